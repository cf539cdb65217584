/*
 * dae_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, fp32) of the
 * denoising-autoencoder scoring path of hojinYang/spotify_recSys_challenge_2018 in the
 * CANONICAL SUMMATION ORDER the HIP kernels are specified to reproduce bit-for-bit.
 *
 * PARITY UNPINNED at the TensorFlow boundary: the reference's arithmetic lives in TensorFlow 1.x
 * ("v1.5.0", readme.md:33; not vendored, not installable here) and the reference ships no tests,
 * golden vectors or fixtures for this path.  This file is pinned instead
 * against oracle/dae_numpy.py (the literal dense restatement of models/DAEs.py) within fp32
 * re-association tolerance, and the pure-Python pieces of the reference that DO import
 * (utils/metrics.py get_r_precision, utils/data_reader.py layouts) are pinned by the golden
 * fixtures under tests/golden/ generated from the real reference (tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (spotify_recsys_challenge_2018_amd/) never does.
 *
 * Each function cites the reference lines it follows (paths relative to /root/reference).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; every fused multiply-add is an explicit fmaf).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * canonical scalar functions
 * ---------------------------------------------------------------------------------------------- */

static inline float as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t as_u32(float f)   { uint32_t u; memcpy(&u, &f, 4); return u; }

/* Canonical sigmoid 1/(1+exp(-x)) -- tf.nn.sigmoid at DAEs.py:67,75,143.  Built only from
 * IEEE-exact fp32 operations (mul, fma, rint, divide, exponent insertion) so that the GPU and
 * this file agree bit-for-bit.  |error| vs the real sigmoid <= ~2 ulp. */
float orc_sigmoidf(float x)
{
    float t = -x;
    t = fminf(fmaxf(t, -87.0f), 87.0f);
    float n = rintf(t * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, t);          /* ln2 high part (exact in 16 bits) */
    r = fmaf(n, -1.42860682030941723e-6f, r);            /* ln2 low part                    */
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    float s = as_float((uint32_t)((int32_t)n + 127) << 23);   /* 2^n, n in [-126,126] */
    float e = p * s;
    return 1.0f / (1.0f + e);
}

static inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

/* Counter-based uniform in [0,1) standing in for TF's random_uniform inside tf.nn.dropout
 * (DAEs.py:40, :68).  TF's generator cannot be reproduced; the DISTRIBUTION (Bernoulli keep with
 * 1/keep_prob rescale) is what the reference specifies.  stream 0 = input dropout, 1 = hidden. */
float orc_uniform(uint32_t seed, uint32_t stream, uint32_t row, uint32_t col)
{
    uint32_t x = mix32(seed + 0x9E3779B9U * (stream + 1U));
    x = mix32(x ^ row);
    x = mix32(x ^ col);
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

/* Order-preserving map fp32 -> u32 (bigger key = bigger float); -0 is canonicalised to +0. */
uint32_t orc_okey(float f)
{
    uint32_t u = as_u32(f);
    if ((u << 1) == 0) u = 0;
    return (u & 0x80000000U) ? ~u : (u | 0x80000000U);
}

/* ------------------------------------------------------------------------------------------------
 * encode -- DAEs.py:33-35 (sparse->dense, here CSR already de-duplicated, last-wins, by the host
 * shim), :40 input dropout, :41-42 row normalise (divide BEFORE the matmul, as the reference),
 * :66-68 encoder matmul + bias + sigmoid + hidden dropout.
 * Canonical order: non-zeros in ascending column order, one fmaf per non-zero per hidden unit.
 * ---------------------------------------------------------------------------------------------- */
void orc_encode(const int32_t* row_ptr, const int32_t* col, const float* val,
                const float* W_enc, const float* b_enc, int V, int H, int B,
                float ikp, float kp, uint32_t seed, float* h_out)
{
    (void)V;
    float* acc = (float*)malloc(sizeof(float) * (size_t)H);
    for (int r = 0; r < B; ++r) {
        const int beg = row_ptr[r], end = row_ptr[r + 1];
        const int nnz = end - beg;
        float* xd = (float*)malloc(sizeof(float) * (size_t)(nnz > 0 ? nnz : 1));
        float s = 0.0f;
        for (int i = 0; i < nnz; ++i) {
            float x = val[beg + i];
            if (ikp < 1.0f) {                                   /* tf.nn.dropout: x/kp*floor(kp+u) */
                float u = orc_uniform(seed, 0U, (uint32_t)r, (uint32_t)col[beg + i]);
                x = (x / ikp) * floorf(ikp + u);
            }
            xd[i] = x;
            s += x;                                             /* reduce_sum, DAEs.py:41 */
        }
        const float denom = s + 1e-10f;                         /* DAEs.py:42 */
        for (int j = 0; j < H; ++j) acc[j] = 0.0f;
        for (int i = 0; i < nnz; ++i) {
            const float w = xd[i] / denom;
            const float* wrow = W_enc + (size_t)col[beg + i] * (size_t)H;
            for (int j = 0; j < H; ++j) acc[j] = fmaf(w, wrow[j], acc[j]);
        }
        for (int j = 0; j < H; ++j) {
            float hv = orc_sigmoidf(acc[j] + b_enc[j]);         /* DAEs.py:66-67 */
            if (kp < 1.0f) {                                    /* DAEs.py:68 */
                float u = orc_uniform(seed, 1U, (uint32_t)r, (uint32_t)j);
                hv = (hv / kp) * floorf(kp + u);
            }
            h_out[(size_t)r * H + j] = hv;
        }
        free(xd);
    }
    free(acc);
}

/* ------------------------------------------------------------------------------------------------
 * decode -- DAEs.py:75-76 (tied) / :143-144 (untied): logits = h . W_dec^T + b_dec.
 * Canonical order: acc = +0; for k = 0..H-1: acc = fmaf(h[k], W[c,k], acc); logit = acc + b[c].
 * (This is exactly what v_mfma_f32_32x32x2_f32 computes.)  apply_sigmoid -> y_pred.
 * ---------------------------------------------------------------------------------------------- */
void orc_decode(const float* h, const float* W_dec, const float* b_dec, int H, int B,
                int col_lo, int col_hi, int apply_sigmoid, float* out, int64_t ld)
{
    for (int r = 0; r < B; ++r) {
        const float* hr = h + (size_t)r * H;
        for (int c = col_lo; c < col_hi; ++c) {
            const float* w = W_dec + (size_t)c * H;
            float acc = 0.0f;
            for (int k = 0; k < H; ++k) acc = fmaf(hr[k], w[k], acc);
            float z = acc + b_dec[c];
            out[(size_t)r * ld + (c - col_lo)] = apply_sigmoid ? orc_sigmoidf(z) : z;
        }
    }
}

/* bf16 variant (BASELINE.json config 5): operands rounded to bf16 (round-to-nearest-even),
 * products accumulated in fp32 in ascending k.  The GPU's bf16 MFMA accumulates 16 products per
 * instruction with unspecified internal order, so this one is a TOLERANCE reference only. */
static inline float bf16_round(float f)
{
    uint32_t u = as_u32(f);
    uint32_t lsb = (u >> 16) & 1U;
    u += 0x7FFFU + lsb;
    u &= 0xFFFF0000U;
    return as_float(u);
}
void orc_decode_bf16(const float* h, const float* W_dec, const float* b_dec, int H, int B,
                     int col_lo, int col_hi, int apply_sigmoid, float* out, int64_t ld)
{
    float* hb = (float*)malloc(sizeof(float) * (size_t)H);
    for (int r = 0; r < B; ++r) {
        for (int k = 0; k < H; ++k) hb[k] = bf16_round(h[(size_t)r * H + k]);
        for (int c = col_lo; c < col_hi; ++c) {
            const float* w = W_dec + (size_t)c * H;
            float acc = 0.0f;
            for (int k = 0; k < H; ++k) acc += hb[k] * bf16_round(w[k]);
            float z = acc + b_dec[c];
            out[(size_t)r * ld + (c - col_lo)] = apply_sigmoid ? orc_sigmoidf(z) : z;
        }
    }
    free(hb);
}

/* ------------------------------------------------------------------------------------------------
 * rank -- main_challenge.py:28-36 / metrics.py:59-68: argsort descending, remove the seed tracks,
 * keep k (=500).  numpy's argsort tie order is unspecified; the canonical rule is
 * (logit desc, column index asc).  Ranking on the LOGIT is ranking on the EXACT sigmoid; it REFINES -- it does not
 * equal -- the order of fp32 sigmoid outputs the reference sorts: equal outputs (a saturated plateau at 1.0f) are
 * told apart by their logits, and orc_sigmoidf above is monotone only to one unit in the last place (920 one-ulp
 * inversions in [-88, 88]), so at such a pair the two orders disagree.  tests/test_gpu_rank_seam.py pins both.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t key; int32_t idx; float logit; } cand_t;

static int cand_cmp(const void* a, const void* b)
{
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->key != y->key) return x->key > y->key ? -1 : 1;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return 0;
}

/* One row.  logits[0..n) are columns col_base..col_base+n; seeds = GLOBAL column ids to drop
 * (any order, duplicates allowed, ids outside the range ignored).  Writes k entries; missing
 * entries get idx -1 / score -inf.  out_kind 0 = canonical sigmoid, 1 = logit. */
void orc_topk_row(const float* logits, int n, int col_base,
                  const int32_t* seeds, int nseeds, int k, int out_kind,
                  float* out_score, int32_t* out_idx)
{
    unsigned char* drop = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int i = 0; i < nseeds; ++i) {
        int64_t p = (int64_t)seeds[i] - col_base;
        if (p >= 0 && p < n) drop[p] = 1;
    }
    cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (drop[i]) continue;
        if (logits[i] == -INFINITY) continue;
        c[m].key = orc_okey(logits[i]); c[m].idx = col_base + i; c[m].logit = logits[i]; ++m;
    }
    qsort(c, (size_t)m, sizeof(cand_t), cand_cmp);
    for (int i = 0; i < k; ++i) {
        if (i < m) {
            out_idx[i] = c[i].idx;
            out_score[i] = out_kind ? c[i].logit : orc_sigmoidf(c[i].logit);
        } else {
            out_idx[i] = -1;
            out_score[i] = -INFINITY;
        }
    }
    free(c); free(drop);
}

void orc_topk(const float* logits, int64_t ld, int B, int n, int col_base,
              const int32_t* seed_row_ptr, const int32_t* seed_col, int k, int out_kind,
              float* out_score, int32_t* out_idx)
{
    for (int r = 0; r < B; ++r) {
        const int32_t* s = seed_col ? seed_col + seed_row_ptr[r] : NULL;
        int ns = seed_col ? seed_row_ptr[r + 1] - seed_row_ptr[r] : 0;
        orc_topk_row(logits + (size_t)r * ld, n, col_base, s, ns, k, out_kind,
                     out_score + (size_t)r * k, out_idx + (size_t)r * k);
    }
}

/* Merge G shard lists [G,B,k] of (logit, idx) into the global top-k. */
void orc_topk_merge(int G, int B, int k, const float* cand_logit, const int32_t* cand_idx,
                    int out_kind, float* out_score, int32_t* out_idx)
{
    cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)G * (size_t)k);
    for (int r = 0; r < B; ++r) {
        int m = 0;
        for (int g = 0; g < G; ++g)
            for (int i = 0; i < k; ++i) {
                size_t o = ((size_t)g * B + r) * k + i;
                if (cand_idx[o] < 0) continue;
                c[m].key = orc_okey(cand_logit[o]); c[m].idx = cand_idx[o];
                c[m].logit = cand_logit[o]; ++m;
            }
        qsort(c, (size_t)m, sizeof(cand_t), cand_cmp);
        for (int i = 0; i < k; ++i) {
            size_t o = (size_t)r * k + i;
            if (i < m) {
                out_idx[o] = c[i].idx;
                out_score[o] = out_kind ? c[i].logit : orc_sigmoidf(c[i].logit);
            } else { out_idx[o] = -1; out_score[o] = -INFINITY; }
        }
    }
    free(c);
}

/* ------------------------------------------------------------------------------------------------
 * whole scoring path for one batch: encode -> decode(track columns) -> rank.  Used by bench.py's
 * cpu_baseline leg ("port", 1 thread) and by smoke().  Returns nothing; scratch is internal.
 * ---------------------------------------------------------------------------------------------- */
void orc_score_batch(const int32_t* row_ptr, const int32_t* col, const float* val,
                     const float* W_enc, const float* b_enc,
                     const float* W_dec, const float* b_dec,
                     int V, int H, int B, int n_cols_decoded, int n_tracks,
                     const int32_t* seed_row_ptr, const int32_t* seed_col, int k,
                     float* out_score, int32_t* out_idx)
{
    float* h = (float*)malloc(sizeof(float) * (size_t)B * H);
    float* z = (float*)malloc(sizeof(float) * (size_t)B * n_cols_decoded);
    orc_encode(row_ptr, col, val, W_enc, b_enc, V, H, B, 1.0f, 1.0f, 0U, h);
    orc_decode(h, W_dec, b_dec, H, B, 0, n_cols_decoded, 0, z, n_cols_decoded);
    orc_topk(z, n_cols_decoded, B, n_tracks < n_cols_decoded ? n_tracks : n_cols_decoded, 0,
             seed_row_ptr, seed_col, k, 0, out_score, out_idx);
    free(h); free(z);
}
