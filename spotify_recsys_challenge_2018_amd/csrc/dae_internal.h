// dae_internal.h -- shared between the translation units of libdae_hip.so (gfx950 only).
// Context object, error plumbing, canonical scalar device functions, kernel launcher prototypes.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/dae_hip.h"

// ---------------------------------------------------------------------------------------------
// geometry constants of the decode path (see DESIGN.md "HBM layout")
// ---------------------------------------------------------------------------------------------
constexpr int DAE_TITLE_MAX_SIZES = 8;   // filter sizes a title scorer may have (title.hip's kernels, the table, dae_pipeline_create_titled)
constexpr int DAE_VT = 32;        // vocabulary columns per wave tile (MFMA M = 32)
constexpr int DAE_KG = 8;         // k values per packed group (4 MFMA 32x32x2 steps)
constexpr int DAE_HPAD = 32;      // hidden size is zero-padded to a multiple of this
constexpr int DAE_MAX_K = 1024;   // largest top-k supported
constexpr int DAE_NUM_CU = 256;   // MI355X
constexpr int DAE_NUM_XCD = 8;
constexpr int DAE_REFINED_CAP = 4096;   // survivors per row the exact mode's compact lists hold (more: refined in place)
constexpr size_t DAE_GUARD_BYTES = 2 * sizeof(int);   // a context's guard words {violations, a violating column}

struct dae_buf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct dae_packed {            // one prepacked decoder image
    bool valid = false;
    bool borrowed = false;      // W / bias / bias16* / eps / W32 belong to ANOTHER context (dae_share_decoder): never freed here
    int V = 0, H = 0, Hp = 0;   // Hp = H padded to DAE_HPAD
    int col_lo = 0, col_hi = 0;
    int ntiles = 0;            // ceil((col_hi-col_lo)/32)
    dae_buf W;                 // fp32: [ntiles][Hp/8][64 lanes][4]   bf16: see decode_bf16.hip
    dae_buf bias;              // [ntiles*32] fp32, zero padded
    dae_buf bias16;            // bf16 image: [ntiles][64] uint4 bias fragments (decode_f32.hip)
    // tiles ordered by the largest bias among their rankable columns, descending (the threshold
    // sample of the fused path takes the head of this list); rebuilt when the image or the number
    // of rankable columns changes
    dae_buf order;             // [ntiles] int32
    dae_buf ident;             // [ntiles] int32: 0, 1, 2, ... (the tile list of "all tiles")
    int order_nrank = -1;      // rankable columns the order was built for (-1: none)
    int order_nsamp = -1;
    long long order_gen = 0;   // process-wide stamp of the last rebuild of `order` (what a context's band list was cut from)
    // DAE_DTYPE_BF16_EXACT (bf16 image only; decode_f32.hip exact_bounds_kernel): per column c a rigorous bound
    // eps_c >= |z32(r, c) - z16(r, c)| for every hidden row with entries in [0, 1], the bias fragments of
    // b - eps (phase A: lower bounds of the fp32 logits) and b + eps (filter: upper bounds), and a row-major fp32
    // copy of the image's decoder rows for the exact re-scoring of the survivors (topk.hip ExactSrc)
    bool exact = false;
    dae_buf eps;               // [ntiles*32] fp32, zero padded; then one more float: the maximum (eps_max)
    dae_buf bias16_lo, bias16_hi;   // [ntiles][64] uint4, as bias16
    dae_buf W32;               // [col_hi - col_lo][H] fp32 row-major
    // the same image as the TITLE side of the exact title mix (mixexact.hip): hidden rows with entries in [-F_r, F_r],
    // |z32 - z16| <= F_r alpha_c + beta_c; the bias fragments carry b -+ beta in k-slots 0..2 and -+alpha (bf16) in slot 3
    dae_buf mix_alpha, mix_beta;    // [ntiles*32] fp32 (alpha: the bf16 value the fragments hold)
    dae_buf mix16_lo, mix16_hi;     // [ntiles][64] uint4
};

struct dae_rowgeom {        // how B rows are cut into row groups for the decode kernels
    int R_TILE;             // rows per group: 128, 64 or 32 (LDS-resident h tile)
    int n_rg;               // ceil(B / R_TILE)
    int Bpad;               // n_rg * R_TILE
    int nb_rg;              // thread blocks per row group
    int grid;               // n_rg * nb_rg
    int waves;              // waves per workgroup (4 or 8)
};

struct dae_topk_state {     // what the second half of a fused scoring call needs from the first (api.hip topk_phase_a / _b)
    bool valid = false;
    const dae_packed* pk = nullptr;
    dae_rowgeom g{};
    int B = 0, k = 0, dtype = 0, S = 0, n_samp = 0, n_other = 0, n_valid_col = 0, nrank = 0;
    bool exact = false, fused = false, mixed = false, whole_b = false;
    int64_t ld_s = 0;
    const int* order = nullptr;
    int* sample_cnt = nullptr;
    // dae_score_topk_begin / _finish: arguments of the pending call
    int pend_H = 0, pend_dtype = 0, pend_n_tracks = 0;
    const float* pend_h32 = nullptr;
};

struct dae_ctx {
    int device = 0;
    // per-context (= per-device) one-time setup done so far, e.g. hipFuncSetAttribute of a kernel: a process-wide
    // `static bool` would skip it for the second device of a multi-device process
    std::unordered_set<const void*> first_use_done;
    hipStream_t stream = nullptr;
    std::string err;
    size_t scratch_total = 0;

    dae_packed pk_f32, pk_bf16;

    // scratch (grown lazily, never shrunk)
    dae_buf h_packed;          // [n_rg][Hp/8][RB][64][4] fp32 (or bf16 image)
    dae_buf h_packed16;        // bf16 image of the hidden tile [n_rg][Hp/16][RB][64][8 bf16]
    dae_buf h_scratch;         // [B,H] fp32 hidden activations when the caller does not keep them
    long long h_geom_key = -1; // (B, H, R_TILE) whose pad region of h_packed is known to be zero
    void* h_geom_ptr = nullptr;
    long long h16_geom_key = -1;   // the same for the bf16 image h_packed16 when the encode writes it directly
    void* h16_geom_ptr = nullptr;
    dae_buf sample;            // phase-A dense logits [Bpad][n_sample_cols]
    dae_buf gmax;              // phase-A group maxima [Bpad][4 * n_sample_tiles]
    dae_buf tau;               // [Bpad] fp32
    dae_buf sample_top;        // [Bpad][k] (logit, idx) pairs
    dae_buf cand;              // [nb_rg][Bpad][cap] pairs
    dae_buf cand_cnt;          // [nb_rg][Bpad] int
    dae_buf dense_tmp;         // unfused fallback logits
    dae_buf train_a, train_b, train_c, train_d;
    float* arm_m = nullptr; float* arm_v = nullptr; float arm_alpha = 0.f, arm_b1 = 0.f, arm_b2 = 0.f, arm_eps = 0.f;  // dae_arm_decoder_adam
    const float* mixT = nullptr; int64_t mix_ld = 0; const float* mix_w = nullptr; int mix_ncols = 0;   // dae_set_score_mix (caller-owned)
    hipEvent_t gate_wait = nullptr, gate_record = nullptr;   // dae_set_decode_gate (caller-owned events)
    int enc_grad_prezeroed = 0;        // untied gW_enc is all-zero on entry (dae_adam_rows_apply re-zeroes what it reads)
    int adam_t = 0; float adam_b1 = 0.f, adam_b2 = 0.f, adam_b1p = 1.f, adam_b2p = 1.f;   // running beta powers of dae_adam_alpha
    int train_dtype = DAE_DTYPE_F32;   // arithmetic of the training forward GEMM (dae_set_train_dtype)
    dae_buf csr_tmp;           // COO -> CSR scratch (csr.hip)
    dae_topk_state tk;
    int overlap_hint = 0;      // dae_set_overlap_hint: other batches are in flight on other streams -> kernel shapes that share CUs
    dae_buf row_bad;           // DAE_DTYPE_BF16_EXACT via dae_decode_topk: [Bpad] int32, 1 = the caller's hidden row leaves [0, 1]
    dae_buf refined;           // DAE_DTYPE_BF16_EXACT: [Bpad][DAE_REFINED_CAP] (fp32 logit, column) pairs + [Bpad] counts (refine.hip)
    dae_buf refstat; int refstat_rows = 0;   // DAE_DTYPE_BF16_EXACT: [rows][2] {candidates, recomputed} of the last refine launch
    dae_buf guard;             // DAE_DTYPE_BF16_EXACT: {violations of the bound seen by the refine launches, a violating column}
    float exact_margin = 1.0f; // dae_set_exact_margin: factor on every eps_c at the next exact prepack
    int margin_lo = 0, margin_hi = 0; float margin_scale = 1.0f;   // dae_set_exact_margin_range: columns [lo, hi) take this factor instead
    // the audit of dropped columns (audit.hip): every audit_every-th exact scoring launch checks audit_tiles random tiles
    int audit_every = 64, audit_tiles = 16;
    uint64_t audit_seq = 0, audits_run = 0;
    dae_buf title_y1;          // dae_title_rank (fp32 / bf16): the DAE term of the track columns, transposed
    dae_buf audit_stat;        // {elements checked, violations} | the sampled tile ids (allocated once)
    dae_buf audit;             // upper bounds of the sampled tiles [Bpad][tiles * 32]
    dae_buf mix_fhat;          // dae_mix_topk_exact: [Bpad] bf16 bits of the rows' feature bounds
    dae_buf title_scratch;     // dae_title_score_exact: CSR, seed lists, hidden rows, features, mixing weights of the launch
    // dae_title_prepack_features (title.hip): the convolutions of a FROZEN title scorer as a table over (filter size, offset,
    // character) -- which variables it was built from
    dae_buf title_tab; const float* ttab_emb = nullptr; const float* ttab_w = nullptr; int ttab_nchar = 0, ttab_E = 0, ttab_F = 0,
        ttab_nsizes = 0; int ttab_fs[DAE_TITLE_MAX_SIZES] = {0};
    // the bias-ordered tile list with its sample RE-DEALT for a launch geometry (dae_launch_tile_band): which order it was cut from
    dae_buf tile_band; long long band_gen = -1; int band_nsamp = 0, band_nbrg = 0, band_waves = 0;

    // profiling of the dominant kernel
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;   // pairs (start, stop)
    size_t prof_used = 0;
    bool prof_armed = false;           // the next decode launch takes prof_ev[prof_used], [prof_used+1]
    std::string prof_kernel;           // symbol of the kernel the last armed pair bracketed (dae_profile_kernel)
};

extern thread_local std::string g_dae_create_err;

int dae_fail(dae_ctx* ctx, int code, const char* fmt, ...);
int dae_reserve(dae_ctx* ctx, dae_buf& b, size_t bytes);

#define DAE_HIP_CHECK(ctx, expr)                                                              \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return dae_fail((ctx), DAE_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                            hipGetErrorString(_e), __FILE__, __LINE__);                       \
    } while (0)

#define DAE_CHECK_LAUNCH(ctx, name)                                                           \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess)                                                                 \
            return dae_fail((ctx), DAE_ERR_HIP, "launch of %s failed: %s", name,              \
                            hipGetErrorString(_e));                                           \
    } while (0)

static inline int dae_round_up(int x, int m) { return (x + m - 1) / m * m; }

// Experiment switches (A/B variants, stage bisection: DESIGN.md section 6b) exist only in builds made with
// -DDAE_EXPERIMENTS (`DAE_EXPERIMENTS=1 python -m spotify_recsys_challenge_2018_amd.build --force`): the default
// library reads no environment variable and its kernels carry no early-outs.
#ifdef DAE_EXPERIMENTS
static inline const char* dae_exp_env(const char* name) { return getenv(name); }
#define DAE_EXP_ON(x) (x)
#else
static inline const char* dae_exp_env(const char*) { return nullptr; }
#define DAE_EXP_ON(x) false
#endif

// ---------------------------------------------------------------------------------------------
// canonical scalar device functions -- the SPECIFICATION is DESIGN.md "canonical order"; the
// oracle (oracle/dae_oracle.c) restates the same arithmetic independently.  Built with
// -ffp-contract=off: every fused multiply-add below is explicit.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dae_sigmoidf(float x)
{
    float t = -x;
    t = fminf(fmaxf(t, -87.0f), 87.0f);
    float n = rintf(t * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, t);
    r = fmaf(n, -1.42860682030941723e-6f, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    float s = __uint_as_float((uint32_t)((int)n + 127) << 23);
    float e = p * s;
    return 1.0f / (1.0f + e);
}

// fp32 -> bf16, round to nearest even (oracle: bf16_round)
__device__ __forceinline__ unsigned dae_bf16_rne(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

__device__ __forceinline__ uint32_t dae_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ float dae_uniform(uint32_t seed, uint32_t stream, uint32_t row,
                                             uint32_t col)
{
    uint32_t x = dae_mix32(seed + 0x9E3779B9U * (stream + 1U));
    x = dae_mix32(x ^ row);
    x = dae_mix32(x ^ col);
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// order-preserving fp32 -> u32 (bigger key = bigger float, -0 == +0)
__device__ __forceinline__ uint32_t dae_okey(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u << 1) == 0) u = 0;
    return (u & 0x80000000U) ? ~u : (u | 0x80000000U);
}
__device__ __forceinline__ float dae_okey_inv(uint32_t k)
{
    uint32_t u = (k & 0x80000000U) ? (k & 0x7FFFFFFFU) : ~k;
    return __uint_as_float(u);
}
constexpr uint32_t DAE_KEY_NEG_INF = 0x007FFFFFU;   // dae_okey(-inf): "absent" marker

// ---------------------------------------------------------------------------------------------
// launchers (each defined next to its kernels)
// ---------------------------------------------------------------------------------------------
dae_rowgeom dae_row_geometry(int B, int Hp);
dae_rowgeom dae_row_geometry_bf16(int B, int Hp);
int dae_launch_prepack_bf16(dae_ctx* ctx, const float* W, const float* b, int V, int H,
                            int col_lo, int col_hi, int exact = 0);
// row_bad (nullable): [g.Bpad] int32, 1 where a row of h has an entry outside [0, 1] (the exact mode's precondition)
int dae_launch_pack_h_bf16(dae_ctx* ctx, const float* h, int B, int H, const dae_rowgeom& g, int* row_bad = nullptr);

// encode.hip
int dae_launch_encode(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val,
                      const float* W_enc, const float* b_enc, int V, int H, int B,
                      float ikp, float kp, uint32_t seed, float* h_out,
                      float* h_packed, int G, int RB, float* sg_out = nullptr,
                      float* xhat_out = nullptr, unsigned short* h_packed16 = nullptr, int NS = 0);

// decode_f32.hip
int dae_launch_prepack_f32(dae_ctx* ctx, const float* W, const float* b, int V, int H,
                           int col_lo, int col_hi);
int dae_launch_pack_h(dae_ctx* ctx, const float* h, int B, int H, const dae_rowgeom& g);
int dae_launch_tile_iota(dae_ctx* ctx, int* dst, int ntiles);      // dst[i] = i
// (re)build pk.order for `nrank` rankable columns; the first n_samp entries are the threshold sample
int dae_filter_block_tiles(const dae_rowgeom& g, int n_items, int dtype, int Hp, bool mixed = false);
int dae_launch_tile_order(dae_ctx* ctx, dae_packed& pk, int nrank, int n_samp, int S);
// band[0 .. n_samp): the sample of `order` dealt to the phase-A launch's slots so that the tiles ONE workgroup decodes in a round
// (item = round * nb_rg * waves + wave * nb_rg + bir) come from `waves` different popularity bands; band[n_samp ..) = order
int dae_launch_tile_band(dae_ctx* ctx, const int* order, int ntiles, int n_samp, int nb_rg, int waves, int* band);
bool dae_sample_wave_groups(const dae_rowgeom& g, int Hp, int n_samp);

struct dae_tileset {        // which wave tiles a decode launch walks
    int n_items;            // number of tiles in the set
    int stride;             // S
    int mode;               // informational: 0 = all tiles (identity list), 3 = ordered list
    const int* list;        // tile of item i; never null
};
// dense epilogue: out[row*ld + item*32 + vl] (item = position of the tile in the set)
// gmax (nullable): [B][ld_gmax] maxima of every lane's two 8-column groups, 4 per (row, item) -- the threshold
// sample's input to dae_launch_tau_select
int dae_launch_decode_dense_f32(dae_ctx* ctx, const dae_rowgeom& g, int B, const dae_tileset& ts,
                                int apply_sigmoid, int mask_from_col, float* out, int64_t ld,
                                int fill_pad, int dtype = DAE_DTYPE_F32, float* gmax = nullptr,
                                int64_t ld_gmax = 0, int gmax_per_wave = 0, int bias_sel = 0);
// gmax_per_wave: 0 = one maximum per (workgroup, round, position) over the workgroup's waves; 1 = every wave slot's value (small
// samples); 3 = per-WAVE groups over all of a wave's tiles, 8 wave slots per workgroup (dae_sample_wave_groups: ld_gmax = 8 nb_rg 32);
// 4 = the same with waves w and w + 4 sharing a group (ld_gmax = 4 nb_rg 32)
// filter epilogue: append (logit, global col) with logit >= tau[row] and col < n_valid_col
int dae_launch_decode_filter_f32(dae_ctx* ctx, const dae_rowgeom& g, int B, const dae_tileset& ts,
                                 const float* tau, int n_valid_col, uint2* cand, int* cand_cnt,
                                 int cap, int dtype = DAE_DTYPE_F32, int bias_sel = 0);
// bias_sel (bf16 image prepacked with DAE_DTYPE_BF16_EXACT): 0 = b, 1 = b - eps (lower bounds), 2 = b + eps (upper bounds)

int dae_launch_decode_scaled_T(dae_ctx* ctx, const dae_rowgeom& g, int B, const dae_tileset& ts, const float* row_scale,
                               float* outT, int64_t ldT, int dtype = DAE_DTYPE_F32);

// training forward: all tiles; the epilogue takes every element as a negative (target 0), writes dL/dz
// TRANSPOSED ([ncols, ldT]: what both backward GEMMs read) and one loss partial per workgroup
// (DAEs.py:98-100); the positives are redone afterwards by train.hip's loss_fixup_kernel.
int dae_launch_decode_loss_f32(dae_ctx* ctx, const dae_rowgeom& g, int B, float inv_n_batch,
                               float* dzT, int64_t ldT, float* loss_part, int dtype = DAE_DTYPE_F32,
                               int dz16 = 0);

int dae_launch_decode_loss_rowmajor(dae_ctx* ctx, const dae_rowgeom& g, int B, int V, int H, const float* W,
                                    const float* bias, const float* h, float inv_n_batch, float* dzT, int64_t ldT,
                                    float* loss_part, int dtype = DAE_DTYPE_F32, int dz16 = 0);

int dae_launch_decode_loss_dh(dae_ctx* ctx, const dae_rowgeom& g, int B, int V, int H, const float* W, const float* bias,
                              const float* h, float inv_n_batch, float* dzT, int64_t ldT, float* loss_part, float* part, int Bpad64);

// train.hip
int dae_train_step_f32(dae_ctx* ctx,
        const int32_t* x_row_ptr, const int32_t* x_col, const float* x_val,
        const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
        const float* W_enc, const float* b_enc, const float* W_dec, const float* b_dec,
        int V, int H, int B, int n_batch, int tied,
        float ikp, float kp, uint32_t seed, float reg_lambda,
        float* gW_enc, float* gb_enc, float* gW_dec, float* gb_dec, float* cost_out);
int dae_train_shard_encode_f32(dae_ctx* ctx, const int32_t* x_row_ptr, const int32_t* x_col,
                               const float* x_val, const float* W_enc_loc, int col_lo, int col_hi,
                               int H, int B, float ikp, uint32_t seed, float* pre_partial);
int dae_train_shard_decode_f32(dae_ctx* ctx, const float* pre, const float* b_enc,
                               const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
                               const float* W_enc_loc, const float* W_dec_loc, const float* b_dec_loc,
                               int col_lo, int col_hi, int H, int B, int n_batch, int tied,
                               float kp, uint32_t seed, float reg_lambda,
                               float* gW_out, float* gb_dec_loc, float* dh_partial, float* cost_partial);
int dae_train_shard_finish_f32(dae_ctx* ctx, const float* dh, const int32_t* x_row_ptr,
                               const int32_t* x_col, const float* x_val,
                               const float* W_enc_loc, const float* b_enc, const float* W_dec_loc,
                               const float* b_dec_loc, int col_lo, int col_hi, int H, int B, int tied,
                               float ikp, float kp, uint32_t seed, float reg_lambda,
                               float* gW_enc_loc, float* gb_enc, float* gW_dec_loc, float* gb_dec_loc);
int dae_launch_grad_w(dae_ctx* ctx, const float* dzT, int64_t ldT, const float* h, int H, int B, int V,
                      float* gW, float* gb);
int dae_launch_grad_h(dae_ctx* ctx, const float* dzT, int64_t ldT, const float* W, int H, int V, int B, float* dh);
inline bool dae_first_use(dae_ctx* ctx, const void* key) { return ctx->first_use_done.insert(key).second; }

int dae_launch_adam_rows(dae_ctx* ctx, int mode, float* param, float* m, float* v, float* grad, int32_t* last,
                         int32_t* mark, float* lr_tab, int n_rows, int row_len, const int32_t* rows,
                         const int32_t* n_listed_dev, int n_listed_max, float lr_t, float beta1, float beta2,
                         float eps, int step);
int dae_launch_adam(dae_ctx* ctx, float* param, float* m, float* v, const float* grad, int64_t n,
                    float lr_t, float beta1, float beta2, float eps);

// csr.hip
int dae_launch_coo_to_csr(dae_ctx* ctx, const int64_t* positions, const float* values, int values_broadcast,
                          int64_t nnz, int n_rows, int n_cols, int32_t* row_ptr, int32_t* col, float* val,
                          int32_t* status);
// a titled launch's intermediates (api.hip dae_title_prepare / dae_title_rank; device pointers, caller-owned)
struct dae_title_bufs {
    int32_t *rp, *col, *srp, *sc;     // CSR of the feed [B + 1], [nnz]; seed lists [B + 1], [nnz]
    float *val, *h, *feat, *wt, *wp;  // values [nnz]; DAE hidden rows [B][H]; title features [B][ld_feat]; mixing weights [B]
};
int dae_title_prepare(dae_ctx* tc, dae_ctx* dc, const int64_t* positions, const float* values, int values_broadcast, int64_t nnz,
                      int B, int V, const float* W_enc, const float* b_enc, int H, const int32_t* titles, int L, const float* emb,
                      int n_char, int E, const float* conv_w, const float* conv_b, const int32_t* filter_sizes, int n_sizes, int F,
                      int ld_feat, const float* titles_use, int n_tracks, const dae_title_bufs& b, int32_t* csr_status);
int dae_title_rank(dae_ctx* tc, dae_ctx* dc, int dtype, int B, int V, int H, int ld_feat, const dae_title_bufs& b, int n_tracks, int k,
                   float* out_score, int32_t* out_idx, int32_t* guard_out);
int dae_launch_coo64_to_csr_seeds(dae_ctx* ctx, const int64_t* positions, const float* values, int values_broadcast,
                                  int64_t nnz, int n_rows, int n_cols, int32_t* row_ptr, int32_t* col, float* val,
                                  int32_t* status, int n_tracks, int32_t* seed_row_ptr, int32_t* seed_col);
int dae_launch_coo32_to_csr_seeds(dae_ctx* ctx, const int32_t* positions, const float* values, int values_broadcast,
                                  int64_t nnz, int n_rows, int n_cols, int32_t* row_ptr, int32_t* col, float* val,
                                  int32_t* status, int n_tracks, int32_t* seed_row_ptr, int32_t* seed_col);

int dae_launch_seeds_from_csr(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, int B, int n_tracks,
                              int32_t* seed_row_ptr, int32_t* seed_col);

// title.hip
int dae_launch_title_features(dae_ctx* ctx, const int32_t* titles, int B, int L, const float* emb, int n_char,
                              int E, const float* conv_w, const float* conv_b, const int32_t* filter_sizes,
                              int n_sizes, int F, float kp, uint32_t seed, float* feat, int64_t ld,
                              int32_t* argmax, float* feat_raw);
int dae_launch_title_table(dae_ctx* ctx, const float* emb, int n_char, int E, const float* conv_w, const int32_t* filter_sizes,
                           int n_sizes, int F);
int dae_launch_row_sums(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val, int B,
                        float ikp, uint32_t seed, float* out);
int dae_launch_mix_weights(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val, int B, float ikp,
                           uint32_t seed, const float* use, float* w_t, float* w_p);
int dae_launch_title_loss_backward(dae_ctx* ctx, const float* zt, int64_t ld_z, const float* dae_score, int64_t ld_d,
                                   const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
                                   const float* w_title, const float* w_playlist, int B, int V, int n_batch,
                                   const float* feat, int ld, const float* Output_WT, float* gOutput_WT,
                                   float* gOutput_b, float* dfeat, float* cost_out);
int dae_launch_title_conv_backward(dae_ctx* ctx, const int32_t* titles, int B, int L, const float* emb, int n_char,
                                   int E, const float* conv_w, const int32_t* filter_sizes, int n_sizes, int F,
                                   const int32_t* argmax, const float* feat_raw, const float* dfeat, int64_t ld,
                                   float kp, uint32_t seed, float* g_emb, float* g_conv_w, float* g_conv_b);
int dae_launch_mix_scores(dae_ctx* ctx, const float* title_score, int64_t ld_t, float* dae_score, int64_t ld_d,
                          const float* w_title, const float* w_playlist, int B, int ncols);

// mixexact.hip (DAE_DTYPE_BF16_EXACT under the title mix)
int dae_launch_mix_title_bounds(dae_ctx* ctx, const float* W, const float* b, int H, int Hp, int col_lo, int col_hi,
                                int ntiles, dae_packed& pk);
int dae_mix_topk_exact_impl(dae_ctx* tc, dae_ctx* dc, const float* feat, int64_t ld_feat, const float* h, int64_t ld_h, int B,
                            const float* w_title, const float* w_playlist, int n_tracks, const int32_t* seed_row_ptr,
                            const int32_t* seed_col, int k, float* out_score, int32_t* out_idx);

// topk.hip
struct dae_dense_src {      // element p of row r = logits[r*ld + p], p in [0,n)
    const float* logits; int64_t ld; int n;
    int col_base;           // global column of tile 0
    int tile_stride;        // column = col_base + (p/32)*32*tile_stride + p%32
    const int* tile_list;   // non-null: column = col_base + tile_list[p/32]*32 + p%32
};
struct dae_pair_group {     // element (seg, r, i) = base[seg*seg_stride + r*row_stride + i]
    const uint2* base; const int* cnt; int64_t seg_stride; int64_t row_stride;
    int64_t cnt_seg_stride; int nseg; int fixed_cnt;
};
struct dae_topk_args {
    int B, k, out_kind;           // out_kind: DAE_OUT_SCORE / DAE_OUT_LOGIT
    int bitmap_base, bitmap_n;    // seed bitmap covers global columns [base, base+n)
    const int32_t* seed_row_ptr; const int32_t* seed_col;
    float* out_score; int32_t* out_idx;   // [B,k]  (may be null)
    uint2* out_pairs;                     // [B,k] (logit bits, idx) (may be null)
    float* out_tau;                       // [B] k-th logit or -inf (may be null)
    int pairs_stride;             // row stride of out_pairs
    int lean, sort_cap;           // set by the launcher (topk.hip): LDS mode, sort buffer keys
    int prefer_small;             // candidate lists: the 256-thread shape (shares a CU with a bf16 filter workgroup)
    const float* row_min;         // nullable: [B] elements with logit < row_min[row] are absent (an exchanged threshold)
};
int dae_launch_topk_dense(dae_ctx* ctx, const dae_dense_src& src, const dae_topk_args& a);
// Threshold of the fused path + the sample's survivors (topk.hip tau_select_kernel):
//   tau[row] = a valid lower bound of the row's k-th largest rankable non-seed logit: the (k + n_seeds(row))-th
//   largest of gmax[row][0..n_g) (-inf = absent) to 16 key bits; -inf when the row holds fewer finite maxima;
//   out_pairs[row * pairs_stride + i], i < out_cnt[row]: the (logit, column) of every dense sample logit
//   samp[row][0..n_s) >= tau (column of element q = col_lo + samp_list[q >> 5] * 32 + (q & 31)), unordered
int dae_launch_tau_select(dae_ctx* ctx, const float* gmax, int64_t ld_g, int n_g, const float* samp, int64_t ld_s,
                          int n_s, const int* samp_list, int col_lo, int B, int k, const int32_t* seed_row_ptr,
                          float* tau, uint2* out_pairs, int64_t pairs_stride, int* out_cnt);
int dae_launch_topk_pairs(dae_ctx* ctx, const dae_pair_group& g0, const dae_pair_group& g1,
                          const dae_topk_args& a);
int dae_launch_topk_soa(dae_ctx* ctx, int G, const float* logit, const int32_t* idx,
                        const dae_topk_args& a);
// DAE_DTYPE_BF16_EXACT (refine.hip): the candidate lists of the bf16 filter launch hold upper bounds u of the fp32
// logits; in place, every candidate that can still be among the k best non-seeds gets its logit RECOMPUTED in fp32 --
// the canonical fmaf chain over k = 0..H-1 of h[row][k] * W32[col - col_lo][k], + bias[col - col_lo] -- and every other
// one -inf (absent for the selection kernel)
struct dae_exact_src {
    const float* h; int64_t ld_h; int H;          // fp32 hidden rows [B][ld_h]
    const float* W32; const float* bias; int col_lo;
    const int* row_bad;                           // nullable: rows flagged 1 return no candidates (idx -1)
    const float* eps_max;                         // device scalar: max over the image's columns of eps_c
    const float* eps;                             // [ncols] the per-column bounds (the guard tests each recomputed survivor)
    int* guard;                                   // nullable: device {violations, a violating column} (dae_exact_guard_read)
};
// out / out_cnt / out_cap (nullable): per row a compact list [out_cap] of the survivors' (fp32 logit, column) pairs and its
// length; rows that fit get it filled and their g1 lists emptied (counts zeroed), the others are refined in place.
// fused (nullable; dae_exact_refine_can_fuse): the launch ENDS the scoring call -- seeds removed, the k best of every row in
// order to fused->out_score / out_idx, exactly what dae_launch_topk_pairs over the refined lists returns -- and the lists it
// leaves behind are not meant to be read
// audit.hip: one audit of the exact scoring launch in progress (n_tiles random ranked tiles x all B rows -> the guard words)
int dae_launch_exact_audit(dae_ctx* ctx, const dae_rowgeom& g, int B, const dae_exact_src& x, int nrank, int n_tiles);
// (pieces the exact title mix's audit shares: mixexact.hip mix_audit) this audit's tile ids + the context's totals; the canonical
// logits of those tiles for rows [0, B) into zout[B][n_tiles * 32]
int dae_audit_pick_tiles(dae_ctx* ctx, int n_rank_tiles, int n_tiles, unsigned long long** stat_out, int** tiles_out);
int dae_launch_audit_chains(dae_ctx* ctx, const float* h, int64_t ld_h, int H, const float* W32, const float* bias, int ncols,
                            int col_bound, int B, const int* tiles, int n_tiles, float* zout);
bool dae_exact_refine_can_fuse(const dae_topk_args& a);
int dae_launch_exact_refine(dae_ctx* ctx, const dae_pair_group& g1, const dae_exact_src& x, int B, int k,
                            const int32_t* seed_row_ptr, uint2* out = nullptr, int* out_cnt = nullptr, int out_cap = 0,
                            int* stat = nullptr, const dae_topk_args* fused = nullptr);
