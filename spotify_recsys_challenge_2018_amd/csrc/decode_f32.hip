// decode_f32.hip -- K2: decoder GEMM logits[r, c] = h[r,:] . W_dec[c,:] + b_dec[c] in EXACT fp32
// on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), reference models/DAEs.py:73-77 (tied) and
// :141-145 (untied), with two epilogues:
//   EPI_DENSE  : store logits / sigmoid scores (the reference's y_pred, main_train.py:66)
//   EPI_FILTER : keep only (logit, column) pairs with logit >= tau[row] -- the fused front half
//                of the top-500 ranking (main_challenge.py:28-36); nothing dense reaches HBM.
//
// Structure (DESIGN.md "decode kernel"):
//   * v_mfma_f32_32x32x2_f32 is bit-for-bit the fmaf chain acc = fma(a_k, b_k, acc) over
//     ascending k, i.e. exactly oracle/dae_oracle.c:orc_decode.  A = W_dec tile (32 vocabulary
//     columns x 2 k), B = h^T (2 k x 32 playlists); D[i = column][j = playlist], so every lane
//     holds 16 different columns of ONE playlist (j = lane & 31) and the per-playlist threshold
//     lives in one register.
//   * the hidden tile of a row group (R_TILE playlists x H, 128 KiB at 128 x 256) is copied into
//     LDS ONCE per persistent workgroup; the main loop has no LDS writes and no barriers.
//   * W_dec is streamed straight from HBM into VGPRs from the prepacked image (one coalesced
//     1 KiB global_load_dwordx4 per wave per 8 k), through a 4-deep register ring; each wave owns
//     whole 32-column tiles, so W_dec is read exactly once per row group.
//   * the fp32 matrix pipe issues one MFMA per 64 cycles per SIMD; with 4 independent
//     accumulators a single wave per SIMD saturates it (MI355X_MICROARCH.md), so the workgroup
//     is 4 waves = one per SIMD, one workgroup per CU.
//   * blockIdx -> (row group, slot) is XCD-aware: the workgroups that walk the SAME column tiles
//     for different row groups sit on the same XCD (blockIdx % 8), so the second..n-th read of a
//     W tile hits that XCD's L2 instead of HBM.
#include <climits>

#include <atomic>
#include "dae_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int DT_F32 = 0;          // v_mfma_f32_32x32x2_f32, exact fp32 (bit-equal to the oracle)
constexpr int DT_BF16 = 1;         // v_mfma_f32_32x32x16_bf16, fp32 accumulate (BASELINE configs[4])

__device__ __forceinline__ bf16x8 as_bf16x8(const uint4 u) { return __builtin_bit_cast(bf16x8, u); }

__device__ __forceinline__ unsigned bf16_rne(float f) { return dae_bf16_rne(f); }

// bf16 kernels take the bias through the matrix pipe: b = e0 + e1 + e2 (three bf16 terms, exact to
// 2^-25 |b|) sits in k-slots 0..2 of an extra A fragment per tile and is multiplied by this B fragment
// of ones, so the accumulators START at the bias: no bias loads or adds in any epilogue, and every
// bf16 kernel produces the same logits for the same (row, column).
__device__ __forceinline__ uint4 bf16_ones_fragment(int hi)
{
    return hi == 0 ? make_uint4(0x3F803F80u, 0x00003F80u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
}

constexpr int EPI_DENSE = 0;
constexpr int EPI_FILTER = 1;
constexpr int EPI_LOSS = 2;        // training: logits -> loss + dL/dz (DAEs.py:98-100)
constexpr int EPI_GMAX = 3;        // EPI_DENSE (raw logits) + cross-wave group maxima: the threshold sample (phase A)

struct DecP {
    const float4* Wp;      // f32: [ntiles][G][64] float4   bf16: [ntiles][G][64] uint4 (8 bf16)
    const float* bias;     // [ntiles*32]
    const uint4* bias16;   // bf16 image only: [ntiles][64] A-operand fragments holding b as 3 bf16 terms
    const float4* hp;      // f32: [n_rg][G][RB][64] float4  bf16: [n_rg][G][RB][64] uint4
    int G;                 // k groups per tile: Hp / 8 (f32, 4 MFMA each) or Hp / 16 (bf16, 1 MFMA)
    int ncols;             // col_hi - col_lo of the prepacked image
    int col_lo;
    int B, n_rg, nb_rg, Bpad;
    dae_tileset ts;
    // dense epilogue
    float* out; int64_t ld; int apply_sigmoid; int mask_from_col; int fill_pad; int vec_ok;
    // EPI_GMAX (threshold sample, phase A of the fused path): besides the dense logits, the maximum over the NW
    // tiles a workgroup decodes in one round of every (row, position in the tile):
    //   gmax[row * ld_gmax + (round * nb_rg + bir) * 32 + c]  = max over waves of z[row][tile(wave)][c]
    // The groups are disjoint sets of columns, so the maxima are distinct elements of the row and their k-th
    // largest is a valid lower bound of the row's k-th largest logit (tau_select_kernel, topk.hip).  The tiles of
    // one workgroup sit nb_rg items apart in the bias-ordered list, i.e. in different popularity bands: with ids
    // = popularity ranks the winners are packed into the first tiles, and a group then holds at most one of them
    // (maxima over neighbouring columns would lose 7 of 8: measured, 4 000 instead of 600 survivors per row).
    float* gmax; int64_t ld_gmax;
    long long* stamps;             // experiments build: stage stamps of workgroup 0 / wave 0 (DAE_DBG_A)
    int gmax_per_wave;             // small samples (vocabulary shards): no cross-wave maximum, slot = (round * n_ws + wave * nb_rg + bir)
    // filter epilogue
    const float* tau; int n_valid_col; uint2* cand; int* cand_cnt; int cap;
    // title mix (models/DAEs.py:180 of the reference: y = title_score * w_title + dae_score * w_playlist):
    //   DAE side, EPI_DENSE: outT[column * ld_outT + row] = sigmoid(z) * row_scale[row] -- the second term, transposed
    //   title side, EPI_GMAX / EPI_FILTER: every value becomes sigmoid(z) * mix_w[row] + mixT[column * mix_ld + row]
    //   before it is stored / compared, i.e. the launch ranks the MIXED score; no [B, V] matrix of either scorer exists
    float* outT; int64_t ld_outT; const float* row_scale;
    const float* mixT; int64_t mix_ld; const float* mix_w; int mix_ncols;   // mixT holds global columns [0, mix_ncols)
    // loss epilogue
    float inv_nb; float* dzT; int64_t ldT; float* loss_part;
    int dz16;                      // dzT holds bf16 (the bf16 backward GEMMs read it as such)
};

__device__ __forceinline__ int tile_of_item(const dae_tileset& ts, int i)
{
    return ts.list[i];          // always a list (the identity for "all tiles"): no branch around a load
}

#ifdef DAE_EXPERIMENTS
#define ASTAMP(i) if (EPI == EPI_GMAX && p.stamps && blockIdx.x == 0 && threadIdx.x == 0) p.stamps[i] = __builtin_readcyclecounter();
#else
#define ASTAMP(i)
#endif

// GT > 0: hidden size known at compile time (G = GT groups of 8 k) -> the k loop is fully
// unrolled, so no loop header sits between the register-ring loads and their use (hipcc drains
// vmcnt to 0 at every loop header; with the loop gone the waits are exact counted vmcnt(3)).
// HALF (phase A of the fp32 fused path, one round of tiles): the workgroup takes HALF a row group of the packed hidden
// image (RB row blocks of its 2 RB) and two workgroups share a CU -- two waves per SIMD, each with half the rows: the one's
// MFMAs run under the other's prologue (hidden tile -> LDS) and epilogue (exchange, sample store), which a single wave per
// SIMD leaves the matrix pipe idle for (28 us for 13.4 us of matrix work).  The exchange slots take the hidden tile's
// place in LDS (dead after the only round), so two workgroups fit: 2 x 64.5 KiB.  Same groups, same chains, same bits.
template <int RB, int EPI, int GT, int NW, int DT, int HALF = 0>
__global__ __launch_bounds__(NW * 64, HALF ? 2 : NW / 4) void decode_f32_kernel(const DecP p)
{
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int j = lane & 31;
    const int G = GT > 0 ? GT : p.G;
    constexpr int R_TILE = RB * 32;

    // XCD-aware block -> (row group, slot in row group)
    const int gs = DAE_NUM_XCD * p.n_rg;
    const int q = blockIdx.x / gs, rem = blockIdx.x % gs;
    const int rg = rem / DAE_NUM_XCD;
    const int bir = q * DAE_NUM_XCD + (rem % DAE_NUM_XCD);

    ASTAMP(0)
    // wave-major slots: consecutive tiles go to different workgroups, so a partial round of tiles is
    // spread over all CUs (and, with two waves per SIMD, over all SIMDs) instead of filling a few
    const int n_ws = p.nb_rg * NW;
    const int item0 = wave * p.nb_rg + bir;
    const bool has0 = item0 < p.ts.n_items;
    // Tile indices come from a list in global memory.  A vector load that the code then waits for
    // drains the WHOLE in-order load queue (s_waitcnt vmcnt(0)), i.e. the W prefetch ring; so the
    // index of a tile is fetched two tiles ahead, before that tile's predecessor issues its W loads --
    // and the first two go out HERE, ahead of the hidden tile's loads, so that the W ring can be started before
    // the workgroup meets (the straight order -- tile, barrier, ids, W -- was one more dependent trip to memory
    // in front of the first MFMA).
    const int tv_cur = tile_of_item(p.ts, has0 ? item0 : 0);
    const int tv_nxt = tile_of_item(p.ts, has0 ? (item0 + n_ws < p.ts.n_items ? item0 + n_ws : item0) : 0);
    // ---- hidden tile of this row group -> LDS, once ------------------------------------------
    const int n_h4 = RB * 64 * G;
    {
        // 8 independent 16 B loads in flight per thread (a load->wait->ds_write chain per element
        // costs one L2 round trip each: ~25k cycles for the 128 KiB tile, measured with SQ_WAIT_ANY)
        const float4* src = p.hp + (HALF ? (size_t)(rg >> 1) * (2 * n_h4) : (size_t)rg * n_h4);
        // HALF: the image is [g][2 RB row blocks][64]; this workgroup's RB blocks of every g
        auto sidx = [&](int i) -> int {
            return HALF ? (i / (RB * 64)) * (2 * RB * 64) + (rg & 1) * (RB * 64) + (i % (RB * 64)) : i;
        };
        constexpr int NT = NW * 64;
        int i = tid;
        for (; i + 7 * NT < n_h4; i += 8 * NT) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[sidx(i + u * NT)];
#pragma unroll
            for (int u = 0; u < 8; ++u) lds4[i + u * NT] = v[u];
        }
        for (; i < n_h4; i += NT) lds4[i] = src[sidx(i)];
    }
    int* lcnt = reinterpret_cast<int*>(lds4 + n_h4);
    float* ltau = reinterpret_cast<float*>(lcnt + R_TILE);
    if (EPI == EPI_FILTER) {
        for (int i = tid; i < R_TILE; i += NW * 64) {
            lcnt[i] = 0;
            ltau[i] = rg * R_TILE + i < p.B ? p.tau[rg * R_TILE + i] : __builtin_inff();
        }
    }

    ASTAMP(1)
    float loss_acc = 0.0f;
    // W stream: the wave's tiles back to back; the register ring always holds the next 4 groups
    // of that stream, so the prefetch runs across tile boundaries (and under the epilogue).
    float4 wb0, wb1, wb2, wb3;
    float4 bA[RB], bB[RB];
    // bf16: one 16-byte load = the A operand of ONE MFMA (K = 16); the ring holds a whole tile
    // (16 steps at hidden = 256): the next tile streams in while this one is multiplied
    constexpr int QR = 16;
    uint4 wq[QR];
    uint4 cb[2][RB];              // hidden fragments: in use / next step
    uint4 bfrag = make_uint4(0u, 0u, 0u, 0u);                     // bias fragment of the wave's next tile
    const uint4 ones = bf16_ones_fragment(hi);
    const uint4* ldsq = reinterpret_cast<const uint4*>(lds4);
    int t_cur = __builtin_amdgcn_readfirstlane(tv_cur), t_nxt = __builtin_amdgcn_readfirstlane(tv_nxt);
    // the ring's first loads: unconditional (a wave without a tile reads the first listed tile and never uses it)
    {
        const float4* w0 = p.Wp + (size_t)t_cur * G * 64 + lane;
        if (DT == DT_F32) {
            wb0 = w0[0]; wb1 = w0[64]; wb2 = w0[128]; wb3 = w0[192];
        } else {
            bfrag = p.bias16[(size_t)t_cur * 64 + lane];
            const uint4* q0 = reinterpret_cast<const uint4*>(w0);
#pragma unroll
            for (int u = 0; u < QR; ++u) wq[u] = q0[(size_t)(u < G ? u : G - 1) * 64];
        }
    }
    __syncthreads();

    float tau_r[RB];
    if (EPI == EPI_FILTER) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) tau_r[rb] = ltau[rb * 32 + j];
    }
    if (DT == DT_F32) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) bA[rb] = lds4[rb * 64 + lane];
    } else {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) cb[0][rb] = ldsq[rb * 64 + lane];
    }

    // EPI_GMAX: one exchange per round of tiles, joined by EVERY wave of the workgroup (a wave without a tile in
    // the round contributes -inf): row block by row block through 16 B x 256 slots per wave behind the hidden
    // tile -- wave w writes its masked logits as float4 (slot = (quad, half, playlist): conflict-free), thread
    // (quad, half, playlist) takes the maximum over the waves and stores 4 maxima of its playlist's row
    auto gmax_round = [&](bool has, int round, const f32x16* accv, const float4* bqv, int tcol0v) {
        // HALF: the slots ARE the hidden tile's LDS (one round only: every wave is past its k loop at the first barrier)
        float4* xl = HALF ? lds4 : reinterpret_cast<float4*>(lcnt + R_TILE);
        if (p.gmax_per_wave) {
            // a sample too small for groups of NW (a vocabulary shard: 61 tiles for 64 wave slots): every element is
            // its own "group" -- the wave stores its masked logits, -inf where it had no tile this round
            const size_t slot = ((size_t)round * p.nb_rg * NW + (size_t)wave * p.nb_rg + bir) * 32;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int row = rg * R_TILE + rb * 32 + j;
                if (row >= p.B) continue;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    float4 v = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
                    if (has) {
                        const int lc = tcol0v + 8 * qd;
                        float z[4] = {accv[rb][4 * qd + 0] + bqv[qd].x, accv[rb][4 * qd + 1] + bqv[qd].y,
                                      accv[rb][4 * qd + 2] + bqv[qd].z, accv[rb][4 * qd + 3] + bqv[qd].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (p.col_lo + lc + e >= p.mask_from_col || lc + e >= p.ncols) z[e] = -__builtin_inff();
                        v = make_float4(z[0], z[1], z[2], z[3]);
                    }
                    *reinterpret_cast<float4*>(p.gmax + (size_t)row * p.ld_gmax + slot + 4 * hi + 8 * qd) = v;
                }
            }
            return;
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            __syncthreads();                                     // the previous row block's slots were read
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                float4 v = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
                if (has) {
                    const int lc = tcol0v + 8 * qd;
                    float z[4] = {accv[rb][4 * qd + 0] + bqv[qd].x, accv[rb][4 * qd + 1] + bqv[qd].y,
                                  accv[rb][4 * qd + 2] + bqv[qd].z, accv[rb][4 * qd + 3] + bqv[qd].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (p.col_lo + lc + e >= p.mask_from_col || lc + e >= p.ncols) z[e] = -__builtin_inff();
                    v = make_float4(z[0], z[1], z[2], z[3]);
                }
                xl[wave * 256 + (qd * 2 + hi) * 32 + j] = v;
            }
            __syncthreads();
            for (int sl = tid; sl < 256; sl += NW * 64) {
                float4 m = xl[sl];
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    const float4 o = xl[w * 256 + sl];
                    m = make_float4(fmaxf(m.x, o.x), fmaxf(m.y, o.y), fmaxf(m.z, o.z), fmaxf(m.w, o.w));
                }
                const int row = rg * R_TILE + rb * 32 + (sl & 31);
                if (row < p.B)
                    *reinterpret_cast<float4*>(p.gmax + (size_t)row * p.ld_gmax +
                                               ((size_t)round * p.nb_rg + bir) * 32 + (sl >> 5) * 4) = m;
            }
            // the dense sample rows of this row block, from the same slots: thread (tile w, playlist jj, half) writes 64
            // contiguous bytes, a wave 32 whole 128-byte rows -- the accumulator layout itself would store 16-byte
            // pieces of 64 different rows per instruction (4.9 us of the launch, measured with stage stamps)
            for (int t2 = tid; t2 < NW * 64 && p.out; t2 += NW * 64) {   // p.out == null: the launch leaves maxima only
                const int w = t2 >> 6, jj = (t2 & 63) >> 1, half = t2 & 1;
                const int item_w = w * p.nb_rg + bir + round * (p.nb_rg * NW);
                const int row = rg * R_TILE + rb * 32 + jj;
                if (item_w < p.ts.n_items && row < p.B) {
                    float* orow = p.out + (size_t)row * p.ld + (size_t)item_w * 32 + half * 16;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        *reinterpret_cast<float4*>(orow + 4 * q4) = xl[w * 256 + (half * 4 + q4) * 32 + jj];
                }
            }
        }
    };

    for (int item = item0; item < p.ts.n_items; item += n_ws) {
        const int t = t_cur;
        const float4* wp = p.Wp + (size_t)t * G * 64 + lane;
        // next tile of this wave (or this one again at the end: in-bounds, values unused)
        const int item_n = item + n_ws < p.ts.n_items ? item + n_ws : item;
        const float4* wn = p.Wp + (size_t)t_nxt * G * 64 + lane;
        const int item_nn = item_n + n_ws < p.ts.n_items ? item_n + n_ws : item_n;
        const int t_nn_v = tile_of_item(p.ts, item_nn);            // consumed at the end of this tile

        // bias of the tile's 32 columns, fetched now so the epilogue never waits on memory:
        // lane holds columns v_local(reg) = (reg & 3) + 8 * (reg >> 2) + 4 * hi, reg = 0..15
        const float* bp = p.bias + (size_t)t * 32 + 4 * hi;
        float4 bq[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
            bq[qd] = DT == DT_BF16 ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(bp + 8 * qd);

        // title mix: the other scorer's term of this tile's elements, requested now, consumed in the epilogue
        float mixv[(EPI == EPI_GMAX || EPI == EPI_FILTER) ? RB : 1][16];
        if ((EPI == EPI_GMAX || EPI == EPI_FILTER) && p.mixT) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int row = rg * R_TILE + rb * 32 + j;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int lc = t * 32 + 4 * hi + (e & 3) + 8 * (e >> 2);
                    mixv[rb][e] = (row < p.B && lc < p.ncols && p.col_lo + lc < p.mix_ncols)
                                      ? p.mixT[(size_t)(p.col_lo + lc) * p.mix_ld + row] : 0.0f;
                }
            }
        }

        f32x16 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rb][e] = 0.0f;
        if (DT == DT_BF16) {
            const uint4 bcur = bfrag;
            bfrag = p.bias16[(size_t)t_nxt * 64 + lane];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(bcur), as_bf16x8(ones), acc[rb], 0, 0, 0);
        }

// one k-group (8 k = 4 MFMA steps per accumulator): consume ring slot WB with hidden fragments
// BC, refill the slot from PF, and fetch the NEXT group's hidden fragments into BN.
#define DAE_STEP(WB, PF, BC, BN, GNEXT)                                                        \
    {                                                                                          \
        const float4 a = WB;                                                                   \
        WB = *(PF);                                                                            \
        const float4* hl = lds4 + (size_t)(GNEXT) * (RB * 64) + lane;                          \
        _Pragma("unroll") for (int rb = 0; rb < RB; ++rb) BN[rb] = hl[rb * 64];                \
        __builtin_amdgcn_sched_barrier(0); /* keep the prefetches AHEAD of this group's MFMAs */\
        _Pragma("unroll") for (int rb = 0; rb < RB; ++rb)                                      \
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, BC[rb].x, acc[rb], 0, 0, 0);   \
        _Pragma("unroll") for (int rb = 0; rb < RB; ++rb)                                      \
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, BC[rb].y, acc[rb], 0, 0, 0);   \
        _Pragma("unroll") for (int rb = 0; rb < RB; ++rb)                                      \
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, BC[rb].z, acc[rb], 0, 0, 0);   \
        _Pragma("unroll") for (int rb = 0; rb < RB; ++rb)                                      \
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, BC[rb].w, acc[rb], 0, 0, 0);   \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }

        if (DT == DT_BF16) {
            const uint4* wq_cur = reinterpret_cast<const uint4*>(wp);
            const uint4* wq_nxt = reinterpret_cast<const uint4*>(wn);
            if (GT > 0) {
                // hidden size known: fully unrolled, ring slot and fragment buffer are static
#pragma unroll
                for (int s = 0; s < (GT > 0 ? GT : 1); ++s) {
                    const uint4 a = wq[s % QR];
                    wq[s % QR] = (s + QR < GT) ? wq_cur[(size_t)(s + QR) * 64]
                                               : wq_nxt[(size_t)(s + QR - GT) * 64];
                    const int sn = (s + 1) % (GT > 0 ? GT : 1);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) cb[(s + 1) & 1][rb] = ldsq[(size_t)(sn * RB + rb) * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
                        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(cb[s & 1][rb]),
                                                                          acc[rb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // generic hidden size (G even): two steps per iteration, no deep ring
                for (int s = 0; s < G; s += 2) {
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const uint4 a = wq_cur[(size_t)(s + h2) * 64];
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) {
                            const uint4 b = ldsq[(size_t)((s + h2) * RB + rb) * 64 + lane];
                            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b),
                                                                              acc[rb], 0, 0, 0);
                        }
                    }
                }
            }
        } else {
        int g = 0;
#pragma unroll
        for (; g < G - 4; g += 4) {
            const float4* pf = wp + (size_t)(g + 4) * 64;
            DAE_STEP(wb0, pf,       bA, bB, g + 1)
            DAE_STEP(wb1, pf + 64,  bB, bA, g + 2)
            DAE_STEP(wb2, pf + 128, bA, bB, g + 3)
            DAE_STEP(wb3, pf + 192, bB, bA, g + 4)
        }
        // last 4 groups of the tile: refill from the next tile, wrap the hidden fragments to g = 0
        DAE_STEP(wb0, wn,       bA, bB, g + 1)
        DAE_STEP(wb1, wn + 64,  bB, bA, g + 2)
        DAE_STEP(wb2, wn + 128, bA, bB, g + 3)
        DAE_STEP(wb3, wn + 192, bB, bA, 0)
        }
#undef DAE_STEP

        ASTAMP(2)
        // ---- epilogue -----------------------------------------------------------------------
        // lane holds, for playlist j of row block rb, the columns
        //   v_local(reg) = (reg & 3) + 8 * (reg >> 2) + 4 * hi          (reg = 0..15)
        const int tcol0 = t * 32 + 4 * hi;                // local column of reg 0 in the image

        if ((EPI == EPI_GMAX || EPI == EPI_FILTER) && p.mixT) {
            // the accumulators become the mixed scores (same operations, same order as mix_scores_kernel of the
            // unfused path: title * w_title + dae * w_playlist, no contraction); the bias is consumed here
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int row = rg * R_TILE + rb * 32 + j;
                const float wt = row < p.B ? p.mix_w[row] : 0.0f;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float zb[4] = {bq[qd].x, bq[qd].y, bq[qd].z, bq[qd].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ts = dae_sigmoidf(acc[rb][4 * qd + e] + zb[e]) * wt;
                        acc[rb][4 * qd + e] = ts + mixv[rb][4 * qd + e];
                    }
                }
            }
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) bq[qd] = make_float4(0.f, 0.f, 0.f, 0.f);
        }

        if (EPI == EPI_DENSE && p.outT) {
            // DAE term of the title mix, transposed: a store instruction writes 32 consecutive rows of one column
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int row = rg * R_TILE + rb * 32 + j;
                if (row >= p.B) continue;
                const float sc = p.row_scale[row];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float zb[4] = {bq[qd].x, bq[qd].y, bq[qd].z, bq[qd].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int lc = tcol0 + 8 * qd + e;
                        if (lc < p.ncols)
                            p.outT[(size_t)(p.col_lo + lc) * p.ld_outT + row] = dae_sigmoidf(acc[rb][4 * qd + e] + zb[e]) * sc;
                    }
                }
            }
        } else if (EPI == EPI_DENSE || (EPI == EPI_GMAX && p.gmax_per_wave)) {
            // (EPI_GMAX with the cross-wave exchange stores its dense rows from LDS, inside gmax_round)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int row = rg * R_TILE + rb * 32 + j;
                if (row >= p.B || (EPI == EPI_GMAX && !p.out)) continue;      // maxima only (bf16 whole-launch filter)
                float* orow = p.out + (size_t)row * p.ld + (size_t)item * 32 + 4 * hi;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int lc = tcol0 + 8 * qd;        // first of 4 consecutive local columns
                    float z[4] = {acc[rb][4 * qd + 0] + bq[qd].x, acc[rb][4 * qd + 1] + bq[qd].y,
                                  acc[rb][4 * qd + 2] + bq[qd].z, acc[rb][4 * qd + 3] + bq[qd].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (p.apply_sigmoid) z[e] = dae_sigmoidf(z[e]);
                        if (p.col_lo + lc + e >= p.mask_from_col || lc + e >= p.ncols)
                            z[e] = -__builtin_inff();
                    }
                    if (lc + 3 < p.ncols || p.fill_pad) {
                        if (p.vec_ok) {
                            *reinterpret_cast<float4*>(orow + 8 * qd) =
                                make_float4(z[0], z[1], z[2], z[3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) orow[8 * qd + e] = z[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (lc + e < p.ncols) orow[8 * qd + e] = z[e];
                    }
                }
            }
            ASTAMP(3)
            if (EPI == EPI_GMAX) gmax_round(true, (item - item0) / n_ws, acc, bq, tcol0);
            ASTAMP(4)
        } else if (EPI == EPI_GMAX) {
            ASTAMP(3)
            gmax_round(true, (item - item0) / n_ws, acc, bq, tcol0);      // maxima AND the dense rows, through LDS
            ASTAMP(4)
        } else if (EPI == EPI_LOSS) {
            // Every element is treated as a NEGATIVE (target 0) here; the few positives of the batch (~100
            // of 170 000 columns per row) are redone from their own dot products by loss_fixup_kernel
            // (train.hip), so no dense target matrix exists.  dL/dz (mean over n_batch folded in) is
            // written transposed, the layout both backward GEMMs read.
            // L = -[y log(p+1e-10) + 0.55 (1-y) log(1-p+1e-10)], p = sigmoid(z), y = 0   (DAEs.py:98-99)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int row = rg * R_TILE + rb * 32 + j;
                if (row >= p.B) continue;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int lc = tcol0 + 8 * qd;
                    const float zb[4] = {bq[qd].x, bq[qd].y, bq[qd].z, bq[qd].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (lc + e < p.ncols) {
                            // training only (parity by tolerance): hardware exp2 / log2 / rcp instead
                            // of the canonical sigmoid and IEEE divides the ranking path needs
                            const float zz = acc[rb][4 * qd + e] + zb[e];
                            const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * zz));
                            const float a0 = 1.0f - pr + 1e-10f;
                            loss_acc -= (0.69314718f * 0.55f) * __builtin_amdgcn_logf(a0);
                            const float dzv = 0.55f * __builtin_amdgcn_rcpf(a0) * pr * (1.0f - pr) * p.inv_nb;
                            if (DT == DT_BF16 && p.dz16)
                                reinterpret_cast<unsigned short*>(p.dzT)[(size_t)(lc + e) * p.ldT + row] =
                                    (unsigned short)bf16_rne(dzv);
                            else
                                p.dzT[(size_t)(lc + e) * p.ldT + row] = dzv;
                        }
                    }
                }
            }
        } else if ((t * 32 < p.ncols) && (p.col_lo + t * 32 < p.n_valid_col)) {
            // filter: tiles without a rankable column (the artist columns) need no epilogue at all; in
            // the others the common case -- this launch walks the LOW-bias tiles -- is "no value of
            // the row block reaches tau": 16 adds, a max-reduction and one compare.  Masks, the LDS
            // atomic for the list slots and the stores only where a lane really passes.
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float tv = tau_r[rb];
                float z[16];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    z[4 * qd + 0] = acc[rb][4 * qd + 0] + bq[qd].x;
                    z[4 * qd + 1] = acc[rb][4 * qd + 1] + bq[qd].y;
                    z[4 * qd + 2] = acc[rb][4 * qd + 2] + bq[qd].z;
                    z[4 * qd + 3] = acc[rb][4 * qd + 3] + bq[qd].w;
                }
                float mx = z[0];
#pragma unroll
                for (int e = 1; e < 16; ++e) mx = fmaxf(mx, z[e]);
                if (mx >= tv) {
                    unsigned m = 0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int lc = tcol0 + (e & 3) + 8 * (e >> 2);
                        if (z[e] >= tv && lc < p.ncols && p.col_lo + lc < p.n_valid_col) m |= 1u << e;
                    }
                    if (m) {
                        const int rloc = rb * 32 + j;
                        const int row = rg * R_TILE + rloc;
                        int base = atomicAdd(&lcnt[rloc], __popc(m));
                        uint2* dst = p.cand + ((size_t)bir * p.Bpad + row) * (size_t)p.cap;
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            if (m & (1u << reg)) {
                                const int lc = tcol0 + (reg & 3) + 8 * (reg >> 2);
                                dst[base++] = make_uint2(__float_as_uint(z[reg]), (unsigned)(p.col_lo + lc));
                            }
                        }
                    }
                }
            }
        }
        t_cur = t_nxt;
        t_nxt = __builtin_amdgcn_readfirstlane(t_nn_v);
    }

    if (EPI == EPI_GMAX) {
        // rounds this wave had no tile for: still joins the exchange (block-uniform trip count in total)
        const int rounds = (p.ts.n_items + n_ws - 1) / n_ws;
        const int mine = item0 < p.ts.n_items ? (p.ts.n_items - item0 + n_ws - 1) / n_ws : 0;
        f32x16 dummy_acc[RB];
        float4 dummy_b[4];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int e = 0; e < 16; ++e) dummy_acc[rb][e] = 0.0f;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) dummy_b[qd] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = mine; r < rounds; ++r) gmax_round(false, r, dummy_acc, dummy_b, 0);
    }
    if (EPI == EPI_FILTER) {
        __syncthreads();
        if (tid < R_TILE) p.cand_cnt[(size_t)bir * p.Bpad + rg * R_TILE + tid] = lcnt[tid];
    }
    if (EPI == EPI_LOSS) {
        // deterministic: lanes -> wave (shuffle tree), waves -> block (fixed order), one slot/block
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) loss_acc += __shfl_xor(loss_acc, d);
        float* wsum = reinterpret_cast<float*>(lcnt);
        __syncthreads();
        if (lane == 0) wsum[wave] = loss_acc;
        __syncthreads();
        if (tid == 0) {
            float s = 0.0f;
            for (int w = 0; w < NW; ++w) s += wsum[w];
            p.loss_part[blockIdx.x] = s * p.inv_nb;          // reduce_mean over the fixed n_batch
        }
    }
}

// ---- training forward (K5) straight from the ROW-MAJOR decoder, hidden = 256 -------------------------------------------
// The decoder changes every training step, so the packed image the scoring kernels stream had to be rebuilt every step (65 us
// for 174 MB) only to be read once: K5 reads the matrix as the optimiser leaves it.  Until round 6 a lane read ITS decoder row
// in 16-byte (fp32) / 32-byte (bf16) pieces -- 32 rows, 64 cache lines per load instruction; the two kernels below take a tile's
// rows as plain 1 KB reads through LDS instead.  The k order of a sum differs from the canonical chain: this is the training
// path, compared by tolerance.  Loss epilogue as EPI_LOSS above (every element a negative; the positives are redone by train.hip's
// loss_fixup_kernel).
struct LossRmP {
    const float* W; const float* bias; const float* h;       // [V][H], [V], [B][H] row-major
    int V, H, B, n_rg, nb_rg;
    float inv_nb; float* dzT; int64_t ldT; float* loss_part;
};

// ---- K5, bf16 operands, hidden 256, batches of at most 256 playlists: the W tile through LDS (round 6) ---------------
// Until round 6 a wave was a tile of 32 decoder rows x 128 playlists, every lane reading ITS decoder row in 32-byte pieces:
// 64 lanes, 32 rows -- 64 different cache lines per load
// instruction, one tag lookup each.  Measured (rocprofv3 counters + the launch with one part removed at a time,
// profiles/r06_notes.md 8): 90 us with, 48 us without the W loads; halving the epilogue's instruction count changed nothing.
// Here a workgroup of 8 waves takes a tile x ALL playlists (58.8 us):
//   * the 8 waves fetch the tile's 32 rows as 32 plain 1 KB row reads (4 per wave), round them to bf16 and put them in LDS
//     ([row][k], 528-byte rows: conflict-free both ways), two tiles in rotation, one barrier per tile;
//   * a wave is one block of 32 playlists: its hidden fragments (16 k-steps x 16 B) live in REGISTERS for the whole launch,
//     the A fragments (decoder rows) come from LDS, 16 MFMAs per tile, then the epilogue of its 32 x 32 logits;
//   * every W byte is read once per launch by one workgroup (the 128-row kernel read it per row group, through L2).
// The k order of every dot product is what the 128-row kernel's was (16 steps of 16, fp32 accumulate in the matrix pipe): the
// same loss, bit for bit.
template <bool DZ16>
__global__ __launch_bounds__(512, 1) void decode_loss_shared_bf16_kernel(const LossRmP p)
{
    constexpr int NW = 8, LDW = 132;                                   // dwords per staged decoder row (128 + 4 of padding)
    __shared__ __attribute__((aligned(16))) unsigned wt[2][32 * LDW];
    __shared__ float wsum[NW];
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = wave * 32 + j;                                     // this lane's playlist
    const int H4 = p.H >> 2;                                           // (64)
    const float4* W4 = reinterpret_cast<const float4*>(p.W);
    const int n_tiles = (p.V + 31) >> 5;
    const int nb = gridDim.x;

    // hidden fragments of the lane's playlist: step s = k 16 s + 8 hi .. + 7 (zeros past the batch)
    uint4 hb[16];
    {
        const float4* hr = reinterpret_cast<const float4*>(p.h) + (size_t)(row < p.B ? row : 0) * H4 + 2 * hi;
        float4 ha[16], hc[16];                                         // (all 32 loads in flight at once)
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) { ha[s_] = hr[4 * s_]; hc[s_] = hr[4 * s_ + 1]; }
        const unsigned keep = row < p.B ? 0xFFFFFFFFu : 0u;            // bf16(0) = 0
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            const float4 a = ha[s_], b = hc[s_];
            hb[s_] = make_uint4((bf16_rne(a.x) | (bf16_rne(a.y) << 16)) & keep, (bf16_rne(a.z) | (bf16_rne(a.w) << 16)) & keep,
                                (bf16_rne(b.x) | (bf16_rne(b.y) << 16)) & keep, (bf16_rne(b.z) | (bf16_rne(b.w) << 16)) & keep);
        }
    }
    // the wave's four rows of a tile: lane l holds floats 4 l .. 4 l + 3 of each (rows past V: the last row, masked later)
    auto load_rows = [&](int t, float4 (&wr)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int v = t * 32 + 4 * wave + r;
            wr[r] = W4[(size_t)(v < p.V ? v : p.V - 1) * H4 + lane];
        }
    };
    auto stage_rows = [&](int buf, const float4 (&wr)[4]) {
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4_t f = {wr[r].x, wr[r].y, wr[r].z, wr[r].w};
            const uint2 pk = __builtin_bit_cast(uint2, __builtin_convertvector(f, bf16x4_t));        // 2 x v_cvt_pk_bf16_f32 (RNE)
            *reinterpret_cast<uint2*>(&wt[buf][(4 * wave + r) * LDW + 2 * lane]) = pk;
        }
    };

    // the bias of the lane's 16 columns of a tile (columns past V: b[V - 1], never used)
    auto load_bias = [&](int t, float4 (&b)[4]) {
        if (t * 32 + 32 <= p.V) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) b[qd] = *reinterpret_cast<const float4*>(p.bias + t * 32 + 4 * hi + 8 * qd);
        } else {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int c = t * 32 + 4 * hi + 8 * qd, last = p.V - 1;
                b[qd] = make_float4(p.bias[c < last ? c : last], p.bias[c + 1 < last ? c + 1 : last],
                                    p.bias[c + 2 < last ? c + 2 : last], p.bias[c + 3 < last ? c + 3 : last]);
            }
        }
    };

    float loss_acc = 0.0f;
    float4 wr[4], bq[4], bq_n[4];
    int t = blockIdx.x;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) bq[qd] = bq_n[qd] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < n_tiles) {
        load_rows(t, wr);
        load_bias(t, bq);
        stage_rows(0, wr);
        if (t + nb < n_tiles) load_rows(t + nb, wr);
    }
    __syncthreads();
    const float k1 = 0.55f * p.inv_nb;
    const bool rows_in = wave * 32 + 32 <= p.B;                        // wave-uniform
    int buf = 0;
    for (; t < n_tiles; t += nb, buf ^= 1) {
        // everything requested one iteration ago has arrived: W rows of tile t + nb (registers), this tile's bias
        if (t + nb < n_tiles) {
            stage_rows(buf ^ 1, wr);                                   // (that buffer's readers passed the barrier)
            load_bias(t + nb, bq_n);
        }
        if (t + 2 * nb < n_tiles) load_rows(t + 2 * nb, wr);           // consumed one iteration from now
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        const unsigned* wl = &wt[buf][j * LDW + 4 * hi];
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            const uint4 af = *reinterpret_cast<const uint4*>(wl + 8 * s_);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(af), as_bf16x8(hb[s_]), acc, 0, 0, 0);
        }
        // epilogue: lane (j, hi) holds, for playlist `row`, the columns t 32 + 4 hi + 8 qd + e
        if (t * 32 + 32 <= p.V && rows_in) {
            const unsigned lane_off = (unsigned)(4 * hi) * (unsigned)p.ldT + (unsigned)row;
            unsigned short* const d16 = reinterpret_cast<unsigned short*>(p.dzT) + (size_t)t * 32 * p.ldT;
            float* const d32 = p.dzT + (size_t)t * 32 * p.ldT;
            if (DZ16) {
                // dL/dz = 0.55 y (1 - y) / (1 - y + 1e-10) (DAEs.py:98-99's negatives); the quotient is 1 to 2^-13 -- far below a
                // bf16's half unit -- unless 1 - y < 1e-6 (a logit above 13.8): those elements are redone below
                float q_min = 1.0f;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float zb[4] = {bq[qd].x, bq[qd].y, bq[qd].z, bq[qd].w};
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        float dzv[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const float zz = acc[4 * qd + e + u] + zb[e + u];
                            const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * zz));
                            const float q = 1.0f - pr;
                            loss_acc -= (0.69314718f * 0.55f) * __builtin_amdgcn_logf(q + 1e-10f);
                            q_min = fminf(q_min, q);
                            dzv[u] = k1 * pr;
                        }
                        typedef float f32x2_t __attribute__((ext_vector_type(2)));
                        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                        const f32x2_t d2 = {dzv[0], dzv[1]};
                        const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(d2, bf16x2_t));
                        (d16 + (size_t)(8 * qd + e) * p.ldT)[lane_off] = (unsigned short)(pk & 0xFFFFu);
                        (d16 + (size_t)(8 * qd + e + 1) * p.ldT)[lane_off] = (unsigned short)(pk >> 16);
                    }
                }
                if (__builtin_expect(__ballot(q_min < 1e-6f) != 0ull, 0)) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const float zb[4] = {bq[qd].x, bq[qd].y, bq[qd].z, bq[qd].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float zz = acc[4 * qd + e] + zb[e];
                            const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * zz));
                            const float q = 1.0f - pr;
                            if (q < 1e-6f)
                                (d16 + (size_t)(8 * qd + e) * p.ldT)[lane_off] =
                                    (unsigned short)bf16_rne(k1 * __builtin_amdgcn_rcpf(q + 1e-10f) * pr * q);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float zb[4] = {bq[qd].x, bq[qd].y, bq[qd].z, bq[qd].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float zz = acc[4 * qd + e] + zb[e];
                        const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * zz));
                        const float a0 = 1.0f - pr + 1e-10f;
                        loss_acc -= (0.69314718f * 0.55f) * __builtin_amdgcn_logf(a0);
                        (d32 + (size_t)(8 * qd + e) * p.ldT)[lane_off] = 0.55f * __builtin_amdgcn_rcpf(a0) * pr * (1.0f - pr) * p.inv_nb;
                    }
                }
            }
        } else if (row < p.B) {
            const int tcol0 = t * 32 + 4 * hi;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int lc = tcol0 + 8 * qd;
                const float zb[4] = {bq[qd].x, bq[qd].y, bq[qd].z, bq[qd].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (lc + e < p.V) {
                        const float zz = acc[4 * qd + e] + zb[e];
                        const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * zz));
                        const float a0 = 1.0f - pr + 1e-10f;
                        loss_acc -= (0.69314718f * 0.55f) * __builtin_amdgcn_logf(a0);
                        const float dzv = 0.55f * __builtin_amdgcn_rcpf(a0) * pr * (1.0f - pr) * p.inv_nb;
                        if (DZ16)
                            reinterpret_cast<unsigned short*>(p.dzT)[(size_t)(lc + e) * p.ldT + row] = (unsigned short)bf16_rne(dzv);
                        else
                            p.dzT[(size_t)(lc + e) * p.ldT + row] = dzv;
                    }
                }
            }
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) bq[qd] = bq_n[qd];
        __syncthreads();
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) loss_acc += __shfl_xor(loss_acc, d);
    if (lane == 0) wsum[wave] = loss_acc;
    __syncthreads();
    if (tid == 0) {
        float sm = 0.0f;
        for (int w = 0; w < NW; ++w) sm += wsum[w];
        p.loss_part[blockIdx.x] = sm * p.inv_nb;
    }
}

// ---- K5 + K7 in one launch (bf16 operands, dz^T as bf16, hidden 256, batches <= 256): dh folded into the forward (round 6) ----
// K7 (train.hip grad_hidden_kernel) reads the whole decoder a second time and dz^T back from HBM to form dh = dz W_dec.  Here
// the tile of decoder rows is in LDS already and a lane holds its playlist's 16 dz values of the tile when the epilogue ends:
// the wave multiplies them (A operand: straight from the epilogue's packed bf16 pairs -- the MFMA's k-slots are assigned to
// the tile's rows the way the accumulator layout hands them out) by the tile (B operand: a TRANSPOSED bf16 copy of the tile
// in LDS, [hidden unit][row], the rows permuted the same way) into 8 accumulators = dh[its 32 playlists][256], kept for the
// whole launch and left as one partial per workgroup (part[blockIdx.x][playlist][hidden]: the fixed-order reduce of
// hidden_backward_kernel sums them, K7's chunks before).  Every element is still a NEGATIVE here: loss_fixup_kernel<.., CORR>
// adds (dz_positive - dz_negative) W_dec[v] for the ~25 k positives as one more partial.  Products: bf16(dz) x bf16(W), fp32
// accumulate, as K7's.  Registers: 128 (dh) + 32 (half of the hidden fragments; the other half in LDS) + the forward's.
__device__ __forceinline__ unsigned k5d_pk2(float a, float b)
{
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__global__ __launch_bounds__(512, 1) void decode_loss_dh_bf16_kernel(const LossRmP p, float* __restrict__ part, int Bpad64)
{
    constexpr int NW = 8, LDW = 132, LDT = 18;                         // dwords per staged row: forward copy / transposed copy
    extern __shared__ __attribute__((aligned(16))) unsigned dyn[];
    unsigned* const wt = dyn;                                          // [2][32 * LDW]          forward copy  [row v][k]
    unsigned* const wt2 = dyn + 2 * 32 * LDW;                          // [2][256 * LDT]         transposed    [hidden][row, permuted]
    uint4* const hq = reinterpret_cast<uint4*>(dyn + 2 * 32 * LDW + 2 * 256 * LDT);     // [8 steps][8 waves][64]: steps 8 .. 15
    float* const wsum = reinterpret_cast<float*>(hq + 8 * 8 * 64);
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = wave * 32 + j;
    const int H4 = p.H >> 2;
    const float4* W4 = reinterpret_cast<const float4*>(p.W);
    const int n_tiles = (p.V + 31) >> 5;
    const int nb = gridDim.x;
    const int t_last = n_tiles - 1;

    // hidden fragments of the lane's playlist: steps 0 .. 7 in registers, 8 .. 15 in LDS (zeros past the batch)
    uint4 hb[8];
    {
        const float4* hr = reinterpret_cast<const float4*>(p.h) + (size_t)(row < p.B ? row : 0) * H4 + 2 * hi;
        float4 ha[16], hc[16];
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) { ha[s_] = hr[4 * s_]; hc[s_] = hr[4 * s_ + 1]; }
        const unsigned keep = row < p.B ? 0xFFFFFFFFu : 0u;
#define K5D_FRAG(A, B) make_uint4((bf16_rne(A.x) | (bf16_rne(A.y) << 16)) & keep, (bf16_rne(A.z) | (bf16_rne(A.w) << 16)) & keep, \
                                  (bf16_rne(B.x) | (bf16_rne(B.y) << 16)) & keep, (bf16_rne(B.z) | (bf16_rne(B.w) << 16)) & keep)
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) hb[s_] = K5D_FRAG(ha[s_], hc[s_]);
#pragma unroll
        for (int s_ = 8; s_ < 16; ++s_) hq[((s_ - 8) * NW + wave) * 64 + lane] = K5D_FRAG(ha[s_], hc[s_]);
#undef K5D_FRAG
    }
    // the wave's four rows of a tile (lane l: floats 4 l .. 4 l + 3 of each), staged twice: as they are ([row][k]) and transposed
    // ([hidden 4 l + c][the wave's 4 rows]: position of row v = 16 s + 4 h + 8 g + i in its 32: (2 s + h) 8 + 4 g + i, with
    // (s, g, h) = bits of the wave index -- the k-slot order of the dh MFMAs below)
    float4 w0, w1, w2, w3;
    const int tpos = (((wave >> 2) << 1) | (wave & 1)) * 4 + ((wave >> 1) & 1) * 2;          // dwords into a transposed row
#define K5D_LOAD(T)                                                                            \
    {                                                                                          \
        const int v_ = (T) * 32 + 4 * wave;                                                    \
        w0 = W4[(size_t)(v_ < p.V ? v_ : p.V - 1) * H4 + lane];                                \
        w1 = W4[(size_t)(v_ + 1 < p.V ? v_ + 1 : p.V - 1) * H4 + lane];                        \
        w2 = W4[(size_t)(v_ + 2 < p.V ? v_ + 2 : p.V - 1) * H4 + lane];                        \
        w3 = W4[(size_t)(v_ + 3 < p.V ? v_ + 3 : p.V - 1) * H4 + lane];                        \
    }
#define K5D_PK(A, B) k5d_pk2((A), (B))                                       /* v_cvt_pk_bf16_f32 (RNE) */
#define K5D_STAGE(BUF)                                                                         \
    {                                                                                          \
        unsigned* d_ = wt + (BUF) * 32 * LDW + (4 * wave) * LDW + 2 * lane;                    \
        *reinterpret_cast<uint2*>(d_) = make_uint2(K5D_PK(w0.x, w0.y), K5D_PK(w0.z, w0.w));    \
        *reinterpret_cast<uint2*>(d_ + LDW) = make_uint2(K5D_PK(w1.x, w1.y), K5D_PK(w1.z, w1.w)); \
        *reinterpret_cast<uint2*>(d_ + 2 * LDW) = make_uint2(K5D_PK(w2.x, w2.y), K5D_PK(w2.z, w2.w)); \
        *reinterpret_cast<uint2*>(d_ + 3 * LDW) = make_uint2(K5D_PK(w3.x, w3.y), K5D_PK(w3.z, w3.w)); \
        unsigned* e_ = wt2 + (BUF) * 256 * LDT + (4 * lane) * LDT + tpos;                      \
        *reinterpret_cast<uint2*>(e_) = make_uint2(K5D_PK(w0.x, w1.x), K5D_PK(w2.x, w3.x));    \
        *reinterpret_cast<uint2*>(e_ + LDT) = make_uint2(K5D_PK(w0.y, w1.y), K5D_PK(w2.y, w3.y)); \
        *reinterpret_cast<uint2*>(e_ + 2 * LDT) = make_uint2(K5D_PK(w0.z, w1.z), K5D_PK(w2.z, w3.z)); \
        *reinterpret_cast<uint2*>(e_ + 3 * LDT) = make_uint2(K5D_PK(w0.w, w1.w), K5D_PK(w2.w, w3.w)); \
    }
    auto load_bias = [&](int t) -> float {
        const int c = t * 32 + j;
        return p.bias[c < p.V ? c : p.V - 1];
    };

    f32x16 dh[8];
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) dh[b][e] = 0.0f;
    float loss_acc = 0.0f;
    float bl, bl_n = 0.0f;
    int t = blockIdx.x;
    K5D_LOAD(t < t_last ? t : t_last)
    bl = load_bias(t < t_last ? t : t_last);
    K5D_STAGE(0)
    K5D_LOAD(t + nb < t_last ? t + nb : t_last)
    __syncthreads();
    const float k1 = 0.55f * p.inv_nb;
    const bool row_in = row < p.B;
    int buf = 0;
    for (; t < n_tiles; t += nb, buf ^= 1) {
        K5D_STAGE(buf ^ 1)                                             // tile t + nb (or a clamped copy nobody reads)
        bl_n = load_bias(t + nb < t_last ? t + nb : t_last);
        K5D_LOAD(t + 2 * nb < t_last ? t + 2 * nb : t_last)
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        const unsigned* wl = wt + buf * 32 * LDW + j * LDW + 4 * hi;
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            const uint4 af = *reinterpret_cast<const uint4*>(wl + 8 * s_);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(af), as_bf16x8(hb[s_]), acc, 0, 0, 0);
        }
#pragma unroll
        for (int s_ = 8; s_ < 16; ++s_) {
            const uint4 af = *reinterpret_cast<const uint4*>(wl + 8 * s_);
            const uint4 bf = hq[((s_ - 8) * NW + wave) * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(af), as_bf16x8(bf), acc, 0, 0, 0);
        }
        // epilogue: lane (j, hi) holds, for playlist `row`, the columns t 32 + 4 hi + 8 qd + e; dz of columns past V and of
        // playlists past the batch is 0 (and not stored)
        float dzv[16];
        float q_min = 1.0f;
        const int tcol0 = t * 32 + 4 * hi;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float zz = acc[4 * qd + e] + __shfl(bl, 4 * hi + 8 * qd + e);
                const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * zz));
                const float q = 1.0f - pr;
                const bool live = row_in && tcol0 + 8 * qd + e < p.V;
                loss_acc -= live ? (0.69314718f * 0.55f) * __builtin_amdgcn_logf(q + 1e-10f) : 0.0f;
                q_min = fminf(q_min, live ? q : 1.0f);
                // dL/dz = 0.55 y (1 - y) / (1 - y + 1e-10) (DAEs.py:98-99's negatives): the quotient is 1 to 2^-13 unless 1 - y < 1e-6
                dzv[4 * qd + e] = live ? k1 * pr : 0.0f;
            }
        }
        if (__builtin_expect(__ballot(q_min < 1e-6f) != 0ull, 0)) {   // (a logit above 13.8 somewhere in the wave: the exact form)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float zz = acc[4 * qd + e] + __shfl(bl, 4 * hi + 8 * qd + e);
                    const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * zz));
                    const float q = 1.0f - pr;
                    const bool live = row_in && tcol0 + 8 * qd + e < p.V;
                    if (live && q < 1e-6f) dzv[4 * qd + e] = k1 * __builtin_amdgcn_rcpf(q + 1e-10f) * pr * q;
                }
        }
        unsigned pk[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) pk[x] = K5D_PK(dzv[2 * x], dzv[2 * x + 1]);
        if (row_in) {
            unsigned short* const d16 = reinterpret_cast<unsigned short*>(p.dzT) + (size_t)t * 32 * p.ldT;
            const unsigned lane_off = (unsigned)(4 * hi) * (unsigned)p.ldT + (unsigned)row;
            if (t * 32 + 32 <= p.V) {
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const int c = 8 * (x >> 1) + 2 * (x & 1);              // column offset of the pair's first element
                    (d16 + (size_t)c * p.ldT)[lane_off] = (unsigned short)(pk[x] & 0xFFFFu);
                    (d16 + (size_t)(c + 1) * p.ldT)[lane_off] = (unsigned short)(pk[x] >> 16);
                }
            } else {
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const int c = 8 * (x >> 1) + 2 * (x & 1);
                    if (tcol0 + c < p.V) (d16 + (size_t)c * p.ldT)[lane_off] = (unsigned short)(pk[x] & 0xFFFFu);
                    if (tcol0 + c + 1 < p.V) (d16 + (size_t)(c + 1) * p.ldT)[lane_off] = (unsigned short)(pk[x] >> 16);
                }
            }
        }
        // dh[playlist][hidden] += dz[playlist][the tile's rows] W[rows][hidden]: A = dz (k-slot i of lane half hi, step s: row
        // 16 s + 4 hi + i for i < 4, + 8 + (i - 4) above), B = the transposed tile, block b of 32 hidden units
        const uint4 a0 = make_uint4(pk[0], pk[1], pk[2], pk[3]), a1 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        const unsigned* tl = wt2 + buf * 256 * LDT + j * LDT + 4 * hi;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint2 x0 = *reinterpret_cast<const uint2*>(tl + b * 32 * LDT), x1 = *reinterpret_cast<const uint2*>(tl + b * 32 * LDT + 2);
            const uint2 y0 = *reinterpret_cast<const uint2*>(tl + b * 32 * LDT + 8), y1 = *reinterpret_cast<const uint2*>(tl + b * 32 * LDT + 10);
            dh[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0), as_bf16x8(make_uint4(x0.x, x0.y, x1.x, x1.y)), dh[b], 0, 0, 0);
            dh[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a1), as_bf16x8(make_uint4(y0.x, y0.y, y1.x, y1.y)), dh[b], 0, 0, 0);
        }
        bl = bl_n;
        __syncthreads();
    }
    // the workgroup's partial of dh: register reg of lane (j, hi), block b = playlist 32 wave + (reg & 3) + 8 (reg >> 2) + 4 hi,
    // hidden 32 b + j
    {
        float* pw = part + (size_t)blockIdx.x * Bpad64 * p.H;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hi;
                if (r < Bpad64) pw[(size_t)r * p.H + b * 32 + j] = dh[b][reg];
            }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) loss_acc += __shfl_xor(loss_acc, d);
    if (lane == 0) wsum[wave] = loss_acc;
    __syncthreads();
    if (tid == 0) {
        float sm = 0.0f;
        for (int w = 0; w < NW; ++w) sm += wsum[w];
        p.loss_part[blockIdx.x] = sm * p.inv_nb;
    }
}
#undef K5D_LOAD
#undef K5D_PK
#undef K5D_STAGE

// ---- K5, fp32 operands, hidden 256, batches of at most 256 playlists: the same shape on v_mfma_f32_32x32x2_f32 (round 6) ----
// The row-gathering fp32 kernel it replaces (64 cache lines per load instruction, as the 128-row bf16 kernel's): 219 us for
// 142 us of fp32 matrix work at the nominal clock; this one 209 us.  Here: a workgroup = a tile of 32 decoder rows x all playlists, the rows
// as plain 1 KB reads into LDS (fp32, 1 040-byte rows, two tiles in rotation), a wave = 32 playlists whose hidden row sits in
// 128 registers, the A fragments from LDS (a float4 = four MFMAs).  The k order inside a dot product is the row-major kernel's
// (pairs (8 g + c, 8 g + 4 + c)); training compares by tolerance.
__global__ __launch_bounds__(512, 1) void decode_loss_shared_f32_kernel(const LossRmP p)
{
    constexpr int NW = 8, LDW = 260;                                   // dwords per staged decoder row (256 + 4 of padding)
    extern __shared__ __attribute__((aligned(16))) float wtf[];       // [2][32 * LDW] | wsum[NW]
    float* const wsum = wtf + 2 * 32 * LDW;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = wave * 32 + j;
    const int H4 = p.H >> 2;
    const float4* W4 = reinterpret_cast<const float4*>(p.W);
    const int n_tiles = (p.V + 31) >> 5;
    const int nb = gridDim.x;

    // hidden row of the lane's playlist: group g = k 8 g + 4 hi .. + 3 (zeros past the batch)
    float4 hb[32];
    {
        const float4* hr = reinterpret_cast<const float4*>(p.h) + (size_t)(row < p.B ? row : 0) * H4 + hi;
#pragma unroll
        for (int g = 0; g < 32; ++g) hb[g] = hr[2 * g];
        if (row >= p.B) {
#pragma unroll
            for (int g = 0; g < 32; ++g) hb[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // the wave's four rows of a tile, lane l holding floats 4 l .. 4 l + 3 of each: four named registers (an array behind a
    // lambda's reference stayed in scratch here)
    float4 w0, w1, w2, w3;
#define K5F_LOAD(T)                                                                            \
    {                                                                                          \
        const int v_ = (T) * 32 + 4 * wave;                                                    \
        w0 = W4[(size_t)(v_ < p.V ? v_ : p.V - 1) * H4 + lane];                                \
        w1 = W4[(size_t)(v_ + 1 < p.V ? v_ + 1 : p.V - 1) * H4 + lane];                        \
        w2 = W4[(size_t)(v_ + 2 < p.V ? v_ + 2 : p.V - 1) * H4 + lane];                        \
        w3 = W4[(size_t)(v_ + 3 < p.V ? v_ + 3 : p.V - 1) * H4 + lane];                        \
    }
#define K5F_STAGE(BUF)                                                                         \
    {                                                                                          \
        float* d_ = &wtf[(BUF) * 32 * LDW + (4 * wave) * LDW + 4 * lane];                      \
        *reinterpret_cast<float4*>(d_) = w0;                                                   \
        *reinterpret_cast<float4*>(d_ + LDW) = w1;                                             \
        *reinterpret_cast<float4*>(d_ + 2 * LDW) = w2;                                         \
        *reinterpret_cast<float4*>(d_ + 3 * LDW) = w3;                                         \
    }
    // the tile's 32 bias values, one per lane (lane l: column 32 t + (l & 31)); a lane takes its 16 by shuffle when it needs them
    auto load_bias = [&](int t) -> float {
        const int c = t * 32 + j;
        return p.bias[c < p.V ? c : p.V - 1];
    };
    float loss_acc = 0.0f;
    float bl, bl_n = 0.0f;
    int t = blockIdx.x;
    // (every request is UNCONDITIONAL, on a tile clamped to the last one: a conditionally written register array goes to scratch,
    // and the store to scratch waits for the load it has just issued)
    const int t_last = n_tiles - 1;
    K5F_LOAD(t < t_last ? t : t_last)
    bl = load_bias(t < t_last ? t : t_last);
    K5F_STAGE(0)
    K5F_LOAD(t + nb < t_last ? t + nb : t_last)
    __syncthreads();
    const bool rows_in = wave * 32 + 32 <= p.B;
    // (Tried: the second wave of each SIMD one tile late with its epilogue, so that one wave's MFMAs run under the other's
    // epilogue -- 214 against 209 us: the launch is not losing its time to coinciding phases.)
    auto epilogue = [&](int te, const f32x16& ac, float bv) {
            float4 bq[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                bq[qd] = make_float4(__shfl(bv, 4 * hi + 8 * qd), __shfl(bv, 4 * hi + 8 * qd + 1), __shfl(bv, 4 * hi + 8 * qd + 2),
                                     __shfl(bv, 4 * hi + 8 * qd + 3));
            if (te * 32 + 32 <= p.V && rows_in) {
                const unsigned lane_off = (unsigned)(4 * hi) * (unsigned)p.ldT + (unsigned)row;
                float* const d32 = p.dzT + (size_t)te * 32 * p.ldT;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float zb[4] = {bq[qd].x, bq[qd].y, bq[qd].z, bq[qd].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float zz = ac[4 * qd + e] + zb[e];
                        const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * zz));
                        const float a0 = 1.0f - pr + 1e-10f;
                        loss_acc -= (0.69314718f * 0.55f) * __builtin_amdgcn_logf(a0);
                        (d32 + (size_t)(8 * qd + e) * p.ldT)[lane_off] = 0.55f * __builtin_amdgcn_rcpf(a0) * pr * (1.0f - pr) * p.inv_nb;
                    }
                }
            } else if (row < p.B) {
                const int tcol0 = te * 32 + 4 * hi;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int lc = tcol0 + 8 * qd;
                    const float zb[4] = {bq[qd].x, bq[qd].y, bq[qd].z, bq[qd].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (lc + e < p.V) {
                            const float zz = ac[4 * qd + e] + zb[e];
                            const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * zz));
                            const float a0 = 1.0f - pr + 1e-10f;
                            loss_acc -= (0.69314718f * 0.55f) * __builtin_amdgcn_logf(a0);
                            p.dzT[(size_t)(lc + e) * p.ldT + row] = 0.55f * __builtin_amdgcn_rcpf(a0) * pr * (1.0f - pr) * p.inv_nb;
                        }
                    }
                }
            }
    };
    int buf = 0;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    for (; t < n_tiles; t += nb, buf ^= 1) {
        K5F_STAGE(buf ^ 1)                                             // tile t + nb (or a clamped copy nobody reads)
        bl_n = load_bias(t + nb < t_last ? t + nb : t_last);
        K5F_LOAD(t + 2 * nb < t_last ? t + 2 * nb : t_last)
        __builtin_amdgcn_sched_barrier(0);                             // (hipcc sinks these requests below the 128 MFMAs otherwise)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        const float* wl = &wtf[buf * 32 * LDW + j * LDW + 4 * hi];
        // (one accumulator: a chain of 128 dependent MFMAs per wave, and two -- even / odd groups, summed -- measured the same
        // 209 - 213 us: with two waves per SIMD the pipe is busy either way; the launch is at the fp32 MFMA rate of its clock)
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(wl + 8 * g);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, hb[g].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, hb[g].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, hb[g].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, hb[g].w, acc, 0, 0, 0);
        }
        epilogue(t, acc, bl);
        bl = bl_n;
        __syncthreads();
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) loss_acc += __shfl_xor(loss_acc, d);
    if (lane == 0) wsum[wave] = loss_acc;
    __syncthreads();
    if (tid == 0) {
        float sm = 0.0f;
        for (int w = 0; w < NW; ++w) sm += wsum[w];
        p.loss_part[blockIdx.x] = sm * p.inv_nb;
    }
}
#undef K5F_LOAD
#undef K5F_STAGE

// ---- fp32, hidden = 256, filter epilogue (phase B of the fused path): the generic kernel above with
// one addition, TAIL BALANCE.  A launch of n tiles over n_ws wave slots runs floor(n / n_ws) whole rounds
// and a last round with `rem` tiles; when that round is at most half full (222 of 512 slots at batch 256,
// i.e. 10 rounds of time for 9.43 rounds of work) each of its tiles is split between TWO waves by row
// blocks (128 playlists -> 2 x 64), so the round costs half a tile time.  The k chain of every output is
// untouched (bit-exact), only which wave owns which row block changes.
template <int N>
struct IntC { static constexpr int value = N; };

template <int DUMMY>
__global__ __launch_bounds__(256, 1) void decode_f32_h256_filter_kernel(const DecP p)
{
    constexpr int RB = 4, G = 32, NW = 4, R_TILE = 128;
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int j = lane & 31;

    const int gs = DAE_NUM_XCD * p.n_rg;
    const int q = blockIdx.x / gs, rem_b = blockIdx.x % gs;
    const int rg = rem_b / DAE_NUM_XCD;
    const int bir = q * DAE_NUM_XCD + (rem_b % DAE_NUM_XCD);

    constexpr int n_h4 = RB * 64 * G;
    // this wave's work: whole tiles ws + r * n_ws (r < R), then possibly one tile -- or half of one -- of
    // the last round
    const int n_items = p.ts.n_items;
    const int n_ws = p.nb_rg * NW;
    const int ws = wave * p.nb_rg + bir;
    const int R = n_items / n_ws, rem = n_items - R * n_ws;
    const bool split = rem > 0 && 2 * rem <= n_ws;
    const int n_it = R + (ws < (split ? 2 * rem : rem) ? 1 : 0);
    auto item_at = [&](int r) {
        r = r < n_it - 1 ? r : n_it - 1;
        r = r < 0 ? 0 : r;
        const int it = r < R ? ws + r * n_ws : R * n_ws + (split ? (ws >> 1) : ws);
        return it < n_items ? it : 0;
    };
    // the first two tile ids go out ahead of the hidden tile's loads, so that the W ring can start before the workgroup
    // meets (tile, barrier, ids, W was one more dependent trip to memory in front of the first MFMA)
    const int tv_cur = tile_of_item(p.ts, item_at(0)), tv_nxt = tile_of_item(p.ts, item_at(1));
    {
        const float4* src = p.hp + (size_t)rg * n_h4;
        constexpr int NT = NW * 64;
#pragma unroll
        for (int i0 = 0; i0 < n_h4; i0 += 8 * NT) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[i0 + u * NT + tid];
#pragma unroll
            for (int u = 0; u < 8; ++u) lds4[i0 + u * NT + tid] = v[u];
        }
    }
    int* lcnt = reinterpret_cast<int*>(lds4 + n_h4);
    float* ltau = reinterpret_cast<float*>(lcnt + R_TILE);
    if (tid < R_TILE) {
        lcnt[tid] = 0;
        ltau[tid] = rg * R_TILE + tid < p.B ? p.tau[rg * R_TILE + tid] : __builtin_inff();
    }
    float4 wb0, wb1, wb2, wb3;
    float4 bA[RB], bB[RB];
    int t_cur = __builtin_amdgcn_readfirstlane(tv_cur), t_nxt = __builtin_amdgcn_readfirstlane(tv_nxt);
    {
        const float4* w0 = p.Wp + (size_t)t_cur * G * 64 + lane;     // (a wave without work reads a listed tile and drops it)
        wb0 = w0[0]; wb1 = w0[64]; wb2 = w0[128]; wb3 = w0[192];
    }
    __syncthreads();

    float tau_r[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) tau_r[rb] = ltau[rb * 32 + j];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) bA[rb] = lds4[rb * 64 + lane];

    for (int r = 0; r < n_it; ++r) {
        const int t = t_cur;
        const float4* wp = p.Wp + (size_t)t * G * 64 + lane;
        const float4* wn = p.Wp + (size_t)t_nxt * G * 64 + lane;
        const int t_nn_v = tile_of_item(p.ts, item_at(r + 2));        // two tiles ahead (see decode_f32_kernel)
        const float* bp = p.bias + (size_t)t * 32 + 4 * hi;
        float4 bq[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) bq[qd] = *reinterpret_cast<const float4*>(bp + 8 * qd);
        const int tcol0 = t * 32 + 4 * hi;
        const bool rankable = (t * 32 < p.ncols) && (p.col_lo + t * 32 < p.n_valid_col);

        // one tile over row blocks [R0, R0 + RN): GEMM with the 4-deep W ring, then the filter epilogue
        auto tile = [&](auto r0c, auto rnc) {
            constexpr int R0 = decltype(r0c)::value, RN = decltype(rnc)::value;
            f32x16 acc[RN];
#pragma unroll
            for (int i = 0; i < RN; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
#define DAE_STEPR(WB, PF, BC, BN, GNEXT)                                                              \
    {                                                                                                 \
        const float4 a = WB;                                                                          \
        WB = *(PF);                                                                                   \
        const float4* hl = lds4 + (size_t)(GNEXT) * (RB * 64) + lane;                                 \
        _Pragma("unroll") for (int i = 0; i < RN; ++i) BN[R0 + i] = hl[(R0 + i) * 64];                \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        _Pragma("unroll") for (int i = 0; i < RN; ++i)                                                \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, BC[R0 + i].x, acc[i], 0, 0, 0);        \
        _Pragma("unroll") for (int i = 0; i < RN; ++i)                                                \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, BC[R0 + i].y, acc[i], 0, 0, 0);        \
        _Pragma("unroll") for (int i = 0; i < RN; ++i)                                                \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, BC[R0 + i].z, acc[i], 0, 0, 0);        \
        _Pragma("unroll") for (int i = 0; i < RN; ++i)                                                \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, BC[R0 + i].w, acc[i], 0, 0, 0);        \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
            int g = 0;
#pragma unroll
            for (; g < G - 4; g += 4) {
                const float4* pf = wp + (size_t)(g + 4) * 64;
                DAE_STEPR(wb0, pf,       bA, bB, g + 1)
                DAE_STEPR(wb1, pf + 64,  bB, bA, g + 2)
                DAE_STEPR(wb2, pf + 128, bA, bB, g + 3)
                DAE_STEPR(wb3, pf + 192, bB, bA, g + 4)
            }
            DAE_STEPR(wb0, wn,       bA, bB, g + 1)
            DAE_STEPR(wb1, wn + 64,  bB, bA, g + 2)
            DAE_STEPR(wb2, wn + 128, bA, bB, g + 3)
            // the next tile may own OTHER row blocks (a full tile's successor is never narrower than
            // ... a half tile is always last): fetch group 0 for ALL row blocks
            {
                const float4 a = wb3;
                wb3 = wn[192];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) bA[rb] = lds4[rb * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < RN; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bB[R0 + i].x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RN; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bB[R0 + i].y, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RN; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bB[R0 + i].z, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RN; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bB[R0 + i].w, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef DAE_STEPR
            if (!rankable) return;
#pragma unroll
            for (int i = 0; i < RN; ++i) {
                constexpr int dummy = 0; (void)dummy;
                const int rb = R0 + i;
                const float tv = tau_r[rb];
                float z[16];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    z[4 * qd + 0] = acc[i][4 * qd + 0] + bq[qd].x;
                    z[4 * qd + 1] = acc[i][4 * qd + 1] + bq[qd].y;
                    z[4 * qd + 2] = acc[i][4 * qd + 2] + bq[qd].z;
                    z[4 * qd + 3] = acc[i][4 * qd + 3] + bq[qd].w;
                }
                float mx = z[0];
#pragma unroll
                for (int e = 1; e < 16; ++e) mx = fmaxf(mx, z[e]);
                if (mx >= tv) {
                    unsigned m = 0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int lc = tcol0 + (e & 3) + 8 * (e >> 2);
                        if (z[e] >= tv && lc < p.ncols && p.col_lo + lc < p.n_valid_col) m |= 1u << e;
                    }
                    if (m) {
                        const int rloc = rb * 32 + j;
                        int base = atomicAdd(&lcnt[rloc], __popc(m));
                        uint2* dst = p.cand + ((size_t)bir * p.Bpad + rg * R_TILE + rloc) * (size_t)p.cap;
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            if (m & (1u << reg)) {
                                const int lc = tcol0 + (reg & 3) + 8 * (reg >> 2);
                                dst[base++] = make_uint2(__float_as_uint(z[reg]), (unsigned)(p.col_lo + lc));
                            }
                        }
                    }
                }
            }
        };

        if (!(split && r == R)) tile(IntC<0>{}, IntC<RB>{});
        else if ((ws & 1) == 0) tile(IntC<0>{}, IntC<RB / 2>{});
        else tile(IntC<RB / 2>{}, IntC<RB / 2>{});

        t_cur = t_nxt;
        t_nxt = __builtin_amdgcn_readfirstlane(t_nn_v);
    }
    __syncthreads();
    if (tid < R_TILE) p.cand_cnt[(size_t)bir * p.Bpad + rg * R_TILE + tid] = lcnt[tid];
}

// ---- bf16, hidden = 256, filter epilogue (phase B of the fused path) --------------------------------
// A wave owns NT column tiles x RB row blocks.  Measured (profiles/r01_notes.md, us at batch 256 / 1024):
//   <NT = 1, RB = 4, ring 8, 2 waves per SIMD>  19.9 / 55.7   <- default: 4 accumulators per wave
//   <NT = 1, RB = 4, ring 16, 1 wave per SIMD>  23.9 / 65.3
//   <NT = 2, RB = 4, ring 16, 1 wave per SIMD>  25.6 / 61.5   (DAE_BF16_PAIR: a hidden fragment feeds 2 MFMAs)
//   <NT = 1, RB = 8, ring 16, 1 wave per SIMD>  27.6 / 64.1   (DAE_BF16_RTILE=256: W read once at batch 256)
//   3 and 4 waves per SIMD spill (168 / 128 registers) and are slower.
// A micro-benchmark of the inner pattern (scripts/micro/mfma_peak.hip: 4 accumulators, hidden fragments
// from LDS one step ahead, no global memory) reaches 2.3-2.4 PFLOP/s, so neither LDS nor the MFMA issue
// limits it; what the real kernel adds is the W ring, tile indices two groups ahead, the bias MFMA and the
// epilogue (a max-reduction and one compare per row block unless some lane really passes).
// ---- phase A for launches of many rows: per-WAVE group maxima, no exchange ---------------------------------------------------
// The generic phase-A kernel (decode_f32_kernel<4, EPI_GMAX, 16, 4, DT_BF16>) takes the maximum over the tiles the FOUR waves of a
// workgroup decode in a round: row block by row block through LDS, two barriers each, one wave per SIMD (468 registers) -- at
// 2 048 rows the sample (1 / 9 of the tiles) costs 40 us where the filter launch decodes everything in 113.  When a launch's
// sample gives every one of the filter kernel's wave slots (8 per workgroup) at least two tiles, the group can be the tiles ONE
// wave decodes: wave w of workgroup bir takes items w nb_rg + bir + r n_ws, r = 0, 1, ... -- in the plain bias order, so its tiles
// sit n_ws places apart (round r = popularity band r) -- and keeps the running maximum per (row, position in the tile) in
// registers: no LDS traffic beyond the hidden fragments, no barrier after the prologue, two waves per SIMD.  The structure of
// decode_bf16_h256_filter_kernel<1, 4, 8, 8> (same hidden tile, W ring, bias through the matrix pipe: the same logits bit for
// bit) with a static tile assignment and fmax as the epilogue.  Groups are disjoint sets of columns as before: the (k + seeds)-th
// largest of the 8 nb_rg x 32 maxima of a row is a valid threshold.  gmax[row][(wave nb_rg + bir) 32 + position].
__global__ __launch_bounds__(512, 1) void decode_bf16_h256_wavemax_kernel(const DecP p)
{
    constexpr int NS = 16, RB = 4, QR = 8, NW = 8;
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int j = lane & 31;
    const int gs = DAE_NUM_XCD * p.n_rg;
    const int q = blockIdx.x / gs, rem = blockIdx.x % gs;
    const int rg = rem / DAE_NUM_XCD;
    const int bir = q * DAE_NUM_XCD + (rem % DAE_NUM_XCD);
    constexpr int n_h4 = RB * 64 * NS;
    constexpr int NTH = NW * 64;
    constexpr int PER = n_h4 / NTH;                              // 8 uint4 of the hidden tile per thread
    const int n_items = p.ts.n_items;
    const int n_ws = p.nb_rg * NW;
    const int it0 = wave * p.nb_rg + bir;
    const uint4* Wq = reinterpret_cast<const uint4*>(p.Wp);
    const uint4* ldsq = reinterpret_cast<const uint4*>(lds4);
    const uint4 ones = bf16_ones_fragment(hi);
    const bool has = it0 < n_items;
    const int tv0 = tile_of_item(p.ts, has ? it0 : 0);
    const int tv1 = tile_of_item(p.ts, has ? (it0 + n_ws < n_items ? it0 + n_ws : it0) : 0);
    {
        const float4* hsrc = p.hp + (size_t)rg * n_h4;
        float4 hv[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) hv[e] = hsrc[e * NTH + tid];
#pragma unroll
        for (int e = 0; e < PER; ++e) lds4[e * NTH + tid] = hv[e];
    }
    __builtin_amdgcn_sched_barrier(0);
    int t = __builtin_amdgcn_readfirstlane(tv0), u = __builtin_amdgcn_readfirstlane(tv1);
    uint4 wq[QR];
    uint4 cb[2][RB];
    uint4 bfr = p.bias16[(size_t)t * 64 + lane];
#pragma unroll
    for (int k = 0; k < QR; ++k) wq[k] = Wq[(size_t)t * (NS * 64) + k * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) cb[0][rb] = ldsq[rb * 64 + lane];

    f32x16 mx[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx[rb][e] = -__builtin_inff();

    for (int it = it0; it < n_items; it += n_ws) {
        const int it_nn = it + 2 * n_ws;
        const int wv = tile_of_item(p.ts, it_nn < n_items ? it_nn : it);   // consumed at the end of this tile
        const uint4* cur = Wq + (size_t)t * (NS * 64);
        const uint4* nxt = Wq + (size_t)u * (NS * 64);
        f32x16 acc[RB];
        {
            f32x16 zero;
#pragma unroll
            for (int e = 0; e < 16; ++e) zero[e] = 0.0f;
            const uint4 bc = bfr;
            bfr = p.bias16[(size_t)u * 64 + lane];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(bc), as_bf16x8(ones), zero, 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int sn = (s + 1) % NS;
            const uint4 a = wq[s % QR];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(cb[s & 1][rb]), acc[rb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                cb[(s + 1) & 1][rb] = ldsq[(sn * RB + rb) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
            }
            wq[s % QR] = (s + QR < NS) ? cur[(s + QR) * 64 + lane] : nxt[(s + QR - NS) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
        }
        // register reg of the tile is column 32 t + (reg & 3) + 8 (reg >> 2) + 4 hi; columns that are not ranked never enter a maximum
        const bool whole = t * 32 + 31 < p.ncols && p.col_lo + t * 32 + 31 < p.mask_from_col;          // wave-uniform
        if (whole) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) mx[rb][reg] = fmaxf(mx[rb][reg], acc[rb][reg]);
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lc = t * 32 + 4 * hi + (reg & 3) + 8 * (reg >> 2);
                const bool ok = lc < p.ncols && p.col_lo + lc < p.mask_from_col;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) mx[rb][reg] = ok ? fmaxf(mx[rb][reg], acc[rb][reg]) : mx[rb][reg];
            }
        }
        t = u; u = __builtin_amdgcn_readfirstlane(wv);
    }
    // a wave without a tile leaves -inf: absent for the threshold kernel
    if (p.gmax_per_wave == 4) {
        // few tiles per wave (1 024 rows: 2.3): waves w and w + 4 share a group -- ONE exchange at the end of the launch, through the
        // hidden tile's LDS (dead by now): 4 nb_rg x 32 maxima per row instead of 8 nb_rg x 32, so that the threshold kernel reads
        // 4 096, not 8 192 (its 16-key shape).  Groups of ~4.6 tiles from as many bands, as a wave's own tiles are at 2 048 rows.
        // (Pairing NEIGHBOURING positions instead doubles the candidates: in the hottest tiles every column is a winner.)
        __syncthreads();                                          // every wave is past its last hidden fragment
        float* xl = reinterpret_cast<float*>(lds4);
        if (wave >= 4) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    *reinterpret_cast<float4*>(xl + ((((wave - 4) * RB + rb) * 4 + qd) * 64 + lane) * 4) =
                        make_float4(mx[rb][4 * qd], mx[rb][4 * qd + 1], mx[rb][4 * qd + 2], mx[rb][4 * qd + 3]);
        }
        __syncthreads();
        if (wave >= 4) return;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int row = rg * 128 + rb * 32 + j;
            float* gp = p.gmax + (size_t)row * p.ld_gmax + (size_t)(wave * p.nb_rg + bir) * 32 + 4 * hi;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float4 o = *reinterpret_cast<const float4*>(xl + (((wave * RB + rb) * 4 + qd) * 64 + lane) * 4);
                if (row < p.B)
                    *reinterpret_cast<float4*>(gp + 8 * qd) = make_float4(fmaxf(mx[rb][4 * qd], o.x), fmaxf(mx[rb][4 * qd + 1], o.y),
                                                                          fmaxf(mx[rb][4 * qd + 2], o.z), fmaxf(mx[rb][4 * qd + 3], o.w));
            }
        }
        return;
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int row = rg * 128 + rb * 32 + j;
        if (row >= p.B) continue;
        float* gp = p.gmax + (size_t)row * p.ld_gmax + (size_t)(wave * p.nb_rg + bir) * 32 + 4 * hi;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
            *reinterpret_cast<float4*>(gp + 8 * qd) = make_float4(mx[rb][4 * qd], mx[rb][4 * qd + 1], mx[rb][4 * qd + 2], mx[rb][4 * qd + 3]);
    }
}

template <int NT, int RB, int QR, int NW>
__global__ __launch_bounds__(NW * 64, 1) void decode_bf16_h256_filter_kernel(const DecP p)
{
    constexpr int NS = 16, R_TILE = RB * 32;
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int j = lane & 31;
#ifdef DAE_EXPERIMENTS          // stage stamps of wave 0 of workgroups 0 and 100 (DAE_DBG_F): cycle counter at the marked points
#define FSTAMP(i) if (p.stamps && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) p.stamps[(blockIdx.x ? 16 : 0) + (i)] = __builtin_readcyclecounter();
#else
#define FSTAMP(i)
#endif
    FSTAMP(0)

    const int gs = DAE_NUM_XCD * p.n_rg;
    const int q = blockIdx.x / gs, rem = blockIdx.x % gs;
    const int rg = rem / DAE_NUM_XCD;
    const int bir = q * DAE_NUM_XCD + (rem % DAE_NUM_XCD);

    constexpr int n_h4 = RB * 64 * NS;
    constexpr int NTH = NW * 64;
    constexpr int PER = (n_h4 + NTH - 1) / NTH;                  // float4 of the hidden tile per thread
    constexpr int CH = PER < 8 ? PER : 8;                        // ... of which in flight at once
    const int n_items = p.ts.n_items;
    const int n_grp = (n_items + NT - 1) / NT;                   // groups of NT tiles
    const int n_ws = p.nb_rg * NW;
    const int grp0 = wave * p.nb_rg + bir;
    const uint4* Wq = reinterpret_cast<const uint4*>(p.Wp);      // uniform base; the lane is the index
    const uint4* ldsq = reinterpret_cast<const uint4*>(lds4);
    const uint4 ones = bf16_ones_fragment(hi);
    // item of (group, nt), clamped to the group's first item when the last group is ragged
    auto item_of = [&](int grp, int nt) { const int i = NT * grp + nt; return i < n_items ? i : NT * grp; };

    // ---- prologue: the requests that depend on nothing go out together (tile ids, thresholds, the hidden tile), the W
    // ring as soon as the ids are here and the tile has been handed to LDS -- it streams from HBM while the workgroup
    // meets.  The straight order (tile, barrier, thresholds, ids, W) was one more dependent trip to memory before the
    // first MFMA; keeping the tile's registers live across the ring's loads made the compiler park them in scratch.
    // WHICH groups a wave decodes: its first two by position (wave w: the w-th and the (NW + w)-th group of the workgroup's
    // share {bir + m nb_rg}), every further one CLAIMED from a counter in LDS, two groups ahead of its use.  A wave that
    // drew one of the hottest tiles of the bias-ordered list (nearly every value passes: 16 - 20 k cycles of appends against
    // 4 k for an ordinary tile, stage stamps DAE_DBG_F) then simply takes fewer tiles, instead of setting the end of the
    // launch with the static share it had before (42 k cycles against 32 k for a workgroup without a hot tile).
    int t[NT], u[NT];                                            // tiles of this / the next group (uniform)
    int tv0[NT], tv1[NT];
    const bool has = grp0 < n_grp;
    {
        const int gn = grp0 + n_ws < n_grp ? grp0 + n_ws : grp0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            tv0[nt] = tile_of_item(p.ts, has ? item_of(grp0, nt) : 0);
            tv1[nt] = tile_of_item(p.ts, has ? item_of(gn, nt) : 0);
        }
    }
    float tau_g[(R_TILE + NTH - 1) / NTH];
#pragma unroll
    for (int e = 0; e < (R_TILE + NTH - 1) / NTH; ++e) {
        const int i = e * NTH + tid;
        tau_g[e] = (i < R_TILE && rg * R_TILE + i < p.B) ? p.tau[rg * R_TILE + i] : __builtin_inff();
    }
    int* lcnt = reinterpret_cast<int*>(lds4 + n_h4);
    float* ltau = reinterpret_cast<float*>(lcnt + R_TILE);
    int* claim = reinterpret_cast<int*>(ltau + R_TILE);          // next unclaimed group of the workgroup (in units of nb_rg)
    if (tid == 0) *claim = 2 * NW;
    {
        const float4* hsrc = p.hp + (size_t)rg * n_h4;
#pragma unroll
        for (int e0 = 0; e0 < PER; e0 += CH) {
            float4 hv[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int i = (e0 + e) * NTH + tid;
                hv[e] = hsrc[i < n_h4 ? i : n_h4 - 1];            // unconditional: a guarded load would push hv[] to scratch
            }
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int i = (e0 + e) * NTH + tid;
                if (e0 + e < PER && i < n_h4) lds4[i] = hv[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < (R_TILE + NTH - 1) / NTH; ++e) {
        const int i = e * NTH + tid;
        if (i < R_TILE) { lcnt[i] = 0; ltau[i] = tau_g[e]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    uint4 wq[NT][QR];
    uint4 cb[2][RB];
    uint4 bfr[NT];                                               // bias fragments of the NEXT group
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        t[nt] = __builtin_amdgcn_readfirstlane(tv0[nt]);
        u[nt] = __builtin_amdgcn_readfirstlane(tv1[nt]);
    }
    // (unconditional: a wave without a tile reads tile list[0]'s fragments and never uses them)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bfr[nt] = p.bias16[(size_t)t[nt] * 64 + lane];
#pragma unroll
    for (int k = 0; k < QR; ++k)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wq[nt][k] = Wq[(size_t)t[nt] * (NS * 64) + k * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    FSTAMP(1)                                                    // every request of the prologue is out
    __syncthreads();
    FSTAMP(2)                                                    // the workgroup has met

    float tau_r[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) tau_r[rb] = ltau[rb * 32 + j];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) cb[0][rb] = ldsq[rb * 64 + lane];

    int g_nxt = grp0 + n_ws;
    [[maybe_unused]] int n_done = 0;                            // (the stage stamps of the experiments build count tiles with it)
    for (int grp = grp0; grp < n_grp;) {
        int g_nn;
        {
            int n2 = 0;
            if (lane == 0) n2 = atomicAdd(claim, 1);
            g_nn = __builtin_amdgcn_readfirstlane(n2) * p.nb_rg + bir;
        }
        const int gnn = g_nn < n_grp ? g_nn : grp;                // (a group that exists: its tile id is read, never used)
        int wv[NT];
        const uint4 *cur[NT], *nxt[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            wv[nt] = tile_of_item(p.ts, item_of(gnn, nt));        // consumed at the end of this group
            cur[nt] = Wq + (size_t)t[nt] * (NS * 64);
            nxt[nt] = Wq + (size_t)u[nt] * (NS * 64);
        }

        // the accumulators start at the bias (see bf16_ones_fragment)
        f32x16 acc[NT][RB];
        {
            f32x16 zero;
#pragma unroll
            for (int e = 0; e < 16; ++e) zero[e] = 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const uint4 bc = bfr[nt];
                bfr[nt] = p.bias16[(size_t)u[nt] * 64 + lane];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc[nt][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(bc), as_bf16x8(ones), zero, 0, 0, 0);
            }
        }

        // One wave per SIMD issues in order: a burst of loads in front of the MFMAs would leave the
        // matrix pipe idle while the burst issues.  So every MFMA (32 cycles in the pipe) is followed
        // by ONE memory instruction of the prefetch -- the next step's hidden fragments from LDS, then
        // the W fragments of step s + QR -- and sched_barrier pins that order.
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int sn = (s + 1) % NS;
            uint4 a[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a[nt] = wq[nt][s % QR];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    acc[nt][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[nt]), as_bf16x8(cb[s & 1][rb]),
                                                                          acc[nt][rb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const int mi = nt * RB + rb;                      // MFMA index within the step
                    if (mi < RB) {                                    // next step's fragment rb = mi
                        cb[(s + 1) & 1][mi] = ldsq[(sn * RB + mi) * 64 + lane];
                    } else if (mi - RB < NT && (mi - RB) <= nt - 1) { // W of step s+QR for a tile already consumed
                        const int w = mi - RB;
                        wq[w][s % QR] = (s + QR < NS) ? cur[w][(s + QR) * 64 + lane] : nxt[w][(s + QR - NS) * 64 + lane];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // W fragments not yet refilled (their tile's MFMAs of this step had to finish first)
#pragma unroll
            for (int w = 0; w < NT; ++w) {
                const bool done_inline = (NT * RB > RB + w) && (w <= ((RB + w) / RB) - 1);
                if (!done_inline)
                    wq[w][s % QR] = (s + QR < NS) ? cur[w][(s + QR) * 64 + lane] : nxt[w][(s + QR - NS) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue: lane = playlist j of row block rb; register reg of tile t is column
        // 32 t + (reg & 3) + 8 (reg >> 2) + 4 hi.  The tiles of this launch are the LOW-bias ones:
        // most hold no value above tau at all, so the common case is a max-reduction and one compare
        // per row block; masks, list slots (LDS atomic) and stores only where a lane really passes.
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const float tv = tau_r[rb];
            float mx = acc[0][rb][0];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) mx = fmaxf(mx, acc[nt][rb][reg]);
            if (mx >= tv) {
                unsigned m[NT];
                int cnt = 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    m[nt] = 0;
                    const bool live = NT * grp + nt < n_items;            // ragged last group
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int lc = t[nt] * 32 + 4 * hi + (reg & 3) + 8 * (reg >> 2);
                        if (live && acc[nt][rb][reg] >= tv && lc < p.ncols && p.col_lo + lc < p.n_valid_col)
                            m[nt] |= 1u << reg;
                    }
                    cnt += __popc(m[nt]);
                }
                if (cnt) {
                    int at = atomicAdd(&lcnt[rb * 32 + j], cnt);
                    uint2* dst = p.cand + ((size_t)bir * p.Bpad + rg * R_TILE + rb * 32 + j) * (size_t)p.cap;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int cg = p.col_lo + t[nt] * 32 + 4 * hi;
                        // two neighbouring values that both pass leave as ONE 16-byte store: in the hottest tiles of the
                        // bias-ordered list nearly every value passes, and their 16 scattered 8-byte stores per lane and row
                        // block made those tiles cost 21 k cycles against 4 k (profiles/r03_notes.md) -- the launch's tail
#pragma unroll
                        for (int reg = 0; reg < 16; reg += 2) {
                            const unsigned two = (m[nt] >> reg) & 3u;
                            const uint2 e0 = make_uint2(__float_as_uint(acc[nt][rb][reg]), (unsigned)(cg + (reg & 3) + 8 * (reg >> 2)));
                            const uint2 e1 = make_uint2(__float_as_uint(acc[nt][rb][reg + 1]),
                                                        (unsigned)(cg + ((reg + 1) & 3) + 8 * ((reg + 1) >> 2)));
                            if (two == 3u) {
                                *reinterpret_cast<uint4*>(dst + at) = make_uint4(e0.x, e0.y, e1.x, e1.y);
                                at += 2;
                            } else if (two == 1u) {
                                dst[at++] = e0;
                            } else if (two == 2u) {
                                dst[at++] = e1;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { t[nt] = u[nt]; u[nt] = __builtin_amdgcn_readfirstlane(wv[nt]); }
        FSTAMP(3 + (n_done < 9 ? n_done : 9))                    // tile (group) done, epilogue included
        ++n_done;
        grp = g_nxt; g_nxt = g_nn;
    }
    FSTAMP(13)
    __syncthreads();
    FSTAMP(14)
    for (int i = tid; i < R_TILE; i += NW * 64) p.cand_cnt[(size_t)bir * p.Bpad + rg * R_TILE + i] = lcnt[i];
}
#undef FSTAMP

// ---- bf16, hidden = 256, filter epilogue, MFMA-bound batches (>= 512 playlists per launch) ------------------------------
// The kernel above reads every B operand (hidden fragment) of every MFMA from LDS: 1 KiB per v_mfma_f32_32x32x16_bf16,
// i.e. 128 B/clk for the CU's four matrix pipes at their peak rate -- exactly the LDS bandwidth, so LDS and matrix pipes are
// co-bound and the launch sits at 0.58 of the bf16 peak at batch 1024 (profiles/r03_notes.md).  Here the row group's
// hidden tile lives in REGISTERS: one wave per SIMD owns all 512 registers of a lane (256 VGPR + 256 AGPR on gfx950), 256 of
// them hold the 16 x 4 B fragments of the 128-row tile, and the main loop touches no LDS at all -- a W fragment from the
// 16-deep ring (a whole tile ahead), four MFMAs, one prefetch.  The epilogue of a tile (max-reduce + one compare per row
// block in the common case) is issued UNDER the first MFMAs of the next tile: two accumulator sets alternate.
// Same tiles, same operands, same accumulation as decode_bf16_h256_filter_kernel: the candidate lists are identical.
// MEASURED AND NOT THE DEFAULT (round 4, batch 1024, one batch in flight): 0.157 ms per step with two row blocks in
// registers against 0.146 ms for the all-LDS kernel, 0.225 ms with three (hipcc then spills fragments to scratch and reloads
// them every tile).  With one wave per SIMD hipcc issues a tile's 68 MFMAs back to back and the epilogue after them; the
// second wave per SIMD of the all-LDS kernel hides more than the LDS reads cost.  Experiments build only (DAE_BF16_REGB).
#ifdef DAE_EXPERIMENTS
template <int RBR, int QR>
__global__ __launch_bounds__(256, 1) void decode_bf16_h256_regb_filter_kernel(const DecP p)
{
    constexpr int NS = 16, RB = 4, RBL = RB - RBR, R_TILE = 128, NW = 4;
    __shared__ int lcnt[R_TILE];
    __shared__ uint4 hl[NS * RBL * 64];      // the fragments of the row blocks RBR .. 3 (16 KiB each): see below
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int j = lane & 31;
    const int gs = DAE_NUM_XCD * p.n_rg;
    const int q = blockIdx.x / gs, rem = blockIdx.x % gs;
    const int rg = rem / DAE_NUM_XCD;
    const int bir = q * DAE_NUM_XCD + (rem % DAE_NUM_XCD);
    const int n_items = p.ts.n_items;
    const int n_ws = p.nb_rg * NW;
    const int it0 = wave * p.nb_rg + bir;
    const uint4* Wq = reinterpret_cast<const uint4*>(p.Wp);
    const uint4 ones = bf16_ones_fragment(hi);

    // tile ids two ahead, the thresholds, the hidden tile -> registers, the first tile's W ring
    const bool has = it0 < n_items;
    const int tv0 = tile_of_item(p.ts, has ? it0 : 0);
    const int tv1 = tile_of_item(p.ts, has && it0 + n_ws < n_items ? it0 + n_ws : (has ? it0 : 0));
    float tau_r[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int row = rg * R_TILE + rb * 32 + j;
        tau_r[rb] = row < p.B ? p.tau[row] : __builtin_inff();
    }
    // Register budget of the lane (512): 3 of the 4 row blocks' fragments (192), two accumulator sets (128), a W ring one whole
    // tile deep (64: 2 176 cycles of prefetch distance).  The fourth row block's fragments would be 64 more -- hipcc then
    // parks 64 of them in scratch and reloads them every tile -- so they stay in LDS (16 KiB, shared by the four waves):
    // one ds_read_b128 per step and wave, a quarter of the LDS traffic that bounds the all-LDS kernel.
    uint4 hf[NS][RBR];
    {
        const uint4* hsrc = reinterpret_cast<const uint4*>(p.hp) + (size_t)rg * (NS * RB * 64);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int rb = 0; rb < RBR; ++rb) hf[s][rb] = hsrc[(s * RB + rb) * 64 + lane];
#pragma unroll
        for (int e = 0; e < NS * RBL * 64 / (NW * 64); ++e) {
            const int i = e * (NW * 64) + tid;                   // ((step, row block - RBR), lane)
            const int f = i >> 6;
            hl[i] = hsrc[((f / RBL) * RB + RBR + f % RBL) * 64 + (i & 63)];
        }
    }
    if (tid < R_TILE) lcnt[tid] = 0;
    int t = __builtin_amdgcn_readfirstlane(tv0), u = __builtin_amdgcn_readfirstlane(tv1);
    uint4 wq[QR];
    uint4 bfr = p.bias16[(size_t)t * 64 + lane];
#pragma unroll
    for (int s = 0; s < QR; ++s) wq[s] = Wq[(size_t)t * (NS * 64) + s * 64 + lane];
    __syncthreads();

    // the filter epilogue of one finished tile (see decode_bf16_h256_filter_kernel): `mx` = the lane's maxima per row block
    auto appends = [&](const f32x16 (&acc)[RB], const float (&mx)[RB], int tt) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const float tv = tau_r[rb];
            if (mx[rb] >= tv) {
                unsigned m = 0;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int lc = tt * 32 + 4 * hi + (reg & 3) + 8 * (reg >> 2);
                    if (acc[rb][reg] >= tv && lc < p.ncols && p.col_lo + lc < p.n_valid_col) m |= 1u << reg;
                }
                if (m) {
                    int at = atomicAdd(&lcnt[rb * 32 + j], __popc(m));
                    uint2* dst = p.cand + ((size_t)bir * p.Bpad + rg * R_TILE + rb * 32 + j) * (size_t)p.cap;
                    const int cg = p.col_lo + tt * 32 + 4 * hi;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        if (m & (1u << reg))
                            dst[at++] = make_uint2(__float_as_uint(acc[rb][reg]), (unsigned)(cg + (reg & 3) + 8 * (reg >> 2)));
                }
            }
        }
    };
    auto maxima = [&](const f32x16 (&acc)[RB], float (&mx)[RB]) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float m = acc[rb][0];
#pragma unroll
            for (int reg = 1; reg < 16; ++reg) m = fmaxf(m, acc[rb][reg]);
            mx[rb] = m;
        }
    };
    // one tile: accumulators start at the bias (through the matrix pipe, as every bf16 kernel), 16 steps; after step s the
    // ring slot s takes the same step of the NEXT tile
    auto tile = [&](f32x16 (&acc)[RB], int tc, int tn) {
        const uint4* cur = Wq + (size_t)tc * (NS * 64) + lane;
        const uint4* nxt = Wq + (size_t)tn * (NS * 64) + lane;
        const uint4 bc = bfr;
        bfr = p.bias16[(size_t)tn * 64 + lane];
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(bc), as_bf16x8(ones), zero, 0, 0, 0);
        uint4 cb[RBL];
#pragma unroll
        for (int r = 0; r < RBL; ++r) cb[r] = hl[r * 64 + lane];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint4 a = wq[s % QR];
            uint4 cbn[RBL];                                       // the LDS-resident fragments of the next step
#pragma unroll
            for (int r = 0; r < RBL; ++r) cbn[r] = hl[(((s + 1) % NS) * RBL + r) * 64 + lane];
#pragma unroll
            for (int rb = 0; rb < RBR; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(hf[s][rb]), acc[rb], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RBL; ++r) {
                acc[RBR + r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(cb[r]), acc[RBR + r], 0, 0, 0);
                cb[r] = cbn[r];
            }
            wq[s % QR] = (s + QR < NS) ? cur[(s + QR) * 64] : nxt[(s + QR - NS) * 64];
        }
    };

    f32x16 accA[RB], accB[RB];
    float mx[RB];
    int it = it0;
    int t_prev = 0;
    bool pending = false;                                        // accB / accA of the previous tile still wait for their epilogue
    // two tiles per trip: A then B; the epilogue of each runs after the MFMAs of the following tile have been issued
    while (it < n_items) {
        const int i1 = it + n_ws, i2 = i1 + n_ws;
        const int w1 = tile_of_item(p.ts, i2 < n_items ? i2 : it);      // ids two tiles ahead
        const int w2 = tile_of_item(p.ts, i2 + n_ws < n_items ? i2 + n_ws : it);
        tile(accA, t, u);                                        // tile t; the ring refills with tile u
        if (pending) { maxima(accB, mx); appends(accB, mx, t_prev); }
        const int tA = t;
        t = u; u = __builtin_amdgcn_readfirstlane(w1);
        if (i1 < n_items) {
            tile(accB, t, u);
            maxima(accA, mx); appends(accA, mx, tA);
            t_prev = t;
            t = u; u = __builtin_amdgcn_readfirstlane(w2);
            pending = true;
        } else {
            maxima(accA, mx); appends(accA, mx, tA);
            pending = false;
        }
        it = i2;
    }
    if (pending) { maxima(accB, mx); appends(accB, mx, t_prev); }
    __syncthreads();
    for (int i = tid; i < R_TILE; i += NW * 64) p.cand_cnt[(size_t)bir * p.Bpad + rg * R_TILE + i] = lcnt[i];
}

#endif  // DAE_EXPERIMENTS

// ---- prepack: W_dec rows -> MFMA A-operand order ----------------------------------------------
// One workgroup per 32-column tile: the tile's 32 rows of W (32 x H floats, contiguous 4 H bytes each) are read
// with coalesced 16-byte loads into LDS and written out in operand order with coalesced 16-byte stores.
// (A thread gathering its own 4 / 8 strided scalars straight from HBM took 139 us for the 174 MB matrix --
// 2.5 TB/s of traffic; the training step re-tiles the decoder every step.)
//   fp32: out float4 index = (t*G + g)*64 + lane, lane = hi*32 + i; component e = W[col_lo+32t+i][8g + 2e + hi]
//   bf16: out uint4  index = (t*NS + s)*64 + lane: bf16 of W[col_lo+32t+i][16s + 8hi + 0..7]
//         bias fragments: lane (hi = 0, i) of tile t carries b[col_lo + 32 t + i] = e0 + e1 + e2 in k-slots 0..2
// (zero outside the matrix)
constexpr int PP_PAD = 4;          // LDS row stride Hp + 4 floats: rows stay 16-byte aligned

template <int DT>
__global__ __launch_bounds__(256) void prepack_tile_kernel(const float* __restrict__ W,
                                                           const float* __restrict__ b, int H, int Hp,
                                                           int col_lo, int col_hi, int ntiles,
                                                           void* __restrict__ Wp_,
                                                           float* __restrict__ bias,
                                                           uint4* __restrict__ bias16)
{
    extern __shared__ __attribute__((aligned(16))) float pp_tile[];      // [32][Hp + PP_PAD]
    const int tid = threadIdx.x;
    const int ldt = Hp + PP_PAD;
    const int Hp4 = Hp >> 2;
    const bool vec = (H & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int v0 = col_lo + t * 32;
        for (int idx = tid; idx < 32 * Hp4; idx += 256) {
            const int r = idx / Hp4, c4 = idx - r * Hp4;
            const int v = v0 + r, k = 4 * c4;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < col_hi) {
                const float* src = W + (size_t)v * H + k;
                if (vec && k + 3 < H) {
                    x = *reinterpret_cast<const float4*>(src);
                } else {
                    if (k < H) x.x = src[0];
                    if (k + 1 < H) x.y = src[1];
                    if (k + 2 < H) x.z = src[2];
                    if (k + 3 < H) x.w = src[3];
                }
            }
            *reinterpret_cast<float4*>(pp_tile + r * ldt + k) = x;
        }
        __syncthreads();
        if (DT == DT_F32) {
            float4* Wp = static_cast<float4*>(Wp_);
            const int G = Hp >> 3;
            for (int o = tid; o < G * 64; o += 256) {
                const int lane = o & 63, g = o >> 6;
                const float* row = pp_tile + (lane & 31) * ldt + 8 * g + (lane >> 5);
                Wp[(size_t)t * G * 64 + o] = make_float4(row[0], row[2], row[4], row[6]);
            }
        } else {
            uint4* Wp = static_cast<uint4*>(Wp_);
            const int NS = Hp >> 4;
            for (int o = tid; o < NS * 64; o += 256) {
                const int lane = o & 63, sidx = o >> 6;
                const float* row = pp_tile + (lane & 31) * ldt + 16 * sidx + 8 * (lane >> 5);
                const float4 lo = *reinterpret_cast<const float4*>(row), hi4 = *reinterpret_cast<const float4*>(row + 4);
                Wp[(size_t)t * NS * 64 + o] =
                    make_uint4(bf16_rne(lo.x) | (bf16_rne(lo.y) << 16), bf16_rne(lo.z) | (bf16_rne(lo.w) << 16),
                               bf16_rne(hi4.x) | (bf16_rne(hi4.y) << 16), bf16_rne(hi4.z) | (bf16_rne(hi4.w) << 16));
            }
        }
        if (tid < 32) bias[t * 32 + tid] = v0 + tid < col_hi ? b[v0 + tid] : 0.0f;
        if (DT == DT_BF16 && tid < 64) {
            const int v = v0 + (tid & 31);
            uint4 f = make_uint4(0u, 0u, 0u, 0u);
            if ((tid >> 5) == 0 && v < col_hi) {
                const float bv = b[v];
                const unsigned e0 = bf16_rne(bv);
                const float r1 = bv - __uint_as_float(e0 << 16);
                const unsigned e1 = bf16_rne(r1);
                const float r2 = r1 - __uint_as_float(e1 << 16);
                const unsigned e2 = bf16_rne(r2);
                f.x = e0 | (e1 << 16); f.y = e2;
            }
            bias16[t * 64 + tid] = f;
        }
        __syncthreads();
    }
}

// ---- DAE_DTYPE_BF16_EXACT: per-column bound of |fp32 logit - bf16 logit| -----------------------------------------
// z32(r, c) = the canonical fp32 chain acc = fmaf(h[k], W[c][k], acc), + b[c]  (oracle orc_decode, DAEs.py:141-145)
// z16(r, c) = what the bf16 kernels of this file leave in an accumulator: bias terms e0 + e1 + e2 and the products
//             bf16(h[k]) * bf16(W[c][k]) (exact in fp32) summed by v_mfma_f32_32x32x16_bf16 in an unspecified order.
// With h[k] in [0, 1] (sigmoid outputs): |bf16(h) - h| <= 2^-9, bf16(h) <= 1, hence against the real-number value
//   | sum bf16(h) bf16(W) - sum h W | <= d_c + 2^-9 n_c,   d_c = sum_k |bf16(W[c][k]) - W[c][k]|,  n_c = sum_k |W[c][k]|
// (d_c is the rounding this image really made: on average a third of the worst case 2^-8 n_c);
//   accumulation, bf16 MFMA: every term runs through at most Hp + 3 additions of unknown order; an addition is taken
//     to err by <= 2^-23 relative (TWICE fp32's unit roundoff: covers a truncating adder), and the total is doubled
//     again: A16 = (Hp + 16) 2^-22 times the sum of the magnitudes (n_c + d_c + |b| + eps);
//     tests/test_gpu_exact.py pins the assumption: measured |z16 - exact| stays below a quarter of this term;
//   accumulation, fp32 chain: (H + 2) 2^-24 (1 + 2^-10) (n_c + |b|)   (standard recursive-summation bound, fma);
//   the three-term bf16 split of b -+ eps: exact to 2^-24 relative (taken as 2^-23).
// Everything in double, rounded away from b when stored.  One 256-thread workgroup per 32-column tile: 8 threads per
// column.  bias16_lo / bias16_hi: bias fragments (see prepack_tile_kernel) of b - eps and b + eps.
__device__ __forceinline__ uint4 bias_fragment(float bv)
{
    const unsigned e0 = bf16_rne(bv);
    const float r1 = bv - __uint_as_float(e0 << 16);
    const unsigned e1 = bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(e1 << 16);
    const unsigned e2 = bf16_rne(r2);
    return make_uint4(e0 | (e1 << 16), e2, 0u, 0u);
}

__global__ __launch_bounds__(256) void exact_bounds_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                           int H, int Hp, int col_lo, int col_hi, int ntiles,
                                                           float* __restrict__ eps, uint4* __restrict__ bias16_lo,
                                                           uint4* __restrict__ bias16_hi, float margin,
                                                           int m_lo, int m_hi, float m_scale)
{
    float* eps_max = eps + (size_t)ntiles * 32;          // zeroed by the launcher; positive floats order like their bits
    const int tid = threadIdx.x;
    const int c = tid >> 3, part = tid & 7;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int v = col_lo + t * 32 + c;
        double n = 0.0, d = 0.0;
        if (v < col_hi) {
            const float* row = W + (size_t)v * H;
            for (int k = part; k < H; k += 8) {
                const float w = row[k];
                const float w16 = __uint_as_float(bf16_rne(w) << 16);
                n += fabs((double)w);
                d += fabs((double)w16 - (double)w);
            }
        }
#pragma unroll
        for (int sh = 1; sh < 8; sh <<= 1) { n += __shfl_xor(n, sh); d += __shfl_xor(d, sh); }
        if (part == 0) {
            float e_f = 0.0f, lo_f = 0.0f, hi_f = 0.0f;
            if (v < col_hi) {
                const double bv = (double)b[v], ab = fabs(bv);
                const double A16 = (double)(Hp + 16) * 0x1p-22;
                const double A32 = (double)(H + 2) * 0x1p-24 * (1.0 + 0x1p-10);
                double e = d + 0x1p-9 * n + A16 * (n + d + 1.01 * ab) + A32 * (n + ab);
                e = e * (1.0 + 4.0 * A16) + 0x1p-23 * (ab + e) + 1e-30;      // eps feeds back through the shifted bias; split error
                e *= 1.0 + 1e-6;
                // dae_set_exact_margin: 1 by default; < 1 voids the bound (the guard's test hook; _range: for some columns only)
                const bool in_range = v >= m_lo && v < m_hi;
                e *= (double)(in_range && m_scale > 0.0f ? m_scale : margin);
                e_f = (float)e;
                if ((double)e_f < e) e_f = __uint_as_float(__float_as_uint(e_f) + 1u);      // e > 0: next float up
                double lo = bv - (double)e_f, hi = bv + (double)e_f;
                if (in_range && m_scale < 0.0f) hi = bv + (double)m_scale;     // (a FORGED filter: the upper bound |scale| logits low)
                lo_f = (float)lo; if ((double)lo_f > lo) lo_f = nextafterf(lo_f, -__builtin_inff());
                hi_f = (float)hi; if ((double)hi_f < hi) hi_f = nextafterf(hi_f, __builtin_inff());
            }
            eps[t * 32 + c] = e_f;
            if (e_f > 0.0f) atomicMax(reinterpret_cast<unsigned*>(eps_max), __float_as_uint(e_f));
            bias16_lo[t * 64 + c] = bias_fragment(lo_f);
            bias16_hi[t * 64 + c] = bias_fragment(hi_f);
            bias16_lo[t * 64 + 32 + c] = make_uint4(0u, 0u, 0u, 0u);
            bias16_hi[t * 64 + 32 + c] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

// ---- tile order for the fused path's threshold sample -------------------------------------------
// The sample only has to be SOME subset of the rankable columns (its k-th largest logit is a lower
// bound of the row's k-th largest whatever the subset), but the tighter that bound, the fewer
// candidates phase B has to keep.  Vocabulary ids are popularity ranks and the trained b_dec is the
// popularity prior, so the tiles with the largest bias hold most of every row's winners: sample
// those.  One workgroup: key = (ordered max bias over the tile's rankable columns, ~tile) sorted
// descending by a bitonic network in LDS.
constexpr int ORDER_MAX_TILES = 8192;
__global__ __launch_bounds__(1024) void tile_order_kernel(const float* __restrict__ bias, int ntiles,
                                                          int nrank, int* __restrict__ order)
{
    __shared__ unsigned long long keys[ORDER_MAX_TILES];
    int n2 = 1024;
    while (n2 < ntiles) n2 <<= 1;
    for (int i = threadIdx.x; i < n2; i += 1024) {
        unsigned long long k = 0ULL;                       // padding sorts last
        if (i < ntiles) {
            float m = -__builtin_inff();
            for (int c = 0; c < 32; ++c) {
                const int col = i * 32 + c;
                if (col < nrank) m = fmaxf(m, bias[col]);
            }
            k = ((unsigned long long)dae_okey(m) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < n2; i += 1024) {
                const int jx = i ^ stride;
                if (jx > i) {
                    const unsigned long long a = keys[i], b = keys[jx];
                    const bool desc = (i & size) == 0;
                    if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[jx] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < ntiles; i += 1024)
        order[i] = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFULL));
}

__global__ __launch_bounds__(256) void tile_iota_kernel(int n, int* __restrict__ out)
{
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) out[t] = t;
}

// fallback for images of more than ORDER_MAX_TILES tiles (and the DAE_SAMPLE=strided experiment):
// every S-th tile first, then the others
// The threshold sample re-dealt for a launch of several ROUNDS (dae_launch_tile_band).  Phase A takes, per (row, position in the
// tile), the maximum over the `waves` tiles a workgroup decodes together in a round; the threshold is the (k + seeds)-th largest
// of these maxima, so two winners in one group cost one of them.  Item i of the sample goes to round i / n_ws, wave (i % n_ws) /
// nb_rg, workgroup i % nb_rg: with ONE round the group's tiles sit nb_rg places apart in the bias order (128 at batch 256) --
// different popularity bands; with ten rounds (2 048 rows: 16 workgroups per row group) they sit 16 apart, round 0 is the 64 most
// popular tiles in 16 groups of 4, ~550 winners share 512 maxima, the threshold drops into the next round's maxima and 1 155
// candidates per row pass instead of 534.  Here wave w's items (all rounds, all workgroups) take the w-th band of the order.
__global__ __launch_bounds__(256) void tile_band_kernel(const int* __restrict__ order, int ntiles, int n_samp, int nb_rg,
                                                        int waves, int* __restrict__ band)
{
    const int n_ws = nb_rg * waves;
    const int R = n_samp / n_ws, rem_last = n_samp - R * n_ws;
    for (int it = blockIdx.x * 256 + threadIdx.x; it < ntiles; it += gridDim.x * 256) {
        if (it >= n_samp) { band[it] = order[it]; continue; }
        const int round = it / n_ws, rem = it - round * n_ws, w = rem / nb_rg, bir = rem - w * nb_rg;
        int rank = round * nb_rg + bir;                       // items of wave w in front of this one: every lower (round, bir) exists
        for (int wp = 0; wp < w; ++wp) {                      // + all items of the waves before it
            int last = rem_last - wp * nb_rg;
            last = last < 0 ? 0 : (last > nb_rg ? nb_rg : last);
            rank += R * nb_rg + last;
        }
        band[it] = order[rank];
    }
}

__global__ __launch_bounds__(256) void tile_order_strided_kernel(int ntiles, int n_samp, int S,
                                                                 int* __restrict__ order)
{
    for (int t = blockIdx.x * 256 + threadIdx.x; t < ntiles; t += gridDim.x * 256) {
        if (t % S == 0) order[t / S] = t;
        else order[n_samp + (t / S) * (S - 1) + (t % S) - 1] = t;
    }
}

// ---- pack h [B,H] -> MFMA B-operand order per row group ---------------------------------------
// out float4 index = ((rg*G + g)*RB + rb)*64 + lane, lane = hi*32 + j; component e holds
// h[rg*R_TILE + rb*32 + j][8g + 2e + hi]  (zero outside).
__global__ __launch_bounds__(256) void pack_h_kernel(const float* __restrict__ h, int B, int H,
                                                     int G, int RB, int n_rg,
                                                     float4* __restrict__ hp)
{
    const size_t total = (size_t)n_rg * G * RB * 64;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total;
         o += (size_t)gridDim.x * 256) {
        const int lane = (int)(o & 63);
        size_t x = o >> 6;
        const int rb = (int)(x % RB); x /= RB;
        const int g = (int)(x % G);
        const int rg = (int)(x / G);
        const int hi = lane >> 5, jj = lane & 31;
        const int r = (rg * RB + rb) * 32 + jj;
        float e[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = 8 * g + 2 * c + hi;
            e[c] = (r < B && k < H) ? h[(size_t)r * H + k] : 0.0f;
        }
        hp[o] = make_float4(e[0], e[1], e[2], e[3]);
    }
}

// out uint4 index = ((rg*NS + s)*RB + rb)*64 + lane: bf16 of h[(rg*RB+rb)*32+j][16s+8hi+0..7]
// row_bad (nullable, zeroed by the launcher): set to 1 for rows with an entry outside [0, 1] (or NaN) -- the
// precondition of DAE_DTYPE_BF16_EXACT's bound
__global__ __launch_bounds__(256) void pack_h_bf16_kernel(const float* __restrict__ h, int B, int H,
                                                          int NS, int RB, int n_rg,
                                                          uint4* __restrict__ hp, int* __restrict__ row_bad)
{
    const size_t total = (size_t)n_rg * NS * RB * 64;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int lane = (int)(o & 63);
        size_t x = o >> 6;
        const int rb = (int)(x % RB); x /= RB;
        const int s = (int)(x % NS);
        const int rg = (int)(x / NS);
        const int hi = lane >> 5, jj = lane & 31;
        const int r = (rg * RB + rb) * 32 + jj;
        unsigned e[8];
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = 16 * s + 8 * hi + c;
            const float hv = (r < B && k < H) ? h[(size_t)r * H + k] : 0.0f;
            bad = bad || !(hv >= 0.0f && hv <= 1.0f);
            e[c] = bf16_rne(hv);
        }
        if (row_bad && bad) row_bad[r] = 1;
        hp[o] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    }
}

template <int RB, int EPI, int GT, int NW, int DT>
int launch_decode(dae_ctx* ctx, const dae_rowgeom& g, const DecP& p)
{
    const size_t lds = (size_t)RB * 64 * p.G * sizeof(float4) + (size_t)RB * 32 * sizeof(int) +
                       (EPI == EPI_GMAX ? (size_t)NW * 256 * sizeof(float4) : 0) +
                       (EPI == EPI_FILTER ? (size_t)RB * 32 * sizeof(float) : 0);
    static const char attr_set_key = 0;     // per template instantiation
    if (dae_first_use(ctx, &attr_set_key)) {
        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(
                               reinterpret_cast<const void*>(&decode_f32_kernel<RB, EPI, GT, NW, DT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (ctx->prof_armed) {
        hipExtLaunchKernelGGL((decode_f32_kernel<RB, EPI, GT, NW, DT>), dim3(g.grid), dim3(NW * 64), lds,
                              ctx->stream, ctx->prof_ev[ctx->prof_used], ctx->prof_ev[ctx->prof_used + 1], 0, p);
        ctx->prof_armed = false;
        ctx->prof_used += 2;
        char name[96];
        snprintf(name, sizeof(name), "decode_f32_kernel<%d, %d, %d, %d, %d>", RB, EPI, GT, NW, DT);
        ctx->prof_kernel = name;
    } else {
        hipLaunchKernelGGL((decode_f32_kernel<RB, EPI, GT, NW, DT>), dim3(g.grid), dim3(NW * 64), lds,
                           ctx->stream, p);
    }
    DAE_CHECK_LAUNCH(ctx, "decode_f32_kernel");
    return DAE_OK;
}

template <int EPI>
int launch_decode_rb(dae_ctx* ctx, const dae_rowgeom& g, const DecP& p)
{
    // the shipped configs all use hidden = 256 (config.ini:12): G = 32 gets the unrolled body
    if (g.R_TILE == 128 && p.G == 32) {
        return launch_decode<4, EPI, 32, 4, DT_F32>(ctx, g, p);
    }
    if (g.waves != 4) return dae_fail(ctx, DAE_ERR_ARG, "bad wave count %d", g.waves);
    switch (g.R_TILE) {
        case 128: return launch_decode<4, EPI, 0, 4, DT_F32>(ctx, g, p);
        case 64:  return launch_decode<2, EPI, 0, 4, DT_F32>(ctx, g, p);
        case 32:  return launch_decode<1, EPI, 0, 4, DT_F32>(ctx, g, p);
    }
    return dae_fail(ctx, DAE_ERR_ARG, "bad R_TILE %d", g.R_TILE);
}

template <int EPI>
int launch_decode_rb_bf16(dae_ctx* ctx, const dae_rowgeom& g, const DecP& p)
{
    // hidden = 256 -> 16 steps of K = 16: unrolled body with the 8-deep register ring
    // two waves per SIMD here: with 16x faster MFMAs the VALU epilogue of a tile is comparable to
    // its matrix time, and the second wave's MFMAs cover it (DAE_DECODE_WAVES_BF16=4 for the A/B)
    if (g.waves != 4) return dae_fail(ctx, DAE_ERR_ARG, "bad wave count %d", g.waves);
    if (g.R_TILE == 256 && p.G == 16) return launch_decode<8, EPI, 16, 4, DT_BF16>(ctx, g, p);
    if (g.R_TILE == 128 && p.G == 16) return launch_decode<4, EPI, 16, 4, DT_BF16>(ctx, g, p);
    switch (g.R_TILE) {
        case 128: return launch_decode<4, EPI, 0, 4, DT_BF16>(ctx, g, p);
        case 64:  return launch_decode<2, EPI, 0, 4, DT_BF16>(ctx, g, p);
        case 32:  return launch_decode<1, EPI, 0, 4, DT_BF16>(ctx, g, p);
    }
    return dae_fail(ctx, DAE_ERR_ARG, "bad R_TILE %d", g.R_TILE);
}

bool bf16_pair_variant()
{
    static const bool v = dae_exp_env("DAE_BF16_PAIR") != nullptr;     // A/B: two column tiles per wave, one wave per SIMD
    return v;
}
// the dedicated phase-B kernel: hidden = 256 (16 steps), one wave per SIMD, 128- or 256-row groups
bool bf16_fast_filter(const dae_rowgeom& g, int dtype, int G)
{
    static const bool off = dae_exp_env("DAE_BF16_GENERIC") != nullptr;                // A/B against the generic body
    return dtype == DAE_DTYPE_BF16 && G == 16 && g.waves == 4 && (g.R_TILE == 128 || g.R_TILE == 256) && !off;
}

// ... and its MFMA-bound form (hidden fragments in registers, decode_bf16_h256_regb_filter_kernel): launches of four or
// more row groups (>= 385 playlists), where W passes through every CU's L1 once per row group and the all-LDS kernel
// is bound by its LDS reads.  0 = off; 2 / 3 = row blocks held in registers (experiments: DAE_BF16_REGB).
int bf16_regb_variant(const dae_rowgeom& g)
{
    static const int env = dae_exp_env("DAE_BF16_REGB") ? atoi(dae_exp_env("DAE_BF16_REGB")) : -1;
    if (g.R_TILE != 128) return 0;
    return (env == 2 || env == 3) ? env : 0;                 // measured slower than the all-LDS kernel: never the default
}

int fill_common(dae_ctx* ctx, const dae_rowgeom& g, int B, const dae_tileset& ts, DecP& p,
                int dtype = DAE_DTYPE_F32, int bias_sel = 0)
{
    const dae_packed& pk = dtype == DAE_DTYPE_F32 ? ctx->pk_f32 : ctx->pk_bf16;
    const dae_buf& hb = dtype == DAE_DTYPE_F32 ? ctx->h_packed : ctx->h_packed16;
    if (!pk.valid) return dae_fail(ctx, DAE_ERR_STATE, "decoder not prepacked (dtype %d)", dtype);
    if (!hb.p) return dae_fail(ctx, DAE_ERR_STATE, "hidden tile not packed");
    memset(&p, 0, sizeof(p));
    p.Wp = static_cast<const float4*>(pk.W.p);
    p.bias = static_cast<const float*>(pk.bias.p);
    p.bias16 = static_cast<const uint4*>(pk.bias16.p);
    if (bias_sel) {                                        // DAE_DTYPE_BF16_EXACT: bounds instead of the logits
        if (dtype != DAE_DTYPE_BF16 || !pk.exact)
            return dae_fail(ctx, DAE_ERR_STATE, "decoder not prepacked with DAE_DTYPE_BF16_EXACT");
        p.bias16 = static_cast<const uint4*>(bias_sel == 1 ? pk.bias16_lo.p : pk.bias16_hi.p);
    }
    p.hp = static_cast<const float4*>(hb.p);
    p.G = dtype == DAE_DTYPE_F32 ? pk.Hp / DAE_KG : pk.Hp / 16;
    p.ncols = pk.col_hi - pk.col_lo;
    p.col_lo = pk.col_lo;
    p.B = B; p.n_rg = g.n_rg; p.nb_rg = g.nb_rg; p.Bpad = g.Bpad;
    p.ts = ts;
    p.mixT = ctx->mixT; p.mix_ld = ctx->mix_ld; p.mix_w = ctx->mix_w; p.mix_ncols = ctx->mix_ncols;      // dae_set_score_mix (GMAX / FILTER epilogues)
    return DAE_OK;
}

}  // namespace

// most tiles one workgroup of the filter launch can walk (sizes its private candidate lists)
int dae_filter_block_tiles(const dae_rowgeom& g, int n_items, int dtype, int Hp, bool mixed)
{
    const int n_ws = g.nb_rg * g.waves;
    if (bf16_fast_filter(g, dtype, Hp / 16) && !mixed) {
        const bool pair = g.R_TILE != 256 && bf16_pair_variant();
        const int nt = pair ? 2 : 1;
        const int nw = (g.R_TILE == 256 || pair || bf16_regb_variant(g)) ? 4 : 8;
        const int n_grp = (n_items + nt - 1) / nt;
        const int n_ws2 = g.nb_rg * nw;
        return nt * nw * ((n_grp + n_ws2 - 1) / n_ws2);
    }
    return g.waves * ((n_items + n_ws - 1) / n_ws);
}

// Rows are cut into groups of R_TILE playlists whose hidden tile (R_TILE x Hp fp32) stays in LDS.
dae_rowgeom dae_row_geometry(int B, int Hp)
{
    dae_rowgeom g;
    int rt = 128;
    while (rt > 32 && (size_t)rt * Hp * 4 > 128 * 1024) rt >>= 1;   // <= 128 KiB of LDS
    while (rt > 32 && B <= rt / 2) rt >>= 1;                        // small batches
    g.R_TILE = rt;
    g.n_rg = (B + rt - 1) / rt;
    g.Bpad = g.n_rg * rt;
    int nb = (DAE_NUM_CU / g.n_rg) / DAE_NUM_XCD * DAE_NUM_XCD;
    if (nb < DAE_NUM_XCD) nb = DAE_NUM_XCD;
    g.nb_rg = nb;
    g.grid = g.n_rg * nb;
    // one wave per SIMD: with 4 independent accumulators it saturates the fp32 matrix pipe (two per SIMD
    // measured slower, profiles/r01_notes.md)
    g.waves = 4;
    return g;
}

// bf16: 128-playlist tiles (64 KiB of LDS at H = 256).  256-playlist tiles fit LDS too but need
// 394 registers per wave, which rules out the second wave per SIMD that hides the epilogue.
dae_rowgeom dae_row_geometry_bf16(int B, int Hp)
{
    dae_rowgeom g;
    // 128-playlist row groups.  256-row groups (every W fragment feeds 8 MFMAs, a batch of 256 reads W
    // exactly once) measured the same kernel time but a slower phase A: DAE_BF16_RTILE=256 for the A/B.
    int rt = 128;
    if (const char* e = dae_exp_env("DAE_BF16_RTILE")) { if (atoi(e) == 256 && Hp == 256 && B > 128) rt = 256; }
    while (rt > 32 && (size_t)rt * Hp * 2 > 128 * 1024) rt >>= 1;
    while (rt > 32 && B <= rt / 2) rt >>= 1;
    g.R_TILE = rt;
    g.n_rg = (B + rt - 1) / rt;
    g.Bpad = g.n_rg * rt;
    int nb = (DAE_NUM_CU / g.n_rg) / DAE_NUM_XCD * DAE_NUM_XCD;
    if (nb < DAE_NUM_XCD) nb = DAE_NUM_XCD;
    g.nb_rg = nb;
    g.grid = g.n_rg * nb;
    g.waves = 4;
    return g;
}

namespace {
template <int DT>
int launch_prepack_tiles(dae_ctx* ctx, const float* W, const float* b, int H, int Hp, int col_lo, int col_hi,
                         int ntiles, void* Wp, float* bias, uint4* bias16)
{
    if (ntiles <= 0) return DAE_OK;
    const size_t lds = (size_t)32 * (Hp + PP_PAD) * sizeof(float);
    static const char attr_set_key = 0;     // per instantiation
    if (dae_first_use(ctx, &attr_set_key)) {
        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&prepack_tile_kernel<DT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int blocks = ntiles < 8 * DAE_NUM_CU ? ntiles : 8 * DAE_NUM_CU;
    hipLaunchKernelGGL(prepack_tile_kernel<DT>, dim3(blocks), dim3(256), lds, ctx->stream, W, b, H, Hp, col_lo, col_hi,
                       ntiles, Wp, bias, bias16);
    DAE_CHECK_LAUNCH(ctx, "prepack_tile_kernel");
    return DAE_OK;
}
}  // namespace

int dae_launch_prepack_bf16(dae_ctx* ctx, const float* W, const float* b, int V, int H,
                            int col_lo, int col_hi, int exact)
{
    dae_packed& pk = ctx->pk_bf16;
    pk.valid = false; pk.order_nrank = -1; pk.exact = false;
    if (exact && (H & 3)) return dae_fail(ctx, DAE_ERR_ARG, "DAE_DTYPE_BF16_EXACT needs H %% 4 == 0 (H=%d)", H);
    const int Hp = dae_round_up(H, DAE_HPAD);
    if ((size_t)32 * Hp * 2 > 128 * 1024)
        return dae_fail(ctx, DAE_ERR_ARG, "hidden size %d too large", H);
    const int ntiles = (col_hi - col_lo + DAE_VT - 1) / DAE_VT;
    const int NS = Hp / 16;
    int rc = dae_reserve(ctx, pk.W, (size_t)ntiles * NS * 64 * sizeof(uint4));
    if (rc) return rc;
    rc = dae_reserve(ctx, pk.bias, (size_t)ntiles * 32 * sizeof(float));
    if (rc) return rc;
    rc = dae_reserve(ctx, pk.bias16, (size_t)ntiles * 64 * sizeof(uint4));
    if (rc) return rc;
    rc = launch_prepack_tiles<DT_BF16>(ctx, W, b, H, Hp, col_lo, col_hi, ntiles, pk.W.p,
                                       static_cast<float*>(pk.bias.p), static_cast<uint4*>(pk.bias16.p));
    if (rc) return rc;
    pk.V = V; pk.H = H; pk.Hp = Hp; pk.col_lo = col_lo; pk.col_hi = col_hi; pk.ntiles = ntiles;
    rc = dae_reserve(ctx, pk.ident, (size_t)(ntiles > 0 ? ntiles : 1) * sizeof(int));
    if (rc) return rc;
    hipLaunchKernelGGL(tile_iota_kernel, dim3((ntiles + 255) / 256 > 0 ? (ntiles + 255) / 256 : 1), dim3(256), 0,
                       ctx->stream, ntiles, static_cast<int*>(pk.ident.p));
    DAE_CHECK_LAUNCH(ctx, "tile_iota_kernel");
    if (exact) {
        rc = dae_reserve(ctx, pk.eps, ((size_t)ntiles * 32 + 1) * sizeof(float));
        if (rc) return rc;
        DAE_HIP_CHECK(ctx, hipMemsetAsync(static_cast<float*>(pk.eps.p) + (size_t)ntiles * 32, 0, sizeof(float), ctx->stream));
        rc = dae_reserve(ctx, pk.bias16_lo, (size_t)ntiles * 64 * sizeof(uint4));
        if (rc) return rc;
        rc = dae_reserve(ctx, pk.bias16_hi, (size_t)ntiles * 64 * sizeof(uint4));
        if (rc) return rc;
        const size_t wbytes = (size_t)(col_hi - col_lo) * H * sizeof(float);
        rc = dae_reserve(ctx, pk.W32, wbytes);
        if (rc) return rc;
        const int blocks = ntiles < 8 * DAE_NUM_CU ? ntiles : 8 * DAE_NUM_CU;
        hipLaunchKernelGGL(exact_bounds_kernel, dim3(blocks), dim3(256), 0, ctx->stream, W, b, H, Hp, col_lo, col_hi,
                           ntiles, static_cast<float*>(pk.eps.p), static_cast<uint4*>(pk.bias16_lo.p),
                           static_cast<uint4*>(pk.bias16_hi.p), ctx->exact_margin, ctx->margin_lo, ctx->margin_hi, ctx->margin_scale);
        DAE_CHECK_LAUNCH(ctx, "exact_bounds_kernel");
        DAE_HIP_CHECK(ctx, hipMemcpyAsync(pk.W32.p, W + (size_t)col_lo * H, wbytes, hipMemcpyDeviceToDevice, ctx->stream));
        // the same image read as the title side of the exact title mix: row-scaled bounds (mixexact.hip)
        rc = dae_launch_mix_title_bounds(ctx, W, b, H, Hp, col_lo, col_hi, ntiles, pk);
        if (rc) return rc;
        pk.exact = true;
    }
    pk.valid = true;
    return DAE_OK;
}

int dae_launch_pack_h_bf16(dae_ctx* ctx, const float* h, int B, int H, const dae_rowgeom& g, int* row_bad)
{
    const int Hp = dae_round_up(H, DAE_HPAD);
    const int NS = Hp / 16, RB = g.R_TILE / 32;
    const size_t total = (size_t)g.n_rg * NS * RB * 64;
    int rc = dae_reserve(ctx, ctx->h_packed16, total * sizeof(uint4));
    if (rc) return rc;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (row_bad) DAE_HIP_CHECK(ctx, hipMemsetAsync(row_bad, 0, (size_t)g.Bpad * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(pack_h_bf16_kernel, dim3(blocks), dim3(256), 0, ctx->stream, h, B, H, NS, RB,
                       g.n_rg, static_cast<uint4*>(ctx->h_packed16.p), row_bad);
    DAE_CHECK_LAUNCH(ctx, "pack_h_bf16_kernel");
    ctx->h16_geom_key = ((long long)B << 32) | ((long long)H << 12) | (long long)g.R_TILE;   // whole image rewritten, pads zero
    ctx->h16_geom_ptr = ctx->h_packed16.p;
    return DAE_OK;
}

int dae_launch_prepack_f32(dae_ctx* ctx, const float* W, const float* b, int V, int H,
                           int col_lo, int col_hi)
{
    dae_packed& pk = ctx->pk_f32;
    pk.valid = false; pk.order_nrank = -1;
    const int Hp = dae_round_up(H, DAE_HPAD);
    if ((size_t)32 * Hp * 4 > 128 * 1024)
        return dae_fail(ctx, DAE_ERR_ARG, "hidden size %d too large (max 1024)", H);
    const int ntiles = (col_hi - col_lo + DAE_VT - 1) / DAE_VT;
    const int G = Hp / DAE_KG;
    int rc = dae_reserve(ctx, pk.W, (size_t)ntiles * G * 64 * sizeof(float4));
    if (rc) return rc;
    rc = dae_reserve(ctx, pk.bias, (size_t)ntiles * 32 * sizeof(float));
    if (rc) return rc;
    rc = launch_prepack_tiles<DT_F32>(ctx, W, b, H, Hp, col_lo, col_hi, ntiles, pk.W.p,
                                      static_cast<float*>(pk.bias.p), nullptr);
    if (rc) return rc;
    pk.V = V; pk.H = H; pk.Hp = Hp; pk.col_lo = col_lo; pk.col_hi = col_hi; pk.ntiles = ntiles;
    rc = dae_reserve(ctx, pk.ident, (size_t)(ntiles > 0 ? ntiles : 1) * sizeof(int));
    if (rc) return rc;
    hipLaunchKernelGGL(tile_iota_kernel, dim3((ntiles + 255) / 256 > 0 ? (ntiles + 255) / 256 : 1), dim3(256), 0,
                       ctx->stream, ntiles, static_cast<int*>(pk.ident.p));
    DAE_CHECK_LAUNCH(ctx, "tile_iota_kernel");
    pk.valid = true;
    return DAE_OK;
}

int dae_launch_tile_iota(dae_ctx* ctx, int* dst, int ntiles)
{
    hipLaunchKernelGGL(tile_iota_kernel, dim3((ntiles + 255) / 256 > 0 ? (ntiles + 255) / 256 : 1), dim3(256), 0,
                       ctx->stream, ntiles, dst);
    DAE_CHECK_LAUNCH(ctx, "tile_iota_kernel");
    return DAE_OK;
}

int dae_launch_tile_order(dae_ctx* ctx, dae_packed& pk, int nrank, int n_samp, int S)
{
    if (pk.order_nrank == nrank && pk.order_nsamp == n_samp && pk.order.p) return DAE_OK;
    int rc = dae_reserve(ctx, pk.order, (size_t)pk.ntiles * sizeof(int));
    if (rc) return rc;
    static const bool strided = dae_exp_env("DAE_SAMPLE") && !strcmp(dae_exp_env("DAE_SAMPLE"), "strided");
    if (pk.ntiles > ORDER_MAX_TILES || strided) {
        hipLaunchKernelGGL(tile_order_strided_kernel, dim3((pk.ntiles + 255) / 256), dim3(256), 0, ctx->stream,
                           pk.ntiles, n_samp, S, static_cast<int*>(pk.order.p));
    } else {
        hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                           static_cast<const float*>(pk.bias.p), pk.ntiles, nrank, static_cast<int*>(pk.order.p));
    }
    DAE_CHECK_LAUNCH(ctx, "tile_order_kernel");
    pk.order_nrank = nrank; pk.order_nsamp = n_samp;
    static std::atomic<long long> gen{0};
    pk.order_gen = ++gen;
    return DAE_OK;
}

// phase A with per-WAVE group maxima (decode_bf16_h256_wavemax_kernel): bf16 image of hidden 256, 128-row groups, and a sample
// that gives each of the 8 wave slots per workgroup at least two tiles
bool dae_sample_wave_groups(const dae_rowgeom& g, int Hp, int n_samp)
{
    static const bool off = dae_exp_env("DAE_NO_WAVEMAX") != nullptr;                    // A/B (experiments build)
    return !off && Hp == 256 && g.R_TILE == 128 && g.waves == 4 && n_samp >= 2 * g.nb_rg * 8;
}

int dae_launch_tile_band(dae_ctx* ctx, const int* order, int ntiles, int n_samp, int nb_rg, int waves, int* band)
{
    if (ntiles <= 0) return DAE_OK;
    hipLaunchKernelGGL(tile_band_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, ctx->stream, order, ntiles, n_samp, nb_rg,
                       waves, band);
    DAE_CHECK_LAUNCH(ctx, "tile_band_kernel");
    return DAE_OK;
}

int dae_launch_pack_h(dae_ctx* ctx, const float* h, int B, int H, const dae_rowgeom& g)
{
    const int Hp = dae_round_up(H, DAE_HPAD);
    const int G = Hp / DAE_KG, RB = g.R_TILE / 32;
    const size_t total = (size_t)g.n_rg * G * RB * 64;
    int rc = dae_reserve(ctx, ctx->h_packed, total * sizeof(float4));
    if (rc) return rc;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_h_kernel, dim3(blocks), dim3(256), 0, ctx->stream, h, B, H, G, RB,
                       g.n_rg, static_cast<float4*>(ctx->h_packed.p));
    DAE_CHECK_LAUNCH(ctx, "pack_h_kernel");
    return DAE_OK;
}

int dae_launch_decode_dense_f32(dae_ctx* ctx, const dae_rowgeom& g, int B, const dae_tileset& ts,
                                int apply_sigmoid, int mask_from_col, float* out, int64_t ld,
                                int fill_pad, int dtype, float* gmax, int64_t ld_gmax, int gmax_per_wave, int bias_sel)
{
    DecP p;
    int rc = fill_common(ctx, g, B, ts, p, dtype, bias_sel);
    if (rc) return rc;
    p.out = out; p.ld = ld; p.apply_sigmoid = apply_sigmoid; p.mask_from_col = mask_from_col;
    p.gmax = gmax; p.ld_gmax = ld_gmax; p.gmax_per_wave = gmax_per_wave;
    if (!out && !gmax) return dae_fail(ctx, DAE_ERR_ARG, "dense decode without an output");
#ifdef DAE_EXPERIMENTS
    static const bool dbgA = dae_exp_env("DAE_DBG_A") != nullptr;
    static long long* abuf = nullptr;
    static int acalls = 0;
    if (dbgA && gmax) {
        if (!abuf) (void)hipMalloc(&abuf, 8 * 8);
        p.stamps = abuf;
        if ((++acalls % 50) == 0) {
            long long h[8];
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemcpy(h, abuf, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "PHASE_A wg0: fill %lld  prologue+kloop %lld  dense epi %lld  gmax %lld\n", h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3]);
        }
    }
#endif
    // fill_pad: the (internal) buffer covers whole tiles; columns past the image get -inf
    p.fill_pad = (out && fill_pad && ld >= (int64_t)ts.n_items * 32) ? 1 : 0;
    p.vec_ok = ((ld % 4) == 0 && (reinterpret_cast<uintptr_t>(out) % 16) == 0) ? 1 : 0;
    if (gmax && (gmax_per_wave == 3 || gmax_per_wave == 4)) {
        // per-wave group maxima (the caller sized gmax for 8 wave slots per workgroup: dae_sample_wave_groups)
        if (dtype != DAE_DTYPE_BF16 || out || p.G != 16 || g.R_TILE != 128 || p.mixT)
            return dae_fail(ctx, DAE_ERR_ARG, "per-wave group maxima: bf16, hidden 256, 128-row groups, maxima only");
        const size_t lds = (size_t)4 * 64 * 16 * sizeof(float4);
        static const char wm_key = 0;
        if (dae_first_use(ctx, &wm_key))
            DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_bf16_h256_wavemax_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(decode_bf16_h256_wavemax_kernel, dim3(g.grid), dim3(512), lds, ctx->stream, p);
        DAE_CHECK_LAUNCH(ctx, "decode_bf16_h256_wavemax_kernel");
        return DAE_OK;
    }
    if (gmax) {
        if (g.waves != 4) return dae_fail(ctx, DAE_ERR_ARG, "group maxima need 4-wave workgroups");
        static const bool no_half = dae_exp_env("DAE_GMAX_FULL") != nullptr;             // A/B against one workgroup per CU
        if (dtype == DAE_DTYPE_F32 && g.R_TILE == 128 && p.G == 32 && !gmax_per_wave && !p.mixT && !no_half &&
            ts.n_items <= g.nb_rg * 4) {
            // one round of tiles (the threshold sample at batch <= 256): half row groups, two workgroups per CU
            p.n_rg = 2 * g.n_rg;
            const size_t lds = (size_t)2 * 64 * p.G * sizeof(float4) + (size_t)2 * 32 * sizeof(int);
            static const char attr_key = 0;
            if (dae_first_use(ctx, &attr_key))
                DAE_HIP_CHECK(ctx, hipFuncSetAttribute(
                                       reinterpret_cast<const void*>(&decode_f32_kernel<2, EPI_GMAX, 32, 4, DT_F32, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL((decode_f32_kernel<2, EPI_GMAX, 32, 4, DT_F32, 1>), dim3(2 * g.grid), dim3(256), lds,
                               ctx->stream, p);
            DAE_CHECK_LAUNCH(ctx, "decode_f32_kernel (half row groups)");
            return DAE_OK;
        }
        return dtype == DAE_DTYPE_F32 ? launch_decode_rb<EPI_GMAX>(ctx, g, p) : launch_decode_rb_bf16<EPI_GMAX>(ctx, g, p);
    }
    return dtype == DAE_DTYPE_F32 ? launch_decode_rb<EPI_DENSE>(ctx, g, p)
                                  : launch_decode_rb_bf16<EPI_DENSE>(ctx, g, p);
}

// DAE term of the title mix: outT[c * ldT + r] = sigmoid(logit[r, c]) * row_scale[r] for the first n_items tiles
int dae_launch_decode_scaled_T(dae_ctx* ctx, const dae_rowgeom& g, int B, const dae_tileset& ts, const float* row_scale,
                               float* outT, int64_t ldT, int dtype)
{
    DecP p;
    int rc = fill_common(ctx, g, B, ts, p, dtype);
    if (rc) return rc;
    p.mixT = nullptr;
    p.outT = outT; p.ld_outT = ldT; p.row_scale = row_scale; p.mask_from_col = INT_MAX;
    return dtype == DAE_DTYPE_F32 ? launch_decode_rb<EPI_DENSE>(ctx, g, p) : launch_decode_rb_bf16<EPI_DENSE>(ctx, g, p);
}

// K5 from the row-major decoder (fp32, hidden = 256, 128-row groups); returns DAE_ERR_STATE when the shape does not apply
int dae_launch_decode_loss_rowmajor(dae_ctx* ctx, const dae_rowgeom& g, int B, int V, int H, const float* W,
                                    const float* bias, const float* h, float inv_n_batch, float* dzT, int64_t ldT,
                                    float* loss_part, int dtype, int dz16)
{
    if (H != 256 || g.R_TILE != 128 || g.waves != 4) return DAE_ERR_STATE;
    if ((uint64_t)ldT * 4 + (uint64_t)g.Bpad >= (1ull << 30)) return DAE_ERR_STATE;          // (32-bit lane offsets in the epilogues)
    LossRmP p;
    p.W = W; p.bias = bias; p.h = h; p.V = V; p.H = H; p.B = B; p.n_rg = g.n_rg; p.nb_rg = g.nb_rg;
    p.inv_nb = inv_n_batch; p.dzT = dzT; p.ldT = ldT; p.loss_part = loss_part;
    if (dtype == DAE_DTYPE_BF16) {
        // the W tile through LDS, a workgroup per tile x all playlists (training batches are <= 256: train.hip): g.grid
        // workgroups, one loss partial each, as the caller sized them
        if (B > 256) return DAE_ERR_STATE;
        if (dz16) hipLaunchKernelGGL(decode_loss_shared_bf16_kernel<true>, dim3(g.grid), dim3(512), 0, ctx->stream, p);
        else hipLaunchKernelGGL(decode_loss_shared_bf16_kernel<false>, dim3(g.grid), dim3(512), 0, ctx->stream, p);
        DAE_CHECK_LAUNCH(ctx, "decode_loss_shared_bf16_kernel");
        return DAE_OK;
    }
    if (B > 256) return DAE_ERR_STATE;
    const size_t lds_s = ((size_t)2 * 32 * 260 + 8) * sizeof(float);
    static const char rs_key = 0;
    if (dae_first_use(ctx, &rs_key))
        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_loss_shared_f32_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
    hipLaunchKernelGGL(decode_loss_shared_f32_kernel, dim3(g.grid), dim3(512), lds_s, ctx->stream, p);
    DAE_CHECK_LAUNCH(ctx, "decode_loss_shared_f32_kernel");
    return DAE_OK;
}

// K5 + K7 fused (decode_loss_dh_bf16_kernel): g.grid partials of dh at `part` ([g.grid][Bpad64][H]); DAE_ERR_STATE when the shape does not apply
int dae_launch_decode_loss_dh(dae_ctx* ctx, const dae_rowgeom& g, int B, int V, int H, const float* W, const float* bias,
                              const float* h, float inv_n_batch, float* dzT, int64_t ldT, float* loss_part, float* part, int Bpad64)
{
    if (H != 256 || B > 256 || g.R_TILE != 128 || g.waves != 4) return DAE_ERR_STATE;
    if ((uint64_t)ldT * 4 + (uint64_t)g.Bpad >= (1ull << 30)) return DAE_ERR_STATE;
    LossRmP p;
    p.W = W; p.bias = bias; p.h = h; p.V = V; p.H = H; p.B = B; p.n_rg = g.n_rg; p.nb_rg = g.nb_rg;
    p.inv_nb = inv_n_batch; p.dzT = dzT; p.ldT = ldT; p.loss_part = loss_part;
    const size_t lds = ((size_t)2 * 32 * 132 + (size_t)2 * 256 * 18) * 4 + (size_t)8 * 8 * 64 * 16 + 64;
    static const char key = 0;
    if (dae_first_use(ctx, &key))
        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_loss_dh_bf16_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(decode_loss_dh_bf16_kernel, dim3(g.grid), dim3(512), lds, ctx->stream, p, part, Bpad64);
    DAE_CHECK_LAUNCH(ctx, "decode_loss_dh_bf16_kernel");
    return DAE_OK;
}

int dae_launch_decode_loss_f32(dae_ctx* ctx, const dae_rowgeom& g, int B, float inv_n_batch,
                               float* dzT, int64_t ldT, float* loss_part, int dtype, int dz16)
{
    DecP p;
    const dae_packed& pk = dtype == DAE_DTYPE_F32 ? ctx->pk_f32 : ctx->pk_bf16;
    dae_tileset ts{pk.ntiles, 1, 0, static_cast<const int*>(pk.ident.p)};
    int rc = fill_common(ctx, g, B, ts, p, dtype);
    if (rc) return rc;
    p.dzT = dzT; p.ldT = ldT; p.loss_part = loss_part; p.inv_nb = inv_n_batch;
    p.dz16 = (dtype == DAE_DTYPE_BF16 && dz16) ? 1 : 0;
    if (dtype == DAE_DTYPE_BF16) {
        // bf16 operands, fp32 accumulate (BASELINE.json configs[3]): the matrix time drops to ~1/16, the launch is
        // bound by its VALU epilogue and the dz^T store; two waves per SIMD overlap those with the MFMAs
        if (g.R_TILE == 128 && p.G == 16) return launch_decode<4, EPI_LOSS, 16, 8, DT_BF16>(ctx, g, p);
        return launch_decode_rb_bf16<EPI_LOSS>(ctx, g, p);
    }
    // A/B: DAE_LOSS_WAVES=8 runs two waves per SIMD on the 128-row image so that one wave's VALU epilogue
    // (4 transcendentals per element) sits under the other's MFMAs; measured 240 us against 229 us for the
    // default one wave per SIMD (V = 170 000, B = 256)
    static const bool w8 = dae_exp_env("DAE_LOSS_WAVES") && atoi(dae_exp_env("DAE_LOSS_WAVES")) == 8;
    if (g.R_TILE == 128 && p.G == 32 && w8) return launch_decode<4, EPI_LOSS, 32, 8, DT_F32>(ctx, g, p);
    return launch_decode_rb<EPI_LOSS>(ctx, g, p);
}

int dae_launch_decode_filter_f32(dae_ctx* ctx, const dae_rowgeom& g, int B, const dae_tileset& ts,
                                 const float* tau, int n_valid_col, uint2* cand, int* cand_cnt,
                                 int cap, int dtype, int bias_sel)
{
    DecP p;
    int rc = fill_common(ctx, g, B, ts, p, dtype, bias_sel);
    if (rc) return rc;
    p.tau = tau; p.n_valid_col = n_valid_col; p.cand = cand; p.cand_cnt = cand_cnt; p.cap = cap;
    static const bool f32_generic = dae_exp_env("DAE_F32_GENERIC") != nullptr;          // A/B against the generic body
    if (dtype == DAE_DTYPE_F32 && g.R_TILE == 128 && p.G == 32 && g.waves == 4 && !f32_generic && !p.mixT) {
        const size_t lds = (size_t)4 * 64 * 32 * sizeof(float4) + 128 * sizeof(int) + 128 * sizeof(float);
        static const char attr_set_key = 0;
        if (dae_first_use(ctx, &attr_set_key)) {
            DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_f32_h256_filter_kernel<0>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (ctx->prof_armed) {
            e0 = ctx->prof_ev[ctx->prof_used]; e1 = ctx->prof_ev[ctx->prof_used + 1];
            ctx->prof_armed = false;
            ctx->prof_used += 2;
            ctx->prof_kernel = "decode_f32_h256_filter_kernel<0>";
        }
        hipExtLaunchKernelGGL(decode_f32_h256_filter_kernel<0>, dim3(g.grid), dim3(256), lds, ctx->stream, e0, e1, 0, p);
        DAE_CHECK_LAUNCH(ctx, "decode_f32_h256_filter_kernel");
        return DAE_OK;
    }
    if (bf16_fast_filter(g, dtype, p.G) && !p.mixT) {
#ifdef DAE_EXPERIMENTS
        static const bool dbgF = dae_exp_env("DAE_DBG_F") != nullptr;     // stage stamps of the dedicated bf16 filter kernel
        static long long* fbuf = nullptr;
        static int fcalls = 0;
        if (dbgF) {
            if (!fbuf) { (void)hipMalloc(&fbuf, 32 * 8); (void)hipMemset(fbuf, 0, 32 * 8); }
            p.stamps = fbuf;
            if ((++fcalls % 100) == 0) {
                long long h[32];
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipMemcpy(h, fbuf, sizeof(h), hipMemcpyDeviceToHost);
                for (int w = 0; w < 2; ++w) {
                    fprintf(stderr, "FILTER wg%d:", w ? 100 : 0);
                    for (int i = 1; i < 15; ++i) if (h[16 * w + i]) fprintf(stderr, " [%d]%lld", i, h[16 * w + i] - h[16 * w]);
                    fprintf(stderr, "\n");
                }
            }
        }
#endif
        const size_t lds = (size_t)(g.R_TILE / 32) * 64 * 16 * sizeof(float4) + (size_t)g.R_TILE * (sizeof(int) + sizeof(float)) + 16;     // (+ the claim counter)
        static const char attr_set_key = 0;
        if (dae_first_use(ctx, &attr_set_key)) {
            DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_bf16_h256_filter_kernel<1, 4, 8, 8>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_bf16_h256_filter_kernel<2, 4, 16, 4>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_bf16_h256_filter_kernel<1, 8, 16, 4>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const int regb = bf16_regb_variant(g);
#ifdef DAE_EXPERIMENTS
        if (regb) {
            if (ctx->prof_armed) {
                e0 = ctx->prof_ev[ctx->prof_used]; e1 = ctx->prof_ev[ctx->prof_used + 1];
                ctx->prof_armed = false;
                ctx->prof_used += 2;
                ctx->prof_kernel = regb == 2 ? "decode_bf16_h256_regb_filter_kernel<2, 16>" : "decode_bf16_h256_regb_filter_kernel<3, 8>";
            }
            if (regb == 2)
                hipExtLaunchKernelGGL((decode_bf16_h256_regb_filter_kernel<2, 16>), dim3(g.grid), dim3(256), 0, ctx->stream, e0, e1, 0, p);
            else
                hipExtLaunchKernelGGL((decode_bf16_h256_regb_filter_kernel<3, 8>), dim3(g.grid), dim3(256), 0, ctx->stream, e0, e1, 0, p);
            DAE_CHECK_LAUNCH(ctx, "decode_bf16_h256_regb_filter_kernel");
            return DAE_OK;
        }
#else
        (void)regb;
#endif
        if (ctx->prof_armed) {
            e0 = ctx->prof_ev[ctx->prof_used]; e1 = ctx->prof_ev[ctx->prof_used + 1];
            ctx->prof_armed = false;
            ctx->prof_used += 2;
            ctx->prof_kernel = g.R_TILE == 256 ? "decode_bf16_h256_filter_kernel<1, 8, 16, 4>"
                             : bf16_pair_variant() ? "decode_bf16_h256_filter_kernel<2, 4, 16, 4>"
                                                   : "decode_bf16_h256_filter_kernel<1, 4, 8, 8>";
        }
        if (g.R_TILE == 256)
            hipExtLaunchKernelGGL((decode_bf16_h256_filter_kernel<1, 8, 16, 4>), dim3(g.grid), dim3(256), lds,
                                  ctx->stream, e0, e1, 0, p);
        else if (bf16_pair_variant())
            hipExtLaunchKernelGGL((decode_bf16_h256_filter_kernel<2, 4, 16, 4>), dim3(g.grid), dim3(256), lds,
                                  ctx->stream, e0, e1, 0, p);
        else
            hipExtLaunchKernelGGL((decode_bf16_h256_filter_kernel<1, 4, 8, 8>), dim3(g.grid), dim3(512), lds,
                                  ctx->stream, e0, e1, 0, p);
        DAE_CHECK_LAUNCH(ctx, "decode_bf16_h256_filter_kernel");
        return DAE_OK;
    }
    return dtype == DAE_DTYPE_F32 ? launch_decode_rb<EPI_FILTER>(ctx, g, p)
                                  : launch_decode_rb_bf16<EPI_FILTER>(ctx, g, p);
}
