// mixexact.hip -- DAE_DTYPE_BF16_EXACT under the title mix (`--challenge` with titles: main_challenge.py:80-90 of the
// reference always scores through DAE_title, DAEs.py:153-181):
//
//     y[r, c] = sigmoid(z_t[r, c]) * w_title[r] + sigmoid(z_d[r, c]) * w_playlist[r]                      (DAEs.py:180)
//     z_d = h[r, :] . W_dec[c, :] + b_dec[c]            (hidden 256: the frozen DAE, DAEs.py:165-171)
//     z_t = feat[r, :] . Output_W[:, c] + Output_b[c]   (400 title features in rows of 448, Char_CNN.py:64-72)
//
// The fp32 path runs both vocabulary-wide GEMMs on v_mfma_f32_32x32x2_f32 (the canonical fmaf chains) and ranks y.  Here
// BOTH run on v_mfma_f32_32x32x16_bf16 in ONE launch per pass -- a wave decodes a 32-column tile against a row group's
// hidden rows of both scorers (704 k values, two accumulator sets) -- on BOUNDS of the two logits, and only the survivors are
// recomputed in fp32:
//   sample launch (MODE 1): biases shifted DOWN by the bounds -> lower bounds l of y over the head of the bias-ordered tile
//          list; their maxima over groups of 4 columns go to the threshold kernel (tau_select_kernel, topk.hip), whose tau is
//          a valid lower bound of the row's k-th largest rankable non-seed y;
//   filter launch (MODE 0): biases shifted UP -> upper logits (u_t, u_d); a column is listed with them when its upper bound
//          can reach tau.  The test needs no sigmoid and no division: with E = exp(-z),
//              w_t / (1 + E_t) + w_p / (1 + E_d) >= tau   <=>   w_t (1 + E_d) + w_p (1 + E_t) >= tau (1 + E_t)(1 + E_d),
//          two v_exp_f32 and a handful of multiply-adds per element, compared with 2^-15 of slack (mix_can_reach).  The
//          lane's pair of MAXIMA over its 16 columns takes the test first: most (row block, tile) pairs of the low-bias
//          tiles stop there.  (The canonical sigmoid in the epilogue -- ~35 VALU instructions, twice per element -- made the
//          launch VALU-bound at 316 us for 750 rows; two logit thresholds per row, u_t >= thT or u_d >= thD, cost 3
//          instructions but list EVERY column when one scorer is a flat background, e.g. an untrained title model.)
//   refine launch, one 1024-thread workgroup per row: (1) the mixed bounds [l, u] of every listed column, densely (a lane
//          per candidate, hardware exp2 / rcp, 2^-15 wider); the need-th largest l is a threshold tau' that `need` columns
//          provably reach, so only columns with u >= tau' can be among the k best -- both keys staged in LDS for rows of up
//          to 8 192 candidates, three streamed passes with a two-level fixed-bin histogram beyond; (2) the survivors are
//          recomputed: z_t, z_d with the canonical chains, mixed with the operations of mix_scores_kernel (title.hip) in
//          their order -- the value the fp32 path ranks -- tested against the promise (bound guard, as refine.hip) and left
//          as one compact (y, column) list per row for the selection kernel.  Rows with both weights 0 (no input, no title)
//          have y = +0 everywhere: they list nothing and get the first k + n_seeds columns directly.
// BOUNDS.  DAE side: hidden rows lie in [0, 1], eps_c of exact_bounds_kernel (decode_f32.hip) as in the plain exact mode.
// Title side: the features are ReLU maxima, not confined to [0, 1]; with F_r = max_k |feat[r][k]| every term of that
// derivation that is linear in the hidden row scales by F_r:
//     |z16_t - z32_t| <= F_r alpha_c + beta_c,
//     alpha_c = (1 + 2^-8) d_c + 2^-9 n_c + A16 (1 + 2^-8)(n_c + d_c) + A32 n_c,   beta_c = (1.01 A16 + A32 + 2^-23) |b_c|
// (d_c, n_c, A16, A32 as there; both inflated for the feedback of the shift through the accumulation).  The shift itself
// goes through the matrix pipe like the bias: k-slot 3 of the tile's bias fragment holds +-alpha_c (bf16, rounded up), k-slot
// 3 of the row's "ones" fragment holds F_r (bf16, rounded up), slots 0..2 the three-term split of b_c +- beta_c.
// The canonical sigmoid is monotone only to one unit in the last place (920 one-ulp inversions among the 2.2 10^9 floats of
// [-88, 88], checked exhaustively), so bounds of the mixed value are widened (2^-15 with the hardware's exp2 / rcp, whose
// results lie within 6e-6 of the canonical ones) before they are compared; the guard compares LOGITS, which need none.
#include <climits>

#include "dae_internal.h"
#include "rank_lds.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 as_bf16x8(const uint4 u) { return __builtin_bit_cast(bf16x8, u); }

constexpr int MX_RB = 3;            // row blocks of 32 playlists per row group: 96 rows x 704 k x 2 B = 132 KiB of LDS
constexpr int MX_NW = 8;            // waves per workgroup (two per SIMD: one's epilogue under the other's MFMAs)
constexpr int MX_QR = 8;            // W ring depth (steps)
constexpr int MX_REF_CAP = 8192;    // survivors per row the refine launch's compact lists hold

// the mixed score with the operations of mix_scores_kernel (title.hip) / the fp32 mix epilogue (decode_f32.hip), in their order
__device__ __forceinline__ float mixf(float zt, float zd, float wt, float wp)
{
    const float ts = dae_sigmoidf(zt) * wt;
    const float pd = dae_sigmoidf(zd) * wp;
    return ts + pd;
}
// The threshold sample evaluates every element, so it takes the hardware's exp2 / rcp (1 ulp each) instead of the canonical
// polynomial (4 instructions against ~35): within 6.1e-6 relative of the canonical value for every finite logit (the
// argument's rounding, |z| log2(e) 2^-24 <= 7.6e-6 in the exponent, dominates), hence a lower bound after 2^-15 relative.
__device__ __forceinline__ float sig_fast(float z)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896341f));
}
// Can w_t sigmoid(z_t) + w_p sigmoid(z_d) reach tau?  True whenever the canonical mixed value, widened by 2^-20, does (the
// caller passes tau (1 - 2^-17): rounding, the canonical sigmoid's <= 3 ulp and its 1-ulp inversions); false only with 2^-15 of room.  With E = exp(-z) (v_exp_f32: 1 ulp, its argument's
// rounding <= 6e-6 relative in E; arguments capped at 2^60, which only raises the left side's share):
//     w_t / (1 + E_t) + w_p / (1 + E_d) >= tau  <=>  w_t (1 + E_d) + w_p (1 + E_t) >= tau (1 + E_t)(1 + E_d)
__device__ __forceinline__ bool mix_can_reach(float zt, float zd, float wt, float wp, float tau)
{
    const float et = 1.0f + __builtin_amdgcn_exp2f(fminf(zt * -1.44269504088896341f, 60.0f));
    const float ed = 1.0f + __builtin_amdgcn_exp2f(fminf(zd * -1.44269504088896341f, 60.0f));
    const float lhs = fmaf(wp, et, wt * ed);
    const float rhs = (tau * et) * ed;
    return fmaf(lhs, 0x1p-15f, lhs) >= rhs;
}
// Per-lane LOGIT thresholds that every column passing mix_can_reach must meet, given the lane's maxima mT >= z_t, mD >= z_d over its
// columns: w_t s(z_t) + w_p s(z_d) >= tau with s(z_d) <= s(mD) needs s(z_t) >= (tau - w_p s(mD)) / w_t =: a, i.e. z_t >= logit(a) -- and the
// same with the roles swapped.  One compare per element instead of two v_exp_f32 and five multiply-adds: on a title scorer whose
// score is a flat background (an untrained one: bench.py's) the pair of maxima passes in nearly every lane, and the per-element tests
// were a third of the launch.  NECESSARY conditions only (the exact test still decides): every bound is taken on the safe side --
// tau 2^-12 lower (mix_can_reach's own slack is 2^-15 plus a few 1e-7 of hardware exp / rcp error), the maxima's sigmoids 2^-18 higher,
// the quotient 2^-18 lower, the logit 2^-16 (relative and absolute) lower; a >= 0.999 (a column that needs a saturated sigmoid) keeps
// logit(0.999), a <= 0 (the other scorer's maximum alone reaches tau) keeps -inf.  A scorer with weight 0 puts no condition on its logit.
__device__ __forceinline__ float mix_logit_floor(float a)
{
    if (!(a > 1e-30f)) return -__builtin_inff();
    a = fminf(a, 0.999f);
    const float l = (__builtin_amdgcn_logf(a) - __builtin_amdgcn_logf(1.0f - a)) * 0.69314718f;        // ln(a / (1 - a))
    return l - (fabsf(l) * 0x1p-16f + 0x1p-16f);
}
__device__ __forceinline__ void mix_thresholds(float mT, float mD, float wt, float wp, float tau, float& th_t, float& th_d)
{
    const float sT = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fminf(mT * -1.44269504088896341f, 60.0f))) * (1.0f + 0x1p-18f);
    const float sD = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fminf(mD * -1.44269504088896341f, 60.0f))) * (1.0f + 0x1p-18f);
    const float tl = tau * (1.0f - 0x1p-12f);
    const float nt = tl - wp * sD, nd = tl - wt * sT;            // what the title / the DAE term has to supply at least
    th_t = wt > 0.0f ? mix_logit_floor(nt > 0.0f ? nt * __builtin_amdgcn_rcpf(wt) * (1.0f - 0x1p-18f) : -1.0f) : -__builtin_inff();
    th_d = wp > 0.0f ? mix_logit_floor(nd > 0.0f ? nd * __builtin_amdgcn_rcpf(wp) * (1.0f - 0x1p-18f) : -1.0f) : -__builtin_inff();
}
__device__ __forceinline__ float mix_fast_dn(float zt, float zd, float wt, float wp)
{
    const float y = sig_fast(zt) * wt + sig_fast(zd) * wp;
    return fmaf(y, -0x1p-15f, y);
}
__device__ __forceinline__ float mix_fast_up(float zt, float zd, float wt, float wp)
{
    const float y = sig_fast(zt) * wt + sig_fast(zd) * wp;
    return fmaf(y, 0x1p-15f, y);
}

struct MixP {
    const uint4* WqD; const uint4* WqT;          // bf16 images [tile][NS][64] of the two scorers (same tiles: column 32 t + i)
    const uint4* biasD; const uint4* biasT;      // [tile][64] bias fragments: the lower or the upper shift
    const uint4* hp;                             // hidden rows of both scorers [n_rg][NSD + NST][RB][64] (mix_pack_kernel)
    const unsigned* fhat;                        // [Bpad] bf16 bits of F_r, rounded up
    const float* w_t; const float* w_p;          // [B]
    const float* tau;                            // [B] (filter)
    const int* list; int n_items;                // tiles of this launch
    int n_valid_col;                             // rankable columns (the images start at column 0)
    int B, n_rg, nb_rg, Bpad;
    uint4* cand; int* cand_cnt; int cap;         // filter: [bir][Bpad][cap] entries (u_t, u_d, column, 0) + [bir][Bpad] counts
    float* samp; int64_t ld_s;                   // sample: [B][ld_s] maxima of 4-column groups, element item * 8 + 4 hi + qd
    int exp_mode;                                // experiments build: 1 = no epilogue at all, 2 = the maxima's test only
};

template <int NSD, int NST, int RB, int QR, int NW, int MODE>
__global__ __launch_bounds__(NW * 64, 1) void mix_bf16_kernel(const MixP p)
{
    constexpr int NS = NSD + NST, R_TILE = RB * 32, NTH = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int j = lane & 31;

    // the workgroups that walk the same tiles for different row groups sit on one XCD (decode_f32.hip)
    const int gs = DAE_NUM_XCD * p.n_rg;
    const int q = blockIdx.x / gs, rem = blockIdx.x % gs;
    const int rg = rem / DAE_NUM_XCD;
    const int bir = q * DAE_NUM_XCD + (rem % DAE_NUM_XCD);

    constexpr int n_h4 = RB * 64 * NS;
    constexpr int PER = (n_h4 + NTH - 1) / NTH;
    constexpr int CH = PER < 8 ? PER : 8;
    const int n_items = p.n_items;
    const int n_ws = p.nb_rg * NW;
    const int it0 = wave * p.nb_rg + bir;
    const uint4* ldsq = reinterpret_cast<const uint4*>(lds4);

    // a wave's tiles: its first two by position, every further one claimed from a counter in LDS (decode_f32.hip)
    const bool has = it0 < n_items;
    const int tv0 = p.list[has ? it0 : 0];
    const int tv1 = p.list[has ? (it0 + n_ws < n_items ? it0 + n_ws : it0) : 0];
    // filter: the row's threshold (rows beyond B list nothing)
    static_assert(R_TILE <= NTH, "one thread per row");
    // (every load of this prologue is UNCONDITIONAL, on a clamped row: a guarded load is a branch, and hipcc ends each guarded
    // group on s_waitcnt vmcnt(0) -- eight dependent trips to memory before the first MFMA, round 6's ISA reading)
    float tau_g = __builtin_inff();
    if (MODE == 0) {
        const int rr = rg * R_TILE + (tid < R_TILE ? tid : 0);
        const int rc = rr < p.B ? rr : p.B - 1;
        const float tau = p.tau[rc];
        const float a_t = p.w_t[rc], a_p = p.w_p[rc];
        if (tid < R_TILE && rr < p.B) {
            // the compared bound is widened by 2^-20 and rounded twice; tau <= 0 (or -inf: fewer than `need` sample values,
            // or NaN): every column passes
            tau_g = tau > 0.0f ? tau * (1.0f - 0x1p-17f) : 0.0f;
            // a row without input and without title (the padding rows of a reader's last batch, main_challenge.py:75-78): both
            // weights are 0, y is +0 for every column -- nothing is listed, the refine launch writes its list directly
            if (a_t == 0.0f && a_p == 0.0f) tau_g = __builtin_inff();
        }
    }
    int* lcnt = reinterpret_cast<int*>(lds4 + n_h4);
    float* ltau = reinterpret_cast<float*>(lcnt + R_TILE);
    int* claim = reinterpret_cast<int*>(ltau + R_TILE);
    if (tid == 0) *claim = 2 * NW;
    {
        const uint4* hsrc = p.hp + (size_t)rg * n_h4;
        uint4* l4 = reinterpret_cast<uint4*>(lds4);
#pragma unroll
        for (int e0 = 0; e0 < PER; e0 += CH) {
            uint4 hv[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int i = (e0 + e) * NTH + tid;
                hv[e] = hsrc[i < n_h4 ? i : n_h4 - 1];
            }
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int i = (e0 + e) * NTH + tid;
                if (e0 + e < PER && i < n_h4) l4[i] = hv[e];
            }
        }
    }
    if (tid < R_TILE) { lcnt[tid] = 0; ltau[tid] = tau_g; }
    // per lane: the mixing weights and the row's feature bound of its RB rows
    float wt_r[RB], wp_r[RB];
    unsigned oy[RB];                                              // .y of the title side's "ones" fragment (k-slots 2, 3)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int row = rg * R_TILE + rb * 32 + j;
        const bool in = row < p.B;
        const int rc = in ? row : p.B - 1;
        const float a_t = p.w_t[rc], a_p = p.w_p[rc];
        const unsigned fh = p.fhat[rc];
        wt_r[rb] = in ? a_t : 0.0f;
        wp_r[rb] = in ? a_p : 0.0f;
        oy[rb] = hi == 0 ? (0x3F80u | ((in ? fh : 0u) << 16)) : 0u;
    }
    __builtin_amdgcn_sched_barrier(0);
    int t = __builtin_amdgcn_readfirstlane(tv0), u = __builtin_amdgcn_readfirstlane(tv1);
    // step sg of tile x: the DAE image's steps first, then the title image's
    auto wsrc = [&](int x, int sg) -> const uint4* {
        return sg < NSD ? p.WqD + ((size_t)x * NSD + sg) * 64 + lane : p.WqT + ((size_t)x * NST + (sg - NSD)) * 64 + lane;
    };
    uint4 wq[QR];
    uint4 cb[2][RB];
    uint4 bfD = p.biasD[(size_t)t * 64 + lane], bfT = p.biasT[(size_t)t * 64 + lane];
#pragma unroll
    for (int k = 0; k < QR; ++k) wq[k] = *wsrc(t, k);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();

    float tau_r[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) tau_r[rb] = ltau[rb * 32 + j];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) cb[0][rb] = ldsq[rb * 64 + lane];
    const uint4 onesD = hi == 0 ? make_uint4(0x3F803F80u, 0x00003F80u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);

    int g_nxt = it0 + n_ws;
    for (int it = it0; it < n_items;) {
        int g_nn;
        {
            int n2 = 0;
            if (lane == 0) n2 = atomicAdd(claim, 1);
            g_nn = __builtin_amdgcn_readfirstlane(n2) * p.nb_rg + bir;
        }
        const int wv = p.list[g_nn < n_items ? g_nn : it];       // consumed at the end of this tile

        f32x16 accD[RB], accT[RB];
        {
            f32x16 zero;
#pragma unroll
            for (int e = 0; e < 16; ++e) zero[e] = 0.0f;
            const uint4 bd = bfD, bt = bfT;
            bfD = p.biasD[(size_t)u * 64 + lane];
            bfT = p.biasT[(size_t)u * 64 + lane];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                accD[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(bd), as_bf16x8(onesD), zero, 0, 0, 0);
                const uint4 ot = hi == 0 ? make_uint4(0x3F803F80u, oy[rb], 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
                accT[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(bt), as_bf16x8(ot), zero, 0, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int sn = (s + 1) % NS;
            const uint4 a = wq[s % QR];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if (s < NSD)
                    accD[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(cb[s & 1][rb]), accD[rb], 0, 0, 0);
                else
                    accT[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(cb[s & 1][rb]), accT[rb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                cb[(s + 1) & 1][rb] = ldsq[(sn * RB + rb) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
            }
            wq[s % QR] = (s + QR < NS) ? *wsrc(t, s + QR) : *wsrc(u, s + QR - NS);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue: lane = playlist j of row block rb; register reg is column 32 t + 4 hi + (reg & 3) + 8 (reg >> 2)
#ifdef DAE_EXPERIMENTS
        if (p.exp_mode == 1) {
            float sink = 0.0f;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) sink += accT[rb][0] + accD[rb][5];
            if (sink == 123.456f) p.cand_cnt[0] = 1;
        } else
#endif
        if (MODE == 0) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float tv = tau_r[rb], wt = wt_r[rb], wp = wp_r[rb];
                float mT = accT[rb][0], mD = accD[rb][0];
#pragma unroll
                for (int reg = 1; reg < 16; ++reg) { mT = fmaxf(mT, accT[rb][reg]); mD = fmaxf(mD, accD[rb][reg]); }
                // (the pair of maxima bounds every pair of the lane's 16 columns)
#ifdef DAE_EXPERIMENTS
                if (p.exp_mode == 2) { if (mix_can_reach(mT, mD, wt, wp, tv) && tv == 123.456f) p.cand_cnt[0] = 1; continue; }
#endif
                if (mix_can_reach(mT, mD, wt, wp, tv)) {
                    // one compare pair per element first (mix_thresholds); the exact test only for the registers in which SOME lane
                    // of the wave still has a column in play (wave-uniform: __ballot)
                    float th_t, th_d;
                    mix_thresholds(mT, mD, wt, wp, tv, th_t, th_d);
                    unsigned pm = 0;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        if (accT[rb][reg] >= th_t && accD[rb][reg] >= th_d) pm |= 1u << reg;
                    unsigned m = 0;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        if (__ballot((pm >> reg) & 1u)) {
                            const int lc = t * 32 + 4 * hi + (reg & 3) + 8 * (reg >> 2);
                            if (((pm >> reg) & 1u) && mix_can_reach(accT[rb][reg], accD[rb][reg], wt, wp, tv) && lc < p.n_valid_col)
                                m |= 1u << reg;
                        }
                    }
                    if (m) {
                        int at = atomicAdd(&lcnt[rb * 32 + j], __popc(m));
                        uint4* dst = p.cand + ((size_t)bir * p.Bpad + rg * R_TILE + rb * 32 + j) * (size_t)p.cap;
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            if (m & (1u << reg))
                                dst[at++] = make_uint4(__float_as_uint(accT[rb][reg]), __float_as_uint(accD[rb][reg]),
                                                       (unsigned)(t * 32 + 4 * hi + (reg & 3) + 8 * (reg >> 2)), 0u);
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int row = rg * R_TILE + rb * 32 + j;
                float o[4];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    float mx = -__builtin_inff();
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int lc = t * 32 + 4 * hi + e + 8 * qd;
                        const float y = mix_fast_dn(accT[rb][4 * qd + e], accD[rb][4 * qd + e], wt_r[rb], wp_r[rb]);
                        if (lc < p.n_valid_col) mx = fmaxf(mx, y);
                    }
                    o[qd] = mx;
                }
                if (row < p.B)
                    *reinterpret_cast<float4*>(p.samp + (size_t)row * p.ld_s + (size_t)it * 8 + 4 * hi) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        t = u; u = __builtin_amdgcn_readfirstlane(wv);
        it = g_nxt; g_nxt = g_nn;
    }
    if (MODE == 0) {
        __syncthreads();
        for (int i = tid; i < R_TILE; i += NTH) p.cand_cnt[(size_t)bir * p.Bpad + rg * R_TILE + i] = lcnt[i];
    }
}

// ---- the title image's row-scaled bounds (see the header) ---------------------------------------------------------------
// One 256-thread workgroup per 32-column tile, 8 threads per column (as exact_bounds_kernel).  alpha_hat[c]: the bf16
// value (as a float) the fragments carry; beta[c]: the shift folded into the bias split.
__device__ __forceinline__ uint4 mix_bias_fragment(float bv, unsigned slot3)
{
    const unsigned e0 = dae_bf16_rne(bv);
    const float r1 = bv - __uint_as_float(e0 << 16);
    const unsigned e1 = dae_bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(e1 << 16);
    const unsigned e2 = dae_bf16_rne(r2);
    return make_uint4(e0 | (e1 << 16), e2 | (slot3 << 16), 0u, 0u);
}

__global__ __launch_bounds__(256) void mix_title_bounds_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                               int H, int Hp, int col_lo, int col_hi, int ntiles,
                                                               float* __restrict__ alpha_hat, float* __restrict__ beta,
                                                               uint4* __restrict__ frag_lo, uint4* __restrict__ frag_hi,
                                                               float margin, int m_lo, int m_hi, float m_scale)
{
    const int tid = threadIdx.x;
    const int c = tid >> 3, part = tid & 7;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int v = col_lo + t * 32 + c;
        double n = 0.0, d = 0.0;
        if (v < col_hi) {
            const float* row = W + (size_t)v * H;
            for (int k = part; k < H; k += 8) {
                const float w = row[k];
                const float w16 = __uint_as_float(dae_bf16_rne(w) << 16);
                n += fabs((double)w);
                d += fabs((double)w16 - (double)w);
            }
        }
#pragma unroll
        for (int sh = 1; sh < 8; sh <<= 1) { n += __shfl_xor(n, sh); d += __shfl_xor(d, sh); }
        if (part == 0) {
            float a_f = 0.0f, be_f = 0.0f, lo_f = 0.0f, hi_f = 0.0f;
            unsigned a16 = 0u;
            if (v < col_hi) {
                const double bv = (double)b[v], ab = fabs(bv);
                const double A16 = (double)(Hp + 16) * 0x1p-22;
                const double A32 = (double)(H + 2) * 0x1p-24 * (1.0 + 0x1p-10);
                const double up = 1.0 + 0x1p-8;               // bf16(f) <= F_r (1 + 2^-8); bf16(F_r) and bf16(alpha) round up
                double al = up * d + 0x1p-9 * n + A16 * up * (n + d) + A32 * n;
                double be = (1.01 * A16 + A32 + 0x1p-23) * ab;
                al = al * (1.0 + 4.0 * A16 * up * up) * (1.0 + 1e-6) + 1e-30;      // the shift feeds back through the accumulation
                be = be * (1.0 + 4.0 * A16) * (1.0 + 1e-6) + 0x1p-23 * be + 1e-30;
                // dae_set_exact_margin (< 1 voids the bound: the guard's test hook); _range: some columns only
                const bool in_range = v >= m_lo && v < m_hi;
                const double mg = (double)(in_range && m_scale > 0.0f ? m_scale : margin);
                al *= mg; be *= mg;
                a_f = (float)al;
                if ((double)a_f < al) a_f = __uint_as_float(__float_as_uint(a_f) + 1u);
                a16 = (__float_as_uint(a_f) + 0xFFFFu) >> 16;                     // bf16, rounded up (a_f > 0)
                a_f = __uint_as_float(a16 << 16);
                be_f = (float)be;
                if ((double)be_f < be) be_f = __uint_as_float(__float_as_uint(be_f) + 1u);
                double lo = bv - (double)be_f, hi = bv + (double)be_f;
                if (in_range && m_scale < 0.0f) hi = bv + (double)m_scale;     // (a FORGED filter: the upper bound |scale| logits low)
                lo_f = (float)lo; if ((double)lo_f > lo) lo_f = nextafterf(lo_f, -__builtin_inff());
                hi_f = (float)hi; if ((double)hi_f < hi) hi_f = nextafterf(hi_f, __builtin_inff());
            }
            alpha_hat[t * 32 + c] = a_f;
            beta[t * 32 + c] = be_f;
            frag_lo[t * 64 + c] = v < col_hi ? mix_bias_fragment(lo_f, a16 | 0x8000u) : make_uint4(0u, 0u, 0u, 0u);
            frag_hi[t * 64 + c] = v < col_hi ? mix_bias_fragment(hi_f, a16) : make_uint4(0u, 0u, 0u, 0u);
            frag_lo[t * 64 + 32 + c] = make_uint4(0u, 0u, 0u, 0u);
            frag_hi[t * 64 + 32 + c] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

// ---- per row: F_r (bf16, rounded up) and whether the bound's preconditions hold -----------------------------------------
// one wave per row.  row_bad[r] = 1: a DAE hidden entry outside [0, 1], a feature that is not finite, or a mixing weight
// outside [0, 1] -- such rows return no recommendations (idx -1), as in the plain exact mode
__device__ __forceinline__ void mix_rowprep_body(int block, const float* __restrict__ h, int64_t ld_h, int HD,
                                                 const float* __restrict__ feat, int64_t ld_f, int HT,
                                                 const float* __restrict__ w_t, const float* __restrict__ w_p,
                                                 int B, int Bpad, unsigned* __restrict__ fhat, int* __restrict__ row_bad)
{
    const int lane = threadIdx.x & 63;
    const int row = block * 4 + (threadIdx.x >> 6);
    if (row >= Bpad) return;
    float mx = 0.0f;
    bool bad = false;
    if (row < B) {
        for (int k = lane; k < HT; k += 64) {
            const float f = fabsf(feat[(size_t)row * ld_f + k]);
            bad = bad || !(f <= 3.0e38f);
            mx = fmaxf(mx, f);
        }
        for (int k = lane; k < HD; k += 64) {
            const float x = h[(size_t)row * ld_h + k];
            bad = bad || !(x >= 0.0f && x <= 1.0f);
        }
        const float a = w_t[row], c = w_p[row];
        bad = bad || !(a >= 0.0f && a <= 1.0f) || !(c >= 0.0f && c <= 1.0f);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    const bool any_bad = __any(bad);
    if (lane == 0) {
        fhat[row] = mx > 0.0f ? (__float_as_uint(mx) + 0xFFFFu) >> 16 : 0u;
        row_bad[row] = any_bad ? 1 : 0;
    }
}

// hidden rows of both scorers in B-operand order: out uint4 index = ((rg * NS + s) * RB + rb) * 64 + lane holds the bf16 of
// h[r][16 s + 8 hi + 0..7] (s < NSD) or feat[r][16 (s - NSD) + 8 hi + 0..7], r = (rg RB + rb) 32 + j  (zero outside)
__device__ __forceinline__ void mix_pack_body(int block, int n_blocks, const float* __restrict__ h, int64_t ld_h, int HD,
                                              const float* __restrict__ feat, int64_t ld_f, int HT,
                                              int B, int NSD, int NS, int RB, int n_rg, uint4* __restrict__ hp)
{
    const size_t total = (size_t)n_rg * NS * RB * 64;
    for (size_t o = (size_t)block * 256 + threadIdx.x; o < total; o += (size_t)n_blocks * 256) {
        const int lane = (int)(o & 63);
        size_t x = o >> 6;
        const int rb = (int)(x % RB); x /= RB;
        const int s = (int)(x % NS);
        const int rg = (int)(x / NS);
        const int hi = lane >> 5, jj = lane & 31;
        const int r = (rg * RB + rb) * 32 + jj;
        const bool dae = s < NSD;
        const float* src = dae ? h + (size_t)r * ld_h : feat + (size_t)r * ld_f;
        const int k0 = 16 * (dae ? s : s - NSD) + 8 * hi;
        const int Hs = dae ? HD : HT;
        unsigned e[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) e[c] = dae_bf16_rne((r < B && k0 + c < Hs) ? src[k0 + c] : 0.0f);
        hp[o] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    }
}

// both in ONE launch (the first `prep_blocks` workgroups prepare rows, the others pack: two launches of ~5 us each sat on the
// launch's critical path -- profiles/r04_title_timeline.txt)
__global__ __launch_bounds__(256) void mix_prep_pack_kernel(const float* __restrict__ h, int64_t ld_h, int HD,
                                                            const float* __restrict__ feat, int64_t ld_f, int HT,
                                                            const float* __restrict__ w_t, const float* __restrict__ w_p, int B,
                                                            int Bpad, unsigned* __restrict__ fhat, int* __restrict__ row_bad,
                                                            int prep_blocks, int NSD, int NS, int RB, int n_rg,
                                                            uint4* __restrict__ hp)
{
    if ((int)blockIdx.x < prep_blocks)
        mix_rowprep_body((int)blockIdx.x, h, ld_h, HD, feat, ld_f, HT, w_t, w_p, B, Bpad, fhat, row_bad);
    else
        mix_pack_body((int)blockIdx.x - prep_blocks, (int)gridDim.x - prep_blocks, h, ld_h, HD, feat, ld_f, HT, B, NSD, NS, RB, n_rg, hp);
}

// ---- refine: bounds of every candidate, the ones that can still be among the k best recomputed in fp32 -------------------
constexpr int MR_THREADS = 1024, MR_WAVES = 16;
// ring of 2 blocks per chain: hipcc waits for every load in flight at each block anyway (run-time bounds around the refills), so a deeper
// ring bought nothing but registers -- at depth 4 the kernel spilled 12 of them into its chains (137 -> 131 us per 750 rows)
constexpr int MR_DEPTH = 2;
constexpr int MR_ROWSTRIDE = 20;       // dwords per candidate in a wave's transposition buffer (refine.hip)
constexpr int MR_MAX_SEG = 1024;
constexpr int MR_STAGE = 8192;         // candidates of a row whose two keys fit LDS (next to the 16 waves' 80 KiB of buffers)
constexpr int MR_BINS = 2048;

struct MixRefP {
    const uint4* base; const int* cnt; int64_t seg_stride, row_stride, cnt_seg_stride; int nseg;
    const float* hD; int64_t ld_hD; int HD; const float* W32D; const float* biasD; const float* epsD;
    const float* hT; int64_t ld_hT; int HT; const float* W32T; const float* biasT; const float* alphaT; const float* betaT;
    const unsigned* fhat; const float* w_t; const float* w_p; const int* row_bad;
    const int32_t* seed_row_ptr; int k; int n_valid_col;
    uint2* out; int* out_cnt; int out_cap;
    int* guard;                        // {violations, a violating column}
    int* stat;                         // [B][2] {candidates, recomputed}
    long long* stamps;                 // experiments build: stage stamps of row 0's workgroup (DAE_DBG_MR)
    // FUSED SELECTION (round 5, as refine.hip; k <= 512): the launch ends the call -- the row's seeds out, its k best mixed
    // scores in order (main_challenge.py:26-36 on DAEs.py:180's y) to fo.out_score / out_idx; `out` then only takes the rows
    // with more survivors than the ordering stage holds
    int fuse;
    dae_rank_out fo;
    const int32_t* seed_col;
};

// One group of 64 candidates, a lane each: the canonical chain acc = fmaf(hrow[k], W32[rowidx][k], acc), k = 0 .. H-1, from
// +0 (oracle orc_decode; what v_mfma_f32_32x32x2_f32 performs in the fp32 kernels).  The rows are fetched quad-wise -- the
// 4 lanes of a quad read 64 contiguous bytes of one row per instruction -- and handed to their lanes through the wave's
// LDS buffer; the hidden factors travel by v_readlane (refine.hip rescore_group, which this restates for two scorers).
// H % 16 == 0; hrow zero beyond H up to the next multiple of 64.
template <int DEPTH>
__device__ __forceinline__ float mix_chain64(const float* __restrict__ W32, int H, const float* hrow, float* tbuf, int lane, int rowidx)
{
    const int Q = lane >> 2, q = lane & 3;
    const int H16 = H >> 4;
    const float4* rp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = __shfl(rowidx, 4 * Q + i);
        rp[i] = reinterpret_cast<const float4*>(W32 + (size_t)ci * H) + q;
    }
    float4 v[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[d][i] = d < H16 ? rp[i][4 * d] : make_float4(0.f, 0.f, 0.f, 0.f);
    float acc = 0.0f;
    constexpr int U = DEPTH < 4 ? 4 : DEPTH;
    for (int j0 = 0; j0 < H16; j0 += U) {
        float hq[U / 4];
#pragma unroll
        for (int c = 0; c < U / 4; ++c) hq[c] = hrow[16 * j0 + 64 * c + lane];
#pragma unroll
        for (int dd = 0; dd < U; ++dd) {
            const int jb = j0 + dd;
            const int d = dd & (DEPTH - 1);
            if (jb < H16) {                                       // wave-uniform
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(tbuf + (4 * Q + i) * MR_ROWSTRIDE + 4 * q) = v[d][i];
                __builtin_amdgcn_wave_barrier();
                // the ring is refilled in PAIRS of blocks: the two halves of a row's 128-byte line are requested back to back, so
                // the second finds the line pending instead of fetching it again (refine.hip rescore_group has the measurement)
                if (dd & 1) {
                    const int dp = (dd - 1) & (DEPTH - 1);
                    const int jn0 = jb - 1 + DEPTH, jn1 = jb + DEPTH;
                    if (jn0 < H16) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[dp][i] = rp[i][4 * jn0];
                    }
                    if (jn1 < H16) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[d][i] = rp[i][4 * jn1];
                    }
                }
                float4 w[4];
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) w[tq] = *reinterpret_cast<const float4*>(tbuf + lane * MR_ROWSTRIDE + 4 * tq);
                __builtin_amdgcn_wave_barrier();
                const int hc = (int)__float_as_uint(hq[dd >> 2]);
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    const int l0 = 16 * (dd & 3) + 4 * tq;
                    acc = fmaf(__uint_as_float((unsigned)__builtin_amdgcn_readlane(hc, l0)), w[tq].x, acc);
                    acc = fmaf(__uint_as_float((unsigned)__builtin_amdgcn_readlane(hc, l0 + 1)), w[tq].y, acc);
                    acc = fmaf(__uint_as_float((unsigned)__builtin_amdgcn_readlane(hc, l0 + 2)), w[tq].z, acc);
                    acc = fmaf(__uint_as_float((unsigned)__builtin_amdgcn_readlane(hc, l0 + 3)), w[tq].w, acc);
                }
            }
        }
    }
    return acc;
}

// two floats further from the value than the rounded subtraction left it
__device__ __forceinline__ float two_down(float x) { return dae_okey_inv(dae_okey(x) - 2u); }

// (The library is built with -fno-vectorize: hipcc's loop vectorizer miscompiled this kernel's per-lane staging loop in round 4 --
// spotify_recsys_challenge_2018_amd/build.py, scripts/probe/vec_repro.hip.)
#ifdef DAE_EXPERIMENTS
#define MSTAMP(i) if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0) p.stamps[i] = __builtin_readcyclecounter();
#else
#define MSTAMP(i)
#endif
__global__ __launch_bounds__(MR_THREADS, 1) void mix_refine_kernel(const MixRefP p)
{
    __shared__ int seg_prefix[MR_MAX_SEG + 2];
    __shared__ float hrowD[1024];
    __shared__ float hrowT[1024];
    __shared__ __attribute__((aligned(16))) float tb[MR_WAVES * 64 * MR_ROWSTRIDE];      // the waves' buffers; before: the histogram
    // (static LDS beyond 64 KiB compiles, but its arrays then overlap at run time: the two key arrays are the launch's
    // dynamic LDS, as refine.hip's staging area)
    extern __shared__ __attribute__((aligned(16))) unsigned mr_dyn[];
    unsigned* kl = mr_dyn;                       // keys of the lower bounds; afterwards the survivors' flat indices
    unsigned* ku = mr_dyn + MR_STAGE;            // keys of the upper bounds
    __shared__ unsigned cnts[32];
    __shared__ int s_n;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const int nseg = p.nseg;
    const bool bad = p.row_bad && p.row_bad[row] != 0;
    // fused selection: the row's seeds are requested with the prologue's loads (one per thread in a register; playlists with
    // more than MR_THREADS seed tracks read the rest when the bitmap is built)
    const int seed_b = (p.fuse && p.seed_col) ? p.seed_row_ptr[row] : 0;
    const int seed_e = (p.fuse && p.seed_col) ? p.seed_row_ptr[row + 1] : 0;
    const int my_seed = seed_b + tid < seed_e ? p.seed_col[seed_b + tid] : -1;
    __shared__ unsigned f_range[2];              // {min, max} of the high words of the keys written to LDS
    for (int s = tid; s < nseg; s += MR_THREADS) seg_prefix[s + 1] = p.cnt[(size_t)s * p.cnt_seg_stride + row];
    if (tid == 0) { seg_prefix[0] = 0; s_n = 0; f_range[0] = 0xFFFFFFFFu; f_range[1] = 0u; }
    if (tid < 32) cnts[tid] = (tid == 1 || tid == 3) ? 0xFFFFFFFFu : 0u;      // [1], [3]: minima
    for (int i = tid; i < 1024; i += MR_THREADS) {
        hrowD[i] = i < p.HD ? p.hD[(size_t)row * p.ld_hD + i] : 0.0f;
        hrowT[i] = i < p.HT ? p.hT[(size_t)row * p.ld_hT + i] : 0.0f;
    }
    __syncthreads();
    if (tid < 64) {
        int carry = 0;
        for (int b0 = 0; b0 < nseg; b0 += 64) {
            const int i = b0 + tid;
            int v = i < nseg ? seg_prefix[i + 1] : 0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(v, d);
                if (tid >= d) v += o;
            }
            if (i < nseg) seg_prefix[i + 1] = v + carry;
            carry += __shfl(v, 63);
        }
    }
    __syncthreads();
    const int total = seg_prefix[nseg];
    MSTAMP(0)
    auto give_up = [&](int code) {
        // a row the launch cannot vouch for (more candidates than its buffers hold): counted like a bound failure -- the
        // callers then re-score the launch with the fp32 kernels
        if (tid == 0) { p.out_cnt[row] = 0; atomicAdd(p.guard, 1); p.guard[1] = code; }
        if (p.fuse) dae_rank_pad<MR_THREADS>(tid, row, 0u, p.fo);
    };
    // ---- fused selection: LDS of the upper-bound keys once they are dead (after step 3; the streamed path never uses them):
    // [keys of the ordering stage: DAE_RANK_MAX x 8 B][seed bitmap over the rankable columns: <= 24 KiB]
    dae_u64* const fkey = reinterpret_cast<dae_u64*>(ku);
    unsigned* const bitmap = ku + 2 * DAE_RANK_MAX;
    const int bm_words = p.fuse ? (p.n_valid_col + 31) >> 5 : 0;
    auto build_bitmap = [&]() {                                   // (every thread; the callers' barrier before it freed the area)
        for (int w = tid; w < bm_words; w += MR_THREADS) bitmap[w] = 0u;
        __syncthreads();
        for (int i = seed_b + tid; i < seed_e; i += MR_THREADS) {
            const int pc = i == seed_b + tid ? my_seed : p.seed_col[i];
            if (pc >= 0 && pc < p.n_valid_col) atomicOr(&bitmap[pc >> 5], 1u << (pc & 31));
        }
        __syncthreads();
    };
    auto fkey_of = [&](float y, int colv) -> dae_u64 {            // composite key of a score; 0 = absent (a seed, -inf)
        const unsigned key = dae_okey(y);
        if (colv < 0 || key <= DAE_KEY_NEG_INF) return 0ull;
        if (colv < p.n_valid_col && ((bitmap[colv >> 5] >> (colv & 31)) & 1u)) return 0ull;
        return ((dae_u64)key << 32) | (dae_u64)(~(unsigned)colv);
    };
    // the ordering stage's buffers: the waves' transposition buffers, free once the recomputation is over
    dae_u64* const sorted = reinterpret_cast<dae_u64*>(tb);
    unsigned* const above = reinterpret_cast<unsigned*>(sorted + DAE_RANK_MAX);
    unsigned* const fhist = above + DAE_RANK_BINS;
    uint2* const orow0 = p.out + (size_t)row * p.out_cap;
    auto select_from_list = [&](int n_list) {                     // the row's list in global memory -> its final lists
        dae_rank_select_emit<MR_THREADS>([&](auto f) {
            for (int e0 = 0; e0 < n_list; e0 += MR_THREADS) {
                const int e = e0 + tid;
                dae_u64 ck = 0ull;
                if (e < n_list) { const uint2 pr = orow0[e]; ck = fkey_of(__uint_as_float(pr.x), (int)pr.y); }
                f(ck);
            }
        }, fkey, sorted, fhist, above, tid, row, p.fo);
    };
    if (!bad && p.w_t[row] == 0.0f && p.w_p[row] == 0.0f) {
        // y = sigmoid(.) * 0 + sigmoid(.) * 0 = +0 for every column: the fp32 path ranks (y desc, column asc), i.e. the first
        // k non-seed columns -- the first k + n_seeds columns cover them
        const int nl = min(p.k + (p.seed_row_ptr ? p.seed_row_ptr[row + 1] - p.seed_row_ptr[row] : 0), min(p.n_valid_col, p.out_cap));
        for (int i = tid; i < nl; i += MR_THREADS) p.out[(size_t)row * p.out_cap + i] = make_uint2(0u, (unsigned)i);
        if (tid == 0) { p.out_cnt[row] = nl; if (p.stat) { p.stat[2 * row] = 0; p.stat[2 * row + 1] = 0; } }
        if (p.fuse) { build_bitmap(); select_from_list(nl); }
        return;
    }
    if (bad || total == 0) {
        if (tid == 0) { p.out_cnt[row] = 0; if (p.stat) { p.stat[2 * row] = total; p.stat[2 * row + 1] = 0; } }
        if (p.fuse) dae_rank_pad<MR_THREADS>(tid, row, 0u, p.fo);
        return;
    }
    const int need = p.k + (p.seed_row_ptr ? p.seed_row_ptr[row + 1] - p.seed_row_ptr[row] : 0);
    const float wt = p.w_t[row], wp = p.w_p[row];
    const float F = __uint_as_float(p.fhat[row] << 16);

    // ---- 1. the mixed bounds of every candidate, as order-preserving keys.  Flat over the row's candidates, four per thread
    // and round with their loads issued together: the entry first, then the three bound terms of its column -- two trips to
    // memory per round.  (A wave per segment, 64 entries at a time, made every 64 entries two DEPENDENT trips: 16 of them
    // per wave and row.)
    auto offset_of = [&](int e) -> int64_t {
        int lo = 0, hi = nseg;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (seg_prefix[mid] <= e) lo = mid; else hi = mid;
        }
        return (int64_t)lo * p.seg_stride + (int64_t)row * p.row_stride + (e - seg_prefix[lo]);
    };
    constexpr int MR_U = 4;
    // body(i, key of l, key of u) for every candidate i of the round starting at c0
    auto bounds_round = [&](int c0, auto body) {
        uint4 en[MR_U];
#pragma unroll
        for (int q = 0; q < MR_U; ++q) {
            const int i = c0 + q * MR_THREADS + tid;
            en[q] = p.base[offset_of(i < total ? i : 0)];
        }
        float al[MR_U], be[MR_U], ep[MR_U];
#pragma unroll
        for (int q = 0; q < MR_U; ++q) {
            const int col = (int)en[q].z;
            al[q] = p.alphaT[col]; be[q] = p.betaT[col]; ep[q] = p.epsD[col];
        }
#pragma unroll
        for (int q = 0; q < MR_U; ++q) {
            const int i = c0 + q * MR_THREADS + tid;
            const float uT = __uint_as_float(en[q].x), uD = __uint_as_float(en[q].y);
            const float wdT = 2.0f * fmaf(al[q], F, be[q]) * 1.000001f;
            const float wdD = 2.0f * ep[q] * 1.000001f;
            // (the narrowing only needs bounds of the canonical value: the hardware's exp2 / rcp, 2^-15 wider)
            const float up = mix_fast_up(uT, uD, wt, wp);
            const float lo = mix_fast_dn(two_down(uT - wdT), two_down(uD - wdD), wt, wp);
            body(i, i < total, dae_okey(lo), dae_okey(up));
        }
    };
    const bool staged = total <= MR_STAGE;                        // both keys of every candidate fit LDS
    unsigned kmx = 0u, kmn = 0xFFFFFFFFu;
    if (staged) {
        for (int c0 = 0; c0 < total; c0 += MR_U * MR_THREADS)
            bounds_round(c0, [&](int i, bool in, unsigned a, unsigned u) {
                if (in) { kl[i] = a; ku[i] = u; kmx = a > kmx ? a : kmx; kmn = a < kmn ? a : kmn; }
            });
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned a = __shfl_xor(kmx, d), b = __shfl_xor(kmn, d);
        kmx = a > kmx ? a : kmx; kmn = b < kmn ? b : kmn;
    }
    MSTAMP(1)
    if (lane == 0) { atomicMax(&cnts[0], kmx); atomicMin(&cnts[1], kmn); }
    unsigned* hist = reinterpret_cast<unsigned*>(tb);
    for (int i = tid; i < MR_BINS; i += MR_THREADS) hist[i] = 0u;
    __syncthreads();

    // ---- 2. tau': the need-th largest lower bound (to a 2048-bin histogram over [min, max] of the row's keys and one pass
    // for the smallest key of the selected bin, as refine.hip): `need` distinct columns provably reach it
    unsigned P = 0u;                                             // key of tau' (0: everything survives)
    // the bin that holds the need-th largest key of the histogram, counted from the top (block-uniform, via cnts[2])
    auto select_bin = [&](int need_) -> int {                     // (cnts[5]: how many keys lie in the bins above it)
        constexpr int BPT = MR_BINS / MR_THREADS;
        const int top = MR_BINS - 1 - BPT * tid;
        unsigned hc[BPT], own = 0;
#pragma unroll
        for (int e = 0; e < BPT; ++e) { hc[e] = hist[top - e]; own += hc[e]; }
        unsigned incl = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) cnts[8 + wave] = incl;
        __syncthreads();
        unsigned pre = 0;
        for (int w = 0; w < wave; ++w) pre += cnts[8 + w];
        incl += pre;
        const unsigned excl = incl - own;
        if (excl < (unsigned)need_ && (unsigned)need_ <= incl) {
            unsigned run = excl;
            bool done = false;
#pragma unroll
            for (int e = 0; e < BPT; ++e) {
                if (!done && run + hc[e] >= (unsigned)need_) { cnts[2] = (unsigned)(top - e); cnts[5] = run; done = true; }
                run += hc[e];
            }
        }
        __syncthreads();
        return (int)cnts[2];
    };
    int* surv = reinterpret_cast<int*>(kl);                       // the survivors' flat indices (the staged path: in kl's place)
    auto append = [&](bool keep, int i) {                         // (whole waves call this)
        const unsigned long long bal = __ballot(keep);
        if (bal) {
            const int leader = __ffsll((long long)bal) - 1;
            int b = 0;
            if (lane == leader) b = atomicAdd(&s_n, __popcll(bal));
            b = __shfl(b, leader);
            const int at = b + __popcll(bal & ((1ull << lane) - 1ull));
            if (keep && at < 2 * MR_STAGE) surv[at] = i;          // (beyond out_cap <= 2 MR_STAGE the row gives up below)
        }
    };
    if (!staged) {
        // STREAMED (a flat DAE bias: the threshold sample says little and a row lists tens of thousands of columns): nothing
        // is staged.  Pass A: histogram of the lower bounds over FIXED bins (1/32 of an octave of y between 2^-40 and 1),
        // A2 inside the bin holding the need-th largest; the lower edge of the sub-bin found is tau'.  Pass B: the bounds
        // again, survivors listed.
        constexpr unsigned K0 = 0x80000000u | 0x2B800000u;        // key of 2^-40
        auto fbin = [&](unsigned key) -> int {
            if (key < K0) return 0;
            const unsigned bq = (key - K0) >> 18;
            return (int)(bq < (unsigned)(MR_BINS - 1) ? bq : (unsigned)(MR_BINS - 1));
        };
        for (int c0 = 0; c0 < total; c0 += MR_U * MR_THREADS)
            bounds_round(c0, [&](int, bool in, unsigned a, unsigned) { if (in) atomicAdd(&hist[fbin(a)], 1u); });
        __syncthreads();
        if (total > need) {
            const int Bsel = select_bin(need);
            P = Bsel > 0 ? K0 + ((unsigned)Bsel << 18) : 0u;
            if (Bsel > 0 && Bsel < MR_BINS - 1) {
                // ... refined: with a flat bias thousands of bounds share that bin (2 % of y wide).  Pass A2: the keys of the
                // selected bin alone, in 2048 sub-bins of 128 key units (1.5e-5 of y)
                const int need2 = need - (int)cnts[5];
                __syncthreads();                                  // (everybody has read cnts[5] and the histogram)
                for (int i = tid; i < MR_BINS; i += MR_THREADS) hist[i] = 0u;
                __syncthreads();
                const unsigned e0 = P;
                for (int c0 = 0; c0 < total; c0 += MR_U * MR_THREADS)
                    bounds_round(c0, [&](int, bool in, unsigned a, unsigned) {
                        if (in && fbin(a) == Bsel) atomicAdd(&hist[(a - e0) >> 7], 1u);
                    });
                __syncthreads();
                const int B2 = select_bin(need2);
                P = e0 + ((unsigned)B2 << 7);
            }
        }
        for (int c0 = 0; c0 < total; c0 += MR_U * MR_THREADS)
            bounds_round(c0, [&](int i, bool in, unsigned, unsigned u) { append(in && u >= P, i); });
    } else if (total > need) {
        const unsigned kmin = cnts[1], kmax = cnts[0];
        const float scale = 2047.999f / ((float)(kmax - kmin) + 1.0f);
        auto bin_of = [&](unsigned key) -> int {
            const unsigned bq = (unsigned)((float)(key - kmin) * scale);
            return (int)(bq < (unsigned)(MR_BINS - 1) ? bq : (unsigned)(MR_BINS - 1));
        };
        for (int i = tid; i < total; i += MR_THREADS) atomicAdd(&hist[bin_of(kl[i])], 1u);
        __syncthreads();
        const int Bsel = select_bin(need);
        unsigned kb = 0xFFFFFFFFu;
        for (int i = tid; i < total; i += MR_THREADS) {
            const unsigned key = kl[i];
            if (bin_of(key) == Bsel) kb = key < kb ? key : kb;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const unsigned o = __shfl_xor(kb, d); kb = o < kb ? o : kb; }
        if (lane == 0 && kb != 0xFFFFFFFFu) atomicMin(&cnts[3], kb);
        __syncthreads();
        if (cnts[3] != 0xFFFFFFFFu) P = cnts[3];
    }
    __syncthreads();                                             // (the last reads of kl and of the histogram)

    MSTAMP(2)
    // ---- 3. the survivors: upper bound >= tau'.  Their flat indices take kl's place.
    if (staged) {
        for (int c0 = 0; c0 < total; c0 += MR_THREADS) {
            const int i = c0 + tid;
            append(i < total && ku[i] >= P, i);                   // (slot <= i: the list never overtakes the keys still to be read)
        }
    }
    __syncthreads();
    const int n = s_n;
    if (tid == 0 && p.stat) { p.stat[2 * row] = total; p.stat[2 * row + 1] = n; }
    if (n > p.out_cap) { give_up(-3); return; }
    MSTAMP(3)
    const bool fast = p.fuse && n <= DAE_RANK_MAX;               // the survivors' keys stay in LDS
    if (p.fuse) build_bitmap();

    // ---- 4. recompute the survivors
    float* tbuf = tb + wave * (64 * MR_ROWSTRIDE);
    uint2* orow = p.out + (size_t)row * p.out_cap;
    const int gstep = MR_WAVES * 64;
    uint4 pr = make_uint4(0u, 0u, 0u, 0u);
    int g0 = wave * 64;
    if (g0 < n) pr = p.base[offset_of(surv[g0 + lane < n ? g0 + lane : g0])];
    for (; g0 < n; g0 += gstep) {
        const int e = g0 + lane;
        const bool in = e < n;
        const uint4 cur = pr;
        const int gn = g0 + gstep;
        if (gn < n) pr = p.base[offset_of(surv[gn + lane < n ? gn + lane : gn])];       // the next group's entries, under this group's rows
        const int col = (int)cur.z;
        const float zT = mix_chain64<MR_DEPTH>(p.W32T, p.HT, hrowT, tbuf, lane, col) + p.biasT[col];
        const float zD = mix_chain64<MR_DEPTH>(p.W32D, p.HD, hrowD, tbuf, lane, col) + p.biasD[col];
        const float y = mixf(zT, zD, wt, wp);
        if (in) {
            // BOUND GUARD: the filter launch promised u - width <= z32 <= u for both logits of every listed column
            const float uT = __uint_as_float(cur.x), uD = __uint_as_float(cur.y);
            const float wdT = 2.0f * fmaf(p.alphaT[col], F, p.betaT[col]) * 1.000001f;
            const float wdD = 2.0f * p.epsD[col] * 1.000001f;
            const bool ok = zT <= uT && zT >= two_down(uT - wdT) && zD <= uD && zD >= two_down(uD - wdD);
            if (!ok) { atomicAdd(p.guard, 1); p.guard[1] = col; }
            if (!fast) orow[e] = make_uint2(__float_as_uint(y), (unsigned)col);
        }
        if (fast) {                                               // (whole waves: the range of the keys by two DPP reductions)
            const dae_u64 ck = in ? fkey_of(y, col) : 0ull;
            if (in) fkey[e] = ck;
            const unsigned hi = (unsigned)(ck >> 32);
            const unsigned mx = dae_wave_max_u32(hi), mn = dae_wave_min_u32(ck != 0ull ? hi : 0xFFFFFFFFu);
            if (lane == 0 && mx != 0u) { atomicMin(&f_range[0], mn); atomicMax(&f_range[1], mx); }
        }
    }
    MSTAMP(4)
    if (tid == 0) p.out_cnt[row] = n;
    if (!p.fuse) return;
    __syncthreads();                                             // the recomputation is over: its buffers become the ordering stage's
    if (fast) {
        for (int b = tid; b < DAE_RANK_BINS; b += MR_THREADS) fhist[b] = 0u;
        dae_rank_emit<MR_THREADS>(fkey, (unsigned)n, sorted, fhist, above, tid, row, p.fo, f_range);
    } else {
        select_from_list(n);
    }
}
#undef MSTAMP

// ---- the audit of dropped columns (round 6; the plain exact mode's is audit.hip) ---------------------------------------------
// The refine launch's guard sees survivors only; a column the filter launch dropped is never recomputed.  Every N-th launch
// (dae_set_exact_audit on the TITLE context) a random set of ranked tiles is recomputed in fp32 for every row -- z_t and z_d
// with the canonical chains (audit.hip's kernel, both images), mixed as the lists' scores are -- and tested against the
// lists THIS launch wrote: an element whose score lies above the row's k-th listed score must be in the list or among the
// row's seeds.  Anything else is a column the filter dropped although it belongs in the list: counted in the title context's
// guard words (DAE_title.recommend / dae_pipeline_poll re-score such a launch with the fp32 kernels, as for a survivor).
// This tests the OUTCOME (main_challenge.py:26-36 on DAEs.py:180's y), whatever the bound's shape; ties with the k-th score
// are not judged.
struct MixAuditP {
    const float* zt; const float* zd; int64_t ld_z;  // [B][n_tiles * 32] canonical logits of the sampled tiles
    const int* tiles; int n_tiles; int B, k, n_valid_col;
    const float* w_t; const float* w_p; const int* row_bad;
    const float* out_score; const int32_t* out_idx;  // the launch's lists [B][k]
    const int32_t* seed_row_ptr; const int32_t* seed_col;
    int* guard; unsigned long long* stat;
};

__global__ __launch_bounds__(256) void mix_audit_kernel(const MixAuditP p)
{
    const int per_row = p.n_tiles * 32;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(e / per_row), j = (int)(e % per_row);
    bool counted = false, bad = false;
    if (row < p.B && !p.row_bad[row]) {
        const float wt = p.w_t[row], wp = p.w_p[row];
        const int col = p.tiles[j >> 5] * 32 + (j & 31);
        const float zt = p.zt[(size_t)row * p.ld_z + j], zd = p.zd[(size_t)row * p.ld_z + j];
        if (col < p.n_valid_col && zt > -__builtin_inff() && zd > -__builtin_inff() && !(wt == 0.0f && wp == 0.0f)) {
            counted = true;
            const float s = mixf(zt, zd, wt, wp);
            const size_t last = (size_t)row * p.k + (p.k - 1);
            const float thr = p.out_idx[last] < 0 ? -__builtin_inff() : p.out_score[last];      // (a short list: every column belongs)
            if (s > thr) {
                bool found = false;
                for (int i = 0; i < p.k && !found; ++i) found = p.out_idx[(size_t)row * p.k + i] == col;
                if (!found && p.seed_row_ptr)
                    for (int i = p.seed_row_ptr[row]; i < p.seed_row_ptr[row + 1] && !found; ++i) found = p.seed_col[i] == col;
                bad = !found;
                if (bad) { atomicAdd(p.guard, 1); p.guard[1] = col; }
            }
        }
    }
    const unsigned nc = (unsigned)__popcll(__ballot(counted)), nb = (unsigned)__popcll(__ballot(bad));
    if ((threadIdx.x & 63) == 0 && nc) {
        atomicAdd(p.stat, (unsigned long long)nc);
        if (nb) atomicAdd(p.stat + 1, (unsigned long long)nb);
    }
}

}  // namespace

static int mix_audit(dae_ctx* tc, dae_ctx* dc, const float* feat, int64_t ld_feat, const float* h, int64_t ld_h, int B,
                     const float* w_title, const float* w_playlist, int n_valid_col, const int32_t* seed_row_ptr,
                     const int32_t* seed_col, int k, const float* out_score, const int32_t* out_idx)
{
    if (tc->audit_every <= 0 || tc->audit_tiles <= 0 || !out_score || !out_idx) return DAE_OK;
    if ((++tc->audit_seq % (uint64_t)tc->audit_every) != 0) return DAE_OK;
    const dae_packed& pt = tc->pk_bf16;
    const dae_packed& pd = dc->pk_bf16;
    const int n_tiles = tc->audit_tiles > 64 ? 64 : tc->audit_tiles;
    unsigned long long* stat; int* tiles;
    int rc = dae_audit_pick_tiles(tc, (n_valid_col + 31) / 32, n_tiles, &stat, &tiles);
    if (rc) return rc;
    const size_t per = (size_t)B * n_tiles * 32;
    rc = dae_reserve(tc, tc->audit, 2 * per * sizeof(float));
    if (rc) return rc;
    float* zt = static_cast<float*>(tc->audit.p);
    float* zd = zt + per;
    rc = dae_launch_audit_chains(tc, feat, ld_feat, pt.H, static_cast<const float*>(pt.W32.p), static_cast<const float*>(pt.bias.p),
                                 pt.col_hi - pt.col_lo, n_valid_col, B, tiles, n_tiles, zt);
    if (rc) return rc;
    rc = dae_launch_audit_chains(tc, h, ld_h, pd.H, static_cast<const float*>(pd.W32.p), static_cast<const float*>(pd.bias.p),
                                 pd.col_hi - pd.col_lo, n_valid_col, B, tiles, n_tiles, zd);
    if (rc) return rc;
    MixAuditP a;
    a.zt = zt; a.zd = zd; a.ld_z = (int64_t)n_tiles * 32; a.tiles = tiles; a.n_tiles = n_tiles; a.B = B; a.k = k;
    a.n_valid_col = n_valid_col; a.w_t = w_title; a.w_p = w_playlist; a.row_bad = static_cast<const int*>(tc->row_bad.p);
    a.out_score = out_score; a.out_idx = out_idx; a.seed_row_ptr = seed_row_ptr; a.seed_col = seed_col;
    a.guard = static_cast<int*>(tc->guard.p); a.stat = stat;
    const int64_t total = (int64_t)per;
    hipLaunchKernelGGL(mix_audit_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, tc->stream, a);
    DAE_CHECK_LAUNCH(tc, "mix_audit_kernel");
    ++tc->audits_run;
    return DAE_OK;
}

// ---- host side ----------------------------------------------------------------------------------------------------------
int dae_launch_mix_title_bounds(dae_ctx* ctx, const float* W, const float* b, int H, int Hp, int col_lo, int col_hi,
                                int ntiles, dae_packed& pk)
{
    int rc = dae_reserve(ctx, pk.mix_alpha, (size_t)ntiles * 32 * sizeof(float));
    if (rc) return rc;
    rc = dae_reserve(ctx, pk.mix_beta, (size_t)ntiles * 32 * sizeof(float));
    if (rc) return rc;
    rc = dae_reserve(ctx, pk.mix16_lo, (size_t)ntiles * 64 * sizeof(uint4));
    if (rc) return rc;
    rc = dae_reserve(ctx, pk.mix16_hi, (size_t)ntiles * 64 * sizeof(uint4));
    if (rc) return rc;
    const int blocks = ntiles < 8 * DAE_NUM_CU ? ntiles : 8 * DAE_NUM_CU;
    hipLaunchKernelGGL(mix_title_bounds_kernel, dim3(blocks), dim3(256), 0, ctx->stream, W, b, H, Hp, col_lo, col_hi, ntiles,
                       static_cast<float*>(pk.mix_alpha.p), static_cast<float*>(pk.mix_beta.p),
                       static_cast<uint4*>(pk.mix16_lo.p), static_cast<uint4*>(pk.mix16_hi.p), ctx->exact_margin,
                       ctx->margin_lo, ctx->margin_hi, ctx->margin_scale);
    DAE_CHECK_LAUNCH(ctx, "mix_title_bounds_kernel");
    return DAE_OK;
}

int dae_mix_topk_exact_impl(dae_ctx* tc, dae_ctx* dc, const float* feat, int64_t ld_feat, const float* h, int64_t ld_h, int B,
                            const float* w_title, const float* w_playlist, int n_tracks, const int32_t* seed_row_ptr,
                            const int32_t* seed_col, int k, float* out_score, int32_t* out_idx)
{
    constexpr int NSD = 16, NST = 28, RB = MX_RB, NW = MX_NW, QR = MX_QR, NS = NSD + NST, R_TILE = RB * 32;
    dae_packed& pt = tc->pk_bf16;
    dae_packed& pd = dc->pk_bf16;
    if (!pt.valid || !pt.exact || !pd.valid || !pd.exact)
        return dae_fail(tc, DAE_ERR_STATE, "both contexts need their weights prepacked with DAE_DTYPE_BF16_EXACT");
    if (pd.Hp != NSD * 16 || pt.Hp != NST * 16 || pd.H != pd.Hp || (pt.H & 15))
        return dae_fail(tc, DAE_ERR_ARG, "the exact title mix is built for hidden 256 (DAE) and 448-wide feature rows "
                        "(got %d and %d): use DAE_DTYPE_F32", pd.H, pt.H);
    if (pt.col_lo != 0 || pd.col_lo != 0 || pt.col_hi != pd.col_hi)
        return dae_fail(tc, DAE_ERR_ARG, "both images must hold the same columns, starting at 0");
    if (B <= 0) return DAE_OK;
    if (B > 4096) return dae_fail(tc, DAE_ERR_ARG, "at most 4096 rows per call (got %d)", B);
    hipStream_t st = tc->stream;
    int rc;
    const int n_valid_col = n_tracks < pt.col_hi ? n_tracks : pt.col_hi;
    const int ntr = (n_valid_col + 31) / 32;                      // tiles with rankable columns
    if (ntr <= 0) return dae_fail(tc, DAE_ERR_ARG, "no rankable column");

    // geometry: 96-row groups; the workgroups of a row group in multiples of the XCD count
    const int n_rg = (B + R_TILE - 1) / R_TILE;
    const int Bpad = n_rg * R_TILE;
    int nb = (DAE_NUM_CU / n_rg) / DAE_NUM_XCD * DAE_NUM_XCD;
    if (nb < DAE_NUM_XCD) nb = DAE_NUM_XCD;
    while (nb > DAE_NUM_XCD && (nb - DAE_NUM_XCD) * NW >= ntr) nb -= DAE_NUM_XCD;     // small images: no workgroups without a tile
    const int grid = n_rg * nb;

    // tiles by their largest DAE bias (the popularity prior): the head of the list is the threshold sample
    // (an eighth of the tiles, in whole rounds of 64, at least 256)
    int n_samp = (ntr / 8) / 64 * 64;
    if (n_samp < 256) n_samp = 256;
    if (n_samp > ntr) n_samp = ntr;
    if (dc->stream != st) return dae_fail(tc, DAE_ERR_STATE, "both contexts must be bound to the same stream");
    if (!(pd.order.p && pd.order_nrank == n_valid_col && pd.ntiles <= 8192)) {      // (the bias order does not depend on the sample size)
        rc = dae_launch_tile_order(dc, pd, n_valid_col, n_samp, 1);
        if (rc) return dae_fail(tc, rc, "%s", dc->err.c_str());
    }
    const int* order = static_cast<const int*>(pd.order.p);

    // per row: F_r, preconditions; the packed hidden rows
    rc = dae_reserve(tc, tc->row_bad, (size_t)Bpad * sizeof(int)); if (rc) return rc;
    rc = dae_reserve(tc, tc->mix_fhat, (size_t)Bpad * sizeof(unsigned)); if (rc) return rc;
    rc = dae_reserve(tc, tc->h_packed16, (size_t)n_rg * NS * RB * 64 * sizeof(uint4)); if (rc) return rc;
    tc->h16_geom_key = -1;                                       // (the buffer no longer holds a plain image's pad region)
    {
        const size_t total = (size_t)n_rg * NS * RB * 64;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4 * DAE_NUM_CU) blocks = 4 * DAE_NUM_CU;
        const int prep_blocks = (Bpad + 3) / 4;
        hipLaunchKernelGGL(mix_prep_pack_kernel, dim3(prep_blocks + blocks), dim3(256), 0, st, h, ld_h, pd.H, feat, ld_feat, pt.H,
                           w_title, w_playlist, B, Bpad, static_cast<unsigned*>(tc->mix_fhat.p), static_cast<int*>(tc->row_bad.p),
                           prep_blocks, NSD, NS, RB, n_rg, static_cast<uint4*>(tc->h_packed16.p));
        DAE_CHECK_LAUNCH(tc, "mix_prep_pack_kernel");
    }

    MixP p;
    memset(&p, 0, sizeof(p));
    p.WqD = static_cast<const uint4*>(pd.W.p); p.WqT = static_cast<const uint4*>(pt.W.p);
    p.hp = static_cast<const uint4*>(tc->h_packed16.p);
    p.fhat = static_cast<const unsigned*>(tc->mix_fhat.p);
    p.w_t = w_title; p.w_p = w_playlist;
    p.n_valid_col = n_valid_col;
    p.B = B; p.n_rg = n_rg; p.nb_rg = nb; p.Bpad = Bpad;
    const size_t lds = (size_t)RB * 64 * NS * sizeof(uint4) + (size_t)R_TILE * 8 + 16;
    auto kf = mix_bf16_kernel<NSD, NST, RB, QR, NW, 0>;
    auto ks = mix_bf16_kernel<NSD, NST, RB, QR, NW, 1>;
#ifdef DAE_EXPERIMENTS
    static const int exp_mode = dae_exp_env("DAE_MIX_EXP") ? atoi(dae_exp_env("DAE_MIX_EXP")) : 0;   // stage bisection of the filter launch
    p.exp_mode = exp_mode;
    static const int qr_env = dae_exp_env("DAE_MIX_QR") ? atoi(dae_exp_env("DAE_MIX_QR")) : 0;      // A/B: W ring depth
    if (qr_env == 12) { kf = mix_bf16_kernel<NSD, NST, RB, 12, NW, 0>; ks = mix_bf16_kernel<NSD, NST, RB, 12, NW, 1>; }
    if (qr_env == 16) { kf = mix_bf16_kernel<NSD, NST, RB, 16, NW, 0>; ks = mix_bf16_kernel<NSD, NST, RB, 16, NW, 1>; }
    if (qr_env == 4) { kf = mix_bf16_kernel<NSD, NST, RB, 4, NW, 0>; ks = mix_bf16_kernel<NSD, NST, RB, 4, NW, 1>; }
#endif
    static const char attr_key = 0;
    if (dae_first_use(tc, &attr_key)) {
        DAE_HIP_CHECK(tc, hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DAE_HIP_CHECK(tc, hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }

    // ---- sample: lower bounds over the head of the list -> tau
    const int64_t ld_s = (int64_t)n_samp * 8;
    rc = dae_reserve(tc, tc->gmax, (size_t)Bpad * ld_s * sizeof(float)); if (rc) return rc;
    rc = dae_reserve(tc, tc->tau, (size_t)Bpad * sizeof(float)); if (rc) return rc;
    p.biasD = static_cast<const uint4*>(pd.bias16_lo.p); p.biasT = static_cast<const uint4*>(pt.mix16_lo.p);
    p.list = order; p.n_items = n_samp;
    p.samp = static_cast<float*>(tc->gmax.p); p.ld_s = ld_s;
    {
        // the sample pass on a grid of its own when other launches are in flight (round 6, as the plain path's sample launch:
        // api.hip topk_phase_a): an eighth of the tiles does not need a workgroup on every CU -- ~8 items per wave slot; its
        // maxima are stored per ITEM, so the geometry changes nothing but where they are computed
        int nb_s = nb;
        if (tc->overlap_hint) {
            nb_s = ((n_samp + 8 * NW - 1) / (8 * NW) + DAE_NUM_XCD - 1) / DAE_NUM_XCD * DAE_NUM_XCD;
            if (nb_s > nb) nb_s = nb;
        }
        static const int nbs_env = dae_exp_env("DAE_MIX_SAMPLE_NB") ? atoi(dae_exp_env("DAE_MIX_SAMPLE_NB")) : 0;       // A/B (experiments build)
        if (nbs_env > 0) nb_s = nbs_env >= nb ? nb : nbs_env / DAE_NUM_XCD * DAE_NUM_XCD;
        if (nb_s < DAE_NUM_XCD) nb_s = DAE_NUM_XCD;
        MixP ps = p;
        ps.nb_rg = nb_s;
        hipLaunchKernelGGL(ks, dim3(n_rg * nb_s), dim3(NW * 64), lds, st, ps);
    }
    DAE_CHECK_LAUNCH(tc, "mix_bf16_kernel<sample>");
    rc = dae_reserve(tc, tc->sample_top, (size_t)Bpad * (sizeof(uint2) + sizeof(int))); if (rc) return rc;
    rc = dae_launch_tau_select(tc, p.samp, ld_s, (int)ld_s, p.samp, 0, 0, order, 0, B, k, seed_row_ptr,
                               static_cast<float*>(tc->tau.p), static_cast<uint2*>(tc->sample_top.p), 1,
                               reinterpret_cast<int*>(static_cast<uint2*>(tc->sample_top.p) + Bpad));        // (no sample list: count 0)
    if (rc) return rc;

    // ---- filter: upper bounds over every rankable tile
    const int cap = ((ntr + nb - 1) / nb) * 32;
    rc = dae_reserve(tc, tc->cand, (size_t)nb * Bpad * cap * sizeof(uint4)); if (rc) return rc;
    rc = dae_reserve(tc, tc->cand_cnt, (size_t)nb * Bpad * sizeof(int)); if (rc) return rc;
    p.biasD = static_cast<const uint4*>(pd.bias16_hi.p); p.biasT = static_cast<const uint4*>(pt.mix16_hi.p);
    p.list = order; p.n_items = ntr;
    p.tau = static_cast<const float*>(tc->tau.p);
    p.cand = static_cast<uint4*>(tc->cand.p); p.cand_cnt = static_cast<int*>(tc->cand_cnt.p); p.cap = cap;
    p.samp = nullptr; p.ld_s = 0;
    hipLaunchKernelGGL(kf, dim3(grid), dim3(NW * 64), lds, st, p);
    DAE_CHECK_LAUNCH(tc, "mix_bf16_kernel<filter>");

    // ---- refine + selection
    if (!tc->guard.p) {
        rc = dae_reserve(tc, tc->guard, DAE_GUARD_BYTES); if (rc) return rc;
        DAE_HIP_CHECK(tc, hipMemsetAsync(tc->guard.p, 0, DAE_GUARD_BYTES, st));
    }
    rc = dae_reserve(tc, tc->refined, (size_t)Bpad * MX_REF_CAP * sizeof(uint2) + (size_t)Bpad * sizeof(int)); if (rc) return rc;
    uint2* rf = static_cast<uint2*>(tc->refined.p);
    int* rf_cnt = reinterpret_cast<int*>(rf + (size_t)Bpad * MX_REF_CAP);
    rc = dae_reserve(tc, tc->refstat, (size_t)Bpad * 2 * sizeof(int)); if (rc) return rc;
    tc->refstat_rows = B;
    MixRefP r;
    memset(&r, 0, sizeof(r));
    r.base = p.cand; r.cnt = p.cand_cnt; r.seg_stride = (int64_t)Bpad * cap; r.row_stride = cap; r.cnt_seg_stride = Bpad; r.nseg = nb;
    r.hD = h; r.ld_hD = ld_h; r.HD = pd.H; r.W32D = static_cast<const float*>(pd.W32.p);
    r.biasD = static_cast<const float*>(pd.bias.p); r.epsD = static_cast<const float*>(pd.eps.p);
    r.hT = feat; r.ld_hT = ld_feat; r.HT = pt.H; r.W32T = static_cast<const float*>(pt.W32.p);
    r.biasT = static_cast<const float*>(pt.bias.p);
    r.alphaT = static_cast<const float*>(pt.mix_alpha.p); r.betaT = static_cast<const float*>(pt.mix_beta.p);
    r.fhat = p.fhat; r.w_t = w_title; r.w_p = w_playlist; r.row_bad = static_cast<const int*>(tc->row_bad.p);
    r.seed_row_ptr = seed_row_ptr; r.k = k; r.n_valid_col = n_valid_col;
    r.out = rf; r.out_cnt = rf_cnt; r.out_cap = MX_REF_CAP;
    r.guard = static_cast<int*>(tc->guard.p); r.stat = static_cast<int*>(tc->refstat.p);
    // k <= 512 and a seed bitmap that fits behind the ordering keys: the refine launch ends the call (no selection launch)
    const bool fuse = k <= DAE_RANK_MAX / 2 && ((size_t)((n_valid_col + 31) >> 5) + 2 * DAE_RANK_MAX) <= (size_t)MR_STAGE;
    r.fuse = fuse ? 1 : 0;
    r.fo = dae_rank_out{k, DAE_OUT_LOGIT, out_score, out_idx};   // (the mixed score is a probability already: it goes out as it is)
    r.seed_col = seed_col;
    if (nb > MR_MAX_SEG) return dae_fail(tc, DAE_ERR_ARG, "too many candidate segments (%d)", nb);
#ifdef DAE_EXPERIMENTS
    static const bool dbg_mr = dae_exp_env("DAE_DBG_MR") != nullptr;
    static long long* mr_stamps = nullptr;
    static int mr_calls = 0;
    if (dbg_mr) { if (!mr_stamps) (void)hipMalloc(&mr_stamps, 8 * 8); r.stamps = mr_stamps; }
#endif
    static const char ref_key = 0;
    if (dae_first_use(tc, &ref_key))
        DAE_HIP_CHECK(tc, hipFuncSetAttribute(reinterpret_cast<const void*>(mix_refine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)(2 * MR_STAGE * sizeof(unsigned))));
    hipLaunchKernelGGL(mix_refine_kernel, dim3(B), dim3(MR_THREADS), 2 * MR_STAGE * sizeof(unsigned), st, r);
    DAE_CHECK_LAUNCH(tc, "mix_refine_kernel");
#ifdef DAE_EXPERIMENTS
    if (dbg_mr && (++mr_calls % 4) == 0) {
        long long hst[8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hst, mr_stamps, sizeof(hst), hipMemcpyDeviceToHost);
        fprintf(stderr, "MIX_REFINE row0 cycles: bounds %lld | threshold %lld | list %lld | recompute %lld\n", hst[1] - hst[0], hst[2] - hst[1],
                hst[3] - hst[2], hst[4] - hst[3]);
    }
#endif

    if (!fuse) {
        dae_topk_args ta;
    memset(&ta, 0, sizeof(ta));
    ta.B = B; ta.k = k;
    ta.bitmap_base = 0; ta.bitmap_n = n_valid_col;
    ta.seed_row_ptr = seed_row_ptr; ta.seed_col = seed_col;
    ta.out_kind = DAE_OUT_LOGIT;                                 // the mixed score is a probability already: it goes out as it is
    ta.out_score = out_score; ta.out_idx = out_idx;
    dae_pair_group gr{rf, rf_cnt, 0, MX_REF_CAP, 0, 1, 0};
    dae_pair_group g_none{nullptr, nullptr, 0, 0, 0, 0, 0};
    rc = dae_launch_topk_pairs(tc, gr, g_none, ta);
    if (rc) return rc;
    }
    return mix_audit(tc, dc, feat, ld_feat, h, ld_h, B, w_title, w_playlist, n_valid_col, seed_row_ptr, seed_col, k, out_score, out_idx);
}
