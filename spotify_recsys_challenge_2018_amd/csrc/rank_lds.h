// rank_lds.h -- the tail of the ranking (main_challenge.py:26-36: argsort descending, drop the seeds, keep k) for a row whose
// candidates already sit in LDS as unique composite keys  (okey(score) << 32) | ~column  -- the order topk.hip defines:
// score descending, column ascending.  Used by the kernels that end a scoring call with the row's candidates in hand
// (refine.hip: the exact mode's recomputed survivors; mixexact.hip: the exact title mix), so that no separate selection
// launch has to read them back.  Same stages as topk_kernel's step 5a (order by histogram rank), same results.
#pragma once
#include "dae_internal.h"

typedef unsigned long long dae_u64;

constexpr int DAE_RANK_BINS = 2048;
constexpr int DAE_RANK_MAX = 1024;      // keys one call orders (k <= 512: what a narrowing to "fits the sort buffer" leaves)

struct dae_rank_out {
    int k, out_kind;                    // out_kind: DAE_OUT_SCORE (sigmoid of the key's float) / DAE_OUT_LOGIT (the float itself)
    float* out_score; int32_t* out_idx; // [rows][k], nullable
};

__device__ __forceinline__ dae_u64 dae_wave_min_u64(dae_u64 v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const dae_u64 o = __shfl_xor(v, d); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ dae_u64 dae_wave_max_u64(dae_u64 v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const dae_u64 o = __shfl_xor(v, d); v = o > v ? o : v; }
    return v;
}

// Wave-wide maximum / minimum of a 32-bit value on the DPP path (quad permutes and row mirrors inside the rows of 16 lanes, the
// four row leaders through v_readlane): ~10 instructions, no LDS -- six ds_bpermute round trips (__shfl_xor) cost ~1 k cycles on
// the last wave to finish.  EVERY lane of the wave must be active; the result is wave-uniform.
__device__ __forceinline__ unsigned dae_wave_max_u32(unsigned v)
{
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); v = t > v ? t : v;      // quad_perm [1,0,3,2]
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); v = t > v ? t : v;      // quad_perm [2,3,0,1]
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); v = t > v ? t : v;     // row_half_mirror
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false); v = t > v ? t : v;     // row_mirror
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                   c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
__device__ __forceinline__ unsigned dae_wave_min_u32(unsigned v) { return ~dae_wave_max_u32(~v); }

// rows without a single candidate: the reference's cand[:k] of an empty list, padded as topk_kernel pads short rows
template <int NTH>
__device__ __forceinline__ void dae_rank_pad(int tid, int row, unsigned from, const dae_rank_out& o)
{
    for (unsigned i = from + (unsigned)tid; i < (unsigned)o.k; i += NTH) {
        const size_t at = (size_t)row * o.k + i;
        if (o.out_idx) o.out_idx[at] = -1;
        if (o.out_score) o.out_score[at] = -__builtin_inff();
    }
}

// Block-wide (NTH threads, every thread calls): which bin, scanning from the top, holds the `need`-th element, and how many
// elements sit in bins above it.  Thread t owns the BPT bins from 2047 - BPT t downwards.  (topk.hip find_bin.)
template <int NTH>
__device__ __forceinline__ void dae_rank_find_bin(const unsigned* hist, unsigned* wave_tot, int tid, unsigned need, int* s_bin,
                                                  unsigned* s_above)
{
    constexpr int BPT = DAE_RANK_BINS / NTH;
    const int top = DAE_RANK_BINS - 1 - BPT * tid;
    unsigned c[BPT];
    unsigned own = 0;
#pragma unroll
    for (int e = 0; e < BPT; ++e) { c[e] = hist[top - e]; own += c[e]; }
    unsigned v = own;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    if (lane == 63) wave_tot[wv] = v;
    __syncthreads();
    unsigned pre = 0;
    for (int w = 0; w < wv; ++w) pre += wave_tot[w];
    const unsigned incl = pre + v, excl = incl - own;
    if (excl < need && need <= incl) {
        unsigned run = excl;
        bool done = false;
#pragma unroll
        for (int e = 0; e < BPT; ++e) {
            if (!done && run + c[e] >= need) { *s_bin = top - e; *s_above = run; done = true; }
            run += c[e];
        }
    }
    __syncthreads();
}

// keys[0 .. c), c <= DAE_RANK_MAX: unique composite keys, 0 = absent.  Emits the min(k, present) largest in descending order
// to row `row` of the outputs and pads the rest.  LDS scratch: sorted[DAE_RANK_MAX], hist[DAE_RANK_BINS],
// above[DAE_RANK_BINS]; `keys` itself is overwritten.  Every thread of the NTH-thread workgroup calls; starts with a barrier.
// range (nullable, LDS): {a lower bound, an upper bound} of the present keys' HIGH words ({0xFFFFFFFF, 0}: none present), kept
// by the caller while it produced the keys -- the caller has then ALSO zeroed hist[] before the call; both save a pass and two
// barriers here (stage stamps: 3.5 k of the ordering's 9.2 k cycles at 16 waves).
template <int NTH>
__device__ __forceinline__ void dae_rank_emit(dae_u64* keys, unsigned c, dae_u64* sorted, unsigned* hist, unsigned* above,
                                              int tid, int row, const dae_rank_out& o, const unsigned* range = nullptr,
                                              long long* dbg = nullptr)
{
#ifdef DAE_EXPERIMENTS         // stage stamps of workgroup 0 (the caller's debug switch hands the buffer in)
#define RKSTAMP(i) if (dbg && blockIdx.x == 0 && tid == 0) dbg[i] = __builtin_readcyclecounter();
#else
#define RKSTAMP(i)
#endif
    constexpr int PER = DAE_RANK_MAX / NTH, BPT = DAE_RANK_BINS / NTH, NW = NTH / 64;
    __shared__ dae_u64 rk_mm[2];
    __shared__ unsigned rk_wave_tot[NW];
    __shared__ unsigned rk_total;
    const int lane = tid & 63;
    __syncthreads();                                             // keys complete; the scratch regions' last readers are done
    RKSTAMP(0)
    dae_u64 mine[PER];
    dae_u64 mn = ~0ull, mx = 0ull;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const unsigned i = (unsigned)(e * NTH + tid);
        mine[e] = i < c ? keys[i] : 0ull;
        if (mine[e] != 0ull) { mn = mine[e] < mn ? mine[e] : mn; mx = mine[e] > mx ? mine[e] : mx; }
    }
    dae_u64 rlo, rhi;
    if (range) {                                                 // (block-uniform)
        rlo = (dae_u64)range[0] << 32;
        rhi = range[1] ? ((dae_u64)range[1] << 32) | 0xFFFFFFFFull : 0ull;
    } else {
        if (tid == 0) { rk_mm[0] = ~0ull; rk_mm[1] = 0ull; }
        for (int b = tid; b < DAE_RANK_BINS; b += NTH) hist[b] = 0u;
        __syncthreads();
        mn = dae_wave_min_u64(mn); mx = dae_wave_max_u64(mx);
        if (lane == 0 && mx != 0ull) { atomicMin(&rk_mm[0], mn); atomicMax(&rk_mm[1], mx); }
        __syncthreads();
        rlo = rk_mm[0]; rhi = rk_mm[1];
    }
    RKSTAMP(1)
    if (rhi == 0ull) { dae_rank_pad<NTH>(tid, row, 0u, o); return; }       // nothing present (block-uniform)
    // bins linear in the SCORE, not in its bit pattern (topk.hip step 5a has the measurement behind this); when the scores'
    // range is degenerate -- every key shares its float -- the bit pattern of the whole key (the column breaks the ties)
    int shift = 64 - 11 - __clzll((rhi - rlo) | 1ull);
    if (shift < 0) shift = 0;
    const float zlo = dae_okey_inv((unsigned)(rlo >> 32)), zhi = dae_okey_inv((unsigned)(rhi >> 32));
    const float zspan = zhi - zlo;
    const bool lin = zspan > 1e-30f && zspan < 3.0e38f;
    const float zscale = lin ? (float)(DAE_RANK_BINS - 1) / zspan : 0.0f;
    unsigned mbin[PER], mpos[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        unsigned bn = 0u;
        if (mine[e] != 0ull) {
            if (lin) {
                const float zz = dae_okey_inv((unsigned)(mine[e] >> 32));
                bn = (unsigned)fminf(fmaxf((zz - zlo) * zscale, 0.0f), (float)(DAE_RANK_BINS - 1));
            } else {
                bn = (unsigned)((mine[e] - rlo) >> shift);
            }
        }
        mbin[e] = bn;
        mpos[e] = mine[e] != 0ull ? atomicAdd(&hist[bn], 1u) : 0u;
    }
    __syncthreads();
    RKSTAMP(2)
    {   // above[b] = keys in bins > b
        const int top = DAE_RANK_BINS - 1 - BPT * tid;
        unsigned cb[BPT], own = 0;
#pragma unroll
        for (int e = 0; e < BPT; ++e) { cb[e] = hist[top - e]; own += cb[e]; }
        unsigned v = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned ov = __shfl_up(v, d);
            if (lane >= d) v += ov;
        }
        if (lane == 63) rk_wave_tot[tid >> 6] = v;
        __syncthreads();
        unsigned run = v - own;
        for (int w = 0; w < (tid >> 6); ++w) run += rk_wave_tot[w];
#pragma unroll
        for (int e = 0; e < BPT; ++e) { above[top - e] = run; run += cb[e]; }
        if (tid == NTH - 1) rk_total = run;
    }
    __syncthreads();
    RKSTAMP(3)
    const unsigned present = rk_total;
    const unsigned k_eff = present < (unsigned)o.k ? present : (unsigned)o.k;
#pragma unroll
    for (int e = 0; e < PER; ++e)
        if (mine[e] != 0ull) sorted[above[mbin[e]] + mpos[e]] = mine[e];
    __syncthreads();
    RKSTAMP(4)
    unsigned rk[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        rk[e] = 0xFFFFFFFFu;
        if (mine[e] == 0ull) continue;
        const unsigned base = above[mbin[e]], nb = hist[mbin[e]];
        unsigned rank = base;
        for (unsigned i = 0; i < nb; ++i) rank += sorted[base + i] > mine[e] ? 1u : 0u;
        rk[e] = rank;
    }
    // the winners in rank order through LDS, then out in rank order: thread i writes position i (whole lines per wave)
    RKSTAMP(5)
    dae_u64* fin = keys;                                         // (every thread took its keys into registers above)
#pragma unroll
    for (int e = 0; e < PER; ++e)
        if (rk[e] < k_eff) fin[rk[e]] = mine[e];
    __syncthreads();
    RKSTAMP(6)
    for (unsigned i = (unsigned)tid; i < k_eff; i += NTH) {
        const dae_u64 key = fin[i];
        const float z = dae_okey_inv((unsigned)(key >> 32));
        const size_t at = (size_t)row * o.k + i;
        if (o.out_idx) o.out_idx[at] = (int)(~(unsigned)(key & 0xFFFFFFFFull));
        if (o.out_score) o.out_score[at] = o.out_kind == DAE_OUT_SCORE ? dae_sigmoidf(z) : z;
    }
    dae_rank_pad<NTH>(tid, row, k_eff, o);
    RKSTAMP(7)
#undef RKSTAMP
}

// The general case: the row's candidates sit in global memory, possibly more than DAE_RANK_MAX of them.  for_keys(f) calls
// f(key) for every candidate (0 = absent; whole workgroup, block-uniform trip count -- it may be called several times and
// re-reads its source each time).  topk_kernel's range-adaptive narrowing (topk.hip step 3b) until the keys above the cut fit,
// the collect into keys[], then dae_rank_emit.  Rare by construction (logits packed within the bounds' width of the cut).
// LDS scratch as dae_rank_emit.  Starts with a barrier.
template <int NTH, typename ForKeys>
__device__ __forceinline__ void dae_rank_select_emit(ForKeys for_keys, dae_u64* keys, dae_u64* sorted, unsigned* hist, unsigned* above,
                                                     int tid, int row, const dae_rank_out& o)
{
    __shared__ unsigned rs_cnt, rs_above;
    __shared__ int rs_bin;
    __shared__ dae_u64 rs_min, rs_max;
    __shared__ unsigned rs_wave_tot[NTH / 64];
    const int lane = tid & 63;
    __syncthreads();
    if (tid == 0) { rs_cnt = 0u; rs_min = ~0ull; rs_max = 0ull; }
    __syncthreads();
    {
        unsigned cnt = 0; dae_u64 mn = ~0ull, mx = 0ull;
        for_keys([&](dae_u64 ck) {
            if (ck != 0ull) { ++cnt; mn = ck < mn ? ck : mn; mx = ck > mx ? ck : mx; }
        });
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
        mn = dae_wave_min_u64(mn); mx = dae_wave_max_u64(mx);
        if (lane == 0 && cnt) { atomicAdd(&rs_cnt, cnt); atomicMin(&rs_min, mn); atomicMax(&rs_max, mx); }
    }
    __syncthreads();
    const unsigned m = rs_cnt;
    const unsigned k_rank = m < (unsigned)o.k ? m : (unsigned)o.k;
    dae_u64 lo = rs_min, hi = rs_max;
    unsigned abv = 0;                                            // keys > hi
    if (m > (unsigned)DAE_RANK_MAX) {
        for (int it = 0; it < 8; ++it) {
            int shift = 64 - 11 - __clzll((hi - lo) | 1ull);
            if (shift < 0) shift = 0;
            for (int b = tid; b < DAE_RANK_BINS; b += NTH) hist[b] = 0u;
            __syncthreads();
            for_keys([&](dae_u64 ck) {
                if (ck != 0ull && ck >= lo && ck <= hi) atomicAdd(&hist[(unsigned)((ck - lo) >> shift)], 1u);
            });
            __syncthreads();
            dae_rank_find_bin<NTH>(hist, rs_wave_tot, tid, k_rank - abv, &rs_bin, &rs_above);
            const unsigned b = (unsigned)rs_bin;
            const unsigned cnt_b = hist[b];
            const unsigned new_abv = abv + rs_above;
            const dae_u64 nlo = lo + ((dae_u64)b << shift);
            dae_u64 nhi = nlo + ((1ull << shift) - 1ull);
            if (nhi > hi) nhi = hi;
            __syncthreads();                                     // hist / rs_bin consumed
            lo = nlo;
            if (new_abv + cnt_b <= (unsigned)DAE_RANK_MAX) break;       // the keys >= lo fit (unique keys: at shift 0 a bin holds one)
            hi = nhi;
            abv = new_abv;
        }
    }
    if (tid == 0) rs_cnt = 0u;
    __syncthreads();
    for_keys([&](dae_u64 ck) {
        if (ck != 0ull && ck >= lo) {
            const unsigned slot = atomicAdd(&rs_cnt, 1u);
            if (slot < (unsigned)DAE_RANK_MAX) keys[slot] = ck;
        }
    });
    __syncthreads();
    const unsigned c = rs_cnt < (unsigned)DAE_RANK_MAX ? rs_cnt : (unsigned)DAE_RANK_MAX;
    dae_rank_emit<NTH>(keys, c, sorted, hist, above, tid, row, o);
}
