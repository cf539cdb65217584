// topk.hip -- K3/K4: exact top-k ranking of one playlist row per workgroup.
// Reference: main_challenge.py:26-36 (cand_generate) == metrics.py:59-68 (single_eval ranking):
//     cand = argsort(-scores); for s in seed: cand.remove(s); cand = cand[:500]
// numpy's argsort leaves tie order unspecified, so the canonical rule is
//     (fp32 logit descending, column index ascending)
// implemented as an exact MSB radix select on the order-preserving key of the logit (3 passes of
// 11/11/10 bits), an index radix select only when the k-th logit is tied across the cut, a
// collect pass, and a bitonic sort of the <= 1024 survivors on the 64-bit composite key.
// Seed tracks are removed through a per-row LDS bitmap over the ranked column range (the
// reference's O(n) list.remove per seed becomes one LDS bit test per candidate).
//
// Two element sources share the code:
//   dense : a row of logits (phase-A sample buffer of the fused path, dae_topk_dense)
//   pairs : segments of (logit, column) pairs (phase-B candidate lists, shard merge K4)
#include "dae_internal.h"

namespace {

constexpr int TK_THREADS = 256;
constexpr int TK_BINS = 2048;
constexpr int TK_MAX_SEG = 1024;

struct DenseSrc {
    dae_dense_src s;
    template <typename F>
    __device__ __forceinline__ void for_each(int row, int tid, int* /*seg_prefix*/, F f) const
    {
        const float* rp = s.logits + (size_t)row * s.ld;
        const int step = 32 * s.tile_stride;
        for (int p = tid; p < s.n; p += TK_THREADS) {
            const int colv = s.col_base + (p >> 5) * step + (p & 31);
            f(rp[p], colv);
        }
    }
    __device__ __forceinline__ void prepare(int, int, int*) const {}
};

struct PairSrc {
    dae_pair_group g0, g1;

    __device__ __forceinline__ int seg_count(const dae_pair_group& g, int seg, int row) const
    {
        return g.cnt ? g.cnt[(size_t)seg * g.cnt_seg_stride + row] : g.fixed_cnt;
    }
    // exclusive prefix of segment sizes in LDS: seg_prefix[0..nseg], nseg = g0.nseg + g1.nseg
    __device__ __forceinline__ void prepare(int row, int tid, int* seg_prefix) const
    {
        const int nseg = g0.nseg + g1.nseg;
        for (int s = tid; s < nseg; s += TK_THREADS)
            seg_prefix[s + 1] = s < g0.nseg ? seg_count(g0, s, row) : seg_count(g1, s - g0.nseg, row);
        if (tid == 0) seg_prefix[0] = 0;
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int s = 1; s <= nseg; ++s) { run += seg_prefix[s]; seg_prefix[s] = run; }
        }
        __syncthreads();
    }
    template <typename F>
    __device__ __forceinline__ void for_each(int row, int tid, int* seg_prefix, F f) const
    {
        const int nseg = g0.nseg + g1.nseg;
        const int total = seg_prefix[nseg];
        for (int e = tid; e < total; e += TK_THREADS) {
            int lo = 0, hiq = nseg;                 // largest s with seg_prefix[s] <= e
            while (hiq - lo > 1) {
                const int mid = (lo + hiq) >> 1;
                if (seg_prefix[mid] <= e) lo = mid; else hiq = mid;
            }
            const int i = e - seg_prefix[lo];
            const dae_pair_group& g = lo < g0.nseg ? g0 : g1;
            const int seg = lo < g0.nseg ? lo : lo - g0.nseg;
            const uint2 pr = g.base[(size_t)seg * g.seg_stride + (size_t)row * g.row_stride + i];
            const int colv = (int)pr.y;
            f(colv < 0 ? -__builtin_inff() : __uint_as_float(pr.x), colv);
        }
    }
};

// shard lists gathered by RCCL: logit[(g*B + row)*k + i], idx[...]  (K4 merge)
struct SoaSrc {
    const float* logit; const int32_t* idx; int G, B, k;
    __device__ __forceinline__ void prepare(int, int, int*) const {}
    template <typename F>
    __device__ __forceinline__ void for_each(int row, int tid, int*, F f) const
    {
        const int total = G * k;
        for (int e = tid; e < total; e += TK_THREADS) {
            const int g = e / k, i = e - g * k;
            const size_t o = ((size_t)g * B + row) * k + i;
            const int colv = idx[o];
            f(colv < 0 ? -__builtin_inff() : logit[o], colv);
        }
    }
};

// Find, scanning bins from the top, the bin where the running count reaches `need`.
// hist[TK_BINS] in LDS; returns bin, count strictly above it (via LDS scalars).
__device__ __forceinline__ void find_bin(const unsigned* hist, unsigned* part, int tid,
                                         unsigned need, int* s_bin, unsigned* s_above)
{
    // part[t] = sum of the 8 bins owned by thread t (bins 8t .. 8t+7)
    unsigned s = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) s += hist[tid * 8 + b];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0;
        int t = TK_THREADS - 1;
        for (; t > 0; --t) {
            if (run + part[t] >= need) break;
            run += part[t];
        }
        int b = t * 8 + 7;
        for (; b > t * 8; --b) {
            if (run + hist[b] >= need) break;
            run += hist[b];
        }
        *s_bin = b;
        *s_above = run;
    }
    __syncthreads();
}

template <typename Src>
__global__ __launch_bounds__(TK_THREADS) void topk_kernel(const Src src, const dae_topk_args a)
{
    extern __shared__ unsigned dyn_bitmap[];             // ceil(bitmap_n/32) words
    __shared__ unsigned hist[TK_BINS];
    __shared__ unsigned part[TK_THREADS];
    __shared__ unsigned long long skey[DAE_MAX_K];
    __shared__ int seg_prefix[TK_MAX_SEG + 2];
    __shared__ int s_bin;
    __shared__ unsigned s_above;
    __shared__ unsigned s_cnt;

    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    const int k = a.k;

    // ---- seed bitmap ---------------------------------------------------------------------------
    const int bm_words = (a.bitmap_n + 31) >> 5;
    for (int w = tid; w < bm_words; w += TK_THREADS) dyn_bitmap[w] = 0;
    __syncthreads();
    if (a.seed_col && bm_words > 0) {
        const int sb = a.seed_row_ptr[row], se = a.seed_row_ptr[row + 1];
        for (int i = sb + tid; i < se; i += TK_THREADS) {
            const int pcol = a.seed_col[i] - a.bitmap_base;
            if (pcol >= 0 && pcol < a.bitmap_n) atomicOr(&dyn_bitmap[pcol >> 5], 1u << (pcol & 31));
        }
    }
    src.prepare(row, tid, seg_prefix);
    __syncthreads();

    auto key_of = [&](float z, int colv) -> unsigned {
        // 0 = absent (masked seed, -inf padding, missing entry); valid keys are > DAE_KEY_NEG_INF
        if (colv < 0) return 0u;
        const unsigned key = dae_okey(z);
        if (key <= DAE_KEY_NEG_INF) return 0u;
        const int pcol = colv - a.bitmap_base;
        if (pcol >= 0 && pcol < a.bitmap_n && ((dyn_bitmap[pcol >> 5] >> (pcol & 31)) & 1u))
            return 0u;
        return key;
    };

    // ---- radix select on the logit key: 11 + 11 + 10 bits ---------------------------------------
    unsigned prefix = 0;          // decided high bits of the k-th key
    unsigned need = 0;            // how many still to take from the current bucket
    unsigned k_eff = 0;
    {
        for (int b = tid; b < TK_BINS; b += TK_THREADS) hist[b] = 0;
        __syncthreads();
        src.for_each(row, tid, seg_prefix, [&](float z, int colv) {
            const unsigned key = key_of(z, colv);
            if (key) atomicAdd(&hist[key >> 21], 1u);
        });
        __syncthreads();
        // total valid
        unsigned s = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) s += hist[tid * 8 + b];
        part[tid] = s;
        __syncthreads();
        if (tid == 0) {
            unsigned tot = 0;
            for (int t = 0; t < TK_THREADS; ++t) tot += part[t];
            s_cnt = tot;
        }
        __syncthreads();
        const unsigned total_valid = s_cnt;
        k_eff = total_valid < (unsigned)k ? total_valid : (unsigned)k;
        __syncthreads();
    }

    unsigned T = 0;               // k-th key
    unsigned n_eq = 0;            // elements with key == T
    if (k_eff > 0) {
        find_bin(hist, part, tid, k_eff, &s_bin, &s_above);
        prefix = (unsigned)s_bin << 21;
        need = k_eff - s_above;
        __syncthreads();

        // pass B: bits 20..10 among keys with the same top 11 bits
        for (int b = tid; b < TK_BINS; b += TK_THREADS) hist[b] = 0;
        __syncthreads();
        src.for_each(row, tid, seg_prefix, [&](float z, int colv) {
            const unsigned key = key_of(z, colv);
            if (key && (key >> 21) == (prefix >> 21)) atomicAdd(&hist[(key >> 10) & 0x7FFu], 1u);
        });
        __syncthreads();
        find_bin(hist, part, tid, need, &s_bin, &s_above);
        prefix |= (unsigned)s_bin << 10;
        need -= s_above;
        __syncthreads();

        // pass C: bits 9..0 among keys with the same top 22 bits
        for (int b = tid; b < TK_BINS; b += TK_THREADS) hist[b] = 0;
        __syncthreads();
        src.for_each(row, tid, seg_prefix, [&](float z, int colv) {
            const unsigned key = key_of(z, colv);
            if (key && (key >> 10) == (prefix >> 10)) atomicAdd(&hist[key & 0x3FFu], 1u);
        });
        __syncthreads();
        find_bin(hist, part, tid, need, &s_bin, &s_above);
        T = prefix | (unsigned)s_bin;
        need -= s_above;
        n_eq = hist[s_bin];
        __syncthreads();
    }

    // ---- tie at the cut: take the `need` smallest column ids among key == T ----------------------
    unsigned idx_cut = 0xFFFFFFFFu;     // take key == T elements with (unsigned)col <= idx_cut
    if (k_eff > 0 && need < n_eq) {
        // radix select the need-th SMALLEST column: work on inverted ids so "largest" logic applies
        unsigned ipre = 0;
        unsigned ineed = need;
        // pass 1: bits 31..21
        for (int b = tid; b < TK_BINS; b += TK_THREADS) hist[b] = 0;
        __syncthreads();
        src.for_each(row, tid, seg_prefix, [&](float z, int colv) {
            if (key_of(z, colv) == T) atomicAdd(&hist[(~(unsigned)colv) >> 21], 1u);
        });
        __syncthreads();
        find_bin(hist, part, tid, ineed, &s_bin, &s_above);
        ipre = (unsigned)s_bin << 21; ineed -= s_above;
        __syncthreads();
        for (int b = tid; b < TK_BINS; b += TK_THREADS) hist[b] = 0;
        __syncthreads();
        src.for_each(row, tid, seg_prefix, [&](float z, int colv) {
            const unsigned ik = ~(unsigned)colv;
            if (key_of(z, colv) == T && (ik >> 21) == (ipre >> 21))
                atomicAdd(&hist[(ik >> 10) & 0x7FFu], 1u);
        });
        __syncthreads();
        find_bin(hist, part, tid, ineed, &s_bin, &s_above);
        ipre |= (unsigned)s_bin << 10; ineed -= s_above;
        __syncthreads();
        for (int b = tid; b < TK_BINS; b += TK_THREADS) hist[b] = 0;
        __syncthreads();
        src.for_each(row, tid, seg_prefix, [&](float z, int colv) {
            const unsigned ik = ~(unsigned)colv;
            if (key_of(z, colv) == T && (ik >> 10) == (ipre >> 10))
                atomicAdd(&hist[ik & 0x3FFu], 1u);
        });
        __syncthreads();
        find_bin(hist, part, tid, ineed, &s_bin, &s_above);
        ipre |= (unsigned)s_bin;
        idx_cut = ~ipre;                 // column ids are unique, so exactly `need` are <= idx_cut
        __syncthreads();
    }

    // ---- collect the k_eff winners, sort them ----------------------------------------------------
    int npow2 = 1;
    while (npow2 < k) npow2 <<= 1;
    for (int i = tid; i < npow2; i += TK_THREADS) skey[i] = 0ull;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (k_eff > 0) {
        src.for_each(row, tid, seg_prefix, [&](float z, int colv) {
            const unsigned key = key_of(z, colv);
            if (key > T || (key == T && (unsigned)colv <= idx_cut)) {
                const unsigned slot = atomicAdd(&s_cnt, 1u);
                if (slot < (unsigned)npow2)
                    skey[slot] = ((unsigned long long)key << 32) | (unsigned)(~(unsigned)colv);
            }
        });
    }
    __syncthreads();

    // bitonic sort, descending
    for (int size = 2; size <= npow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (npow2 >> 1); i += TK_THREADS) {
                const int lo = (i / stride) * (stride << 1) + (i % stride);
                const int hi2 = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long x = skey[lo], y = skey[hi2];
                if ((x < y) == desc) { skey[lo] = y; skey[hi2] = x; }
            }
            __syncthreads();
        }
    }

    // ---- write ----------------------------------------------------------------------------------
    for (int i = tid; i < k; i += TK_THREADS) {
        const unsigned long long ck = skey[i];
        const bool present = (unsigned)i < k_eff && ck != 0ull;
        const float z = present ? dae_okey_inv((unsigned)(ck >> 32)) : -__builtin_inff();
        const int colv = present ? (int)(~(unsigned)(ck & 0xFFFFFFFFull)) : -1;
        const size_t o = (size_t)row * k + i;
        if (a.out_idx) a.out_idx[o] = colv;
        if (a.out_score)
            a.out_score[o] = (present && a.out_kind == DAE_OUT_SCORE) ? dae_sigmoidf(z) : z;
        if (a.out_pairs) a.out_pairs[o] = make_uint2(__float_as_uint(z), (unsigned)colv);
    }
    if (a.out_tau && tid == 0)
        a.out_tau[row] = (k_eff == (unsigned)k) ? dae_okey_inv(T) : -__builtin_inff();
}

template <typename Src>
int launch_topk(dae_ctx* ctx, const Src& src, const dae_topk_args& a)
{
    if (a.k < 1 || a.k > DAE_MAX_K) return dae_fail(ctx, DAE_ERR_ARG, "k=%d out of [1,%d]", a.k, DAE_MAX_K);
    if (a.B <= 0) return DAE_OK;
    const size_t dyn = (size_t)((a.bitmap_n + 31) / 32) * sizeof(unsigned);
    if (dyn > 128 * 1024)
        return dae_fail(ctx, DAE_ERR_ARG, "ranked column range %d too wide for the LDS seed bitmap",
                        a.bitmap_n);
    static bool attr_set = false;
    if (!attr_set) {
        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_kernel<Src>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               128 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((topk_kernel<Src>), dim3(a.B), dim3(TK_THREADS), dyn, ctx->stream, src, a);
    DAE_CHECK_LAUNCH(ctx, "topk_kernel");
    return DAE_OK;
}

}  // namespace

int dae_launch_topk_dense(dae_ctx* ctx, const dae_dense_src& s, const dae_topk_args& a)
{
    DenseSrc src{s};
    return launch_topk(ctx, src, a);
}

int dae_launch_topk_soa(dae_ctx* ctx, int G, const float* logit, const int32_t* idx,
                        const dae_topk_args& a)
{
    SoaSrc src{logit, idx, G, a.B, a.k};
    return launch_topk(ctx, src, a);
}

int dae_launch_topk_pairs(dae_ctx* ctx, const dae_pair_group& g0, const dae_pair_group& g1,
                          const dae_topk_args& a)
{
    if (g0.nseg + g1.nseg > TK_MAX_SEG)
        return dae_fail(ctx, DAE_ERR_ARG, "too many candidate segments (%d)", g0.nseg + g1.nseg);
    PairSrc src{g0, g1};
    return launch_topk(ctx, src, a);
}
