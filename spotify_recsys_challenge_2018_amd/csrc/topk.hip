// topk.hip -- K3/K4: exact top-k ranking of one playlist row per workgroup.
// Reference: main_challenge.py:26-36 (cand_generate) == metrics.py:59-68 (single_eval ranking):
//     cand = argsort(-scores); for s in seed: cand.remove(s); cand = cand[:500]
// numpy's argsort leaves tie order unspecified, so the canonical rule is
//     (fp32 logit descending, column index ascending)
// i.e. descending order of the UNIQUE 64-bit composite key  (okey(logit) << 32) | ~column.
//
// Algorithm per row (one 1024-thread workgroup, everything after the first read stays in LDS):
//   1. seed tracks -> per-row LDS bitmap over the ranked column range (the reference's O(n)
//      list.remove per seed becomes one LDS bit test per element);
//   2. every valid element's composite key is appended to an LDS key buffer (wave-aggregated
//      append), with the running min / max key;
//   3. range-adaptive MSB radix narrowing: histogram (key - lo) >> shift over 2048 bins with shift
//      chosen from the live range [lo, hi], parallel suffix scan to find the bin holding the
//      k-th key, shrink [lo, hi] to that bin; stop as soon as the keys >= lo fit the sort buffer
//      (keys are unique, so no tie handling exists anywhere);
//   4. collect those keys; k <= 512: order them by HISTOGRAM RANK (position = keys in higher bins + keys of the
//      own bin above; one LDS atomic per key, a suffix scan, a count inside the bin); k > 512: hybrid bitonic sort
//      (wave shuffles + LDS stages); emit k.
// Rows whose elements do not fit the LDS key buffer (dae_topk_dense over a whole vocabulary row)
// run the same steps with step 3 re-reading the source instead of LDS.
//
// Element sources:
//   dense : a row of logits (dae_topk_dense; small problems of the fused path)
//   pairs : segments of (logit, column) pairs (the sample's survivors from tau_select_kernel + phase-B candidate lists)
//   soa   : [G, B, k] shard lists gathered by RCCL (K4 merge)
#include "dae_internal.h"

namespace {

[[maybe_unused]] constexpr int TK_THREADS = 1024;      // shadowed by the per-kernel thread count NTH inside the templates
constexpr int TK_BINS = 2048;
constexpr int TK_MAX_SEG = 1024;
constexpr int TK_SORT_MAX = 2048;
typedef unsigned long long u64;

struct DenseSrc {
    static constexpr int kSegs = 0;
    static constexpr bool kFixedSlots = true;      // for_each visits q = tid, tid + 1024, ... in order
    dae_dense_src s;
    int max_keys() const { return s.n; }                  // host: most keys a row can hold
    template <int NTH> __device__ __forceinline__ void prepare(int, int, int*) const {}
    __device__ __forceinline__ int count(int, const int*) const { return s.n; }
    template <int NTH, typename F>
    __device__ __forceinline__ void for_each(int row, int tid, const int*, F f) const
    {
        constexpr int TK_THREADS = NTH;                // (shadows the file-level constant inside this body)
        const float* rp = s.logits + (size_t)row * s.ld;
        const int step = 32 * s.tile_stride;
        int base = 0;                                  // block-uniform loop bounds
        for (; base + 4 * TK_THREADS <= s.n; base += 4 * TK_THREADS) {
            float z[4];
            int tb[4];                                 // tile bases: loaded with the logits, not after
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = base + u * TK_THREADS + tid;
                z[u] = rp[q];
                tb[u] = s.tile_list ? s.tile_list[q >> 5] * 32 : (q >> 5) * step;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = base + u * TK_THREADS + tid;
                f(z[u], s.col_base + tb[u] + (q & 31), true);
            }
        }
        for (; base < s.n; base += TK_THREADS) {
            const int q = base + tid;
            const bool in = q < s.n;
            const float z = in ? rp[q] : 0.f;
            const int tb = s.tile_list ? s.tile_list[in ? (q >> 5) : 0] * 32 : (q >> 5) * step;
            f(z, s.col_base + tb + (q & 31), in);
        }
    }
};

struct PairSrc {
    static constexpr int kSegs = TK_MAX_SEG;
    static constexpr bool kFixedSlots = false;
    dae_pair_group g0, g1;
    int max_keys() const { return 2048; }                 // typical rows are far below; larger rows re-read
    __device__ __forceinline__ int seg_count(const dae_pair_group& g, int seg, int row) const
    {
        return g.cnt ? g.cnt[(size_t)seg * g.cnt_seg_stride + row] : g.fixed_cnt;
    }
    // exclusive prefix of segment sizes in LDS: seg_prefix[0..nseg]
    template <int NTH>
    __device__ __forceinline__ void prepare(int row, int tid, int* seg_prefix) const
    {
        constexpr int TK_THREADS = NTH;
        const int nseg = g0.nseg + g1.nseg;
        for (int s = tid; s < nseg; s += TK_THREADS)
            seg_prefix[s + 1] = s < g0.nseg ? seg_count(g0, s, row) : seg_count(g1, s - g0.nseg, row);
        if (tid == 0) seg_prefix[0] = 0;
        __syncthreads();
        if (tid < 64) {           // one wave scans (nseg <= 1024: 16 chunks of 64)
            int carry = 0;
            for (int base = 0; base < nseg; base += 64) {
                const int i = base + tid;
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
    }
    __device__ __forceinline__ int count(int, const int* seg_prefix) const
    {
        return seg_prefix[g0.nseg + g1.nseg];
    }
    template <int NTH, typename F>
    __device__ __forceinline__ void for_each(int row, int tid, const int* seg_prefix, F f) const
    {
        constexpr int TK_THREADS = NTH, TK_WAVES = NTH / 64;
        // group 0 (few, long segments: the sample winners): flat over the whole workgroup
        for (int sg = 0; sg < g0.nseg; ++sg) {
            const int cnt = seg_prefix[sg + 1] - seg_prefix[sg];
            const uint2* base = g0.base + (size_t)sg * g0.seg_stride + (size_t)row * g0.row_stride;
            for (int i0 = 0; i0 < cnt; i0 += TK_THREADS) {
                const int i = i0 + tid;
                const bool in = i < cnt;
                const uint2 pr = in ? base[i] : make_uint2(0u, 0xFFFFFFFFu);
                f(__uint_as_float(pr.x), (int)pr.y, in);
            }
        }
        // group 1 (many short segments: one candidate list per decode workgroup): a wave per
        // segment, 4 segments in flight, no per-element search
        const int lane = tid & 63, wave = tid >> 6;
        for (int s0 = wave; s0 < g1.nseg; s0 += 4 * TK_WAVES) {
            int cnt[4];
            const uint2* base[4];
            int mx = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sg = s0 + u * TK_WAVES;
                const bool ok = sg < g1.nseg;
                cnt[u] = ok ? seg_prefix[g0.nseg + sg + 1] - seg_prefix[g0.nseg + sg] : 0;
                base[u] = g1.base + (size_t)(ok ? sg : 0) * g1.seg_stride + (size_t)row * g1.row_stride;
                mx = cnt[u] > mx ? cnt[u] : mx;
            }
            for (int i0 = 0; i0 < mx; i0 += 64) {
                uint2 pr[4];
                bool in[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    in[u] = i0 + lane < cnt[u];
                    pr[u] = in[u] ? base[u][i0 + lane] : make_uint2(0u, 0xFFFFFFFFu);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 < cnt[u]) f(__uint_as_float(pr[u].x), (int)pr[u].y, in[u]);
            }
        }
    }
};

struct SoaSrc {
    static constexpr int kSegs = 0;
    static constexpr bool kFixedSlots = false;
    const float* logit; const int32_t* idx; int G, B, k;
    int max_keys() const { return G * k; }
    template <int NTH> __device__ __forceinline__ void prepare(int, int, int*) const {}
    __device__ __forceinline__ int count(int, const int*) const { return G * k; }
    template <int NTH, typename F>
    __device__ __forceinline__ void for_each(int row, int tid, const int*, F f) const
    {
        constexpr int TK_THREADS = NTH;
        const int total = G * k;
        for (int e0 = 0; e0 < total; e0 += TK_THREADS) {
            const int e = e0 + tid;
            const bool in = e < total;
            float z = 0.f; int colv = -1;
            if (in) {
                const int g = e / k, i = e - g * k;
                const size_t o = ((size_t)g * B + row) * k + i;
                colv = idx[o]; z = logit[o];
            }
            f(z, colv, in);
        }
    }
};

// Block-wide: which bin (scanning from the top) holds the `need`-th element, and how many
// elements sit in bins above it.  Thread t owns the BPT bins from 2047 - BPT t downwards.
template <int NTH>
__device__ __forceinline__ void find_bin(const unsigned* hist, unsigned* wave_tot, int tid,
                                         unsigned need, int* s_bin, unsigned* s_above)
{
    constexpr int BPT = TK_BINS / NTH;
    const int top = TK_BINS - 1 - BPT * tid;
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

// NTH threads per row: 1024 for long rows (a phase-A sample, a dense row), 256 for short ones (candidate
// lists, small shards, gathered shard lists) -- a short row is bound by the fixed cost of every stage
// (histogram clears, bin scans, barriers joining 16 waves), which a quarter of the threads cuts to a
// quarter, and up to 8 such workgroups share a CU.  Same stages, same results.
template <typename Src, int NTH>
__global__ __launch_bounds__(NTH) void topk_kernel(const Src src, const dae_topk_args a,
                                                   const int key_cap, const int dbg_stop)
{
    constexpr int TK_THREADS = NTH, TK_WAVES = NTH / 64;       // shadow the file-level constants
    constexpr int MAXES = 1024 / NTH;                            // per-thread maxima kept for the 3a cut (1024 in all)
    constexpr int EMAX = TK_SORT_MAX / NTH;                      // sort-buffer keys per thread, at most
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ unsigned hist[TK_BINS];
    __shared__ int seg_prefix[Src::kSegs + 2];
    __shared__ unsigned wave_tot[TK_WAVES];
    __shared__ int s_bin;
    __shared__ unsigned s_above;
    __shared__ unsigned s_cnt, s_cnt2, s_slots;
    __shared__ u64 s_min, s_max, s_min2;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int row = blockIdx.x;
    const int k = a.k;
    // dynamic LDS: [seed bitmap (bitmap mode only)] [skey: sort buffer, sort_cap keys] [key cache]
    const bool lean = a.lean != 0;
    const int bm_words = lean ? 0 : (a.bitmap_n + 31) >> 5;
    unsigned* bitmap = reinterpret_cast<unsigned*>(dyn);
    u64* skey = reinterpret_cast<u64*>(dyn + (((size_t)bm_words * 4 + 15) & ~(size_t)15));
    u64* keys = skey + a.sort_cap;

    // ---- 1. seeds -----------------------------------------------------------------------------------
    // bitmap mode: per-row LDS bitmap over the ranked columns, tested for every element.
    // lean mode (small LDS footprint, so that this kernel can share a CU with the decode workgroups
    // of another batch): no bitmap.  With ns seeds in the row, the (k + ns)-th largest key over ALL
    // elements is a valid cut for the k-th largest non-seed key, so the narrowing stages run seed-blind
    // with rank k + ns, and the seeds are removed once, from the <= sort_n keys that were collected.
    // Rows with k + ns above the sort buffer test every element against the seed list instead.
    const int seed_b = a.seed_col ? a.seed_row_ptr[row] : 0;
    const int seed_e = a.seed_col ? a.seed_row_ptr[row + 1] : 0;
    const int ns = seed_e - seed_b;
    for (int w = tid; w < bm_words; w += TK_THREADS) bitmap[w] = 0;
    if (tid == 0) { s_cnt = 0; s_min = ~0ull; s_max = 0ull; s_slots = 0; }
    __syncthreads();
    if (!lean && a.seed_col && bm_words > 0) {
        for (int i = seed_b + tid; i < seed_e; i += TK_THREADS) {
            const int pcol = a.seed_col[i] - a.bitmap_base;
            if (pcol >= 0 && pcol < a.bitmap_n) atomicOr(&bitmap[pcol >> 5], 1u << (pcol & 31));
        }
    }
    src.template prepare<NTH>(row, tid, seg_prefix);
    __syncthreads();
    if (DAE_EXP_ON(dbg_stop == 1)) return;

    int sort_n = 1;
    while (sort_n < k) sort_n <<= 1;
    if (sort_n < 1024) sort_n = 1024;                            // always <= TK_SORT_MAX (k <= 1024)
    if (k > 512) sort_n = TK_SORT_MAX;
    const bool seed_blind = lean && ns > 0 && k + ns <= sort_n;  // remove the seeds after the collect
    const bool seed_scan = lean && ns > 0 && !seed_blind;         // (rare) test every element
    auto is_seed = [&](int colv) -> bool {
        bool hit = false;
        for (int i = seed_b; i < seed_e; ++i) hit = hit || (a.seed_col[i] == colv);   // uniform address
        return hit;
    };

    // composite key of one element; 0 = absent (masked seed, -inf padding, missing entry)
    const float row_min = a.row_min ? a.row_min[row] : -__builtin_inff();
    auto ckey = [&](float z, int colv, bool in) -> u64 {
        if (!in || colv < 0 || z < row_min) return 0ull;
        const unsigned key = dae_okey(z);
        if (key <= DAE_KEY_NEG_INF) return 0ull;
        if (!lean) {
            const int pcol = colv - a.bitmap_base;
            if (pcol >= 0 && pcol < a.bitmap_n && ((bitmap[pcol >> 5] >> (pcol & 31)) & 1u)) return 0ull;
        } else if (seed_scan) {
            if (is_seed(colv)) return 0ull;
        }
        return ((u64)key << 32) | (u64)(~(unsigned)colv);
    };

    // ---- 2. first read: keys -> LDS (if they fit), count, min, max ---------------------------------
    // this thread's largest keys, one per class of its elements (class = visit number mod MAXES): MAXES x NTH
    // = 1024 distinct elements of the row in all
    u64 tmaxv[MAXES];
#pragma unroll
    for (int c = 0; c < MAXES; ++c) tmaxv[c] = 0ull;
    int visit = 0;
    auto note_max = [&](u64 ck) {
#pragma unroll
        for (int c = 0; c < MAXES; ++c)
            if ((visit % MAXES) == c) tmaxv[c] = ck > tmaxv[c] ? ck : tmaxv[c];
        ++visit;
    };
    {
        u64 mn = ~0ull, mx = 0ull;
        const int n_src = src.count(row, seg_prefix);
        if (Src::kFixedSlots && key_cap > 0 && n_src <= key_cap) {
            // dense source: element q goes to cache slot q (0 = absent) -- no compaction, hence no
            // ballot / leader atomic / shuffle per element group; the count is one atomic per wave
            unsigned cnt = 0;
            int calls = 0;
            src.template for_each<NTH>(row, tid, seg_prefix, [&](float z, int colv, bool in) {
                const u64 ck = ckey(z, colv, in);
                note_max(ck);
                const int slot = tid + TK_THREADS * calls++;
                if (slot < n_src) keys[slot] = ck;
                if (ck != 0ull) {
                    ++cnt;
                    mn = ck < mn ? ck : mn;
                    mx = ck > mx ? ck : mx;
                }
            });
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
            if (lane == 0 && cnt) atomicAdd(&s_cnt, cnt);
            if (tid == 0) s_slots = (unsigned)n_src;
        } else if (key_cap > 0) {
            src.template for_each<NTH>(row, tid, seg_prefix, [&](float z, int colv, bool in) {
                const u64 ck = ckey(z, colv, in);
                note_max(ck);
                const bool v = ck != 0ull;
                const u64 bal = __ballot(v);
                if (bal) {
                    const int leader = __ffsll((long long)__ballot(1)) - 1;
                    unsigned base = 0;
                    if (lane == leader) base = atomicAdd(&s_cnt, (unsigned)__popcll(bal));
                    base = __shfl(base, leader);
                    if (v) {
                        const unsigned slot = base + __popcll(bal & ((1ull << lane) - 1ull));
                        if (slot < (unsigned)key_cap) keys[slot] = ck;
                        mn = ck < mn ? ck : mn;
                        mx = ck > mx ? ck : mx;
                    }
                }
            });
        } else {
            // nothing is cached: count, min and max only -- one atomic per wave instead of one per element group
            unsigned cnt = 0;
            src.template for_each<NTH>(row, tid, seg_prefix, [&](float z, int colv, bool in) {
                const u64 ck = ckey(z, colv, in);
                note_max(ck);
                if (ck != 0ull) {
                    ++cnt;
                    mn = ck < mn ? ck : mn;
                    mx = ck > mx ? ck : mx;
                }
            });
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
            if (lane == 0 && cnt) atomicAdd(&s_cnt, cnt);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const u64 omn = __shfl_xor(mn, d), omx = __shfl_xor(mx, d);
            mn = omn < mn ? omn : mn;
            mx = omx > mx ? omx : mx;
        }
        if (lane == 0) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); }
    }
    __syncthreads();
    if (DAE_EXP_ON(dbg_stop == 2)) return;
    const unsigned m = s_cnt;                                   // valid elements
    const unsigned n_cached = s_slots ? s_slots : m;            // cache slots in use (fixed slots: the source size)
    const bool in_lds = key_cap > 0 && n_cached <= (unsigned)key_cap;   // all of them were kept in LDS
    unsigned k_eff = m < (unsigned)k ? m : (unsigned)k;        // outputs of the row (final after seed removal)
    const unsigned k_sel = seed_blind ? (unsigned)(k + ns) : (unsigned)k;
    const unsigned k_rank = m < k_sel ? m : k_sel;               // rank the narrowing stages cut at

    // every valid key, from LDS or (big rows) from the source again.  f(key) is called by WHOLE
    // waves (key == 0 for lanes without an element) so it may use wave-level aggregation.
    auto for_keys = [&](auto f) {
        if (in_lds) {
            for (unsigned i0 = 0; i0 < n_cached; i0 += TK_THREADS) {
                const unsigned i = i0 + tid;
                f(i < n_cached ? keys[i] : 0ull);
            }
        } else {
            src.template for_each<NTH>(row, tid, seg_prefix, [&](float z, int colv, bool in) {
                f(ckey(z, colv, in));
            });
        }
    };
    // histogram update with wave-level aggregation: the bulk of a row's logits falls into a
    // handful of bins, and same-address LDS atomics serialise; up to 6 rounds of "leader's bin:
    // one atomic for all lanes that share it", the (few, scattered) rest as plain atomics.
    auto hist_add = [&](unsigned bin, bool valid) {
        u64 todo = __ballot(valid);
        for (int it = 0; it < 6 && todo; ++it) {
            const int leader = __ffsll((long long)todo) - 1;
            const unsigned lb = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
            const u64 same = __ballot(valid && bin == lb);
            if (lane == leader) atomicAdd(&hist[lb], (unsigned)__popcll(same));
            if (bin == lb) valid = false;
            todo &= ~same;
        }
        if (valid) atomicAdd(&hist[bin], 1u);
    };

    // ---- 3a. cheap lower bound from ONE value per thread --------------------------------------------
    // The per-thread maxima are distinct elements of the row, so their k-th largest is <= the row's
    // k-th largest key: a valid cut.  With the row spread over 1024 threads it is also tight
    // (~1.4 k keys survive for k = 500), and it costs one histogram over <= 1024 values instead of
    // one over the whole row.
    u64 lo = s_min, hi = s_max;
    unsigned above = 0;                                          // keys > hi
    bool narrowed = false;
    if (m > (unsigned)sort_n) {
        if (tid == 0) { s_cnt2 = 0; s_min2 = ~0ull; }
        __syncthreads();
        {
            unsigned nhas = 0;
            u64 wmn = ~0ull;
#pragma unroll
            for (int c = 0; c < MAXES; ++c) {
                nhas += tmaxv[c] != 0ull ? 1u : 0u;
                wmn = (tmaxv[c] != 0ull && tmaxv[c] < wmn) ? tmaxv[c] : wmn;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                const u64 o = __shfl_xor(wmn, d); wmn = o < wmn ? o : wmn;
                nhas += __shfl_xor(nhas, d);
            }
            if (lane == 0 && nhas) { atomicAdd(&s_cnt2, nhas); atomicMin(&s_min2, wmn); }
        }
        __syncthreads();
        const unsigned n_max = s_cnt2;
        if (n_max >= k_rank) {
            const u64 mlo = s_min2, mhi = s_max;                 // the row maximum is a thread maximum
            const u64 range = mhi - mlo;
            int shift = 64 - 11 - __clzll(range | 1ull);
            if (shift < 0) shift = 0;
            for (int b = tid; b < TK_BINS; b += TK_THREADS) hist[b] = 0;
            __syncthreads();
#pragma unroll
            for (int c = 0; c < MAXES; ++c)
                if (tmaxv[c] != 0ull) atomicAdd(&hist[(unsigned)((tmaxv[c] - mlo) >> shift)], 1u);
            __syncthreads();
            find_bin<NTH>(hist, wave_tot, tid, k_rank, &s_bin, &s_above);
            const u64 cut = mlo + ((u64)(unsigned)s_bin << shift);   // <= k-th largest thread maximum
            __syncthreads();
            // count the row's keys >= cut
            unsigned cnt = 0;
            for_keys([&](u64 ck) { cnt += (ck != 0ull && ck >= cut) ? 1u : 0u; });
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
            if (tid == 0) s_cnt2 = 0;
            __syncthreads();
            if (lane == 0) atomicAdd(&s_cnt2, cnt);
            __syncthreads();
            if (s_cnt2 <= (unsigned)sort_n) { lo = cut; narrowed = true; }
            else { lo = cut; }                                   // still a valid cut: narrow inside it
            __syncthreads();
        }
    }

    // ---- 3b. general narrowing of [lo, hi] until the keys >= lo fit the sort buffer ----------------
    if (m > (unsigned)sort_n && !narrowed) {
        for (int it = 0; it < 8; ++it) {
            const u64 range = hi - lo;
            int shift = 64 - 11 - __clzll(range | 1ull);
            if (shift < 0) shift = 0;
            for (int b = tid; b < TK_BINS; b += TK_THREADS) hist[b] = 0;
            __syncthreads();
            for_keys([&](u64 ck) {
                const bool v = ck != 0ull && ck >= lo && ck <= hi;
                hist_add(v ? (unsigned)((ck - lo) >> shift) : 0u, v);
            });
            __syncthreads();
            find_bin<NTH>(hist, wave_tot, tid, k_rank - above, &s_bin, &s_above);
            const unsigned b = (unsigned)s_bin;
            const unsigned cnt_b = hist[b];
            const unsigned new_above = above + s_above;
            const u64 nlo = lo + ((u64)b << shift);
            u64 nhi = nlo + ((1ull << shift) - 1ull);
            if (nhi > hi) nhi = hi;
            __syncthreads();                                     // hist / s_bin consumed
            lo = nlo;
            if (new_above + cnt_b <= (unsigned)sort_n) break;    // keys >= lo fit
            hi = nhi;
            above = new_above;
        }
    }

    if (DAE_EXP_ON(dbg_stop == 3)) return;
    // ---- 4. collect keys >= lo, sort descending, emit -------------------------------------------------
    for (int i = tid; i < sort_n; i += TK_THREADS) skey[i] = 0ull;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (in_lds) {
        for (unsigned i0 = 0; i0 < n_cached; i0 += TK_THREADS) {
            const unsigned i = i0 + tid;
            const u64 ck = i < n_cached ? keys[i] : 0ull;
            const bool v = ck != 0ull && ck >= lo;
            const u64 bal = __ballot(v);
            if (bal) {
                const int leader = __ffsll((long long)__ballot(1)) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(&s_cnt, (unsigned)__popcll(bal));
                base = __shfl(base, leader);
                const unsigned slot = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (v && slot < (unsigned)sort_n) skey[slot] = ck;
            }
        }
    } else {
        src.template for_each<NTH>(row, tid, seg_prefix, [&](float z, int colv, bool in) {
            const u64 ck = ckey(z, colv, in);
            const bool v = ck != 0ull && ck >= lo;
            const u64 bal = __ballot(v);
            if (bal) {
                const int leader = __ffsll((long long)__ballot(1)) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(&s_cnt, (unsigned)__popcll(bal));
                base = __shfl(base, leader);
                const unsigned slot = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (v && slot < (unsigned)sort_n) skey[slot] = ck;
            }
        });
    }
    __syncthreads();

    // ---- 4a. lean mode: drop the seeds among the collected keys (order is irrelevant here) ------------
    if (seed_blind) {
        u64 kk[EMAX];
        bool keep[EMAX];
        const unsigned c = s_cnt < (unsigned)sort_n ? s_cnt : (unsigned)sort_n;
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const unsigned i = (unsigned)(e * TK_THREADS + tid);
            kk[e] = i < c ? skey[i] : 0ull;
            keep[e] = kk[e] != 0ull && !is_seed((int)(~(unsigned)(kk[e] & 0xFFFFFFFFull)));
        }
        __syncthreads();                                         // every skey read is done
        if (tid == 0) s_cnt = 0;
        for (int i = tid; i < sort_n; i += TK_THREADS) skey[i] = 0ull;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const u64 bal = __ballot(keep[e]);
            if (bal) {
                const int leader = __ffsll((long long)bal) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(&s_cnt, (unsigned)__popcll(bal));
                base = __shfl(base, leader);
                if (keep[e]) skey[base + __popcll(bal & ((1ull << lane) - 1ull))] = kk[e];
            }
        }
        __syncthreads();
        // >= k + ns keys were above the cut and <= ns of them were seeds; short rows were collected whole
        k_eff = s_cnt < (unsigned)k ? s_cnt : (unsigned)k;
    }

    if (DAE_EXP_ON(dbg_stop == 4)) return;
    // ---- 5a. k <= 512 (at most 1024 keys were collected): order by HISTOGRAM RANK.  The output position of a key is
    // the number of keys above it = (keys in higher bins) + (keys of its own bin above it): a 2048-bin histogram over
    // the live key range, a suffix scan over the bins, the keys dropped bin by bin into a second buffer, and a count
    // over the (few) keys that share the bin.  ~6 barriers and one LDS atomic per key; it takes whatever was
    // collected, so no separate "refine to <= 512 keys" stage exists (that stage + a 512-key network cost 3.9 + 13.9 us
    // of the final launch, rank-by-counting over 512 keys 3.9 + 5.7 us; profiles/r02_notes.md).
    if (DAE_EXP_ON(dbg_stop == 5)) return;
    if (sort_n == 1024) {
        constexpr int PER = 1024 / NTH;                          // collected keys per thread (slot e * NTH + tid)
        constexpr int BPT = TK_BINS / NTH;
        const unsigned c = s_cnt < 1024u ? s_cnt : 1024u;
        u64 mine[PER];
        unsigned mbin[PER], mpos[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) mine[e] = (unsigned)(e * NTH + tid) < c ? skey[e * NTH + tid] : 0ull;
        const u64 rlo = lo, rhi = s_max;                         // every collected key lies in [lo, s_max]
        int shift = 64 - 11 - __clzll((rhi - rlo) | 1ull);
        if (shift < 0) shift = 0;
        // Bins are linear in the LOGIT, not in its bit pattern: the order-preserving key spends half of its range on
        // magnitudes below 2^-126, so a row whose winners straddle zero (popular tracks at p ~ 0.5) had all of its keys
        // in ~70 of the 2048 bit-pattern bins, and the same-bin loop below went quadratic (17 of the 40 us of a
        // 1024-row launch).  Any monotone map of the key keeps the result: bins only have to respect the order.
        const float zlo = dae_okey_inv((unsigned)(rlo >> 32)), zhi = dae_okey_inv((unsigned)(rhi >> 32));
        const float zspan = zhi - zlo;
        const bool lin = zspan > 1e-30f && zspan < 3.0e38f;      // finite, non-degenerate range; else the bit pattern
        const float zscale = lin ? (float)(TK_BINS - 1) / zspan : 0.0f;
        for (int b2 = tid; b2 < TK_BINS; b2 += TK_THREADS) hist[b2] = 0;
        __syncthreads();                                         // also: every skey read above is done
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            unsigned bn = 0u;
            if (mine[e] != 0ull) {
                if (lin) {
                    const float zz = dae_okey_inv((unsigned)(mine[e] >> 32));
                    const float fb = fminf(fmaxf((zz - zlo) * zscale, 0.0f), (float)(TK_BINS - 1));
                    bn = (unsigned)fb;
                } else {
                    bn = (unsigned)((mine[e] - rlo) >> shift);
                }
            }
            mbin[e] = bn;
            mpos[e] = mine[e] != 0ull ? atomicAdd(&hist[bn], 1u) : 0u;
        }
        __syncthreads();
        if (DAE_EXP_ON(dbg_stop == 8)) { if (mpos[0] == 12345u) a.out_idx[0] = 1; return; }
        // above[b] = keys in bins > b.  Thread t owns the BPT bins from 2047 - BPT t downwards.
        unsigned* above = reinterpret_cast<unsigned*>(skey);     // 2048 x 4 B = the sort buffer's 1024 x 8 B
        u64* sorted = keys;                                      // key cache region: >= 1024 keys (launch_topk)
        {
            const int top = TK_BINS - 1 - BPT * tid;
            unsigned cb[BPT], own = 0;
#pragma unroll
            for (int e = 0; e < BPT; ++e) { cb[e] = hist[top - e]; own += cb[e]; }
            unsigned v = own;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned o = __shfl_up(v, d);
                if (lane >= d) v += o;
            }
            if (lane == 63) wave_tot[tid >> 6] = v;
            __syncthreads();
            unsigned run = v - own;
            for (int w = 0; w < (tid >> 6); ++w) run += wave_tot[w];
#pragma unroll
            for (int e = 0; e < BPT; ++e) { above[top - e] = run; run += cb[e]; }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < PER; ++e)
            if (mine[e] != 0ull) sorted[above[mbin[e]] + mpos[e]] = mine[e];
        __syncthreads();
        if (DAE_EXP_ON(dbg_stop == 6)) return;
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
        // the winners in rank order through LDS (the `above` table is dead now), then out in rank order: thread i writes
        // position i, a wave 256 contiguous bytes.  Storing from the rank loop put every 4-byte value into a line of its
        // own -- ~1000 partial-line writes per row, the larger half of this launch at 1024 rows.
        if (DAE_EXP_ON(dbg_stop == 7)) { if (rk[0] == 12345u) a.out_idx[0] = 1; return; }
        __syncthreads();
        u64* fin = skey;
#pragma unroll
        for (int e = 0; e < PER; ++e)
            if (rk[e] < k_eff) fin[rk[e]] = mine[e];
        __syncthreads();
        for (unsigned i = tid; i < k_eff; i += NTH) {
            const u64 key = fin[i];
            const float z = dae_okey_inv((unsigned)(key >> 32));
            const int colv = (int)(~(unsigned)(key & 0xFFFFFFFFull));
            const size_t o = (size_t)row * k + i;
            if (a.out_idx) a.out_idx[o] = colv;
            if (a.out_score) a.out_score[o] = a.out_kind == DAE_OUT_SCORE ? dae_sigmoidf(z) : z;
            if (a.out_pairs) a.out_pairs[(size_t)row * a.pairs_stride + i] = make_uint2(__float_as_uint(z), (unsigned)colv);
            if (a.out_tau && i == (unsigned)k - 1) a.out_tau[row] = z;
        }
    } else
    // Hybrid bitonic sort, descending.  Thread t holds elements t, t + NTH, ... (E = sort_n / NTH of them,
    // at least one).  Strides < 64 exchange through wave shuffles, larger ones through LDS -- also when both
    // elements of a pair sit in the same thread (every key is in LDS for that step anyway).
    {
        const int E = sort_n > TK_THREADS ? sort_n / TK_THREADS : 1;
        u64 kr[EMAX];
#pragma unroll
        for (int e = 0; e < EMAX; ++e) kr[e] = (e < E && e * TK_THREADS + tid < sort_n) ? skey[e * TK_THREADS + tid] : 0ull;
        for (int size = 2; size <= sort_n; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                if (stride >= 64) {
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < EMAX; ++e)
                        if (e < E && e * TK_THREADS + tid < sort_n) skey[e * TK_THREADS + tid] = kr[e];
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < EMAX; ++e) {
                        if (e < E) {
                            const int i = e * TK_THREADS + tid;
                            const u64 other = i < sort_n ? skey[i ^ stride] : 0ull;
                            const bool desc = ((i & size) == 0);
                            const bool lower = ((i & stride) == 0);
                            const u64 mx = kr[e] > other ? kr[e] : other;
                            const u64 mn = kr[e] > other ? other : kr[e];
                            kr[e] = (lower == desc) ? mx : mn;
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < EMAX; ++e) {
                        if (e < E) {
                            const int i = e * TK_THREADS + tid;
                            const u64 other = __shfl_xor(kr[e], stride);
                            const bool desc = ((i & size) == 0);
                            const bool lower = ((i & stride) == 0);
                            const u64 mx = kr[e] > other ? kr[e] : other;
                            const u64 mn = kr[e] > other ? other : kr[e];
                            kr[e] = (lower == desc) ? mx : mn;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const unsigned i = (unsigned)(e * TK_THREADS + tid);
            if (e < E && i < k_eff) {
                const u64 mine = kr[e];
                const float z = dae_okey_inv((unsigned)(mine >> 32));
                const int colv = (int)(~(unsigned)(mine & 0xFFFFFFFFull));
                const size_t o = (size_t)row * k + i;
                if (a.out_idx) a.out_idx[o] = colv;
                if (a.out_score) a.out_score[o] = a.out_kind == DAE_OUT_SCORE ? dae_sigmoidf(z) : z;
                if (a.out_pairs) a.out_pairs[(size_t)row * a.pairs_stride + i] = make_uint2(__float_as_uint(z), (unsigned)colv);
                if (a.out_tau && i == (unsigned)k - 1) a.out_tau[row] = z;
            }
        }
    }
    // fewer than k valid elements: pad like the reference's cand[:500] of a short list
    for (unsigned i = k_eff + tid; i < (unsigned)k; i += TK_THREADS) {
        const size_t o = (size_t)row * k + i;
        if (a.out_idx) a.out_idx[o] = -1;
        if (a.out_score) a.out_score[o] = -__builtin_inff();
        if (a.out_pairs) a.out_pairs[(size_t)row * a.pairs_stride + i] = make_uint2(__float_as_uint(-__builtin_inff()), 0xFFFFFFFFu);
    }
    if (a.out_tau && tid == 0 && k_eff < (unsigned)k) a.out_tau[row] = -__builtin_inff();
}

// ---- tau of the fused path from the sample's group maxima, and the sample's survivors ---------------------------
// Phase A leaves, per row, the dense logits of the sample tiles and the maxima of disjoint groups of sample columns
// (one column from each of the popularity bands a workgroup decodes; non-rankable columns masked to -inf).  The
// maxima are distinct elements of the row, at most n_seeds of them seeds, so the (k + n_seeds)-th largest of them is
// <= the row's k-th largest rankable non-seed logit: a valid threshold for phase B.
// One 256-thread workgroup per row.
//   1. the row's dense sample logits are requested first (up to 16 float4 per thread stay in registers: their
//      latency hides under the search);
//   2. tau: 4-ary search on the 16 leading bits of the order-preserving key for the largest prefix P with
//      count(maxima >= P << 16) >= k + n_seeds -- 8 steps, each 3 x 16 compares per thread counted with s_bcnt1 on
//      the compare mask and ONE barrier (histogram passes cost more: the logits of a row share a few exponents, so
//      4 096 LDS atomics land on a handful of addresses -- 12 us; a shuffle reduction per probe 13.7 us).
//      tau = the float of P << 16: at most 2^-7 (relative) below the element of that rank;
//   3. the sample logits >= tau go out as (logit, column) pairs, one flat list per row (slots from a block-wide
//      exclusive scan of the per-thread counts: no atomics) -- group 0 of the final selection.  Phase B never
//      looks at the sample again.
constexpr int TAU_PRE = 16;        // float4 of dense sample logits per thread requested before the search
struct TauP {
    const float* gmax; int64_t ld_g; int n_g;                    // group maxima [B][ld_g], n_g per row
    const float* samp; int64_t ld_s; int n_s;                    // dense sample logits [B][ld_s], n_s per row (% 32 == 0)
    const int* samp_list; int col_lo;                            // column of sample element q = col_lo + list[q >> 5] * 32 + (q & 31)
    const int32_t* seed_row_ptr; int k;
    float* tau; uint2* out_pairs; int64_t pairs_stride; int* out_cnt;
    long long* dbg;
};
// HAS_S = false: no dense sample to scan (n_s == 0: the bf16 / exact filter launch decodes every tile itself) -- tau only,
// a third of the registers, so that the workgroup fits on a CU NEXT to a filter workgroup of another batch.
template <int TAU_PER, bool HAS_S = true>   // TAU_PER: maxima per thread held in registers (rows of <= 256 * TAU_PER maxima in one sweep)
__global__ __launch_bounds__(256) void tau_select_kernel(const TauP p)
{
    __shared__ unsigned wcnt[2][4][3];
    __shared__ unsigned wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = blockIdx.x;
#ifdef DAE_EXPERIMENTS         // stage stamps of rows 0 and 100 (DAE_DBG_TAU=1 prints them every 50 launches)
#define TSTAMP(i) if (p.dbg && tid == 0 && (row == 0 || row == 100)) p.dbg[(row ? 8 : 0) + (i)] = __builtin_readcyclecounter();
#else
#define TSTAMP(i)
#endif
    TSTAMP(0)
    const float* g = p.gmax + (size_t)row * p.ld_g;
    const int n = p.n_g;
    const unsigned ns = p.seed_row_ptr ? (unsigned)(p.seed_row_ptr[row + 1] - p.seed_row_ptr[row]) : 0u;
    const unsigned need = (unsigned)p.k + ns;
    // ---- 1. requests: the group maxima ALONE first -- the search needs them and 256 rows asking for their 62 KB of
    // dense logits at the same moment put every row's maxima behind everybody's dense lines (maxima arrived after
    // 8.3 k cycles).  The dense row (TAU_PRE float4 per thread, kept in registers) is requested once the maxima are
    // here and stays in flight under the search.
    const bool one_sweep = n <= 256 * TAU_PER;
    float graw[TAU_PER];
#pragma unroll
    for (int u = 0; u < TAU_PER; ++u) {
        const int i = u * 256 + tid;
        graw[u] = (one_sweep && i < n) ? g[i] : -__builtin_inff();
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned kreg[TAU_PER];
#pragma unroll
    for (int u = 0; u < TAU_PER; ++u) kreg[u] = dae_okey(graw[u]);     // -inf -> NEG_INF key: never counted
    __builtin_amdgcn_sched_barrier(0);
    const float4* srow = reinterpret_cast<const float4*>(p.samp + (size_t)row * p.ld_s);
    const int n4 = HAS_S ? p.n_s >> 2 : 0;
    constexpr int NPRE = HAS_S ? TAU_PRE : 1;
    float4 zpre[NPRE];
    // tiles of the preloaded part (one id per 8 float4) -> LDS: consumed only where something passes
    __shared__ int ltile[HAS_S ? TAU_PRE * 256 / 8 : 1];
    if (HAS_S) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int f = u * 256 + tid;
            zpre[u] = srow[f < n4 ? f : 0];
        }
        for (int i = tid; i < TAU_PRE * 256 / 8; i += 256) ltile[i] = p.samp_list[i < (n4 + 7) / 8 ? i : 0];
    }
    __builtin_amdgcn_sched_barrier(0);                           // keep every request above ahead of the search
    // ---- 2. tau -------------------------------------------------------------------------------------------------
    // counts of keys >= each of 3 probes over the row (block-uniform results); absent / -inf keys never count
    auto count3 = [&](unsigned q0, unsigned q1, unsigned q2, int it, unsigned (&c)[3]) {
        unsigned a0 = 0, a1 = 0, a2 = 0;
        if (one_sweep) {
            // all compare masks of a probe first, then their popcounts: a VALU compare -> SALU popcount pair stalls
            // the (only) wave of the SIMD for the VALU -> SALU hand-over; interleaved pair by pair a step cost 2 000 cycles
            unsigned long long m[TAU_PER];
#pragma unroll
            for (int u = 0; u < TAU_PER; ++u) m[u] = __ballot(kreg[u] >= q0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < TAU_PER; ++u) a0 += (unsigned)__popcll(m[u]);
#pragma unroll
            for (int u = 0; u < TAU_PER; ++u) m[u] = __ballot(kreg[u] >= q1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < TAU_PER; ++u) a1 += (unsigned)__popcll(m[u]);
#pragma unroll
            for (int u = 0; u < TAU_PER; ++u) m[u] = __ballot(kreg[u] >= q2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < TAU_PER; ++u) a2 += (unsigned)__popcll(m[u]);
        } else {
            for (int i0 = 0; i0 < n; i0 += 256) {
                const unsigned key = i0 + tid < n ? dae_okey(g[i0 + tid]) : 0u;
                a0 += (unsigned)__popcll(__ballot(key >= q0));
                a1 += (unsigned)__popcll(__ballot(key >= q1));
                a2 += (unsigned)__popcll(__ballot(key >= q2));
            }
        }
        if (lane == 0) { wcnt[it & 1][wv][0] = a0; wcnt[it & 1][wv][1] = a1; wcnt[it & 1][wv][2] = a2; }
        __syncthreads();                                         // slots alternate: one barrier per step is enough
#pragma unroll
        for (int e = 0; e < 3; ++e)
            c[e] = wcnt[it & 1][0][e] + wcnt[it & 1][1][e] + wcnt[it & 1][2][e] + wcnt[it & 1][3][e];
    };
    // prefixes below (NEG_INF >> 16) + 1 would count -inf / absent entries.  The search runs over [p_min - 1, 0xFFFF]:
    // p_min - 1 stands for "fewer than `need` maxima exist" (count taken as >= need by convention, never probed --
    // every probe is > lo), so no separate existence pass is needed.
    const unsigned p_min = (DAE_KEY_NEG_INF >> 16) + 1u;
    unsigned lo = p_min - 1u, hi = 0xFFFFu;                      // the answer lies in [lo, hi]
    int it = 0;
    unsigned c[3];
    TSTAMP(1)
    TSTAMP(2)
    while (lo < hi) {                                            // invariant: count(lo) >= need, count(hi + 1) < need
        const unsigned span = hi - lo;                           // >= 1; three probes cut [lo + 1, hi] into four parts
        const unsigned m1 = lo + (span + 3u) / 4u, m2 = lo + (2u * span + 3u) / 4u, m3 = lo + (3u * span + 3u) / 4u;
        count3(m1 << 16, m2 << 16, m3 << 16, it++, c);           // lo < m1 <= m2 <= m3 <= hi
        if (c[2] >= need) lo = m3;
        else if (c[1] >= need) { lo = m2; hi = m3 - 1u; }
        else if (c[0] >= need) { lo = m1; hi = m2 - 1u; }
        else hi = m1 - 1u;
    }
    const bool found = lo >= p_min;
    TSTAMP(3)
    // No dense sample (n_s == 0: the bf16 filter launch decodes the sample tiles AGAIN and every candidate comes from it):
    // the element tau was taken from has to pass a compare in ANOTHER kernel.  Both kernels run the same MFMA sequence on
    // the same operands, so the values agree bit for bit (tests/test_gpu_bf16.py); 4 ulp of slack make the row's k
    // candidates independent of that (a tau sitting exactly on its element, and a last-bit difference, would otherwise
    // leave the row one candidate short: -1 padding instead of an error).
    const unsigned tkey = (p.n_s == 0 && (lo << 16) > DAE_KEY_NEG_INF + 8u) ? (lo << 16) - 4u : (lo << 16);
    const float tv = found ? dae_okey_inv(tkey) : -__builtin_inff();
    if (tid == 0) p.tau[row] = tv;
    if (!HAS_S) {                                                // no sample to scan: the (empty) survivor list
        if (tid == 0) p.out_cnt[row] = 0;
        return;
    }
    // ---- 3. the sample's survivors (never -inf: masked columns and pads are not candidates) ------------------------
    auto passes = [&](float z) { return z >= tv && z > -__builtin_inff(); };
    unsigned mine = 0;
    unsigned live = 0;                                           // bit u: some lane of this wave has a survivor in zpre[u]
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
        const float mx = fmaxf(fmaxf(zpre[u].x, zpre[u].y), fmaxf(zpre[u].z, zpre[u].w));
        if (__ballot(u * 256 + tid < n4 && passes(mx))) {        // wave-uniform: most groups hold nothing above tau
            live |= 1u << u;
            if (u * 256 + tid < n4)
                mine += (passes(zpre[u].x) ? 1u : 0u) + (passes(zpre[u].y) ? 1u : 0u) + (passes(zpre[u].z) ? 1u : 0u) +
                        (passes(zpre[u].w) ? 1u : 0u);
        }
    }
    for (int f = TAU_PRE * 256 + tid; f < n4; f += 256) {       // rows longer than the preloaded part: read again
        const float4 z = srow[f];
        mine += (passes(z.x) ? 1u : 0u) + (passes(z.y) ? 1u : 0u) + (passes(z.z) ? 1u : 0u) + (passes(z.w) ? 1u : 0u);
    }
    TSTAMP(4)
    unsigned incl = mine;                                        // block-wide exclusive scan of the counts
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads();
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned at = incl - mine;
    for (int w = 0; w < wv; ++w) at += wsum[w];
    if (tid == 255) p.out_cnt[row] = (int)(at + mine);
    TSTAMP(5)
    uint2* dst = p.out_pairs + (size_t)row * p.pairs_stride;
    auto emit4 = [&](const float4 z, int f, int tile) {
        const unsigned cb = (unsigned)(p.col_lo + tile * 32 + ((4 * f) & 31));
        if (passes(z.x)) dst[at++] = make_uint2(__float_as_uint(z.x), cb);
        if (passes(z.y)) dst[at++] = make_uint2(__float_as_uint(z.y), cb + 1u);
        if (passes(z.z)) dst[at++] = make_uint2(__float_as_uint(z.z), cb + 2u);
        if (passes(z.w)) dst[at++] = make_uint2(__float_as_uint(z.w), cb + 3u);
    };
#pragma unroll
    for (int u = 0; u < NPRE; ++u)
        if (((live >> u) & 1u) && u * 256 + tid < n4) emit4(zpre[u], u * 256 + tid, ltile[(u * 256 + tid) >> 3]);
    for (int f = TAU_PRE * 256 + tid; f < n4; f += 256) emit4(srow[f], f, p.samp_list[f >> 3]);
    TSTAMP(6)
#ifdef DAE_EXPERIMENTS
    if (p.dbg && tid == 0 && row == 0) p.dbg[7] = it;
#endif
}

// The same for launches of MANY SHORT rows (vocabulary shards, where every rank scores the whole global batch over its
// slice of the columns: 2048 rows x 1952 sample logits at 8 ranks): one WAVE per row, four rows per workgroup.  The
// maxima (<= 64 * KPL) and the dense sample (<= 64 * PRE float4) of a row sit in the registers of its wave; a search
// step is 3 x KPL compare masks and their s_bcnt1 -- the counts are wave-uniform scalars, so there is no LDS, no
// barrier and no other wave to wait for; the survivors' slots come from one wave scan.  Same tau, same survivor set
// (list order differs: the final selection does not depend on it).
template <int KPL, int PRE>
__global__ __launch_bounds__(256) void tau_select_wave_kernel(const TauP p, const int B)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (row >= B) return;                                        // wave-uniform
    const float* g = p.gmax + (size_t)row * p.ld_g;
    const int n = p.n_g;
    const unsigned ns = p.seed_row_ptr ? (unsigned)(p.seed_row_ptr[row + 1] - p.seed_row_ptr[row]) : 0u;
    const unsigned need = (unsigned)p.k + ns;
    float graw[KPL];
#pragma unroll
    for (int u = 0; u < KPL; ++u) {
        const int i = u * 64 + lane;
        graw[u] = i < n ? g[i] : -__builtin_inff();
    }
    const float4* srow = reinterpret_cast<const float4*>(p.samp + (size_t)row * p.ld_s);
    const int n4 = p.n_s >> 2;
    float4 zpre[PRE];
    int tl[PRE];
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int f = u * 64 + lane;
        zpre[u] = srow[f < n4 ? f : 0];
        tl[u] = p.samp_list[(f < n4 ? f : 0) >> 3];
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned kreg[KPL];
#pragma unroll
    for (int u = 0; u < KPL; ++u) kreg[u] = dae_okey(graw[u]);
    auto count3 = [&](unsigned q0, unsigned q1, unsigned q2, unsigned (&c)[3]) {
        unsigned long long m[KPL];
        unsigned a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int u = 0; u < KPL; ++u) m[u] = __ballot(kreg[u] >= q0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KPL; ++u) a0 += (unsigned)__popcll(m[u]);
#pragma unroll
        for (int u = 0; u < KPL; ++u) m[u] = __ballot(kreg[u] >= q1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KPL; ++u) a1 += (unsigned)__popcll(m[u]);
#pragma unroll
        for (int u = 0; u < KPL; ++u) m[u] = __ballot(kreg[u] >= q2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KPL; ++u) a2 += (unsigned)__popcll(m[u]);
        c[0] = a0; c[1] = a1; c[2] = a2;
    };
    const unsigned p_min = (DAE_KEY_NEG_INF >> 16) + 1u;         // see tau_select_kernel: [p_min - 1, 0xFFFF], sentinel lo
    unsigned lo = p_min - 1u, hi = 0xFFFFu;
    unsigned c[3];
    while (lo < hi) {
        const unsigned span = hi - lo;
        const unsigned m1 = lo + (span + 3u) / 4u, m2 = lo + (2u * span + 3u) / 4u, m3 = lo + (3u * span + 3u) / 4u;
        count3(m1 << 16, m2 << 16, m3 << 16, c);
        if (c[2] >= need) lo = m3;
        else if (c[1] >= need) { lo = m2; hi = m3 - 1u; }
        else if (c[0] >= need) { lo = m1; hi = m2 - 1u; }
        else hi = m1 - 1u;
    }
    const unsigned tkey = (p.n_s == 0 && (lo << 16) > DAE_KEY_NEG_INF + 8u) ? (lo << 16) - 4u : (lo << 16);   // see tau_select_kernel
    const float tv = lo >= p_min ? dae_okey_inv(tkey) : -__builtin_inff();
    if (lane == 0) p.tau[row] = tv;
    auto passes = [&](float z) { return z >= tv && z > -__builtin_inff(); };
    unsigned mine = 0;
#pragma unroll
    for (int u = 0; u < PRE; ++u)
        if (u * 64 + lane < n4)
            mine += (passes(zpre[u].x) ? 1u : 0u) + (passes(zpre[u].y) ? 1u : 0u) + (passes(zpre[u].z) ? 1u : 0u) +
                    (passes(zpre[u].w) ? 1u : 0u);
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    unsigned at = incl - mine;
    if (lane == 63) p.out_cnt[row] = (int)incl;
    uint2* dst = p.out_pairs + (size_t)row * p.pairs_stride;
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int f = u * 64 + lane;
        if (f < n4) {
            const unsigned cb = (unsigned)(p.col_lo + tl[u] * 32 + ((4 * f) & 31));
            const float4 z = zpre[u];
            if (passes(z.x)) dst[at++] = make_uint2(__float_as_uint(z.x), cb);
            if (passes(z.y)) dst[at++] = make_uint2(__float_as_uint(z.y), cb + 1u);
            if (passes(z.z)) dst[at++] = make_uint2(__float_as_uint(z.z), cb + 2u);
            if (passes(z.w)) dst[at++] = make_uint2(__float_as_uint(z.w), cb + 3u);
        }
    }
}

int launch_tau_select(dae_ctx* ctx, const TauP& p, int B)
{
    if (B <= 0) return DAE_OK;
    TauP q = p;
#ifdef DAE_EXPERIMENTS
    static const bool dbg = dae_exp_env("DAE_DBG_TAU") != nullptr;
    static long long* dbuf = nullptr;
    static int calls = 0;
    if (dbg) { if (!dbuf) (void)hipMalloc(&dbuf, 16 * 8); q.dbg = dbuf; }
#endif
    // many short rows (>= 4 per CU, <= 2048 maxima and sample logits each: the 8-rank shard of the global batch): a wave
    // per row, everything in registers, no barriers -- 30.1 -> 22.3 us for 2048 rows.  Longer rows lose (4096 maxima per
    // wave = 64 key registers and 128 mask SGPRs per probe: 37 vs 18.6 us at 4 ranks) and keep the workgroup kernel.
    static const bool no_wave = dae_exp_env("DAE_TAU_NO_WAVE") != nullptr;            // A/B against the workgroup-per-row kernel
    if (B >= 1024 && p.n_g <= 64 * 32 && (p.n_s >> 2) <= 64 * 8 && !no_wave) {
        hipLaunchKernelGGL((tau_select_wave_kernel<32, 8>), dim3((B + 3) / 4), dim3(256), 0, ctx->stream, q, B);
        DAE_CHECK_LAUNCH(ctx, "tau_select_wave_kernel");
        return DAE_OK;
    }
    if (p.n_g <= 256 * 16 && p.n_s == 0)
        hipLaunchKernelGGL((tau_select_kernel<16, false>), dim3(B), dim3(256), 0, ctx->stream, q);
    else if (p.n_g <= 256 * 32 && p.n_s == 0)
        hipLaunchKernelGGL((tau_select_kernel<32, false>), dim3(B), dim3(256), 0, ctx->stream, q);
    else if (p.n_g <= 256 * 16)
        hipLaunchKernelGGL(tau_select_kernel<16>, dim3(B), dim3(256), 0, ctx->stream, q);
    else if (p.n_g <= 256 * 32)        // batch 1024 on one GPU: 5 120 maxima per row -- half the compares of the 64-key shape
        hipLaunchKernelGGL(tau_select_kernel<32>, dim3(B), dim3(256), 0, ctx->stream, q);
    else
        hipLaunchKernelGGL(tau_select_kernel<64>, dim3(B), dim3(256), 0, ctx->stream, q);
    DAE_CHECK_LAUNCH(ctx, "tau_select_kernel");
#ifdef DAE_EXPERIMENTS
    if (dbg && (++calls % 50) == 0) {
        long long h[16];
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "TAU row0:");
        for (int i = 1; i < 7; ++i) fprintf(stderr, " %lld", h[i] - h[0]);
        fprintf(stderr, " steps %lld | row100:", h[7]);
        for (int i = 1; i < 7; ++i) fprintf(stderr, " %lld", h[8 + i] - h[8]);
        fprintf(stderr, "\n");
    }
#endif
    return DAE_OK;
}

template <typename Src>
int launch_topk(dae_ctx* ctx, const Src& src, const dae_topk_args& a)
{
    if (a.k < 1 || a.k > DAE_MAX_K)
        return dae_fail(ctx, DAE_ERR_ARG, "k=%d out of [1,%d]", a.k, DAE_MAX_K);
    if (a.B <= 0) return DAE_OK;
    int sort_n = 1024;
    if (a.k > 512) sort_n = TK_SORT_MAX;
    dae_topk_args aa = a;
    aa.sort_cap = sort_n;
    // Two LDS modes, both covered by the parity tests:
    //   bitmap (default): per-row seed bitmap + a key cache filling the CU's LDS -- fastest alone.
    //   lean (DAE_TOPK_LEAN=1): no bitmap (seed-blind narrowing at rank k + n_seeds, seeds removed once
    //     from the collected keys), sort buffer + small cache only: <= 30 KiB and <= 56 registers per
    //     thread, so a workgroup fits on a CU NEXT to a decode workgroup of another batch.  Measured
    //     (profiles/r01_notes.md): the co-resident decode launch slows down by about what the overlap
    //     gains, and the kernel alone is slower (re-reads its source), so it is not the default.
    static const bool want_lean = dae_exp_env("DAE_TOPK_LEAN") != nullptr;
    aa.lean = want_lean ? 1 : 0;
    // a ranked range too wide for the LDS bitmap (> ~1 M columns) takes the bitmap-free mode instead of failing
    if ((((size_t)((a.bitmap_n + 31) / 32) * 4) + 15) + (size_t)sort_n * 8 + 22 * 1024 > (size_t)160 * 1024) aa.lean = 1;
    const size_t lds_total = 160 * 1024, lds_static = 14 * 1024;    // hist 8K + seg_prefix 4K + scalars
    size_t dyn;
    int key_cap;
    if (aa.lean) {
        const size_t budget = 30 * 1024 - (sizeof(unsigned) * TK_BINS + (Src::kSegs ? (Src::kSegs + 2) * 4 : 8) + 256);
        const size_t sk = (size_t)sort_n * 8;
        key_cap = budget > sk ? (int)((budget - sk) / 8) : 0;
        dyn = sk + (size_t)key_cap * 8;
    } else {
        const size_t bm_bytes = (((size_t)((a.bitmap_n + 31) / 32) * 4) + 15) & ~(size_t)15;
        if (bm_bytes + (size_t)sort_n * 8 + lds_static + 8 * 1024 > lds_total)
            return dae_fail(ctx, DAE_ERR_ARG, "ranked column range %d too wide for the LDS seed bitmap",
                            a.bitmap_n);
        // bitmap + sort buffer + key cache.  The cache is sized to what a row can hold, not to the CU:
        // with many rows of few keys (large batches, vocabulary shards) two workgroups then share a CU.
        const size_t room = lds_total - lds_static - bm_bytes - (size_t)sort_n * 8;
        size_t want = (size_t)(src.max_keys() > 0 ? src.max_keys() : 8192) * 8;
        if (want < 1024 * 8) want = 1024 * 8;                   // the ordering stage's second buffer (1024 keys)
        if (want > room) want = room;
        key_cap = (int)(want / 8);
        dyn = bm_bytes + (size_t)sort_n * 8 + want;
    }
    static const char attr_set_key = 0;
    if (dae_first_use(ctx, &attr_set_key)) {
        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_kernel<Src, 1024>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)(lds_total - lds_static)));
        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_kernel<Src, 256>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)(lds_total - lds_static)));

    }
    // debug: DAE_TOPK_STOP=n stops the phase-A (tau-producing) kernel after stage n,
    // DAE_TOPK_STOP=-n the other launches (bisecting stage costs under rocprofv3)
    static const int dbg_env = dae_exp_env("DAE_TOPK_STOP") ? atoi(dae_exp_env("DAE_TOPK_STOP")) : 0;
    const int dbg_stop = dbg_env > 0 ? (a.out_tau ? dbg_env : 0) : (a.out_tau ? 0 : -dbg_env);
    // threads per row: 256 for rows known to be short (see topk_kernel): dense rows of <= 4096 columns (the
    // sample of a small vocabulary shard) and gathered shard lists.  Candidate lists (PairSrc) keep 1024: their
    // cost is walking the per-workgroup segments (one wave per segment), which 4 waves do slower than 16
    // (measured: 407 vs 354 us per step at --sim-world 8).  DAE_TOPK_THREADS=1024|256 forces one shape (A/B).
    static const int nth_env = dae_exp_env("DAE_TOPK_THREADS") ? atoi(dae_exp_env("DAE_TOPK_THREADS")) : 0;
    const int bound = Src::kSegs ? 0 : src.max_keys();
    // ... and candidate lists when the launch has many rows (>= 4 per CU: large batches, vocabulary shards): a row then
    // holds few candidates, four 256-thread workgroups share a CU, and the per-row fixed cost is what counts
    // (--sim-world 8, 2048 rows: step 322 -> 284 us; at 256 rows 256 threads lose: 22 vs 14 us)
    const bool small = nth_env ? nth_env == 256 : ((!Src::kSegs && bound <= 4096) || (Src::kSegs && (a.B >= 1024 || a.prefer_small)));
    if (small)
        hipLaunchKernelGGL((topk_kernel<Src, 256>), dim3(a.B), dim3(256), dyn, ctx->stream, src, aa, key_cap, dbg_stop);
    else
        hipLaunchKernelGGL((topk_kernel<Src, 1024>), dim3(a.B), dim3(1024), dyn, ctx->stream, src, aa, key_cap, dbg_stop);
    DAE_CHECK_LAUNCH(ctx, "topk_kernel");
    return DAE_OK;
}

}  // namespace

int dae_launch_tau_select(dae_ctx* ctx, const float* gmax, int64_t ld_g, int n_g, const float* samp, int64_t ld_s,
                          int n_s, const int* samp_list, int col_lo, int B, int k, const int32_t* seed_row_ptr,
                          float* tau, uint2* out_pairs, int64_t pairs_stride, int* out_cnt)
{
    if ((n_s & 31) || (ld_s & 3) || (reinterpret_cast<uintptr_t>(samp) & 15))
        return dae_fail(ctx, DAE_ERR_ARG, "sample rows must be whole tiles, 16-byte aligned");
    TauP p{gmax, ld_g, n_g, samp, ld_s, n_s, samp_list, col_lo, seed_row_ptr, k, tau, out_pairs, pairs_stride, out_cnt, nullptr};
    return launch_tau_select(ctx, p, B);
}

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
