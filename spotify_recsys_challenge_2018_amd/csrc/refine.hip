// refine.hip -- DAE_DTYPE_BF16_EXACT: between the bf16 filter launch and the final selection, turn the candidates'
// stored values into what the fp32 path would have computed -- or into "absent".
//
// The filter launch (decode_f32.hip, bias b + eps) leaves per (decode workgroup, row) lists of (u, column) where
// u is an UPPER bound of the column's fp32 logit z32 with z32 >= u - 2 eps_c (api.hip decode_topk_core; the bound is
// derived next to exact_bounds_kernel).  One 512-thread workgroup per row:
//   1. narrow: with need = k + n_seeds, the need-th largest u (to 20 key bits) minus 2 eps_max is a threshold tau'
//      that `need` distinct columns provably reach in fp32, so a candidate with u < tau' cannot be among the k best
//      non-seeds: it is overwritten with -inf (absent for the selection kernel).  What is left is ~k columns plus the
//      ones within 2 eps of the cut -- not the thousands a loose phase-A threshold lets through when the bias says
//      nothing (--bias zeros: ~6 700 candidates per row, ~900 after this step).  Skipped when it cannot pay
//      (fewer than 1.25 need candidates);
//   2. recompute: the survivors' logits with the canonical chain acc = fmaf(h[k], W[c][k], acc) over k = 0 .. H-1
//      from +0, then + b[c] -- oracle/dae_oracle.c orc_decode, the operation v_mfma_f32_32x32x2_f32 performs in the
//      fp32 kernels (main_challenge.py:26-36 ranks those values) -- written back IN PLACE.
//      A lane owns a candidate and runs its chain; the decoder rows (1 KiB each, row-major fp32 copy of the image)
//      are fetched so that the 4 lanes of a quad read 64 CONTIGUOUS bytes of one row per instruction -- a lane reading
//      its own row 16 bytes at a time made every load instruction touch 64 lines and the address unit, not the
//      memory, was the limit (14 us per launch) -- and re-distributed through 5 KiB of LDS per wave (80-byte row
//      stride: conflict-free b128 accesses both ways); 8 blocks of 16 k (32 x 16 bytes per lane) are in flight.
// The selection kernel (topk.hip, PairSrc) then ranks the lists as it does for the other modes.
#include "dae_internal.h"
#include "rank_lds.h"

namespace {

constexpr int RF_STAGE = 16384;        // candidates of a row whose u fit the staging area (floats; the waves' buffers reuse it)
constexpr int RF_SURV = 4096;          // survivors listed per flush (offsets only)
constexpr int RF_BINS = 2048;          // histogram of the narrowing step (lives in the survivor list's LDS)
constexpr int RF_UNROLL = 4;           // flat passes: elements per thread and round, their loads in flight together
constexpr int RF_MAX_SEG = 1024;
constexpr int RF_ROWSTRIDE = 20;       // dwords per candidate in a wave's transposition buffer (16 data + 4 pad)
// Two shapes: 512 threads, 8 blocks of 16 k in flight per lane -- fastest alone (~190 registers, two waves per SIMD) --
// and 256 threads, 2 blocks -- 1 wave per SIMD at <= 112 registers, which fits on a CU NEXT to the two filter waves per SIMD of
// another batch's decode launch (dae_set_overlap_hint): slower alone, but it then runs under that launch instead of
// waiting for it.

struct RefineP {
    uint2* base; int* cnt; int64_t seg_stride, row_stride, cnt_seg_stride; int nseg;
    dae_exact_src x;
    const int32_t* seed_row_ptr; int k;
    uint2* out; int* out_cnt; int out_cap;        // compact output lists [row][out_cap] + counts (see the header comment)
    int* stat;                                    // nullable: [B][2] {candidates, recomputed} of this launch
    int stage_cap;                                // candidates of a row the staging area holds (the launch's dynamic LDS)
    // FUSED SELECTION (round 5; k <= 512): the launch ends with the row's final list -- seeds removed, the k best in order
    // (main_challenge.py:26-36) -- instead of handing its survivors to a selection launch that reads them back
    int fuse;
    dae_rank_out fo;
    const int32_t* seed_col; int bitmap_base, bitmap_n;     // the row's seeds -> an LDS bitmap over the ranked columns
    const float* row_min;                         // nullable: an exchanged threshold (dae_score_topk_finish)
    int bm_off;                                   // byte offset of the seed bitmap in the dynamic LDS (behind the staging area)
    long long* stamps;                            // experiments build: stage stamps of workgroup 0 (DAE_DBG_R)
    // SHARED RECOMPUTATION (round 5, launches of many rows): exact_rescore_shared_kernel ran before this launch and left the
    // fp32 logit in place of the bound u for every candidate of the rows that are recomputed WITHOUT narrowing (`staged` false
    // below: the same predicate there) -- such rows only read their pairs back and order them
    int pre;
    int B;
};

#ifdef DAE_EXPERIMENTS
#define RSTAMP(i) if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0) p.stamps[i] = __builtin_readcyclecounter();
#else
#define RSTAMP(i)
#endif

// HT > 0: hidden = 16 HT known at compile time.  A trip of the recomputation's block loop then has no branch around a load, and
// hipcc counts the waits inside it: with the run-time bounds EVERY wait of the loop was vmcnt(0) -- a block waited for the ring's
// refill it had just requested, a memory round trip per pair of blocks whatever the ring's depth (ISA of round 5; the same finding
// as in exact_rescore_shared_kernel).  The trip loop itself stays rolled: fully unrolled, hipcc hoists the chain's loads and spills
// hundreds of registers (128 are all a 1 024-thread workgroup has).
template <int RF_THREADS, int RF_DEPTH, int HT = 0>
__device__ __forceinline__ void refine_body(const RefineP& p)
{
    constexpr int RF_WAVES = RF_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char rf_dyn[];      // staging floats, then the waves' buffers
    __shared__ int seg_prefix[RF_MAX_SEG + 2];
    __shared__ float hrow[1024];
    // flat indices of the listed candidates (before that: the narrowing's histogram; with the fused selection the upper half
    // holds the row's final keys -- a row that takes that path lists at most 1024 -- and the lower half the ordering's histogram)
    __shared__ __attribute__((aligned(16))) int surv_off[RF_SURV];
    __shared__ unsigned cnts[32];
    __shared__ int s_n;
    __shared__ unsigned f_range[2];              // fused selection: {min, max} of the high words of the keys written to LDS

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const int nseg = p.nseg;
    RSTAMP(0)
    const bool bad = p.x.row_bad && p.x.row_bad[row] != 0;          // precondition of the bound violated: nothing survives
    for (int s = tid; s < nseg; s += RF_THREADS) seg_prefix[s + 1] = p.cnt[(size_t)s * p.cnt_seg_stride + row];
    if (tid == 0) { seg_prefix[0] = 0; s_n = 0; f_range[0] = 0xFFFFFFFFu; f_range[1] = 0u; }
    if (tid < 32) cnts[tid] = (tid == 1 || tid == 3) ? 0xFFFFFFFFu : 0u;      // [1], [3]: minima
    for (int i = tid; i < 1024; i += RF_THREADS) hrow[i] = i < p.x.H ? p.x.h[(size_t)row * p.x.ld_h + i] : 0.0f;
    // fused selection: the row's seeds as a bitmap over the ranked columns, built HERE -- its two dependent loads (seed_row_ptr,
    // seed_col) travel with the prologue's instead of standing in front of the ordering at the end
    unsigned* const bitmap = reinterpret_cast<unsigned*>(rf_dyn + p.bm_off);
    const int bm_words = p.fuse ? (p.bitmap_n + 31) >> 5 : 0;
    const int seed_b = (p.fuse && p.seed_col) ? p.seed_row_ptr[row] : 0;
    const int seed_e = (p.fuse && p.seed_col) ? p.seed_row_ptr[row + 1] : 0;
    int my_seed = seed_b + tid < seed_e ? p.seed_col[seed_b + tid] : -1;
    for (int w = tid; w < bm_words; w += RF_THREADS) bitmap[w] = 0u;
    __syncthreads();
    for (int i = seed_b + tid; i < seed_e; i += RF_THREADS) {
        const int pc = (i == seed_b + tid ? my_seed : p.seed_col[i]) - p.bitmap_base;
        if (pc >= 0 && pc < p.bitmap_n) atomicOr(&bitmap[pc >> 5], 1u << (pc & 31));
    }
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
    RSTAMP(1)                                                    // counts + hidden row in LDS, prefix done
    const int total = seg_prefix[nseg];
    if (total == 0) {
        if (tid == 0 && p.out_cnt) p.out_cnt[row] = 0;
        if (tid == 0 && p.stat) { p.stat[2 * row] = 0; p.stat[2 * row + 1] = 0; }
        if (p.fuse) dae_rank_pad<RF_THREADS>(tid, row, 0u, p.fo);
        return;
    }
    const int need = p.k + (p.seed_row_ptr ? p.seed_row_ptr[row + 1] - p.seed_row_ptr[row] : 0);

    // flat candidate index -> offset of its pair in the lists.  Every pass below walks the FLAT index space, so the loads
    // of a pass are independent of each other (a wave walking its segments one after another waited a memory round trip
    // per segment: 16 trips per pass at 128 segments, 28 of the launch's 45 us on a model whose rows rank differently)
    auto offset_of = [&](int e) -> int {
        int lo = 0, hi = nseg;                              // largest s with seg_prefix[s] <= e
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (seg_prefix[mid] <= e) lo = mid; else hi = mid;
        }
        return (int)((int64_t)lo * p.seg_stride + (int64_t)row * p.row_stride + (e - seg_prefix[lo]));
    };

    // ---- 1. narrow ------------------------------------------------------------------------------------------------
    unsigned* skey = reinterpret_cast<unsigned*>(rf_dyn);         // staged bounds as order-preserving keys
    float taup = bad ? __builtin_inff() : -__builtin_inff();
    unsigned ktau = 0u;                                          // its key (narrowed rows)
    const bool staged = !bad && total <= p.stage_cap && total > need + (need >> 1);   // narrowing pays when it can drop a third
    int n_kept = bad ? 0 : total;
    if (staged) {
        // the candidates' bounds as order-preserving KEYS, staged by segment: wave w takes segments w, w + RF_WAVES, ...,
        // eight of them in flight at a time (a lane per entry; entries beyond 64 of a long segment follow serially) -- no
        // search for the segment of a flat index (7 dependent LDS reads per element: 22 k cycles for 7 400 candidates)
        unsigned kmx = 0u, kmn = 0xFFFFFFFFu;
        for (int sg0 = wave; sg0 < nseg; sg0 += 8 * RF_WAVES) {
            unsigned uv[8]; int b0[8], cn[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int sg = sg0 + q * RF_WAVES;
                const bool ok = sg < nseg;
                b0[q] = seg_prefix[ok ? sg : 0];
                cn[q] = ok ? seg_prefix[sg + 1] - b0[q] : 0;
                const uint2* sp = p.base + ((int64_t)(ok ? sg : 0) * p.seg_stride + (int64_t)row * p.row_stride);
                uv[q] = lane < cn[q] ? sp[lane].x : 0u;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (lane < cn[q]) {
                    const unsigned key = dae_okey(__uint_as_float(uv[q]));
                    skey[b0[q] + lane] = key;
                    kmx = key > kmx ? key : kmx; kmn = key < kmn ? key : kmn;
                }
                if (cn[q] > 64) {                                 // wave-uniform
                    const uint2* sp = p.base + ((int64_t)(sg0 + q * RF_WAVES) * p.seg_stride + (int64_t)row * p.row_stride);
                    for (int i = 64 + lane; i < cn[q]; i += 64) {
                        const unsigned key = dae_okey(__uint_as_float(sp[i].x));
                        skey[b0[q] + i] = key;
                        kmx = key > kmx ? key : kmx; kmn = key < kmn ? key : kmn;
                    }
                }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned a = __shfl_xor(kmx, d), b = __shfl_xor(kmn, d);
            kmx = a > kmx ? a : kmx; kmn = b < kmn ? b : kmn;
        }
        if (lane == 0) { atomicMax(&cnts[0], kmx); atomicMin(&cnts[1], kmn); }
        unsigned* hist = reinterpret_cast<unsigned*>(surv_off);
        for (int i = tid; i < RF_BINS; i += RF_THREADS) hist[i] = 0u;
        __syncthreads();
        RSTAMP(2)                                                // bounds staged
        // P = a staged key with count(key >= P) >= need, as large as a 2048-bin histogram over [min, max] of the row's keys
        // resolves (bins linear in the KEY: any monotone map keeps the argument): one pass of LDS atomics, one scan from
        // the top for the bin B holding the need-th largest key, one pass for the smallest key of that bin.  Everything in
        // B survives, i.e. at most (bin population - 1) candidates more than an exact selection would keep -- a handful
        // against the hundreds inside the 2 eps band.  (The ten 2-bit search steps of round 3 cost 33 k cycles on 7 400
        // keys, a wave reduction and a barrier each.)
        const unsigned kmin = cnts[1], kmax = cnts[0];
        const float scale = 2047.999f / ((float)(kmax - kmin) + 1.0f);
        auto bin_of = [&](unsigned key) -> int {
            const unsigned bq = (unsigned)((float)(key - kmin) * scale);
            return (int)(bq < (unsigned)(RF_BINS - 1) ? bq : (unsigned)(RF_BINS - 1));
        };
#pragma unroll 4
        for (int i = tid; i < total; i += RF_THREADS) atomicAdd(&hist[bin_of(skey[i])], 1u);
        __syncthreads();
        constexpr int BPT = RF_BINS / RF_THREADS;
        const int top = RF_BINS - 1 - BPT * tid;
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
        if (excl < (unsigned)need && (unsigned)need <= incl) {
            unsigned run = excl;
            bool done = false;
#pragma unroll
            for (int e = 0; e < BPT; ++e) {
                if (!done && run + hc[e] >= (unsigned)need) { cnts[2] = (unsigned)(top - e); done = true; }
                run += hc[e];
            }
        }
        __syncthreads();
        const int Bsel = (int)cnts[2];
        unsigned kb = 0xFFFFFFFFu;
#pragma unroll 4
        for (int i = tid; i < total; i += RF_THREADS) {
            const unsigned key = skey[i];
            if (bin_of(key) == Bsel) kb = key < kb ? key : kb;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const unsigned o = __shfl_xor(kb, d); kb = o < kb ? o : kb; }
        if (lane == 0 && kb != 0xFFFFFFFFu) atomicMin(&cnts[3], kb);
        __syncthreads();
        const unsigned P = cnts[3];
        if (P != 0xFFFFFFFFu && P > DAE_KEY_NEG_INF + 2u) {
            taup = dae_okey_inv(P) - 2.0f * p.x.eps_max[0] * 1.000001f;
            taup = dae_okey_inv(dae_okey(taup) - 2u);          // two floats further down: the subtraction rounded
        }
        RSTAMP(3)                                                // selection done
        // an upper bound of how many pass (whole bins down to tau''s): decides where the results go
        ktau = dae_okey(taup);
        const int Bt = ktau > kmin ? bin_of(ktau < kmax ? ktau : kmax) : 0;
        if (top >= Bt && Bt > top - BPT) {                        // the thread that owns bin Bt
            unsigned run = excl;
#pragma unroll
            for (int e = 0; e < BPT; ++e) { run += hc[e]; if (top - e == Bt) cnts[4] = run; }
        }
        __syncthreads();
        n_kept = (int)cnts[4];
        RSTAMP(4)
    }
    // COMPACT: the survivors' (fp32 logit, column) pairs go to this row's own list p.out[row][0 .. n_kept) and the
    // filter launch's per-workgroup lists of the row are emptied (their counts zeroed): the selection kernel then reads one
    // short list per row instead of walking 128 segments that hold mostly "absent" marks.  Rows with more survivors
    // than the list holds (thousands of logits within 2 eps of the cut) are refined IN PLACE, as in round 3.
    const bool compact = p.out != nullptr && n_kept <= p.out_cap;
    uint2* const orow = compact ? p.out + (size_t)row * p.out_cap : nullptr;
    // fused selection, the common case: everything that is recomputed fits the ordering stage -- the survivors' keys stay
    // in LDS and never see global memory.  (Other rows: their lists as before, then a narrowing over them at the end.)
    const bool fast = p.fuse && compact && n_kept <= DAE_RANK_MAX;
    dae_u64* const fkey = reinterpret_cast<dae_u64*>(surv_off + RF_SURV / 2);
    const float row_min = p.row_min ? p.row_min[row] : -__builtin_inff();
    // composite key of a recomputed survivor; 0 = absent (a seed, below an exchanged threshold, -inf)
    auto fkey_of = [&](float z, int colv) -> dae_u64 {
        const unsigned key = dae_okey(z);
        if (colv < 0 || z < row_min || key <= DAE_KEY_NEG_INF) return 0ull;
        const int pc = colv - p.bitmap_base;
        if (pc >= 0 && pc < p.bitmap_n && ((bitmap[pc >> 5] >> (pc & 31)) & 1u)) return 0ull;
        return ((dae_u64)key << 32) | (dae_u64)(~(unsigned)colv);
    };
    // ... into slot i of the LDS list (lanes with `in`), with the range of the keys' high words kept up to date for the ordering
    // stage: one wave reduction on the DPP path and one LDS atomic each way per group (called by WHOLE waves)
    auto fkey_put = [&](int i, float z, int colv, bool in) {
        const dae_u64 ck = in ? fkey_of(z, colv) : 0ull;
        if (in) fkey[i] = ck;
        const unsigned hi = (unsigned)(ck >> 32);
        const unsigned mx = dae_wave_max_u32(hi), mn = dae_wave_min_u32(ck != 0ull ? hi : 0xFFFFFFFFu);
        if (lane == 0 && mx != 0u) { atomicMin(&f_range[0], mn); atomicMax(&f_range[1], mx); }
    };

    // ---- 2. recompute the survivors -------------------------------------------------------------------------------
    float* tbuf = reinterpret_cast<float*>(rf_dyn) + wave * (64 * RF_ROWSTRIDE);      // (shares the staging area: see the barriers)
    const int H16 = HT > 0 ? HT : (p.x.H >> 4);                  // blocks of 16 k (H % 16 remainder handled below)
    const int Hrem4 = HT > 0 ? 0 : ((p.x.H & 15) >> 2);          // float4 left over after the whole blocks

    // one group of <= 64 candidates, a lane each: its column `colv` (any valid column for lanes without one: `in` false)
    // -> the fp32 logit.  The rows are fetched quad-wise (see the header), the lanes' columns travel by shuffle.
    auto rescore_group = [&](int colv, bool in) -> float {
        const int Q = lane >> 2, q = lane & 3;
        const float4* rp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = __shfl(colv, 4 * Q + i);
            rp[i] = reinterpret_cast<const float4*>(p.x.W32 + (size_t)(ci - p.x.col_lo) * p.x.H) + q;
        }
        float4 v[RF_DEPTH][4];
#pragma unroll
        for (int d = 0; d < RF_DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[d][i] = d < H16 ? rp[i][4 * d] : make_float4(0.f, 0.f, 0.f, 0.f);
        float acc = 0.0f;
        // hidden activations: the same for every lane.  Lane l holds h[64 c + l] of the current 64-k chunk c in ONE register and
        // every product takes its factor by v_readlane (an SGPR operand of the v_fma) -- as four broadcast ds_read_b128 per
        // 16 k they were a third of the LDS instructions of this loop, which is LDS-bound (stage stamps, DAE_DBG_R)
        constexpr int RF_U = RF_DEPTH < 4 ? 4 : RF_DEPTH;         // blocks per trip: whole 64-k chunks, so a block's lanes are constants
        auto trip = [&](const int j0) {
            float hq[RF_U / 4];
#pragma unroll
            for (int c = 0; c < RF_U / 4; ++c) hq[c] = hrow[(16 * j0 + 64 * c + lane) & 1023];      // (zero beyond H)
#pragma unroll
            for (int dd = 0; dd < RF_U; ++dd) {
                const int j = j0 + dd;
                constexpr int dmask = RF_DEPTH - 1;
                const int d = dd & dmask;                         // ring slot (RF_DEPTH is a power of two)
                if (HT > 0 || j < H16) {                          // wave-uniform (HT: whole trips, 16 HT % RF_U == 0)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<float4*>(tbuf + (4 * Q + i) * RF_ROWSTRIDE + 4 * q) = v[d][i];
                    __builtin_amdgcn_wave_barrier();              // (a wave's LDS accesses execute in order; keep the compiler from moving them)
                    // The ring is refilled in PAIRS of blocks: a quad reads 64 contiguous bytes of its row per block, half a
                    // 128-byte line.  Requested one block apart, the two halves of a line were two fetches from L2 -- the ~2 000
                    // lines a CU has in flight do not survive in its 256-line L1 (stage stamps: 18 k cycles per group of 64
                    // against 8.5 k for its bytes at 64 B per cycle).  Requested back to back, the second finds the line pending.
                    if (dd & 1) {
                        const int dp = (dd - 1) & dmask;
                        const int jn0 = j - 1 + RF_DEPTH, jn1 = j + RF_DEPTH;
                        if (HT > 0) {
                            // no branch around a load: the last trip's refills re-read the row's last blocks (never used).  With the
                            // branches every wait of the trip was vmcnt(0) -- a block waited for the refill it had just requested
                            const int c0 = jn0 < HT ? jn0 : HT - 1, c1 = jn1 < HT ? jn1 : HT - 1;
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[dp][i] = rp[i][4 * c0];
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[d][i] = rp[i][4 * c1];
                        } else {
                            if (jn0 < H16) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[dp][i] = rp[i][4 * jn0];
                            }
                            if (jn1 < H16) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[d][i] = rp[i][4 * jn1];
                            }
                        }
                    }                                             // (RF_DEPTH is even: every block beyond the prologue's has a partner)
                    float4 w[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) w[t] = *reinterpret_cast<const float4*>(tbuf + lane * RF_ROWSTRIDE + 4 * t);
                    __builtin_amdgcn_wave_barrier();
                    const int hc = (int)__float_as_uint(hq[dd >> 2]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int l0 = 16 * (dd & 3) + 4 * t;
                        acc = fmaf(__uint_as_float((unsigned)__builtin_amdgcn_readlane(hc, l0)), w[t].x, acc);
                        acc = fmaf(__uint_as_float((unsigned)__builtin_amdgcn_readlane(hc, l0 + 1)), w[t].y, acc);
                        acc = fmaf(__uint_as_float((unsigned)__builtin_amdgcn_readlane(hc, l0 + 2)), w[t].z, acc);
                        acc = fmaf(__uint_as_float((unsigned)__builtin_amdgcn_readlane(hc, l0 + 3)), w[t].w, acc);
                    }
                }
            }
        };
        for (int j0 = 0; j0 < H16; j0 += RF_U) trip(j0);
        if (Hrem4 && in) {                                       // hidden sizes that are not a multiple of 16: the tail, lane-owned
            const float4* wr = reinterpret_cast<const float4*>(p.x.W32 + (size_t)(colv - p.x.col_lo) * p.x.H);
            for (int t = 0; t < Hrem4; ++t) {
                const float4 wv = wr[4 * H16 + t];
                const int hb = 16 * H16 + 4 * t;
                acc = fmaf(hrow[hb], wv.x, acc);
                acc = fmaf(hrow[hb + 1], wv.y, acc);
                acc = fmaf(hrow[hb + 2], wv.z, acc);
                acc = fmaf(hrow[hb + 3], wv.w, acc);
            }
        }
        return acc + p.x.bias[colv - p.x.col_lo];
    };

    // BOUND GUARD.  The filter launch promised u - 2 eps_c <= z32 <= u for the stored upper bound u of column c; the
    // term of eps_c that covers the accumulation inside v_mfma_f32_32x32x16_bf16 rests on an error MODEL of that
    // instruction (decode_f32.hip exact_bounds_kernel), so every recomputed survivor is tested against the promise --
    // both numbers are in registers -- and a violation is COUNTED in the context's guard words (dae_exact_guard_read):
    // a column whose bound fails may have a sibling that was wrongly filtered out, and the caller must know.
    // (The lower end is taken two floats down: the subtraction rounds.)
    auto guard = [&](float z, float u, int colv, bool in) {
        if (!in || !p.x.guard) return;
        const float e2 = 2.0f * p.x.eps[colv - p.x.col_lo] * 1.000001f;
        const float lo = dae_okey_inv(dae_okey(u - e2) - 2u);
        if (!(z <= u && z >= lo)) {
            atomicAdd(p.x.guard, 1);
            p.x.guard[1] = colv;
        }
    };
    // statistics of the launch (dae_exact_stats_read): per row, the candidates the filter launch left and the candidates
    // recomputed -- PLAIN stores into the row's own slots (three device-scope atomics per row on shared words cost the launch
    // 7 of its 27 us: 256 workgroups queueing on one address)
    if (tid == 0 && p.stat) {
        p.stat[2 * row] = total;
        if (!staged) p.stat[2 * row + 1] = n_kept;                // (narrowed rows store their count when it is known)
    }
    // the row's per-workgroup lists are empty from here on (compact), its own list holds n_kept entries
    auto finish_compact = [&](int n_written) {
        if (p.fuse) return;                                       // (nobody reads the lists after this launch)
        for (int s = tid; s < nseg; s += RF_THREADS) p.cnt[(size_t)s * p.cnt_seg_stride + row] = 0;
        if (tid == 0) p.out_cnt[row] = n_written;
    };
    if (!compact && tid == 0 && p.out_cnt) p.out_cnt[row] = 0;

    // ---- 3. fused selection (p.fuse): the row's k best non-seeds in order, straight to the caller's outputs -------------------
    // n_list = entries of the row's list: keys in LDS (fast), pairs in the row's compact list, or the flat index space of its
    // per-workgroup lists (refined in place; what failed the narrowing is marked -inf there).  The recomputation is over, so the
    // staging area is free: [ordering buffer][bin offsets].
    auto final_select = [&](int n_list) {
        __syncthreads();                                          // the recomputation's last LDS and list accesses
        dae_u64* sorted = reinterpret_cast<dae_u64*>(rf_dyn);     // (the staging area is free now; the bitmap sits behind it)
        unsigned* above = reinterpret_cast<unsigned*>(sorted + DAE_RANK_MAX);
        unsigned* fhist = reinterpret_cast<unsigned*>(surv_off);
        if (fast) {
            for (int b = tid; b < RF_BINS; b += RF_THREADS) fhist[b] = 0u;
            dae_rank_emit<RF_THREADS>(fkey, (unsigned)n_list, sorted, fhist, above, tid, row, p.fo, f_range,
                                      p.stamps ? p.stamps + 16 : nullptr);
            return;
        }
        // more survivors than the ordering stage takes (logits packed within 2 eps of the cut), or a list refined in place:
        // the narrowing over the row's list in global memory (rank_lds.h), every pass re-reading it
        dae_rank_select_emit<RF_THREADS>([&](auto f) {
            for (int e0 = 0; e0 < n_list; e0 += RF_THREADS) {
                const int e = e0 + tid;
                dae_u64 ck = 0ull;
                if (e < n_list) {
                    const uint2 pr = compact ? orow[e] : p.base[offset_of(e)];
                    ck = fkey_of(__uint_as_float(pr.x), (int)pr.y);
                }
                f(ck);
            }
        }, fkey, sorted, fhist, above, tid, row, p.fo);
    };

    if (bad) {                                                   // a row that must return nothing
        if (p.fuse) { dae_rank_pad<RF_THREADS>(tid, row, 0u, p.fo); return; }
        if (compact) { finish_compact(0); return; }
        for (int e = tid; e < total; e += RF_THREADS) p.base[offset_of(e)].x = __float_as_uint(-__builtin_inff());
        return;
    }

    if (!staged) {
        // every candidate is recomputed (few enough that narrowing cannot pay, or too many to stage): the waves work
        // independently on flat groups of 64 -- no list, no barrier; slot of the result = the flat index
        for (int g0 = wave * 64; g0 < total; g0 += RF_WAVES * 64) {
            const int e = g0 + lane;
            const bool in = e < total;
            const int off = offset_of(in ? e : g0);
            const uint2 pr = p.base[off];
            if (g0 == 0) { RSTAMP(5) }
            // (p.pre: recomputed and guarded by exact_rescore_shared_kernel, a decoder row fetched once for 32 playlists)
            const float z = p.pre ? __uint_as_float(pr.x) : rescore_group((int)pr.y, in);
            if (g0 == 0) { RSTAMP(9) }
            if (!p.pre) guard(z, __uint_as_float(pr.x), (int)pr.y, in);
            if (fast) fkey_put(e, z, (int)pr.y, in);
            else if (in) {
                if (compact) orow[e] = make_uint2(__float_as_uint(z), pr.y);
                else p.base[off].x = __float_as_uint(z);
            }
        }
        RSTAMP(6)
        if (compact) finish_compact(total);
        if (p.fuse) final_select(total);
        RSTAMP(8)
        return;
    }

    // narrowed: the flat index space in rounds of RF_UNROLL x RF_THREADS; what passes tau' is listed in LDS by its flat index
    // (the keys are staged: no global load and no segment search in this loop), and the list is recomputed -- a lane per
    // entry, its pair located and fetched one group ahead -- whenever the next round could overflow it and at the end.
    // In place, what fails is marked absent (-inf).
    __syncthreads();                                             // (the count's last reads of the staging area)
    int n_out = 0;                                               // compact: entries written so far (block-uniform)
    constexpr int RND = RF_UNROLL * RF_THREADS;
    for (int c0 = 0; c0 < total; c0 += RND) {
#pragma unroll
        for (int q = 0; q < RF_UNROLL; ++q) {
            const int i = c0 + q * RF_THREADS + tid;
            const bool has = i < total;
            const bool keep = has && skey[has ? i : 0] >= ktau;
            if (has && !keep && !compact) p.base[offset_of(i)].x = __float_as_uint(-__builtin_inff());
            const unsigned long long bal = __ballot(keep);
            if (bal) {
                const int leader = __ffsll((long long)bal) - 1;
                int b = 0;
                if (lane == leader) b = atomicAdd(&s_n, __popcll(bal));
                b = __shfl(b, leader);
                if (keep) surv_off[b + __popcll(bal & ((1ull << lane) - 1ull))] = i;
            }
        }
        __syncthreads();
        const int n = s_n;
        const bool last = c0 + RND >= total;
        if (last) { RSTAMP(5) }                                   // listed
        if (n + RND > RF_SURV || last) {                          // the list could overflow next round, or this was the last
            // (the recomputation's buffers take the head of the staging area: rounds not yet visited are re-staged after it)
            const int gstep = RF_WAVES * 64;
            int g0 = wave * 64;
            uint2 pr = make_uint2(0u, 0u);
            int off = 0;
            if (g0 < n) { off = offset_of(surv_off[g0 + lane < n ? g0 + lane : g0]); pr = p.base[off]; }
            for (; g0 < n; g0 += gstep) {
                const int e = g0 + lane;
                const bool in = e < n;
                const uint2 cur = pr;
                const int off_cur = off;
                const int gn = g0 + gstep;
                if (gn < n) { off = offset_of(surv_off[gn + lane < n ? gn + lane : gn]); pr = p.base[off]; }   // next group's pairs, under this group's rows
                const float z = rescore_group((int)cur.y, in);
                guard(z, __uint_as_float(cur.x), (int)cur.y, in);
                if (fast) fkey_put(n_out + e, z, (int)cur.y, in);
                else if (in) {
                    if (compact) orow[n_out + e] = make_uint2(__float_as_uint(z), cur.y);
                    else p.base[off_cur].x = __float_as_uint(z);
                }
            }
            n_out += n;
            __syncthreads();
            if (tid == 0) s_n = 0;
            if (!last)                                            // tbuf overwrote the head of the staging area
                for (int e = c0 + RND + tid; e < total && e < RF_WAVES * 64 * RF_ROWSTRIDE; e += RF_THREADS)
                    skey[e] = dae_okey(__uint_as_float(p.base[offset_of(e)].x));
            __syncthreads();
        } else {
            // every thread has read n BEFORE anyone appends again: a fast wave's next-round atomicAdd could otherwise change
            // s_n under a slow wave, and the waves would disagree on the flush branch and meet different barriers
            __syncthreads();
        }
    }
    RSTAMP(6)                                                    // recomputed
    if (tid == 0 && p.stat) p.stat[2 * row + 1] = n_out;
    if (compact) finish_compact(n_out);
    RSTAMP(7)
    if (p.fuse) final_select(compact ? n_out : total);
    RSTAMP(8)
}

// ---- shared recomputation: one decoder row for 32 playlists ------------------------------------------------------------------
// A row's workgroup above fetches every survivor's decoder row (1 KiB) for ONE playlist: 534 KiB per playlist through a CU's
// L1, a lane-owned chain of 256 v_readlane + v_fma pairs per candidate.  With one row per CU that is the launch's length
// (~18 us); with many rows per CU the launch grows with rows x candidates -- 2 048 rows: 149 us against 109 for the filter
// launch that decodes the whole vocabulary.  But the rows of a launch mostly want the SAME decoder rows (the popular head of
// the vocabulary: on the bench model every playlist has the same 534 candidates, on a trained one most of them), and the
// canonical chain acc = fmaf(h[k], W[c][k], acc), k ascending, is what v_mfma_f32_32x32x2_f32 performs for 32 columns x 32
// playlists at a time (the fp32 kernels of decode_f32.hip: A = W, B = h, MFMA (g, e) takes k = 8 g + 2 e + hi).  So, for
// launches of many rows, BEFORE the per-row launch:
//   workgroup (segment s, group of 32 playlists): the candidates the filter workgroup s listed for those rows -> the UNION of
//   their columns (an LDS hash table, dense ranks by a scan) -> per tile of 32 union columns the 128 MFMAs over the group's
//   hidden rows (LDS, B-operand order) with the columns' decoder rows straight from the row-major fp32 copy -> a table
//   z[union column][playlist] in LDS -> every candidate takes its own logit (+ bias: the same fp32 addition), is tested
//   against the filter's promise (the bound guard, as above) and REPLACES its bound u in the list.
// Only rows that the per-row launch would recompute without narrowing (same predicate); it then orders what it reads back.
// A decoder row is fetched once per (segment, 32 playlists) instead of once per playlist; columns of the union that a
// playlist did not list are computed for it and ignored (the matrix pipe is idle in this launch anyway).  Bits: identical --
// the same instruction, operand order and k order as the fp32 path's logits, which the lane-owned fmaf chain reproduces.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int RS_ROWS = 32;            // playlists per workgroup (one MFMA row block)
constexpr int RS_PR = 128;             // candidates per playlist and round
constexpr int RS_PMAX = RS_ROWS * RS_PR;
constexpr int RS_HASH = 2 * RS_PMAX;   // slots of the union's hash table (load <= 1/2)
constexpr int RS_UPASS = 128;          // union columns per pass: one tile of 32 per wave
constexpr int RS_ZLD = 33;             // row stride of the z table (floats): conflict-free both ways
constexpr int RS_MAXH = 256;           // hidden sizes the kernel takes (H % 8 == 0): 32 KiB of hidden rows in LDS
constexpr int RS_MAXSEG = 32;          // segments (filter workgroups' lists) one workgroup takes
static_assert(RS_HASH == 8192, "the hash below keeps 13 bits");

struct SharedP {
    uint2* base; const int* cnt; int64_t seg_stride, row_stride, cnt_seg_stride; int nseg;
    dae_exact_src x;
    const int32_t* seed_row_ptr; int k, B, stage_cap;
    int segb, nblk;                    // segments per workgroup, workgroups per group of 32 playlists
    long long* stamps;                 // experiments build: stage stamps of workgroup 0 (DAE_DBG_S)
};
#ifdef DAE_EXPERIMENTS
#define SSTAMP(i) if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0) p.stamps[i] = __builtin_readcyclecounter();
#else
#define SSTAMP(i)
#endif

// grid = nblk x groups of 32 playlists, ~one workgroup per CU: the launch is a chain of memory round trips (counts -> pairs ->
// decoder rows), so a workgroup takes as many segments as give its four waves a tile each.  (Half the tables and two workgroups
// per CU: 20.4 us against 22 alone at 1 024 rows, 8.8 against 9.0 M playlists/s with four batches in flight.)
// GT > 0: hidden = 8 GT known at compile time -- the chain's loop has no branches, so the ring's waits are exact counts (with a
// run-time bound hipcc waits for EVERY load in flight at each step: 30 k cycles per tile against 9 k)
template <int GT>
__global__ __launch_bounds__(256, 1) void exact_rescore_shared_kernel(const SharedP p)
{
    __shared__ __attribute__((aligned(16))) float4 hB[(RS_MAXH / 8) * 64];    // [g][lane = hi 32 + j] {e = 0..3}: h[row j][8 g + 2 e + hi]
    __shared__ float zt[RS_UPASS * RS_ZLD];
    __shared__ unsigned hkey[RS_HASH];             // column + 1 (0: empty); after the scan: the column's rank in the union
    __shared__ int ucol[RS_PMAX];                  // union columns by rank
    __shared__ unsigned short pslot[RS_PMAX];      // hash slot of pair (j, q) of this round
    __shared__ int r_pre[RS_ROWS][RS_MAXSEG + 1];  // per playlist: prefix of its counts over the workgroup's segments
    __shared__ int r_tot[RS_ROWS];
    __shared__ int s_wtot[4], s_maxc;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
    const int blk = blockIdx.x % p.nblk, grp = blockIdx.x / p.nblk;
    const int s0 = blk * p.segb;
    const int ns = p.nseg - s0 < p.segb ? p.nseg - s0 : p.segb;       // segments of this workgroup (>= 1 by the launcher)
    const int row0 = grp * RS_ROWS;
    const int H = GT > 0 ? 8 * GT : p.x.H, G8 = GT > 0 ? GT : (H >> 3);

    // ---- which rows take part, and how many candidates the workgroup's segments hold for them --------------------------------
    SSTAMP(0)
    if (tid < RS_ROWS) r_tot[tid] = 0;
    if (tid == 0) s_maxc = 0;
    {
        uint4* hk4 = reinterpret_cast<uint4*>(hkey);
#pragma unroll
        for (int i = 0; i < RS_HASH / 4 / 256; ++i) hk4[tid + 256 * i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    {
        // (every load of the prologue is requested before the first is used: counts in fours, then the hidden rows)
        const int j = tid & 31, part = tid >> 5;
        const int row = row0 + j;
        const int rowc = row < p.B ? row : p.B - 1;
        float4 v0[4], v1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                  // the group's hidden rows (rows beyond the batch: zeros below)
            const int g = part + 8 * i;
            const float4* src = reinterpret_cast<const float4*>(p.x.h + (size_t)rowc * p.x.ld_h) + 2 * (g < G8 ? g : 0);
            v0[i] = src[0]; v1[i] = src[1];
        }
        int sum = 0;
        for (int sg0 = part; sg0 < p.nseg; sg0 += 32) {
            int c[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sg = sg0 + 8 * i;
                c[i] = p.cnt[(size_t)(sg < p.nseg ? sg : 0) * p.cnt_seg_stride + rowc];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sg = sg0 + 8 * i;
                if (sg < p.nseg && row < p.B) {
                    sum += c[i];
                    if (sg >= s0 && sg < s0 + ns) r_pre[j][sg - s0 + 1] = c[i];
                }
            }
        }
        if (sum) atomicAdd(&r_tot[j], sum);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = part + 8 * i;
            const bool ok = row < p.B && g < G8;
            hB[g * 64 + j] = ok ? make_float4(v0[i].x, v0[i].z, v1[i].x, v1[i].z) : make_float4(0.f, 0.f, 0.f, 0.f);        // hi 0: k = 8 g + 0, 2, 4, 6
            hB[g * 64 + 32 + j] = ok ? make_float4(v0[i].y, v0[i].w, v1[i].y, v1[i].w) : make_float4(0.f, 0.f, 0.f, 0.f);   // hi 1: k = 8 g + 1, 3, 5, 7
        }
    }
    __syncthreads();
    if (tid < RS_ROWS) {
        const int row = row0 + tid;
        bool act = false;
        if (row < p.B) {
            const int total = r_tot[tid];
            const bool bad = p.x.row_bad && p.x.row_bad[row] != 0;
            const int need = p.k + (p.seed_row_ptr ? p.seed_row_ptr[row + 1] - p.seed_row_ptr[row] : 0);
            const bool staged = !bad && total <= p.stage_cap && total > need + (need >> 1);      // (refine_body's predicate)
            act = !bad && !staged && total > 0;
        }
        int run = 0;
        r_pre[tid][0] = 0;
        for (int i = 1; i <= ns; ++i) { run += act ? r_pre[tid][i] : 0; r_pre[tid][i] = run; }
        if (run) atomicMax(&s_maxc, run);
    }
    __syncthreads();
    const int maxc = s_maxc;
    SSTAMP(1)
    if (maxc == 0) return;
    for (int r0 = 0; r0 < maxc; r0 += RS_PR) {
        // ---- the round's pairs -> the union of their columns ------------------------------------------------------------------
        // (a thread's 16 pairs, then their bias and bound, are requested together: one memory round trip each, not sixteen)
        constexpr int NPT = RS_PMAX / 256;
        uint2 pr[NPT]; int off[NPT]; float pb[NPT], pe[NPT];
        {
            // pair -> (segment, position): the segment by COUNTING the prefix entries at or below the flat index -- for all of
            // a thread's pairs together, one independent LDS read each per segment (a search per pair is a chain of them)
            int sgn[NPT], en[NPT], tot[NPT];
#pragma unroll
            for (int n = 0; n < NPT; ++n) {
                const int pid = tid + 256 * n;
                sgn[n] = 0; en[n] = r0 + (pid % RS_PR); tot[n] = r_pre[pid / RS_PR][ns];
            }
            for (int i = 1; i < ns; ++i) {
#pragma unroll
                for (int n = 0; n < NPT; ++n) sgn[n] += r_pre[(tid + 256 * n) / RS_PR][i] <= en[n];
            }
#pragma unroll
            for (int n = 0; n < NPT; ++n) {
                const int j = (tid + 256 * n) / RS_PR;
                off[n] = en[n] < tot[n] ? (int)((int64_t)(s0 + sgn[n]) * p.seg_stride + (int64_t)(row0 + j) * p.row_stride +
                                                (en[n] - r_pre[j][sgn[n]])) : -1;
            }
        }
#pragma unroll
        for (int n = 0; n < NPT; ++n) pr[n] = p.base[off[n] >= 0 ? off[n] : 0];          // (unconditional: no branch between the loads)
#pragma unroll
        for (int n = 0; n < NPT; ++n) {
            const int ci = off[n] >= 0 ? (int)pr[n].y - p.x.col_lo : 0;
            pb[n] = p.x.bias[ci];
            pe[n] = p.x.eps[ci];
        }
        // insert: every first attempt goes out before any answer is looked at; what collided probes on
        unsigned sl[NPT], old[NPT];
#pragma unroll
        for (int n = 0; n < NPT; ++n) {
            sl[n] = (pr[n].y * 2654435761u) >> 19;                             // 13 bits: RS_HASH = 8192 slots
            old[n] = off[n] >= 0 ? atomicCAS(&hkey[sl[n]], 0u, pr[n].y + 1u) : 0u;
        }
#pragma unroll
        for (int n = 0; n < NPT; ++n) {
            if (off[n] >= 0) {
                const unsigned key = pr[n].y + 1u;
                unsigned o = old[n], s_ = sl[n];
                while (o != 0u && o != key) {
                    s_ = (s_ + 1u) & (unsigned)(RS_HASH - 1);
                    o = atomicCAS(&hkey[s_], 0u, key);
                }
                pslot[tid + 256 * n] = (unsigned short)s_;
            }
        }
        __syncthreads();
        if (r0 == 0) { SSTAMP(2) }
        // dense ranks: thread t owns RS_HASH / 256 consecutive slots
        constexpr int SPT = RS_HASH / 256;
        unsigned kv_[SPT];
        {
            const uint4* hk4 = reinterpret_cast<const uint4*>(hkey) + tid * (SPT / 4);
#pragma unroll
            for (int e = 0; e < SPT / 4; ++e) { const uint4 q4 = hk4[e]; kv_[4 * e] = q4.x; kv_[4 * e + 1] = q4.y; kv_[4 * e + 2] = q4.z; kv_[4 * e + 3] = q4.w; }
        }
        int own = 0;
#pragma unroll
        for (int e = 0; e < SPT; ++e) own += kv_[e] != 0u;
        int incl = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        int pre = incl - own;
        for (int w = 0; w < wave; ++w) pre += s_wtot[w];
        const int U = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
        if (own) {
#pragma unroll
            for (int e = 0; e < SPT; ++e) {
                const unsigned kv = kv_[e];
                if (kv != 0u) { ucol[pre] = (int)(kv - 1u); hkey[tid * SPT + e] = (unsigned)pre; ++pre; }
            }
        }
        __syncthreads();
        if (r0 == 0) { SSTAMP(3) if (p.stamps && blockIdx.x == 0 && tid == 0) p.stamps[8] = U; }

        for (int u0 = 0; u0 < U; u0 += RS_UPASS) {
            // ---- one tile of 32 union columns per wave: the canonical chains of 32 x 32 (column, playlist) pairs -------------
            const int t0 = u0 + 32 * wave;
            if (t0 < U) {                                                      // wave-uniform
                const int ui = t0 + (lane & 31);
                const int colv = ucol[ui < U ? ui : U - 1];
                const float4* wr = reinterpret_cast<const float4*>(p.x.W32 + (size_t)(colv - p.x.col_lo) * H);
                // lane (i, hi) fetches k = 8 g + 4 hi .. + 3 of its column's row -- ONE 16-byte load per lane and group, 32 bytes per
                // row and instruction (both halves fetching the row's 32 bytes and selecting took twice the instructions, and the
                // launch was bound by the address unit: 32 rows per instruction) -- and v_permlane32_swap hands each half the
                // k it multiplies: MFMA e takes k = 8 g + 2 e + hi
                constexpr int QD = 8;                                          // k-groups of 8 in flight
                const float4* wq = wr + hi;
                float4 wv[QD];
#pragma unroll
                for (int d = 0; d < QD; ++d)
                    if (GT > 0 || d < G8) wv[d] = wq[2 * d];
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
                for (int g = 0; g < (GT > 0 ? GT : RS_MAXH / 8); ++g) {
                    if (GT > 0 || g < G8) {                                    // wave-uniform (H < 256: the chain ends at H)
                        const float4 w = wv[g % QD];
                        if (GT > 0 ? (g + QD < GT) : (g + QD < G8)) wv[g % QD] = wq[2 * (g + QD)];
                        const float4 b = hB[g * 64 + lane];
                        __builtin_amdgcn_sched_barrier(0);                     // the refill stays AHEAD of this group's MFMAs (hipcc sinks it to its use)
                        // lower half holds k = 0..3 of the group, upper half k = 4..7: swap(x, y) -> {k0 | k1}, {k4 | k5}; swap(z, w) -> {k2 | k3}, {k6 | k7}
                        const auto s01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(w.x), __float_as_uint(w.y), false, false);
                        const auto s23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(w.z), __float_as_uint(w.w), false, false);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(s01[0]), b.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(s23[0]), b.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(s01[1]), b.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(s23[1]), b.w, acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // lane (j, hi), register r: column (r & 3) + 8 (r >> 2) + 4 hi of the tile, playlist j
                const int j = lane & 31;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    zt[(32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hi) * RS_ZLD + j] = acc[r];
            }
            if (r0 == 0 && u0 == 0) { SSTAMP(4) }
            __syncthreads();
            if (r0 == 0 && u0 == 0) { SSTAMP(5) }
            // ---- every pair of the round whose column sits in this pass takes its logit ---------------------------------------
#pragma unroll
            for (int n = 0; n < NPT; ++n) {
                if (off[n] >= 0) {
                    const int pid = tid + 256 * n;
                    const int rk = (int)hkey[pslot[pid]] - u0;
                    if (rk >= 0 && rk < RS_UPASS) {
                        const int colv = (int)pr[n].y;
                        const float z = zt[rk * RS_ZLD + pid / RS_PR] + pb[n];
                        if (p.x.guard) {                                       // the bound guard (refine_body)
                            const float u = __uint_as_float(pr[n].x);
                            const float e2 = 2.0f * pe[n] * 1.000001f;
                            const float lo = dae_okey_inv(dae_okey(u - e2) - 2u);
                            if (!(z <= u && z >= lo)) {
                                atomicAdd(p.x.guard, 1);
                                p.x.guard[1] = colv;
                            }
                        }
                        p.base[off[n]].x = __float_as_uint(z);
                    }
                }
            }
            __syncthreads();
        }
        if (r0 + RS_PR < maxc) {
            uint4* hk4 = reinterpret_cast<uint4*>(hkey);
#pragma unroll
            for (int i = 0; i < RS_HASH / 4 / 256; ++i) hk4[tid + 256 * i] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
        }
        if (r0 == 0) { SSTAMP(6) }
    }
    SSTAMP(7)
}

template <int HT>
__global__ __launch_bounds__(512) void exact_refine_kernel(const RefineP p) { refine_body<512, 8, HT>(p); }
// one row per CU (launches of <= 512 rows): 16 waves, so that the ~530 - 700 recomputed survivors of a row are ONE group of 64 per
// wave -- with 8 waves the ninth group made wave 0 run two groups one after the other, and a group is a chain of memory round
// trips (~9 us): the launch's length was that wave's
template <int HT>
__global__ __launch_bounds__(1024) void exact_refine_wide_kernel(const RefineP p) { refine_body<1024, 4, HT>(p); }
// the shape that shares a CU with another batch's filter workgroup: one wave per SIMD within the 112 registers those leave
template <int HT>
__global__ __launch_bounds__(256) void exact_refine_slim_kernel(const RefineP p)
{
    refine_body<256, 2, HT>(p);
}

}  // namespace

// LDS the fused selection adds BEHIND the staging area: the seed bitmap over the ranked columns (built in the prologue; the
// ordering's buffers reuse the staging area, which holds at least DAE_RANK_MAX * 8 + DAE_RANK_BINS * 4 bytes)
static size_t fuse_bitmap_bytes(int bitmap_n) { return (((size_t)((bitmap_n + 31) >> 5)) * 4 + 15) & ~(size_t)15; }
constexpr size_t RF_DYN_MAX = 128 * 1024;      // dynamic LDS a refine workgroup may ask for (static: ~25 KB)
static_assert((size_t)(RF_STAGE / 2) * sizeof(float) >= (size_t)DAE_RANK_MAX * 8 + (size_t)DAE_RANK_BINS * 4, "ordering buffers");
static_assert(RF_BINS == DAE_RANK_BINS, "the narrowing's and the ordering's histograms share their LDS");

bool dae_exact_refine_can_fuse(const dae_topk_args& a)
{
    return a.k <= DAE_RANK_MAX / 2 && !a.out_pairs && !a.out_tau &&
           (size_t)RF_STAGE * sizeof(float) + fuse_bitmap_bytes(a.bitmap_n) <= RF_DYN_MAX;
}

int dae_launch_exact_refine(dae_ctx* ctx, const dae_pair_group& g1, const dae_exact_src& x, int B, int k,
                            const int32_t* seed_row_ptr, uint2* out, int* out_cnt, int out_cap, int* stat,
                            const dae_topk_args* fa)
{
    if (B <= 0) return DAE_OK;
    if (g1.nseg > RF_MAX_SEG) return dae_fail(ctx, DAE_ERR_ARG, "too many candidate segments (%d)", g1.nseg);
    if (!x.h || !x.W32 || !x.bias || !x.eps_max || !x.eps || (x.H & 3) || x.H > 1024 || !g1.cnt)
        return dae_fail(ctx, DAE_ERR_ARG, "exact refine: bad arguments (H=%d)", x.H);
    RefineP p;
    p.base = const_cast<uint2*>(g1.base); p.cnt = const_cast<int*>(g1.cnt); p.seg_stride = g1.seg_stride; p.row_stride = g1.row_stride;
    p.cnt_seg_stride = g1.cnt_seg_stride; p.nseg = g1.nseg; p.x = x; p.seed_row_ptr = seed_row_ptr; p.k = k;
    p.stat = stat;
    p.fuse = 0; p.fo = dae_rank_out{k, DAE_OUT_LOGIT, nullptr, nullptr}; p.seed_col = nullptr; p.bitmap_base = 0; p.bitmap_n = 0;
    p.row_min = nullptr;
    if (fa) {
        if (!dae_exact_refine_can_fuse(*fa) || fa->k != k || fa->seed_row_ptr != seed_row_ptr)
            return dae_fail(ctx, DAE_ERR_ARG, "exact refine: this selection cannot be fused (k=%d)", fa->k);
        p.fuse = 1; p.fo = dae_rank_out{fa->k, fa->out_kind, fa->out_score, fa->out_idx};
        p.seed_col = fa->seed_col; p.bitmap_base = fa->bitmap_base; p.bitmap_n = fa->bitmap_n; p.row_min = fa->row_min;
    }
    p.out = (out && out_cnt && out_cap > 0) ? out : nullptr; p.out_cnt = p.out ? out_cnt : nullptr; p.out_cap = p.out ? out_cap : 0;
    if ((int64_t)g1.nseg * g1.seg_stride >= ((int64_t)1 << 31))
        return dae_fail(ctx, DAE_ERR_ARG, "exact refine: candidate lists too large for 32-bit offsets");
    p.stamps = nullptr;
#ifdef DAE_EXPERIMENTS
    static const bool dbgR = dae_exp_env("DAE_DBG_R") != nullptr;
    static long long* rbuf = nullptr;
    static int rcalls = 0;
    if (dbgR) {
        if (!rbuf) { (void)hipMalloc(&rbuf, 32 * 8); (void)hipMemset(rbuf, 0, 32 * 8); }
        p.stamps = rbuf;
        if ((++rcalls % 100) == 0) {
            long long h[32];
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemcpy(h, rbuf, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "REFINE wg0:");
            for (int i = 1; i < 10; ++i) if (h[i]) fprintf(stderr, " [%d]%lld", i, h[i] - h[0]);
            for (int i = 16; i < 24; ++i) if (h[i]) fprintf(stderr, " rk%d:%lld", i - 16, h[i] - h[0]);
            fprintf(stderr, "\n");
        }
    }
#endif
    // Many rows per CU (>= 3 at 768 rows): the 256-thread shape with HALF the staging area -- 57 KB of LDS and ~100 registers,
    // so two or three rows share a CU and one's memory round trips (counts, pairs, decoder rows) hide under another's
    // arithmetic; a row is latency-bound on its own (batch 1024: 60 -> see profiles/r04_notes.md).  Rows with more candidates
    // than the smaller area holds are recomputed without narrowing (slower, same bits).
    const bool many = B >= 768;
    bool slim = ctx->overlap_hint || many;
    int shape = slim ? 0 : 1;                                                                  // 0 slim (256), 1 (512), 2 wide (1024)
    // fused selection, one row per CU: the wide shape, whatever the overlap hint says -- measured (profiles/r05_notes.md), four
    // batches in flight: slim + fused 46.1 us per step, 512 threads + fused 41.9, slim + a selection launch 42.2
    if (p.fuse && !many) shape = 2;
    static const char* shape_env = dae_exp_env("DAE_RF_SHAPE");                               // A/B (experiments build)
    if (shape_env) shape = atoi(shape_env);
    slim = shape == 0;
    p.stage_cap = many ? RF_STAGE / 2 : RF_STAGE;
    // launches of many rows: the rows that are recomputed without narrowing get their logits from the shared recomputation
    // (exact_rescore_shared_kernel: a decoder row fetched once per 32 playlists, the chains on the matrix pipe) first
    bool shared = many && (x.H & 7) == 0 && x.H <= RS_MAXH;
    static const char* shared_env = dae_exp_env("DAE_RF_SHARED");                             // A/B (experiments build)
    if (shared_env) shared = atoi(shared_env) != 0 && (x.H & 7) == 0 && x.H <= RS_MAXH;
    p.pre = shared ? 1 : 0; p.B = B;
    if (shared) {
        SharedP sp;
        sp.base = p.base; sp.cnt = p.cnt; sp.seg_stride = p.seg_stride; sp.row_stride = p.row_stride;
        sp.cnt_seg_stride = p.cnt_seg_stride; sp.nseg = p.nseg; sp.x = x; sp.seed_row_ptr = seed_row_ptr; sp.k = k; sp.B = B;
        sp.stage_cap = p.stage_cap;
        const int ngrp = (B + RS_ROWS - 1) / RS_ROWS;
        int nblk = DAE_NUM_CU / ngrp;                                      // ~ one workgroup per CU
        if (nblk < 1) nblk = 1;
        if (nblk > p.nseg) nblk = p.nseg;
        int segb = (p.nseg + nblk - 1) / nblk;
        if (segb > RS_MAXSEG) segb = RS_MAXSEG;
        nblk = (p.nseg + segb - 1) / segb;
        sp.segb = segb; sp.nblk = nblk;
        sp.stamps = nullptr;
#ifdef DAE_EXPERIMENTS
        static const bool dbgS = dae_exp_env("DAE_DBG_S") != nullptr;
        static long long* sbuf = nullptr;
        static int scalls = 0;
        if (dbgS) {
            if (!sbuf) { (void)hipMalloc(&sbuf, 16 * 8); (void)hipMemset(sbuf, 0, 16 * 8); }
            sp.stamps = sbuf;
            if ((++scalls % 100) == 0) {
                long long h[16];
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipMemcpy(h, sbuf, sizeof(h), hipMemcpyDeviceToHost);
                fprintf(stderr, "SHARED wg0 (grid %d x %d, segb %d):", ngrp, nblk, segb);
                for (int i = 1; i < 8; ++i) fprintf(stderr, " [%d]%lld", i, h[i] - h[0]);
                fprintf(stderr, " U=%lld\n", h[8]);
            }
        }
#endif
        if (x.H == 256) hipLaunchKernelGGL(exact_rescore_shared_kernel<32>, dim3((unsigned)(ngrp * nblk)), dim3(256), 0, ctx->stream, sp);
        else hipLaunchKernelGGL(exact_rescore_shared_kernel<0>, dim3((unsigned)(ngrp * nblk)), dim3(256), 0, ctx->stream, sp);
        DAE_CHECK_LAUNCH(ctx, "exact_rescore_shared_kernel");
    }
    // dynamic LDS of a shape: the staging area or the waves' transposition buffers, whichever is larger, + the fused selection's
    // seed bitmap behind them.  dae_exact_refine_can_fuse admits what the 512-thread shape holds; the wide shape's buffers are
    // 16 KB larger, so a vocabulary of 393 217 .. 524 288 ranked columns takes the 512-thread shape instead
    auto dyn_of = [&](int sh) {
        size_t d = (size_t)p.stage_cap * sizeof(float);
        const size_t tb = (size_t)(sh == 0 ? 4 : sh == 1 ? 8 : 16) * 64 * RF_ROWSTRIDE * sizeof(float);
        return d < tb ? tb : d;
    };
    if (p.fuse && shape == 2 && dyn_of(2) + fuse_bitmap_bytes(p.bitmap_n) > RF_DYN_MAX) shape = 1;
    slim = shape == 0;
    size_t dyn = dyn_of(shape);
    p.bm_off = (int)dyn;
    if (p.fuse) dyn += fuse_bitmap_bytes(p.bitmap_n);
    if (dyn > RF_DYN_MAX)
        return dae_fail(ctx, DAE_ERR_ARG, "exact refine: %zu bytes of LDS for %d ranked columns (shape %d)", dyn, p.bitmap_n, shape);
    static const char key = 0;
    if (dae_first_use(ctx, &key)) {
        const int mx = (int)RF_DYN_MAX;
        static_assert(RF_DYN_MAX >= (size_t)RF_STAGE * sizeof(float), "staging area");
        const void* ks[] = {reinterpret_cast<const void*>(&exact_refine_kernel<0>), reinterpret_cast<const void*>(&exact_refine_kernel<16>),
                            reinterpret_cast<const void*>(&exact_refine_slim_kernel<0>),
                            reinterpret_cast<const void*>(&exact_refine_wide_kernel<0>), reinterpret_cast<const void*>(&exact_refine_wide_kernel<16>)};
        for (const void* kf : ks) DAE_HIP_CHECK(ctx, hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    }
    static const bool no_ht = dae_exp_env("DAE_RF_NO_HT") != nullptr;                         // A/B (experiments build)
    const bool h256 = x.H == 256 && !no_ht;                                                    // the compile-time block loop
    if (shape == 0) {
        // (the slim shape keeps the run-time loop: 91 registers against 137 -- it exists to fit next to another batch's filter waves)
        hipLaunchKernelGGL(exact_refine_slim_kernel<0>, dim3(B), dim3(256), dyn, ctx->stream, p);
    } else if (shape == 2) {
        if (h256) hipLaunchKernelGGL(exact_refine_wide_kernel<16>, dim3(B), dim3(1024), dyn, ctx->stream, p);
        else hipLaunchKernelGGL(exact_refine_wide_kernel<0>, dim3(B), dim3(1024), dyn, ctx->stream, p);
    } else {
        if (h256) hipLaunchKernelGGL(exact_refine_kernel<16>, dim3(B), dim3(512), dyn, ctx->stream, p);
        else hipLaunchKernelGGL(exact_refine_kernel<0>, dim3(B), dim3(512), dyn, ctx->stream, p);
    }
    DAE_CHECK_LAUNCH(ctx, "exact_refine_kernel");
    return DAE_OK;
}
