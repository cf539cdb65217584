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

namespace {

constexpr int RF_STAGE = 16384;        // candidates of a row whose u fit the staging area (floats; the waves' buffers reuse it)
constexpr int RF_SURV = 4096;          // survivors listed per flush (offsets only)
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
    long long* stamps;                            // experiments build: stage stamps of workgroup 0 (DAE_DBG_R)
};

#ifdef DAE_EXPERIMENTS
#define RSTAMP(i) if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0) p.stamps[i] = __builtin_readcyclecounter();
#else
#define RSTAMP(i)
#endif

template <int RF_THREADS, int RF_DEPTH>
__device__ __forceinline__ void refine_body(const RefineP& p)
{
    constexpr int RF_WAVES = RF_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char rf_dyn[];      // staging floats, then the waves' buffers
    __shared__ int seg_prefix[RF_MAX_SEG + 2];
    __shared__ __attribute__((aligned(16))) float hrow[1024];
    __shared__ int surv_off[RF_SURV];            // offsets of the listed pairs
    __shared__ unsigned cnts[32];
    __shared__ int s_n;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const int nseg = p.nseg;
    RSTAMP(0)
    const bool bad = p.x.row_bad && p.x.row_bad[row] != 0;          // precondition of the bound violated: nothing survives
    for (int s = tid; s < nseg; s += RF_THREADS) seg_prefix[s + 1] = p.cnt[(size_t)s * p.cnt_seg_stride + row];
    if (tid == 0) { seg_prefix[0] = 0; s_n = 0; }
    if (tid < 32) cnts[tid] = 0u;
    for (int i = tid; i < (p.x.H >> 2); i += RF_THREADS)
        reinterpret_cast<float4*>(hrow)[i] = reinterpret_cast<const float4*>(p.x.h + (size_t)row * p.x.ld_h)[i];
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
    RSTAMP(1)                                                    // counts + hidden row in LDS, prefix done
    const int total = seg_prefix[nseg];
    if (total == 0) {
        if (tid == 0 && p.out_cnt) p.out_cnt[row] = 0;
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
    float* stage_u = reinterpret_cast<float*>(rf_dyn);
    float taup = bad ? __builtin_inff() : -__builtin_inff();
    const bool staged = !bad && total <= RF_STAGE && total > need + (need >> 1);   // narrowing pays when it can drop a third
    int n_kept = bad ? 0 : total;
    if (staged) {
        for (int e0 = tid; e0 < total; e0 += RF_UNROLL * 2 * RF_THREADS) {          // 8 independent loads per thread in flight
            unsigned uv[RF_UNROLL * 2];
#pragma unroll
            for (int q = 0; q < RF_UNROLL * 2; ++q) {
                const int e = e0 + q * RF_THREADS;
                uv[q] = p.base[offset_of(e < total ? e : total - 1)].x;
            }
#pragma unroll
            for (int q = 0; q < RF_UNROLL * 2; ++q) {
                const int e = e0 + q * RF_THREADS;
                if (e < total) stage_u[e] = __uint_as_float(uv[q]);
            }
        }
        __syncthreads();
        RSTAMP(2)                                                // bounds staged
        // largest 20-bit key prefix P with count(key >= P) >= need: 10 four-way steps, counts by ballot, one barrier each
        unsigned P = 0u;
        int step = 0;
        for (int bp = 30; bp >= 12; bp -= 2, ++step) {
            const unsigned c1 = P + (1u << bp), c2 = P + (2u << bp), c3 = P + (3u << bp);
            unsigned n1 = 0, n2 = 0, n3 = 0;
            for (int i0 = 0; i0 < total; i0 += RF_THREADS) {
                const int i = i0 + tid;
                const bool v = i < total;
                const unsigned key = v ? dae_okey(stage_u[i]) : 0u;
                n1 += (unsigned)__popcll(__ballot(v && key >= c1));
                n2 += (unsigned)__popcll(__ballot(v && key >= c2));
                n3 += (unsigned)__popcll(__ballot(v && key >= c3));
            }
            if (lane == 0) {
                if (n1) atomicAdd(&cnts[3 * step + 0], n1);
                if (n2) atomicAdd(&cnts[3 * step + 1], n2);
                if (n3) atomicAdd(&cnts[3 * step + 2], n3);
            }
            __syncthreads();
            const unsigned un = (unsigned)need;
            P = cnts[3 * step + 2] >= un ? c3 : cnts[3 * step + 1] >= un ? c2 : cnts[3 * step + 0] >= un ? c1 : P;
        }
        if (P > DAE_KEY_NEG_INF + 2u) {
            taup = dae_okey_inv(P) - 2.0f * p.x.eps_max[0] * 1.000001f;
            taup = dae_okey_inv(dae_okey(taup) - 2u);          // two floats further down: the subtraction rounded
        }
        RSTAMP(3)                                                // search done
        // how many pass: decides where the results go
        unsigned nk = 0;
        for (int i0 = 0; i0 < total; i0 += RF_THREADS) {
            const int i = i0 + tid;
            nk += (unsigned)__popcll(__ballot(i < total && stage_u[i] >= taup));
        }
        if (lane == 0 && nk) atomicAdd(&cnts[30], nk);
        __syncthreads();
        n_kept = (int)cnts[30];
        RSTAMP(4)
    }
    // COMPACT: the survivors' (fp32 logit, column) pairs go to this row's own list p.out[row][0 .. n_kept) and the
    // filter launch's per-workgroup lists of the row are emptied (their counts zeroed): the selection kernel then reads one
    // short list per row instead of walking 128 segments that hold mostly "absent" marks.  Rows with more survivors
    // than the list holds (thousands of logits within 2 eps of the cut) are refined IN PLACE, as in round 3.
    const bool compact = p.out != nullptr && n_kept <= p.out_cap;
    uint2* const orow = compact ? p.out + (size_t)row * p.out_cap : nullptr;

    // ---- 2. recompute the survivors -------------------------------------------------------------------------------
    float* tbuf = reinterpret_cast<float*>(rf_dyn) + wave * (64 * RF_ROWSTRIDE);      // (shares the staging area: see the barriers)
    const int H16 = p.x.H >> 4;                                  // blocks of 16 k (H % 16 remainder handled below)
    const int Hrem4 = (p.x.H & 15) >> 2;                         // float4 left over after the whole blocks

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
        for (int j0 = 0; j0 < H16; j0 += RF_DEPTH) {
#pragma unroll
            for (int d = 0; d < RF_DEPTH; ++d) {
                const int j = j0 + d;
                if (j < H16) {                                    // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<float4*>(tbuf + (4 * Q + i) * RF_ROWSTRIDE + 4 * q) = v[d][i];
                    __builtin_amdgcn_wave_barrier();              // (a wave's LDS accesses execute in order; keep the compiler from moving them)
                    const int jn = j + RF_DEPTH;
                    if (jn < H16) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[d][i] = rp[i][4 * jn];
                    }
                    float4 w[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) w[t] = *reinterpret_cast<const float4*>(tbuf + lane * RF_ROWSTRIDE + 4 * t);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float4 hv = *reinterpret_cast<const float4*>(hrow + 16 * j + 4 * t);
                        acc = fmaf(hv.x, w[t].x, acc);
                        acc = fmaf(hv.y, w[t].y, acc);
                        acc = fmaf(hv.z, w[t].z, acc);
                        acc = fmaf(hv.w, w[t].w, acc);
                    }
                }
            }
        }
        if (Hrem4 && in) {                                       // hidden sizes that are not a multiple of 16: the tail, lane-owned
            const float4* wr = reinterpret_cast<const float4*>(p.x.W32 + (size_t)(colv - p.x.col_lo) * p.x.H);
            for (int t = 0; t < Hrem4; ++t) {
                const float4 wv = wr[4 * H16 + t];
                const float4 hv = *reinterpret_cast<const float4*>(hrow + 16 * H16 + 4 * t);
                acc = fmaf(hv.x, wv.x, acc);
                acc = fmaf(hv.y, wv.y, acc);
                acc = fmaf(hv.z, wv.z, acc);
                acc = fmaf(hv.w, wv.w, acc);
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
    // statistics of the context (dae_exact_stats_read): rows refined, candidates the filter launch left, candidates recomputed
    if (tid == 0 && p.x.guard) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(p.x.guard + 2);
        atomicAdd(st + 0, 1ull); atomicAdd(st + 1, (unsigned long long)total); atomicAdd(st + 2, (unsigned long long)n_kept);
    }
    // the row's per-workgroup lists are empty from here on (compact), its own list holds n_kept entries
    auto finish_compact = [&]() {
        for (int s = tid; s < nseg; s += RF_THREADS) p.cnt[(size_t)s * p.cnt_seg_stride + row] = 0;
        if (tid == 0) p.out_cnt[row] = n_kept;
    };
    if (!compact && tid == 0 && p.out_cnt) p.out_cnt[row] = 0;

    if (bad) {                                                   // a row that must return nothing
        if (compact) { finish_compact(); return; }
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
            const float z = rescore_group((int)pr.y, in);
            guard(z, __uint_as_float(pr.x), (int)pr.y, in);
            if (in) {
                if (compact) orow[e] = make_uint2(__float_as_uint(z), pr.y);
                else p.base[off].x = __float_as_uint(z);
            }
        }
        if (compact) finish_compact();
        return;
    }

    // narrowed: the flat index space in rounds of RF_UNROLL x RF_THREADS; what passes tau' is listed in LDS by the offset of
    // its pair (the bounds are staged: no global load in this loop), and the list is recomputed -- a lane per entry, the
    // pair fetched one group ahead -- whenever the next round could overflow it and at the end.  In place, what fails is
    // marked absent (-inf).
    __syncthreads();                                             // (the count's last reads of the staging area)
    int n_out = 0;                                               // compact: entries written so far (block-uniform)
    constexpr int RND = RF_UNROLL * RF_THREADS;
    for (int c0 = 0; c0 < total; c0 += RND) {
#pragma unroll
        for (int q = 0; q < RF_UNROLL; ++q) {
            const int i = c0 + q * RF_THREADS + tid;
            const bool has = i < total;
            const bool keep = has && stage_u[has ? i : 0] >= taup;
            int off = 0;
            if (keep || (has && !compact)) off = offset_of(i);
            if (has && !keep && !compact) p.base[off].x = __float_as_uint(-__builtin_inff());
            const unsigned long long bal = __ballot(keep);
            if (bal) {
                const int leader = __ffsll((long long)bal) - 1;
                int b = 0;
                if (lane == leader) b = atomicAdd(&s_n, __popcll(bal));
                b = __shfl(b, leader);
                if (keep) surv_off[b + __popcll(bal & ((1ull << lane) - 1ull))] = off;
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
            if (g0 < n) pr = p.base[surv_off[g0 + lane < n ? g0 + lane : g0]];
            for (; g0 < n; g0 += gstep) {
                const int e = g0 + lane;
                const bool in = e < n;
                const uint2 cur = pr;
                const int gn = g0 + gstep;
                if (gn < n) pr = p.base[surv_off[gn + lane < n ? gn + lane : gn]];      // next group's pairs, under this group's rows
                const float z = rescore_group((int)cur.y, in);
                guard(z, __uint_as_float(cur.x), (int)cur.y, in);
                if (in) {
                    if (compact) orow[n_out + e] = make_uint2(__float_as_uint(z), cur.y);
                    else p.base[surv_off[e]].x = __float_as_uint(z);
                }
            }
            n_out += n;
            __syncthreads();
            if (tid == 0) s_n = 0;
            if (!last)                                            // tbuf overwrote the head of the staging area
                for (int e = c0 + RND + tid; e < total && e < RF_WAVES * 64 * RF_ROWSTRIDE; e += RF_THREADS)
                    stage_u[e] = __uint_as_float(p.base[offset_of(e)].x);
            __syncthreads();
        } else {
            // every thread has read n BEFORE anyone appends again: a fast wave's next-round atomicAdd could otherwise change
            // s_n under a slow wave, and the waves would disagree on the flush branch and meet different barriers
            __syncthreads();
        }
    }
    RSTAMP(6)                                                    // recomputed
    if (compact) finish_compact();
    RSTAMP(7)
}

__global__ __launch_bounds__(512) void exact_refine_kernel(const RefineP p) { refine_body<512, 8>(p); }
// the shape that shares a CU with another batch's filter workgroup: one wave per SIMD within the 112 registers those leave
__global__ __launch_bounds__(256) void exact_refine_slim_kernel(const RefineP p)
{
    refine_body<256, 2>(p);
}

}  // namespace

int dae_launch_exact_refine(dae_ctx* ctx, const dae_pair_group& g1, const dae_exact_src& x, int B, int k,
                            const int32_t* seed_row_ptr, uint2* out, int* out_cnt, int out_cap)
{
    if (B <= 0) return DAE_OK;
    if (g1.nseg > RF_MAX_SEG) return dae_fail(ctx, DAE_ERR_ARG, "too many candidate segments (%d)", g1.nseg);
    if (!x.h || !x.W32 || !x.bias || !x.eps_max || !x.eps || (x.H & 3) || x.H > 1024 || !g1.cnt)
        return dae_fail(ctx, DAE_ERR_ARG, "exact refine: bad arguments (H=%d)", x.H);
    RefineP p;
    p.base = const_cast<uint2*>(g1.base); p.cnt = const_cast<int*>(g1.cnt); p.seg_stride = g1.seg_stride; p.row_stride = g1.row_stride;
    p.cnt_seg_stride = g1.cnt_seg_stride; p.nseg = g1.nseg; p.x = x; p.seed_row_ptr = seed_row_ptr; p.k = k;
    p.out = (out && out_cnt && out_cap > 0) ? out : nullptr; p.out_cnt = p.out ? out_cnt : nullptr; p.out_cap = p.out ? out_cap : 0;
    if ((int64_t)g1.nseg * g1.seg_stride >= ((int64_t)1 << 31))
        return dae_fail(ctx, DAE_ERR_ARG, "exact refine: candidate lists too large for 32-bit offsets");
    p.stamps = nullptr;
#ifdef DAE_EXPERIMENTS
    static const bool dbgR = dae_exp_env("DAE_DBG_R") != nullptr;
    static long long* rbuf = nullptr;
    static int rcalls = 0;
    if (dbgR) {
        if (!rbuf) { (void)hipMalloc(&rbuf, 16 * 8); (void)hipMemset(rbuf, 0, 16 * 8); }
        p.stamps = rbuf;
        if ((++rcalls % 100) == 0) {
            long long h[16];
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemcpy(h, rbuf, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "REFINE wg0:");
            for (int i = 1; i < 8; ++i) if (h[i]) fprintf(stderr, " [%d]%lld", i, h[i] - h[0]);
            fprintf(stderr, "\n");
        }
    }
#endif
    size_t dyn = (size_t)RF_STAGE * sizeof(float);
    if (dyn < (size_t)8 * 64 * RF_ROWSTRIDE * sizeof(float)) dyn = (size_t)8 * 64 * RF_ROWSTRIDE * sizeof(float);
    static const char key = 0;
    if (dae_first_use(ctx, &key)) {
        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&exact_refine_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&exact_refine_slim_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    }
    if (ctx->overlap_hint)
        hipLaunchKernelGGL(exact_refine_slim_kernel, dim3(B), dim3(256), dyn, ctx->stream, p);
    else
        hipLaunchKernelGGL(exact_refine_kernel, dim3(B), dim3(512), dyn, ctx->stream, p);
    DAE_CHECK_LAUNCH(ctx, "exact_refine_kernel");
    return DAE_OK;
}
