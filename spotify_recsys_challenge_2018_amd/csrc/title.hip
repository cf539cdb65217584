// title.hip -- the title scorer of the reference's challenge path:
//   models/title_models/Char_CNN.py:6-75  characters -> embedding -> one "wide" convolution per filter size
//                                         (VALID over the title) -> ReLU -> max over time -> concat -> dropout
//   models/DAEs.py:153-181                y = title_score * w_title + dae_score * w_playlist
// The heavy part of the scorer, sigmoid(features . Output_W + Output_b) over the whole vocabulary, is the
// decoder GEMM again (K2 with hidden = n_sizes * filter_num): the caller prepacks Output_W^T and runs
// dae_decode_dense on the features this file produces.  What is here is small per playlist (25 characters,
// 2.3 MFLOP): one workgroup per playlist, the embedded title in LDS, one thread per (size, filter).
// Padding characters (-1, spotify_reader.py:36) embed to zero (tf.nn.embedding_lookup's GPU behaviour).
#include "dae_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int T_MAX_SIZES = DAE_TITLE_MAX_SIZES;
constexpr int T_MAX_LEN = 64;
constexpr int T_MAX_EMB = 128;

struct TitleP {
    const int32_t* titles; int B, L;
    const float* emb; int n_char, E;
    const float* conv_w;           // size i: [fs_i][E][F] at w_off[i]
    const float* conv_b;           // [n_sizes][F]
    int fs[T_MAX_SIZES]; int w_off[T_MAX_SIZES]; int n_sizes, F;
    float kp; uint32_t seed;
    float* feat; int64_t ld;       // [B, ld], zero beyond n_sizes*F
    int32_t* argmax;               // [B, n_sizes*F] or null: position of the max (training)
    float* feat_raw;               // [B, n_sizes*F] or null: features before dropout (training)
};

__global__ __launch_bounds__(256) void title_features_kernel(const TitleP p)
{
    extern __shared__ float xs[];                     // [L][E]
    const int row = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < p.L * p.E; i += 256) {
        const int pos = i / p.E, c = i - pos * p.E;
        const int t = p.titles[(size_t)row * p.L + pos];
        xs[i] = (t >= 0 && t < p.n_char) ? p.emb[(size_t)t * p.E + c] : 0.0f;
    }
    __syncthreads();
    const int nf = p.n_sizes * p.F;
    for (int fi = tid; fi < nf; fi += 256) {
        const int i = fi / p.F, f = fi - i * p.F;
        const int fs = p.fs[i];
        const float* W = p.conv_w + p.w_off[i] + f;
        const float b = p.conv_b[fi];
        float best = 0.0f;                            // ReLU: max over time of max(conv, 0)
        int arg = 0;
        bool first = true;
        for (int pos = 0; pos + fs <= p.L; ++pos) {
            float acc = b;
            const float* x = xs + pos * p.E;
            for (int q = 0; q < fs * p.E; ++q) acc = fmaf(x[q], W[(size_t)q * p.F], acc);
            acc = acc > 0.0f ? acc : 0.0f;
            if (first || acc > best) { best = acc; arg = pos; first = false; }
        }
        if (p.argmax) p.argmax[(size_t)row * nf + fi] = arg;
        if (p.feat_raw) p.feat_raw[(size_t)row * nf + fi] = best;
        float v = best;
        if (p.kp < 1.0f) v = (v / p.kp) * floorf(p.kp + dae_uniform(p.seed, 2U, (uint32_t)row, (uint32_t)fi));
        p.feat[(size_t)row * p.ld + fi] = v;
    }
    for (int fi = nf + tid; fi < p.ld; fi += 256) p.feat[(size_t)row * p.ld + fi] = 0.0f;
}

// The same features, organised for the machine: one WAVE per (filter size, block of 64 filters), a lane per
// filter, and the accumulators of ALL window positions of that filter in registers.  The weight W[q][f] is
// then loaded once per q (coalesced over the lanes) instead of once per (position, q), and the P positions
// give P independent fmaf chains; x comes from LDS as a broadcast.  Every chain is the one the kernel above
// runs (acc = b, then q ascending), so the two produce the same bits.  1.6 ms -> tens of us at the reference's
// shapes (150 titles x 25 characters, sizes 3/5/7/9 x 100 filters, embedding 50).
// PMAX = compile-time bound of the window positions (L - min(fs) + 1); the LDS image of the title is zero
// padded so that the unused positions read in bounds (their accumulators are ignored).
template <int PMAX, int NW>
__global__ __launch_bounds__(NW * 64) void title_features_wave_kernel(const TitleP p, int lpad)
{
    extern __shared__ float xs[];                     // [lpad][E], rows >= L are zero
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < lpad * p.E; i += NW * 64) {
        const int pos = i / p.E, c = i - pos * p.E;
        float v = 0.0f;
        if (pos < p.L) {
            const int t = p.titles[(size_t)row * p.L + pos];
            if (t >= 0 && t < p.n_char) v = p.emb[(size_t)t * p.E + c];
        }
        xs[i] = v;
    }
    __syncthreads();
    const int nf = p.n_sizes * p.F;
    const int fblocks = (p.F + 63) >> 6;
    for (int task = wave; task < p.n_sizes * fblocks; task += NW) {
        const int i = task / fblocks, fb = task - i * fblocks;
        const int fs = p.fs[i];
        const int P = p.L - fs + 1;                   // >= 1 (checked by the launcher)
        const int f = fb * 64 + lane;
        const bool live = f < p.F;
        const int fc = live ? f : p.F - 1;
        const float* W = p.conv_w + p.w_off[i] + fc;
        const float b = p.conv_b[i * p.F + fc];
        float acc[PMAX];
#pragma unroll
        for (int pos = 0; pos < PMAX; ++pos) acc[pos] = b;
        const int K = fs * p.E;
        // weights in groups of 8: the next group's loads (one L2 round trip, ~600 cycles) are issued under this
        // group's 8 * PMAX fmafs; a single weight ahead left the wave waiting on every q
        constexpr int QB = 8;
        float wb[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) wb[u] = W[(size_t)(u < K ? u : K - 1) * p.F];
        for (int q0 = 0; q0 < K; q0 += QB) {
            float wn[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                const int qn = q0 + QB + u;
                wn[u] = W[(size_t)(qn < K ? qn : K - 1) * p.F];
            }
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                if (q0 + u < K) {                     // wave-uniform
                    const float* x = xs + q0 + u;
#pragma unroll
                    for (int pos = 0; pos < PMAX; ++pos) acc[pos] = fmaf(x[pos * p.E], wb[u], acc[pos]);
                }
            }
#pragma unroll
            for (int u = 0; u < QB; ++u) wb[u] = wn[u];
        }
        float best = 0.0f;
        int arg = 0;
#pragma unroll
        for (int pos = 0; pos < PMAX; ++pos) {
            const float a = acc[pos] > 0.0f ? acc[pos] : 0.0f;                // ReLU, then max over time
            if (pos < P && (pos == 0 || a > best)) { best = a; arg = pos; }
        }
        if (live) {
            const int fi = i * p.F + f;
            if (p.argmax) p.argmax[(size_t)row * nf + fi] = arg;
            if (p.feat_raw) p.feat_raw[(size_t)row * nf + fi] = best;
            float v = best;
            if (p.kp < 1.0f) v = (v / p.kp) * floorf(p.kp + dae_uniform(p.seed, 2U, (uint32_t)row, (uint32_t)fi));
            p.feat[(size_t)row * p.ld + fi] = v;
        }
    }
    for (int fi = nf + tid; fi < p.ld; fi += NW * 64) p.feat[(size_t)row * p.ld + fi] = 0.0f;
}

// The same features on the matrix cores: for one filter size the convolution of a title is the product
// X[pos][q] . W[q][f] with X[pos][q] = xs[pos * E + q] (the overlapping windows ARE the rows of the LDS image), q < fs * E.
// v_mfma_f32_32x32x2_f32 accumulates its two k values as the fmaf chain does (one rounding per product, ascending k:
// DESIGN.md section 2), so with the accumulators preset to the filter's bias the 32 x 32 block of (position, filter)
// sums equals the chains of the two kernels above bit for bit.  One wave = one (filter size, block of 32 filters);
// a workgroup's 8 waves take the tasks from both ends of the (size-major) list so that a wave with a long kernel
// also gets a short one.  A operand: one LDS word per lane (the stride E = 50 puts the 32 positions on 32 different
// even banks, the second k value on the odd ones); B operand: one coalesced 128-byte row piece of W per k value, 16
// k-steps requested ahead.  105 -> ~15 us at the reference's shapes (150 titles x 25 characters, 3/5/7/9 x 100).
template <int NW>
__global__ __launch_bounds__(NW * 64) void title_features_mfma_kernel(const TitleP p, int lpad)
{
    extern __shared__ float xs[];                     // [lpad][E], rows >= L are zero
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, j = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < lpad * p.E; i += NW * 64) {
        const int pos = i / p.E, c = i - pos * p.E;
        float v = 0.0f;
        if (pos < p.L) {
            const int t = p.titles[(size_t)row * p.L + pos];
            if (t >= 0 && t < p.n_char) v = p.emb[(size_t)t * p.E + c];
        }
        xs[i] = v;
    }
    __syncthreads();
    const int nf = p.n_sizes * p.F;
    const int fblocks = (p.F + 31) >> 5;
    const int n_tasks = p.n_sizes * fblocks;
    for (int r = 0;; ++r) {
        if (r * NW + wave >= n_tasks) break;          // r * NW + wave tasks are taken before this one: the ends never cross
        const int fwd = (r >> 1) * NW + wave;
        const int task = (r & 1) ? n_tasks - 1 - fwd : fwd;
        const int i = task / fblocks, fb = task - i * fblocks;
        const int fs = p.fs[i];
        const int P = p.L - fs + 1;                   // 1 .. 32 (checked by the launcher)
        const int f = fb * 32 + j;
        const bool live = f < p.F;
        const int fc = live ? f : p.F - 1;
        const float* W = p.conv_w + p.w_off[i] + fc + (size_t)hi * p.F;        // k = 2 s + hi
        const float* xa = xs + j * p.E + hi;
        const float b = p.conv_b[i * p.F + fc];
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = b;
        const int S = (fs * p.E) >> 1;                // k-steps of two
        constexpr int QB = 8;
        float w0[QB], w1[QB];
        const size_t kstride = (size_t)2 * p.F;
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            w0[u] = W[(size_t)(u < S ? u : S - 1) * kstride];
            w1[u] = W[(size_t)(QB + u < S ? QB + u : S - 1) * kstride];
        }
        for (int s0 = 0; s0 < S; s0 += QB) {
            float wn[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                const int sn = s0 + 2 * QB + u;
                wn[u] = W[(size_t)(sn < S ? sn : S - 1) * kstride];
            }
            float xq[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u) xq[u] = xa[2 * (s0 + u < S ? s0 + u : S - 1)];
#pragma unroll
            for (int u = 0; u < QB; ++u)
                if (s0 + u < S)                       // wave-uniform
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xq[u], w0[u], acc, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < QB; ++u) { w0[u] = w1[u]; w1[u] = wn[u]; }
        }
        // accumulator e of lane (j, hi): position 8 (e / 4) + 4 hi + e % 4, filter j.  ReLU, then the FIRST maximum over
        // the positions in ascending order (the chain kernels' rule): per lane first, then across the two halves
        float best = 0.0f;
        int arg = 0;
        bool any = false;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int pos = 8 * (e >> 2) + 4 * hi + (e & 3);
            const float a = acc[e] > 0.0f ? acc[e] : 0.0f;
            if (pos < P && (!any || a > best)) { best = a; arg = pos; any = true; }
        }
        const float ob = __shfl_xor(best, 32);
        const int oa = __shfl_xor(arg, 32);
        const bool oany = __shfl_xor((int)any, 32) != 0;
        if (oany && (!any || ob > best || (ob == best && oa < arg))) { best = ob; arg = oa; }
        if (live && hi == 0) {
            const int fi = i * p.F + f;
            if (p.argmax) p.argmax[(size_t)row * nf + fi] = arg;
            if (p.feat_raw) p.feat_raw[(size_t)row * nf + fi] = best;
            float v = best;
            if (p.kp < 1.0f) v = (v / p.kp) * floorf(p.kp + dae_uniform(p.seed, 2U, (uint32_t)row, (uint32_t)fi));
            p.feat[(size_t)row * p.ld + fi] = v;
        }
    }
    for (int fi = nf + tid; fi < p.ld; fi += NW * 64) p.feat[(size_t)row * p.ld + fi] = 0.0f;
}

// ---- a FROZEN scorer's convolutions as a table (dae_title_prepack_features) --------------------------------------------------
// Scoring (`--challenge`, main_challenge.py:80-90) runs the title scorer with fixed variables on ~10^4 batches.  A title is a
// row of CHARACTER IDS (41 of them + padding, spotify_reader.py:36), so the convolution's inner sum over the embedding,
//     conv[pos][f] = b[f] + sum_d sum_e emb[t[pos + d]][e] W[d][e][f],
// has only fs x (n_char + 1) distinct inner terms per filter: T[d][c][f] = sum_e emb[c][e] W[d][e][f] (the fmaf chain over e
// from +0; padding and out-of-range ids: the zero row).  With the table conv[pos][f] = b[f] + T[0][t[pos]][f] + ... (d
// ascending) costs fs additions instead of fs x E multiply-adds: 24 x 42 x 100 floats = 403 KB, resident in L2, built in
// microseconds; 750 titles: 106 us of fp32 MFMA (title_features_mfma_kernel) -> a few us of table reads.
// ARITHMETIC: the same real number as the chain kernels above, summed in another order (e first, then d) -- a different fp32
// rounding of it, as any other order would be (TF's conv2d has no documented order: the title path is compared with the
// reference restatement to a TOLERANCE, oracle/title_numpy.py; tests/test_gpu_title.py bounds the two orders against each
// other).  Used ONLY for inference calls (keep_prob = 1, no argmax / raw features wanted) on a context whose table was built
// from the very arrays the call passes; training keeps the chains.  fp32 and exact_bf16 title scoring read the same features,
// so their lists stay bit-identical to each other.
__global__ __launch_bounds__(128) void title_table_build_kernel(const float* __restrict__ emb, int n_char, int E,
                                                                const float* __restrict__ Wd, int F, float* __restrict__ T)
{
    // blockIdx.x = d * (n_char + 1) + c of ONE filter size (Wd = its [fs][E][F] weights, T = its [fs][n_char + 1][F] table)
    const int nc1 = n_char + 1;
    const int d = blockIdx.x / nc1, c = blockIdx.x - d * nc1;
    for (int f = threadIdx.x; f < F; f += 128) {
        float acc = 0.0f;
        if (c < n_char)
            for (int e = 0; e < E; ++e) acc = fmaf(emb[(size_t)c * E + e], Wd[((size_t)d * E + e) * F + f], acc);
        T[(size_t)blockIdx.x * F + f] = acc;
    }
}

constexpr int TT_PMAX = 32;        // window positions per filter size (the MFMA kernel's bound as well)
__global__ __launch_bounds__(256) void title_features_table_kernel(const TitleP p, const float* __restrict__ T)
{
    __shared__ int ts[T_MAX_LEN + TT_PMAX];           // the title's table rows (ids; padding / invalid -> the zero row), zero-row padded
    const int row = blockIdx.x, tid = threadIdx.x;
    const int nc1 = p.n_char + 1;
    for (int i = tid; i < T_MAX_LEN + TT_PMAX; i += 256) {
        int t = p.n_char;
        if (i < p.L) { const int v = p.titles[(size_t)row * p.L + i]; if (v >= 0 && v < p.n_char) t = v; }
        ts[i] = t;
    }
    __syncthreads();
    const int nf = p.n_sizes * p.F;
    for (int fi = tid; fi < nf; fi += 256) {
        const int i = fi / p.F, f = fi - i * p.F;
        const int fs = p.fs[i];
        const int P = p.L - fs + 1;                   // 1 .. 32 (the launcher checks)
        int t_off = 0;
        for (int q = 0; q < i; ++q) t_off += p.fs[q] * nc1 * p.F;
        const float* Tf = T + t_off + f;
        const float b = p.conv_b[fi];
        float acc[TT_PMAX];
#pragma unroll
        for (int pos = 0; pos < TT_PMAX; ++pos) acc[pos] = b;
        for (int d = 0; d < fs; ++d) {                // d ascending; the 32 positions' reads are independent of each other
            const float* Td = Tf + (size_t)d * nc1 * p.F;
            // (positions beyond the size's own P read the ZERO row -- one cached line -- instead of being skipped: a predicated
            // load is a branch, and hipcc then waits for every load before the next one goes out: 460 memory round trips in a
            // row, 50.7 us for 750 titles against 33.9 us.  Measured on top and dropped: 512 threads (one pass over the 400
            // filters) 34.7 us; two offsets d per round and 24 positions 43.3 us)
            float tv[TT_PMAX];
#pragma unroll
            for (int pos = 0; pos < TT_PMAX; ++pos) tv[pos] = Td[(size_t)(pos < P ? ts[pos + d] : p.n_char) * p.F];
#pragma unroll
            for (int pos = 0; pos < TT_PMAX; ++pos) acc[pos] += tv[pos];
        }
        float best = 0.0f;                            // ReLU, then the first maximum over the positions
        bool first = true;
#pragma unroll
        for (int pos = 0; pos < TT_PMAX; ++pos) {
            const float a = acc[pos] > 0.0f ? acc[pos] : 0.0f;
            if (pos < P && (first || a > best)) { best = a; first = false; }
        }
        p.feat[(size_t)row * p.ld + fi] = best;
    }
    for (int fi = nf + tid; fi < p.ld; fi += 256) p.feat[(size_t)row * p.ld + fi] = 0.0f;
}

// y = title * w_title[row] + dae * w_playlist[row], written over the dae scores (DAEs.py:180)
__global__ __launch_bounds__(256) void mix_scores_kernel(const float* __restrict__ ts, int64_t ld_t,
                                                         float* __restrict__ ds, int64_t ld_d,
                                                         const float* __restrict__ wt,
                                                         const float* __restrict__ wp, int B, int ncols)
{
    const int row = blockIdx.y;
    const float a = wt[row], b = wp[row];
    const float* t = ts + (size_t)row * ld_t;
    float* d = ds + (size_t)row * ld_d;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < ncols; c += gridDim.x * 256)
        d[c] = t[c] * a + d[c] * b;
}

// ---- training of the title variables (DAEs.py:183-198: the DAE constants are frozen) -------------------

// reduce_sum of DAEs.py:41 -- the row sums of the dropped-out input, with the encode kernel's draws
__global__ __launch_bounds__(256) void row_sums_kernel(const int32_t* __restrict__ row_ptr,
                                                       const int32_t* __restrict__ col,
                                                       const float* __restrict__ val, int B, float ikp,
                                                       uint32_t seed, float* __restrict__ out)
{
    // one wave per row: 64 entries per load, added one after the other in entry order -- the encode kernels' order
    // (one thread per row with a dependent load per entry took 37 us for 150 rows)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const int beg = row_ptr[row], end = row_ptr[row + 1];
    float s = 0.0f;
    for (int base = beg; base < end; base += 64) {
        const int n = min(64, end - base);
        float x = 0.0f;
        if (lane < n) {
            x = val[base + lane];
            if (ikp < 1.0f)
                x = (x / ikp) * floorf(ikp + dae_uniform(seed, 0U, (uint32_t)row, (uint32_t)col[base + lane]));
        }
        for (int i = 0; i < n; ++i) s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), i));
    }
    if (lane == 0) out[row] = s;
}

// the mixing weights of DAEs.py:159-162 in one launch: x_count = reduce_sum * input_keep_prob (the row sum as above),
// deno = titles_use + x_count + 1e-10, w_title = titles_use / deno, w_playlist = x_count / deno -- fp32 operations in the
// reference's order (what DAE_title._mix_weights did with five elementwise launches after dae_row_sums)
__global__ __launch_bounds__(256) void mix_weights_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                          const float* __restrict__ val, int B, float ikp, uint32_t seed,
                                                          const float* __restrict__ use, float* __restrict__ w_t,
                                                          float* __restrict__ w_p)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const int beg = row_ptr[row], end = row_ptr[row + 1];
    float s = 0.0f;
    for (int base = beg; base < end; base += 64) {
        const int n = min(64, end - base);
        float x = 0.0f;
        if (lane < n) {
            x = val[base + lane];
            if (ikp < 1.0f)
                x = (x / ikp) * floorf(ikp + dae_uniform(seed, 0U, (uint32_t)row, (uint32_t)col[base + lane]));
        }
        for (int i = 0; i < n; ++i) s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), i));
    }
    if (lane == 0) {
        const float u = use[row];
        const float xc = s * ikp;
        const float deno = (u + xc) + 1e-10f;
        w_t[row] = u / deno;
        w_p[row] = xc / deno;
    }
}

__global__ __launch_bounds__(256) void title_scatter_y_kernel(const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ col,
                                                              const float* __restrict__ val, int B, int V,
                                                              float* __restrict__ y)
{
    const int row = blockIdx.x;
    for (int i = row_ptr[row] + threadIdx.x; i < row_ptr[row + 1]; i += 256)
        if (col[i] >= 0 && col[i] < V) y[(size_t)row * V + col[i]] = val[i];
}

// Weighted BCE of the MIXED score and its gradient w.r.t. the title logits, written TRANSPOSED (the layout
// the two backward GEMMs read), one 32 x 32 (column x playlist) tile per workgroup through LDS:
//   yp = sigmoid(zt) * w_t + dae * w_p;  L = -[y log(yp + 1e-10) + 0.55 (1 - y) log(1 - yp + 1e-10)]
//   dzt = dL/dyp * w_t * st (1 - st) / n_batch
__global__ __launch_bounds__(256) void title_loss_kernel(const float* __restrict__ zt, int64_t ld_z,
                                                         const float* __restrict__ dae, int64_t ld_d,
                                                         const float* __restrict__ y, const float* __restrict__ wt,
                                                         const float* __restrict__ wp, int B, int V, float inv_nb,
                                                         float* __restrict__ dzT, int64_t ldT,
                                                         float* __restrict__ loss_part)
{
    __shared__ float tile[32][33];
    __shared__ float wsum[4];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int v0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    float loss = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, v = v0 + tx;
        float dz = 0.0f;
        if (r < B && v < V) {
            const float z = zt[(size_t)r * ld_z + v];
            const float st = 1.0f / (1.0f + __expf(-z));
            const float a = wt[r];
            const float yp = st * a + dae[(size_t)r * ld_d + v] * wp[r];
            const float t = y[(size_t)r * V + v];
            const float a1 = yp + 1e-10f, a0 = 1.0f - yp + 1e-10f;
            loss -= t * __logf(a1) + 0.55f * (1.0f - t) * __logf(a0);
            dz = -(t / a1 - 0.55f * (1.0f - t) / a0) * inv_nb * a * st * (1.0f - st);
        }
        tile[ty + 8 * i][tx] = dz;                                    // [r local][v local]
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = v0 + ty + 8 * i, r = r0 + tx;
        if (v < V && r < ldT) dzT[(size_t)v * ldT + r] = tile[tx][ty + 8 * i];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) loss += __shfl_xor(loss, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0)
        loss_part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) * inv_nb;
}

__global__ void title_cost_kernel(const float* __restrict__ part, int n, float* __restrict__ cost)
{
    __shared__ double ws[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)part[i];   // fixed assignment, fixed tree
    ws[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) ws[threadIdx.x] += ws[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) *cost = (float)ws[0];
}

// back through dropout -> max over time -> ReLU -> convolution -> embedding (Char_CNN.py:23-63).  The gradient
// of one feature goes to the single window that won the max (its argmax), and only if the ReLU was open.
// Three small kernels, none with an atomic per weight element:
//   gate   dg[b, fi] = dfeat * dropout mask / kp if the feature's ReLU was open, else 0
//   wgrad  gW[i][dp][c][f] = sum_b dg[b, i, f] * x[b][arg[b, i, f] + dp][c]   one thread per (dp, c, f), loop over b
//          gb[i][f]        = sum_b dg[b, i, f]
//   egrad  per playlist: gx[pos][c] = sum_{i, f, dp: arg + dp == pos} dg * W[i][dp][c][f] in LDS, then ONE global
//          atomicAdd per (pos, c) into the embedding row of the character at pos
__global__ __launch_bounds__(256) void title_gate_kernel(const TitleP p, const float* __restrict__ dfeat,
                                                         float* __restrict__ dg)
{
    const int nf = p.n_sizes * p.F;
    const size_t n = (size_t)p.B * nf;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (size_t)gridDim.x * 256) {
        const int row = (int)(o / nf), fi = (int)(o - (size_t)row * nf);
        float d = 0.0f;
        if (p.feat_raw[o] > 0.0f) {                       // ReLU open (a row whose windows are all <= 0 gets nothing)
            d = dfeat[(size_t)row * p.ld + fi];
            if (p.kp < 1.0f) d = (d / p.kp) * floorf(p.kp + dae_uniform(p.seed, 2U, (uint32_t)row, (uint32_t)fi));
        }
        dg[o] = d;
    }
}

// grid = (sum of filter sizes, E); block = F threads (rounded up to a wave)
__global__ __launch_bounds__(256) void title_wgrad_kernel(const TitleP p, const float* __restrict__ dg,
                                                          float* __restrict__ g_conv_w, float* __restrict__ g_conv_b)
{
    int i = 0, dp = blockIdx.x;
    while (dp >= p.fs[i]) { dp -= p.fs[i]; ++i; }                     // (size index, offset in the window)
    const int c = blockIdx.y, f = threadIdx.x;
    if (f >= p.F) return;
    const int nf = p.n_sizes * p.F, fi = i * p.F + f;
    float acc = 0.0f, accb = 0.0f;
    for (int b = 0; b < p.B; ++b) {
        const float d = dg[(size_t)b * nf + fi];
        accb += d;
        if (d != 0.0f) {
            const int t = p.titles[(size_t)b * p.L + p.argmax[(size_t)b * nf + fi] + dp];
            if (t >= 0 && t < p.n_char) acc = fmaf(d, p.emb[(size_t)t * p.E + c], acc);
        }
    }
    g_conv_w[p.w_off[i] + (size_t)(dp * p.E + c) * p.F + f] = acc;
    if (dp == 0 && c == 0) g_conv_b[fi] = accb;
}

__global__ __launch_bounds__(256) void title_egrad_kernel(const TitleP p, const float* __restrict__ dg,
                                                          float* __restrict__ g_emb)
{
    extern __shared__ float gx[];                     // [L][E]
    const int row = blockIdx.x, tid = threadIdx.x;
    for (int q = tid; q < p.L * p.E; q += 256) gx[q] = 0.0f;
    __syncthreads();
    const int nf = p.n_sizes * p.F;
    // thread = (window offset dp, channel c) pairs walked for every open feature: conflict-free in c
    for (int fi = 0; fi < nf; ++fi) {
        const float d = dg[(size_t)row * nf + fi];
        if (d == 0.0f) continue;                                       // uniform: dg is read by all threads alike
        const int i = fi / p.F, f = fi - i * p.F;
        const int pos = p.argmax[(size_t)row * nf + fi];
        const float* W = p.conv_w + p.w_off[i] + f;
        for (int q = tid; q < p.fs[i] * p.E; q += 256)                 // q = dp * E + c -> gx[(pos + dp) * E + c]
            atomicAdd(&gx[pos * p.E + q], d * W[(size_t)q * p.F]);       // LDS add: waves run ahead of each other
    }
    __syncthreads();
    for (int q = tid; q < p.L * p.E; q += 256) {
        const int pos = q / p.E, c = q - pos * p.E;
        const int t = p.titles[(size_t)row * p.L + pos];
        if (t >= 0 && t < p.n_char && gx[q] != 0.0f) atomicAdd(&g_emb[(size_t)t * p.E + c], gx[q]);
    }
}

}  // namespace

static int title_fill(dae_ctx* ctx, TitleP& p, const int32_t* titles, int B, int L, const float* emb, int n_char,
                      int E, const float* conv_w, const float* conv_b, const int32_t* filter_sizes, int n_sizes,
                      int F, float kp, uint32_t seed, int64_t ld)
{
    if (n_sizes < 1 || n_sizes > T_MAX_SIZES) return dae_fail(ctx, DAE_ERR_ARG, "n_sizes=%d out of [1,%d]", n_sizes, T_MAX_SIZES);
    if (L < 1 || L > T_MAX_LEN || E < 1 || E > T_MAX_EMB) return dae_fail(ctx, DAE_ERR_ARG, "title length %d / embedding %d too large", L, E);
    if (ld < (int64_t)n_sizes * F) return dae_fail(ctx, DAE_ERR_ARG, "ld=%lld < %d features", (long long)ld, n_sizes * F);
    memset(&p, 0, sizeof(p));
    p.titles = titles; p.B = B; p.L = L; p.emb = emb; p.n_char = n_char; p.E = E;
    p.conv_w = conv_w; p.conv_b = conv_b; p.n_sizes = n_sizes; p.F = F; p.kp = kp; p.seed = seed; p.ld = ld;
    int off = 0;
    for (int i = 0; i < n_sizes; ++i) {
        if (filter_sizes[i] < 1 || filter_sizes[i] > L) return dae_fail(ctx, DAE_ERR_ARG, "filter size %d outside [1,%d]", filter_sizes[i], L);
        p.fs[i] = filter_sizes[i]; p.w_off[i] = off;
        off += filter_sizes[i] * E * F;
    }
    return DAE_OK;
}

int dae_launch_row_sums(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val, int B,
                        float ikp, uint32_t seed, float* out)
{
    hipLaunchKernelGGL(row_sums_kernel, dim3((B + 3) / 4), dim3(256), 0, ctx->stream, row_ptr, col, val, B, ikp,
                       seed, out);
    DAE_CHECK_LAUNCH(ctx, "row_sums_kernel");
    return DAE_OK;
}

int dae_launch_mix_weights(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val, int B, float ikp,
                           uint32_t seed, const float* use, float* w_t, float* w_p)
{
    hipLaunchKernelGGL(mix_weights_kernel, dim3((B + 3) / 4), dim3(256), 0, ctx->stream, row_ptr, col, val, B, ikp, seed, use,
                       w_t, w_p);
    DAE_CHECK_LAUNCH(ctx, "mix_weights_kernel");
    return DAE_OK;
}

int dae_launch_title_loss_backward(dae_ctx* ctx, const float* zt, int64_t ld_z, const float* dae_score, int64_t ld_d,
                                   const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
                                   const float* w_title, const float* w_playlist, int B, int V, int n_batch,
                                   const float* feat, int ld, const float* Output_WT, float* gOutput_WT,
                                   float* gOutput_b, float* dfeat, float* cost_out)
{
    hipStream_t st = ctx->stream;
    int rc;
    const int Bpad64 = (B + 63) / 64 * 64;
    if ((rc = dae_reserve(ctx, ctx->train_a, (size_t)B * V * sizeof(float)))) return rc;        // dense targets
    if ((rc = dae_reserve(ctx, ctx->train_b, (size_t)V * Bpad64 * sizeof(float)))) return rc;   // dz^T
    const dim3 grid((V + 31) / 32, Bpad64 / 32);
    const size_t n_part = (size_t)grid.x * grid.y;
    if ((rc = dae_reserve(ctx, ctx->train_c, n_part * sizeof(float)))) return rc;
    float* y = static_cast<float*>(ctx->train_a.p);
    float* dzT = static_cast<float*>(ctx->train_b.p);
    float* part = static_cast<float*>(ctx->train_c.p);
    DAE_HIP_CHECK(ctx, hipMemsetAsync(y, 0, (size_t)B * V * sizeof(float), st));
    hipLaunchKernelGGL(title_scatter_y_kernel, dim3(B), dim3(256), 0, st, y_row_ptr, y_col, y_val, B, V, y);
    DAE_CHECK_LAUNCH(ctx, "title_scatter_y_kernel");
    hipLaunchKernelGGL(title_loss_kernel, grid, dim3(256), 0, st, zt, ld_z, dae_score, ld_d, y, w_title, w_playlist,
                       B, V, 1.0f / (float)n_batch, dzT, (int64_t)Bpad64, part);
    DAE_CHECK_LAUNCH(ctx, "title_loss_kernel");
    hipLaunchKernelGGL(title_cost_kernel, dim3(1), dim3(256), 0, st, part, (int)n_part, cost_out);
    DAE_CHECK_LAUNCH(ctx, "title_cost_kernel");
    if ((rc = dae_launch_grad_w(ctx, dzT, Bpad64, feat, ld, B, V, gOutput_WT, gOutput_b))) return rc;
    return dae_launch_grad_h(ctx, dzT, Bpad64, Output_WT, ld, V, B, dfeat);
}

int dae_launch_title_conv_backward(dae_ctx* ctx, const int32_t* titles, int B, int L, const float* emb, int n_char,
                                   int E, const float* conv_w, const int32_t* filter_sizes, int n_sizes, int F,
                                   const int32_t* argmax, const float* feat_raw, const float* dfeat, int64_t ld,
                                   float kp, uint32_t seed, float* g_emb, float* g_conv_w, float* g_conv_b)
{
    TitleP p;
    int rc = title_fill(ctx, p, titles, B, L, emb, n_char, E, conv_w, nullptr, filter_sizes, n_sizes, F, kp, seed, ld);
    if (rc) return rc;
    p.argmax = const_cast<int32_t*>(argmax);
    p.feat_raw = const_cast<float*>(feat_raw);
    DAE_HIP_CHECK(ctx, hipMemsetAsync(g_emb, 0, (size_t)n_char * E * sizeof(float), ctx->stream));
    if (F > 256) return dae_fail(ctx, DAE_ERR_ARG, "filter_num %d > 256", F);
    p.B = B;
    rc = dae_reserve(ctx, ctx->train_c, (size_t)B * n_sizes * F * sizeof(float));
    if (rc) return rc;
    float* dg = static_cast<float*>(ctx->train_c.p);
    int blocks = (int)(((size_t)B * n_sizes * F + 255) / 256);
    hipLaunchKernelGGL(title_gate_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, ctx->stream, p, dfeat, dg);
    DAE_CHECK_LAUNCH(ctx, "title_gate_kernel");
    int sum_fs = 0;
    for (int i = 0; i < n_sizes; ++i) sum_fs += filter_sizes[i];
    hipLaunchKernelGGL(title_wgrad_kernel, dim3(sum_fs, E), dim3((F + 63) / 64 * 64), 0, ctx->stream, p, dg, g_conv_w,
                       g_conv_b);
    DAE_CHECK_LAUNCH(ctx, "title_wgrad_kernel");
    hipLaunchKernelGGL(title_egrad_kernel, dim3(B), dim3(256), (size_t)L * E * sizeof(float), ctx->stream, p, dg, g_emb);
    DAE_CHECK_LAUNCH(ctx, "title_egrad_kernel");
    return DAE_OK;
}

// emb == nullptr: drop the table (the variables are about to change)
int dae_launch_title_table(dae_ctx* ctx, const float* emb, int n_char, int E, const float* conv_w, const int32_t* filter_sizes,
                           int n_sizes, int F)
{
    ctx->ttab_emb = nullptr; ctx->ttab_w = nullptr;
    if (!emb) return DAE_OK;
    if (n_sizes < 1 || n_sizes > T_MAX_SIZES || n_char < 1 || E < 1 || E > T_MAX_EMB || F < 1)
        return dae_fail(ctx, DAE_ERR_ARG, "dae_title_prepack_features: bad shape");
    size_t total = 0;
    for (int i = 0; i < n_sizes; ++i) {
        if (filter_sizes[i] < 1 || filter_sizes[i] > T_MAX_LEN) return dae_fail(ctx, DAE_ERR_ARG, "filter size %d", filter_sizes[i]);
        total += (size_t)filter_sizes[i] * (n_char + 1) * F;
    }
    int rc = dae_reserve(ctx, ctx->title_tab, total * sizeof(float));
    if (rc) return rc;
    size_t t_off = 0, w_off = 0;
    for (int i = 0; i < n_sizes; ++i) {
        const int fs = filter_sizes[i];
        hipLaunchKernelGGL(title_table_build_kernel, dim3((unsigned)(fs * (n_char + 1))), dim3(128), 0, ctx->stream, emb, n_char, E,
                           conv_w + w_off, F, static_cast<float*>(ctx->title_tab.p) + t_off);
        DAE_CHECK_LAUNCH(ctx, "title_table_build_kernel");
        t_off += (size_t)fs * (n_char + 1) * F; w_off += (size_t)fs * E * F;
        ctx->ttab_fs[i] = fs;
    }
    ctx->ttab_emb = emb; ctx->ttab_w = conv_w; ctx->ttab_nchar = n_char; ctx->ttab_E = E; ctx->ttab_F = F; ctx->ttab_nsizes = n_sizes;
    return DAE_OK;
}

int dae_launch_title_features(dae_ctx* ctx, const int32_t* titles, int B, int L, const float* emb, int n_char,
                              int E, const float* conv_w, const float* conv_b, const int32_t* filter_sizes,
                              int n_sizes, int F, float kp, uint32_t seed, float* feat, int64_t ld,
                              int32_t* argmax, float* feat_raw)
{
    TitleP p;
    int rc = title_fill(ctx, p, titles, B, L, emb, n_char, E, conv_w, conv_b, filter_sizes, n_sizes, F, kp, seed, ld);
    if (rc) return rc;
    p.feat = feat; p.argmax = argmax; p.feat_raw = feat_raw;
    int fs_min = L, fs_max = 1;
    for (int i = 0; i < n_sizes; ++i) { fs_min = p.fs[i] < fs_min ? p.fs[i] : fs_min; fs_max = p.fs[i] > fs_max ? p.fs[i] : fs_max; }
    const int p_max = L - fs_min + 1;                 // most window positions of any size
    // inference on a context that holds the table of exactly these variables (dae_title_prepack_features): fs additions per
    // (position, filter) instead of fs x E multiply-adds
    static const bool no_table = dae_exp_env("DAE_TITLE_NO_TABLE") != nullptr;                // A/B (experiments build)
    if (ctx->title_tab.p && ctx->ttab_emb == emb && ctx->ttab_w == conv_w && ctx->ttab_nchar == n_char && ctx->ttab_E == E &&
        ctx->ttab_F == F && ctx->ttab_nsizes == n_sizes && kp == 1.0f && !argmax && !feat_raw && p_max >= 1 && p_max <= TT_PMAX &&
        !no_table) {
        bool same = true;
        for (int i = 0; i < n_sizes; ++i) same = same && ctx->ttab_fs[i] == p.fs[i];
        if (same) {
            hipLaunchKernelGGL(title_features_table_kernel, dim3(B), dim3(256), 0, ctx->stream, p,
                               static_cast<const float*>(ctx->title_tab.p));
            DAE_CHECK_LAUNCH(ctx, "title_features_table_kernel");
            return DAE_OK;
        }
    }
    static const bool generic = dae_exp_env("DAE_TITLE_GENERIC") != nullptr;          // A/B against the first kernel
    bool even_k = true;
    for (int i = 0; i < n_sizes; ++i) even_k = even_k && ((p.fs[i] * E) % 2 == 0);
    static const bool no_mfma = dae_exp_env("DAE_TITLE_WAVE") != nullptr;              // A/B against the fmaf-chain kernel
    if (p_max >= 1 && p_max <= 32 && fs_max <= L && even_k && !generic && !no_mfma) {
        // positions 0 .. 31 are computed for every size: rows up to 31 + fs_max - 1 of the LDS image are read
        const int lpad = 32 + fs_max;
        const size_t lds = (size_t)lpad * E * sizeof(float);
        hipLaunchKernelGGL((title_features_mfma_kernel<8>), dim3(B), dim3(512), lds, ctx->stream, p, lpad);
        DAE_CHECK_LAUNCH(ctx, "title_features_mfma_kernel");
        return DAE_OK;
    }
    if (p_max >= 1 && p_max <= 32 && fs_max <= L && !generic) {
        // positions up to PMAX - 1 + fs_max - 1 are read: pad the LDS image with zero rows
        const int pm = p_max <= 24 ? 24 : 32;
        const int lpad = pm + fs_max;
        const size_t lds = (size_t)lpad * E * sizeof(float);
        if (pm == 24) hipLaunchKernelGGL((title_features_wave_kernel<24, 8>), dim3(B), dim3(512), lds, ctx->stream, p, lpad);
        else hipLaunchKernelGGL((title_features_wave_kernel<32, 8>), dim3(B), dim3(512), lds, ctx->stream, p, lpad);
        DAE_CHECK_LAUNCH(ctx, "title_features_wave_kernel");
        return DAE_OK;
    }
    hipLaunchKernelGGL(title_features_kernel, dim3(B), dim3(256), (size_t)L * E * sizeof(float), ctx->stream, p);
    DAE_CHECK_LAUNCH(ctx, "title_features_kernel");
    return DAE_OK;
}

int dae_launch_mix_scores(dae_ctx* ctx, const float* title_score, int64_t ld_t, float* dae_score, int64_t ld_d,
                          const float* w_title, const float* w_playlist, int B, int ncols)
{
    int bx = (ncols + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(mix_scores_kernel, dim3(bx, B), dim3(256), 0, ctx->stream, title_score, ld_t, dae_score,
                       ld_d, w_title, w_playlist, B, ncols);
    DAE_CHECK_LAUNCH(ctx, "mix_scores_kernel");
    return DAE_OK;
}
