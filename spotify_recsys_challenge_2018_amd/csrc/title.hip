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

constexpr int T_MAX_SIZES = 8;
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

}  // namespace

int dae_launch_title_features(dae_ctx* ctx, const int32_t* titles, int B, int L, const float* emb, int n_char,
                              int E, const float* conv_w, const float* conv_b, const int32_t* filter_sizes,
                              int n_sizes, int F, float kp, uint32_t seed, float* feat, int64_t ld,
                              int32_t* argmax, float* feat_raw)
{
    if (n_sizes < 1 || n_sizes > T_MAX_SIZES) return dae_fail(ctx, DAE_ERR_ARG, "n_sizes=%d out of [1,%d]", n_sizes, T_MAX_SIZES);
    if (L < 1 || L > T_MAX_LEN || E < 1 || E > T_MAX_EMB) return dae_fail(ctx, DAE_ERR_ARG, "title length %d / embedding %d too large", L, E);
    if (ld < (int64_t)n_sizes * F) return dae_fail(ctx, DAE_ERR_ARG, "ld=%lld < %d features", (long long)ld, n_sizes * F);
    TitleP p;
    memset(&p, 0, sizeof(p));
    p.titles = titles; p.B = B; p.L = L; p.emb = emb; p.n_char = n_char; p.E = E;
    p.conv_w = conv_w; p.conv_b = conv_b; p.n_sizes = n_sizes; p.F = F; p.kp = kp; p.seed = seed;
    p.feat = feat; p.ld = ld; p.argmax = argmax; p.feat_raw = feat_raw;
    int off = 0;
    for (int i = 0; i < n_sizes; ++i) {
        if (filter_sizes[i] < 1 || filter_sizes[i] > L) return dae_fail(ctx, DAE_ERR_ARG, "filter size %d outside [1,%d]", filter_sizes[i], L);
        p.fs[i] = filter_sizes[i]; p.w_off[i] = off;
        off += filter_sizes[i] * E * F;
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
