// encode.hip -- K1: sparse multi-hot playlist x encoder matrix (reference models/DAEs.py:40-42
// input dropout + row normalise, :64-70 encoder).  HBM-bandwidth bound row gather.
//
// One 64-lane wavefront owns one playlist row.  Lane l owns hidden units [4l, 4l+4) (+256 per
// extra pass when H > 256), so each non-zero is ONE fully coalesced 16 B/lane read of a W_enc row
// (1 KiB per wave instruction at H = 256) and the accumulation needs no cross-lane reduction at
// all: the per-lane fmaf chain over the row's non-zeros in ascending column order IS the canonical
// order of DESIGN.md, bit-identical to oracle/dae_oracle.c:orc_encode.  Column ids / weights of up
// to 64 non-zeros live one-per-lane in VGPRs and are broadcast with v_readlane (scalar), which
// makes the row base address scalar and leaves 8 independent 1 KiB loads in flight per wave.
#include "dae_internal.h"

namespace {

constexpr int ENC_UNROLL = 8;

__device__ __forceinline__ float rl_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__global__ __launch_bounds__(256) void encode_kernel(
    const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ W, const float* __restrict__ b_enc,
    int H, int B, float ikp, float kp, uint32_t seed, float* __restrict__ h_out)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_total = gridDim.x * 4;

    for (int row = blockIdx.x * 4 + wave; row < B; row += waves_total) {
        const int beg = row_ptr[row], end = row_ptr[row + 1];

        // ---- pass 1: s = sum of (dropped-out) weights, sequential in column order -------------
        float s = 0.0f;
        for (int base = beg; base < end; base += 64) {
            const int n = min(64, end - base);
            float x = 0.0f;
            if (lane < n) {
                x = val[base + lane];
                if (ikp < 1.0f) {
                    const float u = dae_uniform(seed, 0U, (uint32_t)row, (uint32_t)col[base + lane]);
                    x = (x / ikp) * floorf(ikp + u);
                }
            }
            for (int i = 0; i < n; ++i) s += rl_f(x, i);
        }
        const float denom = s + 1e-10f;

        // ---- pass 2: gather, one hidden pass of 256 units at a time ---------------------------
        for (int hbase = 0; hbase < H; hbase += 256) {
            const int hoff = hbase + lane * 4;
            const bool active = hoff < H;
            const float* Wl = W + (active ? hoff : 0);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

            for (int base = beg; base < end; base += 64) {
                const int n = min(64, end - base);
                int c_l = 0;
                float w_l = 0.0f;
                if (lane < n) {
                    c_l = col[base + lane];
                    float x = val[base + lane];
                    if (ikp < 1.0f) {
                        const float u = dae_uniform(seed, 0U, (uint32_t)row, (uint32_t)c_l);
                        x = (x / ikp) * floorf(ikp + u);
                    }
                    w_l = x / denom;
                }
                int i = 0;
                for (; i + ENC_UNROLL <= n; i += ENC_UNROLL) {
                    float4 wv[ENC_UNROLL];
                    float ws[ENC_UNROLL];
#pragma unroll
                    for (int u = 0; u < ENC_UNROLL; ++u) {
                        const int c = __builtin_amdgcn_readlane(c_l, i + u);
                        ws[u] = rl_f(w_l, i + u);
                        wv[u] = active ? *reinterpret_cast<const float4*>(Wl + (size_t)c * H)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < ENC_UNROLL; ++u) {
                        acc.x = fmaf(ws[u], wv[u].x, acc.x);
                        acc.y = fmaf(ws[u], wv[u].y, acc.y);
                        acc.z = fmaf(ws[u], wv[u].z, acc.z);
                        acc.w = fmaf(ws[u], wv[u].w, acc.w);
                    }
                }
                for (; i < n; ++i) {
                    const int c = __builtin_amdgcn_readlane(c_l, i);
                    const float w = rl_f(w_l, i);
                    if (active) {
                        const float4 wv = *reinterpret_cast<const float4*>(Wl + (size_t)c * H);
                        acc.x = fmaf(w, wv.x, acc.x);
                        acc.y = fmaf(w, wv.y, acc.y);
                        acc.z = fmaf(w, wv.z, acc.z);
                        acc.w = fmaf(w, wv.w, acc.w);
                    }
                }
            }

            if (active) {
                const float4 be = *reinterpret_cast<const float4*>(b_enc + hoff);
                float hv[4] = {dae_sigmoidf(acc.x + be.x), dae_sigmoidf(acc.y + be.y),
                               dae_sigmoidf(acc.z + be.z), dae_sigmoidf(acc.w + be.w)};
                if (kp < 1.0f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = dae_uniform(seed, 1U, (uint32_t)row, (uint32_t)(hoff + e));
                        hv[e] = (hv[e] / kp) * floorf(kp + u);
                    }
                }
                *reinterpret_cast<float4*>(h_out + (size_t)row * H + hoff) =
                    make_float4(hv[0], hv[1], hv[2], hv[3]);
            }
        }
    }
}

}  // namespace

int dae_launch_encode(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val,
                      const float* W_enc, const float* b_enc, int V, int H, int B,
                      float ikp, float kp, uint32_t seed, float* h_out)
{
    (void)V;
    if (B <= 0) return DAE_OK;
    // one wave per row; 4 waves per block; cap the grid and stride the rest (guide G11)
    int blocks = (B + 3) / 4;
    if (blocks > DAE_NUM_CU * 8) blocks = DAE_NUM_CU * 8;
    hipLaunchKernelGGL(encode_kernel, dim3(blocks), dim3(256), 0, ctx->stream,
                       row_ptr, col, val, W_enc, b_enc, H, B, ikp, kp, seed, h_out);
    DAE_CHECK_LAUNCH(ctx, "encode_kernel");
    return DAE_OK;
}
