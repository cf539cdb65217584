// encode.hip -- K1: sparse multi-hot playlist x encoder matrix (reference models/DAEs.py:40-42
// input dropout + row normalise, :64-70 encoder).  HBM-bandwidth bound row gather.
//
// One 64-lane wavefront owns one playlist row.  Lane l owns hidden units [4l, 4l+4) (+256 per
// extra pass when H > 256), so each non-zero is ONE fully coalesced 16 B/lane read of a W_enc row
// (1 KiB per wave instruction at H = 256) and the accumulation needs no cross-lane reduction at
// all: the per-lane fmaf chain over the row's non-zeros in ascending column order IS the canonical
// order of DESIGN.md, bit-identical to oracle/dae_oracle.c:orc_encode.  Column ids / weights of up
// to 64 non-zeros live one-per-lane in VGPRs and are broadcast with v_readlane (scalar), which
// makes the row base address scalar.  Loads are issued in groups of 16 rows, double buffered: up
// to 32 independent 1 KiB row reads (32 KiB) are in flight per wave, which is what hides HBM /
// Infinity-Cache latency at batch 256 where only 256 waves exist.
//
// Optionally the hidden row is also written in the MFMA B-operand order of decode_f32.hip
// (the fused dae_score_topk path), which removes the separate re-pack pass.
#include "dae_internal.h"

namespace {

constexpr int ENC_GRP = 16;

__device__ __forceinline__ float rl_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

struct EncP {
    const int32_t* row_ptr; const int32_t* col; const float* val;
    const float* W; const float* b_enc;
    int H, B;
    float ikp, kp; uint32_t seed;
    float* h_out;        // [B,H] row-major or null
    float* hp;           // packed [n_rg][G][RB][2][32][4] or null
    int G, RB;           // packed geometry (G = Hp/8, RB = R_TILE/32)
    float* sg_out;       // [B,H] sigmoid BEFORE hidden dropout (training backward) or null
    float* xhat_out;     // [nnz] normalised, dropped-out input weights (training backward) or null
};

// issue the loads of one group of 16 non-zeros (indices base..base+15 of the current 64-chunk);
// indices past n re-read the chunk's last row (cache hit) and get weight 0: fmaf(0, w, acc) == acc.
#define ENC_LOAD(X, WS, BASE)                                                                  \
    _Pragma("unroll") for (int u = 0; u < ENC_GRP; ++u) {                                      \
        const int ii = (BASE) + u;                                                             \
        const int ic = ii < n ? ii : n - 1;                                                    \
        const int c = __builtin_amdgcn_readlane(c_l, ic);                                      \
        const float w = rl_f(w_l, ic);                                                         \
        WS[u] = ii < n ? w : 0.0f;                                                             \
        X[u] = *reinterpret_cast<const float4*>(Wl + (size_t)c * H);                           \
    }
#define ENC_FMA(X, WS)                                                                         \
    _Pragma("unroll") for (int u = 0; u < ENC_GRP; ++u) {                                      \
        acc.x = fmaf(WS[u], X[u].x, acc.x);                                                    \
        acc.y = fmaf(WS[u], X[u].y, acc.y);                                                    \
        acc.z = fmaf(WS[u], X[u].z, acc.z);                                                    \
        acc.w = fmaf(WS[u], X[u].w, acc.w);                                                    \
    }

template <int NW>
__global__ __launch_bounds__(NW * 64) void encode_kernel(const EncP p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_total = gridDim.x * NW;
    const int H = p.H;

    for (int row = blockIdx.x * NW + wave; row < p.B; row += waves_total) {
        const int beg = p.row_ptr[row], end = p.row_ptr[row + 1];

        // ---- pass 1: s = sum of (dropped-out) weights, sequential in column order -------------
        float s = 0.0f;
        for (int base = beg; base < end; base += 64) {
            const int n = min(64, end - base);
            float x = 0.0f;
            if (lane < n) {
                x = p.val[base + lane];
                if (p.ikp < 1.0f) {
                    const float u = dae_uniform(p.seed, 0U, (uint32_t)row, (uint32_t)p.col[base + lane]);
                    x = (x / p.ikp) * floorf(p.ikp + u);
                }
            }
            for (int i = 0; i < n; ++i) s += rl_f(x, i);
        }
        const float denom = s + 1e-10f;

        // ---- pass 2: gather, one hidden pass of 256 units at a time ---------------------------
        for (int hbase = 0; hbase < H; hbase += 256) {
            const int hoff = hbase + lane * 4;
            const bool active = hoff < H;
            const float* Wl = p.W + (active ? hoff : 0);     // idle lanes read column 0 (unused)
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

            for (int base = beg; base < end; base += 64) {
                const int n = min(64, end - base);           // >= 1
                int c_l = 0;
                float w_l = 0.0f;
                if (lane < n) {
                    c_l = p.col[base + lane];
                    float x = p.val[base + lane];
                    if (p.ikp < 1.0f) {
                        const float u = dae_uniform(p.seed, 0U, (uint32_t)row, (uint32_t)c_l);
                        x = (x / p.ikp) * floorf(p.ikp + u);
                    }
                    w_l = x / denom;
                    if (p.xhat_out && hbase == 0) p.xhat_out[base + lane] = w_l;
                }
                float4 xa[ENC_GRP], xb[ENC_GRP];
                float wa[ENC_GRP], wb[ENC_GRP];
                ENC_LOAD(xa, wa, 0)
                if (n > 16) { ENC_LOAD(xb, wb, 16) }
                ENC_FMA(xa, wa)
                if (n > 16) {
                    if (n > 32) { ENC_LOAD(xa, wa, 32) }
                    ENC_FMA(xb, wb)
                    if (n > 32) {
                        if (n > 48) { ENC_LOAD(xb, wb, 48) }
                        ENC_FMA(xa, wa)
                        if (n > 48) { ENC_FMA(xb, wb) }
                    }
                }
            }

            if (active) {
                const float4 be = *reinterpret_cast<const float4*>(p.b_enc + hoff);
                float hv[4] = {dae_sigmoidf(acc.x + be.x), dae_sigmoidf(acc.y + be.y),
                               dae_sigmoidf(acc.z + be.z), dae_sigmoidf(acc.w + be.w)};
                if (p.sg_out)
                    *reinterpret_cast<float4*>(p.sg_out + (size_t)row * H + hoff) =
                        make_float4(hv[0], hv[1], hv[2], hv[3]);
                if (p.kp < 1.0f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = dae_uniform(p.seed, 1U, (uint32_t)row, (uint32_t)(hoff + e));
                        hv[e] = (hv[e] / p.kp) * floorf(p.kp + u);
                    }
                }
                if (p.h_out)
                    *reinterpret_cast<float4*>(p.h_out + (size_t)row * H + hoff) =
                        make_float4(hv[0], hv[1], hv[2], hv[3]);
                if (p.hp) {
                    // k = hoff + e  ->  group g = k >> 3, slot (e2 = (k & 7) >> 1, hi = k & 1)
                    const int R_TILE = p.RB * 32;
                    const int rg = row / R_TILE, rl = row - rg * R_TILE;
                    const int rb = rl >> 5, j = rl & 31;
                    const int g = hoff >> 3;
                    const int e2 = (hoff & 7) >> 1;                       // 0 or 2
                    float* base4 = p.hp + ((((size_t)rg * p.G + g) * p.RB + rb) * 64 + j) * 4;
                    base4[e2] = hv[0];                 // hi = 0, slot e2
                    base4[32 * 4 + e2] = hv[1];        // hi = 1, slot e2
                    base4[e2 + 1] = hv[2];             // hi = 0, slot e2 + 1
                    base4[32 * 4 + e2 + 1] = hv[3];    // hi = 1, slot e2 + 1
                }
            }
        }
    }
}

}  // namespace

int dae_launch_encode(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val,
                      const float* W_enc, const float* b_enc, int V, int H, int B,
                      float ikp, float kp, uint32_t seed, float* h_out,
                      float* h_packed, int G, int RB, float* sg_out, float* xhat_out)
{
    (void)V;
    if (B <= 0) return DAE_OK;
    EncP p;
    p.row_ptr = row_ptr; p.col = col; p.val = val; p.W = W_enc; p.b_enc = b_enc;
    p.H = H; p.B = B; p.ikp = ikp; p.kp = kp; p.seed = seed;
    p.h_out = h_out; p.hp = h_packed; p.G = G; p.RB = RB;
    p.sg_out = sg_out; p.xhat_out = xhat_out;
    if (B <= 2048) {
        // few rows: one wave per workgroup spreads the rows over all CUs (latency bound)
        hipLaunchKernelGGL(encode_kernel<1>, dim3(B), dim3(64), 0, ctx->stream, p);
    } else {
        int blocks = (B + 3) / 4;
        if (blocks > DAE_NUM_CU * 8) blocks = DAE_NUM_CU * 8;    // grid-stride the rest
        hipLaunchKernelGGL(encode_kernel<4>, dim3(blocks), dim3(256), 0, ctx->stream, p);
    }
    DAE_CHECK_LAUNCH(ctx, "encode_kernel");
    return DAE_OK;
}
