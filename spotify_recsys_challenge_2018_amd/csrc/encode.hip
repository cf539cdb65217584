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

// W_enc through a buffer descriptor: row offset (wave-uniform, from v_readlane) in the SCALAR offset
// operand, lane offset constant in the vector offset -- v_readlane + s_mul + buffer_load per
// non-zero instead of a 64-bit scalar multiply/add chain per global_load (measured: the gather
// was instruction-bound, 17 of 24 us at batch 256, with every read redirected to row 0 as slow).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t w_rsrc(const float* W, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float ld1(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float4 ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

struct EncP {
    const int32_t* row_ptr; const int32_t* col; const float* val;
    const float* W; const float* b_enc;
    int H, B;
    float ikp, kp; uint32_t seed;
    float* h_out;        // [B,H] row-major or null
    float* hp;           // packed [n_rg][G][RB][2][32][4] or null
    int G, RB;           // packed geometry (G = Hp/8, RB = R_TILE/32)
    unsigned short* hp16;   // bf16 image [n_rg][NS][RB][64][8] (decode_f32.hip pack_h_bf16_kernel's layout) or null:
    int NS;                 // the bf16 decode reads the hidden rows rounded (RNE) straight from the encode; NS = Hp/16
    float* sg_out;       // [B,H] sigmoid BEFORE hidden dropout (training backward) or null
    float* xhat_out;     // [nnz] normalised, dropped-out input weights (training backward) or null
    unsigned w_bytes;    // V * H * 4 (< 4 GiB: buffer descriptor range)
    int dbg_row0;        // experiment: read row 0 instead of the real rows
    int dbg_stop;        // experiment: leave the kernel after stage n
};

// issue the loads of one group of 16 non-zeros (indices base..base+15 of the current 64-chunk);
// indices past n re-read the chunk's last row (cache hit) and get weight 0: fmaf(0, w, acc) == acc.
#define ENC_LOAD(X, WS, BASE)                                                                  \
    _Pragma("unroll") for (int u = 0; u < ENC_GRP; ++u) {                                      \
        const int ii = (BASE) + u;             /* lanes >= n hold column 0 / weight 0 */       \
        const int c = __builtin_amdgcn_readlane(c_l, ii);                                      \
        WS[u] = rl_f(w_l, ii);                                                                 \
        X[u] = ld4(rs, voff, c * hbytes);                                                      \
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
    const int hbytes = H * 4;
    const __amdgpu_buffer_rsrc_t rs = w_rsrc(p.W, p.w_bytes);

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
            const int voff = (active ? hoff : 0) * 4;        // idle lanes read column 0 (unused)
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
                if (p.hp16) {
                    // k = hoff + e -> step s = k >> 4, lane half hi = (k >> 3) & 1, slot k & 7: 4 consecutive slots
                    const int R_TILE = p.RB * 32;
                    const int rg = row / R_TILE, rl = row - rg * R_TILE;
                    const int rb = rl >> 5, j = rl & 31;
                    const int st = hoff >> 4, hi2 = (hoff >> 3) & 1, e0 = hoff & 7;
                    unsigned short* d = p.hp16 + ((((size_t)rg * p.NS + st) * p.RB + rb) * 64 + hi2 * 32 + j) * 8 + e0;
                    *reinterpret_cast<uint2*>(d) = make_uint2(dae_bf16_rne(hv[0]) | (dae_bf16_rne(hv[1]) << 16),
                                                              dae_bf16_rne(hv[2]) | (dae_bf16_rne(hv[3]) << 16));
                }
            }
        }
    }
}

// Latency mode for small batches: a row is split over HS waves by hidden units (wave q owns units
// [q*H/HS, (q+1)*H/HS), one or more floats per lane), so HS times more row reads are in flight per
// playlist.  The per-unit fmaf chain over the non-zeros is unchanged -> same bits as encode_kernel.
// WGW = waves per workgroup (1: a workgroup per (row, quarter); HS: a workgroup per row, its waves the quarters -- the same waves, a
// quarter of the workgroups to dispatch: round 6's SQ counters showed waves living 6.4 us in a 12.3 us launch of 1 024 one-wave
// workgroups)
template <int HS, int WGW = 1>
__global__ __launch_bounds__(64 * WGW) void encode_split_kernel(const EncP p)
{
    const int lane = threadIdx.x & 63;
    const int unit = WGW == 1 ? (int)blockIdx.x : (int)blockIdx.x * WGW + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int row = unit / HS;
    const int q = unit % HS;
    const int H = p.H;
    const int hq = H / HS;                       // hidden units of this wave (multiple of 4)
    const int hbytes = H * 4;
    const __amdgpu_buffer_rsrc_t rs = w_rsrc(p.W, p.w_bytes);
    if (DAE_EXP_ON(p.dbg_stop == 1)) return;
    const int beg = p.row_ptr[row], end = p.row_ptr[row + 1];
    const int nnz = end - beg;
    if (DAE_EXP_ON(p.dbg_stop == 2)) { if (nnz == -7) p.h_out[0] = 0.f; return; }

    // Typical rows (<= 256 non-zeros: a playlist holds <= 250 items, spotify_reader.py:84) keep
    // their (column, value) entries in 4 registers per lane, fetched by 8 INDEPENDENT loads; the
    // chunk-by-chunk version paid one dependent round trip per 64 entries, twice.
    int cl[4];
    float xl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = beg + 64 * c + lane;
        const bool in = 64 * c + lane < nnz;
        cl[c] = in ? p.col[i] : 0;
        xl[c] = in ? p.val[i] : 0.0f;
    }
    if (p.ikp < 1.0f) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (64 * c + lane < nnz)
                xl[c] = (xl[c] / p.ikp) * floorf(p.ikp + dae_uniform(p.seed, 0U, (uint32_t)row, (uint32_t)cl[c]));
    }
    if (DAE_EXP_ON(p.dbg_stop == 3)) { if (xl[0] + xl[1] + xl[2] + xl[3] + cl[0] == -7.f) p.h_out[0] = 0.f; return; }
    // s = sum of the weights in column order (sequential: canonical order)
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = min(64, nnz - 64 * c);
        for (int i = 0; i < n; ++i) s += rl_f(xl[c], i);
    }
    for (int base = beg + 256; base < end; base += 64) {         // rows beyond 256 entries (rare)
        const int n = min(64, end - base);
        float x = 0.0f;
        if (lane < n) {
            x = p.val[base + lane];
            if (p.ikp < 1.0f)
                x = (x / p.ikp) * floorf(p.ikp + dae_uniform(p.seed, 0U, (uint32_t)row, (uint32_t)p.col[base + lane]));
        }
        for (int i = 0; i < n; ++i) s += rl_f(x, i);
    }
    const float denom = s + 1e-10f;
    if (DAE_EXP_ON(p.dbg_stop == 4)) { if (denom == -7.f) p.h_out[0] = 0.f; return; }
    float wl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        wl[c] = (64 * c + lane < nnz) ? xl[c] / denom : 0.0f;
        if (p.xhat_out && q == 0 && 64 * c + lane < nnz) p.xhat_out[beg + 64 * c + lane] = wl[c];
    }

// lanes past the row's end hold column 0 / weight 0, so no clamps or selects are needed; 16 slots
// at a time so that short rows do not pay for a whole chunk.  OFF = lane offset of the chunk inside its 64-entry register
#define ENC_ISSUE(X, CL, OFF, N, CK)                                                           \
        _Pragma("unroll") for (int k16 = 0; k16 < (CK) / 16; ++k16)                            \
            if (16 * k16 < (N)) {                                                              \
                _Pragma("unroll") for (int u = 16 * k16; u < 16 * k16 + 16; ++u)               \
                    X[u] = ld1(rs, voff, __builtin_amdgcn_readlane(CL, (OFF) + u) * hbytes);   \
            }
#define ENC_CHAIN(X, WL, OFF, N, CK)                                                           \
        _Pragma("unroll") for (int k16 = 0; k16 < (CK) / 16; ++k16)                            \
            if (16 * k16 < (N)) {                                                              \
                _Pragma("unroll") for (int u = 16 * k16; u < 16 * k16 + 16; ++u)               \
                    acc = fmaf(rl_f(WL, (OFF) + u), X[u], acc);                                \
            }

    for (int hb = 0; hb < hq; hb += 64) {
        const int hu = q * hq + hb + lane;       // this lane's hidden unit
        const bool active = hb + lane < hq;
        const int voff = (active ? hu : 0) * 4;
        float acc = 0.0f;
        // The row's first 256 entries sit in registers (cl / wl); their W rows are fetched in chunks of 32, double
        // buffered: chunk c + 1's 32 row reads are issued before chunk c's ordered fmaf chain runs, so 64 rows are in
        // flight.  (Chunks of 64 -- 128 rows in flight, 128 registers of buffers -- were not faster, and at 159 registers
        // a wave could not share a SIMD with the two filter waves of another batch's decode launch.)
        constexpr int CK = 32;
        float xa[CK], xb[CK];
        const int nch = (min(nnz, 256) + CK - 1) / CK;            // chunks held in registers (wave-uniform)
        // WHOLE chunks, issued unconditionally (round 5): the registers beyond the row's end hold column 0 / weight 0, so a full chunk
        // reads W row 0 (one cached line) for them and adds +0 -- as the partial groups of 16 always did.  With the per-group guards
        // and the "is there a next chunk" branch around the issue, hipcc could not count the loads in flight and every chain waited
        // with vmcnt(0): for the chunk it needed AND the one it had just requested -- a memory round trip per chunk instead of a
        // double buffer (ISA, profiles/r05_notes.md 15).  Now chunk c + 1 always goes out before chunk c's chain (the last of the
        // eight excepted, at compile time), and the wait in between is a count.
#define ENC_ISSUE_ALL(X, CL, OFF)                                                              \
        _Pragma("unroll") for (int u = 0; u < CK; ++u)                                         \
            X[u] = ld1(rs, voff, __builtin_amdgcn_readlane(CL, (OFF) + u) * hbytes);
#define ENC_CHAIN_ALL(X, WL, OFF)                                                              \
        _Pragma("unroll") for (int u = 0; u < CK; ++u)                                         \
            acc = fmaf(rl_f(WL, (OFF) + u), X[u], acc);
        if (nch > 0) { ENC_ISSUE_ALL(xa, cl[0], 0) }
#pragma unroll
        for (int c2 = 0; c2 < 256 / CK; ++c2) {
            if (c2 < nch) {
                if (c2 & 1) {
                    if (c2 + 1 < 256 / CK) { ENC_ISSUE_ALL(xa, cl[(((c2 + 1) * CK) >> 6) & 3], ((c2 + 1) * CK) & 63) }
                    ENC_CHAIN_ALL(xb, wl[(c2 * CK) >> 6], (c2 * CK) & 63)
                } else {
                    if (c2 + 1 < 256 / CK) { ENC_ISSUE_ALL(xb, cl[(((c2 + 1) * CK) >> 6) & 3], ((c2 + 1) * CK) & 63) }
                    ENC_CHAIN_ALL(xa, wl[(c2 * CK) >> 6], (c2 * CK) & 63)
                }
            }
        }
#undef ENC_ISSUE_ALL
#undef ENC_CHAIN_ALL
        for (int base = beg + 256; base < end; base += 64) {     // rare long tail, chunk by chunk
            const int n = min(64, end - base);
            int c_l = 0; float w_l = 0.0f;
            if (lane < n) {
                c_l = p.col[base + lane];
                float x = p.val[base + lane];
                if (p.ikp < 1.0f)
                    x = (x / p.ikp) * floorf(p.ikp + dae_uniform(p.seed, 0U, (uint32_t)row, (uint32_t)c_l));
                w_l = x / denom;
                if (p.xhat_out && q == 0 && hb == 0) p.xhat_out[base + lane] = w_l;
            }
            ENC_ISSUE(xa, c_l, 0, min(n, CK), CK)
            ENC_CHAIN(xa, w_l, 0, min(n, CK), CK)
            if (n > CK) {
                ENC_ISSUE(xb, c_l, CK, n - CK, CK)
                ENC_CHAIN(xb, w_l, CK, n - CK, CK)
            }
        }
        if (DAE_EXP_ON(p.dbg_stop == 5)) { if (acc == -7.f) p.h_out[0] = 0.f; return; }
        if (active) {
            float hv = dae_sigmoidf(acc + p.b_enc[hu]);
            if (p.sg_out) p.sg_out[(size_t)row * H + hu] = hv;
            if (p.kp < 1.0f) {
                const float u = dae_uniform(p.seed, 1U, (uint32_t)row, (uint32_t)hu);
                hv = (hv / p.kp) * floorf(p.kp + u);
            }
            if (p.h_out) p.h_out[(size_t)row * H + hu] = hv;
            if (p.hp) {
                const int R_TILE = p.RB * 32;
                const int rg = row / R_TILE, rl = row - rg * R_TILE;
                const int rb = rl >> 5, j = rl & 31;
                const int g = hu >> 3, e2 = (hu & 7) >> 1, hi2 = hu & 1;
                p.hp[((((size_t)rg * p.G + g) * p.RB + rb) * 64 + hi2 * 32 + j) * 4 + e2] = hv;
            }
            if (p.hp16) {
                const int R_TILE = p.RB * 32;
                const int rg = row / R_TILE, rl = row - rg * R_TILE;
                const int rb = rl >> 5, j = rl & 31;
                const int st = hu >> 4, hi2 = (hu >> 3) & 1, e0 = hu & 7;
                p.hp16[((((size_t)rg * p.NS + st) * p.RB + rb) * 64 + hi2 * 32 + j) * 8 + e0] =
                    (unsigned short)dae_bf16_rne(hv);
            }
        }
    }
#undef ENC_ISSUE
#undef ENC_CHAIN
}

}  // namespace

int dae_launch_encode(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val,
                      const float* W_enc, const float* b_enc, int V, int H, int B,
                      float ikp, float kp, uint32_t seed, float* h_out,
                      float* h_packed, int G, int RB, float* sg_out, float* xhat_out, unsigned short* h_packed16, int NS)
{
    if (B <= 0) return DAE_OK;
    EncP p;
    p.hp16 = h_packed16; p.NS = NS;
    p.row_ptr = row_ptr; p.col = col; p.val = val; p.W = W_enc; p.b_enc = b_enc;
    p.H = H; p.B = B; p.ikp = ikp; p.kp = kp; p.seed = seed;
    p.h_out = h_out; p.hp = h_packed; p.G = G; p.RB = RB;
    p.sg_out = sg_out; p.xhat_out = xhat_out;
    if ((size_t)V * H * 4 >= 0xFFFFFFFFull)
        return dae_fail(ctx, DAE_ERR_ARG, "W_enc of %d x %d exceeds the 4 GiB buffer range", V, H);
    p.w_bytes = (unsigned)((size_t)V * H * 4);
    static const int dbg_row0 = dae_exp_env("DAE_DBG_ENC_ROW0") ? atoi(dae_exp_env("DAE_DBG_ENC_ROW0")) : 0;
    p.dbg_row0 = dbg_row0;
    static const int dbg_stop = dae_exp_env("DAE_DBG_ENC_STOP") ? atoi(dae_exp_env("DAE_DBG_ENC_STOP")) : 0;
    p.dbg_stop = dbg_stop;
    if (B <= 1024 && (H % 16) == 0 && H >= 64) {
        // small batch: latency bound -> 4 waves per row (by hidden units), 4x the bytes in flight
        static const bool enc_wg1 = dae_exp_env("DAE_ENC_WG1") != nullptr;                     // A/B: one wave per workgroup
        if (enc_wg1) hipLaunchKernelGGL((encode_split_kernel<4, 1>), dim3(B * 4), dim3(64), 0, ctx->stream, p);
        else hipLaunchKernelGGL((encode_split_kernel<4, 4>), dim3(B), dim3(256), 0, ctx->stream, p);
    } else if (B <= 2048) {
        // few rows: one wave per workgroup spreads the rows over all CUs
        hipLaunchKernelGGL(encode_kernel<1>, dim3(B), dim3(64), 0, ctx->stream, p);
    } else {
        int blocks = (B + 3) / 4;
        if (blocks > DAE_NUM_CU * 8) blocks = DAE_NUM_CU * 8;    // grid-stride the rest
        hipLaunchKernelGGL(encode_kernel<4>, dim3(blocks), dim3(256), 0, ctx->stream, p);
    }
    DAE_CHECK_LAUNCH(ctx, "encode_kernel");
    return DAE_OK;
}
