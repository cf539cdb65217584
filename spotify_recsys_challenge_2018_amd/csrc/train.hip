// train.hip -- the training step of the reference graph (models/DAEs.py:98-102; SURVEY.md 8a
// rows a8-a10, App. B.5), fp32 on the CDNA4 matrix cores (bf16 operands for the three GEMMs under
// dae_set_train_dtype, BASELINE.json configs[3]):
//
//   forward   K1 encode with dropout (encode.hip) -> K5 decode GEMM whose epilogue turns logits
//             into the weighted-BCE loss and dL/dz, every element taken as a negative
//             (decode_f32.hip EPI_LOSS), writing dz^T [V,B]; loss_fixup_kernel redoes the positives
//             of the target CSR (no dense target matrix)
//   K6        gW_dec[v,:] = sum_r dz[r,v] h[r,:]   (+ gb_dec = column sums)      contraction B
//   K7        dh[r,:]     = sum_v dz[r,v] W_dec[v,:]  split over V, partials reduced   contraction V
//   K8        dpre = dh * dropout-mask/kp * s(1-s); gb_enc; gW_enc[c,:] += xhat[r,c] dpre[r,:]
//   K9        TF1 Adam, dense (moments decay on zero-gradient rows too)
//
// Both backward GEMMs use v_mfma_f32_32x32x2_f32 with D[i = hidden unit][j = v or r]; operands are
// read in their natural row-major layouts because the contraction index is the slow dimension of
// both, and the 4 (2) tiles a wave owns along i (j) are interleaved (hidden = hc0 + 4 i + a) so
// that one float4 (float2) per lane feeds 4 (2) MFMAs.  Summation orders differ from the oracle's
// float64 reference: parity is by tolerance (tests/test_gpu_train.py), not bitwise.
#include "dae_internal.h"

namespace {

// streamed-once 16-byte accesses (Adam state: every byte is read and written exactly once per step)
typedef float nt4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_ld4(const float* a)
{
    const nt4_t t = __builtin_nontemporal_load(reinterpret_cast<const nt4_t*>(a));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nt_st4(float* a, const float4 x)
{
    const nt4_t t = {x.x, x.y, x.z, x.w};
    __builtin_nontemporal_store(t, reinterpret_cast<nt4_t*>(a));
}


typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int N> struct IntC { static constexpr int value = N; };
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ unsigned pk_bf16(float a, float b)
{
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// eight floats -> the 8 k-slots a lane holds of one 32x32x16 bf16 MFMA operand
__device__ __forceinline__ bf16x8_t pk_bf16x8(float a0, float a1, float a2, float a3, float a4, float a5, float a6,
                                              float a7)
{
    return __builtin_bit_cast(bf16x8_t, make_uint4(pk_bf16(a0, a1), pk_bf16(a2, a3), pk_bf16(a4, a5), pk_bf16(a6, a7)));
}

// ---- the positives of the loss -----------------------------------------------------------------------
// K5 (decode_f32.hip, EPI_LOSS) treats all B x V elements as negatives.  A batch holds ~100 positives per row
// out of 170 000 columns, so instead of a dense [B, V] target matrix (174 MB zeroed, scattered into and read
// back per step) each target entry (row, col, y) is redone here: the same logit -- the fmaf chain over
// k = 0..H-1 from +0, then + bias, which is what the fp32 MFMA computes -- then the full loss term and
// dL/dz of DAEs.py:98-99; dL/dz OVERWRITES K5's value and the loss partial holds L(y) - L(0).
// One workgroup per row (h row in LDS), one thread per target entry.  The target CSR holds one entry per
// (row, col) (include/dae_hip.h: the CSR contract), so no two threads own the same element.
constexpr int FIX_MAXH = 1024;
// value of the bf16 nearest (ties to even) to f, as the prepack / pack_h kernels round the MFMA operands
__device__ __forceinline__ float bf16_value(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xFFFF0000u);
}
// BF16: the forward GEMM ran on bf16 operands (dae_set_train_dtype): W and h are rounded the same way here
// DZ16: dL/dz is kept as bf16 (the bf16 backward GEMMs read it as such)
// CORR (with BF16 and DZ16: the forward launch has already folded dh = dz W_dec into itself with every element a negative):
// the row's dh correction sum_i (bf16(dz_i) - bf16(dz_i as a negative)) bf16(W_dec[v_i]) goes out as one more partial
// (corr_out[row][k]); the value the forward launch stored is read back before it is overwritten, so the difference is exact.
template <bool BF16, bool DZ16 = false, bool CORR = false>
__global__ __launch_bounds__(256) void loss_fixup_kernel(const int32_t* __restrict__ row_ptr,
                                                         const int32_t* __restrict__ col,
                                                         const float* __restrict__ val, int B, int H,
                                                         int col_lo, int col_hi,
                                                         const float* __restrict__ h,      // [B, H] after dropout
                                                         const float* __restrict__ Wd,     // [col_hi - col_lo, H]
                                                         const float* __restrict__ bias,   // local column index
                                                         float inv_nb, float* __restrict__ dzT, int64_t ldT,
                                                         float* __restrict__ loss_part, float* __restrict__ corr_out = nullptr)
{
    __shared__ float4 sh[FIX_MAXH / 4];
    __shared__ float wsum[4];
    constexpr int CCAP = CORR ? 1024 : 1;
    __shared__ float cdel[CCAP];
    __shared__ int ccol[CCAP];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int H4 = H >> 2;
    for (int i = tid; i < H4; i += 256) {
        float4 v = reinterpret_cast<const float4*>(h + (size_t)row * H)[i];
        if (BF16) v = make_float4(bf16_value(v.x), bf16_value(v.y), bf16_value(v.z), bf16_value(v.w));
        sh[i] = v;
    }
    __syncthreads();
    float corr = 0.0f;
    for (int i = row_ptr[row] + tid; i < row_ptr[row + 1]; i += 256) {
        const int c = col[i];
        if (c < col_lo || c >= col_hi) continue;
        const int lc = c - col_lo;
        const float y = val[i];
        const float4* w = reinterpret_cast<const float4*>(Wd + (size_t)lc * H);
        float z = 0.0f;
        int k = 0;
        for (; k + 16 <= H4; k += 16) {                        // (16 loads in flight: the same chain, half the round trips)
            float4 wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = w[k + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (BF16) wv[u] = make_float4(bf16_value(wv[u].x), bf16_value(wv[u].y), bf16_value(wv[u].z), bf16_value(wv[u].w));
                const float4 hv = sh[k + u];
                z = fmaf(wv[u].x, hv.x, z); z = fmaf(wv[u].y, hv.y, z);
                z = fmaf(wv[u].z, hv.z, z); z = fmaf(wv[u].w, hv.w, z);
            }
        }
        for (; k + 8 <= H4; k += 8) {
            float4 wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w[k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (BF16) wv[u] = make_float4(bf16_value(wv[u].x), bf16_value(wv[u].y), bf16_value(wv[u].z), bf16_value(wv[u].w));
                const float4 hv = sh[k + u];
                z = fmaf(wv[u].x, hv.x, z); z = fmaf(wv[u].y, hv.y, z);
                z = fmaf(wv[u].z, hv.z, z); z = fmaf(wv[u].w, hv.w, z);
            }
        }
        for (; k < H4; ++k) {
            float4 wv = w[k];
            const float4 hv = sh[k];
            if (BF16) wv = make_float4(bf16_value(wv.x), bf16_value(wv.y), bf16_value(wv.z), bf16_value(wv.w));
            z = fmaf(wv.x, hv.x, z); z = fmaf(wv.y, hv.y, z); z = fmaf(wv.z, hv.z, z); z = fmaf(wv.w, hv.w, z);
        }
        z += bias[lc];
        const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * z));
        const float a1 = pr + 1e-10f, a0 = 1.0f - pr + 1e-10f;
        const float l1 = __builtin_amdgcn_logf(a1), l0 = __builtin_amdgcn_logf(a0);
        // L(y) - L(0) = -ln2 * y * (log2 a1 - 0.55 log2 a0)
        corr -= 0.69314718f * y * (l1 - 0.55f * l0);
        const float dzv = -(y * __builtin_amdgcn_rcpf(a1) - 0.55f * (1.0f - y) * __builtin_amdgcn_rcpf(a0)) *
                          pr * (1.0f - pr) * inv_nb;
        if (DZ16) {
            unsigned short* dst = reinterpret_cast<unsigned short*>(dzT) + (size_t)lc * ldT + row;
            const unsigned short nw = (unsigned short)(pk_bf16(dzv, 0.0f) & 0xFFFFu);
            if (CORR) {
                const int e = i - row_ptr[row];
                if (e < CCAP) { cdel[e] = __uint_as_float((unsigned)nw << 16) - __uint_as_float((unsigned)*dst << 16); ccol[e] = lc; }
            }
            *dst = nw;
        } else dzT[(size_t)lc * ldT + row] = dzv;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) corr += __shfl_xor(corr, d);
    if ((tid & 63) == 0) wsum[tid >> 6] = corr;
    __syncthreads();
    if (tid == 0) loss_part[row] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * inv_nb;
    if (CORR) {
        // entries outside [col_lo, col_hi) wrote nothing: mark them (a second sweep over the row's entries, in entry order)
        // (a row holds at most CCAP = 1 024 targets here -- a playlist's <= 250 tracks and their artists, spotify_reader.py:84 --;
        // a longer one poisons its dh row with NaN rather than dropping entries silently)
        const bool too_long = row_ptr[row + 1] - row_ptr[row] > CCAP;
        const int beg = row_ptr[row], n = min(row_ptr[row + 1] - beg, CCAP);
        __syncthreads();
        for (int e = tid; e < n; e += 256) { const int c = col[beg + e]; if (c < col_lo || c >= col_hi) { cdel[e] = 0.0f; ccol[e] = 0; } }
        __syncthreads();
        // wave g takes the entries g, g + 4, ... (a lane = four hidden units: plain 1 KB row reads, 8 in flight), the four partial
        // sums meet in LDS and are added in wave order: a fixed order, whatever the timing
        float4* const cacc = sh;                                    // (the hidden row is no longer needed: [4][H / 4] float4 fit)
        __syncthreads();
        const int wv_ = tid >> 6, ln_ = tid & 63;
        for (int k4 = ln_; k4 < (H >> 2); k4 += 64) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            int e = wv_;
            for (; e + 28 < n; e += 32) {
                float4 wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = reinterpret_cast<const float4*>(Wd + (size_t)ccol[e + 4 * u] * H)[k4];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float d = cdel[e + 4 * u];
                    a.x = fmaf(d, bf16_value(wv[u].x), a.x); a.y = fmaf(d, bf16_value(wv[u].y), a.y);
                    a.z = fmaf(d, bf16_value(wv[u].z), a.z); a.w = fmaf(d, bf16_value(wv[u].w), a.w);
                }
            }
            for (; e < n; e += 4) {
                const float4 w1 = reinterpret_cast<const float4*>(Wd + (size_t)ccol[e] * H)[k4];
                const float d = cdel[e];
                a.x = fmaf(d, bf16_value(w1.x), a.x); a.y = fmaf(d, bf16_value(w1.y), a.y);
                a.z = fmaf(d, bf16_value(w1.z), a.z); a.w = fmaf(d, bf16_value(w1.w), a.w);
            }
            cacc[wv_ * (FIX_MAXH / 16) + k4] = a;
        }
        __syncthreads();
        for (int k4 = tid; k4 < (H >> 2); k4 += 256) {
            const float4 p0 = cacc[k4], p1 = cacc[(FIX_MAXH / 16) + k4], p2 = cacc[2 * (FIX_MAXH / 16) + k4], p3 = cacc[3 * (FIX_MAXH / 16) + k4];
            reinterpret_cast<float4*>(corr_out + (size_t)row * H)[k4] =
                too_long ? make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""))
                         : make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                                       ((p0.w + p1.w) + p2.w) + p3.w);
        }
    }
}

// ---- K6: gW[v, hc] (+)= sum_r dz[r, v] * h[r, hc];  gb[v] = sum_r dz[r, v] ----------------------
// block = 4 waves sharing the LDS image of h[:, hc0 : hc0+128] (K = B <= 256 rows); a wave owns
// tiles of 64 vocabulary columns (2 MFMA tiles, v = v0 + 2 j + b) x 128 hidden units (4 tiles).
struct GwP {
    const float* dzT; int64_t ldT;    // [V, ldT] (dz transposed, rows zero padded to ldT)
    const float* h; int H, B, V;
    float* gW;                        // [V, H]
    float* gb;                        // [V] (written by the hc0 == 0 blocks) or null
    int accumulate;                   // gW += instead of =
    int dbg;                          // experiments: 1 = no stores, 2 = no dz^T loads
    int n_half, nb_half;              // H / 128 hidden halves, blocks per half
    // dense TF1-Adam of the [V, H] tensor `ad_p` applied in the epilogue instead of writing gW (dae_arm_decoder_adam)
    float* ad_p; float* ad_m; float* ad_v; float ad_alpha, ad_b1, ad_b2, ad_eps;
};

// NA = hidden tiles per wave (4, 2 or 1): a "half" is 32*NA hidden units, hidden = hc0 + NA*i + a
// BF16 (dae_set_train_dtype, NA = 4 only): the 8 k-steps of a GW_MMA group (16 playlists) become ONE
// v_mfma_f32_32x32x16_bf16 per accumulator -- k-slot x of lane half hi is playlist R0 + 2x + hi in both
// operands, which is exactly what the fp32 steps consume one at a time -- on operands rounded to bf16 in
// registers; loads, LDS image, accumulators and stores are the fp32 kernel's.
// TR (fp32, NA = 4): the two MFMA operands swapped -- D[i = vocabulary row of the lane pair][j = hidden lane] instead of
// D[i = hidden][j = vocabulary row].  Loads, LDS image and column sums are unchanged; what changes is that a lane of the
// accumulators is a hidden unit (hc0 + 4 j + a), so the epilogue writes 512 contiguous bytes of one gW row per half-wave
// instead of 16-byte pieces of 32 rows -- the same shape the transposed bf16 kernel (grad_wdec_t_kernel) has.
template <int NA, int NW = 4, bool BF16 = false, bool DZ16 = false, bool TR = false>
__global__ __launch_bounds__(NW * 64, 1) void grad_wdec_kernel(const GwP p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [Bp][32*NA] floats
    constexpr int HW = 32 * NA;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gs = DAE_NUM_XCD * p.n_half;
    const int q = blockIdx.x / gs, rem = blockIdx.x % gs;
    const int half = rem / DAE_NUM_XCD;
    const int bir = q * DAE_NUM_XCD + (rem % DAE_NUM_XCD);
    const int hc0 = half * HW;
    const int Bp = (p.B + 31) & ~31;           // rows padded to whole 32-row groups (zero rows)

    // LDS image of h[:, hc0 : hc0 + HW]: 8 independent 16-byte loads in flight per thread.  (One 4-byte load ->
    // wait -> ds_write per iteration, 128 iterations per thread, was ~80 us of this kernel's 280: every iteration
    // pays an L2 round trip.)
    if ((reinterpret_cast<uintptr_t>(p.h) & 15) == 0 && (p.H & 3) == 0) {
        constexpr int HW4 = HW / 4, NT = NW * 64;
        const int n4 = Bp * HW4;
        float4* lds4 = reinterpret_cast<float4*>(lds);
        for (int i0 = tid; i0 < n4; i0 += 8 * NT) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * NT, n4 - 1);
                const int r = i / HW4, c4 = i - r * HW4;
                v[u] = *reinterpret_cast<const float4*>(p.h + (size_t)min(r, p.B - 1) * p.H + hc0 + 4 * c4);
                if (r >= p.B) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * NT;
                if (i < n4) lds4[i] = v[u];
            }
        }
    } else {
        for (int i = tid; i < Bp * HW; i += NW * 64) {
            const int r = i / HW, c = i - r * HW;
            lds[i] = r < p.B ? p.h[(size_t)r * p.H + hc0 + c] : 0.0f;
        }
    }
    __syncthreads();

    const int n_tiles = (p.V + 63) / 64;
    const int n_ws = p.nb_half * NW;
    for (int t = bir * NW + wave; t < n_tiles; t += n_ws) {
        const int v0 = t * 64;
        const int vcol = v0 + 2 * j;                                 // this lane's 2 columns
        const bool ok0 = vcol < p.V, ok1 = vcol + 1 < p.V;
        f32x16 acc[NA][2];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;
        float cs0 = 0.f, cs1 = 0.f;

        // B operand from dz^T [V][ldT]: the tile's 64 columns x B rows are ONE contiguous 64 KiB
        // block there (a strip of row-major dz is 256-byte pieces at a 4*V-byte stride: every piece
        // another DRAM page and TLB entry -- 584 us measured).  A lane owns columns vcol, vcol+1 =
        // two rows of dz^T; one float4 per row carries 4 consecutive playlists = 2 k-steps
        // (playlist 4q + 2*step + hi).
        const float* t0p = p.dzT + (size_t)(ok0 ? vcol : 0) * p.ldT;
        const float* t1p = p.dzT + (size_t)(ok1 ? vcol + 1 : 0) * p.ldT;
// unconditional loads (conditional writes to these arrays sent them to scratch memory): columns
// past V read row 0 and accumulate values that are never stored; the prefetch issued in the last
// iteration re-reads the last group
#define GW_LOAD(T0, T1, R0)                                                                    \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                     \
            const int r4 = min((R0) + 4 * q_, Bp4 - 4);                                        \
            T0[q_] = *reinterpret_cast<const float4*>(t0p + r4);                               \
            T1[q_] = *reinterpret_cast<const float4*>(t1p + r4);                               \
        }
#define GW_STEP(DX, DY, R)                                                                     \
        {                                                                                      \
            const float* ap = lds + (size_t)((R) + hi) * HW + NA * j;                          \
            float av[NA];                                                                      \
            if (NA == 4) {                                                                     \
                const float4 t4 = *reinterpret_cast<const float4*>(ap);                        \
                av[0] = t4.x; av[1 % NA] = t4.y; av[2 % NA] = t4.z; av[3 % NA] = t4.w;         \
            } else if (NA == 2) {                                                              \
                const float2 t2 = *reinterpret_cast<const float2*>(ap);                        \
                av[0] = t2.x; av[1 % NA] = t2.y;                                               \
            } else {                                                                           \
                av[0] = ap[0];                                                                 \
            }                                                                                  \
            cs0 += (DX); cs1 += (DY);                                                          \
            _Pragma("unroll") for (int a = 0; a < NA; ++a)                                     \
                acc[a][0] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32((DX), av[a], acc[a][0], 0, 0, 0) \
                               : __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], (DX), acc[a][0], 0, 0, 0); \
            _Pragma("unroll") for (int a = 0; a < NA; ++a)                                     \
                acc[a][1] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32((DY), av[a], acc[a][1], 0, 0, 0) \
                               : __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], (DY), acc[a][1], 0, 0, 0); \
        }
// the upper half-wave takes the odd playlist.  A bit blend (v_bfi), NOT `hi ? t.y : t.x`: the
// optimizer turns that into a dynamically indexed vector extract, which lives in scratch memory.
#define GW_SEL(A, Bv) __uint_as_float((__float_as_uint(Bv) & himask) | (__float_as_uint(A) & ~himask))
#define GW_MMA(T0, T1, R0)                                                                     \
        if (BF16) {                                                                            \
            float avs[8][NA];                                                                  \
            _Pragma("unroll") for (int x_ = 0; x_ < 8; ++x_) {                                 \
                const float4 t4 = *reinterpret_cast<const float4*>(lds + (size_t)((R0) + 2 * x_ + hi) * HW + NA * j); \
                avs[x_][0] = t4.x; avs[x_][1 % NA] = t4.y; avs[x_][2 % NA] = t4.z; avs[x_][3 % NA] = t4.w; \
            }                                                                                  \
            float dx[8], dy[8];                                                                \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                 \
                dx[2 * q_] = GW_SEL(T0[q_].x, T0[q_].y); dx[2 * q_ + 1] = GW_SEL(T0[q_].z, T0[q_].w); \
                dy[2 * q_] = GW_SEL(T1[q_].x, T1[q_].y); dy[2 * q_ + 1] = GW_SEL(T1[q_].z, T1[q_].w); \
            }                                                                                  \
            _Pragma("unroll") for (int x_ = 0; x_ < 8; ++x_) { cs0 += dx[x_]; cs1 += dy[x_]; } \
            const bf16x8_t bx = pk_bf16x8(dx[0], dx[1], dx[2], dx[3], dx[4], dx[5], dx[6], dx[7]); \
            const bf16x8_t by = pk_bf16x8(dy[0], dy[1], dy[2], dy[3], dy[4], dy[5], dy[6], dy[7]); \
            _Pragma("unroll") for (int a = 0; a < NA; ++a) {                                   \
                const bf16x8_t af = pk_bf16x8(avs[0][a], avs[1][a], avs[2][a], avs[3][a], avs[4][a], avs[5][a], \
                                              avs[6][a], avs[7][a]);                           \
                acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bx, acc[a][0], 0, 0, 0); \
                acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, by, acc[a][1], 0, 0, 0); \
            }                                                                                  \
        } else                                                                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                     \
            const int r4 = (R0) + 4 * q_;                                                      \
            GW_STEP(GW_SEL(T0[q_].x, T0[q_].y), GW_SEL(T1[q_].x, T1[q_].y), r4)                \
            GW_STEP(GW_SEL(T0[q_].z, T0[q_].w), GW_SEL(T1[q_].z, T1[q_].w), r4 + 2)            \
        }
        const int Bp4 = Bp;                          // dz^T rows are zero padded to a multiple of 64
        const unsigned himask = hi ? 0xFFFFFFFFu : 0u;
        if (DZ16) {
            // dz^T stored as bf16: 16 playlists of a row = two 16-byte loads; the dword x of a row holds playlists
            // (R0 + 2x, R0 + 2x + 1) and the lane half hi wants R0 + 2x + hi: one v_perm per operand dword
            const unsigned short* z0 = reinterpret_cast<const unsigned short*>(p.dzT) + (size_t)(ok0 ? vcol : 0) * p.ldT;
            const unsigned short* z1 = reinterpret_cast<const unsigned short*>(p.dzT) + (size_t)(ok1 ? vcol + 1 : 0) * p.ldT;
            const unsigned sel = hi ? 0x07060302u : 0x05040100u;
#define GW_LOAD16(U0, U1, R0)                                                                  \
            _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                 \
                const int r8 = min((R0) + 8 * q_, Bp4 - 8);                                    \
                U0[q_] = *reinterpret_cast<const uint4*>(z0 + r8);                             \
                U1[q_] = *reinterpret_cast<const uint4*>(z1 + r8);                             \
            }
#define GW_HALF(D) __uint_as_float(hi ? ((D) & 0xFFFF0000u) : ((D) << 16))
#define GW_MMA16(U0, U1, R0)                                                                   \
            {                                                                                  \
                float avs[8][NA];                                                              \
                _Pragma("unroll") for (int x_ = 0; x_ < 8; ++x_) {                             \
                    const float4 t4 = *reinterpret_cast<const float4*>(lds + (size_t)((R0) + 2 * x_ + hi) * HW + NA * j); \
                    avs[x_][0] = t4.x; avs[x_][1 % NA] = t4.y; avs[x_][2 % NA] = t4.z; avs[x_][3 % NA] = t4.w; \
                }                                                                              \
                const unsigned d0[8] = {U0[0].x, U0[0].y, U0[0].z, U0[0].w, U0[1].x, U0[1].y, U0[1].z, U0[1].w}; \
                const unsigned d1[8] = {U1[0].x, U1[0].y, U1[0].z, U1[0].w, U1[1].x, U1[1].y, U1[1].z, U1[1].w}; \
                _Pragma("unroll") for (int x_ = 0; x_ < 8; ++x_) { cs0 += GW_HALF(d0[x_]); cs1 += GW_HALF(d1[x_]); } \
                const bf16x8_t bx = __builtin_bit_cast(bf16x8_t, make_uint4(                   \
                    __builtin_amdgcn_perm(d0[1], d0[0], sel), __builtin_amdgcn_perm(d0[3], d0[2], sel), \
                    __builtin_amdgcn_perm(d0[5], d0[4], sel), __builtin_amdgcn_perm(d0[7], d0[6], sel))); \
                const bf16x8_t by = __builtin_bit_cast(bf16x8_t, make_uint4(                   \
                    __builtin_amdgcn_perm(d1[1], d1[0], sel), __builtin_amdgcn_perm(d1[3], d1[2], sel), \
                    __builtin_amdgcn_perm(d1[5], d1[4], sel), __builtin_amdgcn_perm(d1[7], d1[6], sel))); \
                _Pragma("unroll") for (int a = 0; a < NA; ++a) {                               \
                    const bf16x8_t af = pk_bf16x8(avs[0][a], avs[1][a], avs[2][a], avs[3][a], avs[4][a], avs[5][a], \
                                                  avs[6][a], avs[7][a]);                       \
                    acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bx, acc[a][0], 0, 0, 0); \
                    acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, by, acc[a][1], 0, 0, 0); \
                }                                                                              \
            }
            uint4 ua0[2], ua1[2], ub0[2], ub1[2];
            GW_LOAD16(ua0, ua1, 0)
            for (int r0 = 0; r0 < Bp; r0 += 32) {
                GW_LOAD16(ub0, ub1, r0 + 16)
                __builtin_amdgcn_sched_barrier(0);
                GW_MMA16(ua0, ua1, r0)
                __builtin_amdgcn_sched_barrier(0);
                GW_LOAD16(ua0, ua1, r0 + 32)
                __builtin_amdgcn_sched_barrier(0);
                GW_MMA16(ub0, ub1, r0 + 16)
                __builtin_amdgcn_sched_barrier(0);
            }
#undef GW_LOAD16
#undef GW_HALF
#undef GW_MMA16
        } else {
        float4 ta0[4], ta1[4], tb0[4], tb1[4];
        GW_LOAD(ta0, ta1, 0)
        for (int r0 = 0; r0 < Bp; r0 += 32) {        // straight-line 16 k-steps per iteration
            GW_LOAD(tb0, tb1, r0 + 16)
            __builtin_amdgcn_sched_barrier(0);       // keep the prefetch AHEAD of the 64 MFMAs below
            GW_MMA(ta0, ta1, r0)                     // (hipcc sinks loads next to their first use)
            __builtin_amdgcn_sched_barrier(0);
            GW_LOAD(ta0, ta1, r0 + 32)               // past the end: re-reads the last group
            __builtin_amdgcn_sched_barrier(0);
            GW_MMA(tb0, tb1, r0 + 16)
            __builtin_amdgcn_sched_barrier(0);
        }
        }
#undef GW_LOAD
#undef GW_STEP
#undef GW_SEL
#undef GW_MMA
        // D[i][j]: hidden unit hc0 + NA * i_idx + a, i_idx = (reg & 3) + 8 (reg >> 2) + 4 hi; column
        // v0 + 2 j + b.  The NA `a` accumulators of one reg are NA consecutive hidden units.
        if (DAE_EXP_ON(p.dbg & 1)) {
            float keep = cs0 + cs1;
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) keep += acc[a][0][e] + acc[a][1][e];
            if (keep == 12345.678f) p.gW[0] = keep;
            continue;
        }
        if (TR && NA == 4) {
            // register reg of accumulator (a, b) is row v0 + 2 i_idx + b, i_idx = (reg & 3) + 8 (reg >> 2) + 4 hi; lane j
            // holds hidden units hc0 + 4 j + a: one float4 per (b, reg)
            const float b1 = p.ad_b1, b2 = p.ad_b2, eps = p.ad_eps, al = p.ad_alpha;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int r4 = 0; r4 < 16; r4 += 4) {
                    if (p.ad_m) {
                        float4 pp[4], mm[4], vv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int reg = r4 + u;
                            const int v = v0 + 2 * ((reg & 3) + 8 * (reg >> 2) + 4 * hi) + b;
                            const size_t o = (size_t)(v < p.V ? v : 0) * p.H + hc0 + 4 * j;
                            pp[u] = nt_ld4(p.ad_p + o);
                            mm[u] = nt_ld4(p.ad_m + o);
                            vv[u] = nt_ld4(p.ad_v + o);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int reg = r4 + u;
                            const int v = v0 + 2 * ((reg & 3) + 8 * (reg >> 2) + 4 * hi) + b;
                            const size_t o = (size_t)(v < p.V ? v : 0) * p.H + hc0 + 4 * j;
                            const float g0 = acc[0][b][reg], g1 = acc[1 % NA][b][reg], g2 = acc[2 % NA][b][reg],
                                        g3 = acc[3 % NA][b][reg];
#define K6_ADAM(P, M, V, G)                                              \
                            M = M + (G - M) * (1.0f - b1);               \
                            V = V + (G * G - V) * (1.0f - b2);           \
                            P = P - (M * al) / (sqrtf(V) + eps);
                            K6_ADAM(pp[u].x, mm[u].x, vv[u].x, g0) K6_ADAM(pp[u].y, mm[u].y, vv[u].y, g1)
                            K6_ADAM(pp[u].z, mm[u].z, vv[u].z, g2) K6_ADAM(pp[u].w, mm[u].w, vv[u].w, g3)
#undef K6_ADAM
                            if (v < p.V) {
                                nt_st4(p.ad_p + o, pp[u]);
                                nt_st4(p.ad_m + o, mm[u]);
                                nt_st4(p.ad_v + o, vv[u]);
                            }
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int reg = r4 + u;
                            const int v = v0 + 2 * ((reg & 3) + 8 * (reg >> 2) + 4 * hi) + b;
                            if (v >= p.V) continue;
                            float4* dst = reinterpret_cast<float4*>(p.gW + (size_t)v * p.H + hc0 + 4 * j);
                            float4 o4 = make_float4(acc[0][b][reg], acc[1 % NA][b][reg], acc[2 % NA][b][reg], acc[3 % NA][b][reg]);
                            if (p.accumulate) { const float4 old = *dst; o4.x += old.x; o4.y += old.y; o4.z += old.z; o4.w += old.w; }
                            *dst = o4;
                        }
                    }
                }
            }
        } else
        if (NA == 4 && p.ad_m) {
            // the gradient tile goes straight into the Adam update of its parameters: W / m / v are read and written
            // in place, gW never reaches memory (7 passes over the tensor + 1 of the gradient become 6).  Same
            // per-element operations as adam_kernel, so the parameters are the bits dae_adam_step would produce.
            const float b1 = p.ad_b1, b2 = p.ad_b2, eps = p.ad_eps, al = p.ad_alpha;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int v = vcol + b;
                if (v >= p.V) continue;
                const size_t rbase = (size_t)v * p.H + hc0;
#pragma unroll
                for (int r4 = 0; r4 < 16; r4 += 4) {
                    float4 pp[4], mm[4], vv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int reg = r4 + u;
                        const size_t o = rbase + 4 * ((reg & 3) + 8 * (reg >> 2) + 4 * hi);
                        pp[u] = nt_ld4(p.ad_p + o);
                        mm[u] = nt_ld4(p.ad_m + o);
                        vv[u] = nt_ld4(p.ad_v + o);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int reg = r4 + u;
                        const size_t o = rbase + 4 * ((reg & 3) + 8 * (reg >> 2) + 4 * hi);
                        const float g0 = acc[0][b][reg], g1 = acc[1 % NA][b][reg], g2 = acc[2 % NA][b][reg],
                                    g3 = acc[3 % NA][b][reg];
#define K6_ADAM(P, M, V, G)                                              \
                        M = M + (G - M) * (1.0f - b1);                   \
                        V = V + (G * G - V) * (1.0f - b2);               \
                        P = P - (M * al) / (sqrtf(V) + eps);
                        K6_ADAM(pp[u].x, mm[u].x, vv[u].x, g0) K6_ADAM(pp[u].y, mm[u].y, vv[u].y, g1)
                        K6_ADAM(pp[u].z, mm[u].z, vv[u].z, g2) K6_ADAM(pp[u].w, mm[u].w, vv[u].w, g3)
#undef K6_ADAM
                        nt_st4(p.ad_p + o, pp[u]);
                        nt_st4(p.ad_m + o, mm[u]);
                        nt_st4(p.ad_v + o, vv[u]);
                    }
                }
            }
        } else
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int v = vcol + b;
            if (v >= p.V) continue;
            float* orow = p.gW + (size_t)v * p.H + hc0;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i_idx = (reg & 3) + 8 * (reg >> 2) + 4 * hi;
                if (NA == 4) {
                    float4* dst = reinterpret_cast<float4*>(orow + 4 * i_idx);
                    float4 o = make_float4(acc[0][b][reg], acc[1 % NA][b][reg], acc[2 % NA][b][reg], acc[3 % NA][b][reg]);
                    if (p.accumulate) { const float4 old = *dst; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
                    *dst = o;
                } else {
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        float* dst = orow + NA * i_idx + a;
                        *dst = p.accumulate ? *dst + acc[a][b][reg] : acc[a][b][reg];
                    }
                }
            }
        }
        if (p.gb && half == 0) {
            cs0 += __shfl_xor(cs0, 32);
            cs1 += __shfl_xor(cs1, 32);
            if (hi == 0) {
                if (ok0) p.gb[vcol] = cs0;
                if (ok1) p.gb[vcol + 1] = cs1;
            }
        }
    }
}

// ---- K6, transposed orientation (bf16 operands, dz^T stored as bf16, hidden a multiple of 128) --------------------
// Same product gW[v, hc] = sum_r dz[r, v] h[r, hc] with the operand roles swapped: A = dz^T (M = vocabulary rows),
// B = h^T (N = hidden units), so that an accumulator lane is a hidden unit and its registers are vocabulary rows.
// What that buys is memory shape on both sides:
//   * A fragments are plain 16-byte loads from the bf16 dz^T row of the lane (8 consecutive playlists): no selects,
//     no permutes, no conversions;
//   * a store instruction writes, per half-wave, 32 lanes x float4 = 512 contiguous bytes of ONE gW row (hidden =
//     hc0 + 4 n + a), where the other orientation writes 16-byte pieces of 32 rows (57 of its 113 us were the store);
//     the armed Adam update (dae_arm_decoder_adam) reads and writes W / m / v with the same shape.
// LDS holds h^T for the workgroup's 128 hidden units as bf16 B fragments in operand order: 4 KB per k-step of 16
// playlists, 64 KB at B = 256.  gb = dz^T 1 comes out of the matrix pipe as well (a ones fragment as B operand).
template <int NW, bool FULL = false>
__global__ __launch_bounds__(NW * 64, 1) void grad_wdec_t_kernel(const GwP p)
{
    extern __shared__ __attribute__((aligned(16))) uint4 ldsq[];      // [S][4][64] B fragments
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gs = DAE_NUM_XCD * p.n_half;
    const int q = blockIdx.x / gs, rem = blockIdx.x % gs;
    const int half = rem / DAE_NUM_XCD;
    const int bir = q * DAE_NUM_XCD + (rem % DAE_NUM_XCD);
    const int hc0 = half * 128;
    const int Bp = (p.B + 31) & ~31;
    const int S = Bp >> 4;                                             // k-steps of 16 playlists (2..16)

    // B fragments: (s, a, lane (n, hi)) = bf16 of h[16 s + 8 hi + x][hc0 + 4 n + a], x = 0..7; rows past B are zero
    for (int f = tid; f < S * 4 * 64; f += NW * 64) {
        const int fl = f & 63, fa = (f >> 6) & 3, fs = f >> 8;
        const int r0 = 16 * fs + 8 * (fl >> 5);
        const float* src = p.h + hc0 + 4 * (fl & 31) + fa;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = r0 + e < p.B ? src[(size_t)(r0 + e) * p.H] : 0.0f;
        ldsq[f] = make_uint4(pk_bf16(x[0], x[1]), pk_bf16(x[2], x[3]), pk_bf16(x[4], x[5]), pk_bf16(x[6], x[7]));
    }
    __syncthreads();

    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u));
    const unsigned short* dz = reinterpret_cast<const unsigned short*>(p.dzT);
    const int n_tiles = (p.V + 63) / 64;
    const int n_ws = p.nb_half * NW;
    constexpr int RING = 4;                                            // k-steps of A fragments in flight per wave (8: spills)
    for (int t = bir * NW + wave; t < n_tiles; t += n_ws) {
        const int v0 = t * 64;
        const int va = v0 + n, vb = v0 + 32 + n;                       // this lane's two A rows
        const unsigned short* ra = dz + (size_t)(va < p.V ? va : 0) * p.ldT + 8 * hi;
        const unsigned short* rb = dz + (size_t)(vb < p.V ? vb : 0) * p.ldT + 8 * hi;
        f32x16 acc[2][4], accg[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int e = 0; e < 16; ++e) accg[m][e] = 0.0f;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][a][e] = 0.0f;
        }
        uint4 qa[RING], qb[RING];
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int su = (FULL || u < S) ? u : S - 1;
            qa[u] = *reinterpret_cast<const uint4*>(ra + 16 * su);
            qb[u] = *reinterpret_cast<const uint4*>(rb + 16 * su);
        }
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            // (FULL: S == 16, a batch of 241 .. 256, known at compile time -- with the wave-uniform tests in the loop hipcc ends
            // every step on s_waitcnt vmcnt(0), i.e. on the ring slot it has just requested)
            if (FULL || s_ < S) {                                      // wave-uniform
                const bf16x8_t fa = __builtin_bit_cast(bf16x8_t, qa[s_ % RING]);
                const bf16x8_t fb = __builtin_bit_cast(bf16x8_t, qb[s_ % RING]);
                if (s_ + RING < 16) {                                  // refill the slot (clamped: values unused past S)
                    const int sn = (FULL || s_ + RING < S) ? s_ + RING : S - 1;
                    qa[s_ % RING] = *reinterpret_cast<const uint4*>(ra + 16 * sn);
                    qb[s_ % RING] = *reinterpret_cast<const uint4*>(rb + 16 * sn);
                }
                uint4 bq[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) bq[a] = ldsq[(s_ * 4 + a) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, bq[a]);
                    acc[0][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, bf, acc[0][a], 0, 0, 0);
                    acc[1][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, bf, acc[1][a], 0, 0, 0);
                }
                accg[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, ones, accg[0], 0, 0, 0);
                accg[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, ones, accg[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (p.gb && half == 0 && n == 0) {                             // every lane holds the row sums; lanes 0 and 32 store
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int v = v0 + 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * hi;
                    if (v < p.V) p.gb[v] = accg[m][reg];
                }
        }
        // lane n holds hidden units hc0 + 4 n + a (a = the 4 accumulators of a register), register reg the row
        // v0 + 32 m + (reg & 3) + 8 (reg >> 2) + 4 hi: one float4 per (m, reg), 512 contiguous bytes per half-wave
        if (p.ad_m) {
            const float b1 = p.ad_b1, b2 = p.ad_b2, eps = p.ad_eps, al = p.ad_alpha;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r4 = 0; r4 < 16; r4 += 4) {
                    float4 pp[4], mm[4], vv[4];
                    size_t o[4];
                    bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int reg = r4 + u;
                        const int v = v0 + 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * hi;
                        ok[u] = v < p.V;
                        o[u] = (size_t)(ok[u] ? v : 0) * p.H + hc0 + 4 * n;
                        pp[u] = nt_ld4(p.ad_p + o[u]);
                        mm[u] = nt_ld4(p.ad_m + o[u]);
                        vv[u] = nt_ld4(p.ad_v + o[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int reg = r4 + u;
                        const float g0 = acc[m][0][reg], g1 = acc[m][1][reg], g2 = acc[m][2][reg], g3 = acc[m][3][reg];
#define K6_ADAM(P, M, V, G)                                              \
                        M = M + (G - M) * (1.0f - b1);                   \
                        V = V + (G * G - V) * (1.0f - b2);               \
                        P = P - (M * al) / (sqrtf(V) + eps);
                        K6_ADAM(pp[u].x, mm[u].x, vv[u].x, g0) K6_ADAM(pp[u].y, mm[u].y, vv[u].y, g1)
                        K6_ADAM(pp[u].z, mm[u].z, vv[u].z, g2) K6_ADAM(pp[u].w, mm[u].w, vv[u].w, g3)
#undef K6_ADAM
                        if (ok[u]) {
                            nt_st4(p.ad_p + o[u], pp[u]);
                            nt_st4(p.ad_m + o[u], mm[u]);
                            nt_st4(p.ad_v + o[u], vv[u]);
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int v = v0 + 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * hi;
                    if (v < p.V)
                        *reinterpret_cast<float4*>(p.gW + (size_t)v * p.H + hc0 + 4 * n) =
                            make_float4(acc[m][0][reg], acc[m][1][reg], acc[m][2][reg], acc[m][3][reg]);
                }
        }
    }
}

// ---- K6, transposed orientation, the armed-Adam form with its state streams kept in flight (round 6) ---------------------------
// grad_wdec_t_kernel above runs a tile in two phases -- 160 MFMAs with 4 KB of dz^T requests in flight per wave, then the Adam
// pass in groups of 12 x 1 KB loads, compute, 12 stores -- and sits at 5.3 TB/s for 1.14 GB with its waves parked 59 % of the time
// (SQ counters, r06 notes 8): too few bytes in flight, not too many instructions.  At 248 registers it has no room for more.
// This form halves the tile (32 decoder rows: 64 accumulator registers instead of 128 + 32) and spends the registers on the
// streams: the first group of p / m / v rows of a tile is requested BEFORE its MFMAs (under which it arrives), and inside the Adam
// pass group g + 1 is requested before group g is computed and stored (two buffers).  Same operands, same k order per
// element, same update operations as above: the parameters stay bit-identical to dense Adam.
template <int NW, bool FULL>
__global__ __launch_bounds__(NW * 64, 1) void grad_wdec_t32_kernel(const GwP p)
{
    extern __shared__ __attribute__((aligned(16))) uint4 ldsq[];      // [S][4][64] B fragments
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gs = DAE_NUM_XCD * p.n_half;
    const int q = blockIdx.x / gs, rem = blockIdx.x % gs;
    const int half = rem / DAE_NUM_XCD;
    const int bir = q * DAE_NUM_XCD + (rem % DAE_NUM_XCD);
    const int hc0 = half * 128;
    const int Bp = (p.B + 31) & ~31;
    const int S = Bp >> 4;

    for (int f = tid; f < S * 4 * 64; f += NW * 64) {
        const int fl = f & 63, fa = (f >> 6) & 3, fs = f >> 8;
        const int r0 = 16 * fs + 8 * (fl >> 5);
        const float* src = p.h + hc0 + 4 * (fl & 31) + fa;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = r0 + e < p.B ? src[(size_t)(r0 + e) * p.H] : 0.0f;
        ldsq[f] = make_uint4(pk_bf16(x[0], x[1]), pk_bf16(x[2], x[3]), pk_bf16(x[4], x[5]), pk_bf16(x[6], x[7]));
    }
    __syncthreads();

    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u));
    const unsigned short* dz = reinterpret_cast<const unsigned short*>(p.dzT);
    const int n_tiles = (p.V + 31) / 32;
    const int n_ws = p.nb_half * NW;
    constexpr int RING = 4;
    const float b1 = p.ad_b1, b2 = p.ad_b2, eps = p.ad_eps, al = p.ad_alpha;
    for (int t = bir * NW + wave; t < n_tiles; t += n_ws) {
        const int v0 = t * 32;
        const int va = v0 + n;
        const unsigned short* ra = dz + (size_t)(va < p.V ? va : 0) * p.ldT + 8 * hi;
        f32x16 acc[4], accg;
#pragma unroll
        for (int e = 0; e < 16; ++e) accg[e] = 0.0f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.0f;
        uint4 qa[RING];
#pragma unroll
        for (int u = 0; u < RING; ++u) qa[u] = *reinterpret_cast<const uint4*>(ra + 16 * ((FULL || u < S) ? u : S - 1));
        // the Adam pass's rows: register reg of lane (n, hi) is decoder row v0 + (reg & 3) + 8 (reg >> 2) + 4 hi, hidden hc0 + 4 n + a
        float4 P[2][4], M[2][4], Vv[2][4];
        size_t off[2][4];
        bool ok[2][4];
        auto issue = [&](int buf, int r4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int reg = r4 + u;
                const int v = v0 + (reg & 3) + 8 * (reg >> 2) + 4 * hi;
                ok[buf][u] = v < p.V;
                off[buf][u] = (size_t)(ok[buf][u] ? v : 0) * p.H + hc0 + 4 * n;
                P[buf][u] = nt_ld4(p.ad_p + off[buf][u]);
                M[buf][u] = nt_ld4(p.ad_m + off[buf][u]);
                Vv[buf][u] = nt_ld4(p.ad_v + off[buf][u]);
            }
        };
        issue(0, 0);                                                   // arrives under the MFMAs
        // (S == 16 -- a batch of 241 .. 256 -- is a template case: with the wave-uniform `s_ < S` tests in the loop hipcc ends every
        // step on s_waitcnt vmcnt(0), i.e. on the ring slot it has just requested: 16 memory round trips per tile instead of a ring)
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            if (FULL || s_ < S) {                                      // wave-uniform
                const bf16x8_t fa = __builtin_bit_cast(bf16x8_t, qa[s_ % RING]);
                if (s_ + RING < 16) {
                    const int sn = (FULL || s_ + RING < S) ? s_ + RING : S - 1;
                    qa[s_ % RING] = *reinterpret_cast<const uint4*>(ra + 16 * sn);
                }
                uint4 bq[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) bq[a] = ldsq[(s_ * 4 + a) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8_t, bq[a]), acc[a], 0, 0, 0);
                accg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, ones, accg, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (p.gb && half == 0 && n == 0) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int v = v0 + (reg & 3) + 8 * (reg >> 2) + 4 * hi;
                if (v < p.V) p.gb[v] = accg[reg];
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cb = g & 1;
            if (g + 1 < 4) issue(cb ^ 1, 4 * (g + 1));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int reg = 4 * g + u;
                float4 pp = P[cb][u], mm = M[cb][u], vv = Vv[cb][u];
                const float g0 = acc[0][reg], g1 = acc[1][reg], g2 = acc[2][reg], g3 = acc[3][reg];
#define K6_ADAM(Pq, Mq, Vq, G)                                           \
                Mq = Mq + (G - Mq) * (1.0f - b1);                        \
                Vq = Vq + (G * G - Vq) * (1.0f - b2);                    \
                Pq = Pq - (Mq * al) / (sqrtf(Vq) + eps);
                K6_ADAM(pp.x, mm.x, vv.x, g0) K6_ADAM(pp.y, mm.y, vv.y, g1)
                K6_ADAM(pp.z, mm.z, vv.z, g2) K6_ADAM(pp.w, mm.w, vv.w, g3)
#undef K6_ADAM
                if (ok[cb][u]) {
                    nt_st4(p.ad_p + off[cb][u], pp);
                    nt_st4(p.ad_m + off[cb][u], mm);
                    nt_st4(p.ad_v + off[cb][u], vv);
                }
            }
        }
    }
}

// ---- K6 with fp32 operands (train_dtype = f32), the armed-Adam form with its state streams kept in flight (round 6) -------------
// grad_wdec_t32_kernel's plan on v_mfma_f32_32x32x2_f32: a tile of 32 decoder rows, A = dz^T (fp32 rows; a float4 = 4
// playlists = two k-steps, the lane half hi taking the even / odd one), B = h^T from LDS (one float4 per lane and k-step: the four
// accumulators' hidden units), 512 MFMAs per tile, the row sums (gb) on the VALU; then the Adam pass of section 12 -- the first
// group of p / m / v rows requested before the MFMAs, group g + 1 before group g is computed.  The generic kernel it replaces
// for this case (grad_wdec_kernel<4, 8, false, false, true>) ran its two phases back to back at 12 KB in flight per wave: 349 us
// for 180 us of matrix work and 1.22 GB.
template <int NW, bool FULL>
__global__ __launch_bounds__(NW * 64, 1) void grad_wdec_t32_f32_kernel(const GwP p)
{
    extern __shared__ __attribute__((aligned(16))) float4 ldsf[];     // [Bp / 2 k-steps][64 lanes]: h[2 g + hi][hc0 + 4 n .. + 3]
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gs = DAE_NUM_XCD * p.n_half;
    const int q_ = blockIdx.x / gs, rem = blockIdx.x % gs;
    const int half = rem / DAE_NUM_XCD;
    const int bir = q_ * DAE_NUM_XCD + (rem % DAE_NUM_XCD);
    const int hc0 = half * 128;
    const int Bp = (p.B + 31) & ~31;
    const int Q = Bp >> 2;                                             // float4 of a dz^T row (4 playlists each)

    for (int f = tid; f < (Bp >> 1) * 64; f += NW * 64) {
        const int fl = f & 63, g = f >> 6;
        const int r = 2 * g + (fl >> 5);
        ldsf[f] = r < p.B ? *reinterpret_cast<const float4*>(p.h + (size_t)r * p.H + hc0 + 4 * (fl & 31)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    const int n_tiles = (p.V + 31) / 32;
    const int n_ws = p.nb_half * NW;
    constexpr int RING = 4;
    const unsigned himask = hi ? 0xFFFFFFFFu : 0u;
    // (Tried: the second wave of each SIMD starting 2 .. 16 x 8 k cycles late, so that one wave's MFMAs run under the other's state
    // streams -- 345 - 352 us at every setting: phases that coincide are not what this launch loses its time to.)
#define K6F_SEL(A, Bv) __uint_as_float((__float_as_uint(Bv) & himask) | (__float_as_uint(A) & ~himask))
    const float b1 = p.ad_b1, b2 = p.ad_b2, eps = p.ad_eps, al = p.ad_alpha;
    for (int t = bir * NW + wave; t < n_tiles; t += n_ws) {
        const int v0 = t * 32;
        const int va = v0 + n;
        const float4* ra = reinterpret_cast<const float4*>(p.dzT + (size_t)(va < p.V ? va : 0) * p.ldT);
        f32x16 acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.0f;
        float cs = 0.0f;
        float4 qa[RING];
#pragma unroll
        for (int u = 0; u < RING; ++u) qa[u] = ra[(FULL || u < Q) ? u : Q - 1];
        float4 P[2][4], M[2][4], Vv[2][4];
        size_t off[2][4];
        bool ok[2][4];
        auto issue = [&](int buf, int r4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int reg = r4 + u;
                const int v = v0 + (reg & 3) + 8 * (reg >> 2) + 4 * hi;
                ok[buf][u] = v < p.V;
                off[buf][u] = (size_t)(ok[buf][u] ? v : 0) * p.H + hc0 + 4 * n;
                P[buf][u] = nt_ld4(p.ad_p + off[buf][u]);
                M[buf][u] = nt_ld4(p.ad_m + off[buf][u]);
                Vv[buf][u] = nt_ld4(p.ad_v + off[buf][u]);
            }
        };
        issue(0, 0);                                                   // arrives under the 512 MFMAs
        const int q_end = FULL ? 64 : Q;
        for (int q0 = 0; q0 < q_end; q0 += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                const int qq = q0 + u;
                const float4 d4 = qa[u];
                {
                    const int qn = qq + RING;
                    qa[u] = ra[(FULL ? qn < 64 : qn < Q) ? qn : q_end - 1];
                }
                const float4 bA = ldsf[(2 * qq) * 64 + lane], bB = ldsf[(2 * qq + 1) * 64 + lane];
                const float dA = K6F_SEL(d4.x, d4.y), dB = K6F_SEL(d4.z, d4.w);
                __builtin_amdgcn_sched_barrier(0);
                cs += dA; cs += dB;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dA, bA.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dA, bA.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(dA, bA.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(dA, bA.w, acc[3], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dB, bB.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dB, bB.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(dB, bB.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(dB, bB.w, acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        cs += __shfl_xor(cs, 32);                                      // the two lane halves hold the even / odd playlists of the row
        if (p.gb && half == 0 && hi == 0 && va < p.V) p.gb[va] = cs;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cb = g & 1;
            if (g + 1 < 4) issue(cb ^ 1, 4 * (g + 1));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int reg = 4 * g + u;
                float4 pp = P[cb][u], mm = M[cb][u], vv = Vv[cb][u];
                const float g0 = acc[0][reg], g1 = acc[1][reg], g2 = acc[2][reg], g3 = acc[3][reg];
#define K6_ADAM(Pq, Mq, Vq, G)                                           \
                Mq = Mq + (G - Mq) * (1.0f - b1);                        \
                Vq = Vq + (G * G - Vq) * (1.0f - b2);                    \
                Pq = Pq - (Mq * al) / (sqrtf(Vq) + eps);
                K6_ADAM(pp.x, mm.x, vv.x, g0) K6_ADAM(pp.y, mm.y, vv.y, g1)
                K6_ADAM(pp.z, mm.z, vv.z, g2) K6_ADAM(pp.w, mm.w, vv.w, g3)
#undef K6_ADAM
                if (ok[cb][u]) {
                    nt_st4(p.ad_p + off[cb][u], pp);
                    nt_st4(p.ad_m + off[cb][u], mm);
                    nt_st4(p.ad_v + off[cb][u], vv);
                }
            }
        }
    }
#undef K6F_SEL
}

// ---- K7: dh partial [chunk][r][hc] = sum_{v in chunk} dzT[v, r] * W[v, hc] -----------------------
// a wave owns one (hidden half of 128, 64 playlists) output tile for one chunk of V: 8 accumulators;
// A = W rows (float4 per lane: hc0 + 4 i + a), B = dz^T rows (float2 per lane: r0 + 2 j + b); no LDS.
struct DhP {
    const float* dzT; int64_t ldT;     // [V, ldT]  (ldT >= Bpad64)
    const float* W; int H, V;
    float* part;                       // [n_chunk][Bpad64][H]
    int n_chunk, chunk, Bpad64, n_half, n_rblk;
    int fast32;                        // W and dz^T both end below 4 GB: whole chunks take 32-bit byte offsets from the matrix base
};

// BF16 (dae_set_train_dtype, NA = 4 only): the 8 k-steps of a block (16 vocabulary rows) become ONE
// v_mfma_f32_32x32x16_bf16 per accumulator: k-slot x of lane half hi is row V0 + 2x + hi in both operands.
// DZ16: dz^T is stored as bf16; a lane's two playlists (r0 + 2j, r0 + 2j + 1) of a row are one dword.
template <int NA, bool BF16 = false, bool DZ16 = false>
__global__ __launch_bounds__(256, 1) void grad_hidden_kernel(const DhP p)
{
    constexpr int HW = 32 * NA;
    const int lane = threadIdx.x & 63, hi = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_out = p.n_half * p.n_rblk;
    const int total = n_out * p.n_chunk;
    // (Round 6, settled with scripts/probe/fetch_calib.hip: this launch really pulls 430 - 474 MB through the fabric for 261 MB
    // algorithmic -- every W row leaves HBM twice.  A chunk's eight output tiles are two workgroups; dealing the pair to ONE XCD
    // (blocks b, b + 8) changed nothing (FETCH_SIZE 215 113 KB before and after): an XCD streams 16 chunks of 1.36 MB at once
    // through 4 MB of L2, the partner's lines are gone before it arrives.  Reading W once needs the eight tiles in one workgroup
    // with the W block shared through LDS -- not built.)
    for (int w = blockIdx.x * 4 + wave; w < total; w += gridDim.x * 4) {
        const int ot = w % n_out, ch = w / n_out;      // neighbours share the W chunk
        // the four waves of a workgroup take the playlist blocks of ONE hidden half (n_rblk = 4: the shipped batch of 256), so
        // every W byte is read by one workgroup only -- its waves ask for the same lines within a few hundred cycles
        const int half = ot / p.n_rblk, rblk = ot % p.n_rblk;
        const int hc0 = half * HW, r0 = rblk * 64;
        const int v_beg = ch * p.chunk;
        int v_end = v_beg + p.chunk;
        if (v_end > p.V) v_end = p.V;
        f32x16 acc[NA][2];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;
        // (byte offsets, 32 bits: the launcher takes the fast form only when both matrices lie below 4 GB)
        const unsigned w_row = (unsigned)p.H * 4u, d_row = (unsigned)p.ldT * (DZ16 ? 2u : 4u);
        const unsigned lane_w = (unsigned)(hc0 + NA * j) * 4u + (unsigned)hi * w_row;
        const unsigned lane_d = (unsigned)(r0 + 2 * j) * (DZ16 ? 2u : 4u) + (unsigned)hi * d_row;
        const float* Wl = p.W + hc0 + NA * j;
        const float* Dl = p.dzT + r0 + 2 * j;
        const unsigned short* Dl16 = reinterpret_cast<const unsigned short*>(p.dzT) + r0 + 2 * j;
        // 16 vocabulary rows (8 k-steps, 64 MFMAs) per block, two register sets: the loads of block n+1 are
        // issued before the MFMAs of block n (one wave per SIMD: nothing else hides the ~2 us of HBM latency;
        // without the second set this kernel ran at 0.6 of its matrix time).  Rows past the chunk read row
        // v_beg and multiply by 0.
#define DH_LOAD(AV, D, V0)                                                                     \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                        \
            const int v = (V0) + 2 * s + hi;                                                   \
            const bool in = FASTV || v < v_end;                                                \
            const int vc = in ? v : v_beg;                                                     \
            /* FASTV: a wave-uniform row base (scalar registers) + one 32-bit lane offset -- the per-lane form costs a */ \
            /* 64-bit multiply (four quarter-rate instructions) per load, ~700 cycles per block next to 256 of MFMA   */ \
            const float* wr = FASTV ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.W) +     \
                                          ((unsigned)((V0) + 2 * s) * w_row + lane_w))                         \
                                    : Wl + (size_t)vc * p.H;                                   \
            if (NA == 4) {                                                                     \
                const float4 t4 = *reinterpret_cast<const float4*>(wr);                        \
                AV[s][0] = t4.x; AV[s][1 % NA] = t4.y; AV[s][2 % NA] = t4.z; AV[s][3 % NA] = t4.w; \
            } else if (NA == 2) {                                                              \
                const float2 t2 = *reinterpret_cast<const float2*>(wr);                        \
                AV[s][0] = t2.x; AV[s][1 % NA] = t2.y;                                         \
            } else {                                                                           \
                AV[s][0] = wr[0];                                                              \
            }                                                                                  \
            const char* dfast = reinterpret_cast<const char*>(p.dzT) + ((unsigned)((V0) + 2 * s) * d_row + lane_d); \
            if (DZ16) D[s].x = __uint_as_float(*reinterpret_cast<const unsigned*>(             \
                FASTV ? dfast : reinterpret_cast<const char*>(Dl16 + (size_t)vc * p.ldT)));    \
            else D[s] = *reinterpret_cast<const float2*>(                                      \
                FASTV ? dfast : reinterpret_cast<const char*>(Dl + (size_t)vc * p.ldT));       \
        }
// the "past the chunk -> 0" select sits HERE, not next to the load: a select on a loaded value in the load
// stage makes the compiler wait for that load before the sched_barrier, i.e. before the MFMAs it should hide under
#define DH_MMA(AV, D, V0)                                                                      \
        if (BF16) {                                                                            \
            bf16x8_t bx, by;                                                                   \
            if (DZ16) {                                                                        \
                unsigned dd[8];                                                                \
                _Pragma("unroll") for (int s = 0; s < 8; ++s)                                  \
                    dd[s] = (FASTV || (V0) + 2 * s + hi < v_end) ? __float_as_uint(D[s].x) : 0u; \
                bx = __builtin_bit_cast(bf16x8_t, make_uint4(                                  \
                    __builtin_amdgcn_perm(dd[1], dd[0], 0x05040100u), __builtin_amdgcn_perm(dd[3], dd[2], 0x05040100u), \
                    __builtin_amdgcn_perm(dd[5], dd[4], 0x05040100u), __builtin_amdgcn_perm(dd[7], dd[6], 0x05040100u))); \
                by = __builtin_bit_cast(bf16x8_t, make_uint4(                                  \
                    __builtin_amdgcn_perm(dd[1], dd[0], 0x07060302u), __builtin_amdgcn_perm(dd[3], dd[2], 0x07060302u), \
                    __builtin_amdgcn_perm(dd[5], dd[4], 0x07060302u), __builtin_amdgcn_perm(dd[7], dd[6], 0x07060302u))); \
            } else {                                                                           \
            float dx[8], dy[8];                                                                \
            _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                    \
                const bool in = FASTV || (V0) + 2 * s + hi < v_end;                            \
                dx[s] = in ? D[s].x : 0.f; dy[s] = in ? D[s].y : 0.f;                          \
            }                                                                                  \
            bx = pk_bf16x8(dx[0], dx[1], dx[2], dx[3], dx[4], dx[5], dx[6], dx[7]);            \
            by = pk_bf16x8(dy[0], dy[1], dy[2], dy[3], dy[4], dy[5], dy[6], dy[7]);            \
            }                                                                                  \
            _Pragma("unroll") for (int a = 0; a < NA; ++a) {                                   \
                const bf16x8_t af = pk_bf16x8(AV[0][a], AV[1][a], AV[2][a], AV[3][a], AV[4][a], AV[5][a], \
                                              AV[6][a], AV[7][a]);                             \
                acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, af, acc[a][0], 0, 0, 0); \
                acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(by, af, acc[a][1], 0, 0, 0); \
            }                                                                                  \
        } else                                                                                 \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                        \
            const bool in = FASTV || (V0) + 2 * s + hi < v_end;                                \
            const float dx = in ? D[s].x : 0.f, dy = in ? D[s].y : 0.f;                        \
            _Pragma("unroll") for (int a = 0; a < NA; ++a)                                     \
                acc[a][0] = NA == 4 ? __builtin_amdgcn_mfma_f32_32x32x2f32(dx, AV[s][a], acc[a][0], 0, 0, 0) \
                                    : __builtin_amdgcn_mfma_f32_32x32x2f32(AV[s][a], dx, acc[a][0], 0, 0, 0); \
            _Pragma("unroll") for (int a = 0; a < NA; ++a)                                     \
                acc[a][1] = NA == 4 ? __builtin_amdgcn_mfma_f32_32x32x2f32(dy, AV[s][a], acc[a][1], 0, 0, 0) \
                                    : __builtin_amdgcn_mfma_f32_32x32x2f32(AV[s][a], dy, acc[a][1], 0, 0, 0); \
        }
        // A chunk that is whole (a multiple of 32 rows, and the prefetch past its end stays inside the matrix -- every
        // chunk but the last) needs no "row past the chunk" clamps and selects: its addresses are then a uniform part
        // plus a per-lane constant, and the ~120 VALU instructions per block that a single wave per SIMD executes
        // with the matrix pipe idle shrink accordingly.
        auto body = [&](auto fastc) {
            constexpr bool FASTV = decltype(fastc)::value;
            float avA[8][NA], avB[8][NA];
            float2 dA[8], dB[8];
            DH_LOAD(avA, dA, v_beg)
            for (int v0 = v_beg; v0 < v_end; v0 += 32) {
                DH_LOAD(avB, dB, v0 + 16)
                __builtin_amdgcn_sched_barrier(0);
                DH_MMA(avA, dA, v0)
                __builtin_amdgcn_sched_barrier(0);
                DH_LOAD(avA, dA, v0 + 32)
                __builtin_amdgcn_sched_barrier(0);
                DH_MMA(avB, dB, v0 + 16)
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (((v_end - v_beg) & 31) == 0 && v_end + 16 <= p.V && p.fast32) body(IntC<1>{});
        else body(IntC<0>{});
#undef DH_LOAD
#undef DH_MMA
        float* prow = p.part + ((size_t)ch * p.Bpad64) * p.H + hc0;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r = r0 + 2 * j + b;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i_idx = (reg & 3) + 8 * (reg >> 2) + 4 * hi;
                if (NA == 4) {
                    // operands swapped (NA = 4): the lane is the hidden unit hc0 + 4 j + a, the register the playlist
                    // r0 + 2 i_idx + b -- 512 contiguous bytes of one partial row per half-wave
                    *reinterpret_cast<float4*>(prow + (size_t)(r0 + 2 * i_idx + b) * p.H + 4 * j) =
                        make_float4(acc[0][b][reg], acc[1 % NA][b][reg], acc[2 % NA][b][reg], acc[3 % NA][b][reg]);
                } else {
#pragma unroll
                    for (int a = 0; a < NA; ++a) prow[(size_t)r * p.H + NA * i_idx + a] = acc[a][b][reg];
                }
            }
        }
    }
}

// ---- K8a: dpre[r, k] = (sum_chunks part) * (h > 0 ? 1/kp : 0) * s (1 - s)  (DAEs.py:67-68) ------
__global__ __launch_bounds__(256) void hidden_backward_kernel(const float* __restrict__ part,
                                                              int n_chunk, int Bpad64, int H, int B,
                                                              const float* __restrict__ h,
                                                              const float* __restrict__ sg, float kp,
                                                              float* __restrict__ dpre)
{
    const size_t n = (size_t)B * H;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (size_t)gridDim.x * 256) {
        float s = 0.f;
        // fixed order; 32 loads in flight per thread (one dependent load per iteration was 32 us for 33 MB, 8 at a time 11 us:
        // 65 536 threads x 128 chunk partials is 16 round trips of 8)
        int c = 0;
        for (; c + 32 <= n_chunk; c += 32) {
            float q[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) q[u] = part[(size_t)(c + u) * Bpad64 * H + o];
#pragma unroll
            for (int u = 0; u < 32; ++u) s += q[u];
        }
        for (; c + 8 <= n_chunk; c += 8) {
            float q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = part[(size_t)(c + u) * Bpad64 * H + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += q[u];
        }
        for (; c < n_chunk; ++c) s += part[(size_t)c * Bpad64 * H + o];
        const float sv = sg[o];
        const float keep = h[o] != 0.0f ? 1.0f / kp : 0.0f;
        dpre[o] = s * keep * sv * (1.0f - sv);
    }
}

// column sums over rows: out[k] = sum_r a[r, k] (+ lambda * base[k])
// one block per 64 columns; 4 row lanes per column, combined in fixed order (deterministic)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, int B, int H,
                                                     float lambda, const float* __restrict__ base,
                                                     float* __restrict__ out)
{
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + c;
    float s = 0.f;
    if (k < H) {
        int r = rl;
        for (; r + 28 < B; r += 32) {                // 8 loads in flight, summed in row order
            float q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = a[(size_t)(r + 4 * u) * H + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += q[u];
        }
        for (; r < B; r += 4) s += a[(size_t)r * H + k];
    }
    part[rl][c] = s;
    __syncthreads();
    if (rl == 0 && k < H)
        out[k] = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) + (lambda != 0.f ? lambda * base[k] : 0.f);
}

// ---- K8b: gW_enc[c, :] += xhat[r, c] * dpre[r, :]  (row-sparse; several rows may share c) ---------
// xhat is recomputed exactly as the encode kernel does (same dropout draws, same order).
__global__ __launch_bounds__(256) void scatter_gwenc_kernel(const int32_t* __restrict__ row_ptr,
                                                            const int32_t* __restrict__ col,
                                                            const float* __restrict__ val, int B,
                                                            int H, float ikp, uint32_t seed,
                                                            int col_lo, int col_hi,
                                                            const float* __restrict__ dpre,
                                                            float* __restrict__ gW)
{
    // the row's entries (dropped-out value, column) are staged in LDS by all threads at once: walking them with
    // one scalar load per iteration, twice, was ~35 us of dependent-load latency for rows of ~100 entries
    constexpr int CAP = 1024;
    __shared__ float xs[CAP];
    __shared__ int cs[CAP];
    const int row = blockIdx.x, tid = threadIdx.x;
    if (row >= B) return;
    const int beg = row_ptr[row], end = row_ptr[row + 1];
    // pass 1: the row sum, in entry order (the order the encode kernel uses)
    float s = 0.0f;
    for (int c0 = beg; c0 < end; c0 += CAP) {
        const int n = min(CAP, end - c0);
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            float x = val[c0 + i];
            const int c = col[c0 + i];
            if (ikp < 1.0f) x = (x / ikp) * floorf(ikp + dae_uniform(seed, 0U, (uint32_t)row, (uint32_t)c));
            xs[i] = x; cs[i] = c;
        }
        __syncthreads();
        for (int i = 0; i < n; ++i) s += xs[i];
    }
    const float denom = s + 1e-10f;
    // pass 2: gW[c, :] += xhat * dpre[row, :]   (a single chunk -- every playlist batch -- is still in LDS)
    for (int c0 = beg; c0 < end; c0 += CAP) {
        const int n = min(CAP, end - c0);
        if (end - beg > CAP) {
            __syncthreads();
            for (int i = tid; i < n; i += 256) {
                float x = val[c0 + i];
                const int c = col[c0 + i];
                if (ikp < 1.0f) x = (x / ikp) * floorf(ikp + dae_uniform(seed, 0U, (uint32_t)row, (uint32_t)c));
                xs[i] = x; cs[i] = c;
            }
            __syncthreads();
        }
        // xhat once per entry (the same division, by one thread instead of by every hidden unit's: ~100 IEEE divides per thread)
        __syncthreads();
        for (int i = tid; i < n; i += 256) xs[i] = xs[i] / denom;
        __syncthreads();
        for (int k = tid; k < H; k += 256) {
            const float dv = dpre[(size_t)row * H + k];
            for (int i = 0; i < n; ++i) {
                const float w = xs[i];
                const int c = cs[i];
                if (w != 0.0f && c >= col_lo && c < col_hi) atomicAdd(&gW[(size_t)(c - col_lo) * H + k], w * dv);
            }
        }
    }
}

// ---- vocabulary-sharded training: partial pre-activation of the encoder -------------
// pre[r, k] = sum over this shard's columns of xhat[r, c] * W_loc[c - col_lo, k]; xhat is normalised
// by the row's GLOBAL sum (the CSR carries the whole row on every rank), no bias, no sigmoid.
__global__ __launch_bounds__(256) void encode_partial_kernel(const int32_t* __restrict__ row_ptr,
                                                             const int32_t* __restrict__ col,
                                                             const float* __restrict__ val, int B,
                                                             int H, float ikp, uint32_t seed,
                                                             int col_lo, int col_hi,
                                                             const float* __restrict__ W,
                                                             float* __restrict__ pre)
{
    const int row = blockIdx.x;
    if (row >= B) return;
    const int beg = row_ptr[row], end = row_ptr[row + 1];
    float s = 0.0f;
    for (int i = beg; i < end; ++i) {
        float x = val[i];
        if (ikp < 1.0f) x = (x / ikp) * floorf(ikp + dae_uniform(seed, 0U, (uint32_t)row, (uint32_t)col[i]));
        s += x;
    }
    const float denom = s + 1e-10f;
    for (int k = threadIdx.x; k < H; k += 256) {
        float acc = 0.0f;
        for (int i = beg; i < end; ++i) {
            const int c = col[i];
            if (c < col_lo || c >= col_hi) continue;
            float x = val[i];
            if (ikp < 1.0f) x = (x / ikp) * floorf(ikp + dae_uniform(seed, 0U, (uint32_t)row, (uint32_t)c));
            acc = fmaf(x / denom, W[(size_t)(c - col_lo) * H + k], acc);
        }
        pre[(size_t)row * H + k] = acc;
    }
}

// sg = sigmoid(pre + b_enc); h = dropout(sg, kp) with the encode kernel's draws (DAEs.py:66-68)
__global__ __launch_bounds__(256) void activate_kernel(const float* __restrict__ pre,
                                                       const float* __restrict__ b_enc, int B, int H,
                                                       float kp, uint32_t seed,
                                                       float* __restrict__ h, float* __restrict__ sg)
{
    const size_t n = (size_t)B * H;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (size_t)gridDim.x * 256) {
        const int row = (int)(o / H), hu = (int)(o - (size_t)row * H);
        float hv = dae_sigmoidf(pre[o] + b_enc[hu]);
        sg[o] = hv;
        if (kp < 1.0f) hv = (hv / kp) * floorf(kp + dae_uniform(seed, 1U, (uint32_t)row, (uint32_t)hu));
        h[o] = hv;
    }
}

// dh[o] = sum over the K7 chunks, fixed order
__global__ __launch_bounds__(256) void sum_chunks_kernel(const float* __restrict__ part, int n_chunk,
                                                         size_t chunk_stride, size_t n,
                                                         float* __restrict__ out)
{
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (size_t)gridDim.x * 256) {
        float s = 0.f;
        int c = 0;
        for (; c + 8 <= n_chunk; c += 8) {           // 8 loads in flight, summed in chunk order
            float q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = part[(size_t)(c + u) * chunk_stride + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += q[u];
        }
        for (; c < n_chunk; ++c) s += part[(size_t)c * chunk_stride + o];
        out[o] = s;
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                   float a, size_t n)
{
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (size_t)gridDim.x * 256)
        y[o] += a * x[o];
}

// sum of squares / 2 of a tensor, one partial per block (tf.nn.l2_loss, DAEs.py:79-82)
__global__ __launch_bounds__(256) void l2_partial_kernel(const float* __restrict__ x, size_t n,
                                                         double* __restrict__ part)
{
    __shared__ double ws[4];
    double s = 0.0;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (size_t)gridDim.x * 256)
        s += (double)x[o] * (double)x[o];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = 0.5 * (ws[0] + ws[1] + ws[2] + ws[3]);
}

// cost = sum(loss partials) + lambda * sum(l2 partials), in double, fixed order (lane-strided sums, then
// a shuffle tree): one wave
__global__ __launch_bounds__(64) void finish_cost_kernel(const float* __restrict__ loss_part, int n_loss,
                                                         const double* __restrict__ l2_part, int n_l2, float lambda,
                                                         float* __restrict__ cost)
{
    const int lane = threadIdx.x;
    double s = 0.0, l2 = 0.0;
    for (int i = lane; i < n_loss; i += 64) s += (double)loss_part[i];
    for (int i = lane; i < n_l2; i += 64) l2 += l2_part[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { s += __shfl_xor(s, d); l2 += __shfl_xor(l2, d); }
    if (lane == 0) *cost = (float)(s + (double)lambda * l2);
}

// ---- K9: TF1 AdamOptimizer, dense ------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ m,
                                                   float* __restrict__ v,
                                                   const float* __restrict__ g, size_t n,
                                                   float lr_t, float b1, float b2, float eps)
{
    const size_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p); float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v); const float4* g4 = reinterpret_cast<const float4*>(g);
// TF1 ApplyAdam functor: m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); var -= (m alpha)/(sqrt(v)+eps)
#define ADAM1(P, M, V, G, c)                                         \
        M.c = M.c + (G.c - M.c) * (1.0f - b1);                       \
        V.c = V.c + (G.c * G.c - V.c) * (1.0f - b2);                 \
        P.c = P.c - (M.c * lr_t) / (sqrtf(V.c) + eps);
    // two float4 groups per iteration: 8 independent 16-byte loads in flight per thread (HBM-bound: 7 passes
    // over the tensor).  Every byte is touched once: nontemporal loads and stores keep the 7 streams out of each
    // other's way in L2.
    typedef float nt4 __attribute__((ext_vector_type(4)));
    auto ld = [](const float4* a) {
        const nt4 t = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(a));
        return make_float4(t.x, t.y, t.z, t.w);
    };
    auto st = [](float4* a, const float4 x) {
        const nt4 t = {x.x, x.y, x.z, x.w};
        __builtin_nontemporal_store(t, reinterpret_cast<nt4*>(a));
    };
    const size_t stride = (size_t)gridDim.x * 256;
    size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; o + stride < n4; o += 2 * stride) {
        const size_t o2 = o + stride;
        float4 pa = ld(p4 + o), ma = ld(m4 + o), va = ld(v4 + o);
        const float4 ga = ld(g4 + o);
        float4 pb = ld(p4 + o2), mb = ld(m4 + o2), vb = ld(v4 + o2);
        const float4 gb = ld(g4 + o2);
        ADAM1(pa, ma, va, ga, x) ADAM1(pa, ma, va, ga, y) ADAM1(pa, ma, va, ga, z) ADAM1(pa, ma, va, ga, w)
        ADAM1(pb, mb, vb, gb, x) ADAM1(pb, mb, vb, gb, y) ADAM1(pb, mb, vb, gb, z) ADAM1(pb, mb, vb, gb, w)
        st(p4 + o, pa); st(m4 + o, ma); st(v4 + o, va);
        st(p4 + o2, pb); st(m4 + o2, mb); st(v4 + o2, vb);
    }
    for (; o < n4; o += stride) {
        float4 pp = p4[o], mm = m4[o], vv = v4[o];
        const float4 gg = g4[o];
        ADAM1(pp, mm, vv, gg, x) ADAM1(pp, mm, vv, gg, y) ADAM1(pp, mm, vv, gg, z) ADAM1(pp, mm, vv, gg, w)
        p4[o] = pp; m4[o] = mm; v4[o] = vv;
    }
#undef ADAM1
    for (size_t o = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; o < n;
         o += (size_t)gridDim.x * 256) {
        const float gg = g[o];
        const float mm = m[o] + (gg - m[o]) * (1.0f - b1);
        const float vv = v[o] + (gg * gg - v[o]) * (1.0f - b2);
        m[o] = mm; v[o] = vv;
        p[o] = p[o] - (mm * lr_t) / (sqrtf(vv) + eps);
    }
}

// ---- K9 on a ROW-SPARSE gradient: the same dense TF1 Adam, without the HBM passes over rows that have none ------
// The untied encoder's gradient is non-zero on the few thousand rows the batch's input names (4 % of 170 000).
// Dense Adam still moves every row (m and v decay, p follows m), which costs 7 passes over 174 MB per step.  A row
// without gradient, however, evolves by a recurrence nobody else reads: its state can stay at the step it was last
// current for (`last[row]`) and be brought up to date -- by running the SAME per-element update with g = 0 once per
// missed step, with the alpha each of those steps used (lr_tab[s]) -- when the row is next needed: before a step
// whose input names it (dae_adam_rows_begin), or for everyone at a sync point (dae_adam_rows_flush).  Every element
// sees exactly the operation sequence dense Adam would have applied, so the parameters are bit-identical
// (tests/test_gpu_train.py); only the memory traffic of untouched rows is gone.
// One wave per listed row; a row listed several times (a track in many playlists) is claimed once per launch through
// mark[row] (atomicExch with a per-launch stamp).
#define ADAM_EL(P, M, V, G, A)                                           \
        M = M + (G - M) * (1.0f - b1);                                   \
        V = V + (G * G - V) * (1.0f - b2);                               \
        P = P - (M * A) / (sqrtf(V) + eps);

// MODE 0: begin  (listed rows -> current at step - 1)
// MODE 1: apply  (listed rows -> current at step - 1, then the update of `step` with their gradient row, which is
//                 zeroed again so that the dense gradient buffer stays all-zero between steps)
template <int MODE>
__global__ __launch_bounds__(256) void adam_rows_kernel(float* __restrict__ p, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ g,
                                                        int* __restrict__ last, int* __restrict__ mark,
                                                        float* __restrict__ lr_tab, int n_rows, int row_len,
                                                        const int32_t* __restrict__ rows,
                                                        const int32_t* __restrict__ n_listed_dev, int n_listed_max,
                                                        float lr_t, float b1, float b2, float eps, int step)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (MODE == 1 && blockIdx.x == 0 && threadIdx.x == 0) lr_tab[step] = lr_t;
    const int n_listed = n_listed_dev ? min(*n_listed_dev, n_listed_max) : n_listed_max;
    if (w >= n_listed) return;
    const int row = rows[w];
    if (row < 0 || row >= n_rows) return;
    const int stamp = 2 * step + MODE;
    // (round 6) Everything that depends on `row` alone is requested TOGETHER: the claim, the row's step and the first 64 float4 of
    // its state (the whole row at hidden 256) -- the wave was four dependent trips to memory (row, claim, step, state) for one
    // update, and a launch is ~100 such waves per CU: 27 + 30 us for 95 MB.  A wave that loses the claim has read lines the
    // winner reads anyway.
    const bool vec = (row_len & 3) == 0;
    const bool first_in = vec && lane < (row_len >> 2);
    const size_t o_first = ((size_t)row * row_len >> 2) + lane;
    float4 pp0 = make_float4(0.f, 0.f, 0.f, 0.f), mm0 = pp0, vv0 = pp0, gg0 = pp0;
    if (first_in) {
        pp0 = reinterpret_cast<float4*>(p)[o_first]; mm0 = reinterpret_cast<float4*>(m)[o_first];
        vv0 = reinterpret_cast<float4*>(v)[o_first];
        if (MODE == 1) gg0 = reinterpret_cast<float4*>(g)[o_first];
    }
    const int from = last[row];
    int claimed = 0;
    if (lane == 0) claimed = atomicExch(&mark[row], stamp) != stamp;
    claimed = __shfl(claimed, 0);
    if (!claimed) return;
    const int upto = step - 1;
    // four elements per lane at a time: the replay is a sequential recurrence per element (sqrt -> divide -> subtract),
    // so independent chains are the only instruction-level parallelism there is
    if (vec) {
        for (int c4 = lane; c4 < (row_len >> 2); c4 += 64) {
            const size_t o = ((size_t)row * row_len >> 2) + c4;
            const bool pre = c4 == lane;                           // the first round was requested above
            float4 pp = pre ? pp0 : reinterpret_cast<float4*>(p)[o], mm = pre ? mm0 : reinterpret_cast<float4*>(m)[o],
                   vv = pre ? vv0 : reinterpret_cast<float4*>(v)[o];
            for (int s_ = from + 1; s_ <= upto; ++s_) {
                const float a = lr_tab[s_];
                const float z = 0.0f;
                ADAM_EL(pp.x, mm.x, vv.x, z, a) ADAM_EL(pp.y, mm.y, vv.y, z, a)
                ADAM_EL(pp.z, mm.z, vv.z, z, a) ADAM_EL(pp.w, mm.w, vv.w, z, a)
            }
            if (MODE == 1) {
                const float4 gg = pre ? gg0 : reinterpret_cast<float4*>(g)[o];
                ADAM_EL(pp.x, mm.x, vv.x, gg.x, lr_t) ADAM_EL(pp.y, mm.y, vv.y, gg.y, lr_t)
                ADAM_EL(pp.z, mm.z, vv.z, gg.z, lr_t) ADAM_EL(pp.w, mm.w, vv.w, gg.w, lr_t)
                reinterpret_cast<float4*>(g)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            reinterpret_cast<float4*>(p)[o] = pp; reinterpret_cast<float4*>(m)[o] = mm;
            reinterpret_cast<float4*>(v)[o] = vv;
        }
    } else
    for (int c = lane; c < row_len; c += 64) {
        const size_t o = (size_t)row * row_len + c;
        float pp = p[o], mm = m[o], vv = v[o];
        for (int s_ = from + 1; s_ <= upto; ++s_) {
            const float a = lr_tab[s_];
            const float z = 0.0f;
            ADAM_EL(pp, mm, vv, z, a)
        }
        if (MODE == 1) {
            const float gg = g[o];
            ADAM_EL(pp, mm, vv, gg, lr_t)
            g[o] = 0.0f;
        }
        p[o] = pp; m[o] = mm; v[o] = vv;
    }
    if (lane == 0) last[row] = MODE == 1 ? step : upto;
}

// every row -> current at `step` (sync points: evaluation, saving, sharding, ...); one wave per row
__global__ __launch_bounds__(256) void adam_rows_flush_kernel(float* __restrict__ p, float* __restrict__ m,
                                                              float* __restrict__ v, int* __restrict__ last,
                                                              const float* __restrict__ lr_tab, int n_rows,
                                                              int row_len, float b1, float b2, float eps, int step)
{
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += gridDim.x * 4) {
        const int from = last[row];
        if (from >= step) continue;
        if ((row_len & 3) == 0) {
            for (int c4 = lane; c4 < (row_len >> 2); c4 += 64) {
                const size_t o = ((size_t)row * row_len >> 2) + c4;
                float4 pp = reinterpret_cast<float4*>(p)[o], mm = reinterpret_cast<float4*>(m)[o],
                       vv = reinterpret_cast<float4*>(v)[o];
                for (int s_ = from + 1; s_ <= step; ++s_) {
                    const float a = lr_tab[s_];
                    const float z = 0.0f;
                    ADAM_EL(pp.x, mm.x, vv.x, z, a) ADAM_EL(pp.y, mm.y, vv.y, z, a)
                    ADAM_EL(pp.z, mm.z, vv.z, z, a) ADAM_EL(pp.w, mm.w, vv.w, z, a)
                }
                reinterpret_cast<float4*>(p)[o] = pp; reinterpret_cast<float4*>(m)[o] = mm;
                reinterpret_cast<float4*>(v)[o] = vv;
            }
        } else
        for (int c = lane; c < row_len; c += 64) {
            const size_t o = (size_t)row * row_len + c;
            float pp = p[o], mm = m[o], vv = v[o];
            for (int s_ = from + 1; s_ <= step; ++s_) {
                const float a = lr_tab[s_];
                const float z = 0.0f;
                ADAM_EL(pp, mm, vv, z, a)
            }
            p[o] = pp; m[o] = mm; v[o] = vv;
        }
        if (lane == 0) last[row] = step;
    }
}
#undef ADAM_EL

int grid_for(size_t n) { size_t b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace

int dae_launch_adam(dae_ctx* ctx, float* param, float* m, float* v, const float* grad, int64_t n,
                    float lr_t, float beta1, float beta2, float eps)
{
    if (n <= 0) return DAE_OK;
    size_t work = (size_t)n / 4;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(work ? work : 1)), dim3(256), 0, ctx->stream, param, m, v,
                       grad, (size_t)n, lr_t, beta1, beta2, eps);
    DAE_CHECK_LAUNCH(ctx, "adam_kernel");
    return DAE_OK;
}

int dae_launch_adam_rows(dae_ctx* ctx, int mode, float* param, float* m, float* v, float* grad, int32_t* last,
                         int32_t* mark, float* lr_tab, int n_rows, int row_len, const int32_t* rows,
                         const int32_t* n_listed_dev, int n_listed_max, float lr_t, float beta1, float beta2,
                         float eps, int step)
{
    if (mode == 2) {
        int blocks = (n_rows + 3) / 4;
        if (blocks > 16 * DAE_NUM_CU) blocks = 16 * DAE_NUM_CU;
        hipLaunchKernelGGL(adam_rows_flush_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, ctx->stream, param, m, v,
                           last, lr_tab, n_rows, row_len, beta1, beta2, eps, step);
        DAE_CHECK_LAUNCH(ctx, "adam_rows_flush_kernel");
        return DAE_OK;
    }
    // mode 1 always launches: its first thread records this step's alpha even when no row is listed
    const int blocks = (n_listed_max + 3) / 4 > 0 ? (n_listed_max + 3) / 4 : 1;
    if (mode == 0)
        hipLaunchKernelGGL(adam_rows_kernel<0>, dim3(blocks), dim3(256), 0, ctx->stream, param, m, v, grad, last, mark,
                           lr_tab, n_rows, row_len, rows, n_listed_dev, n_listed_max, lr_t, beta1, beta2, eps, step);
    else
        hipLaunchKernelGGL(adam_rows_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, param, m, v, grad, last, mark,
                           lr_tab, n_rows, row_len, rows, n_listed_dev, n_listed_max, lr_t, beta1, beta2, eps, step);
    DAE_CHECK_LAUNCH(ctx, "adam_rows_kernel");
    return DAE_OK;
}

namespace {

// scratch carved for one training step over a [Vl, H] weight (shard) and B rows; stable for a given
// (Vl, H, B), so the stages of a sharded step find h / sg where the earlier stage left them
struct TrainPlan {
    int NA, G, RB, Bpad64, n_chunk, chunk, n_fix, dtype, dz16, rm;
    int fuse_dh;        // K5 leaves dh's partials itself (decode_f32.hip decode_loss_dh_bf16_kernel): no K7; n_chunk = g.grid + 1
    dae_rowgeom g;
    size_t bh, hp_bytes;
    float *dzT, *hbuf, *sg, *dpre, *part, *loss_part;
    double* l2_part;
};

int train_plan(dae_ctx* ctx, int Vl, int H, int B, TrainPlan& t)
{
    if ((H % 32) != 0) return dae_fail(ctx, DAE_ERR_ARG, "training kernels need H %% 32 == 0 (H=%d)", H);
    if (B < 1 || B > 256) return dae_fail(ctx, DAE_ERR_ARG, "training batch %d outside [1, 256]", B);
    if (Vl < 1) return dae_fail(ctx, DAE_ERR_ARG, "empty vocabulary shard");
    int rc;
    t.NA = (H % 128) == 0 ? 4 : ((H % 64) == 0 ? 2 : 1);
    const int Hp = dae_round_up(H, DAE_HPAD);
    t.dtype = ctx->train_dtype;
    {   // bf16 GEMMs with the 4-tile backward kernels: dL/dz itself is stored as bf16 (DAE_BWD_F32 / DAE_DZ_F32: A/B)
        static const bool dz_f32 = dae_exp_env("DAE_BWD_F32") != nullptr || dae_exp_env("DAE_DZ_F32") != nullptr;
        t.dz16 = (t.dtype == DAE_DTYPE_BF16 && (H % 128) == 0 && !dz_f32) ? 1 : 0;
    }
    t.g = t.dtype == DAE_DTYPE_BF16 ? dae_row_geometry_bf16(B, Hp) : dae_row_geometry(B, Hp);
    t.G = Hp / DAE_KG; t.RB = t.g.R_TILE / 32;
    {   // hidden 256: K5 reads the row-major decoder and hidden activations directly (fp32, or rounded to bf16 in
        // registers) -- no per-step prepack
        static const bool k5_packed = dae_exp_env("DAE_K5_PACKED") != nullptr;                 // A/B
        t.rm = (H == 256 && t.g.R_TILE == 128 && t.g.waves == 4 && !k5_packed) ? 1 : 0;
    }
    t.Bpad64 = (B + 63) / 64 * 64;
    t.hp_bytes = (size_t)t.g.n_rg * t.G * t.RB * 64 * sizeof(float4);
    if ((rc = dae_reserve(ctx, ctx->h_packed, t.hp_bytes))) return rc;
    if ((rc = dae_reserve(ctx, ctx->train_b, (size_t)Vl * t.Bpad64 * sizeof(float)))) return rc;
    // split of the V contraction of K7: about one (output tile, chunk) work item per wave slot
    const int n_out_tiles = (H / (32 * t.NA)) * (t.Bpad64 / 64);
    int want_chunks = (DAE_NUM_CU * 4) / n_out_tiles;
    if (want_chunks < 1) want_chunks = 1;
    t.chunk = ((Vl + want_chunks - 1) / want_chunks + 15) / 16 * 16;
    if (t.chunk < 16) t.chunk = 16;
    t.n_chunk = (Vl + t.chunk - 1) / t.chunk;
    {   // bf16 GEMMs with dz^T as bf16 at hidden 256: dh comes out of the forward launch, one partial per workgroup + the positives'
        static const bool no_fuse = dae_exp_env("DAE_K5_NOFUSE") != nullptr;                   // A/B: K5, then K7
        t.fuse_dh = (t.rm && t.dtype == DAE_DTYPE_BF16 && t.dz16 && !no_fuse) ? 1 : 0;
        if (t.fuse_dh) t.n_chunk = t.g.grid + 1;
    }
    t.bh = (size_t)B * H;
    t.n_fix = B;                                   // loss partials of the positives: one per row
    const size_t c_floats = 3 * t.bh + (size_t)t.n_chunk * t.Bpad64 * H + (size_t)t.g.grid + t.n_fix + 64;
    if ((rc = dae_reserve(ctx, ctx->train_c, c_floats * sizeof(float) + 4096 * sizeof(double)))) return rc;
    t.dzT = static_cast<float*>(ctx->train_b.p);
    t.hbuf = static_cast<float*>(ctx->train_c.p);
    t.sg = t.hbuf + t.bh;
    t.dpre = t.sg + t.bh;
    t.part = t.dpre + t.bh;
    t.loss_part = t.part + (size_t)t.n_chunk * t.Bpad64 * H;
    t.l2_part = reinterpret_cast<double*>(
        (reinterpret_cast<uintptr_t>(t.loss_part + t.g.grid + t.n_fix) + 63) & ~(uintptr_t)63);
    return DAE_OK;
}

// K5 loss/dz over the prepacked decoder image + the positives' fix-up -> K6 (gW, gb) -> K7 partials of dh.
// h (row-major in t.hbuf and tiled in ctx->h_packed) and ctx->pk_f32 must be current.
int train_decode_backward(dae_ctx* ctx, const TrainPlan& t, int Vl, int H, int B, int n_batch,
                          const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
                          int col_lo, int col_hi, const float* Wd, const float* b_dec, float* gWd, float* gb_dec)
{
    hipStream_t st = ctx->stream;
    const int NA = t.NA;
    int rc;
    if (H > FIX_MAXH) return dae_fail(ctx, DAE_ERR_ARG, "training kernels need H <= %d (H=%d)", FIX_MAXH, H);
    if (t.Bpad64 != B)
        DAE_HIP_CHECK(ctx, hipMemsetAsync(t.dzT, 0, (size_t)Vl * t.Bpad64 * (t.dz16 ? sizeof(unsigned short) : sizeof(float)), st));
    float* const corr_part = t.part + (size_t)(t.n_chunk - 1) * t.Bpad64 * H;         // (fuse_dh: the positives' partial, the last one)
    if (t.fuse_dh) {
        rc = dae_launch_decode_loss_dh(ctx, t.g, B, Vl, H, Wd, b_dec, t.hbuf, 1.0f / (float)n_batch, t.dzT, t.Bpad64, t.loss_part,
                                       t.part, t.Bpad64);
        if (rc == DAE_ERR_STATE) return dae_fail(ctx, DAE_ERR_STATE, "the fused K5 + K7 launch does not take this shape");
        if (t.Bpad64 != B) DAE_HIP_CHECK(ctx, hipMemsetAsync(corr_part, 0, (size_t)t.Bpad64 * H * sizeof(float), st));
    } else if (t.rm)
        rc = dae_launch_decode_loss_rowmajor(ctx, t.g, B, Vl, H, Wd, b_dec, t.hbuf, 1.0f / (float)n_batch, t.dzT, t.Bpad64,
                                             t.loss_part, t.dtype, t.dz16);
    else
        rc = dae_launch_decode_loss_f32(ctx, t.g, B, 1.0f / (float)n_batch, t.dzT, t.Bpad64, t.loss_part, t.dtype, t.dz16);
    if (rc) return rc;
    if (t.fuse_dh)
        hipLaunchKernelGGL((loss_fixup_kernel<true, true, true>), dim3(B), dim3(256), 0, st, y_row_ptr, y_col, y_val, B, H, col_lo,
                           col_hi, t.hbuf, Wd, b_dec, 1.0f / (float)n_batch, t.dzT, (int64_t)t.Bpad64,
                           t.loss_part + t.g.grid, corr_part);
    else if (t.dtype == DAE_DTYPE_BF16 && t.dz16)
        hipLaunchKernelGGL((loss_fixup_kernel<true, true>), dim3(B), dim3(256), 0, st, y_row_ptr, y_col, y_val, B, H, col_lo,
                           col_hi, t.hbuf, Wd, b_dec, 1.0f / (float)n_batch, t.dzT, (int64_t)t.Bpad64,
                           t.loss_part + t.g.grid);
    else if (t.dtype == DAE_DTYPE_BF16)
        hipLaunchKernelGGL(loss_fixup_kernel<true>, dim3(B), dim3(256), 0, st, y_row_ptr, y_col, y_val, B, H, col_lo,
                           col_hi, t.hbuf, Wd, b_dec, 1.0f / (float)n_batch, t.dzT, (int64_t)t.Bpad64,
                           t.loss_part + t.g.grid);
    else
        hipLaunchKernelGGL(loss_fixup_kernel<false>, dim3(B), dim3(256), 0, st, y_row_ptr, y_col, y_val, B, H, col_lo,
                           col_hi, t.hbuf, Wd, b_dec, 1.0f / (float)n_batch, t.dzT, (int64_t)t.Bpad64,
                           t.loss_part + t.g.grid);
    DAE_CHECK_LAUNCH(ctx, "loss_fixup_kernel");

    // ---- K6: decoder gradient ------------------------------------------------------------------------
    auto run_k6 = [&]() -> int {
        GwP p;
        p.ad_p = nullptr; p.ad_m = nullptr; p.ad_v = nullptr; p.ad_alpha = p.ad_b1 = p.ad_b2 = p.ad_eps = 0.0f;
        if (ctx->arm_m) {           // dae_arm_decoder_adam: update Wd in place instead of writing gWd
            p.ad_p = const_cast<float*>(Wd); p.ad_m = ctx->arm_m; p.ad_v = ctx->arm_v; p.ad_alpha = ctx->arm_alpha;
            p.ad_b1 = ctx->arm_b1; p.ad_b2 = ctx->arm_b2; p.ad_eps = ctx->arm_eps;
        }
        p.dzT = t.dzT; p.ldT = t.Bpad64; p.h = t.hbuf; p.H = H; p.B = B; p.V = Vl; p.gW = gWd; p.gb = gb_dec;
        p.accumulate = 0;
        static const int k6dbg = dae_exp_env("DAE_DBG_K6") ? atoi(dae_exp_env("DAE_DBG_K6")) : 0;
        p.dbg = k6dbg;
        p.n_half = H / (32 * NA);
        int nb = (DAE_NUM_CU / p.n_half) / DAE_NUM_XCD * DAE_NUM_XCD;
        if (nb < DAE_NUM_XCD) nb = DAE_NUM_XCD;
        p.nb_half = nb;
        const size_t lds = (size_t)((B + 31) & ~31) * 32 * NA * sizeof(float);
        static const char attr_key = 0;
        if (dae_first_use(ctx, &attr_key)) {
            DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_wdec_kernel<4>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        const dim3 grid(p.n_half * nb), blk(256);
        // two waves per SIMD on the shared h image: 241 us against 257 us with one (V = 170 000, B = H = 256); the
        // second wave covers the dz^T load latency and the gW stores of the first (DAE_K6_WAVES=4 for the A/B)
        static const bool k6w8 = !(dae_exp_env("DAE_K6_WAVES") && atoi(dae_exp_env("DAE_K6_WAVES")) == 4);
        static const bool bwd_f32 = dae_exp_env("DAE_BWD_F32") != nullptr;      // A/B: bf16 forward only
        if (NA == 4 && t.dtype == DAE_DTYPE_BF16 && !bwd_f32) {
            static const char attr16_key = 0;
            if (dae_first_use(ctx, &attr16_key)) {
                DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_wdec_kernel<4, 8, true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
            static const char attr16z_key = 0;
            if (dae_first_use(ctx, &attr16z_key)) {
                DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_wdec_kernel<4, 8, true, true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
            static const bool k6_old = dae_exp_env("DAE_K6_ORIENT") && !strcmp(dae_exp_env("DAE_K6_ORIENT"), "hidden");   // A/B
            if (t.dz16 && !k6_old) {
                const size_t lds_t = (size_t)(((B + 31) & ~31) >> 4) * 4 * 64 * sizeof(uint4);
                static const char k6t_key = 0;
                if (dae_first_use(ctx, &k6t_key))
                    DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_wdec_t_kernel<8>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                static const bool k6_t64 = dae_exp_env("DAE_K6_T64") != nullptr;                       // A/B: the 64-row tile form
                if (p.ad_m && !k6_t64) {
                    static const char attr_t32_key = 0;
                    if (dae_first_use(ctx, &attr_t32_key)) {
                        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_wdec_t32_kernel<8, true>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
                        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_wdec_t32_kernel<8, false>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
                    }
                    if (((B + 31) & ~31) == 256) hipLaunchKernelGGL((grad_wdec_t32_kernel<8, true>), grid, dim3(512), lds_t, st, p);
                    else hipLaunchKernelGGL((grad_wdec_t32_kernel<8, false>), grid, dim3(512), lds_t, st, p);
                } else
                if (((B + 31) & ~31) == 256) {
                    static const char attr_tf_key = 0;
                    if (dae_first_use(ctx, &attr_tf_key))
                        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_wdec_t_kernel<8, true>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
                    hipLaunchKernelGGL((grad_wdec_t_kernel<8, true>), grid, dim3(512), lds_t, st, p);
                } else
                hipLaunchKernelGGL((grad_wdec_t_kernel<8>), grid, dim3(512), lds_t, st, p);
            } else if (t.dz16) hipLaunchKernelGGL((grad_wdec_kernel<4, 8, true, true>), grid, dim3(512), lds, st, p);
            else hipLaunchKernelGGL((grad_wdec_kernel<4, 8, true>), grid, dim3(512), lds, st, p);
        } else if (NA == 4 && k6w8) {
            static const char attr8_key = 0;
            if (dae_first_use(ctx, &attr8_key)) {
                DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_wdec_kernel<4, 8>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
            static const bool k6_old32 = dae_exp_env("DAE_K6_ORIENT") && !strcmp(dae_exp_env("DAE_K6_ORIENT"), "hidden");   // A/B
            static const bool k6_generic = dae_exp_env("DAE_K6_GENERIC") != nullptr;                // A/B: the generic kernel below
            const int Bp32 = (B + 31) & ~31;
            if (p.ad_m && H == 256 && !k6_old32 && !k6_generic && (Bp32 & 15) == 0) {
                // (the ring walks the dz^T row four float4 at a time: whole groups of 16 playlists)
                const size_t lds_f = (size_t)(Bp32 >> 1) * 64 * sizeof(float4);
                static const char attr_tf32_key = 0;
                if (dae_first_use(ctx, &attr_tf32_key)) {
                    DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_wdec_t32_f32_kernel<8, true>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                    DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_wdec_t32_f32_kernel<8, false>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                }
                if (Bp32 == 256) hipLaunchKernelGGL((grad_wdec_t32_f32_kernel<8, true>), grid, dim3(512), lds_f, st, p);
                else hipLaunchKernelGGL((grad_wdec_t32_f32_kernel<8, false>), grid, dim3(512), lds_f, st, p);
            } else
            if (!k6_old32) {
                static const char attr8t_key = 0;
                if (dae_first_use(ctx, &attr8t_key))
                    DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_wdec_kernel<4, 8, false, false, true>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                hipLaunchKernelGGL((grad_wdec_kernel<4, 8, false, false, true>), grid, dim3(512), lds, st, p);
            } else hipLaunchKernelGGL((grad_wdec_kernel<4, 8>), grid, dim3(512), lds, st, p);
        } else if (NA == 4) hipLaunchKernelGGL(grad_wdec_kernel<4>, grid, blk, lds, st, p);
        else if (NA == 2) hipLaunchKernelGGL(grad_wdec_kernel<2>, grid, blk, lds, st, p);
        else hipLaunchKernelGGL(grad_wdec_kernel<1>, grid, blk, lds, st, p);
        DAE_CHECK_LAUNCH(ctx, "grad_wdec_kernel");
        return DAE_OK;
    };

    // ---- K7: dh, split over V ------------------------------------------------------------------------
    auto run_k7 = [&]() -> int {
        DhP p;
        p.dzT = t.dzT; p.ldT = t.Bpad64; p.W = Wd; p.H = H; p.V = Vl; p.part = t.part;
        p.n_chunk = t.n_chunk; p.chunk = t.chunk; p.Bpad64 = t.Bpad64; p.n_half = H / (32 * NA);
        p.n_rblk = t.Bpad64 / 64;
        p.fast32 = ((uint64_t)(Vl + 32) * (uint64_t)H * 4 < (1ull << 32) && (uint64_t)(Vl + 32) * (uint64_t)t.Bpad64 * 4 < (1ull << 32)) ? 1 : 0;
        const int total = p.n_half * p.n_rblk * t.n_chunk;
        int blocks = (total + 3) / 4;
        if (blocks > DAE_NUM_CU) blocks = DAE_NUM_CU;
        static const bool bwd_f32_7 = dae_exp_env("DAE_BWD_F32") != nullptr;
        if (NA == 4 && t.dtype == DAE_DTYPE_BF16 && !bwd_f32_7 && t.dz16)
            hipLaunchKernelGGL((grad_hidden_kernel<4, true, true>), dim3(blocks), dim3(256), 0, st, p);
        else if (NA == 4 && t.dtype == DAE_DTYPE_BF16 && !bwd_f32_7)
            hipLaunchKernelGGL((grad_hidden_kernel<4, true>), dim3(blocks), dim3(256), 0, st, p);
        else if (NA == 4) hipLaunchKernelGGL(grad_hidden_kernel<4>, dim3(blocks), dim3(256), 0, st, p);
        else if (NA == 2) hipLaunchKernelGGL(grad_hidden_kernel<2>, dim3(blocks), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(grad_hidden_kernel<1>, dim3(blocks), dim3(256), 0, st, p);
        DAE_CHECK_LAUNCH(ctx, "grad_hidden_kernel");
        return DAE_OK;
    };
    // K6 and K7 are independent (both read dz^T).  With the armed Adam K6 rewrites Wd in place, and K7 multiplies by
    // the weights the forward pass used: K7 first.
    if (ctx->arm_m) {
        if (NA != 4) { ctx->arm_m = nullptr; return dae_fail(ctx, DAE_ERR_ARG, "the armed decoder Adam needs H %% 128 == 0 (H=%d)", H); }
        if (!t.fuse_dh) { rc = run_k7(); if (rc) return rc; }
        rc = run_k6();
        ctx->arm_m = nullptr; ctx->arm_v = nullptr;         // one step only
        return rc;
    }
    rc = run_k6(); if (rc) return rc;
    return t.fuse_dh ? DAE_OK : run_k7();
}

// cost = sum of the loss partials + lambda * (l2 of the listed tensors)
int train_cost(dae_ctx* ctx, const TrainPlan& t, float reg_lambda, const float* const* ts,
               const size_t* ns, int n_t, float* cost_out)
{
    hipStream_t st = ctx->stream;
    int n_l2 = 0;
    if (reg_lambda != 0.0f) {
        for (int i = 0; i < n_t; ++i) {
            if (!ts[i] || ns[i] == 0) continue;
            const int nb = grid_for(ns[i]) > 1024 ? 1024 : grid_for(ns[i]);
            hipLaunchKernelGGL(l2_partial_kernel, dim3(nb), dim3(256), 0, st, ts[i], ns[i], t.l2_part + n_l2);
            DAE_CHECK_LAUNCH(ctx, "l2_partial_kernel");
            n_l2 += nb;
        }
    }
    hipLaunchKernelGGL(finish_cost_kernel, dim3(1), dim3(64), 0, st, t.loss_part, t.g.grid + t.n_fix, t.l2_part, n_l2,
                       reg_lambda, cost_out);
    DAE_CHECK_LAUNCH(ctx, "finish_cost_kernel");
    return DAE_OK;
}

// dpre from dh (n_chunk partials at `part`), gb_enc, the row-sparse gW_enc of this column range,
// and the lambda terms of the weight gradients
int train_encoder_backward(dae_ctx* ctx, const TrainPlan& t, const float* part, int n_chunk,
                           const int32_t* x_row_ptr, const int32_t* x_col, const float* x_val,
                           int col_lo, int col_hi, int H, int B, int tied, float ikp, float kp,
                           uint32_t seed, float reg_lambda, const float* W_enc, const float* b_enc,
                           const float* W_dec, const float* b_dec,
                           float* gW_enc, float* gb_enc, float* gW_dec, float* gb_dec)
{
    hipStream_t st = ctx->stream;
    const size_t nW = (size_t)(col_hi - col_lo) * H;
    hipLaunchKernelGGL(hidden_backward_kernel, dim3(grid_for(t.bh)), dim3(256), 0, st, part, n_chunk,
                       t.Bpad64, H, B, t.hbuf, t.sg, kp, t.dpre);
    DAE_CHECK_LAUNCH(ctx, "hidden_backward_kernel");
    hipLaunchKernelGGL(colsum_kernel, dim3((H + 63) / 64), dim3(256), 0, st, t.dpre, B, H, reg_lambda,
                       b_enc, gb_enc);
    DAE_CHECK_LAUNCH(ctx, "colsum_kernel");

    // ---- K8: encoder gradient (row-sparse) -----------------------------------------------------------
    // (the rows-Adam keeps the dense buffer all-zero itself: dae_set_enc_grad_prezeroed)
    if (!tied && !ctx->enc_grad_prezeroed) DAE_HIP_CHECK(ctx, hipMemsetAsync(gW_enc, 0, nW * sizeof(float), st));
    hipLaunchKernelGGL(scatter_gwenc_kernel, dim3(B), dim3(256), 0, st, x_row_ptr, x_col, x_val, B, H,
                       ikp, seed, col_lo, col_hi, t.dpre, gW_enc);
    DAE_CHECK_LAUNCH(ctx, "scatter_gwenc_kernel");

    if (reg_lambda != 0.0f) {
        hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(nW)), dim3(256), 0, st, gW_enc, W_enc, reg_lambda, nW);
        if (!tied)
            hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(nW)), dim3(256), 0, st, gW_dec, W_dec, reg_lambda, nW);
        hipLaunchKernelGGL(axpy_kernel, dim3(grid_for((size_t)(col_hi - col_lo))), dim3(256), 0, st, gb_dec,
                           b_dec, reg_lambda, (size_t)(col_hi - col_lo));
        DAE_CHECK_LAUNCH(ctx, "axpy_kernel");
    }
    return DAE_OK;
}

}  // namespace

// ---- the two backward GEMMs on caller-provided buffers (also used by the title scorer's output layer) ---
// gW[v, :] = sum_r dzT[v, r] h[r, :]  and  gb[v] = sum_r dzT[v, r]   (H % 32 == 0, B <= 256)
int dae_launch_grad_w(dae_ctx* ctx, const float* dzT, int64_t ldT, const float* h, int H, int B, int V,
                      float* gW, float* gb)
{
    if ((H % 32) != 0 || B < 1 || B > 256) return dae_fail(ctx, DAE_ERR_ARG, "grad_w: H=%d B=%d unsupported", H, B);
    const int NA = (H % 128) == 0 ? 4 : ((H % 64) == 0 ? 2 : 1);
    GwP p;
    p.ad_p = nullptr; p.ad_m = nullptr; p.ad_v = nullptr; p.ad_alpha = p.ad_b1 = p.ad_b2 = p.ad_eps = 0.0f;
    p.dzT = dzT; p.ldT = ldT; p.h = h; p.H = H; p.B = B; p.V = V; p.gW = gW; p.gb = gb;
    p.accumulate = 0; p.dbg = 0;
    p.n_half = H / (32 * NA);
    int nb = (DAE_NUM_CU / p.n_half) / DAE_NUM_XCD * DAE_NUM_XCD;
    if (nb < DAE_NUM_XCD) nb = DAE_NUM_XCD;
    p.nb_half = nb;
    const size_t lds = (size_t)((B + 31) & ~31) * 32 * NA * sizeof(float);
    static const char attr_key = 0;
    if (dae_first_use(ctx, &attr_key)) {
        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_wdec_kernel<4>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const dim3 grid(p.n_half * nb), blk(256);
    if (NA == 4) hipLaunchKernelGGL(grad_wdec_kernel<4>, grid, blk, lds, ctx->stream, p);
    else if (NA == 2) hipLaunchKernelGGL(grad_wdec_kernel<2>, grid, blk, lds, ctx->stream, p);
    else hipLaunchKernelGGL(grad_wdec_kernel<1>, grid, blk, lds, ctx->stream, p);
    DAE_CHECK_LAUNCH(ctx, "grad_wdec_kernel");
    return DAE_OK;
}

// dh[r, :] = sum_v dzT[v, r] W[v, :]  (split over V into ctx scratch, reduced in fixed order)
int dae_launch_grad_h(dae_ctx* ctx, const float* dzT, int64_t ldT, const float* W, int H, int V, int B, float* dh)
{
    if ((H % 32) != 0 || B < 1 || B > 256) return dae_fail(ctx, DAE_ERR_ARG, "grad_h: H=%d B=%d unsupported", H, B);
    const int NA = (H % 128) == 0 ? 4 : ((H % 64) == 0 ? 2 : 1);
    const int Bpad64 = (B + 63) / 64 * 64;
    if (ldT < Bpad64) return dae_fail(ctx, DAE_ERR_ARG, "grad_h: ldT=%lld < %d", (long long)ldT, Bpad64);
    const int n_out_tiles = (H / (32 * NA)) * (Bpad64 / 64);
    int want_chunks = (DAE_NUM_CU * 4) / n_out_tiles;
    if (want_chunks < 1) want_chunks = 1;
    int chunk = ((V + want_chunks - 1) / want_chunks + 15) / 16 * 16;
    if (chunk < 16) chunk = 16;
    const int n_chunk = (V + chunk - 1) / chunk;
    int rc = dae_reserve(ctx, ctx->train_d, (size_t)n_chunk * Bpad64 * H * sizeof(float));
    if (rc) return rc;
    float* part = static_cast<float*>(ctx->train_d.p);
    DhP p;
    p.dzT = dzT; p.ldT = ldT; p.W = W; p.H = H; p.V = V; p.part = part;
    p.n_chunk = n_chunk; p.chunk = chunk; p.Bpad64 = Bpad64; p.n_half = H / (32 * NA);
    p.n_rblk = Bpad64 / 64;
    p.fast32 = ((uint64_t)(V + 32) * (uint64_t)H * 4 < (1ull << 32) && (uint64_t)(V + 32) * (uint64_t)ldT * 4 < (1ull << 32)) ? 1 : 0;
    const int total = p.n_half * p.n_rblk * n_chunk;
    int blocks = (total + 3) / 4;
    if (blocks > DAE_NUM_CU) blocks = DAE_NUM_CU;
    if (NA == 4) hipLaunchKernelGGL(grad_hidden_kernel<4>, dim3(blocks), dim3(256), 0, ctx->stream, p);
    else if (NA == 2) hipLaunchKernelGGL(grad_hidden_kernel<2>, dim3(blocks), dim3(256), 0, ctx->stream, p);
    else hipLaunchKernelGGL(grad_hidden_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, p);
    DAE_CHECK_LAUNCH(ctx, "grad_hidden_kernel");
    const size_t bh = (size_t)B * H;
    hipLaunchKernelGGL(sum_chunks_kernel, dim3(grid_for(bh)), dim3(256), 0, ctx->stream, part, n_chunk,
                       (size_t)Bpad64 * H, bh, dh);
    DAE_CHECK_LAUNCH(ctx, "sum_chunks_kernel");
    return DAE_OK;
}

int dae_train_step_f32(dae_ctx* ctx,
        const int32_t* x_row_ptr, const int32_t* x_col, const float* x_val,
        const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
        const float* W_enc, const float* b_enc, const float* W_dec, const float* b_dec,
        int V, int H, int B, int n_batch, int tied,
        float ikp, float kp, uint32_t seed, float reg_lambda,
        float* gW_enc, float* gb_enc, float* gW_dec, float* gb_dec, float* cost_out)
{
    hipStream_t st = ctx->stream;
    TrainPlan t;
    int rc = train_plan(ctx, V, H, B, t);
    if (rc) return rc;
    if (ctx->arm_m && (tied || reg_lambda != 0.0f)) {
        ctx->arm_m = nullptr; ctx->arm_v = nullptr;
        return dae_fail(ctx, DAE_ERR_ARG, "the armed decoder Adam needs the untied model and reg_lambda = 0");
    }
    const float* Wd = tied ? W_enc : W_dec;
    // decoder weights change every step: re-tile them for the forward GEMM
    if (!t.rm) {
        rc = t.dtype == DAE_DTYPE_BF16 ? dae_launch_prepack_bf16(ctx, Wd, b_dec, V, H, 0, V)
                                       : dae_launch_prepack_f32(ctx, Wd, b_dec, V, H, 0, V);
        if (rc) return rc;
    }

    // ---- forward ----------------------------------------------------------------------------------
    // the pad rows / pad k of the fp32 image are never written by the encode kernel: zeroed once per geometry and
    // buffer (same key as the scoring path: the two share the image)
    if (t.dtype == DAE_DTYPE_F32 && !t.rm) {
        const long long key = ((long long)B << 32) | ((long long)H << 12) | (long long)t.g.R_TILE;
        if (ctx->h_geom_key != key || ctx->h_geom_ptr != ctx->h_packed.p) {
            DAE_HIP_CHECK(ctx, hipMemsetAsync(ctx->h_packed.p, 0, t.hp_bytes, st));
            ctx->h_geom_key = key;
            ctx->h_geom_ptr = ctx->h_packed.p;
        }
    }
    if (t.rm) {
        rc = dae_launch_encode(ctx, x_row_ptr, x_col, x_val, W_enc, b_enc, V, H, B, ikp, kp, seed, t.hbuf,
                               nullptr, 0, 0, t.sg, nullptr);
    } else if (t.dtype == DAE_DTYPE_BF16) {
        rc = dae_launch_encode(ctx, x_row_ptr, x_col, x_val, W_enc, b_enc, V, H, B, ikp, kp, seed, t.hbuf,
                               nullptr, 0, 0, t.sg, nullptr);
        if (rc) return rc;
        rc = dae_launch_pack_h_bf16(ctx, t.hbuf, B, H, t.g);
    } else {
        rc = dae_launch_encode(ctx, x_row_ptr, x_col, x_val, W_enc, b_enc, V, H, B, ikp, kp, seed, t.hbuf,
                               static_cast<float*>(ctx->h_packed.p), t.G, t.RB, t.sg, nullptr);
    }
    if (rc) return rc;
    rc = train_decode_backward(ctx, t, V, H, B, n_batch, y_row_ptr, y_col, y_val, 0, V, Wd, b_dec,
                               tied ? gW_enc : gW_dec, gb_dec);
    if (rc) return rc;

    // ---- cost (+ lambda * l2) ----------------------------------------------------------------------
    const float* ts[4] = {W_enc, b_dec, b_enc, tied ? nullptr : W_dec};
    const size_t ns[4] = {(size_t)V * H, (size_t)V, (size_t)H, (size_t)V * H};
    rc = train_cost(ctx, t, reg_lambda, ts, ns, 4, cost_out);
    if (rc) return rc;

    rc = train_encoder_backward(ctx, t, t.part, t.n_chunk, x_row_ptr, x_col, x_val, 0, V, H, B, tied,
                                ikp, kp, seed, reg_lambda, W_enc, b_enc, W_dec, b_dec,
                                gW_enc, gb_enc, gW_dec, gb_dec);
    if (rc) return rc;
    if (!t.rm) (t.dtype == DAE_DTYPE_BF16 ? ctx->pk_bf16 : ctx->pk_f32).valid = true;
    return DAE_OK;
}

// ---- vocabulary-row sharded step: three stages around the caller's two all-reduces ------
int dae_train_shard_encode_f32(dae_ctx* ctx, const int32_t* x_row_ptr, const int32_t* x_col,
                               const float* x_val, const float* W_enc_loc, int col_lo, int col_hi,
                               int H, int B, float ikp, uint32_t seed, float* pre_partial)
{
    hipLaunchKernelGGL(encode_partial_kernel, dim3(B), dim3(256), 0, ctx->stream, x_row_ptr, x_col, x_val,
                       B, H, ikp, seed, col_lo, col_hi, W_enc_loc, pre_partial);
    DAE_CHECK_LAUNCH(ctx, "encode_partial_kernel");
    return DAE_OK;
}

int dae_train_shard_decode_f32(dae_ctx* ctx, const float* pre, const float* b_enc,
                               const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
                               const float* W_enc_loc, const float* W_dec_loc, const float* b_dec_loc,
                               int col_lo, int col_hi, int H, int B, int n_batch, int tied,
                               float kp, uint32_t seed, float reg_lambda,
                               float* gW_out, float* gb_dec_loc, float* dh_partial, float* cost_partial)
{
    hipStream_t st = ctx->stream;
    const int Vl = col_hi - col_lo;
    TrainPlan t;
    int rc = train_plan(ctx, Vl, H, B, t);
    if (rc) return rc;
    const float* Wd = tied ? W_enc_loc : W_dec_loc;
    if (!t.rm) {
        rc = t.dtype == DAE_DTYPE_BF16 ? dae_launch_prepack_bf16(ctx, Wd, b_dec_loc, Vl, H, 0, Vl)
                                       : dae_launch_prepack_f32(ctx, Wd, b_dec_loc, Vl, H, 0, Vl);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(activate_kernel, dim3(grid_for(t.bh)), dim3(256), 0, st, pre, b_enc, B, H, kp, seed,
                       t.hbuf, t.sg);
    DAE_CHECK_LAUNCH(ctx, "activate_kernel");
    ctx->h_geom_key = -1;
    if (!t.rm) {
        rc = t.dtype == DAE_DTYPE_BF16 ? dae_launch_pack_h_bf16(ctx, t.hbuf, B, H, t.g)
                                       : dae_launch_pack_h(ctx, t.hbuf, B, H, t.g);
        if (rc) return rc;
    }
    rc = train_decode_backward(ctx, t, Vl, H, B, n_batch, y_row_ptr, y_col, y_val, col_lo, col_hi, Wd, b_dec_loc,
                               gW_out, gb_dec_loc);
    if (rc) return rc;
    // b_enc is replicated: its l2 term is counted once, by the shard that owns column 0
    const float* ts[4] = {W_enc_loc, b_dec_loc, col_lo == 0 ? b_enc : nullptr, tied ? nullptr : W_dec_loc};
    const size_t ns[4] = {(size_t)Vl * H, (size_t)Vl, (size_t)H, (size_t)Vl * H};
    rc = train_cost(ctx, t, reg_lambda, ts, ns, 4, cost_partial);
    if (rc) return rc;
    hipLaunchKernelGGL(sum_chunks_kernel, dim3(grid_for(t.bh)), dim3(256), 0, st, t.part, t.n_chunk,
                       (size_t)t.Bpad64 * H, t.bh, dh_partial);
    DAE_CHECK_LAUNCH(ctx, "sum_chunks_kernel");
    if (!t.rm) (t.dtype == DAE_DTYPE_BF16 ? ctx->pk_bf16 : ctx->pk_f32).valid = true;
    return DAE_OK;
}

int dae_train_shard_finish_f32(dae_ctx* ctx, const float* dh, const int32_t* x_row_ptr,
                               const int32_t* x_col, const float* x_val,
                               const float* W_enc_loc, const float* b_enc, const float* W_dec_loc,
                               const float* b_dec_loc, int col_lo, int col_hi, int H, int B, int tied,
                               float ikp, float kp, uint32_t seed, float reg_lambda,
                               float* gW_enc_loc, float* gb_enc, float* gW_dec_loc, float* gb_dec_loc)
{
    TrainPlan t;
    int rc = train_plan(ctx, col_hi - col_lo, H, B, t);       // same carving as the decode stage
    if (rc) return rc;
    return train_encoder_backward(ctx, t, dh, 1, x_row_ptr, x_col, x_val, col_lo, col_hi, H, B, tied,
                                  ikp, kp, seed, reg_lambda, W_enc_loc, b_enc, W_dec_loc, b_dec_loc,
                                  gW_enc_loc, gb_enc, gW_dec_loc, gb_dec_loc);
}
