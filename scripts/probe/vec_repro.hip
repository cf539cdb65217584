// vec_repro.hip -- reduced reproducer of round 4's "loop vectorizer miscompile" (profiles/r04_notes.md 11d, VERDICT r4 Weak #4):
// the per-lane staging loop of mix_refine_kernel as it stood at commit d614ea5, with and without
// `#pragma clang loop vectorize(disable) interleave(disable)`, checked against a one-thread recomputation of every staged key.
// Build + run: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math scripts/probe/vec_repro.hip -o vec_repro && ./vec_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../spotify_recsys_challenge_2018_amd/csrc/dae_internal.h"

constexpr int STAGE = 8192;
__device__ __forceinline__ float mixf(float zt, float zd, float wt, float wp) { return dae_sigmoidf(zt) * wt + dae_sigmoidf(zd) * wp; }
__device__ __forceinline__ float widen_up(float y) { return fmaf(y, 0x1p-20f, y); }
__device__ __forceinline__ float widen_dn(float y) { return fmaf(y, -0x1p-20f, y); }
__device__ __forceinline__ float two_down(float x) { return dae_okey_inv(dae_okey(x) - 2u); }
__device__ __forceinline__ void keys_of(const uint4 en, const float* al, const float* be, const float* ep, float F, float wt, float wp,
                                        unsigned& a, unsigned& u)
{
    const int col = (int)en.z;
    const float uT = __uint_as_float(en.x), uD = __uint_as_float(en.y);
    const float wdT = 2.0f * fmaf(al[col], F, be[col]) * 1.000001f, wdD = 2.0f * ep[col] * 1.000001f;
    u = dae_okey(widen_up(mixf(uT, uD, wt, wp)));
    a = dae_okey(widen_dn(mixf(two_down(uT - wdT), two_down(uD - wdD), wt, wp)));
}
template <bool PRAGMA>
__global__ __launch_bounds__(1024) void stage_kernel(const uint4* base, const int* seg_prefix, int nseg, const float* al, const float* be,
                                                     const float* ep, float F, float wt, float wp, unsigned* out_l, unsigned* out_u)
{
    extern __shared__ __attribute__((aligned(16))) unsigned mr_dyn[];
    unsigned* kl = mr_dyn; unsigned* ku = mr_dyn + STAGE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned kmx = 0u, kmn = 0xFFFFFFFFu;
    for (int sg = wave; sg < nseg; sg += 16) {
        const int b0 = seg_prefix[sg], cn = seg_prefix[sg + 1] - b0;
        const uint4* sp = base + b0;
        if (PRAGMA) {
#pragma clang loop vectorize(disable) interleave(disable)
            for (int i = lane; i < cn; i += 64) {
                unsigned a, u; keys_of(sp[i], al, be, ep, F, wt, wp, a, u);
                kl[b0 + i] = a; ku[b0 + i] = u; kmx = a > kmx ? a : kmx; kmn = a < kmn ? a : kmn;
            }
        } else {
            for (int i = lane; i < cn; i += 64) {
                unsigned a, u; keys_of(sp[i], al, be, ep, F, wt, wp, a, u);
                kl[b0 + i] = a; ku[b0 + i] = u; kmx = a > kmx ? a : kmx; kmn = a < kmn ? a : kmn;
            }
        }
    }
    __syncthreads();
    const int total = seg_prefix[nseg];
    for (int i = threadIdx.x; i < total; i += 1024) { out_l[i] = kl[i]; out_u[i] = ku[i]; }
    if (kmx == 12345u && kmn == 7u) out_l[0] = 0;                 // (keep the reductions alive)
}
__global__ void ref_kernel(const uint4* base, int total, const float* al, const float* be, const float* ep, float F, float wt, float wp,
                           unsigned* out_l, unsigned* out_u)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int i = 0; i < total; ++i) keys_of(base[i], al, be, ep, F, wt, wp, out_l[i], out_u[i]);
}
int main()
{
    const int nseg = 32, ncol = 4096;
    std::vector<int> pre(nseg + 1, 0);
    for (int s = 0; s < nseg; ++s) pre[s + 1] = pre[s] + (s % 3 == 0 ? 200 : s % 3 == 1 ? 64 : 97);      // segments of more than 64 entries among them
    const int total = pre[nseg];
    std::vector<uint4> en(total);
    std::vector<float> al(ncol), be(ncol), ep(ncol);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (st >> 8) * (1.0f / 16777216.0f); };
    for (int c = 0; c < ncol; ++c) { al[c] = 1e-3f * rnd(); be[c] = 1e-3f * rnd(); ep[c] = 2e-3f * rnd(); }
    for (int i = 0; i < total; ++i) { float a = 6.0f * rnd() - 3.0f, b = 8.0f * rnd() - 6.0f; en[i] = make_uint4(*(unsigned*)&a, *(unsigned*)&b, (unsigned)(rnd() * ncol) % ncol, 0u); }
    uint4* d_en; int* d_pre; float *d_al, *d_be, *d_ep; unsigned* d_out;
    hipMalloc(&d_en, total * sizeof(uint4)); hipMalloc(&d_pre, (nseg + 1) * 4); hipMalloc(&d_al, ncol * 4); hipMalloc(&d_be, ncol * 4);
    hipMalloc(&d_ep, ncol * 4); hipMalloc(&d_out, 6 * total * 4);
    hipMemcpy(d_en, en.data(), total * sizeof(uint4), hipMemcpyHostToDevice); hipMemcpy(d_pre, pre.data(), (nseg + 1) * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_al, al.data(), ncol * 4, hipMemcpyHostToDevice); hipMemcpy(d_be, be.data(), ncol * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_ep, ep.data(), ncol * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE * 4);
    const float F = 1.5f, wt = 0.4f, wp = 0.6f;
    hipLaunchKernelGGL(ref_kernel, dim3(1), dim3(64), 0, 0, d_en, total, d_al, d_be, d_ep, F, wt, wp, d_out, d_out + total);
    hipLaunchKernelGGL(stage_kernel<true>, dim3(1), dim3(1024), 2 * STAGE * 4, 0, d_en, d_pre, nseg, d_al, d_be, d_ep, F, wt, wp, d_out + 2 * total, d_out + 3 * total);
    hipLaunchKernelGGL(stage_kernel<false>, dim3(1), dim3(1024), 2 * STAGE * 4, 0, d_en, d_pre, nseg, d_al, d_be, d_ep, F, wt, wp, d_out + 4 * total, d_out + 5 * total);
    std::vector<unsigned> h(6 * total);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, 6 * total * 4, hipMemcpyDeviceToHost);
    int bad[4] = {0, 0, 0, 0};
    for (int i = 0; i < total; ++i) {
        bad[0] += h[2 * total + i] != h[i]; bad[1] += h[3 * total + i] != h[total + i];
        bad[2] += h[4 * total + i] != h[i]; bad[3] += h[5 * total + i] != h[total + i];
    }
    for (int i = 0, n = 0; i < total && n < 6; ++i)
        if (h[4 * total + i] != h[i]) {
            int sg = 0; while (pre[sg + 1] <= i) ++sg;
            int from = -1;                                        // whose key is it?
            for (int j2 = pre[sg]; j2 < pre[sg + 1]; ++j2) if (h[j2] == h[4 * total + i]) from = j2 - pre[sg];
            printf("  entry %d (segment %d, local %d of %d): got %08x want %08x -- the key of local entry %d\n", i, sg, i - pre[sg],
                   pre[sg + 1] - pre[sg], h[4 * total + i], h[i], from);
            ++n;
        }
    printf("entries %d | pragma: lower-bound keys wrong %d, upper %d | no pragma: lower %d, upper %d\n", total, bad[0], bad[1], bad[2], bad[3]);
    return (bad[2] || bad[3]) ? 1 : 0;
}
