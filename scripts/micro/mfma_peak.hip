// micro-benchmark: what one SIMD sustains on v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32 with
// 1 and 2 waves per SIMD and 4 / 8 independent accumulators (no memory traffic at all).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o gpurun_out/mfma_peak && ./gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int BF>
__global__ void k(float* out, int iters)
{
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    uint4 ua = make_uint4(threadIdx.x, 1, 2, 3), ub = make_uint4(4, 5, threadIdx.x, 7);
    float fa = threadIdx.x * 1e-3f, fb = 2e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            if (BF) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    if (s == 12345.f) out[0] = s;
}

template <int NACC, int BF>
void run(const char* name, int threads, int iters)
{
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;
    hipLaunchKernelGGL((k<NACC, BF>), dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, BF>), dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * NACC * (BF ? 32768.0 : 4096.0);
    printf("%-40s threads=%d  %.3f ms  %.1f TFLOP/s\n", name, threads, ms, flop / ms / 1e9);
    hipFree(d);
}

int main()
{
    run<8, 1>("bf16 32x32x16, 8 acc", 256, 20000);
    run<8, 1>("bf16 32x32x16, 8 acc", 512, 10000);
    run<4, 1>("bf16 32x32x16, 4 acc", 256, 40000);
    run<4, 1>("bf16 32x32x16, 4 acc", 512, 20000);
    run<2, 1>("bf16 32x32x16, 2 acc", 512, 40000);
    run<4, 0>("f32 32x32x2, 4 acc", 256, 20000);
    run<4, 0>("f32 32x32x2, 4 acc", 512, 10000);
    return 0;
}
