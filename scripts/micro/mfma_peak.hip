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

// the decode kernels' inner pattern: every step reads NB fresh B fragments from LDS (double buffered, one
// step ahead) and issues NA x NB MFMAs on them; no global memory.  NA = A fragments (column tiles) per wave.
template <int NA, int NB, int INTERLEAVE>
__global__ void k_lds(float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    for (int i = threadIdx.x; i < 16 * NB * 64; i += blockDim.x) lds[i] = make_uint4(i, 1, 2, 3);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[NA][NB];
    for (int a = 0; a < NA; ++a) for (int b = 0; b < NB; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    uint4 ua[NA];
    for (int a = 0; a < NA; ++a) ua[a] = make_uint4(threadIdx.x, a, 2, 3);
    uint4 cb[2][NB];
    for (int b = 0; b < NB; ++b) cb[0][b] = lds[b * 64 + lane];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int sn = (s + 1) & 15;
            if (!INTERLEAVE) {
#pragma unroll
                for (int b = 0; b < NB; ++b) cb[(s + 1) & 1][b] = lds[(sn * NB + b) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua[a]),
                                                                        __builtin_bit_cast(bf16x8, cb[s & 1][b]), acc[a][b], 0, 0, 0);
                    if (INTERLEAVE) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (a == 0) cb[(s + 1) & 1][b] = lds[(sn * NB + b) * 64 + lane];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = 0.f;
    for (int a = 0; a < NA; ++a) for (int b = 0; b < NB; ++b) for (int e = 0; e < 16; ++e) sum += acc[a][b][e];
    if (sum == 12345.f) out[0] = sum;
}

template <int NA, int NB, int INTERLEAVE>
void run_lds(const char* name, int threads, int iters)
{
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256;
    const size_t lds = (size_t)16 * NB * 64 * sizeof(uint4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds<NA, NB, INTERLEAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k_lds<NA, NB, INTERLEAVE>), dim3(blocks), dim3(threads), lds, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_lds<NA, NB, INTERLEAVE>), dim3(blocks), dim3(threads), lds, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * 16.0 * NA * NB * 32768.0;
    printf("%-52s threads=%d  %.3f ms  %.1f TFLOP/s\n", name, threads, ms, flop / ms / 1e9);
    hipFree(d);
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
    run_lds<1, 4, 0>("bf16 + LDS B frags: 1 tile x 4 row blocks, burst", 256, 2000);
    run_lds<1, 4, 0>("bf16 + LDS B frags: 1 tile x 4 row blocks, burst", 512, 1000);
    run_lds<2, 4, 0>("bf16 + LDS B frags: 2 tiles x 4 row blocks, burst", 256, 1000);
    run_lds<2, 4, 1>("bf16 + LDS B frags: 2 tiles x 4 row blocks, interleaved", 256, 1000);
    run_lds<2, 4, 1>("bf16 + LDS B frags: 2 tiles x 4 row blocks, interleaved", 512, 500);
    run_lds<1, 8, 1>("bf16 + LDS B frags: 1 tile x 8 row blocks, interleaved", 256, 1000);
    run_lds<4, 2, 1>("bf16 + LDS B frags: 4 tiles x 2 row blocks, interleaved", 256, 1000);
    run<4, 0>("f32 32x32x2, 4 acc", 256, 20000);
    run<4, 0>("f32 32x32x2, 4 acc", 512, 10000);
    return 0;
}
