// experiment tool (scripts/coresidency_probe.py): spin kernels that claim a given number of VGPRs / LDS bytes
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int R>
__global__ __launch_bounds__(256) void spin_kernel(float* out, int iters, int prio)
{
    if (prio) __builtin_amdgcn_s_setprio(3);
    extern __shared__ float lds[];
    float a = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    if (R >= 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (R >= 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if (R >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (R >= 160) asm volatile("v_mov_b32 v159, 0" ::: "v159");
    if (R >= 200) asm volatile("v_mov_b32 v199, 0" ::: "v199");
    if (R >= 224) asm volatile("v_mov_b32 v223, 0" ::: "v223");
    if (R >= 232) asm volatile("v_mov_b32 v231, 0" ::: "v231");
    if (R >= 240) asm volatile("v_mov_b32 v239, 0" ::: "v239");
    if (a == 12345.678f) out[0] = a + lds[0];
}

extern "C" int probe_launch(void* stream, int regs, int threads, int blocks, int lds_bytes, int iters, float* out, int prio)
{
    hipStream_t st = (hipStream_t)stream;
#define CASE(R) case R: hipFuncSetAttribute((const void*)&spin_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                        hipLaunchKernelGGL(spin_kernel<R>, dim3(blocks), dim3(threads), lds_bytes, st, out, iters, prio); break;
    switch (regs) {
        CASE(32) CASE(64) CASE(96) CASE(128) CASE(160) CASE(200) CASE(224) CASE(232) CASE(240)
        default: return -1;
    }
    return (int)hipGetLastError();
}
