// Which CUs does a CU-masked stream reach?  (round 6, VERDICT r5 item 1d: a spatial-partition experiment needs to know how the
// bits of hipExtStreamCreateWithCUMask map to XCDs.)  A kernel of many one-wave workgroups records (XCC_ID, SE_ID, CU_ID) of the
// CU it ran on; the host prints, per mask pattern, how many distinct CUs per XCD were reached.
// build: hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <set>
#include <map>
#include <vector>

__global__ void where_kernel(unsigned* out)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the CU busy for a moment so that the grid spreads over every CU the stream may use
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 20000) { }
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xF) << 16) | (hw & 0xFFFF);
}

static void run(const char* name, const std::vector<uint32_t>& mask)
{
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    const int n = 8192;
    unsigned* d; hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    std::map<int, std::set<unsigned>> per;
    for (unsigned v : h) per[(int)(v >> 16)].insert((v >> 8) & 0xFF);     // CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    int total = 0;
    printf("%-28s", name);
    for (auto& kv : per) { printf(" xcc%d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  = %d CUs\n", total);
    hipFree(d); hipStreamDestroy(st);
}

int main()
{
    auto bits = [](auto pred) { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
    run("all 256", bits([](int) { return true; }));
    run("bits 0..63", bits([](int i) { return i < 64; }));
    run("bits 64..127", bits([](int i) { return i >= 64 && i < 128; }));
    run("bits 0..127", bits([](int i) { return i < 128; }));
    run("i % 8 == 0", bits([](int i) { return i % 8 == 0; }));
    run("i % 8 < 2", bits([](int i) { return i % 8 < 2; }));
    run("i % 4 == 0", bits([](int i) { return i % 4 == 0; }));
    run("i % 2 == 0", bits([](int i) { return i % 2 == 0; }));
    run("(i / 8) % 4 == 0", bits([](int i) { return (i / 8) % 4 == 0; }));
    run("(i / 32) % 2 == 0", bits([](int i) { return (i / 32) % 2 == 0; }));
    return 0;
}
