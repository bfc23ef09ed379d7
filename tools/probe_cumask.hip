// probe_cumask.hip — which compute units a CU-masked HIP stream (hipExtStreamCreateWithCUMask) really gets on MI355X: every
// workgroup of a wide launch reports its XCC id and its (shader engine, shader array, compute unit) id; the host prints, per mask,
// the number of distinct compute units per XCC.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_cumask.hip -o /tmp/probe_cumask && /tmp/probe_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void k_where(unsigned* out)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // spin a little so that the launch spreads over every unit the queue may use
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hw; }
}

static void run(const char* name, int first, int count, int total)
{
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    for (int i = first; i < first + count; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
    const int grid = 4096;
    unsigned* d; hipMalloc(&d, grid * 8);
    hipLaunchKernelGGL(k_where, dim3(grid), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(grid * 2);
    hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per;
    std::map<unsigned, int> wg_per;
    for (int b = 0; b < grid; ++b) {
        const unsigned xcc = h[b * 2] & 0xf, hw = h[b * 2 + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;     // gfx9 HW_ID: CU_ID 11:8, SH_ID 12, SE_ID 15:13
        per[xcc].insert((se << 8) | (sh << 4) | cu);
        wg_per[xcc]++;
    }
    printf("%-28s bits [%3d, %3d):", name, first, first + count);
    for (auto& kv : per) printf("  xcc%u: %zu CUs / %d wgs", kv.first, kv.second.size(), wg_per[kv.first]);
    printf("\n");
    if (count <= 16) {
        printf("    units:");
        for (auto& kv : per) for (unsigned u : kv.second) printf(" x%u.se%u.sh%u.cu%u", kv.first, u >> 8, (u >> 4) & 1, u & 15);
        printf("\n");
    }
    hipFree(d); hipStreamDestroy(st);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int total = p.multiProcessorCount;
    printf("compute units: %d\n", total);
    run("all", 0, total, total);
    run("first 8", 0, 8, total);
    run("second 8", 8, 8, total);
    run("first 16", 0, 16, total);
    run("first 32", 0, 32, total);
    run("bits 32..63", 32, 32, total);
    run("all but the first 16", 16, total - 16, total);
    run("all but the first 32", 32, total - 32, total);
    return 0;
}
