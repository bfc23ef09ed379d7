// probe_launch.hip — what a kernel launch costs before its first vector load returns (MI355X, gfx950).
// Every workgroup's thread 0 stamps the 100 MHz wall clock at entry and after ONE load (vector, scalar or none); the host
// prints, per launch of a back-to-back series, the kernel's span (first entry -> last stamp) and the latency of that first
// load by XCD (workgroup b runs on XCD b % 8).  Variants: memory kind of the loaded buffer, load kind, grid size.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_launch.hip -o /tmp/probe_launch && /tmp/probe_launch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void k_probe(const int* __restrict__ buf, long long* __restrict__ stamps, int mode, int stride_ints)
{
    const long long t0 = wall_clock64();
    int v = 0;
    const int* p = buf + (size_t)blockIdx.x * stride_ints;
    if (mode == 1) v = p[threadIdx.x & 31];                                        // vector load, one line per workgroup
    else if (mode == 2) { asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory"); }
    if (mode == 1) asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = t1 + (v == 0x7fffffff); }
}

static void run(const char* name, const int* d_buf, int mode, int grid, int block, int stride_ints, int reps)
{
    long long* d_st; hipMalloc(&d_st, (size_t)reps * grid * 16);
    hipStream_t st; hipStreamCreate(&st);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_probe, dim3(grid), dim3(block), 0, st, d_buf, d_st + (size_t)r * grid * 2, mode, stride_ints);
    hipStreamSynchronize(st);
    std::vector<long long> h((size_t)reps * grid * 2);
    hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
    const int r = reps - 1;                                                          // steady state: the last launch of the series
    const long long* s = h.data() + (size_t)r * grid * 2;
    long long first = s[0], last = s[1];
    for (int b = 0; b < grid; ++b) { first = std::min(first, s[b * 2]); last = std::max(last, s[b * 2 + 1]); }
    long long prev_last = 0;
    if (r > 0) { const long long* q = h.data() + (size_t)(r - 1) * grid * 2; for (int b = 0; b < grid; ++b) prev_last = std::max(prev_last, q[b * 2 + 1]); }
    printf("%-44s grid %5d x %4d: span %6.2f us, gap to previous launch %6.2f us, first-load latency by XCD (median us):", name, grid, block,
           (last - first) / 100.0, r > 0 ? (first - prev_last) / 100.0 : 0.0);
    for (int x = 0; x < 8; ++x) {
        std::vector<double> l;
        for (int b = x; b < grid; b += 8) l.push_back((s[b * 2 + 1] - s[b * 2]) / 100.0);
        std::sort(l.begin(), l.end());
        printf(" %5.2f", l.empty() ? 0.0 : l[l.size() / 2]);
    }
    printf("\n");
    hipFree(d_st); hipStreamDestroy(st);
}

int main()
{
    const size_t bytes = 64u << 20;
    int *d_coarse, *d_unc, *h_fine;
    hipMalloc(&d_coarse, bytes); hipMemset(d_coarse, 0, bytes);
    hipExtMallocWithFlags((void**)&d_unc, bytes, hipDeviceMallocUncached); hipMemset(d_unc, 0, bytes);
    hipHostMalloc((void**)&h_fine, bytes, hipHostMallocDefault);
    hipDeviceSynchronize();
    for (int grid : { 64, 512, 2048 }) {
        run("no load", d_coarse, 0, grid, 256, 64, 12);
        run("scalar load, hipMalloc", d_coarse, 2, grid, 256, 64, 12);
        run("vector load, hipMalloc", d_coarse, 1, grid, 256, 64, 12);
        run("vector load, hipMalloc, 1 MiB apart", d_coarse, 1, grid < 64 ? grid : 64, 256, 1 << 18, 12);
        run("vector load, uncached device memory", d_unc, 1, grid, 256, 64, 12);
        run("vector load, pinned host memory", h_fine, 1, grid, 256, 64, 12);
    }
    run("vector load, hipMalloc, 1024-thread blocks", d_coarse, 1, 512, 1024, 64, 12);
    return 0;
}
