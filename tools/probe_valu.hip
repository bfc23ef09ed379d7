// probe_valu.hip — issue cost (shader cycles per wave-instruction) of the vector instructions the depthwise 3x3 of the
// OSNet chain kernels is made of, on gfx950: which form of "f16 inputs, f32 accumulate" is cheapest.
// One workgroup of `waves` x 4 waves per CU-filling grid; each wave runs `iters` x 64 copies of one instruction on 16
// independent accumulators (no dependent chain shorter than 16 instructions) and stamps s_memtime around the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_valu.hip -o /tmp/probe_valu && /tmp/probe_valu
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(1024) void k_probe(long long* __restrict__ cyc, float* __restrict__ sink, int iters)
{
    float a[16];
    unsigned h[16];
    float2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; h[i] = 0x3c003c00u + threadIdx.x + i; p[i] = { a[i], a[i] + 1.f }; }
    unsigned w = 0x38003800u + (threadIdx.x & 7);
    float wf = 0.5f;
    float2 wp = { 0.5f, 0.25f };
    asm volatile("" : "+v"(w), "+v"(wf));
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KIND == 0) {                 // v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(wf), "v"(a[(i + 1) & 15]));
                REP16(X)
#undef X
            } else if (KIND == 1) {          // v_fma_mix_f32 (f16 x f16 + f32)
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a[i]) : "v"(h[i]), "v"(w));
                REP16(X)
#undef X
            } else if (KIND == 2) {          // v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(wp), "v"(p[(i + 1) & 15]));
                REP16(X)
#undef X
            } else if (KIND == 3) {          // v_cvt_f32_f16
#define X(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(h[i]));
                REP16(X)
#undef X
            } else if (KIND == 4) {          // v_cvt_f16_f32
#define X(i) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h[i]) : "v"(a[i]));
                REP16(X)
#undef X
            } else if (KIND == 5) {          // v_mov_b32 dpp row_shr:1
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(h[i]) : "v"(h[(i + 1) & 15]));
                REP16(X)
#undef X
            } else if (KIND == 6) {          // v_pk_fma_f16
#define X(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(h[i]) : "v"(w), "v"(h[(i + 1) & 15]));
                REP16(X)
#undef X
            } else if (KIND == 7) {          // v_pk_max_f16
#define X(i) asm volatile("v_pk_max_f16 %0, %1, 0" : "=v"(h[i]) : "v"(h[(i + 1) & 15]));
                REP16(X)
#undef X
            } else if (KIND == 8) {          // v_fma_f32 with a DPP source (the neighbour shift folded into the multiply-add)
#define X(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(wf));
                REP16(X)
#undef X
            } else if (KIND == 9) {          // v_cvt_pkrtz? no: v_pack_b32_f16
#define X(i) asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(h[i]) : "v"(h[(i + 1) & 15]), "v"(h[(i + 2) & 15]));
                REP16(X)
#undef X
            } else if (KIND == 10) {         // v_fma_mix_f32 with an f32 accumulator and hi halves
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(a[i]) : "v"(h[i]), "v"(w));
                REP16(X)
#undef X
            } else if (KIND == 11) {         // v_fma_mixlo_f16 (f32 accumulate, f16 result)
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(h[i]) : "v"(h[(i + 1) & 15]), "v"(w), "v"(a[i]));
                REP16(X)
#undef X
            } else if (KIND == 12) {         // v_dot2_f32_f16 (two taps per instruction, f32 accumulate — different rounding, for reference)
#define X(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(h[i]), "v"(w));
                REP16(X)
#undef X
            }
        }
    }
    typedef _Float16 h4v __attribute__((ext_vector_type(4)));
    typedef _Float16 h8v __attribute__((ext_vector_type(8)));
    typedef float f4v __attribute__((ext_vector_type(4)));
    if (KIND >= 13) {                        // matrix-core forms: 64 MFMAs per iteration on 4 (independent) or 1 (dependent) accumulators
        h4v a4 = { (_Float16)a[0], 0, 0, 0 }, b4 = { (_Float16)a[1], 0, 0, 0 };
        h8v a8 = { (_Float16)a[0], 0, 0, 0, 0, 0, 0, 0 }, b8 = { (_Float16)a[1], 0, 0, 0, 0, 0, 0, 0 };
        f4v c[4] = { { a[2], 0, 0, 0 }, { a[3], 0, 0, 0 }, { a[4], 0, 0, 0 }, { a[5], 0, 0, 0 } };
        const long long u0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = (KIND == 15 || KIND == 16) ? 0 : k;
                    if (KIND == 13 || KIND == 15) c[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[j], 0, 0, 0);
                    else if (KIND == 14 || KIND == 16) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[j], 0, 0, 0);
                    else if (KIND == 18) {   // one MFMA then three independent vector instructions (the chain kernels' mix)
                        c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[k], 0, 0, 0);
                        asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a[4 * k + 0]) : "v"(h[0]), "v"(w));
                        asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a[4 * k + 1]) : "v"(h[1]), "v"(w));
                        asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a[4 * k + 2]) : "v"(h[2]), "v"(w));
                    } else if (KIND == 19) { // role by wave: waves 0-3 of the workgroup MFMA only, waves 4-7 (the same SIMDs) vector only
                        if ((threadIdx.x >> 8) == 0) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[k], 0, 0, 0);
                        else {
                            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a[4 * k + 0]) : "v"(h[0]), "v"(w));
                            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a[4 * k + 1]) : "v"(h[1]), "v"(w));
                            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a[4 * k + 2]) : "v"(h[2]), "v"(w));
                        }
                    }
                    else if (KIND == 17) { if (k & 1) c[k] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[k], 0, 0, 0); else c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[k], 0, 0, 0); }
                }
            }
        }
        const long long u1 = __builtin_readcyclecounter();
        if (c[0][0] + c[1][0] + c[2][0] + c[3][0] + a[0] + a[5] + a[10] + a[14] == 12345.678f) sink[1] = c[0][0];
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = u1 - u0;
        return;
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + (float)h[i] + p[i].x + p[i].y;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int waves_per_simd)
{
    const int block = 256 * waves_per_simd, grid = 256, iters = 2000;
    long long* d; float* sink;
    hipMalloc(&d, (size_t)grid * 16 * 8); hipMalloc(&sink, 4);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k_probe<KIND>, dim3(grid), dim3(block), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    std::vector<long long> h((size_t)grid * (block / 64));
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double n = (double)iters * 64;
    // s_memtime / readcyclecounter ticks are shader cycles (guide); per-SIMD cost = per-wave cycles / waves sharing the SIMD
    printf("%-34s %d wave(s)/SIMD: median %6.2f cycles per instruction per wave  -> %5.2f per SIMD issue slot   (min %6.2f, max %6.2f per wave)\n", name, waves_per_simd,
           h[h.size() / 2] / n, h[h.size() / 2] / n / waves_per_simd, h.front() / n, h.back() / n);
    hipFree(d); hipFree(sink);
}

int main()
{
    for (int w : { 1, 2 }) {
        run<0>("v_fma_f32", w);
        run<1>("v_fma_mix_f32 (lo halves)", w);
        run<10>("v_fma_mix_f32 (hi halves)", w);
        run<2>("v_pk_fma_f32", w);
        run<3>("v_cvt_f32_f16", w);
        run<4>("v_cvt_f16_f32", w);
        run<5>("v_mov_b32_dpp row_shr:1", w);
        run<8>("v_fmac_f32_dpp row_shr:1", w);
        run<6>("v_pk_fma_f16", w);
        run<7>("v_pk_max_f16", w);
        run<9>("v_pack_b32_f16", w);
        run<11>("v_fma_mixlo_f16", w);
        run<12>("v_dot2_f32_f16", w);
        run<13>("mfma_f32_16x16x16_f16, 4 accumulators", w);
        run<14>("mfma_f32_16x16x32_f16, 4 accumulators", w);
        run<15>("mfma_f32_16x16x16_f16, dependent", w);
        run<16>("mfma_f32_16x16x32_f16, dependent", w);
        run<17>("mfma 16x16x32 / 16x16x16 alternating", w);
        run<18>("1 mfma + 3 v_fma_mix per slot (per 4 instr)", w);
        if (w == 2) run<19>("mfma wave + v_fma_mix wave on one SIMD", w);
    }
    return 0;
}
