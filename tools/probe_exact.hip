// Probe: which device arithmetic is bit-identical to host IEEE arithmetic on gfx950.
// Decides the oracle's summation orders (DESIGN.md "exactness contract").
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_mfma_chain(const float* A, const float* B, float* C, int K) {
    int l = threadIdx.x;
    f32x16 acc = {0};
    for (int kk = 0; kk < K / 2; ++kk) {
        float a = A[(l & 31) * K + 2 * kk + (l >> 5)];
        float b = B[(l & 31) * K + 2 * kk + (l >> 5)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        C[i * 32 + j] = acc[r];
    }
}
__global__ void k_f64(const double* x, const double* y, double* s, double* q, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s[i] = sqrt(x[i]); q[i] = x[i] / y[i]; }
}
__global__ void k_f32(const float* x, const float* y, float* s, float* q, float* f, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s[i] = sqrtf(x[i]); q[i] = x[i] / y[i]; f[i] = fmaf(x[i], y[i], q[i]); }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    std::mt19937_64 rng(1234);
    std::normal_distribution<float> nd(0.f, 0.05f);
    const int K = 64;
    std::vector<float> A(32 * K), B(32 * K), C(32 * 32), Cref(32 * 32);
    for (auto& v : A) v = nd(rng);
    for (auto& v : B) v = nd(rng);
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    k_mfma_chain<<<1, 64>>>(dA, dB, dC, K);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[j * K + k], acc);
        Cref[i * 32 + j] = acc;
        if (memcmp(&acc, &C[i * 32 + j], 4)) ++bad;
    }
    printf("mfma_f32_32x32x2 chain vs host fmaf chain (natural k order): mismatches=%d/1024\n", bad);

    const int N = 1 << 21;
    std::vector<double> x(N), y(N), s(N), q(N);
    std::uniform_real_distribution<double> ud(1e-6, 1e6);
    for (int i = 0; i < N; ++i) { x[i] = ud(rng) * ud(rng); y[i] = ud(rng); }
    double *dx, *dy, *ds, *dq;
    CK(hipMalloc(&dx, N * 8)); CK(hipMalloc(&dy, N * 8)); CK(hipMalloc(&ds, N * 8)); CK(hipMalloc(&dq, N * 8));
    CK(hipMemcpy(dx, x.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), N * 8, hipMemcpyHostToDevice));
    k_f64<<<(N + 255) / 256, 256>>>(dx, dy, ds, dq, N);
    CK(hipMemcpy(s.data(), ds, N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(q.data(), dq, N * 8, hipMemcpyDeviceToHost));
    int bs = 0, bq = 0;
    for (int i = 0; i < N; ++i) {
        volatile double hs = std::sqrt(x[i]); volatile double hq = x[i] / y[i];
        double a = hs, b = hq;
        if (memcmp(&a, &s[i], 8)) ++bs;
        if (memcmp(&b, &q[i], 8)) ++bq;
    }
    printf("f64 sqrt mismatches=%d/%d  f64 div mismatches=%d/%d\n", bs, N, bq, N);

    std::vector<float> xf(N), yf(N), sf(N), qf(N), ff(N);
    std::uniform_real_distribution<float> uf(1e-4f, 1e4f);
    for (int i = 0; i < N; ++i) { xf[i] = uf(rng) * uf(rng); yf[i] = uf(rng); }
    float *fx, *fy, *fs, *fq, *fff;
    CK(hipMalloc(&fx, N * 4)); CK(hipMalloc(&fy, N * 4)); CK(hipMalloc(&fs, N * 4)); CK(hipMalloc(&fq, N * 4)); CK(hipMalloc(&fff, N * 4));
    CK(hipMemcpy(fx, xf.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(fy, yf.data(), N * 4, hipMemcpyHostToDevice));
    k_f32<<<(N + 255) / 256, 256>>>(fx, fy, fs, fq, fff, N);
    CK(hipMemcpy(sf.data(), fs, N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(qf.data(), fq, N * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ff.data(), fff, N * 4, hipMemcpyDeviceToHost));
    int b1 = 0, b2 = 0, b3 = 0;
    for (int i = 0; i < N; ++i) {
        volatile float hs = sqrtf(xf[i]); volatile float hq = xf[i] / yf[i];
        float a = hs, b = hq; float c = fmaf(xf[i], yf[i], b);
        if (memcmp(&a, &sf[i], 4)) ++b1;
        if (memcmp(&b, &qf[i], 4)) ++b2;
        if (memcmp(&c, &ff[i], 4)) ++b3;
    }
    printf("f32 sqrt mismatches=%d  f32 div mismatches=%d  f32 fma mismatches=%d (of %d)\n", b1, b2, b3, N);
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device=%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    return 0;
}
