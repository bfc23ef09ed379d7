// ss_ops32.hip — the ReID network (row a5, OSNet-x0.25) with fp32 activations and fp32 weights on the f32-input matrix
// instruction v_mfma_f32_16x16x4_f32: the ACCURACY MODE of the path.
//
// Why it exists: BASELINE north_star bounds the float distances at 1e-4 against the CPU reference, and the reference passes no
// half= (/root/reference/yolo_multi_model.py:41): its arithmetic is fp32.  With f16 activations (ss_ops.hip) this network's
// appearance distances are off by 3e-2 (bench.py reid_f16_vs_f32); fp32 activations give 1e-5.  Until round 5 that mode ran on
// the library convolutions.  These five kernels replace them:
//
//   k32_stem    conv 7x7/2 (3 -> 16) + bias + ReLU + max pool 3x3/2            implicit GEMM, K = 147, input band in LDS
//   k32_pw      1x1 convolution + bias (+ shortcut) (+ ReLU)                     weights in LDS, pixels straight from HBM
//   k32_chains  the ten LightConv layers (1x1 linear -> depthwise 3x3 + bias + ReLU) of an OSBlock's four chains for a band of
//               one image, every intermediate in LDS; writes the four chain outputs + their per-band channel sums
//   k32_tail    channel gates (from the sums) -> gate-weighted sum of the four chains -> conv3 + bias + shortcut (identity or
//               1x1 `down`) + ReLU -> [block output] -> the 1x1 ConvBR that follows (next block's conv1 / the stage's ConvBR
//               [+ 2x2 average] / conv5) -- one launch, the block output never leaves the registers between the two products
//   k32_head    global average pool + fully connected layer + bias + ReLU
//
// One operand convention everywhere.  A tensor is NHWC fp32; a pixel's C channels are C/4 CHUNKS of 4 floats (one 16-byte
// vector).  For a 16-pixel tile, lane (kq = lane >> 4, n = lane & 15) of a wave holds chunks kq, kq + 4, kq + 8, .. of pixel n.
// v_mfma_f32_16x16x4_f32 takes A[m][k] from lane (k, m) and B[k][n] from lane (k, n), one float each, so component s of chunk
// kq + 4 jj is B's k-slot kq of the MFMA (jj, s), whose A operand is W[m][4 (kq + 4 jj) + s]: every operand is a 16-byte load and
// the K axis is walked in the order (jj, s, kq).  The result D puts output channels 16 mt + 4 q .. + 3 of pixel n into lane
// (q, n) -- chunk 4 mt + q of the OUTPUT pixel, i.e. the accumulators of M-tile mt ARE the B chunks jj = mt of the next 1x1
// product (k32_tail feeds conv3's accumulators straight into the following convolution).
//
// Sums are fp32 in a fixed order (no atomics), so a launch is reproducible; the order is not PyTorch's, so the results agree
// with the fp32 CPU network to rounding (1e-6 relative per layer), not bit for bit: tests/test_gpu_nets32.py states the bounds.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/strongsort_hip.h"

typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// workgroup barrier that orders LDS traffic only: global loads requested for a later phase stay in flight across it
#define LDS_BARRIER32() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)

static __device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
static __device__ __forceinline__ void st4(float* p, const f4 v) { *reinterpret_cast<f4*>(p) = v; }
static __device__ __forceinline__ f4 zero4() { return f4{ 0.f, 0.f, 0.f, 0.f }; }
static __device__ __forceinline__ f4 fma4(const f4 a, const f4 b, const f4 c)
{
    return f4{ __builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1]), __builtin_fmaf(a[2], b[2], c[2]), __builtin_fmaf(a[3], b[3], c[3]) };
}
static __device__ __forceinline__ f4 relu4(const f4 v)
{
    return f4{ v[0] > 0.f ? v[0] : 0.f, v[1] > 0.f ? v[1] : 0.f, v[2] > 0.f ? v[2] : 0.f, v[3] > 0.f ? v[3] : 0.f };
}
static __device__ __forceinline__ f4 max4(const f4 a, const f4 b)
{
    return f4{ a[0] > b[0] ? a[0] : b[0], a[1] > b[1] ? a[1] : b[1], a[2] > b[2] ? a[2] : b[2], a[3] > b[3] ? a[3] : b[3] };
}

// Weight matrix [N][K] -> LDS rows of KP = 16 JJ + 4 floats, zero in the padding columns and in the rows N .. NP - 1 (24 output channels =
// two M-tiles, the second half empty).  (The + 4 pitch is a 2-way conflict on the A-operand read — lane (kq, n): 16 bytes at row n, chunk kq,
// ds_read_b128's lane groups {0-3, 12-15, 20-27}, ..; + 8 is conflict-free and was measured in round 6: no faster anywhere, and the larger
// tables cost the 24-channel tail its third workgroup per CU, 155 -> 197 us.  k32_conv, whose LDS is small, uses the conflict-free 72.)
template <int K, int N, int NTHR>
static __device__ __forceinline__ void stage_w(float* __restrict__ Ws, const float* __restrict__ w, int tid)
{
    constexpr int CH = K / 4, JJ = (CH + 3) / 4, KP = 16 * JJ + 4, NP = (N + 15) / 16 * 16;
    // up to eight vectors of a thread are requested (from clamped, always valid addresses) before the first LDS store: the rolled
    // `predicated load -> store` loop paid one L2 round trip per NTHR vectors - 3 to 17 of them back to back at the start of a workgroup
    constexpr int TOT = NP * (KP / 4), NIT = (TOT + NTHR - 1) / NTHR, UB = NIT < 8 ? NIT : 8;
    for (int i0 = tid; i0 < TOT; i0 += UB * NTHR) {
        f4 v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = min(i0 + u * NTHR, TOT - 1), r = i / (KP / 4), c = i - r * (KP / 4);
            v[u] = ld4(w + (size_t)min(r, N - 1) * K + 4 * min(c, CH - 1));
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = i0 + u * NTHR, r = i / (KP / 4), c = i - r * (KP / 4);
            if (i < TOT) st4(Ws + r * KP + 4 * c, (r < N && c < CH) ? v[u] : zero4());
        }
    }
}
template <int K, int N>
constexpr int w_lds_floats() { return ((N + 15) / 16 * 16) * (16 * ((K / 4 + 3) / 4) + 4); }

// acc[mt] += W[16 mt .. + 15][:] x B for the 16 pixels of the tile; b[jj] = this lane's chunks kq + 4 jj
template <int K, int N>
static __device__ __forceinline__ void mm_tile(const float* __restrict__ Ws, const f4 (&b)[(K / 4 + 3) / 4], f4 (&acc)[(N + 15) / 16], int kq, int n)
{
    constexpr int JJ = (K / 4 + 3) / 4, KP = 16 * JJ + 4, MT = (N + 15) / 16;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            const f4 a = ld4(Ws + (mt * 16 + n) * KP + 4 * (kq + 4 * jj));
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[mt] = MFMA4(a[s], b[jj][s], acc[mt]);
        }
}

// ---- k32_pw ------------------------------------------------------------------------------------------------------------
// out[p][0..N) = [relu](W x[p] + bias [+ res[p]]), p < M.  A wave owns 16-pixel tiles; workgroup = 4 waves x `tpw` tiles.
template <int K, int N>
__global__ __launch_bounds__(256) void k32_pw(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                              const float* __restrict__ res, float* __restrict__ out, long long M, int relu, int tpw,
                                              const int* __restrict__ n_img, int img_px)
{
    constexpr int CH = K / 4, JJ = (CH + 3) / 4, MT = (N + 15) / 16;
    __shared__ __attribute__((aligned(16))) float Ws[w_lds_floats<K, N>()];
    if (n_img) { const long long mv = (long long)(*n_img) * img_px; if (mv < M) M = mv; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n = lane & 15;
    if ((long long)blockIdx.x * 64 * tpw >= M) return;
    stage_w<K, N, 256>(Ws, w, tid);
    __syncthreads();
    const long long tile0 = ((long long)blockIdx.x * 4 + wave) * tpw;
    for (int it = 0; it < tpw; ++it) {
        const long long p0 = (tile0 + it) * 16;
        if (p0 >= M) break;
        const long long px = p0 + n;
        const bool valid = px < M;
        f4 b[JJ];
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            const int c = kq + 4 * jj;
            b[jj] = (valid && c < CH) ? ld4(x + px * K + 4 * c) : zero4();
        }
        f4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = zero4();
        mm_tile<K, N>(Ws, b, acc, kq, n);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int oc = 16 * mt + 4 * kq;
            if (oc >= N || !valid) continue;
            f4 v = acc[mt] + ld4(bias + oc);
            if (res) v = v + ld4(res + px * N + oc);
            if (relu) v = relu4(v);
            st4(out + px * N + oc, v);
        }
    }
}

// ---- k32_chains ----------------------------------------------------------------------------------------------------------
// The four LightConv chains (1, 2, 3 and 4 layers deep; layer = 1x1 linear -> depthwise 3x3 + bias + ReLU) of an OSBlock on a band
// of R rows of one image (+ HALO rows above and below when the image has several bands: four stacked 3x3 layers need 4).
// LDS: P (the 1x1 output of the current layer) and Q (the layer's output = next layer's input), [RB rows][W][C + 4] floats each;
// the pixel pitch C + 4 (20 / 28 / 36 dwords) spreads the 16-byte accesses of any 16 distinct columns over all 64 banks.
// Rows outside the image hold ZERO in P (they are the depthwise layer's padding); rows outside the LDS window count as zero too,
// which only reaches halo rows whose values are never stored.  Outputs: ys[t] band rows, psum[t][img][band][c] = their channel sums.
#define C32_THREADS 768
template <int C, int W>
__global__ __launch_bounds__(C32_THREADS) void k32_chains(const float* __restrict__ x1, const float* __restrict__ w1 /*[10][C][C]*/,
                                                        const float* __restrict__ w9 /*[10][9][C]*/, const float* __restrict__ bs /*[10][C]*/,
                                                        float* __restrict__ y0, float* __restrict__ y1, float* __restrict__ y2,
                                                        float* __restrict__ y3, float* __restrict__ psum, int Nimg, int H, int R, int HALO,
                                                        const int* __restrict__ n_img)
{
    constexpr int CH = C / 4, JJ = (CH + 3) / 4, MT = (C + 15) / 16, PITCH = C + 4, SLOTS = C32_THREADS / CH, RGS = SLOTS / W;
    static_assert(C32_THREADS % CH == 0 && SLOTS % W == 0, "thread map");
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    const int img = blockIdx.y, band = blockIdx.x, bands = gridDim.x;
    if (n_img && img >= *n_img) return;
    const int RB = R + 2 * HALO, r0 = band * R - HALO;          // image row of LDS row 0
    float* __restrict__ P = smem32;
    float* __restrict__ Q = P + RB * W * PITCH;
    float* __restrict__ S = Q + RB * W * PITCH;                 // [SLOTS][C] partial channel sums of a chain's last layer
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n = lane & 15;
    const int chunk = tid % CH, slot = tid / CH, xx = slot % W, rg = slot / W;
    const int RPG = (RB + RGS - 1) / RGS, ra = rg * RPG, rb = (ra + RPG < RB) ? ra + RPG : RB;
    const int ntiles = RB * W / 16;
    const size_t ibase = (size_t)img * H * W * C;
    int layer = 0;
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
        float* __restrict__ yt = t == 0 ? y0 : t == 1 ? y1 : t == 2 ? y2 : y3;
#pragma unroll 1
        for (int d = 0; d <= t; ++d, ++layer) {
            // ---- 1x1 (no bias, no activation): source x1 (global) for a chain's first layer, Q afterwards
            {
                f4 a[MT][JJ];
                const float* wl = w1 + (size_t)layer * C * C;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int jj = 0; jj < JJ; ++jj) {
                        const int row = 16 * mt + n, c = kq + 4 * jj;
                        a[mt][jj] = (row < C && c < CH) ? ld4(wl + row * C + 4 * c) : zero4();
                    }
                for (int tile = wave; tile < ntiles; tile += C32_THREADS / 64) {
                    const int p = tile * 16 + n, r = p / W, y = r0 + r;
                    const bool inimg = y >= 0 && y < H;
                    f4 b[JJ];
#pragma unroll
                    for (int jj = 0; jj < JJ; ++jj) {
                        const int c = kq + 4 * jj;
                        const int cc = c < CH ? c : 0, yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
                        const f4 v = d == 0 ? ld4(x1 + ibase + ((size_t)yc * W + (p - r * W)) * C + 4 * cc) : ld4(Q + p * PITCH + 4 * cc);
                        b[jj] = (c < CH && (d != 0 || inimg)) ? v : zero4();
                    }
                    f4 acc[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = zero4();
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int jj = 0; jj < JJ; ++jj)
#pragma unroll
                            for (int s = 0; s < 4; ++s) acc[mt] = MFMA4(a[mt][jj][s], b[jj][s], acc[mt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int co = 4 * mt + kq;                 // output chunk
                        if (co < CH) st4(P + p * PITCH + 4 * co, inimg ? acc[mt] : zero4());
                    }
                }
            }
            __syncthreads();
            // ---- depthwise 3x3 + bias + ReLU: thread = (chunk, column, run of RPG rows), a rolling 3x3 window of 16-byte vectors
            {
                const bool last = d == t;
                const float* wt = w9 + (size_t)layer * 9 * C + 4 * chunk;
                f4 k9[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) k9[i] = ld4(wt + i * C);
                const f4 bb = ld4(bs + layer * C + 4 * chunk);
                f4 psa = zero4();
                auto ld = [&](int r, int x) -> f4 {                    // unconditional load from a clamped address + select (no branch per tap)
                    const bool ok = r >= 0 && r < RB && x >= 0 && x < W;
                    const int rc = r < 0 ? 0 : (r >= RB ? RB - 1 : r), xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
                    const f4 v = ld4(P + (rc * W + xc) * PITCH + 4 * chunk);
                    return ok ? v : zero4();
                };
                if (ra < rb) {
                    f4 t0 = ld(ra - 1, xx - 1), t1 = ld(ra - 1, xx), t2 = ld(ra - 1, xx + 1);
                    f4 m0 = ld(ra, xx - 1), m1 = ld(ra, xx), m2 = ld(ra, xx + 1);
                    for (int r = ra; r < rb; ++r) {
                        const f4 b0 = ld(r + 1, xx - 1), b1 = ld(r + 1, xx), b2 = ld(r + 1, xx + 1);
                        f4 o = bb;
                        o = fma4(k9[0], t0, o); o = fma4(k9[1], t1, o); o = fma4(k9[2], t2, o);
                        o = fma4(k9[3], m0, o); o = fma4(k9[4], m1, o); o = fma4(k9[5], m2, o);
                        o = fma4(k9[6], b0, o); o = fma4(k9[7], b1, o); o = fma4(k9[8], b2, o);
                        o = relu4(o);
                        st4(Q + (r * W + xx) * PITCH + 4 * chunk, o);
                        const int y = r0 + r;
                        if (last && r >= HALO && r < HALO + R && y >= 0 && y < H) {
                            st4(yt + ibase + ((size_t)y * W + xx) * C + 4 * chunk, o);
                            psa = psa + o;
                        }
                        t0 = m0; t1 = m1; t2 = m2; m0 = b0; m1 = b1; m2 = b2;
                    }
                }
                if (last) st4(S + slot * C + 4 * chunk, psa);
            }
            __syncthreads();
        }
        if (tid < C) {                                               // channel sums of chain t over this band, slots in order
            float s = 0.f;
            for (int i = 0; i < SLOTS; ++i) s += S[i * C + tid];
            psum[(((size_t)t * Nimg + img) * bands + band) * C + tid] = s;
        }
    }
}

// ---- k32_chains3 ---------------------------------------------------------------------------------------------------------
// k32_chains' two-phase form (1x1 of every tile -> barrier -> depthwise of every pixel -> barrier) with what the first trace asked
// for (745 us per launch at 1024 crops, 46 us per workgroup for work that issues in 17):
//  * one lane map for BOTH phases — lane (kq, n) = chunk kq (+ 4 jj) of column n of a 16-pixel row segment — and a pixel pitch
//    that is 8 mod 16 dwords (24 / 24 / 40 for 16 / 24 / 32 channels): every 16-byte LDS access of the kernel, the depthwise
//    taps at x - 1, x, x + 1 included, is conflict-free (the chunk-fastest thread order of k32_chains on a C + 4 pitch paid 3x on
//    its nine taps);
//  * the weights of a phase are requested one phase ahead (1x1 matrix at the start of the depthwise phase before it, the nine
//    taps + bias at the start of the 1x1 phase before them), and a chain's first 1x1 reads x1 from registers filled during the
//    previous chain's last depthwise phase: no phase starts with a dependent global load;
//  * channel sums by wave shuffles, one LDS vector per (wave, lane row).
// 24 channels = 6 chunks: waves 0..7 take chunks 0..3 of four rows each, waves 8..11 chunks 4, 5 of TWO row runs at once (lane
// rows kq < 2 / kq >= 2), so all twelve waves walk four rows.
template <int C> struct Ch3 {
    static constexpr int PITCH = (C == 24) ? 24 : C + 8;
};
template <int C, int W, bool PRE>
__global__ __launch_bounds__(C32_THREADS) void k32_chains3(const float* __restrict__ x1, const float* __restrict__ w1 /*[10][C][C]*/,
                                                         const float* __restrict__ w9 /*[10][9][C]*/, const float* __restrict__ bs /*[10][C]*/,
                                                         float* __restrict__ y0, float* __restrict__ y1, float* __restrict__ y2,
                                                         float* __restrict__ y3, float* __restrict__ psum, int Nimg, int H, int R, int HALO,
                                                         const int* __restrict__ n_img)
{
    constexpr int CH = C / 4, JJ = (CH + 3) / 4, MT = (C + 15) / 16, PITCH = Ch3<C>::PITCH, NWV = C32_THREADS / 64, RPG = 4;
    constexpr int TPW = 4, PWV = (C == 16) ? 12 : 8;             // 1x1 phase: 4 tiles of 16 pixels per wave on 12 (48 tiles) / 8 (32 tiles) waves
    static_assert((C == 16 && W == 32) || (C == 24 && W == 16), "instantiated for the 64 x 32 x 16 and 32 x 16 x 24 maps");
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    const int img = blockIdx.y, band = blockIdx.x, bands = gridDim.x;
    if (n_img && img >= *n_img) return;
    const int RB = R + 2 * HALO, r0 = band * R - HALO;          // image row of LDS row 0 (host: RB == 24 for C 16, 32 for C 24)
    // P carries a one-pixel ZERO border (rows -1 and RB, columns -1 and W): the nine taps of a pixel are nine plain loads at
    // immediate offsets from one lane base — no bounds test, no clamp, no select (they were 60 % of the kernel's vector instructions)
    constexpr int PW = W + 2;
    float* __restrict__ P = smem32;                             // [(RB + 2)][W + 2][PITCH]
    float* __restrict__ Q = P + (RB + 2) * PW * PITCH;          // [RB][W][PITCH]
    float* __restrict__ S = Q + RB * W * PITCH;                 // [NWV][4 lane rows][4] partial channel sums
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kq = lane >> 4, n = lane & 15;
    // depthwise unit of this lane: chunk dc of column dx, rows dra .. dra + 3
    int dc, dx, dra;
    if (C == 16) { const int xt = wave / 6; dc = kq; dx = 16 * xt + n; dra = (wave - 6 * xt) * RPG; }
    else if (wave < 8) { dc = kq; dx = n; dra = wave * RPG; }
    else { dc = 4 + (kq & 1); dx = n; dra = (2 * (wave - 8) + (kq >> 1)) * RPG; }
    const float* __restrict__ xi = x1 + (size_t)img * H * W * C;

    auto load_a = [&](int layer, f4 (&a)[MT][JJ]) {
        const float* wl = w1 + (size_t)layer * C * C;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) {
                const int row = 16 * mt + n, c = kq + 4 * jj;
                const f4 v = ld4(wl + (unsigned)((row < C ? row : 0) * C + 4 * (c < CH ? c : 0)));
                a[mt][jj] = (row < C && c < CH) ? v : zero4();
            }
    };
    auto load_x1 = [&](f4 (&b)[PRE ? TPW : 1][JJ]) {                        // this wave's tiles of the band, straight from the block's conv1 output
#pragma unroll
        for (int i = 0; i < (PRE ? TPW : 1); ++i) {
            const int tile = (wave < PWV ? wave : 0) + i * PWV, p = tile * 16 + n, r = p / W, y = r0 + r;
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) {
                const int c = kq + 4 * jj;
                const bool ok = c < CH && y >= 0 && y < H;
                const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y), cc = c < CH ? c : 0;
                const f4 v = ld4(xi + (unsigned)((yc * W + (p - r * W)) * C + 4 * cc));      // unconditional load + select (see the depthwise taps)
                b[i][jj] = ok ? v : zero4();
            }
        }
    };
    // PRE: a chain's first 1x1 takes x1 from registers filled one phase earlier (16 / 32 VGPRs); !PRE: it loads x1 in place
    f4 a[MT][JJ], bpre[PRE ? TPW : 1][JJ];
    load_a(0, a);
    if (PRE) load_x1(bpre);
    for (int i = threadIdx.x; i < (RB + 2) * PW * PITCH / 4; i += C32_THREADS) st4(P + 4 * i, zero4());     // the border (the interior is rewritten)
    LDS_BARRIER32();
    int layer = 0;
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
        float* __restrict__ yt = (t == 0 ? y0 : t == 1 ? y1 : t == 2 ? y2 : y3) + (size_t)img * H * W * C;
#pragma unroll 1
        for (int d = 0; d <= t; ++d, ++layer) {
            // ---- 1x1 phase (taps + bias of this layer's depthwise requested first)
            f4 k9[9];
            const float* __restrict__ w9l = w9 + (size_t)layer * 9 * C;
#pragma unroll
            for (int i = 0; i < 9; ++i) k9[i] = ld4(w9l + (unsigned)(i * C + 4 * dc));
            const f4 bb = ld4(bs + layer * C + 4 * dc);
            if (wave < PWV) {                                    // (wave-uniform: the tiles below are straight-line code, their LDS reads and MFMA chains interleave)
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int p = (wave + i * PWV) * 16 + n, r = p / W, y = r0 + r;
                    const bool inimg = y >= 0 && y < H;
                    f4 b[JJ];
#pragma unroll
                    for (int jj = 0; jj < JJ; ++jj) {
                        const int c = kq + 4 * jj;
                        const int cc = c < CH ? c : 0;
                        if (PRE) {
                            const f4 qv = ld4(Q + p * PITCH + 4 * cc);
                            b[jj] = d == 0 ? bpre[i][jj] : (c < CH ? qv : zero4());
                        } else {
                            const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
                            const f4 v = d == 0 ? ld4(xi + (unsigned)((yc * W + (p - r * W)) * C + 4 * cc)) : ld4(Q + p * PITCH + 4 * cc);
                            b[jj] = (c < CH && (d != 0 || inimg)) ? v : zero4();
                        }
                    }
                    f4 acc[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = zero4();
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int jj = 0; jj < JJ; ++jj)
#pragma unroll
                            for (int s = 0; s < 4; ++s) acc[mt] = MFMA4(a[mt][jj][s], b[jj][s], acc[mt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int co = 4 * mt + kq;
                        if (co < CH) st4(P + ((r + 1) * PW + (p - r * W) + 1) * PITCH + 4 * co, inimg ? acc[mt] : zero4());
                    }
                }
            }
            LDS_BARRIER32();
            // ---- depthwise phase (next 1x1 matrix / next chain's x1 tiles requested first)
            const bool last = d == t;
            if (layer < 9) load_a(layer + 1, a);
            if (PRE && last && t < 3) load_x1(bpre);
            f4 psa = zero4();
            {
                const float* __restrict__ pb = P + (dra * PW + dx) * PITCH + 4 * dc;        // tap (row dra - 1, column dx - 1)
                auto ld = [&](int i, int j) -> f4 { return ld4(pb + (i * PW + j) * PITCH); };
                f4 t0 = ld(0, 0), t1 = ld(0, 1), t2 = ld(0, 2);
                f4 m0 = ld(1, 0), m1 = ld(1, 1), m2 = ld(1, 2);
#pragma unroll
                for (int i = 0; i < RPG; ++i) {
                    const int r = dra + i;
                    const f4 b0 = ld(i + 2, 0), b1 = ld(i + 2, 1), b2 = ld(i + 2, 2);
                    f4 o = bb;
                    o = fma4(k9[0], t0, o); o = fma4(k9[1], t1, o); o = fma4(k9[2], t2, o);
                    o = fma4(k9[3], m0, o); o = fma4(k9[4], m1, o); o = fma4(k9[5], m2, o);
                    o = fma4(k9[6], b0, o); o = fma4(k9[7], b1, o); o = fma4(k9[8], b2, o);
                    o = relu4(o);
                    st4(Q + (r * W + dx) * PITCH + 4 * dc, o);
                    const int y = r0 + r;
                    if (last && r >= HALO && r < HALO + R && y >= 0 && y < H) {
                        st4(yt + (unsigned)((y * W + dx) * C + 4 * dc), o);
                        psa = psa + o;
                    }
                    t0 = m0; t1 = m1; t2 = m2; m0 = b0; m1 = b1; m2 = b2;
                }
            }
            if (last) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = psa[j];
                    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                    psa[j] = v;
                }
                if (n == 0) st4(S + (wave * 4 + kq) * 4, psa);
            }
            LDS_BARRIER32();
            if (last && tid < C) {                                   // channel tid of chain t: the (wave, lane row) partials of its chunk, in order
                const int cq = tid >> 2, j = tid & 3;
                float sum = 0.f;
                if (C == 16 || cq < 4) {
                    for (int wv = 0; wv < (C == 16 ? NWV : 8); ++wv) sum += S[(wv * 4 + cq) * 4 + j];
                } else {
                    for (int wv = 8; wv < NWV; ++wv) { sum += S[(wv * 4 + (cq - 4)) * 4 + j]; sum += S[(wv * 4 + (cq - 2)) * 4 + j]; }
                }
                psum[(((size_t)t * Nimg + img) * bands + band) * C + tid] = sum;
            }
        }
    }
}

// ---- k32_chainsR ---------------------------------------------------------------------------------------------------------
// The chains as a REGISTER-resident row stream (round 6): no activation ever touches LDS.  k32_chains3 spends its time moving every
// layer's 1x1 output through LDS (write 16 B per (pixel, chunk), read it back 4.5 times) with the LDS, vector and matrix phases of
// a layer one after the other behind workgroup barriers and one workgroup per CU (profiles/r05_pmc_osnet32.json: MFMA-busy 0.19,
// 1.2 TB/s).  The operand convention makes all of that unnecessary:
//   * after the 1x1 of a 16-pixel row segment lane (kq, n) holds output chunk kq (+ 4 mt) of column n — the depthwise 3x3 is per
//     channel, so its column neighbours are the SAME register of lanes n - 1 / n + 1 of the same 16-lane DPP row (row_shr:1 /
//     row_shl:1; the image's left / right zero padding is DPP's bound_ctrl zero), its row neighbours are the same lane one and
//     two steps ago,
//   * and the depthwise output of lane (kq, n) IS the B operand of the next layer's MFMAs.
// A WAVE walks down the rows of ONE image for ONE chain: when the 1x1 row p of a layer arrives it finishes output row p - 1
// (+ ky = 2 taps), continues row p (+ ky = 1) and starts row p + 1 (bias + ky = 0) — two partial accumulators per layer, summed in
// the order bias, ky 0, 1, 2 (kx 0, 1, 2 inside) — and hands the finished row to the next layer in the same step.  State per layer:
// 2 x NT x MT f4 accumulators (+ the 1x1 matrix when it lives in registers); the nine taps + bias of a layer are read from the
// workgroup's LDS table in every step (16-lane broadcast reads; kept out of registers on purpose: the first register-stream attempt
// of round 5 died of spills because every chain's state and taps lived in one wave).
// Work split: a wave runs chain 3 (4 layers) then chain 0 (1 layer), its partner chain 2 (3) then chain 1 (2): five layer passes
// each, so the two waves of an image and the two images of a workgroup finish together; no barrier after the table is staged, no
// halo (a wave owns the whole image height), channel sums in registers (psum has ONE band).
#define R32_THREADS 256
static __device__ __forceinline__ float dppf_shr1_z(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true)); }
static __device__ __forceinline__ float dppf_shl1_z(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true)); }

template <int C> struct ChR {
    static constexpr int CH = C / 4, JJ = (CH + 3) / 4, MT = JJ, CP = 16 * JJ, KP = CP + 4, NPR = (C % 16) ? C + 1 : C;   // NPR: rows of a staged 1x1 matrix (+ one zero row)
    static constexpr int TAB = 10 * 10 * CP;                       // floats: [layer][9 taps + bias][CP]
};

// One chain of L layers (global layer indices l0 .. l0 + L - 1) over the H rows of one image, by one wave.
// Columns: with two segments per row (W = 32) segment t holds the columns 2 n + t (even / odd), so the left neighbour of an even
// column is the odd segment shifted by one lane and its right neighbour the odd segment itself (and mirrored for the odd columns):
// two DPP moves per value instead of six, no seam between the segments.
// ONE loop, ONE body: a layer runs from the step its first row arrives (a scalar compare per layer); after its last row it runs on
// ZERO rows, which is what the zero padding below the image asks for; "is this row inside the image" is the third operand of the
// ReLU, v_med3_f32(x, 0, lim) with the wave-uniform lim = +inf / 0, so the masking costs no instruction.  (Compile experiments of
// round 6: separate prologue / steady / epilogue copies of the body made the register allocator spill 400-800 values, a body with no
// branch at all 40-560; the per-layer guards keep the layers separate basic blocks and the allocation spill-free.)
// Also measured and not kept: a layer consuming the row the one before finished in the PREVIOUS step (so that one wave's 1x1 of layer
// l + 1 runs beside its depthwise of layer l): 32 more live registers, 36-300 spilled values, 340 instead of 284 us; three waves per
// SIMD at 24 channels (168 registers): 236 spilled values, 538 instead of 198 us.
template <int C, int W, int L, bool ALDS>
struct ChainState {
    static constexpr int JJ = ChR<C>::JJ, MT = ChR<C>::MT, NT = (W >= 16 ? W / 16 : 1);
    f4 a[ALDS ? 1 : L][MT][JJ];
    f4 accA[L][NT][MT], accB[L][NT][MT], sum[MT];
};

// P[.][mt] = rows 16 mt .. of W_l x (this lane's part of the layer's input row)
template <int C, int W, int L, bool ALDS>
static __device__ __forceinline__ void chain_mm(const ChainState<C, W, L, ALDS>& S, int l, int mt, int wl0, int n,
                                                const f4 (&in)[(W >= 16 ? W / 16 : 1)][ChR<C>::JJ], f4 (&P)[(W >= 16 ? W / 16 : 1)][ChR<C>::MT])
{
    constexpr int JJ = ChR<C>::JJ, NT = (W >= 16 ? W / 16 : 1), KP = ChR<C>::KP, NPR = ChR<C>::NPR;
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    int wo = wl0 + l * NPR * KP;
    if (ALDS) asm volatile("" : "+v"(wo));                                      // the matrix reads stay inside the step (not hoisted into registers)
    const float* wp = smem32 + wo;
    f4 am[JJ];
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) {
        const int row = 16 * mt + n;
        am[jj] = ALDS ? ld4(wp + (row < C ? row : C) * KP + 16 * jj) : S.a[ALDS ? 0 : l][mt][jj];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) P[t][mt] = zero4();
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj)                                              // the segments' chains side by side: a dependent MFMA is 40 cycles behind its
#pragma unroll
        for (int s = 0; s < 4; ++s)                                              // predecessor, an independent one 32
#pragma unroll
            for (int t = 0; t < NT; ++t) P[t][mt] = MFMA4(am[jj][s], in[t][jj][s], P[t][mt]);
}

// the 1x1 row P of layer l arrives: finish the output row above (-> out = med3(., 0, lim): ReLU, or zero when that row is outside
// the image), continue this row, start the next
template <int C, int W, int L, bool ALDS>
static __device__ __forceinline__ void chain_dw(ChainState<C, W, L, ALDS>& S, int l, int mt, int tl0, int n, float lim, const f4 (&P)[(W >= 16 ? W / 16 : 1)][ChR<C>::MT],
                                                f4 (&out)[(W >= 16 ? W / 16 : 1)][ChR<C>::MT])
{
    constexpr int NT = (W >= 16 ? W / 16 : 1), CP = ChR<C>::CP;
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    int to = tl0 + l * 10 * CP;
    asm volatile("" : "+v"(to));                                                // the table reads stay inside the step (not hoisted into registers)
    const float* tp = smem32 + to;
    f4 k9[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) k9[k] = ld4(tp + k * CP + 16 * mt);
    f4 lf[NT], rt[NT];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (NT == 1) {
            lf[0][s] = dppf_shr1_z(P[0][mt][s]); rt[0][s] = dppf_shl1_z(P[0][mt][s]);
            if (W == 8) {                                        // two 8-wide images share the 16-lane row: no neighbour across the seam
                lf[0][s] = (n & 7) == 0 ? 0.f : lf[0][s];
                rt[0][s] = (n & 7) == 7 ? 0.f : rt[0][s];
            }
        }
        else {
            lf[0][s] = dppf_shr1_z(P[NT - 1][mt][s]); rt[0][s] = P[NT - 1][mt][s];
            lf[NT - 1][s] = P[0][mt][s];              rt[NT - 1][s] = dppf_shl1_z(P[0][mt][s]);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f4 fin = S.accB[l][t][mt];
        fin = fma4(k9[6], lf[t], fin); fin = fma4(k9[7], P[t][mt], fin); fin = fma4(k9[8], rt[t], fin);
        f4 mid = S.accA[l][t][mt];
        mid = fma4(k9[3], lf[t], mid); mid = fma4(k9[4], P[t][mt], mid); mid = fma4(k9[5], rt[t], mid);
        f4 top = k9[9];
        top = fma4(k9[0], lf[t], top); top = fma4(k9[1], P[t][mt], top); top = fma4(k9[2], rt[t], top);
        S.accB[l][t][mt] = mid;
        S.accA[l][t][mt] = top;
#pragma unroll
        for (int s = 0; s < 4; ++s) out[t][mt][s] = __builtin_amdgcn_fmed3f(fin[s], 0.f, lim);
    }
}

template <int C, int W, int L, bool ALDS>
static __device__ __forceinline__ void chain_rows(const float* __restrict__ xi, float* __restrict__ yo, float* __restrict__ ps,
                                                  const float* __restrict__ w1g, int l0, int H, int kq, int n, int probe, bool second = true)
{
    // W == 8: the 16 lanes of a row are TWO images' eight columns (image A: lanes 0-7, image B = the next one: lanes 8-15; `second`
    // false: there is no image B — its lanes then repeat image A's work on image A's addresses and write nothing of their own)
    const int col0 = W == 8 ? (n & 7) : 0, po = (W == 8 && second) ? (n >> 3) * H * W * C : 0;
    xi += po; yo += po;
    constexpr int CH = ChR<C>::CH, JJ = ChR<C>::JJ, MT = ChR<C>::MT, NT = (W >= 16 ? W / 16 : 1), CP = ChR<C>::CP, KP = ChR<C>::KP, NPR = ChR<C>::NPR;
    extern __shared__ __attribute__((aligned(16))) float smem32[];          // the workgroup's tables: [10][10][CP] taps + bias, then (ALDS) [10][NPR][KP] 1x1 matrices
    ChainState<C, W, L, ALDS> S;
    if (!ALDS) {
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int jj = 0; jj < JJ; ++jj) {
                    const int row = 16 * mt + n, c = kq + 4 * jj;
                    const f4 v = ld4(w1g + (size_t)(l0 + l) * C * C + (unsigned)((row < C ? row : 0) * C + 4 * (c < CH ? c : 0)));
                    S.a[l][mt][jj] = (row < C && c < CH) ? v : zero4();
                }
    }
    const int tl0 = l0 * 10 * CP + 4 * kq;                                     // this lane's chunk of the chain's first tap row (float offset)
    const int wl0 = ChR<C>::TAB + l0 * NPR * KP + 4 * kq;
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f4 bb = ld4(smem32 + tl0 + (l * 10 + 9) * CP + 16 * mt);
#pragma unroll
            for (int t = 0; t < NT; ++t) { S.accA[l][t][mt] = bb; S.accB[l][t][mt] = zero4(); }
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) S.sum[mt] = zero4();
    // this lane's part of an input row: chunks kq + 4 jj of columns NT n + t
    auto ldrow = [&](int r, f4 (&d)[NT][JJ]) {
        const int rc = r < H ? r : H - 1;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) {
                const int c = kq + 4 * jj;
                d[t][jj] = ld4(xi + (unsigned)((rc * W + (W == 8 ? col0 : NT * n + t)) * C + 4 * (c < CH ? c : 0)));
            }
    };
    auto inside = [&](int r) -> float { return (r >= 0 && r < H) ? __builtin_inff() : 0.f; };
    f4 in0[NT][JJ];
    const int iend = H + L;                                       // the chain's last row leaves in step iend - 1
    {
        // Two row buffers used in turn by the two halves of the loop body (no register copies: a copy of a loaded value would wait
        // for it one step after it was requested), rows requested two steps ahead, and the output store of a step UNCONDITIONAL
        // (a row that is not there yet is all zeros — med3's lim — and goes to row 0, which the real row 0 overwrites later): the
        // number of memory operations between a request and its use is fixed, so the wait before the use is a COUNTED vmcnt instead
        // of vmcnt(0) — which was a full memory round trip (and the previous step's stores) in every step, 236 of the kernel's 283 us.
        f4 X[2][NT][JJ];
        const int i0 = -(iend & 1);                                            // an odd number of steps starts with a step that does nothing
        ldrow(i0 < 0 ? 0 : i0, X[0]);
        ldrow(i0 + 1, X[1]);
        auto step = [&](int i, f4 (&Xc)[NT][JJ]) {                             // row i enters, row i - L leaves
            const float lim0 = inside(i);
            f4 cur[2][NT][MT], last[NT][MT];                                    // last: the chain's row of this step (zero until the last layer runs)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) last[t][mt] = zero4();
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) {
                const float lj = (JJ * 4 == CH || kq + 4 * jj < CH) ? lim0 : 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int s = 0; s < 4; ++s) in0[t][jj][s] = __builtin_amdgcn_fmed3f(Xc[t][jj][s], 0.f, lj);   // x1 >= 0 (a ReLU output)
            }
            ldrow(i + 2, Xc);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                if (i >= l) {                                                  // (also what keeps one layer's tables in registers at a time)
                    f4 P[NT][MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) chain_mm<C, W, L, ALDS>(S, l, mt, wl0, n, l == 0 ? in0 : cur[(l - 1) & 1], P);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) chain_dw<C, W, L, ALDS>(S, l, mt, tl0, n, inside(i - l - 1), P, l + 1 < L ? cur[l & 1] : last);
                }
            }
            const int o = i - L < 0 ? 0 : i - L;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (MT * 4 == CH || kq + 4 * mt < CH) {
                        if (!(probe & 1)) st4(yo + (unsigned)((o * W + (W == 8 ? col0 : NT * n + t)) * C + 4 * (kq + 4 * mt)), last[t][mt]);
                        S.sum[mt] = S.sum[mt] + last[t][mt];
                    }
        };
#pragma unroll 1
        for (int i = i0; i < iend; i += 2) {
            step(i, X[0]);
            step(i + 1, X[1]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f4 v = S.sum[mt];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = v[j];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            if (W != 8) s += __shfl_xor(s, 8);
            v[j] = s;
        }
        if (W == 8) { if ((n & 7) == 0 && (n == 0 || second)) st4(ps + (n >> 3) * C + 4 * (kq + 4 * mt), v); }      // (psum rows of consecutive images are C apart)
        else if (n == 0 && (MT * 4 == CH || kq + 4 * mt < CH)) st4(ps + 4 * (kq + 4 * mt), v);
    }
}

template <int C, int W, bool ALDS>
__global__ __launch_bounds__(R32_THREADS, 2) void k32_chainsR(const float* __restrict__ x1, const float* __restrict__ w1 /*[10][C][C]*/,
                                                              const float* __restrict__ w9 /*[10][9][C]*/, const float* __restrict__ bs /*[10][C]*/,
                                                              float* __restrict__ y0, float* __restrict__ y1, float* __restrict__ y2,
                                                              float* __restrict__ y3, float* __restrict__ psum, int Nimg, int H,
                                                              const int* __restrict__ n_img, int probe)
{
    constexpr int CH = ChR<C>::CH, CP = ChR<C>::CP, KP = ChR<C>::KP, NPR = ChR<C>::NPR, TAB = ChR<C>::TAB;
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    float* __restrict__ T = smem32;                              // [10][10][CP]: taps 0..8, bias; zero in the padding channels
    float* __restrict__ Wl = T + TAB;                            // ALDS: [10][NPR][KP], row C (when C % 16) and the padding columns zero
    const int tid = threadIdx.x;
    for (int i = tid; i < TAB; i += R32_THREADS) {
        const int c = i % CP, k = (i / CP) % 10, l = i / (10 * CP);
        float v = 0.f;
        if (c < C) v = k < 9 ? w9[(l * 9 + k) * C + c] : bs[l * C + c];
        T[i] = v;
    }
    if (ALDS) {
        for (int i = tid; i < 10 * NPR * (KP / 4); i += R32_THREADS) {
            const int c4 = i % (KP / 4), r = (i / (KP / 4)) % NPR, l = i / ((KP / 4) * NPR);
            f4 v = zero4();
            if (r < C && c4 < CH) v = ld4(w1 + (size_t)(l * C + r) * C + 4 * c4);
            st4(Wl + (l * NPR + r) * KP + 4 * c4, v);
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kq = lane >> 4, n = lane & 15;
    constexpr int IPW = W == 8 ? 2 : 1;                          // images per wave (two 8-wide images side by side in the 16 lanes of a row)
    const int img = (blockIdx.x * 2 + (wave >> 1)) * IPW, pair = wave & 1;
    int nv = Nimg;
    if (n_img && *n_img < nv) nv = *n_img;
    if (img >= nv) return;
    const bool second = img + 1 < nv;
    const size_t ib = (size_t)img * H * W * C;
    const float* xi = x1 + ib;
    auto ps = [&](int t) { return psum + ((size_t)t * Nimg + img) * C; };
    if (pair == 0) {
        chain_rows<C, W, 4, ALDS>(xi, y3 + ib, ps(3), w1, 6, H, kq, n, probe, second);
        chain_rows<C, W, 1, ALDS>(xi, y0 + ib, ps(0), w1, 0, H, kq, n, probe, second);
    } else {
        chain_rows<C, W, 3, ALDS>(xi, y2 + ib, ps(2), w1, 3, H, kq, n, probe, second);
        chain_rows<C, W, 2, ALDS>(xi, y1 + ib, ps(1), w1, 1, H, kq, n, probe, second);
    }
}

// ---- k32_gates, k32_tail -----------------------------------------------------------------------------------------------------
// k32_gates, per image: gate_t = sigmoid(fc2 relu(fc1 mean_t + b1) + b2) from the chains' channel sums (bands in order) -> gates
// [4][N][MID].  One 128-thread workgroup per image.  (Round 5 computed them at the start of EVERY tail workgroup: three barriers and
// three levels of dependent global loads, four times per image, in front of 2 us of streaming.)
template <int MID>
__global__ __launch_bounds__(128) void k32_gates(const float* __restrict__ psum, int bands, float scale, const float* __restrict__ gw1,
                                                 const float* __restrict__ gb1, const float* __restrict__ gw2, const float* __restrict__ gb2,
                                                 int hidden, float* __restrict__ gates, int Nimg, const int* __restrict__ n_img)
{
    __shared__ float Mn[4 * MID], Hd[16];
    const int img = blockIdx.x, tid = threadIdx.x;
    if (n_img && img >= *n_img) return;
    const int t = tid >> 5, c = tid & 31;
    if (c < MID) {                                            // channel means of the four chain outputs (bands in order)
        float s = 0.f;
        const float* ps = psum + (((size_t)t * Nimg + img) * bands) * MID + c;
        for (int b0 = 0; b0 < bands; b0 += 8) {                // eight band sums per round trip (clamped index), added in band order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ps[(size_t)min(b0 + u, bands - 1) * MID];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (b0 + u < bands) s += v[u];
        }
        Mn[t * MID + c] = s * scale;
    }
    __syncthreads();
    if (tid < 4 * hidden) {
        const int tt = tid / hidden, j = tid - tt * hidden;
        float s = gb1[j];
        f4 wv[MID / 4];
#pragma unroll
        for (int q4 = 0; q4 < MID / 4; ++q4) wv[q4] = ld4(gw1 + j * MID + 4 * q4);
#pragma unroll
        for (int q4 = 0; q4 < MID / 4; ++q4) {
            s = __builtin_fmaf(wv[q4][0], Mn[tt * MID + 4 * q4], s); s = __builtin_fmaf(wv[q4][1], Mn[tt * MID + 4 * q4 + 1], s);
            s = __builtin_fmaf(wv[q4][2], Mn[tt * MID + 4 * q4 + 2], s); s = __builtin_fmaf(wv[q4][3], Mn[tt * MID + 4 * q4 + 3], s);
        }
        Hd[tt * 4 + j] = s > 0.f ? s : 0.f;
    }
    __syncthreads();
    if (c < MID) {
        float s = gb2[c];
        for (int j = 0; j < hidden; ++j) s = __builtin_fmaf(gw2[c * hidden + j], Hd[t * 4 + j], s);
        gates[((size_t)t * Nimg + img) * MID + c] = 1.0f / (1.0f + expf(-s));
    }
}

// k32_tail, per pixel: x2 = sum_t gate_t * y_t; o = relu(W3 x2 + b3 + shortcut), shortcut = idn (C1 == 0) or Wd x + bd (the block
// input x, C1 channels); o -> d_out when asked; o2 = relu(W4 o + b4) -> d_out2, averaged over 2x2 pixels when POOL.  A wave owns
// 16-pixel tiles (POOL: 2 rows x 8 columns).  PERSISTENT: the grid is a fixed number of workgroups (two per CU when the weights
// allow), each stages the three weight matrices ONCE and then walks its share of the images (all tiles of an image by its eight
// waves), so the staging is paid once per CU instead of once per 8-32 tiles.
#define T32_THREADS 512
template <int MID, int C2, int C1, int N2, bool POOL>
__global__ __launch_bounds__(T32_THREADS, 4) void k32_tail(const float* __restrict__ y0, const float* __restrict__ y1, const float* __restrict__ y2,
                                                const float* __restrict__ y3, const float* __restrict__ gates, const float* __restrict__ w3,
                                                const float* __restrict__ b3, const float* __restrict__ xin, const float* __restrict__ wd,
                                                const float* __restrict__ bd, float* __restrict__ out, const float* __restrict__ w4,
                                                const float* __restrict__ b4, float* __restrict__ out2, int Nimg, int H, int W,
                                                const int* __restrict__ n_img, int splits)
{
    constexpr int CHM = MID / 4, JM = (CHM + 3) / 4, MT3 = C2 / 16, CH1 = C1 / 4, J1 = (CH1 + 3) / 4, MT4 = (N2 + 15) / 16, CH4 = N2 / 4;
    constexpr int L3 = w_lds_floats<MID, C2>(), LD = C1 ? w_lds_floats<(C1 ? C1 : 16), C2>() : 0, WV = T32_THREADS / 64;
    static_assert(C2 % 16 == 0, "C2");
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    constexpr int L4 = w_lds_floats<C2, N2>(), LB = L3 + LD + L4, GP = 16 * JM;     // biases behind the matrices, then one gate table per wave
    float* __restrict__ W3s = smem32;
    float* __restrict__ Wds = W3s + L3;
    float* __restrict__ W4s = Wds + LD;
    float* __restrict__ Bs = smem32 + LB;                        // [C2] b3 (+ bd), [16 MT4] b4 (zero past N2)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n = lane & 15;
    float* __restrict__ Gw = Bs + C2 + 16 * MT4 + wave * 4 * GP; // this wave's copy of the image's gates: [4][GP], zero in the padding chunks
    int nv = Nimg;
    if (n_img && *n_img < nv) nv = *n_img;
    // work items = (image, part of its tiles): `splits` parts per image when the batch alone cannot fill the chip (the per-frame call's 32 crops)
    const int items = nv * splits, per = (items + (int)gridDim.x - 1) / (int)gridDim.x, it0 = blockIdx.x * per, it1 = min(it0 + per, items);
    if (it0 >= it1) return;
    stage_w<MID, C2, T32_THREADS>(W3s, w3, tid);
    if (C1) stage_w<(C1 ? C1 : 16), C2, T32_THREADS>(Wds, wd, tid);
    stage_w<C2, N2, T32_THREADS>(W4s, w4, tid);
    if (tid < C2) Bs[tid] = b3[tid] + (C1 ? bd[tid] : 0.f);
    if (tid < 16 * MT4) Bs[C2 + tid] = tid < N2 ? b4[tid] : 0.f;
    __syncthreads();
    const int HW = H * W, tiles_img = HW / 16;
    const float* const ys[4] = { y0, y1, y2, y3 };
    for (int item = it0; item < it1; ++item) {
        const int img = item / splits, part = item - img * splits;
        // the image's four gate vectors -> this wave's LDS table (a wave reads only what it wrote: no barrier)
        for (int i = lane; i < 4 * GP; i += 64) {
            const int t = i / GP, c = i - t * GP;
            Gw[i] = c < MID ? gates[((size_t)t * Nimg + img) * MID + c] : 0.f;
        }
        const size_t pbase = (size_t)img * HW;
        for (int tile = part * WV + wave; tile < tiles_img; tile += WV * splits) {
            int pl;                                              // pixel of this lane inside the image
            if (POOL) { const int tw = W / 8, ty = tile / tw, tx = tile - ty * tw; pl = (2 * ty + (n >> 3)) * W + 8 * tx + (n & 7); }
            else pl = tile * 16 + n;
            const size_t px = pbase + pl;
            // every operand of the tile requested before the first is used
            f4 yv[4][JM], xv[C1 ? (J1 ? J1 : 1) : MT3];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int jj = 0; jj < JM; ++jj) { const int c = kq + 4 * jj; yv[t][jj] = ld4(ys[t] + px * MID + 4 * (c < CHM ? c : 0)); }
            if (C1) {
#pragma unroll
                for (int jj = 0; jj < J1; ++jj) { const int c = kq + 4 * jj; xv[jj] = ld4(xin + px * C1 + 4 * (c < CH1 ? c : 0)); }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT3; ++mt) xv[mt] = ld4(xin + px * C2 + 16 * mt + 4 * kq);
            }
            // (the weight fragments, biases and gates are re-read from LDS for every tile: an offset the compiler cannot see through keeps
            //  it from hoisting the loop-invariant reads into registers — 256 registers and 110-290 spilled values when it did)
            int o3 = 0, od = L3, o4 = L3 + LD, ob = LB + 4 * kq, og = LB + C2 + 16 * MT4 + wave * 4 * GP + 4 * kq;
            asm volatile("" : "+v"(o3), "+v"(od), "+v"(o4), "+v"(ob), "+v"(og));
            f4 bm[JM];
#pragma unroll
            for (int jj = 0; jj < JM; ++jj) {
                f4 v = ld4(smem32 + og + 16 * jj) * yv[0][jj];
#pragma unroll
                for (int t = 1; t < 4; ++t) v = v + ld4(smem32 + og + t * GP + 16 * jj) * yv[t][jj];
                bm[jj] = v;                                      // (padding chunks: gate 0 -> 0)
            }
            f4 acc[MT3];
#pragma unroll
            for (int mt = 0; mt < MT3; ++mt) acc[mt] = zero4();
            mm_tile<MID, C2>(smem32 + o3, bm, acc, kq, n);
            if (C1) {                                            // shortcut = Wd x + bd as a second product into the same accumulators
                f4 bx[J1 ? J1 : 1];
#pragma unroll
                for (int jj = 0; jj < J1; ++jj) bx[jj] = kq + 4 * jj < CH1 ? xv[jj] : zero4();
                mm_tile<(C1 ? C1 : 16), C2>(smem32 + od, bx, acc, kq, n);
            }
#pragma unroll
            for (int mt = 0; mt < MT3; ++mt) {
                const int oc = 16 * mt + 4 * kq;
                f4 v = acc[mt] + ld4(smem32 + ob + 16 * mt);
                if (!C1) v = v + xv[mt];
                v = relu4(v);
                acc[mt] = v;
                if (out) st4(out + px * C2 + oc, v);
            }
            // the 1x1 ConvBR that follows: the block output's accumulators are its B chunks
            f4 acc2[MT4];
#pragma unroll
            for (int mt = 0; mt < MT4; ++mt) acc2[mt] = zero4();
            mm_tile<C2, N2>(smem32 + o4, acc, acc2, kq, n);
#pragma unroll
            for (int mt = 0; mt < MT4; ++mt) {
                const int oc = 16 * mt + 4 * kq;
                f4 v = zero4();
                if (4 * mt + kq < CH4) v = relu4(acc2[mt] + ld4(smem32 + ob + C2 + 16 * mt));
                if (POOL) {
                    f4 s;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { float a = v[j]; a += __shfl_xor(a, 1); a += __shfl_xor(a, 8); s[j] = a * 0.25f; }
                    if (4 * mt + kq < CH4 && n < 8 && !(n & 1)) {
                        const int tw = W / 8, ty = tile / tw, tx = tile - ty * tw;
                        const size_t po = (size_t)img * (HW / 4) + (size_t)ty * (W / 2) + 4 * tx + (n >> 1);
                        st4(out2 + po * N2 + oc, s);
                    }
                } else if (4 * mt + kq < CH4) st4(out2 + px * N2 + oc, v);
            }
        }
    }
}

// ---- k32_stem ------------------------------------------------------------------------------------------------------------
// conv 7x7 / stride 2 / pad 3 (3 -> 16) + bias + ReLU + max pool 3x3 / stride 2 / pad 1 on NHWC fp32 crops [N][256][128][3].
// Workgroup = (image, band of 4 pooled rows): the 23 input rows the band's 9 convolution rows need are staged once (zero
// outside the image; a row = 12 zero floats, 384 values, 8 zero floats), the K axis of the implicit GEMM is k = 21 ky + 3 kx + c --
// a patch row is 21 CONSECUTIVE floats of an NHWC input row -- so B[k][n] is one ds_read_b32 at base(n) + k + (k / 21)(pitch - 21)
// (columns of a tile are 6 floats apart: 32 lanes hit 32 banks), A (16 x 148 weights) lives in 37 registers per lane.
#define ST_ROWP 404
#define ST_PR 4
__global__ __launch_bounds__(C32_THREADS) void k32_stem(const float* __restrict__ x, const float* __restrict__ w /*[16][148]*/,
                                                      const float* __restrict__ bias, float* __restrict__ y, int Nimg, const int* __restrict__ n_img)
{
    constexpr int H = 256, Wd = 128, OW = 64, PH = 64, PW = 32, IN_ROWS = 24, CROWS = 2 * ST_PR + 1, CSP = 16;
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    float* __restrict__ In = smem32;                         // [24][404]
    float* __restrict__ Cs = In + IN_ROWS * ST_ROWP;         // [9][64][16] convolution rows after bias + ReLU
    const int img = blockIdx.y, band = blockIdx.x;
    if (n_img && img >= *n_img) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n = lane & 15;
    const int pr0 = band * ST_PR, cy0 = 2 * pr0 - 1, iy0 = 2 * cy0 - 3;
    const float* xi = x + (size_t)img * H * Wd * 3;
    {   // the band's input rows: every vector of a thread requested (from a clamped, always valid address) before the first LDS store — the
        // rolled `conditional load -> store` loop was four dependent HBM round trips at the start of every workgroup
        constexpr int TOT = IN_ROWS * (ST_ROWP / 4), NIT = (TOT + C32_THREADS - 1) / C32_THREADS;
        f4 v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = min(tid + u * C32_THREADS, TOT - 1), lr = i / (ST_ROWP / 4), c4 = i - lr * (ST_ROWP / 4), iy = iy0 + lr;
            const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), cc = c4 < 3 ? 0 : (c4 >= 99 ? 95 : c4 - 3);
            v[u] = ld4(xi + (size_t)iyc * (Wd * 3) + 4 * cc);
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + u * C32_THREADS, lr = i / (ST_ROWP / 4), c4 = i - lr * (ST_ROWP / 4), iy = iy0 + lr;
            if (i < TOT) st4(In + lr * ST_ROWP + 4 * c4, (iy >= 0 && iy < H && c4 >= 3 && c4 < 99) ? v[u] : zero4());
        }
    }
    float a[37];
#pragma unroll
    for (int s = 0; s < 37; ++s) a[s] = w[n * 148 + 4 * s + kq];
    const f4 bb = ld4(bias + 4 * kq);
    __syncthreads();
    for (int tile = wave; tile < CROWS * 4; tile += C32_THREADS / 64) {
        const int cyl = tile >> 2, cx0 = (tile & 3) * 16, cy = cy0 + cyl;
        if (cy < 0 || cy >= H / 2) continue;
        const float* bp = In + 2 * cyl * ST_ROWP + 3 + 6 * (cx0 + n) + kq;
        f4 acc = zero4();
#pragma unroll
        for (int s = 0; s < 37; ++s) {
            const int k = 4 * s + kq, ky = k / 21;
            acc = MFMA4(a[s], bp[4 * s + ky * (ST_ROWP - 21)], acc);
        }
        st4(Cs + (cyl * OW + cx0 + n) * CSP + 4 * kq, relu4(acc + bb));
    }
    __syncthreads();
    if (tid < ST_PR * PW * 4) {
        const int q = tid & 3, pxl = (tid >> 2) & 31, prl = tid >> 7;
        f4 m = f4{ -INFINITY, -INFINITY, -INFINITY, -INFINITY };
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int cyl = 2 * prl + dy, cy = cy0 + cyl;
            if (cy < 0 || cy >= H / 2) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int cx = 2 * pxl - 1 + dx;
                if (cx < 0 || cx >= OW) continue;
                m = max4(m, ld4(Cs + (cyl * OW + cx) * CSP + 4 * q));
            }
        }
        st4(y + (((size_t)img * PH + pr0 + prl) * PW + pxl) * 16 + 4 * q, m);
    }
}

// ---- k32_stemW: the same operator, a workgroup WALKING down `nb` consecutive bands of one image -------------------------------------
// k32_stem's workgroups run their phases one after the other (stage 23 rows -> 37 weight loads -> matrix -> pool) and the two workgroups of a
// CU start together, so they wait together: MFMA-busy 0.41.  Here the 16 NEW input rows of the next band are requested before the current
// band's matrix phase and stored after it (their HBM latency lies under the MFMAs), the seven rows two bands share stay where they are (the
// 24 input rows are a RING: local row r of a band is ring row (base + r) mod 24, base += 16 per band; a tile's seven row offsets are
// wave-uniform scalars), the convolution row two bands share is computed once (Cs is a ring of 9 slots: slot = (row + 1) mod 9), the 37
// weight registers are loaded once per workgroup, and a wave runs TWO tiles side by side (independent accumulator chains: a dependent MFMA
// issues 40 cycles after its predecessor, an independent one 32).  Per band: [matrix] barrier [store new rows, pool] barrier.  Every sum
// keeps k32_stem's order (k = 0 .. 147 into one accumulator), so the two kernels agree bit for bit.
// U8: the crops arrive as bytes (ss_crop_norm* with SS_DST_U8: the rounded bilinear value per RGB channel, NHWC) and the staging applies
// ((q / 255) - mean) / sd from a 3 x 256 table built with a4's own expression (csrc/ss_front.hip k_crop_hwc8) — the floats that reach the
// ring are the float crop's, bit for bit, and the crops cross HBM as a quarter of the bytes (the crop launch 147 -> 94 us per 862 crops).
#define STW_THREADS 512
template <bool U8>
__global__ __launch_bounds__(STW_THREADS, 4) void k32_stemW(const void* __restrict__ xv, const float* __restrict__ w /*[16][148]*/,
                                                            const float* __restrict__ bias, float* __restrict__ y, int Nimg, const int* __restrict__ n_img, int nb,
                                                            const float* __restrict__ w1 /*[16][16] or null*/, const float* __restrict__ b1, float* __restrict__ y1)
{
    constexpr int H = 256, Wd = 128, OW = 64, PH = 64, PW = 32, RING = 24, CSP = 16, RV = Wd * 3 / 4;   // RV: data vectors of an input row (96)
    constexpr int RB = Wd * 3 / 16;                          // U8: 16-byte vectors of an input row (24)
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    float* __restrict__ In = smem32;                         // [24][404] ring of input rows
    float* __restrict__ Cs = In + RING * ST_ROWP;            // [9][64][16] ring of convolution rows after bias + ReLU
    float* __restrict__ Lut = Cs + 9 * OW * CSP;             // U8: [3][256]
    const int img = blockIdx.y, b0 = blockIdx.x * nb;
    if (n_img && img >= *n_img) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kq = lane >> 4, n = lane & 15;
    const float* xi = reinterpret_cast<const float*>(xv) + (size_t)img * H * Wd * 3;
    const uint8_t* xb = reinterpret_cast<const uint8_t*>(xv) + (size_t)img * H * Wd * 3;
    auto inrow = [&](int iy) { return iy < 0 ? 0 : (iy >= H ? H - 1 : iy); };
    if (tid < RING * 5) {                                    // the rows' zero borders (3 vectors left, 2 right), written once
        const int r = tid / 5, j = tid - 5 * r;
        st4(In + r * ST_ROWP + 4 * (j < 3 ? j : RV + j), zero4());
    }
    // U8: 16 bytes of a row (vector c16: positions 16 c16 .. + 15, channel = position mod 3 = (c16 + j) mod 3) -> 16 floats of the ring row
    auto cvt16 = [&](f4 (&f)[4], const uint4 v, int c16, bool ok) __attribute__((always_inline)) {
        const int c0 = c16 % 3, o0 = c0 * 256, o1 = (c0 == 2 ? 0 : c0 + 1) * 256, o2 = (c0 == 0 ? 2 : c0 - 1) * 256;
        const unsigned wv[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * q + e;
                const int b = (int)((wv[q] >> (8 * e)) & 255u);
                f[q][e] = ok ? Lut[(j % 3 == 0 ? o0 : j % 3 == 1 ? o1 : o2) + b] : 0.f;
            }
    };
    auto put16 = [&](float* dst, const uint4 v, int c16, bool ok) __attribute__((always_inline)) {
        f4 f[4];
        cvt16(f, v, c16, ok);
#pragma unroll
        for (int q = 0; q < 4; ++q) st4(dst + 4 * q, f[q]);
    };
    if (U8) {
        const float mean[3] = { 0.485f, 0.456f, 0.406f }, sd[3] = { 0.229f, 0.224f, 0.225f };
        for (int i = tid; i < 768; i += STW_THREADS) { const int c = i >> 8; const float q = (float)(i & 255) / 255.0f; Lut[i] = (q - mean[c]) / sd[c]; }
        __syncthreads();
        constexpr int TOT = 23 * RB, NIT = (TOT + STW_THREADS - 1) / STW_THREADS;
        const int iy0 = 16 * b0 - 5;
        uint4 v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = min(tid + u * STW_THREADS, TOT - 1), lr = i / RB, c16 = i - lr * RB;
            v[u] = *reinterpret_cast<const uint4*>(xb + (size_t)inrow(iy0 + lr) * (Wd * 3) + 16 * c16);
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + u * STW_THREADS, lr = i / RB, c16 = i - lr * RB, iy = iy0 + lr;
            if (i < TOT) put16(In + lr * ST_ROWP + 12 + 16 * c16, v[u], c16, iy >= 0 && iy < H);
        }
    } else {   // the first band's 23 rows (ring rows 0 .. 22): every vector requested before the first LDS store
        constexpr int TOT = 23 * RV, NIT = (TOT + STW_THREADS - 1) / STW_THREADS;
        const int iy0 = 16 * b0 - 5;
        f4 v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = min(tid + u * STW_THREADS, TOT - 1), lr = i / RV, c4 = i - lr * RV;
            v[u] = ld4(xi + (size_t)inrow(iy0 + lr) * (Wd * 3) + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + u * STW_THREADS, lr = i / RV, c4 = i - lr * RV, iy = iy0 + lr;
            if (i < TOT) st4(In + lr * ST_ROWP + 12 + 4 * c4, (iy >= 0 && iy < H) ? v[u] : zero4());
        }
    }
    float a[37];
#pragma unroll
    for (int s = 0; s < 37; ++s) a[s] = w[n * 148 + 4 * s + kq];
    const f4 bb = ld4(bias + 4 * kq);
#pragma unroll
    for (int s = 0; s < 37; ++s) asm volatile("" ::"v"(a[s]));
    asm volatile("" ::"v"(bb[0]), "v"(bb[1]), "v"(bb[2]), "v"(bb[3]));   // the weights have arrived HERE: inside the loop their wait would also cover the rows requested ahead
    // the first OSBlock's conv1 (1x1, 16 -> 16, + bias + ReLU) on the pooled pixels, from the same launch: the pool threads of a wave hold a
    // 16-pixel tile in the B layout (lane (q, n): chunk q of pixel n), k32_pw's four-MFMA chain runs on it — x1 bit-equal to k32_pw's, x0 not re-read
    f4 wc1 = zero4(), bc1 = zero4();
    if (w1) {
        wc1 = ld4(w1 + n * 16 + 4 * kq); bc1 = ld4(b1 + 4 * kq);
        asm volatile("" ::"v"(wc1[0]), "v"(wc1[1]), "v"(wc1[2]), "v"(wc1[3]), "v"(bc1[0]), "v"(bc1[1]), "v"(bc1[2]), "v"(bc1[3]));
    }
    int base = 0;
    for (int bi = 0; bi < nb; ++bi) {
        const int band = b0 + bi, cy0 = 8 * band - 1, iy0 = 16 * band - 5;
        const bool more = bi + 1 < nb;
        __syncthreads();                                     // this band's rows are in the ring; the previous band's pool is done with Cs
        f4 pf[U8 ? 1 : 3];
        uint4 pb = { 0u, 0u, 0u, 0u };
        if (more) {
            if (U8) {                                        // the next band's 16 new rows: 384 vectors of 16 bytes
                const int i = min(tid, 16 * RB - 1), j = i / RB, c16 = i - j * RB;
                pb = *reinterpret_cast<const uint4*>(xb + (size_t)inrow(iy0 + 23 + j) * (Wd * 3) + 16 * c16);
            } else {
#pragma unroll
                for (int u = 0; u < 3; ++u) {                // the next band's 16 new rows: 1 536 vectors
                    const int i = tid + u * STW_THREADS, j = i / RV, c4 = i - j * RV;
                    pf[u] = ld4(xi + (size_t)inrow(iy0 + 23 + j) * (Wd * 3) + 4 * c4);
                }
            }
        }
        const int first = bi ? 1 : 0;                        // convolution row 0 of a later band is the previous band's row 8
        for (int t = 2 * wave; t < (9 - first) * 4; t += 2 * (STW_THREADS / 64)) {
            const int cyl = first + (t >> 2), cx0 = (t & 3) * 16, cy = cy0 + cyl;
            if (cy < 0 || cy >= H / 2) continue;
            const int r0 = base + 2 * cyl;
            auto ro = [&](int ky) __attribute__((always_inline)) {      // (a function, not an array: a select between two array elements becomes a dynamically indexed stack slot)
                int r = r0 + ky; r = r >= RING ? r - RING : r; r = r >= RING ? r - RING : r;
                return r * ST_ROWP - 21 * ky;
            };
            const int lb = 3 + 6 * (cx0 + n) + kq;
            f4 acc0 = zero4(), acc1 = zero4();
            // B operands four k-steps (eight MFMAs) ahead of their MFMAs (two register groups; the fences keep "request the next group, then
            // multiply this one" — left alone the scheduler reads each operand right before its use and waits for the LDS every step)
            constexpr int SG = 4, NG = (37 + SG - 1) / SG;
            float bv[2][SG][2];
            auto request = [&](int g) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < SG; ++j) {
                    const int s = SG * g + j;
                    if (s < 37) {
                        const int klo = 4 * s, ky_lo = klo / 21, ky_hi = (klo + 3) / 21 > 6 ? 6 : (klo + 3) / 21;
                        const int off = (ky_lo == ky_hi) ? ro(ky_lo) : (kq >= 21 * ky_hi - klo ? ro(ky_hi) : ro(ky_lo));
                        const float* bp = In + lb + off + klo;
                        bv[g & 1][j][0] = bp[0]; bv[g & 1][j][1] = bp[96];
                    }
                }
            };
            request(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) request(g + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < SG; ++j)
                    if (SG * g + j < 37) { acc0 = MFMA4(a[SG * g + j], bv[g & 1][j][0], acc0); acc1 = MFMA4(a[SG * g + j], bv[g & 1][j][1], acc1); }
                __builtin_amdgcn_sched_barrier(0);
            }
            int slot = (cy + 1) % 9;
            float* cs = Cs + (slot * OW + cx0 + n) * CSP + 4 * kq;
            st4(cs, relu4(acc0 + bb));
            st4(cs + 16 * CSP, relu4(acc1 + bb));
        }
        __syncthreads();                                     // Cs complete; local rows 0 .. 15 of the ring are dead
        if (more) {
            if (U8) {
                if (tid < 16 * RB) {
                    const int j = tid / RB, c16 = tid - j * RB, iy = iy0 + 23 + j;
                    int r = base + 23 + j; r = r >= RING ? r - RING : r; r = r >= RING ? r - RING : r;
                    put16(In + r * ST_ROWP + 12 + 16 * c16, pb, c16, iy >= 0 && iy < H);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int i = tid + u * STW_THREADS, j = i / RV, c4 = i - j * RV, iy = iy0 + 23 + j;
                    int r = base + 23 + j; r = r >= RING ? r - RING : r; r = r >= RING ? r - RING : r;
                    st4(In + r * ST_ROWP + 12 + 4 * c4, (iy >= 0 && iy < H) ? pf[u] : zero4());
                }
            }
        }
        {
            const int q = kq, pxl = 16 * (wave & 1) + n, prl = wave >> 1;       // a wave = 16 pixels of one pooled row x 4 channel chunks (lane (q, n))
            f4 m = f4{ -INFINITY, -INFINITY, -INFINITY, -INFINITY };
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int cy = cy0 + 2 * prl + dy;
                if (cy < 0 || cy >= H / 2) continue;
                const int slot = (cy + 1) % 9;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int cx = 2 * pxl - 1 + dx;
                    if (cx < 0 || cx >= OW) continue;
                    m = max4(m, ld4(Cs + (slot * OW + cx) * CSP + 4 * q));
                }
            }
            const size_t po = (((size_t)img * PH + 4 * band + prl) * PW + pxl) * 16 + 4 * q;
            st4(y + po, m);
            if (w1) {
                f4 acc = zero4();
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = MFMA4(wc1[e], m[e], acc);
                st4(y1 + po, relu4(acc + bc1));
            }
        }
        base += 16; base = base >= RING ? base - RING : base;
    }
}

// ---- k32_head ------------------------------------------------------------------------------------------------------------
// out[i][f] = relu(b[f] + sum_c w[f][c] * mean_p x[i][p][c]); 4 images per workgroup, C = 128.
__global__ __launch_bounds__(256) void k32_head(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                float* __restrict__ out, int Nimg, int HW, int F, const int* __restrict__ n_img)
{
    constexpr int C = 128;
    __shared__ __attribute__((aligned(16))) float mean[4][C];
    const int tid = threadIdx.x, i0 = blockIdx.x * 4;
    int nv = Nimg;
    if (n_img && *n_img < nv) nv = *n_img;
    if (i0 >= nv) return;
    const float inv = 1.0f / (float)HW;
#pragma unroll 1
    for (int k = tid; k < 4 * C; k += 256) {
        const int i = k / C, c = k - i * C, img = i0 + i;
        float s = 0.f;
        if (img < nv) {
            const float* p = x + (size_t)img * HW * C + c;
#pragma unroll 8
            for (int q = 0; q < HW; ++q) s += p[(size_t)q * C];
        }
        mean[i][c] = s * inv;
    }
    __syncthreads();
#pragma unroll 1
    for (int f = tid; f < F; f += 256) {
        float s[4] = { bias[f], bias[f], bias[f], bias[f] };
        const float* wr = w + (size_t)f * C;
#pragma unroll 4
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const f4 wv = ld4(wr + 4 * c4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f4 mv = ld4(&mean[i][4 * c4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[i] = __builtin_fmaf(wv[j], mv[j], s[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i0 + i < nv) out[(size_t)(i0 + i) * F + f] = s[i] > 0.f ? s[i] : 0.f;
    }
}

// ---- k32_conv, k32_conv0: the DETECTOR's convolutions in fp32 ---------------------------------------------------------------
// The reference runs its detector in fp32 (no half=: /root/reference/yolo_multi_model.py:18-21, :41) and the f16 kernels of ss_ops.hip
// do not reproduce the fp32 network's NMS keep lists — a CPU rounding model says the loss is in the backbone, so "f16 backbone + fp32
// head" does not help (tools/det_f16_rounding_model.py, profiles/r06_det_f16_rounding_model.txt).  Until round 6 the fp32 detector ran
// on the library convolutions with bias / SiLU / concat as separate passes (6 ms per 32 frames).  k32_conv is ONE implicit-GEMM kernel
// for every 1x1 and 3x3 (stride 1 / 2, pad k / 2) convolution of the YOLO graphs whose channel counts are multiples of 16:
//   out[p][co] = act(bias[co] + sum_{ky,kx,ci} w[co][(ky, kx, ci)] x[p * s + (ky, kx) - pad][ci]) (+ res[p][co] after the activation)
// NHWC fp32 with a PIXEL STRIDE per tensor, so input, output and shortcut may be channel slices of wider tensors: a C2f block runs
// without a single chunk / add / cat pass.  Same operand convention as the ReID kernels (lane (kq, n): chunk kq + 4 j of pixel n; the
// weights are the A operand from LDS).  K is walked in groups of 64 (four 16-channel chunks of one tap or of consecutive taps); the
// weight slice of the next group and the next group's pixel vectors are requested before the matrix work of the current one (registers
// / the other LDS buffer), one barrier per group.
__device__ __attribute__((aligned(16))) float g_zero4[4] = { 0.f, 0.f, 0.f, 0.f };      // (not const: a constant-address-space pointer would turn the selects into generic pointers and the loads into flat loads)   // where the vector of a padding tap / a row past the matrix is read from

template <int KS, int STRIDE, int MT, int PT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k32_conv(const float* __restrict__ x, int xs, const float* __restrict__ w, const float* __restrict__ bias,
                                                const float* __restrict__ res, int rs, float* __restrict__ out, int os, int N, int H, int W, int OH,
                                                int OW, int Cin, int Cout, int act)
{
    constexpr int ROWS = 16 * MT, PITCH = 72, THREADS = 64 * WAVES, WPT = (ROWS * 16 + THREADS - 1) / THREADS;      // WPT: weight vectors a thread stages per group
    __shared__ __attribute__((aligned(16))) float Ws[2][ROWS * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n = lane & 15;
    const int co0 = blockIdx.y * ROWS, K = KS * KS * Cin, c16s = Cin >> 4, NKC = K >> 4, G = (NKC + 3) >> 2;
    const long long M = (long long)N * OH * OW;
    const long long tile0 = ((long long)blockIdx.x * WAVES + wave) * PT;
    long long p[PT];
    int oy[PT], ox[PT];
    const float* xim[PT];
    bool valid[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        p[t] = (tile0 + t) * 16 + n;
        valid[t] = p[t] < M;
        const long long pc = valid[t] ? p[t] : 0;
        const int img = (int)(pc / (OH * OW)), r = (int)(pc - (long long)img * (OH * OW));
        oy[t] = r / OW; ox[t] = r - oy[t] * OW;
        xim[t] = x + (size_t)img * H * W * xs + 4 * kq;
    }
    auto load_b = [&](int g, f4 (&b)[PT][4]) {                        // this lane's chunk of the four k-chunks of group g, per pixel tile
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) {
            const int kc = 4 * g + jc, kcc = kc < NKC ? kc : NKC - 1, tap = kcc / c16s, c16 = kcc - tap * c16s, ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
            for (int t = 0; t < PT; ++t) {
                const int iy = oy[t] * STRIDE + ky - KS / 2, ix = ox[t] * STRIDE + kx - KS / 2;
                // a tap outside the image (or a k-chunk past K) reads the zero vector: the choice is made on the ADDRESS, so nothing
                // touches the loaded value before the matrix instructions of the NEXT group do (a select on the value made the
                // compiler wait for every load right here, i.e. one L2 round trip per 64-wide group in front of the MFMAs)
                const bool ok = valid[t] && kc < NKC && iy >= 0 && iy < H && ix >= 0 && ix < W;
                const float* q = ok ? xim[t] + (size_t)(iy * W + ix) * xs + 16 * c16 : g_zero4;
                b[t][jc] = ld4(q);
            }
        }
    };
    auto load_w = [&](int g, f4 (&wr)[WPT]) {                          // rows co0 .. of the weight matrix, columns 64 g .. 64 g + 63
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int i = tid + u * THREADS, row = i >> 4, c4 = i & 15, k = 64 * g + 4 * c4;
            const bool ok = k < K && co0 + row < Cout && row < ROWS;
            wr[u] = ld4(ok ? w + (size_t)(co0 + row) * K + k : g_zero4);
        }
    };
    auto store_w = [&](int buf, const f4 (&wr)[WPT]) {
#pragma unroll
        for (int u = 0; u < WPT; ++u) { const int i = tid + u * THREADS; if (ROWS * 16 % THREADS == 0 || i < ROWS * 16) st4(&Ws[buf][(i >> 4) * PITCH + 4 * (i & 15)], wr[u]); }
    };
    f4 acc[MT][PT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[mt][t] = zero4();
    auto mma = [&](int buf, const f4 (&b)[PT][4]) {
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) {
            f4 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = ld4(&Ws[buf][(16 * mt + n) * PITCH + 16 * jc + 4 * kq]);
#pragma unroll
            for (int s = 0; s < 4; ++s)                                  // consecutive MFMAs on different accumulators (a dependent one is 40 cycles behind, an independent 32)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < PT; ++t) acc[mt][t] = MFMA4(a[mt][s], b[t][jc][s], acc[mt][t]);
        }
    };
    f4 b0[PT][4], b1[PT][4], wr[WPT];
    load_b(0, b0);
    load_w(0, wr);
    store_w(0, wr);
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g < G; g += 2) {
        if (g + 1 < G) { load_b(g + 1, b1); load_w(g + 1, wr); }
        mma(0, b0);
        if (g + 1 < G) store_w(1, wr);
        __syncthreads();
        if (g + 1 >= G) break;
        if (g + 2 < G) { load_b(g + 2, b0); load_w(g + 2, wr); }
        mma(1, b1);
        if (g + 2 < G) store_w(0, wr);
        __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int oc = co0 + 16 * mt + 4 * kq;
        if (oc >= Cout) continue;
        const f4 bv = ld4(bias + oc);
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            if (!valid[t]) continue;
            f4 v = acc[mt][t] + bv;
            if (act) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] / (1.0f + expf(-v[j]));          // SiLU
            }
            if (res) v = v + ld4(res + (size_t)p[t] * rs + oc);
            st4(out + (size_t)p[t] * os + oc, v);
        }
    }
}

// The first convolution (3 input channels, 3x3, stride 2, pad 1): a thread computes the 16 output channels of one pixel from its 27
// input values; the weights [16][27] are wave-uniform (scalar loads).  x NHWC [N][H][W][3] dense, out NHWC with a pixel stride.
__global__ __launch_bounds__(256) void k32_conv0(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                 float* __restrict__ out, int os, int N, int H, int W, int OH, int OW, int act)
{
    const long long M = (long long)N * OH * OW, pp = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pp >= M) return;
    const int img = (int)(pp / (OH * OW)), r = (int)(pp - (long long)img * (OH * OW)), oy = r / OW, ox = r - oy * OW;
    float v[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float* q = x + ((size_t)img * H * W + (size_t)(ok ? iy : 0) * W + (ok ? ix : 0)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[(ky * 3 + kx) * 3 + c] = ok ? q[c] : 0.f;
        }
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = 4 * c4 + j;
            float s = bias[co];
#pragma unroll
            for (int k = 0; k < 27; ++k) s = __builtin_fmaf(w[co * 27 + k], v[k], s);
            o[j] = act ? s / (1.0f + expf(-s)) : s;
        }
        st4(out + (size_t)pp * os + 4 * c4, o);
    }
}

// cat(nearest-2x-upsample(lo), hi) (lo_first) or cat(hi, up(lo)) along channels in one pass: a thread moves one 16-byte chunk of an
// output pixel.  lo [N][H/2][W/2][.] pixel stride ls, hi [N][H][W][.] pixel stride hs (channel slices allowed), out dense [N][H][W][Cl + Ch].
__global__ __launch_bounds__(256) void k32_upcat(const float* __restrict__ lo, int ls, int Cl, const float* __restrict__ hi, int hs, int Ch,
                                                 float* __restrict__ out, int N, int H, int W, int lo_first)
{
    const int C4 = (Cl + Ch) >> 2;
    const long long tot = (long long)N * H * W * C4, i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= tot) return;
    const int c4 = (int)(i % C4);
    const long long px = i / C4;
    const int x = (int)(px % W), y = (int)((px / W) % H), img = (int)(px / ((long long)W * H));
    const int c = 4 * c4, cl = lo_first ? c : c - Ch;                   // channel inside lo (when in its range)
    f4 v;
    if (cl >= 0 && cl < Cl) v = ld4(lo + ((size_t)(img * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * ls + cl);
    else v = ld4(hi + (size_t)px * hs + (lo_first ? c - Cl : c));
    st4(out + (size_t)px * (Cl + Ch) + c, v);
}

// SPPF's three chained 5x5 / stride 1 / pad 2 max pools and the concat in one pass: out [N][H][W][4 C] = (x, pool5(x), pool5^2(x),
// pool5^3(x)) = x and its maxima over the 5x5, 9x9 and 13x13 windows (clipped at the border, as -inf padding does).  A thread makes
// one 16-byte chunk of the four slices of one pixel.  x may be a channel slice (pixel stride xs).
__global__ __launch_bounds__(256) void k32_sppf(const float* __restrict__ x, int xs, float* __restrict__ out, int N, int H, int W, int C)
{
    const int C4 = C >> 2;
    const long long tot = (long long)N * H * W * C4, i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= tot) return;
    const int c = 4 * (int)(i % C4);
    const long long px = i / C4;
    const int x0 = (int)(px % W), y0 = (int)((px / W) % H), img = (int)(px / ((long long)W * H));
    const float* xi = x + (size_t)img * H * W * xs + c;
    const f4 ninf = f4{ -INFINITY, -INFINITY, -INFINITY, -INFINITY };
    f4 m5 = ninf, m9 = ninf, m13 = ninf;
    for (int dy = -6; dy <= 6; ++dy) {
        const int yy = y0 + dy;
        if (yy < 0 || yy >= H) continue;
        const int ay = dy < 0 ? -dy : dy;
        for (int dx = -6; dx <= 6; ++dx) {
            const int xx = x0 + dx;
            if (xx < 0 || xx >= W) continue;
            const int ax = dx < 0 ? -dx : dx, r = ax > ay ? ax : ay;
            const f4 v = ld4(xi + (size_t)(yy * W + xx) * xs);
            m13 = max4(m13, v);
            if (r <= 4) m9 = max4(m9, v);
            if (r <= 2) m5 = max4(m5, v);
        }
    }
    float* o = out + (size_t)px * 4 * C + c;
    st4(o, ld4(xi + (size_t)(y0 * W + x0) * xs)); st4(o + C, m5); st4(o + 2 * C, m9); st4(o + 3 * C, m13);
}

// YOLOv8 anchor-free head decode in fp32, one pass (DFL softmax expectation, dist2bbox, stride scale, class sigmoid, level concat):
// the branch outputs of the three levels (NHWC float, bias included) -> pred [B][4 + nc + n_ext][A] float, the layout ss_nms reads.
// Thread = one anchor of one image.  ext: the third branch (keypoints: ext_mode 1, Ultralytics Pose.kpts_decode; mask coefficients:
// 0, raw).  The fp32 twin of ss_ops.hip's k_v8_decode, with expf instead of the fast exponential.
struct V8Levels32 {
    const float* box[3]; const float* cls[3]; const float* ext[3];   // [B][H][W][64], [B][H][W][cls_ld], [B][H][W][ext_ld]
    int H[3], W[3], stride[3];
    int n_ext, ext_ld, ext_mode, cls_ld;
};

__global__ __launch_bounds__(128) void k32_v8_decode(V8Levels32 L, int B, int nc, int A, float* __restrict__ pred)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (a >= A) return;
    int l = 0, a0 = 0;
    while (l < 2 && a >= a0 + L.H[l] * L.W[l]) { a0 += L.H[l] * L.W[l]; ++l; }
    const int p = a - a0, hw = L.H[l] * L.W[l];
    const int py = p / L.W[l], px = p - py * L.W[l];
    const float* bx = L.box[l] + ((size_t)b * hw + p) * 64;
    float d[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[16], m = -INFINITY;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4 t = ld4(bx + 16 * s + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[4 * q + k] = t[k]; m = fmaxf(m, t[k]); }
        }
        float se = 0.f, sw = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float e = expf(v[k] - m); se += e; sw += e * (float)k; }
        d[s] = sw / se;
    }
    const float ax = (float)px + 0.5f, ay = (float)py + 0.5f, st = (float)L.stride[l];
    const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
    float* o = pred + (size_t)b * (4 + nc + L.n_ext) * A + a;
    o[0] = (x1 + x2) * 0.5f * st; o[(size_t)A] = (y1 + y2) * 0.5f * st;
    o[(size_t)2 * A] = (x2 - x1) * st; o[(size_t)3 * A] = (y2 - y1) * st;
    if (L.n_ext) {
        const float* e = L.ext[l] + ((size_t)b * hw + p) * L.ext_ld;
        float* oe = o + (size_t)(4 + nc) * A;
        for (int k = 0, j = 0; k < L.n_ext; ++k, j = j == 2 ? 0 : j + 1) {
            const float v = e[k];
            oe[(size_t)k * A] = L.ext_mode == 0 ? v : j == 0 ? (v * 2.0f + (float)px) * st : j == 1 ? (v * 2.0f + (float)py) * st : 1.0f / (1.0f + expf(-v));
        }
    }
    const float* cl = L.cls[l] + ((size_t)b * hw + p) * L.cls_ld;
    for (int k = 0; k < nc; ++k) o[(size_t)(4 + k) * A] = 1.0f / (1.0f + expf(-cl[k]));
}

// ---- C ABI -----------------------------------------------------------------------------------------------------------------
#define OP32_CHECK() do { if (hipGetLastError() != hipSuccess) return SS_ERR_HIP; } while (0)
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies per DEVICE: `done` is a per-kernel bit set indexed by the current device
// (a process-wide bool left the second GPU of a process without the attribute: launches asking for > 64 KB of LDS failed there)
static bool lds_attr_once(const void* fn, unsigned long long& done)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return false;
    if (done >> dev & 1) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) return false;
    done |= 1ull << dev;
    return true;
}
static int g_chains_pre = -1;         // k32_chains3: x1 tiles of a chain's first 1x1 requested a phase ahead: -1 = where it was measured faster (16 channels:
                                      // 618 -> 572 us per launch; 24 channels: 218 -> 330 us, the 32 extra registers spill), 0 / 1 = A/B
static int g_chains_form = 2;        // 2: k32_chainsR (register row stream; 64 x 32 x 16 and 32 x 16 x 24 maps), 1: k32_chains3, 0: k32_chains (the 16 x 8 maps always take k32_chains)
static int g_conv_mt = 0, g_conv_min = 0;      // k32_conv: 16-channel output tiles per workgroup forced (A/B) / the workgroup count below which fewer are taken (0: never; measured: 768 -> 3.64 vs 3.59 ms)
static int g_conv_waves = 0, g_conv_wgs = 1024; // k32_conv: waves per workgroup (0 = 8 when that still leaves g_conv_wgs workgroups, else 4); A/B
static int g_tail_wgs = 0;           // k32_tail: workgroups of the persistent grid (0 = two per CU where they fit); A/B
static int g_chains_probe = 0;       // measurement only: bit 0 = k32_chainsR does not store the chain outputs (what the arithmetic alone costs)
static int g_stem_walk = 0;           // bands per workgroup of the walking stem: 0 = by batch size, -1 = never (k32_stem), 1 / 2 / 4 / 8 / 16 forced
static int g_chains_min_n = 128;     // batches below this take the LDS band forms even with chains_form 2: a row-stream wave walks a whole image (five layer passes,
                                      // ~200 us whatever the batch), the band forms spread an image over 4 workgroups x 12 waves — the per-frame call's 32-crop batches
static bool chains_rowstream(int N, int H, int W, int C)
{
    return g_chains_form == 2 && N >= g_chains_min_n && ((C == 16 && W == 32) || (C == 24 && W == 16) || (C == 32 && W == 8)) && H >= 9;
}

template <int K, int N>
static int launch_pw32(hipStream_t st, const float* x, const float* w, const float* b, const float* res, float* out, long long M, int relu,
                       const int* nv, int img_px)
{
    const long long tiles = (M + 15) / 16;
    int tpw = 1;
    while (tpw < 8 && tiles / (4 * tpw) > 8192) tpw *= 2;      // a workgroup re-stages the weights: amortise over more tiles on large maps
    const long long grid = (tiles + 4 * tpw - 1) / (4 * tpw);
    hipLaunchKernelGGL((k32_pw<K, N>), dim3((unsigned)grid), dim3(256), 0, st, x, w, b, res, out, M, relu, tpw, nv, img_px);
    OP32_CHECK();
    return SS_OK;
}

extern "C" int ss_op32_pointwise(void* stream, const void* d_x, const void* d_w, const void* d_bias, const void* d_res, void* d_out,
                                 long long M, int K, int N, int relu, const int* d_nvalid, int img_px)
{
    if (!d_x || !d_w || !d_bias || !d_out || M < 1 || (d_nvalid && img_px < 1)) return SS_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const float *x = (const float*)d_x, *w = (const float*)d_w, *b = (const float*)d_bias, *r = (const float*)d_res;
    float* o = (float*)d_out;
#define PW32(KK, NN) if (K == KK && N == NN) return launch_pw32<KK, NN>(st, x, w, b, r, o, M, relu, d_nvalid, img_px)
    PW32(16, 16); PW32(16, 64); PW32(64, 16); PW32(64, 24); PW32(64, 64); PW32(64, 96); PW32(96, 24); PW32(96, 32); PW32(96, 96);
    PW32(96, 128); PW32(128, 32); PW32(128, 128);
#undef PW32
    return SS_ERR_INVALID;
}

// bands of the chain launch for an H x W map with C mid channels: stage 1 (64 x 32, 16) 4 bands of 16 rows + 4 halo rows each side,
// the smaller maps one band = the whole image (no halo)
static int chains_rows(int H, int W, int C, int* halo)
{
    const size_t whole = 2ull * H * W * (C + 4) * 4 + (size_t)C32_THREADS * 16;
    if (whole <= 150 * 1024) { *halo = 0; return H; }
    *halo = 4;
    for (int R = H / 2; R >= 8; R /= 2)
        if (H % R == 0 && 2ull * (R + 8) * W * (C + 4) * 4 + (size_t)C32_THREADS * 16 <= 150 * 1024) return R;
    return -1;
}

extern "C" int ss_op32_chains_bands(int N, int H, int W, int C)
{
    if (N < 1 || H < 1 || W < 1 || !(C == 16 || C == 24 || C == 32)) return SS_ERR_INVALID;
    if (chains_rowstream(N, H, W, C)) return 1;
    int halo;
    const int R = chains_rows(H, W, C, &halo);
    return R < 1 ? SS_ERR_CAPACITY : H / R;
}

extern "C" int ss_op32_chains(void* stream, const void* d_x1, const void* d_w1, const void* d_w9, const void* d_bias, void* const* d_ys,
                              float* d_psum, int N, int H, int W, int C, const int* d_nvalid)
{
    if (!d_x1 || !d_w1 || !d_w9 || !d_bias || !d_ys || !d_psum || N < 1 || N > 65535 || (H * W) % 16) return SS_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const float *x1 = (const float*)d_x1, *w1 = (const float*)d_w1, *w9 = (const float*)d_w9, *b = (const float*)d_bias;
    float *y0 = (float*)d_ys[0], *y1 = (float*)d_ys[1], *y2 = (float*)d_ys[2], *y3 = (float*)d_ys[3];
    if (chains_rowstream(N, H, W, C)) {                       // two images (four waves) per workgroup
#define CHR(CC, WW, AL) if (C == CC && W == WW) { \
        const size_t ldsr = (size_t)(ChR<CC>::TAB + (AL ? 10 * ChR<CC>::NPR * ChR<CC>::KP : 0)) * 4; \
        static unsigned long long attr = 0; \
        if (ldsr > 64 * 1024 && !lds_attr_once((const void*)k32_chainsR<CC, WW, AL>, attr)) return SS_ERR_HIP; \
        const int ipw = WW == 8 ? 4 : 2;                      /* images per workgroup */ \
        hipLaunchKernelGGL((k32_chainsR<CC, WW, AL>), dim3((N + ipw - 1) / ipw), dim3(R32_THREADS), ldsr, st, x1, w1, w9, b, y0, y1, y2, y3, d_psum, N, H, d_nvalid, g_chains_probe); \
        OP32_CHECK(); return SS_OK; }
        CHR(16, 32, false) CHR(24, 16, true) CHR(32, 8, true)
#undef CHR
    }
    int halo;
    const int R = chains_rows(H, W, C, &halo);
    if (R < 1) return SS_ERR_CAPACITY;
    const size_t lds = 2ull * (R + 2 * halo) * W * (C + 4) * 4 + (size_t)(C32_THREADS / (C / 4)) * C * 4;
    const dim3 grid(H / R, N);
#define CH32(CC, WW) if (C == CC && W == WW) { \
        static unsigned long long attr = 0; \
        if (!lds_attr_once((const void*)k32_chains<CC, WW>, attr)) return SS_ERR_HIP; \
        hipLaunchKernelGGL((k32_chains<CC, WW>), grid, dim3(C32_THREADS), lds, st, x1, w1, w9, b, y0, y1, y2, y3, d_psum, N, H, R, halo, d_nvalid); \
        OP32_CHECK(); return SS_OK; }
#define CH32C(CC, WW, PRE_) if (C == CC && W == WW && g_chains_form >= 1 && (g_chains_pre < 0 ? CC == 16 : g_chains_pre != 0) == PRE_ && R + 2 * halo == (CC == 16 ? 24 : 32)) { \
        static unsigned long long attr = 0; \
        if (!lds_attr_once((const void*)k32_chains3<CC, WW, PRE_>, attr)) return SS_ERR_HIP; \
        const size_t lds3 = ((size_t)(R + 2 * halo + 2) * (WW + 2) + (size_t)(R + 2 * halo) * WW) * Ch3<CC>::PITCH * 4 + (size_t)(C32_THREADS / 64) * 16 * 4; \
        hipLaunchKernelGGL((k32_chains3<CC, WW, PRE_>), grid, dim3(C32_THREADS), lds3, st, x1, w1, w9, b, y0, y1, y2, y3, d_psum, N, H, R, halo, d_nvalid); \
        OP32_CHECK(); return SS_OK; }
    CH32C(16, 32, true) CH32C(16, 32, false) CH32C(24, 16, true) CH32C(24, 16, false)
#undef CH32C
    CH32(16, 32) CH32(24, 16) CH32(32, 8)
#undef CH32
    return SS_ERR_INVALID;
}

// A/B switch (tests, measurements): "chains_form" 2 = k32_chainsR where it applies (default), 1 = k32_chains3, 0 = k32_chains everywhere
extern "C" int ss_op32_set_option(const char* name, int value)
{
    if (!name) return SS_ERR_INVALID;
    if (!strcmp(name, "chains_pre")) { g_chains_pre = value < 0 ? -1 : (value != 0); return SS_OK; }
    if (!strcmp(name, "conv_mt")) { if (value < 0 || value > 5 || value == 3) return SS_ERR_INVALID; g_conv_mt = value; return SS_OK; }
    if (!strcmp(name, "conv_min")) { if (value < 0) return SS_ERR_INVALID; g_conv_min = value; return SS_OK; }
    if (!strcmp(name, "conv_waves")) { if (!(value == 0 || value == 4 || value == 8)) return SS_ERR_INVALID; g_conv_waves = value; return SS_OK; }
    if (!strcmp(name, "conv_wgs")) { if (value < 1) return SS_ERR_INVALID; g_conv_wgs = value; return SS_OK; }
    if (!strcmp(name, "tail_wgs")) { if (value < 0 || value > 65535) return SS_ERR_INVALID; g_tail_wgs = value; return SS_OK; }
    if (!strcmp(name, "chains_probe")) { g_chains_probe = value; return SS_OK; }
    if (!strcmp(name, "stem_walk")) { if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16) return SS_ERR_INVALID; g_stem_walk = value; return SS_OK; }
    if (!strcmp(name, "chains_min_n")) { if (value < 1) return SS_ERR_INVALID; g_chains_min_n = value; return SS_OK; }
    if (!strcmp(name, "chains_form")) { if (value < 0 || value > 2) return SS_ERR_INVALID; g_chains_form = value; return SS_OK; }
    return SS_ERR_INVALID;
}

template <int MID, int C2, int C1, int N2, bool POOL>
static int launch_tail32(hipStream_t st, const void* const* ys, const float* gates, const float* w3, const float* b3, const float* xin,
                         const float* wd, const float* bd, float* out, const float* w4, const float* b4, float* out2, int N, int H, int W, const int* nv)
{
    constexpr size_t lds = (size_t)(w_lds_floats<MID, C2>() + (C1 ? w_lds_floats<(C1 ? C1 : 16), C2>() : 0) + w_lds_floats<C2, N2>() + C2 + (N2 + 15) / 16 * 16 +
                                    (T32_THREADS / 64) * 4 * 16 * ((MID / 4 + 3) / 4)) * 4;
    static int wgs_dev[64] = { 0 };                            // workgroups of the persistent grid per device: what fits on the chip at once (<= 4 per CU)
    static unsigned long long attr = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63 || !lds_attr_once((const void*)k32_tail<MID, C2, C1, N2, POOL>, attr)) return SS_ERR_HIP;
    if (!wgs_dev[dev]) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k32_tail<MID, C2, C1, N2, POOL>, T32_THREADS, lds) != hipSuccess || cus < 1 || per_cu < 1)
            return SS_ERR_HIP;
        wgs_dev[dev] = cus * (per_cu > 4 ? 4 : per_cu);
    }
    const int wgs = wgs_dev[dev];
    const int tiles_wv = (H * W / 16 + T32_THREADS / 64 - 1) / (T32_THREADS / 64);      // tiles per wave when one workgroup takes a whole image
    int splits = 1;
    while (N * splits * 2 <= wgs && splits * 2 <= tiles_wv) splits *= 2;
    const int grid = g_tail_wgs > 0 ? g_tail_wgs : (N * splits < wgs ? N * splits : wgs);
    hipLaunchKernelGGL((k32_tail<MID, C2, C1, N2, POOL>), dim3(grid), dim3(T32_THREADS), lds, st, (const float*)ys[0], (const float*)ys[1], (const float*)ys[2],
                       (const float*)ys[3], gates, w3, b3, xin, wd, bd, out, w4, b4, out2, N, H, W, nv, splits);
    OP32_CHECK();
    return SS_OK;
}

extern "C" int ss_op32_tail(void* stream, const void* const* d_ys, const float* d_psum, int bands, const void* d_gw1, const void* d_gb1,
                            const void* d_gw2, const void* d_gb2, int hidden, float* d_gates, const void* d_w3, const void* d_b3, const void* d_xin,
                            int C1, const void* d_wd, const void* d_bd, void* d_out, const void* d_w4, const void* d_b4, void* d_out2, int pool, int N,
                            int H, int W, int MID, int C2, int N2, const int* d_nvalid)
{
    if (!d_ys || !d_psum || !d_gw1 || !d_gb1 || !d_gw2 || !d_gb2 || !d_gates || !d_w3 || !d_b3 || !d_xin || !d_w4 || !d_b4 || !d_out2 || N < 1 ||
        N > 65535 || bands < 1 || hidden < 1 || hidden > 4 || (H * W) % 16 || (C1 && (!d_wd || !d_bd)) || (pool && (W % 8 || H % 2)))
        return SS_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const float scale = 1.0f / (float)(H * W);
#define GT32(M_) if (MID == M_) hipLaunchKernelGGL((k32_gates<M_>), dim3(N), dim3(128), 0, st, d_psum, bands, scale, (const float*)d_gw1, (const float*)d_gb1, \
        (const float*)d_gw2, (const float*)d_gb2, hidden, d_gates, N, d_nvalid)
    GT32(16); else GT32(24); else GT32(32); else return SS_ERR_INVALID;
#undef GT32
    OP32_CHECK();
#define TL32(M_, C2_, C1_, N2_, P_) if (MID == M_ && C2 == C2_ && C1 == C1_ && N2 == N2_ && (pool != 0) == P_) \
        return launch_tail32<M_, C2_, C1_, N2_, P_>(st, d_ys, d_gates, (const float*)d_w3, (const float*)d_b3, (const float*)d_xin, (const float*)d_wd, \
            (const float*)d_bd, (float*)d_out, (const float*)d_w4, (const float*)d_b4, (float*)d_out2, N, H, W, d_nvalid)
    TL32(16, 64, 16, 16, false); TL32(16, 64, 0, 64, true); TL32(24, 96, 64, 24, false); TL32(24, 96, 0, 96, true);
    TL32(32, 128, 96, 32, false); TL32(32, 128, 0, 128, false);
#undef TL32
    return SS_ERR_INVALID;
}

static int stem_bands(int N) { return g_stem_walk > 0 ? g_stem_walk : (N >= 1024 ? 16 : N >= 512 ? 8 : N >= 128 ? 4 : 2); }   // bands a workgroup walks down (measured at 1 024 / 860 / 430 / 28 crops: tools/stem32_time.py)

// The stem on BYTE crops [N][256][128][3] (ss_crop_norm* with SS_DST_U8): the normalisation happens while the rows are staged.
// ... and with the first block's conv1 (1x1, 16 -> 16, bias, ReLU) applied to the pooled pixels in the same launch: d_y1 = relu(W1 d_y + b1), bit-equal
// to ss_op32_pointwise on d_y.  x_u8: byte crops (ss_op32_stem_u8) or float crops.
extern "C" int ss_op32_stem_conv1(void* stream, const void* d_x, int x_u8, const void* d_w, const void* d_bias, void* d_y, const void* d_w1, const void* d_b1,
                                  void* d_y1, int N, int H, int W, const int* d_nvalid)
{
    if (!d_x || !d_w || !d_bias || !d_y || !d_w1 || !d_b1 || !d_y1 || N < 1 || N > 65535 || H != 256 || W != 128) return SS_ERR_INVALID;
    constexpr size_t lds = (size_t)(24 * ST_ROWP + (2 * ST_PR + 1) * 64 * 16 + 768) * 4;
    static unsigned long long attr8 = 0, attr32 = 0;
    const int nb = stem_bands(N);
    if (x_u8) {
        if (!lds_attr_once((const void*)k32_stemW<true>, attr8)) return SS_ERR_HIP;
        hipLaunchKernelGGL(k32_stemW<true>, dim3(16 / nb, N), dim3(STW_THREADS), lds, (hipStream_t)stream, d_x, (const float*)d_w, (const float*)d_bias, (float*)d_y, N,
                           d_nvalid, nb, (const float*)d_w1, (const float*)d_b1, (float*)d_y1);
    } else {
        if (!lds_attr_once((const void*)k32_stemW<false>, attr32)) return SS_ERR_HIP;
        hipLaunchKernelGGL(k32_stemW<false>, dim3(16 / nb, N), dim3(STW_THREADS), lds, (hipStream_t)stream, d_x, (const float*)d_w, (const float*)d_bias, (float*)d_y, N,
                           d_nvalid, nb, (const float*)d_w1, (const float*)d_b1, (float*)d_y1);
    }
    OP32_CHECK();
    return SS_OK;
}

extern "C" int ss_op32_stem_u8(void* stream, const void* d_x, const void* d_w, const void* d_bias, void* d_y, int N, int H, int W, const int* d_nvalid)
{
    if (!d_x || !d_w || !d_bias || !d_y || N < 1 || N > 65535 || H != 256 || W != 128) return SS_ERR_INVALID;
    constexpr size_t lds = (size_t)(24 * ST_ROWP + (2 * ST_PR + 1) * 64 * 16 + 768) * 4;
    static unsigned long long attr = 0;
    const int nb = stem_bands(N);
    if (!lds_attr_once((const void*)k32_stemW<true>, attr)) return SS_ERR_HIP;
    hipLaunchKernelGGL(k32_stemW<true>, dim3(16 / nb, N), dim3(STW_THREADS), lds, (hipStream_t)stream, d_x, (const float*)d_w,
                       (const float*)d_bias, (float*)d_y, N, d_nvalid, nb, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
    OP32_CHECK();
    return SS_OK;
}

extern "C" int ss_op32_stem(void* stream, const void* d_x, const void* d_w, const void* d_bias, void* d_y, int N, int H, int W, const int* d_nvalid)
{
    if (!d_x || !d_w || !d_bias || !d_y || N < 1 || N > 65535 || H != 256 || W != 128) return SS_ERR_INVALID;
    constexpr size_t lds = (size_t)(24 * ST_ROWP + (2 * ST_PR + 1) * 64 * 16) * 4;
    static unsigned long long attr = 0, attr_w = 0;
    const int nb = g_stem_walk < 0 ? 0 : stem_bands(N);      // -1: one band per workgroup, k32_stem
    if (nb) {
        if (!lds_attr_once((const void*)k32_stemW<false>, attr_w)) return SS_ERR_HIP;
        hipLaunchKernelGGL(k32_stemW<false>, dim3(16 / nb, N), dim3(STW_THREADS), lds, (hipStream_t)stream, d_x, (const float*)d_w,
                           (const float*)d_bias, (float*)d_y, N, d_nvalid, nb, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
        OP32_CHECK();
        return SS_OK;
    }
    if (!lds_attr_once((const void*)k32_stem, attr)) return SS_ERR_HIP;
    hipLaunchKernelGGL(k32_stem, dim3(64 / ST_PR, N), dim3(C32_THREADS), lds, (hipStream_t)stream, (const float*)d_x, (const float*)d_w,
                       (const float*)d_bias, (float*)d_y, N, d_nvalid);
    OP32_CHECK();
    return SS_OK;
}

extern "C" int ss_op32_head(void* stream, const void* d_x, const void* d_w, const void* d_bias, void* d_out, int N, int HW, int C, int F,
                            const int* d_nvalid)
{
    if (!d_x || !d_w || !d_bias || !d_out || N < 1 || HW < 1 || C != 128 || F < 1) return SS_ERR_INVALID;
    hipLaunchKernelGGL(k32_head, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)d_x, (const float*)d_w, (const float*)d_bias,
                       (float*)d_out, N, HW, F, d_nvalid);
    OP32_CHECK();
    return SS_OK;
}

/* fp32 convolution of the detector graphs (k32_conv): d_x NHWC [N][H][W][.] with pixel stride xs floats (>= Cin, a channel slice of a
 * wider tensor when larger), d_w [Cout][ks][ks][Cin], d_out NHWC [N][OH][OW][.] with pixel stride os, d_res (may be NULL, added AFTER
 * the activation) with pixel stride rs; ks 1 | 3, stride 1 | 2, pad ks / 2; act 1 = SiLU, 0 = none; Cin, Cout multiples of 16 and every
 * base pointer / stride a multiple of 4 floats. */
extern "C" int ss_op32_conv(void* stream, const void* d_x, int xs, const void* d_w, const void* d_bias, const void* d_res, int rs, void* d_out,
                            int os, int N, int H, int W, int Cin, int Cout, int ks, int stride, int act)
{
    if (!d_x || !d_w || !d_bias || !d_out || N < 1 || H < 1 || W < 1 || Cin < 16 || Cin % 16 || Cout < 16 || Cout % 16 || xs < Cin || os < Cout ||
        (xs | os | rs) % 4 || (d_res && rs < Cout) || !(ks == 1 || ks == 3) || !(stride == 1 || stride == 2) || (ks == 1 && stride != 1) ||
        (((uintptr_t)d_x | (uintptr_t)d_out | (uintptr_t)d_res | (uintptr_t)d_w | (uintptr_t)d_bias) & 15))
        return SS_ERR_INVALID;
    const int OH = (H + 2 * (ks / 2) - ks) / stride + 1, OW = (W + 2 * (ks / 2) - ks) / stride + 1;
    const long long M = (long long)N * OH * OW, tiles = (M + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    // output channels per workgroup: 64 (80 for the 80-wide head layers) on the large maps; on the small ones fewer, so that the launch
    // still has ~g_conv_min workgroups (a 12 x 20 map of 32 frames is 480 pixel tiles: 120 workgroups of 64 channels would leave the chip 7/8 idle)
    int mt = Cout % 80 == 0 && Cout <= 80 ? 5 : (Cout % 64 == 0 ? 4 : (Cout % 32 == 0 ? 2 : 1));
    if (g_conv_mt) mt = (Cout % (16 * g_conv_mt) == 0) ? g_conv_mt : mt;
    else while (mt > 1 && (tiles / 4) * ((Cout + 16 * mt - 1) / (16 * mt)) < g_conv_min) mt = mt == 5 ? 1 : mt / 2;
    // enough workgroups for the chip: one pixel tile per wave on the small maps
    const int pt = 1;                                          // (two pixel tiles per wave measured slower everywhere: 4.25 vs 3.68 ms per 32 frames)
    const int wv = g_conv_waves ? g_conv_waves : (tiles / 8 * ((Cout + 16 * mt - 1) / (16 * mt)) >= g_conv_wgs ? 8 : 4);
    const dim3 grid((unsigned)((tiles + wv * pt - 1) / (wv * pt)), (unsigned)((Cout + 16 * mt - 1) / (16 * mt)));
#define CV32(KS_, ST_, MT_, WV_) if (ks == KS_ && stride == ST_ && mt == MT_ && wv == WV_) { \
        hipLaunchKernelGGL((k32_conv<KS_, ST_, MT_, 1, WV_>), grid, dim3(64 * WV_), 0, st, (const float*)d_x, xs, (const float*)d_w, (const float*)d_bias, \
                           (const float*)d_res, rs, (float*)d_out, os, N, H, W, OH, OW, Cin, Cout, act); \
        OP32_CHECK(); return SS_OK; }
#define CV32M(KS_, ST_) CV32(KS_, ST_, 1, 4) CV32(KS_, ST_, 1, 8) CV32(KS_, ST_, 2, 4) CV32(KS_, ST_, 2, 8) CV32(KS_, ST_, 4, 4) CV32(KS_, ST_, 4, 8) \
        CV32(KS_, ST_, 5, 4) CV32(KS_, ST_, 5, 8)
    CV32M(1, 1) CV32M(3, 1) CV32M(3, 2)
#undef CV32M
#undef CV32
    return SS_ERR_INVALID;
}

/* The first convolution of a detector in fp32 (k32_conv0): 3 -> 16 channels, 3x3, stride 2, pad 1; d_x dense NHWC [N][H][W][3],
 * d_w [16][3][3][3] (co, ky, kx, ci), d_out NHWC with pixel stride os. */
extern "C" int ss_op32_conv0(void* stream, const void* d_x, const void* d_w, const void* d_bias, void* d_out, int os, int N, int H, int W, int act)
{
    if (!d_x || !d_w || !d_bias || !d_out || N < 1 || H < 2 || W < 2 || os < 16 || os % 4 || ((uintptr_t)d_out & 15)) return SS_ERR_INVALID;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long long M = (long long)N * OH * OW;
    hipLaunchKernelGGL(k32_conv0, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)d_x, (const float*)d_w,
                       (const float*)d_bias, (float*)d_out, os, N, H, W, OH, OW, act);
    OP32_CHECK();
    return SS_OK;
}

/* cat(upsample2x_nearest(lo), hi) (lo_first != 0) or cat(hi, upsample2x_nearest(lo)) along channels, fp32 NHWC: d_lo [N][H/2][W/2][.]
 * pixel stride ls, d_hi [N][H][W][.] pixel stride hs, d_out dense [N][H][W][Cl + Ch]; Cl, Ch multiples of 4, H and W even. */
extern "C" int ss_op32_upcat(void* stream, const void* d_lo, int ls, int Cl, const void* d_hi, int hs, int Ch, void* d_out, int N, int H, int W, int lo_first)
{
    if (!d_lo || !d_hi || !d_out || N < 1 || H < 2 || W < 2 || (H | W) & 1 || Cl < 4 || Ch < 4 || (Cl | Ch | ls | hs) % 4 || ls < Cl || hs < Ch ||
        (((uintptr_t)d_lo | (uintptr_t)d_hi | (uintptr_t)d_out) & 15))
        return SS_ERR_INVALID;
    const long long tot = (long long)N * H * W * ((Cl + Ch) / 4);
    hipLaunchKernelGGL(k32_upcat, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)d_lo, ls, Cl, (const float*)d_hi, hs, Ch,
                       (float*)d_out, N, H, W, lo_first);
    OP32_CHECK();
    return SS_OK;
}

/* YOLOv8 head decode in fp32: the three levels' branch outputs (dense NHWC float, bias included: box [B][H][W][64], cls [B][H][W][cls_ld]
 * with the first nc channels used, ext [B][H][W][ext_ld] with the first n_ext used, ext_mode 1 = keypoint triplets, 0 = raw) ->
 * d_pred [B][4 + nc + n_ext][A], A = sum of H * W. */
extern "C" int ss_op32_v8_decode(void* stream, const void* const* d_box, const void* const* d_cls, const void* const* d_ext, int n_ext, int ext_ld,
                                 int ext_mode, const int* H, const int* W, const int* strides, int B, int nc, int cls_ld, float* d_pred)
{
    if (!d_box || !d_cls || !H || !W || !strides || !d_pred || B < 1 || B > 65535 || nc < 1 || cls_ld < nc || n_ext < 0 || (n_ext && (!d_ext || ext_ld < n_ext)))
        return SS_ERR_INVALID;
    V8Levels32 L;
    int A = 0;
    for (int l = 0; l < 3; ++l) {
        if (!d_box[l] || !d_cls[l] || (n_ext && !d_ext[l]) || H[l] < 1 || W[l] < 1 || ((uintptr_t)d_box[l] & 15)) return SS_ERR_INVALID;
        L.box[l] = (const float*)d_box[l]; L.cls[l] = (const float*)d_cls[l]; L.ext[l] = n_ext ? (const float*)d_ext[l] : nullptr;
        L.H[l] = H[l]; L.W[l] = W[l]; L.stride[l] = strides[l];
        A += H[l] * W[l];
    }
    L.n_ext = n_ext; L.ext_ld = ext_ld; L.ext_mode = ext_mode; L.cls_ld = cls_ld;
    hipLaunchKernelGGL(k32_v8_decode, dim3((A + 127) / 128, B), dim3(128), 0, (hipStream_t)stream, L, B, nc, A, d_pred);
    OP32_CHECK();
    return SS_OK;
}

/* SPPF's pools + concat in fp32: d_x NHWC [N][H][W][.] pixel stride xs -> d_out dense [N][H][W][4 C] = (x, pool5 x, pool5 pool5 x, ...). */
extern "C" int ss_op32_sppf_pools(void* stream, const void* d_x, int xs, void* d_out, int N, int H, int W, int C)
{
    if (!d_x || !d_out || N < 1 || H < 1 || W < 1 || C < 4 || (C | xs) % 4 || xs < C || (((uintptr_t)d_x | (uintptr_t)d_out) & 15)) return SS_ERR_INVALID;
    const long long tot = (long long)N * H * W * (C / 4);
    hipLaunchKernelGGL(k32_sppf, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)d_x, xs, (float*)d_out, N, H, W, C);
    OP32_CHECK();
    return SS_OK;
}
