// ss_ops.hip — small fused NHWC-f16 operators for the detector / OSNet glue (rows a2 / a5).
//
// The convolutions themselves stay in PyTorch-ROCm (MIOpen -> MFMA).  What rocprofv3 showed at batch 1
// (profiles/r01_rocprofv3_kernel_stats_c2_s1.csv) is ~1000 launches per frame, most of them 4-5 us
// elementwise pieces around the convs: separate bias adds (SubTensorOpWithScalar1d), activations,
// OSNet's depthwise 3x3s falling to MIOpen's naive_conv, and 8 launches per channel gate.  These
// stateless kernels fuse those pieces (CDNA rule: fuse elementwise work into as few passes as possible).
// All tensors are channels-last half: element (n,h,w,c) at ((n*H+h)*W+w)*C+c.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include "../../include/strongsort_hip.h"

// Valid-image count of the OSNet kernels (ss_op_set_valid_images): the ReID crops of a frame group are packed (frame f's
// crops follow frame f-1's), the launches keep their fixed grids (graph replay) and the workgroups of images >= *n leave at
// once.  The setting belongs to ONE HIP stream of ONE host thread (two pipelines in a process use different streams; a
// stream that never set it computes every image): a small thread-local table keyed by the stream handle.
struct NvEntry { void* stream; const int* n; int batch; };
static thread_local NvEntry g_nv[8] = {};
static inline const int* nv_for(void* stream, long long batch)
{
    for (const NvEntry& e : g_nv) if (e.n && e.stream == stream && e.batch == batch) return e.n;
    return nullptr;
}
static inline int nv_batch(void* stream)
{
    for (const NvEntry& e : g_nv) if (e.n && e.stream == stream) return e.batch;
    return 0;
}

// Process-wide A/B switches of the stateless operators (ss_op_set_option; all default to 1)
static int g_opt_pw_epilogue = 1, g_opt_pw_splitk = 1, g_opt_osnet_chains = 1;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ inline float act_apply(float v, int act)
{
    if (act == 1) return v > 0.f ? v : 0.f;                 // relu
    // v_rcp_f32 (1 ulp) instead of the IEEE division this library is otherwise compiled with (-fhip-fp32-correctly-rounded-divide-sqrt:
    // a ~20-instruction sequence per element; eight of them per output vector made the epilogue of a 256-pixel convolution
    // tile 8 us long, in-kernel stamps r03).  The value is rounded to half right after; no tracker arithmetic goes through here.
    if (act == 2) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));          // silu
    if (act == 3) return __builtin_amdgcn_rcpf(1.0f + __expf(-v));              // sigmoid
    return v;
}

// x = act(x + bias[c] (+ res)), in place.  C % 8 == 0: 16-byte vectors.
__global__ __launch_bounds__(256) void k_bias_act8(__half* __restrict__ x, const __half* __restrict__ bias,
                                                  const __half* __restrict__ res, size_t n_vec, int C8, int act)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
        h8 v = reinterpret_cast<h8*>(x)[i];
        h8 b = reinterpret_cast<const h8*>(bias)[i % C8];
        h8 r;
        if (res) r = reinterpret_cast<const h8*>(res)[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float f = (float)v[k] + (float)b[k];
            if (res) f += (float)r[k];
            v[k] = (_Float16)act_apply(f, act);
        }
        reinterpret_cast<h8*>(x)[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_bias_act1(__half* __restrict__ x, const __half* __restrict__ bias,
                                                  const __half* __restrict__ res, size_t n, int C, int act)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float f = __half2float(x[i]) + __half2float(bias[i % C]);
        if (res) f += __half2float(res[i]);
        x[i] = __float2half(act_apply(f, act));
    }
}

// One depthwise tap for 8 channels: acc[k] = fma(float(v[k]), float(w[k]), acc[k]) as v_fma_mix_f32 (f16 sources widened
// inside the instruction, f32 accumulate, one rounding — bit-identical to cvt + v_fma_f32).  hipcc otherwise widens the
// operands with 8 + 8 v_cvt_f32_f16 per tap and keeps the nine tap weights as 72 float registers; measured on
// k_osnet_streams<16>: 245 VALU instructions per (pixel, 8 channels) and 160 VGPRs before (profiles/r02_pmc_nets.json).
__device__ __forceinline__ float ss_mix_lo(unsigned a, unsigned b, float c)
{ float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float ss_mix_hi(unsigned a, unsigned b, float c)
{ float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ void ss_dw_tap8(const h8& v, const h8& w, float (&acc)[8])
{
    const uint4 vu = __builtin_bit_cast(uint4, v), wu = __builtin_bit_cast(uint4, w);
    acc[0] = ss_mix_lo(vu.x, wu.x, acc[0]); acc[1] = ss_mix_hi(vu.x, wu.x, acc[1]);
    acc[2] = ss_mix_lo(vu.y, wu.y, acc[2]); acc[3] = ss_mix_hi(vu.y, wu.y, acc[3]);
    acc[4] = ss_mix_lo(vu.z, wu.z, acc[4]); acc[5] = ss_mix_hi(vu.z, wu.z, acc[5]);
    acc[6] = ss_mix_lo(vu.w, wu.w, acc[6]); acc[7] = ss_mix_hi(vu.w, wu.w, acc[7]);
}

// depthwise 3x3, stride 1, pad 1, + bias + relu.  w9 is [9][C] (tap-major).  thread = (pixel, 8 channels).
__global__ __launch_bounds__(256) void k_dw3x3(const __half* __restrict__ x, const __half* __restrict__ w9,
                                              const __half* __restrict__ bias, __half* __restrict__ y, int N, int H,
                                              int W, int C8, int act)
{
    const size_t total = (size_t)N * H * W * C8;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const size_t p = i / C8;
        const int w = (int)(p % W), h = (int)((p / W) % H);
        const size_t nb = (p / ((size_t)W * H)) * H;
        float acc[8];
        {
            h8 b = reinterpret_cast<const h8*>(bias)[c8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = (float)b[k];
        }
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int hh = h + dy;
            if (hh < 0 || hh >= H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int ww = w + dx;
                if (ww < 0 || ww >= W) continue;
                h8 v = reinterpret_cast<const h8*>(x)[((nb + hh) * W + ww) * C8 + c8];
                h8 k9 = reinterpret_cast<const h8*>(w9)[((dy + 1) * 3 + dx + 1) * C8 + c8];
                ss_dw_tap8(v, k9, acc);
            }
        }
        h8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (_Float16)act_apply(acc[k], act);
        reinterpret_cast<h8*>(y)[i] = o;
    }
}

// Epilogue with placement: out[pix][0..C) = act(x + bias) (+ res after the activation when res_after), where `out`
// is a channel slice of a wider NHWC tensor (row pitch out_ld elements) — a C2f block's concat buffer — and
// channels [c0, c0+cn) are optionally mirrored into a second, dense tensor (the half the next bottleneck
// convolves).  Replaces chunk().contiguous(), the shortcut add and torch.cat's per-input copies.
__global__ __launch_bounds__(256) void k_bias_act_place(const __half* __restrict__ x, const __half* __restrict__ bias,
                                                       const __half* __restrict__ res, size_t n_vec, int C8, int act,
                                                       int res_after, __half* __restrict__ out, int out_ld8,
                                                       __half* __restrict__ out2, int c0_8, int cn_8)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / C8;
        const int c8 = (int)(i - pix * C8);
        h8 v = reinterpret_cast<const h8*>(x)[i];
        const h8 b = reinterpret_cast<const h8*>(bias)[c8];
        h8 r;
        if (res) r = reinterpret_cast<const h8*>(res)[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float f = (float)v[k] + (float)b[k];
            if (res && !res_after) f += (float)r[k];
            f = act_apply(f, act);
            if (res && res_after) f = (float)(_Float16)f + (float)r[k];      // the unfused form rounds the activation first
            v[k] = (_Float16)f;
        }
        reinterpret_cast<h8*>(out)[pix * out_ld8 + c8] = v;
        if (out2 && c8 >= c0_8 && c8 < c0_8 + cn_8) reinterpret_cast<h8*>(out2)[pix * cn_8 + (c8 - c0_8)] = v;
    }
}

// YOLOv8 anchor-free head decode in one pass (DFL softmax expectation, dist2bbox, stride scale, class sigmoid,
// level concat): reads the six branch outputs of the three levels (NHWC half, final 1x1 conv WITHOUT its bias —
// added here) and writes pred [B][4+nc][A] float, the layout ss_nms reads.  thread = one anchor of one image.
struct V8Levels {
    const __half* box[3]; const __half* cls[3];             // [B][H][W][64], [B][H][W][nc]
    const __half* box_bias[3]; const __half* cls_bias[3];
    int H[3], W[3], stride[3];
    // the head's third branch (keypoints / mask coefficients), bias already added: [B][H][W][ext_ld], n_ext of them used ->
    // rows 4 + nc .. of the prediction; ext_mode 1: Ultralytics Pose.kpts_decode ((2 v + cell) * stride for x and y, sigmoid for
    // the visibility of every (x, y, v) triplet), 0: raw (Segment's mask coefficients)
    const __half* ext[3];
    int n_ext, ext_ld, ext_mode;
    int cls_ld;                                             // channels per pixel of the class tensors (>= nc: a 1-class head padded to 8)
};

__global__ __launch_bounds__(128) void k_v8_decode(V8Levels L, int B, int nc, int A, float* __restrict__ pred)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (a >= A) return;
    int l = 0, a0 = 0;
    while (l < 2 && a >= a0 + L.H[l] * L.W[l]) { a0 += L.H[l] * L.W[l]; ++l; }
    const int p = a - a0, hw = L.H[l] * L.W[l];
    const int py = p / L.W[l], px = p - py * L.W[l];
    const h8* bx = reinterpret_cast<const h8*>(L.box[l] + ((size_t)b * hw + p) * 64);
    const h8* bb = reinterpret_cast<const h8*>(L.box_bias[l]);
    float d[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[16], m = -INFINITY;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const h8 t = bx[s * 2 + h], tb = bb[s * 2 + h];
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[h * 8 + k] = (float)t[k] + (float)tb[k]; m = fmaxf(m, v[h * 8 + k]); }
        }
        float se = 0.f, sw = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float e = __expf(v[k] - m); se += e; sw += e * (float)k; }
        d[s] = sw / se;
    }
    const float ax = (float)px + 0.5f, ay = (float)py + 0.5f, st = (float)L.stride[l];
    const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
    float* o = pred + (size_t)b * (4 + nc + L.n_ext) * A + a;
    o[0] = (x1 + x2) * 0.5f * st; o[(size_t)A] = (y1 + y2) * 0.5f * st;
    o[(size_t)2 * A] = (x2 - x1) * st; o[(size_t)3 * A] = (y2 - y1) * st;
    if (L.n_ext) {
        const __half* e = L.ext[l] + ((size_t)b * hw + p) * L.ext_ld;
        float* oe = o + (size_t)(4 + nc) * A;
        for (int k = 0, j = 0; k < L.n_ext; ++k, j = j == 2 ? 0 : j + 1) {
            const float v = __half2float(e[k]);
            oe[(size_t)k * A] = L.ext_mode == 0 ? v : j == 0 ? (v * 2.0f + (float)px) * st : j == 1 ? (v * 2.0f + (float)py) * st : 1.0f / (1.0f + __expf(-v));
        }
    }
    const __half* cl = L.cls[l] + ((size_t)b * hw + p) * L.cls_ld;
    const __half* cb = L.cls_bias[l];
    if (nc % 8 == 0) {
        for (int k8 = 0; k8 < nc / 8; ++k8) {
            const h8 t = reinterpret_cast<const h8*>(cl)[k8], tb = reinterpret_cast<const h8*>(cb)[k8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[(size_t)(4 + k8 * 8 + k) * A] = 1.0f / (1.0f + __expf(-((float)t[k] + (float)tb[k])));
        }
        return;
    }
    for (int k = 0; k < nc; ++k) {
        const float z = __half2float(cl[k]) + __half2float(cb[k]);
        o[(size_t)(4 + k) * A] = 1.0f / (1.0f + __expf(-z));
    }
}

// Pointwise (1x1) convolution + bias + activation (+ shortcut) with placement, on the matrix cores:
//   out[pix][n] = act(sum_k x[pix][k] * w[n][k] + bias[n])        x NHWC half [M][K], w [N][K] half
// MIOpen runs these as zero-fill + implicit GEMM + our bias/activation pass (3 launches, ~20 us at the
// detector's sizes); hipBLASLt needs a separate activation pass and is 2-4x off the HBM roofline for K,N <= 64.
// Workgroup = 4 waves x PT pixel tiles of 16 = BM pixels x BN output channels.  Output channel = MFMA M
// (weights = A operand, staged per 64-wide K chunk in LDS), pixel = MFMA N (B operand straight from global).
// v_mfma_f32_16x16x16_f16 sums over k whatever the k <-> (lane group, element) assignment is, as long as A and
// B use the same one: lane group q takes k = 8q..8q+7 of a 32-wide chunk, elements 0-3 for the first MFMA and
// 4-7 for the second, so both operands are single 16-byte loads.  K % 8 == 0, N % 8 == 0.
//
// CONV3: the same kernel as an implicit GEMM for 3x3 / pad 1 / stride 1|2 convolutions: K = 9*Cin with k = tap*Cin + c
// (weights [N][3][3][Cin]), and a lane's eight k's — always inside one tap because Cin % 8 == 0 — come from the
// input pixel shifted by that tap (zeros outside the image).  Replaces MIOpen's zero-fill + igemm + our epilogue pass.
struct ConvGeom { int H, W, Cin, OH, OW, stride; };

// VEC_EPI: the accumulators (4 channels x 1 pixel per lane and tile) are transposed through a per-wave LDS tile so the
// epilogue reads the shortcut and writes the output as 16-byte vectors, 128 contiguous bytes per 8 lanes, instead
// of 8-byte pieces of 32-byte segments.
struct PwArgs {
    const __half* x; const __half* w; const __half* bias; const __half* res;
    int M, K, N, act, res_after;
    __half* out; int out_ld; __half* out2; int c0, cn;
    ConvGeom g;
    const int* n_img; int img_px;                  // optional: only the first *n_img images (img_px pixels each) are computed
};

// LDS of pw_body<BN, PT, ., VEC_EPI> in halfs: the weight slice, then the four waves' epilogue tiles.  The caller owns the array
// (a kernel that inlines several instantiations - k_pw_group - would otherwise get the SUM of their static arrays).
template <int BN, int PT, bool VEC_EPI>
constexpr int pw_lds_halfs() { return BN * 72 + (VEC_EPI ? 4 * PT * 16 * (BN + 8) : 0); }

template <int BN, int PT, bool CONV3, bool VEC_EPI>
__device__ __forceinline__ void pw_body(const PwArgs& A, const int bx, const int by, _Float16* __restrict__ pw_lds)
{
    const __half* __restrict__ x = A.x; const __half* __restrict__ w = A.w; const __half* __restrict__ bias = A.bias;
    const __half* __restrict__ res = A.res; __half* __restrict__ out = A.out; __half* __restrict__ out2 = A.out2;
    const int K = A.K, N = A.N, act = A.act, res_after = A.res_after, out_ld = A.out_ld, c0 = A.c0, cn = A.cn;
    int M = A.M;
    if (A.n_img) { const long long mv = (long long)(*A.n_img) * A.img_px; if (mv < M) M = (int)mv; }
    const ConvGeom g = A.g;
    constexpr int MT = BN / 16, KC = 64, PITCH = KC + 8, BM = 64 * PT;
    if ((long long)bx * BM >= M) return;                    // (only with n_img: the launch covers the full batch)
    _Float16* __restrict__ Ws = pw_lds;                     // [BN][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, n = lane & 15;
    const int n0 = by * BN;
    const size_t px0 = (size_t)bx * BM + wave * (16 * PT);
    f4 acc[MT][PT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[mt][pt] = f4{ 0.f, 0.f, 0.f, 0.f };
    const h8 z8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    // CONV3: image base and top-left input coordinate of this lane's output pixel(s)
    size_t ibase[PT];
    int iy0[PT], ix0[PT];
    if (CONV3) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const size_t px = px0 + pt * 16 + n;
            const int ohw = g.OH * g.OW;
            const int bimg = (int)(px / ohw), rem = (int)(px - (size_t)bimg * ohw), oy = rem / g.OW, ox = rem - oy * g.OW;
            ibase[pt] = (size_t)bimg * g.H * g.W;
            iy0[pt] = oy * g.stride - 1; ix0[pt] = ox * g.stride - 1;
        }
    }

    // software pipeline over 64-wide K chunks: chunk kc+64's weights (to registers) and pixels are requested before
    // chunk kc's MFMAs, so global latency hides behind the matrix work of the previous chunk
    constexpr int WV = (BN * (KC / 8) + 255) / 256;         // weight vectors staged per thread and chunk
    auto load_w = [&](int kc, h8 (&wr)[WV]) {
#pragma unroll
        for (int j = 0; j < WV; ++j) {
            const int i = tid + j * 256, r = i / (KC / 8), c8 = i - r * (KC / 8);
            const int oc = n0 + r, k = kc + c8 * 8;
            wr[j] = (i < BN * (KC / 8) && oc < N && k < K) ? *reinterpret_cast<const h8*>(w + (size_t)oc * K + k) : z8;
        }
    };
    auto load_b = [&](int kc, h8 (&b)[KC / 32][PT]) {
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const size_t px = px0 + pt * 16 + n;
                const int k = kc + ks * 32 + 8 * q;
                if (!CONV3) {
                    b[ks][pt] = (px < (size_t)M && k < K) ? *reinterpret_cast<const h8*>(x + px * K + k) : z8;
                } else {
                    const int tap = k / g.Cin, c = k - tap * g.Cin, dy = tap / 3, dx = tap - dy * 3;
                    const int iy = iy0[pt] + dy, ix = ix0[pt] + dx;
                    const bool ok = px < (size_t)M && k < K && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                    b[ks][pt] = ok ? *reinterpret_cast<const h8*>(x + (ibase[pt] + (size_t)iy * g.W + ix) * g.Cin + c) : z8;
                }
            }
    };
    // (weights two chunks ahead of their LDS store, pixels one chunk ahead of their MFMAs: a chunk's matrix work is shorter than
    //  the L2 latency of the next chunk's weights, r04)
    h8 wr[WV], wn[WV], b[KC / 32][PT], bn[KC / 32][PT];
    load_w(0, wr);
    load_b(0, b);
    for (int kc = 0; kc < K; kc += KC) {
        __syncthreads();                                    // previous chunk's LDS reads are done
#pragma unroll
        for (int j = 0; j < WV; ++j) {
            const int i = tid + j * 256, r = i / (KC / 8), c8 = i - r * (KC / 8);
            if (i < BN * (KC / 8)) *reinterpret_cast<h8*>(Ws + r * PITCH + c8 * 8) = wr[j];
        }
        if (kc + KC < K) { load_w(kc + KC, wn); load_b(kc + KC, bn); }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks) {
            if (kc + ks * 32 >= K) break;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const h8 a = *reinterpret_cast<const h8*>(Ws + (mt * 16 + n) * PITCH + ks * 32 + 8 * q);
                const h4 a0 = { a[0], a[1], a[2], a[3] }, a1 = { a[4], a[5], a[6], a[7] };
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const h8 bv = b[ks][pt];
                    if (CONV3) {                               // 3x3 layers (detector only): the double-K form, half the matrix-core issue cycles
                        acc[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bv, acc[mt][pt], 0, 0, 0);
                    } else {                                   // 1x1 layers keep the two 16x16x16 steps: OSNet's fused kernels reproduce their bits
                        const h4 b0 = { bv[0], bv[1], bv[2], bv[3] }, b1 = { bv[4], bv[5], bv[6], bv[7] };
                        acc[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, acc[mt][pt], 0, 0, 0);
                        acc[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, b1, acc[mt][pt], 0, 0, 0);
                    }
                }
            }
        }
        if (kc + KC < K) {
#pragma unroll
            for (int j = 0; j < WV; ++j) wr[j] = wn[j];
#pragma unroll
            for (int ks = 0; ks < KC / 32; ++ks)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) b[ks][pt] = bn[ks][pt];
        }
    }

    if (VEC_EPI) {
        constexpr int EP = BN + 8;                                           // tile pitch (halfs)
        _Float16* tile = pw_lds + BN * PITCH + wave * (PT * 16 * EP);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const h4 o = { (_Float16)acc[mt][pt][0], (_Float16)acc[mt][pt][1], (_Float16)acc[mt][pt][2], (_Float16)acc[mt][pt][3] };
                *reinterpret_cast<h4*>(tile + (pt * 16 + n) * EP + mt * 16 + 4 * q) = o;     // conv rounded to half, as unfused
            }
        __builtin_amdgcn_wave_barrier();
        constexpr int CG = BN / 8;                                           // 8-channel groups per row
#pragma unroll
        for (int it = 0; it < (PT * 16 * CG + 63) / 64; ++it) {
            const int i = it * 64 + lane, row = i / CG, cg = i - row * CG;
            const size_t px = px0 + row;
            const int oc = n0 + cg * 8;
            if (i >= PT * 16 * CG || px >= (size_t)M || oc >= N) continue;       // (16 x 10 vectors of an 80-channel tile: 2.5 rounds)
            const h8 v = *reinterpret_cast<const h8*>(tile + row * EP + cg * 8);
            const h8 bb = *reinterpret_cast<const h8*>(bias + oc);
            h8 r = z8;
            if (res) r = *reinterpret_cast<const h8*>(res + px * N + oc);
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = (float)v[j] + (float)bb[j];
                if (res && !res_after) f += (float)r[j];
                f = act_apply(f, act);
                if (res && res_after) f = (float)(_Float16)f + (float)r[j];
                o[j] = (_Float16)f;
            }
            *reinterpret_cast<h8*>(out + px * out_ld + oc) = o;
            if (out2 && oc >= c0 && oc < c0 + cn) *reinterpret_cast<h8*>(out2 + px * cn + (oc - c0)) = o;
        }
        return;
    }
    // epilogue: lane (q, n) holds channels oc0..oc0+3 of pixel n of each tile
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const size_t px = px0 + pt * 16 + n;
        if (px >= (size_t)M) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int oc0 = n0 + mt * 16 + 4 * q;
            if (oc0 >= N) continue;
            const h4 bb = *reinterpret_cast<const h4*>(bias + oc0);
            h4 r = { 0, 0, 0, 0 };
            if (res) r = *reinterpret_cast<const h4*>(res + px * N + oc0);
            h4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // the unfused form rounds the convolution to half before the bias pass
                float f = (float)(_Float16)acc[mt][pt][j] + (float)bb[j];
                if (res && !res_after) f += (float)r[j];
                f = act_apply(f, act);
                if (res && res_after) f = (float)(_Float16)f + (float)r[j];
                o[j] = (_Float16)f;
            }
            *reinterpret_cast<h4*>(out + px * out_ld + oc0) = o;
            if (out2 && oc0 >= c0 && oc0 < c0 + cn) *reinterpret_cast<h4*>(out2 + px * cn + (oc0 - c0)) = o;
        }
    }
}

#ifndef SS_TAIL_BUMP
#define SS_TAIL_BUMP 1
#endif
#ifndef SS_PW_BUMP
#define SS_PW_BUMP 1
#endif
// waves per SIMD asked of the register allocator (it otherwise parks the accumulators in AGPRs and stops a few registers above a step)
template <int BN, int PT, bool CONV3> constexpr int pw_waves() { return !SS_PW_BUMP ? 1 : BN * PT <= 32 ? 8 : (BN == 32 && PT == 2 && CONV3) ? 5 : BN * PT <= 80 ? 6 : BN * PT <= 128 ? 4 : BN * PT <= 160 ? 3 : 2; }   // (<32, 2, 3x3> at 6: spills)
template <int BN, int PT, bool CONV3, bool VEC_EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(pw_waves<BN, PT, CONV3>()))) void k_pw(PwArgs A)
{
    __shared__ __attribute__((aligned(16))) _Float16 pw_lds[pw_lds_halfs<BN, PT, VEC_EPI>()];
    pw_body<BN, PT, CONV3, VEC_EPI>(A, blockIdx.x, blockIdx.y, pw_lds);
}

// Split-K form of k_pw for layers whose pixel count cannot fill the chip with 64-pixel workgroups (the detector's
// stride-32 level at batch 16: 60-120 workgroups for 256 CUs, each walking 18-36 K chunks with two barriers per chunk:
// 17-31 us per layer, latency-bound).  Workgroup = 16 pixels x BN channels; the four waves take the 64-wide K
// chunks round-robin (wave w: chunks w, w+4, ..), both operands straight from global memory as 16-byte vectors in the
// v_mfma_f32_16x16x32_f16 layout (lane (q, n): k = 8q..8q+7 of a 32-wide block; A row = output channel n, B column =
// pixel n), next chunk's loads in flight during this chunk's MFMAs, no LDS and no barrier inside the loop; the four
// partial tiles are added in wave order through LDS and go through the same vector epilogue (conv rounded to half, bias,
// shortcut, activation).  4x the workgroups, 1/4 of the serial K walk per wave; weights are re-read from L2 per 16
// pixels instead of per 64-128, which is why large layers stay on k_pw.
template <int BN, bool CONV3>
__device__ __forceinline__ void pw_splitk_body(const PwArgs& A, const int bx, const int by)
{
    const __half* __restrict__ x = A.x; const __half* __restrict__ w = A.w; const __half* __restrict__ bias = A.bias;
    const __half* __restrict__ res = A.res; __half* __restrict__ out = A.out; __half* __restrict__ out2 = A.out2;
    const int K = A.K, N = A.N, act = A.act, res_after = A.res_after, out_ld = A.out_ld, c0 = A.c0, cn = A.cn;
    int M = A.M;
    if (A.n_img) { const long long mv = (long long)(*A.n_img) * A.img_px; if (mv < M) M = (int)mv; }
    const ConvGeom g = A.g;
    constexpr int MT = BN / 16, KC = 64, RP = 17;
    __shared__ float Red[4 * BN * RP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, n = lane & 15;
    const int n0 = by * BN;
    const size_t px = (size_t)bx * 16 + n;
    const h8 z8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    size_t ibase = 0;
    int iy0 = 0, ix0 = 0;
    if (CONV3) {
        const int ohw = g.OH * g.OW;
        const int bimg = (int)(px / ohw), rem = (int)(px - (size_t)bimg * ohw), oy = rem / g.OW, ox = rem - oy * g.OW;
        ibase = (size_t)bimg * g.H * g.W;
        iy0 = oy * g.stride - 1; ix0 = ox * g.stride - 1;
    }
    auto load_a = [&](int kc, h8 (&a)[MT][2]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int oc = n0 + mt * 16 + n, k = kc + ks * 32 + 8 * q;
                a[mt][ks] = (oc < N && k < K) ? *reinterpret_cast<const h8*>(w + (size_t)oc * K + k) : z8;
            }
    };
    auto load_b = [&](int kc, h8 (&b)[2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k = kc + ks * 32 + 8 * q;
            if (!CONV3) {
                b[ks] = (px < (size_t)M && k < K) ? *reinterpret_cast<const h8*>(x + px * K + k) : z8;
            } else {
                const int tap = k / g.Cin, c = k - tap * g.Cin, dy = tap / 3, dx = tap - dy * 3;
                const int iy = iy0 + dy, ix = ix0 + dx;
                const bool ok = px < (size_t)M && k < K && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                b[ks] = ok ? *reinterpret_cast<const h8*>(x + (ibase + (size_t)iy * g.W + ix) * g.Cin + c) : z8;
            }
        }
    };
    f4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f4{ 0.f, 0.f, 0.f, 0.f };
    h8 a[MT][2], b[2], an[MT][2], bn[2];
    int kc = wave * KC;
    if (kc < K) { load_a(kc, a); load_b(kc, b); }
    for (; kc < K; kc += 4 * KC) {
        const bool more = kc + 4 * KC < K;
        if (more) { load_a(kc + 4 * KC, an); load_b(kc + 4 * KC, bn); }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (kc + ks * 32 >= K) break;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt][ks], b[ks], acc[mt], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { a[mt][0] = an[mt][0]; a[mt][1] = an[mt][1]; }
            b[0] = bn[0]; b[1] = bn[1];
        }
    }
    // partial tiles -> LDS [wave][channel][pixel]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) Red[(wave * BN + mt * 16 + 4 * q + j) * RP + n] = acc[mt][j];
    __syncthreads();
    constexpr int CG = BN / 8;
    if (tid < 16 * CG) {
        const int row = tid / CG, cg = tid - row * CG;
        const size_t p = (size_t)bx * 16 + row;
        const int oc = n0 + cg * 8;
        if (p < (size_t)M && oc < N) {
            const h8 bb = *reinterpret_cast<const h8*>(bias + oc);
            h8 r = z8;
            if (res) r = *reinterpret_cast<const h8*>(res + p * N + oc);
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = cg * 8 + j;
                const float sum = ((Red[(0 * BN + ch) * RP + row] + Red[(1 * BN + ch) * RP + row]) + Red[(2 * BN + ch) * RP + row]) +
                                  Red[(3 * BN + ch) * RP + row];
                float f = (float)(_Float16)sum + (float)bb[j];                // conv rounded to half, then the bias pass
                if (res && !res_after) f += (float)r[j];
                f = act_apply(f, act);
                if (res && res_after) f = (float)(_Float16)f + (float)r[j];
                o[j] = (_Float16)f;
            }
            *reinterpret_cast<h8*>(out + p * out_ld + oc) = o;
            if (out2 && oc >= c0 && oc < c0 + cn) *reinterpret_cast<h8*>(out2 + p * cn + (oc - c0)) = o;
        }
    }
}

template <int BN, bool CONV3>
__global__ __launch_bounds__(256) void k_pw_splitk(PwArgs A) { pw_splitk_body<BN, CONV3>(A, blockIdx.x, blockIdx.y); }

// Several independent convolutions of one shape class (all 3x3 or all 1x1, N <= BN) in ONE launch: the detect head's six
// branches are 18 launches of 60-1920 workgroups each, run one after the other; grouped by depth they are 3 launches whose
// small levels fill the CUs the stride-8 level leaves idle.  blockIdx.x walks the problems' workgroup ranges; a problem is
// either in k_pw's form (64 pixels per workgroup) or in the split-K form (16 pixels, long K walk, few pixels).
#define PW_GROUP_MAX 12
// form[p]: 0 = 64 pixels per workgroup, 1 = split-K.  A problem with N <= 64 in a BN = 80 group (the box branches next to the class
// branches) runs 4, not 5, channel tiles.  (128-pixel workgroups for the stride-8 level were measured too: the 148 VGPRs of that form
// set the occupancy of EVERY problem in the launch to 2 waves per SIMD - detector 1.03 -> 1.05 ms.)
struct PwGroup { PwArgs p[PW_GROUP_MAX]; int start[PW_GROUP_MAX + 1]; int form[PW_GROUP_MAX]; int n; };

// SPLITK = false: no problem of the launch takes the split-K form.  That body's 148 VGPRs and 21 KB of LDS set the occupancy of the
// WHOLE launch (3 waves per SIMD), and at 32 frames no level is small enough to take it: without it the 80-wide 3x3 group is 82 VGPRs /
// 22 KB (5 waves) - yolov8n on 32 frames 0.954 -> 0.914 ms, same bits.  (Also measured, r05: a level's two first layers, which read the
// same pixels, as ONE 144-channel problem whose halves the second layers read in place - half the gathers, 1.8x the matrix work per K
// chunk, but 118 VGPRs / 40 KB: 0.931 ms, slower than the per-branch problems at the higher occupancy.  Not kept.
// Also measured and not kept (r05, detector at 0.87 ms): the split-K form for the plain 3x3 launches up to 8 192 pixels (0.91 ms), 128-wide K
// chunks for the long K walks of the stride-32 level (neutral), 8 x 8 instead of 8 x 16 bottleneck tiles at the stride-8 level (0.88-0.89 ms).)
#ifndef SS_GRP_WAVES
#define SS_GRP_WAVES 6
#endif
template <int BN, bool CONV3, bool SPLITK>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(SPLITK || !SS_GRP_WAVES ? 1 : SS_GRP_WAVES)))
void k_pw_group(PwGroup G)
{
    __shared__ __attribute__((aligned(16))) _Float16 pw_lds[pw_lds_halfs<BN, 1, true>()];
    int bx = blockIdx.x, p = 0;
    for (int i = 1; i < G.n; ++i) if (bx >= G.start[i]) p = i;
    bx -= G.start[p];
    const int form = G.form[p];
    if constexpr (SPLITK) { if (form == 1) { pw_splitk_body<BN, CONV3>(G.p[p], bx, 0); return; } }
    if (BN == 80 && G.p[p].N <= 64) pw_body<64, 1, CONV3, true>(G.p[p], bx, 0, pw_lds);      // (dead code in the narrower groups)
    else pw_body<BN, 1, CONV3, true>(G.p[p], bx, 0, pw_lds);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_bneck — a C2f bottleneck (3x3 conv + SiLU -> 3x3 conv + SiLU (+ shortcut)) in ONE launch, the intermediate in LDS.
// As two k_pw launches each of these layers is a chain of dependent round trips to data the previous launch has just written
// (13-17 us per launch for 0.5 us of matrix work).  Here a workgroup owns a TH x TW tile of the block's output:
// the input tile with a 2-pixel halo is staged once (pixel pitch C + 8 halfs: the 16 pixels of an MFMA column tile spread over
// all LDS banks), the first convolution is evaluated on the tile + 1-pixel ring and written to LDS as half exactly as the
// separate launch rounds it (accumulator -> half, + bias, SiLU, -> half; ZERO outside the image: it is the second convolution's
// padding), the second convolution reads it from there, and the epilogue adds the shortcut from the staged input.  Weights:
// 64-wide slices of k = tap * C + c, double-buffered in LDS, one barrier per slice, every LDS operand of a slice requested
// before its first MFMA (v_mfma_f32_16x16x32_f16, as k_pw's 3x3 form).  Output: a channel slice of the C2f concat buffer (+ the dense copy the next
// bottleneck reads), as ss_op_conv3x3_f16 places it.
struct BnArgs {
    const __half* x; const __half* w1; const __half* b1; const __half* w2; const __half* b2;
    __half* out; int out_ld; __half* out2;
    int B, H, W, add, TW, TH, tiles_x, tiles_y;
};

// 3x3 convolution of PT x 16 pixels per wave out of an LDS tile (pixel pitch CI + 8 halfs): CI input, CO output channels,
// weights [CO][3][3][CI] streamed in 64-wide K slices through Ws [2][CO][72].
template <int CI, int CO, int PT>
__device__ __forceinline__ void bn_conv(const _Float16* __restrict__ In, const int (&pbase)[PT], const int* __restrict__ tapoff,
                                        const __half* __restrict__ w, _Float16* __restrict__ Ws, f4 (&acc)[CO / 16][PT], const int tid)
{
    constexpr int C = CI;
    constexpr int MT = CO / 16, WP = 72, K = 9 * CI, WV = (CO * 8 + 255) / 256;
    const int lane = tid & 63, q = lane >> 4, n = lane & 15;
    const h8 z8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    auto load_w = [&](int k0, h8 (&wr)[WV]) {
#pragma unroll
        for (int j = 0; j < WV; ++j) {
            const int i = tid + j * 256, r = i >> 3, c8 = i & 7, kk = k0 + c8 * 8;
            wr[j] = (i < CO * 8 && kk < K) ? *reinterpret_cast<const h8*>(w + (size_t)r * K + kk) : z8;
        }
    };
    // weight slices are requested TWO slices ahead of their LDS store: a slice's matrix work (~0.2-0.3 us) is shorter than the L2
    // latency of its successor's weights, with one slice of distance every iteration waited for them (r04: k_head / k_bneck
    // spent ~0.8 us per 64-wide slice for 0.15 us of MFMAs)
    // (measured and rejected, r04: requesting the weight slices TWO slices ahead of their LDS store — same box, same run: yolov8n on 32
    //  frames 0.968-0.980 -> 0.971-0.988 ms with it here, 1.04-1.06 ms with it in k_pw's walk as well: the extra staging registers and
    //  loads in flight cost more than the hidden L2 latency returns)
    h8 wr[WV], wn[WV];
    load_w(0, wr);
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += 64, buf ^= 1) {
        _Float16* wb = Ws + buf * (CO * WP);
#pragma unroll
        for (int j = 0; j < WV; ++j) {
            const int i = tid + j * 256, r = i >> 3, c8 = i & 7;
            if (i < CO * 8) *reinterpret_cast<h8*>(wb + r * WP + c8 * 8) = wr[j];
        }
        if (k0 + 64 < K) load_w(k0 + 64, wn);
        __syncthreads();                                           // slice k0 (and whatever the caller wrote to LDS before) is visible
        h8 b[2][PT], a[2][MT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kk = k0 + ks * 32 + 8 * q, tap = kk / C, c = kk % C;
            const bool kv = kk < K;
            const int boff = kv ? tapoff[tap] + c : 0;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                b[ks][pt] = *reinterpret_cast<const h8*>(In + pbase[pt] + boff);
                if (!kv) b[ks][pt] = z8;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[ks][mt] = *reinterpret_cast<const h8*>(wb + (mt * 16 + n) * WP + ks * 32 + 8 * q);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (k0 + ks * 32 >= K) break;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)            // k_pw's 3x3 form (v_mfma_f32_16x16x32_f16 on the same operands): the same bits as the separate launches
                    acc[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks][mt], b[ks][pt], acc[mt][pt], 0, 0, 0);
        }
        if (k0 + 64 < K) {
#pragma unroll
            for (int j = 0; j < WV; ++j) wr[j] = wn[j];
        }
    }
}

template <int C, int PT1, int PT2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(!SS_TAIL_BUMP ? 1 : C == 16 ? 6 : 1))) void k_bneck(BnArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char bn_smem[];
    constexpr int MT = C / 16, P = C + 8, WP = 72;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, n = lane & 15;
    const int TW = A.TW, TH = A.TH, XW = TW + 4, XH = TH + 4, RW = TW + 2, RH = TH + 2;
    _Float16* Xs = reinterpret_cast<_Float16*>(bn_smem);                               // [XH][XW][P]   input, origin (oy0-2, ox0-2)
    _Float16* Ts = Xs + XH * XW * P;                                                   // [RH][RW][P]   first conv's output, origin (oy0-1, ox0-1)
    _Float16* Ws = Ts + ((RH * RW * P + 7) & ~7);                                      // [2][C][WP]
    int* tap1 = reinterpret_cast<int*>(Ws + 2 * C * WP);                               // [16] tap offsets in Xs / Ts
    int* tap2 = tap1 + 16;
    const int tpi = A.tiles_x * A.tiles_y;
    const int img = blockIdx.x / tpi, trm = blockIdx.x - img * tpi, tyi = trm / A.tiles_x, txi = trm - tyi * A.tiles_x;
    const int oy0 = tyi * TH, ox0 = txi * TW, H = A.H, W = A.W;
    const h8 z8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (tid < 9) { const int ky = tid / 3, kx = tid - ky * 3; tap1[tid] = (ky * XW + kx) * P; tap2[tid] = (ky * RW + kx) * P; }
    // ---- stage the input tile (halo 2): a wave per row, 8 loads in flight per lane ----
    const __half* xi = A.x + (size_t)img * H * W * C;
    constexpr int VPC = C / 8;
    const int rowv = XW * VPC;
    for (int r = wave; r < XH; r += 4) {
        const int iy = oy0 - 2 + r;
        const bool rok = iy >= 0 && iy < H;
        const __half* src = xi + (size_t)(rok ? iy : 0) * W * C;
        _Float16* dst = Xs + r * XW * P;
        for (int i0 = lane; i0 < rowv; i0 += 512) {
            h8 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u, c = i / VPC, v = i % VPC, ix = ox0 - 2 + c;
                const bool ok = rok && i < rowv && ix >= 0 && ix < W;
                tmp[u] = ok ? *reinterpret_cast<const h8*>(src + (size_t)ix * C + v * 8) : z8;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u, c = i / VPC, v = i % VPC;
                if (i < rowv) *reinterpret_cast<h8*>(dst + c * P + v * 8) = tmp[u];
            }
        }
    }
    // ---- first convolution on the tile + 1-pixel ring -> Ts ----
    {
        int pb[PT1], tpix[PT1];
        bool inimg[PT1];
#pragma unroll
        for (int pt = 0; pt < PT1; ++pt) {
            const int p = (wave * PT1 + pt) * 16 + n;
            int ry = p / RW, rx = p - ry * RW;
            const bool ok = p < RH * RW;
            if (!ok) { ry = 0; rx = 0; }
            const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
            inimg[pt] = ok && iy >= 0 && iy < H && ix >= 0 && ix < W;
            tpix[pt] = ok ? p : -1;
            pb[pt] = (ry * XW + rx) * P;
        }
        f4 acc[MT][PT1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pt = 0; pt < PT1; ++pt) acc[mt][pt] = f4{ 0.f, 0.f, 0.f, 0.f };
        bn_conv<C, C, PT1>(Xs, pb, tap1, A.w1, Ws, acc, tid);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const h4 bb = *reinterpret_cast<const h4*>(A.b1 + mt * 16 + 4 * q);
#pragma unroll
            for (int pt = 0; pt < PT1; ++pt) {
                if (tpix[pt] < 0) continue;
                h4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float f = act_apply((float)(_Float16)acc[mt][pt][j] + (float)bb[j], 2);
                    asm volatile("" : "+v"(f));                    // product rounded to f32 THEN to half, as k_pw (no v_fma_mixlo_f16: one rounding)
                    o[j] = inimg[pt] ? (_Float16)f : (_Float16)0.f;
                }
                *reinterpret_cast<h4*>(Ts + tpix[pt] * P + mt * 16 + 4 * q) = o;
            }
        }
    }
    // (the first barrier inside bn_conv orders the Ts writes before the second convolution's reads; its weight buffers are free:
    //  every wave has left the first K walk's last slice before it writes Ts ... and the slice written next is the OTHER buffer only
    //  if the slice count is odd, so: one explicit barrier)
    __syncthreads();
    // ---- second convolution on the tile, shortcut from the staged input ----
    {
        int pb[PT2], opix[PT2], xoff[PT2];
#pragma unroll
        for (int pt = 0; pt < PT2; ++pt) {
            const int p = (wave * PT2 + pt) * 16 + n;
            int ty = p / TW, tx = p - ty * TW;
            const bool ok = p < TH * TW && oy0 + ty < H && ox0 + tx < W;
            if (!ok) { ty = 0; tx = 0; }
            pb[pt] = (ty * RW + tx) * P;
            xoff[pt] = ((ty + 2) * XW + tx + 2) * P;
            opix[pt] = ok ? (img * H + oy0 + ty) * W + ox0 + tx : -1;
        }
        f4 acc[MT][PT2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pt = 0; pt < PT2; ++pt) acc[mt][pt] = f4{ 0.f, 0.f, 0.f, 0.f };
        bn_conv<C, C, PT2>(Ts, pb, tap2, A.w2, Ws, acc, tid);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const h4 bb = *reinterpret_cast<const h4*>(A.b2 + mt * 16 + 4 * q);
#pragma unroll
            for (int pt = 0; pt < PT2; ++pt) {
                if (opix[pt] < 0) continue;
                const h4 xr = *reinterpret_cast<const h4*>(Xs + xoff[pt] + mt * 16 + 4 * q);
                h4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float f = act_apply((float)(_Float16)acc[mt][pt][j] + (float)bb[j], 2);
                    asm volatile("" : "+v"(f));
                    if (A.add) f = (float)(_Float16)f + (float)xr[j];
                    o[j] = (_Float16)f;
                }
                *reinterpret_cast<h4*>(A.out + (size_t)opix[pt] * A.out_ld + mt * 16 + 4 * q) = o;
                if (A.out2) *reinterpret_cast<h4*>(A.out2 + (size_t)opix[pt] * C + mt * 16 + 4 * q) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_head — one level of the anchor-free detect head, BOTH branches, all three layers of a branch in one launch:
//   3x3 (CI -> CM) + SiLU -> 3x3 (CM -> CM) + SiLU -> 1x1 (CM -> nout) + bias
// (nets.Detect.cv2[i] / cv3[i]; CM = 64 for the box branch, 80 for the class branch of yolov8n).  As grouped launches per depth
// (k_pw_group) these were the largest kernels of the detector: 126 + 108 + 36 us per 32 frames, each layer a round trip of
// its whole output through HBM / L2.  Here a workgroup owns a TH x TW tile of one image and one branch (blockIdx.y): k_bneck's
// scheme with CI != CM and no shortcut — input tile + 2-pixel halo staged once, first convolution on the tile + 1-pixel ring
// into LDS (zero outside the image), second convolution out of it — and the branch's final 1x1 as a third MFMA product: the
// activated output of the second convolution goes, rounded to half exactly as the separate launch stores it, into a per-wave
// LDS tile (aliased onto the input tile, which is dead by then) and is multiplied with the 1x1 weights staged in the weight
// buffers.  Same MFMA instructions, operand layouts, K order and rounding points as the grouped launches: bit-identical.
struct HeadArgs {
    const __half* x;
    const __half* w1[2]; const __half* b1[2]; const __half* w2[2]; const __half* b2[2]; const __half* w3[2]; const __half* b3[2];
    __half* out[2]; int nout[2];
    int B, H, W, TW, TH, tiles_x, tiles_y;
};

template <int CI, int CM, int PT1, int PT2>
__device__ __forceinline__ void head_body(const HeadArgs& A, const int br, char* smem)
{
    constexpr int MT = CM / 16, PX = CI + 8, PM = CM + 8, WP = 72, KB3 = (CM + 31) / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, n = lane & 15;
    const int TW = A.TW, TH = A.TH, XW = TW + 4, XH = TH + 4, RW = TW + 2, RH = TH + 2;
    _Float16* Xs = reinterpret_cast<_Float16*>(smem);                                  // [XH][XW][PX]  input, origin (oy0-2, ox0-2)
    _Float16* Ts = Xs + ((XH * XW * PX + 7) & ~7);                                     // [RH][RW][PM]  first conv's output, origin (oy0-1, ox0-1)
    _Float16* Ws = Ts + ((RH * RW * PM + 7) & ~7);                                     // [2][80][WP] weight slices; then the 1x1 weights [CM][PM]
    int* tap1 = reinterpret_cast<int*>(Ws + 2 * 80 * WP);                              // [16] tap offsets in Xs / Ts
    int* tap2 = tap1 + 16;
    _Float16* Us = Xs;                                                                 // [4 waves][PT2 * 16][PM]: second conv's output (Xs is dead by then)
    const int tpi = A.tiles_x * A.tiles_y;
    const int img = blockIdx.x / tpi, trm = blockIdx.x - img * tpi, tyi = trm / A.tiles_x, txi = trm - tyi * A.tiles_x;
    const int oy0 = tyi * TH, ox0 = txi * TW, H = A.H, W = A.W;
    const h8 z8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (tid < 9) { const int ky = tid / 3, kx = tid - ky * 3; tap1[tid] = (ky * XW + kx) * PX; tap2[tid] = (ky * RW + kx) * PM; }
    // ---- stage the input tile (halo 2): a wave per row, 8 loads in flight per lane ----
    const __half* xi = A.x + (size_t)img * H * W * CI;
    constexpr int VPC = CI / 8;
    const int rowv = XW * VPC;
    for (int r = wave; r < XH; r += 4) {
        const int iy = oy0 - 2 + r;
        const bool rok = iy >= 0 && iy < H;
        const __half* src = xi + (size_t)(rok ? iy : 0) * W * CI;
        _Float16* dst = Xs + r * XW * PX;
        for (int i0 = lane; i0 < rowv; i0 += 512) {
            h8 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u, c = i / VPC, v = i % VPC, ix = ox0 - 2 + c;
                const bool ok = rok && i < rowv && ix >= 0 && ix < W;
                tmp[u] = ok ? *reinterpret_cast<const h8*>(src + (size_t)ix * CI + v * 8) : z8;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u, c = i / VPC, v = i % VPC;
                if (i < rowv) *reinterpret_cast<h8*>(dst + c * PX + v * 8) = tmp[u];
            }
        }
    }
    // ---- first convolution (CI -> CM) on the tile + 1-pixel ring -> Ts ----
    {
        int pb[PT1], tpix[PT1];
        bool inimg[PT1];
#pragma unroll
        for (int pt = 0; pt < PT1; ++pt) {
            const int p = (wave * PT1 + pt) * 16 + n;
            int ry = p / RW, rx = p - ry * RW;
            const bool ok = p < RH * RW;
            if (!ok) { ry = 0; rx = 0; }
            const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
            inimg[pt] = ok && iy >= 0 && iy < H && ix >= 0 && ix < W;
            tpix[pt] = ok ? p : -1;
            pb[pt] = (ry * XW + rx) * PX;
        }
        f4 acc[MT][PT1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pt = 0; pt < PT1; ++pt) acc[mt][pt] = f4{ 0.f, 0.f, 0.f, 0.f };
        bn_conv<CI, CM, PT1>(Xs, pb, tap1, A.w1[br], Ws, acc, tid);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const h4 bb = *reinterpret_cast<const h4*>(A.b1[br] + mt * 16 + 4 * q);
#pragma unroll
            for (int pt = 0; pt < PT1; ++pt) {
                if (tpix[pt] < 0) continue;
                h4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float f = act_apply((float)(_Float16)acc[mt][pt][j] + (float)bb[j], 2);
                    asm volatile("" : "+v"(f));                    // rounded to f32 THEN to half, as the grouped launch (no v_fma_mixlo_f16)
                    o[j] = inimg[pt] ? (_Float16)f : (_Float16)0.f;
                }
                *reinterpret_cast<h4*>(Ts + tpix[pt] * PM + mt * 16 + 4 * q) = o;
            }
        }
    }
    __syncthreads();                                               // Ts complete; the first K walk's weight buffers and Xs are free
    // ---- second convolution (CM -> CM) on the tile -> per-wave tile Us ----
    int opix[PT2];
    {
        int pb[PT2];
#pragma unroll
        for (int pt = 0; pt < PT2; ++pt) {
            const int p = (wave * PT2 + pt) * 16 + n;
            int ty = p / TW, tx = p - ty * TW;
            const bool ok = p < TH * TW && oy0 + ty < H && ox0 + tx < W;
            if (!ok) { ty = 0; tx = 0; }
            pb[pt] = (ty * RW + tx) * PM;
            opix[pt] = ok ? (img * H + oy0 + ty) * W + ox0 + tx : -1;
        }
        f4 acc[MT][PT2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pt = 0; pt < PT2; ++pt) acc[mt][pt] = f4{ 0.f, 0.f, 0.f, 0.f };
        bn_conv<CM, CM, PT2>(Ts, pb, tap2, A.w2[br], Ws, acc, tid);
        _Float16* ut = Us + wave * (PT2 * 16 * PM);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const h4 bb = *reinterpret_cast<const h4*>(A.b2[br] + mt * 16 + 4 * q);
#pragma unroll
            for (int pt = 0; pt < PT2; ++pt) {
                h4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float f = act_apply((float)(_Float16)acc[mt][pt][j] + (float)bb[j], 2);
                    asm volatile("" : "+v"(f));
                    o[j] = (_Float16)f;
                }
                *reinterpret_cast<h4*>(ut + (pt * 16 + n) * PM + mt * 16 + 4 * q) = o;
            }
        }
    }
    __syncthreads();                                               // every wave has left the second K walk: the weight buffers are free
    // ---- the branch's 1x1 (CM -> nout): weights [nout][CM] -> LDS [CM rows max][PM], rows >= nout zero ----
    const int nout = A.nout[br];
    {
        const __half* w3 = A.w3[br];
        for (int i = tid; i < CM * (CM / 8); i += 256) {
            const int r = i / (CM / 8), c8 = i - r * (CM / 8);
            *reinterpret_cast<h8*>(Ws + r * PM + c8 * 8) = r < nout ? *reinterpret_cast<const h8*>(w3 + (size_t)r * CM + c8 * 8) : z8;
        }
    }
    __syncthreads();
    {
        const _Float16* ut = Us + wave * (PT2 * 16 * PM);
        f4 acc[MT][PT2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pt = 0; pt < PT2; ++pt) acc[mt][pt] = f4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int kb = 0; kb < KB3; ++kb) {
            const int k = kb * 32 + 8 * q;
            const bool kv = k < CM;                                // (CM = 80: the last 32-wide block is half empty, as in k_pw's walk)
            h8 b[PT2];
#pragma unroll
            for (int pt = 0; pt < PT2; ++pt) b[pt] = kv ? *reinterpret_cast<const h8*>(ut + (pt * 16 + n) * PM + k) : z8;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const h8 a = kv ? *reinterpret_cast<const h8*>(Ws + (mt * 16 + n) * PM + k) : z8;
                const h4 a0 = { a[0], a[1], a[2], a[3] }, a1 = { a[4], a[5], a[6], a[7] };
#pragma unroll
                for (int pt = 0; pt < PT2; ++pt) {                 // the 1x1 form of k_pw: two 16x16x16 steps per 32-wide block
                    const h4 b0 = { b[pt][0], b[pt][1], b[pt][2], b[pt][3] }, b1 = { b[pt][4], b[pt][5], b[pt][6], b[pt][7] };
                    acc[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, acc[mt][pt], 0, 0, 0);
                    acc[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, b1, acc[mt][pt], 0, 0, 0);
                }
            }
        }
        __half* out = A.out[br];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int oc0 = mt * 16 + 4 * q;
            if (oc0 >= nout) continue;
            const h4 bb = *reinterpret_cast<const h4*>(A.b3[br] + oc0);
#pragma unroll
            for (int pt = 0; pt < PT2; ++pt) {
                if (opix[pt] < 0) continue;
                h4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (_Float16)((float)(_Float16)acc[mt][pt][j] + (float)bb[j]);
                *reinterpret_cast<h4*>(out + (size_t)opix[pt] * nout + oc0) = o;
            }
        }
    }
}

// blockIdx.y = branch: 0 = box (CM 64), 1 = class (CM 80)
template <int CI, int PT1, int PT2>
__global__ __launch_bounds__(256) void k_head(HeadArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char hd_smem[];
    if (blockIdx.y == 0) head_body<CI, 64, PT1, PT2>(A, 0, hd_smem);
    else head_body<CI, 80, PT1, PT2>(A, 1, hd_smem);
}

// OSNet stem in one pass: conv 7x7 / stride 2 / pad 3 (3 -> 16 channels) + bias + ReLU + max pool 3x3 / stride 2 /
// pad 1, crops [N][H][128][3] half -> [N][H/4][32][16] half.  MIOpen runs the C_in = 3 convolution at 186 us for 256
// crops (5% of its data rate) and bias, ReLU and pooling are three more passes over the 67 MB conv output.
// Workgroup = 4 pooled rows x 32 columns of one crop: the 23 input rows it needs sit in LDS as flat rows of halfs with
// 16 halfs of zero margin in front, so the 21 taps (kx, ch) of conv pixel c on row ky are the window starting at half
// 6c + 7.  The MFMA's K axis is that row cut into ALIGNED 8-half blocks: wave r takes the conv pixels c = 4n + r
// (n = 0..15), whose windows all start (6r + 7) % 8 halfs into block 3n + (6r + 7) / 8 — so one 16-byte ds_read_b128
// per lane (lane (q, n): block 3n + .. + q; 48-byte lane stride = conflict-free) feeds one v_mfma_f32_16x16x32_f16 per
// ky against the weights pre-shifted by (6r + 7) % 8 inside a 32-wide K (host: fused.stem_weight, [4][7][16][32]).
// (Round 1 read 12-byte-strided 8-byte pieces for two 16x16x16 MFMAs per ky and staged with 2-byte LDS stores: 204 us per
// 512 crops, LDS-issue-bound.)  ReLU(conv+bias) lands in LDS as half; out-of-image conv rows / the left pad column hold 0,
// the identity of max over ReLU outputs; the pool phase reads 3x3 windows and writes the pooled tile.
#define STEM_W 128
#define STEM_M 16           // zero margin (halfs) in front of an LDS input row
#define STEM_PITCH 416      // 16 + 384 + 16 halfs
#define STEM_IN_ROWS 23
#define STEM_CR 9           // conv rows per tile
#define STEM_CW 65          // conv columns + left pad
#define STEM_CP 18          // halfs per pixel of the conv-output tile: 9 dwords, so that the wave's 8-byte writes (pixels 4n + r:
                            // 36-dword lane stride) and the pooling reads (2-pixel stride) spread over the LDS banks (16 halfs:
                            // 128-byte lane stride, 16-way conflicts — 46 M conflict cycles per 1024 crops, PMC)
typedef _Float16 h4a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef _Float16 h8a4 __attribute__((ext_vector_type(8), aligned(4)));

__global__ __launch_bounds__(256) void k_osnet_stem(const __half* __restrict__ x, const __half* __restrict__ wp /*[4][7][16][32]*/,
                                                   const __half* __restrict__ bias, __half* __restrict__ y, int H, int tiles,
                                                   const int* __restrict__ nvalid, const __half* __restrict__ w1c,
                                                   const __half* __restrict__ b1c, __half* __restrict__ y1)
{
    if (nvalid && (int)(blockIdx.x / tiles) >= *nvalid) return;
    __shared__ __attribute__((aligned(16))) _Float16 In[STEM_IN_ROWS * STEM_PITCH];
    __shared__ __attribute__((aligned(16))) _Float16 Cv[STEM_CR * STEM_CW * STEM_CP];
    const int tid = threadIdx.x, lane = tid & 63, r = tid >> 6, q = lane >> 4, n = lane & 15;
    const int img = blockIdx.x / tiles, j0 = (blockIdx.x - img * tiles) * 4;           // first pooled row of the tile
    const int OHc = H / 2, OHp = OHc / 2;
    const int in_r0 = 4 * j0 - 5, cv_r0 = 2 * j0 - 1;
    const h8 z8 = { 0, 0, 0, 0, 0, 0, 0, 0 };

    // ---- stage the input rows: 16-byte chunks, zeros outside the image and in the margins ----
    const __half* xi = x + (size_t)img * H * STEM_W * 3;
    constexpr int CH = STEM_PITCH / 8;                                                 // 52 chunks per LDS row
    {
        // all of a thread's chunks are requested before the first LDS store, from clamped addresses (rolled, with a predicated load per
        // iteration, this loop paid five full memory round trips per workgroup - most of a workgroup's life)
        constexpr int NIT = (STEM_IN_ROWS * CH + 255) / 256;
        h8 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = min(tid + 256 * it, STEM_IN_ROWS * CH - 1);
            const int rr = i / CH, ch = i - rr * CH;
            const int gr = min(max(in_r0 + rr, 0), H - 1), cc = min(max(ch - STEM_M / 8, 0), STEM_W * 3 / 8 - 1);
            v[it] = *reinterpret_cast<const h8*>(xi + (size_t)gr * STEM_W * 3 + cc * 8);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + 256 * it;
            if (i < STEM_IN_ROWS * CH) {
                const int rr = i / CH, ch = i - rr * CH, gr = in_r0 + rr;
                const bool ok = ch >= STEM_M / 8 && ch < STEM_M / 8 + STEM_W * 3 / 8 && gr >= 0 && gr < H;
                *reinterpret_cast<h8*>(In + rr * STEM_PITCH + ch * 8) = ok ? v[it] : z8;
            }
        }
    }
    if (tid < STEM_CR * 2) *reinterpret_cast<h8a4*>(Cv + (tid >> 1) * STEM_CW * STEM_CP + (tid & 1) * 8) = z8;      // left pad column
    // weights of this wave's residue: A operand, lane (q, oc = n): k = 8q..8q+7 of every ky
    h8 a[7];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) a[ky] = *reinterpret_cast<const h8*>(wp + ((size_t)(r * 7 + ky) * 16 + n) * 32 + 8 * q);
    const h4 bb = *reinterpret_cast<const h4*>(bias + 4 * q);
    __syncthreads();

    // ---- convolution on the matrix cores: 9 conv rows; this wave's 16 pixels of a row are c = 4n + r ----
    const int c = 4 * n + r;
    const _Float16* bp = In + 8 * (3 * n + (6 * r + 7) / 8 + q);
#pragma unroll 3
    for (int cr = 0; cr < STEM_CR; ++cr) {
        f4 d = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ky], *reinterpret_cast<const h8*>(bp + (2 * cr + ky) * STEM_PITCH), d, 0, 0, 0);
        const int gr = cv_r0 + cr;
        h4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float f = (float)(_Float16)d[j] + (float)bb[j];    // conv rounded to half, then the bias pass (as unfused)
            o[j] = (gr >= 0 && gr < OHc) ? (_Float16)(f > 0.f ? f : 0.f) : (_Float16)0.f;
        }
        *reinterpret_cast<h4a4*>(Cv + ((size_t)(cr * STEM_CW + c + 1) * STEM_CP + 4 * q)) = o;
    }
    __syncthreads();

    // ---- max pool 3x3 / stride 2: thread = (pooled row, pooled column, 8 channels) ----
    {
        const int c8 = tid & 1, pc = (tid >> 1) & 31, pr = tid >> 6;
        if (j0 + pr < OHp) {
            h8 m = *reinterpret_cast<const h8a4*>(Cv + ((size_t)((2 * pr) * STEM_CW + 2 * pc) * STEM_CP + c8 * 8));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const h8 v = *reinterpret_cast<const h8a4*>(Cv + ((size_t)((2 * pr + ky) * STEM_CW + 2 * pc + kx) * STEM_CP + c8 * 8));
#pragma unroll
                    for (int k = 0; k < 8; ++k) m[k] = v[k] > m[k] ? v[k] : m[k];
                }
            reinterpret_cast<h8*>(y)[(((size_t)img * OHp + j0 + pr) * 32 + pc) * 2 + c8] = m;
            if (w1c) *reinterpret_cast<h8*>(In + (size_t)(pr * 32 + pc) * 24 + c8 * 8) = m;       // In is free after the conv phase
        }
    }
    if (!w1c) return;
    // ---- the first OSBlock's conv1 (1x1, 16 -> 16, + bias, ReLU) on the pooled tile: k_pw's products for K = 16 (lane
    // groups q = 0, 1 hold k = 8q..8q+7, two 16x16x16 MFMAs over elements 0-3 / 4-7), its rounding and epilogue ----
    __syncthreads();
    if (j0 + r < OHp) {                                              // wave r = pooled row r of the tile
        const bool kv = q < 2;
        const h8 aw = kv ? *reinterpret_cast<const h8*>(w1c + n * 16 + 8 * q) : z8;
        const h4 a0 = { aw[0], aw[1], aw[2], aw[3] }, a1 = { aw[4], aw[5], aw[6], aw[7] };
        const h4 b1 = *reinterpret_cast<const h4*>(b1c + 4 * q);
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const h8 bv = kv ? *reinterpret_cast<const h8*>(In + (size_t)(r * 32 + pt * 16 + n) * 24 + 8 * q) : z8;
            const h4 v0 = { bv[0], bv[1], bv[2], bv[3] }, v1 = { bv[4], bv[5], bv[6], bv[7] };
            f4 d = { 0.f, 0.f, 0.f, 0.f };
            d = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, v0, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, v1, d, 0, 0, 0);
            h4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float f = (float)(_Float16)d[j] + (float)b1[j];
                o[j] = (_Float16)(f > 0.f ? f : 0.f);
            }
            *reinterpret_cast<h4*>(y1 + (((size_t)img * OHp + j0 + r) * 32 + pt * 16 + n) * 16 + 4 * q) = o;
        }
    }
}

// The detector's first convolution: 3x3 / stride 2 / pad 1 on the 3-channel letterboxed frame + bias + SiLU, [B][H][W][3]
// half -> [B][H/2][W/2][COUT].  (MIOpen's igemm + our bias pass: 32 + 15 us per 16 frames, C_in = 3 wastes its K tiles.)
// Same idea as k_osnet_stem: a row of the NHWC input is a flat array of halfs, the 9 taps (kx, ch) of conv column c on row ky
// are the window starting at flat 6c - 3; with 8 halfs of margin in front of a 64-column tile's span, the columns c = 4n + r
// start (6r + 5) % 8 halfs into the aligned 8-half block 3n + (6r + 5) / 8, and 9 taps from there never leave the next block:
// K = 16 per ky, one v_mfma_f32_16x16x16_f16 per ky against weights pre-shifted per residue r (wave r) — host:
// fused.conv0_weight, [4][3][COUT][16].  Workgroup = 8 conv rows x 64 columns; 17 input rows staged with 16-byte copies.
#define C0_PITCH 400        // halfs per LDS row: 8 margin + 384 (64 columns x 2 x 3) + 8 tail
#define C0_ROWS 8

template <int COUT>
__global__ __launch_bounds__(256) void k_conv0(const __half* __restrict__ x, const __half* __restrict__ wp, const __half* __restrict__ bias,
                                              __half* __restrict__ y, int H, int W, int act)
{
    constexpr int MT = COUT / 16, IN_ROWS = 2 * C0_ROWS + 1, CH = C0_PITCH / 8;
    __shared__ __attribute__((aligned(16))) _Float16 In[IN_ROWS * C0_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, r = tid >> 6, q = lane >> 4, n = lane & 15;
    const int OH = (H - 1) / 2 + 1, OW = W / 2;
    const int c0 = blockIdx.x * 64, oy0 = blockIdx.y * C0_ROWS, b = blockIdx.z;
    const h8 z8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const __half* xi = x + (size_t)b * H * W * 3;
    const int row_halfs = W * 3, f0 = 6 * c0 - 8;                      // flat index of LDS half 0 within an input row
    for (int i = tid; i < IN_ROWS * CH; i += 256) {
        const int rr = i / CH, ch = i - rr * CH, gr = 2 * oy0 - 1 + rr, f = f0 + ch * 8;
        *reinterpret_cast<h8*>(In + rr * C0_PITCH + ch * 8) =
            (gr >= 0 && gr < H && f >= 0 && f + 8 <= row_halfs) ? *reinterpret_cast<const h8*>(xi + (size_t)gr * row_halfs + f) : z8;
    }
    h4 a[MT][3], bb[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) a[mt][ky] = *reinterpret_cast<const h4*>(wp + ((size_t)(r * 3 + ky) * COUT + mt * 16 + n) * 16 + 4 * q);
        bb[mt] = *reinterpret_cast<const h4*>(bias + mt * 16 + 4 * q);
    }
    __syncthreads();
    const _Float16* bp = In + 8 * (3 * n + (6 * r + 5) / 8) + 4 * q;
    const int c = c0 + 4 * n + r;
#pragma unroll 2
    for (int cr = 0; cr < C0_ROWS; ++cr) {
        const int oy = oy0 + cr;
        if (oy >= OH) break;
        h4 bv[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) bv[ky] = *reinterpret_cast<const h4*>(bp + (2 * cr + ky) * C0_PITCH);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f4 d = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) d = __builtin_amdgcn_mfma_f32_16x16x16f16(a[mt][ky], bv[ky], d, 0, 0, 0);
            h4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (_Float16)act_apply((float)(_Float16)d[j] + (float)bb[mt][j], act);
            if (c < OW) *reinterpret_cast<h4*>(y + (((size_t)b * OH + oy) * OW + c) * COUT + mt * 16 + 4 * q) = o;
        }
    }
}

// OSNet "LightConv3x3" in one pass: y = relu(dw3x3(pw1x1(x)) + bias), C in {16, 24, 32}.
//
// The two-launch form (GEMM, then k_dw3x3) writes and re-reads the C-channel intermediate through HBM and
// spends a hipBLASLt launch on a [pixels x C] x [C x C] product whose K is 16..32.  Here a workgroup owns a
// TH-row band of one image: phase 1 computes the pointwise product for the band plus one halo row each side
// on the matrix cores (v_mfma_f32_16x16x16_f16: M = output channel, N = 16 pixels, K = input channel; the
// weights are the A operand and stay in registers) and parks it as f16 in LDS with a zero pad column each
// side; phase 2 is the depthwise 3x3 + bias + ReLU out of LDS (thread = pixel x 8 channels, tap weights in
// registers, same fmaf order as k_dw3x3).  HBM traffic = read x once (+2/TH halo) + write y once.
#define LC_TH 16

// The depthwise 3x3 of a LightConv on the matrix cores.  v_fma_mix_f32 / DPP / conversions issue at one wave-instruction per
// ~4.4 cycles per SIMD on gfx950 (tools/probe_valu.hip), and the 36 multiply-adds per (pixel, 4 channels) of the vector form
// were 55 % of the instructions of k_osnet_chains (190 us per stage-1 launch, VALU-issue bound, matrix cores 5 % busy).  A
// depthwise tap is a diagonal [16 x 16] matrix over the channels, so two taps are one v_mfma_f32_16x16x32_f16: lane (q, n)
// of the B operand holds k-slots 8q..8q+7 = (tap a, channels 4q..4q+3 of pixel n), (tap b, the same channels) — exactly the
// two packed registers the pointwise MFMA's output layout leaves in that lane — and lane (q, m) of the A operand holds row
// m = output channel: w_a[m] at slot (m & 3), w_b[m] at slot 4 + (m & 3) when q == m >> 2, zeros elsewhere.  D comes out as
// channels 4q..4q+3 of pixel n again: the next layer's B operand after the rounding.
// ONE accumulation order for every kernel form (k_lightconv, k_osnet_streams, k_osnet_chains — their outputs are compared
// bit for bit): accumulator = bias (fp32), then for ky = 0, 1, 2 (input rows y-1, y, y+1):
//     acc = MFMA(A1[ky], (left, centre), acc)        taps (ky, 0), (ky, 1)
//     acc = MFMA(A2[ky], (centre, right), acc)       taps  -  (zero weight on the centre), (ky, 2)
// (left, centre, right) are three consecutive register pairs, so both B operands are windows of them.  (The 16-deep
// v_mfma_f32_16x16x16_f16 for the single tap would halve A2's registers, but accumulating it onto the 32-deep form's result
// gave wrong sums with this toolchain — a dependent-MFMA hazard between the two opcodes — so one opcode it is.)  Products of halves are
// exact in fp32 and the zero slots add exact zeros; the hardware's summation inside one MFMA is not fmaf-by-fmaf, so results
// are not those of k_dw3x3's chain bit for bit (tests: nearly every output equals the two-step torch form's, the rest by
// one half ulp).  A non-finite activation poisons the 16 channels of its pixel (0 x inf), not one.
template <int C>
struct DwDiag {
    static constexpr int MT = (C + 15) / 16;
    h8 A1[MT][3], A2[MT][3];
    f4 bf[MT];
    // w9l: [9][C] taps of this layer, biasl: [C]
    __device__ __forceinline__ void init(const __half* __restrict__ w9l, const __half* __restrict__ biasl, int q, int n)
    {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int c = mt * 16 + n, c4 = mt * 16 + 4 * q;
            const bool on = c < C && (n >> 2) == q;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const _Float16 w0 = on ? (_Float16)__half2float(w9l[(ky * 3 + 0) * C + c]) : (_Float16)0.f;
                const _Float16 w1 = on ? (_Float16)__half2float(w9l[(ky * 3 + 1) * C + c]) : (_Float16)0.f;
                const _Float16 w2 = on ? (_Float16)__half2float(w9l[(ky * 3 + 2) * C + c]) : (_Float16)0.f;
                h8 a1 = { 0, 0, 0, 0, 0, 0, 0, 0 }, a2 = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool at = (n & 3) == j;
                    a1[j] = at ? w0 : (_Float16)0.f; a1[4 + j] = at ? w1 : (_Float16)0.f;
                    a2[4 + j] = at ? w2 : (_Float16)0.f;
                }
                A1[mt][ky] = a1; A2[mt][ky] = a2;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) bf[mt][e] = c4 + e < C ? __half2float(biasl[c4 + e]) : 0.f;
        }
    }
    // the same operands out of a DwTab layer (LDS): aoff = ((n >> 2) == q ? n : 16) * 16
    __device__ __forceinline__ void load(const char* tl, int aoff, int q)
    {
        constexpr int OPB = 17 * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                A1[mt][ky] = *reinterpret_cast<const h8*>(tl + (mt * 6 + 2 * ky) * OPB + aoff);
                A2[mt][ky] = *reinterpret_cast<const h8*>(tl + (mt * 6 + 2 * ky + 1) * OPB + aoff);
            }
            bf[mt] = *reinterpret_cast<const f4*>(tl + MT * 6 * OPB + mt * 64 + q * 16);
        }
    }
    // one input row's taps onto an accumulator
    static __device__ __forceinline__ f4 row(const h8& a1, const h8& a2, uint2 L, uint2 Cn, uint2 R, f4 acc)
    {
        const uint4 lc = { L.x, L.y, Cn.x, Cn.y }, cr = { Cn.x, Cn.y, R.x, R.y };
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(h8, lc), acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(h8, cr), acc, 0, 0, 0);
    }
    // the whole 3x3 of one (16 pixels, 16 channels) tile: v[ky][kx] = this lane's 4 channels of pixel (x - 1 + kx, y - 1 + ky)
    __device__ __forceinline__ f4 run(int mt, const uint2 (&v)[3][3]) const
    {
        f4 acc = bf[mt];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) acc = row(A1[mt][ky], A2[mt][ky], v[ky][0], v[ky][1], v[ky][2], acc);
        return acc;
    }
};
// rounding to half, then ReLU (round(max(x, 0)) == max(round(x), 0)): one v_pk_max_f16 per two channels
__device__ __forceinline__ uint2 ss_relu_h4(const f4& f)
{
    const h4 o = { (_Float16)f[0], (_Float16)f[1], (_Float16)f[2], (_Float16)f[3] };
    uint2 v = __builtin_bit_cast(uint2, o);
    asm("v_pk_max_f16 %0, %1, 0" : "=v"(v.x) : "v"(v.x));
    asm("v_pk_max_f16 %0, %1, 0" : "=v"(v.y) : "v"(v.y));
    return v;
}

// LDS layout of the pointwise outputs the LDS-form kernels (k_lightconv, k_osnet_streams) run the depthwise on: PS = MT * 16
// halfs per pixel; with two M-tiles a lane's four channels of tile 0 and of tile 1 sit side by side (half q * 8 + mt * 4), so
// ONE 16-byte read per tap feeds both tiles' B operands and the 64 lanes of a read cover 16 pixels x 64 contiguous bytes.  (In
// channel order the 8-byte reads of a tile hit 32 bytes of every 64: four pixels per bank group, a 4-way conflict on each of
// the 18 reads per 16 pixels — the LDS pipe, not the matrix cores, set the pace: 108 us for the 16x8 maps of 1024 crops.)
// Channels past C of the second tile are written as the zeros the MFMA produces for them (never left uninitialised: they meet
// zero weights, and 0 x NaN would not be 0).
template <int MT>
__device__ __forceinline__ int lc_slot(int mt, int q) { return MT == 2 ? q * 8 + mt * 4 : 4 * q; }
template <int C>
__device__ __forceinline__ void lc_taps(const _Float16* Pp /* pixel (x-1, y-1), this lane's slot of tile 0 */, int WPS /* (W+2) * PS */,
                                        uint2 (&v)[(C + 15) / 16][3][3])
{
    constexpr int MT = (C + 15) / 16, PS = MT * 16;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            if constexpr (MT == 2) {
                const uint4 t = *reinterpret_cast<const uint4*>(Pp + (size_t)ky * WPS + kx * PS);
                v[0][ky][kx] = uint2{ t.x, t.y }; v[1][ky][kx] = uint2{ t.z, t.w };
            } else
                v[0][ky][kx] = *reinterpret_cast<const uint2*>(Pp + (size_t)ky * WPS + kx * PS);
        }
}

// DwDiag's operands of several layers as an LDS table, for the kernels whose registers cannot hold them (32 channels x 4 layers
// = 192 registers): an operand is nonzero in 16 of its 64 lanes (q == n >> 2), so the table keeps 16 entries of 16 bytes plus
// one zero entry per operand and a lane reads its own or the zero entry (one ds_read_b128); the bias as 4 floats per q.
template <int C>
struct DwTab {
    static constexpr int MT = (C + 15) / 16, OPB = 17 * 16, LAYER = MT * 6 * OPB + MT * 64;
    // table bytes global -> LDS, 16 bytes per thread and step, all loads in flight at once
    static __device__ __forceinline__ void copy(char* tab, const char* __restrict__ g, int layers, int tid)
    {
        for (int i = tid; i < layers * LAYER / 16; i += 256) reinterpret_cast<uint4*>(tab)[i] = reinterpret_cast<const uint4*>(g)[i];
    }
    static __device__ __forceinline__ void fill(char* tab, const __half* __restrict__ w9, const __half* __restrict__ bias, int layers, int tid)
    {
        for (int i = tid; i < layers * MT * 6 * 17; i += 256) {
            const int e = i % 17, op = (i / 17) % 6, mt = (i / (17 * 6)) % MT, l = i / (17 * 6 * MT);
            const int ky = op >> 1, c = mt * 16 + e;
            h8 a = { 0, 0, 0, 0, 0, 0, 0, 0 };
            if (e < 16 && c < C) {
                const __half* wl = w9 + (size_t)l * 9 * C;
                if (!(op & 1)) { a[e & 3] = (_Float16)__half2float(wl[(ky * 3 + 0) * C + c]); a[4 + (e & 3)] = (_Float16)__half2float(wl[(ky * 3 + 1) * C + c]); }
                else a[4 + (e & 3)] = (_Float16)__half2float(wl[(ky * 3 + 2) * C + c]);
            }
            *reinterpret_cast<h8*>(tab + (size_t)l * LAYER + (mt * 6 + op) * OPB + e * 16) = a;
        }
        for (int i = tid; i < layers * MT * 16; i += 256) {
            const int c = i % (MT * 16), l = i / (MT * 16);
            reinterpret_cast<float*>(tab + (size_t)l * LAYER + MT * 6 * OPB)[c] = c < C ? __half2float(bias[(size_t)l * C + c]) : 0.f;
        }
    }
};

// the table in global memory, built once per set of weights (ss_op_dwtab_f16): the chain kernels copy it to LDS
template <int C>
__global__ __launch_bounds__(256) void k_dwtab(const __half* __restrict__ w9, const __half* __restrict__ bias, int layers, char* __restrict__ out)
{
    DwTab<C>::fill(out, w9, bias, layers, threadIdx.x);
}

template <int C>
__global__ __launch_bounds__(256) void k_lightconv(const __half* __restrict__ x, const __half* __restrict__ w1,
                                                  const __half* __restrict__ w9, const __half* __restrict__ bias,
                                                  __half* __restrict__ y, int H, int W, int bands)
{
    constexpr int KS = (C + 15) / 16, MT = KS, TH = LC_TH;
    extern __shared__ __attribute__((aligned(16))) char lc_smem[];
    constexpr int PS = MT * 16;
    _Float16* T = (_Float16*)lc_smem;                       // [(TH+2)][W+2][PS] (lc_slot order)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = blockIdx.x / bands, y0 = (blockIdx.x - img * bands) * TH;
    const int WP = W + 2;
    const h4 z4 = { 0, 0, 0, 0 };

    for (int i = tid; i < (TH + 2) * 2 * (PS / 8); i += 256) {    // zero pad columns 0 and W+1
        const int c8 = i % (PS / 8), rc = i / (PS / 8), r = rc >> 1, col = (rc & 1) ? W + 1 : 0;
        h8 z = { 0, 0, 0, 0, 0, 0, 0, 0 };
        *reinterpret_cast<h8*>(T + ((size_t)(r * WP + col) * PS + c8 * 8)) = z;
    }

    // ---- phase 1: pointwise product on the matrix cores ----
    const int q = lane >> 4, n = lane & 15;
    h4 a[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int oc = mt * 16 + n, ic0 = ks * 16 + 4 * q;
            a[mt][ks] = (oc < C && ic0 < C) ? *reinterpret_cast<const h4*>(w1 + (size_t)oc * C + ic0) : z4;
        }
    const __half* xi = x + (size_t)img * H * W * C;
    const int NT = (TH + 2) * W / 16;
    auto load_b = [&](int t, h4 (&b)[KS]) {
        const int p = t * 16 + n, r = p / W, c = p - r * W, gr = y0 - 1 + r;
        const bool ok = t < NT && gr >= 0 && gr < H;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ic0 = ks * 16 + 4 * q;
            b[ks] = (ok && ic0 < C) ? *reinterpret_cast<const h4*>(xi + ((size_t)gr * W + c) * C + ic0) : z4;
        }
    };
    h4 bcur[KS], bnext[KS];
    load_b(wave, bcur);
    for (int t = wave; t < NT; t += 4) {
        load_b(t + 4, bnext);                               // next tile's pixels in flight during this tile's MFMAs
        const int p = t * 16 + n, r = p / W, c = p - r * W;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f4 d = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x16f16(a[mt][ks], bcur[ks], d, 0, 0, 0);
            h4 o = { (_Float16)d[0], (_Float16)d[1], (_Float16)d[2], (_Float16)d[3] };
            *reinterpret_cast<h4*>(T + ((size_t)(r * WP + c + 1) * PS + lc_slot<MT>(mt, q))) = o;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bcur[ks] = bnext[ks];
    }
    __syncthreads();

    // ---- phase 2: depthwise 3x3 + bias + ReLU out of LDS, on the matrix cores (DwDiag) ----
    DwDiag<C> dw;
    dw.init(w9, bias, q, n);
    const int ntile = TH * W / 16;
    for (int ti = wave; ti < ntile; ti += 4) {
        const int p = ti * 16 + n, py = p / W, px = p - py * W, gy = y0 + py;
        uint2 v[MT][3][3];
        lc_taps<C>(T + ((size_t)(py * WP + px) * PS + lc_slot<MT>(0, q)), WP * PS, v);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int c4 = mt * 16 + 4 * q;
            const uint2 o = ss_relu_h4(dw.run(mt, v[mt]));
            if (gy < H && c4 < C) *reinterpret_cast<uint2*>(y + ((((size_t)img * H + gy) * W + px) * C + c4)) = o;
        }
    }
}

// The four LightConv3x3 chains of an OSNet block (1, 2, 3 and 4 layers deep, all reading the same x1) in one launch.
// blockIdx.y = chain; a workgroup takes a TH-row band of one image with a halo of t rows each side and runs its t
// layers back to back out of LDS: layer l's pointwise product (matrix cores) goes to P, its depthwise+bias+ReLU
// reads P and writes the next layer's input X (LDS) or, for the last layer, the chain's output in HBM.  The valid
// rows shrink by one per layer and side (halo recompute factor (TH+t+1)/TH); rows outside the image are forced to
// zero after every layer, as the per-layer zero padding requires.  Per chain the intermediates never touch HBM:
// traffic = x1 band (+halo) in, y_t out, instead of t x (in + out).  The last layer also leaves the band's channel
// sums (psum), so the aggregation gate needs no separate mean pass.  Arithmetic per layer is k_lightconv's, bit for bit.
struct StreamOut { __half* y[4]; };
#define OS_TMAX 4

template <int C>
__global__ __launch_bounds__(256) void k_osnet_streams(const __half* __restrict__ x, const __half* __restrict__ w1,
                                                      const char* __restrict__ gtab,
                                                      StreamOut out, float* __restrict__ psum, int N, int H, int W,
                                                      int bands, const int* __restrict__ nvalid)
{
    if (nvalid && (int)(blockIdx.x / bands) >= *nvalid) return;
    constexpr int KS = (C + 15) / 16, MT = KS, C8 = C / 8, TH = LC_TH;
    extern __shared__ __attribute__((aligned(16))) char lc_smem[];
    const int WP = W + 2;
    constexpr int PS = MT * 16;
    _Float16* P = (_Float16*)lc_smem;                                        // [TH+2*TMAX][W+2][PS] (lc_slot order)
    _Float16* X = P + (size_t)(TH + 2 * OS_TMAX) * WP * PS;                  // [TH+2*TMAX][W][PS] (lc_slot order: one 16-byte read per pixel feeds both K halves of the pointwise MFMA)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, n16 = lane & 15;
    const int t = blockIdx.y + 1, lbase = (t * (t - 1)) / 2;
    const int aoff = ((n16 >> 2) == q ? n16 : 16) * 16;
    const int img = blockIdx.x / bands, band = blockIdx.x - img * bands, y0 = band * TH;
    const int R = TH + 2 * t;                                                // local row i <-> image row y0 - t + i
    const h4 z4 = { 0, 0, 0, 0 };
    const h8 z8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int i = tid; i < R * 2 * (PS / 8); i += 256) {                      // zero pad columns of P
        const int c8 = i % (PS / 8), rc = i / (PS / 8), r = rc >> 1, col = (rc & 1) ? W + 1 : 0;
        *reinterpret_cast<h8*>(P + ((size_t)(r * WP + col) * PS + c8 * 8)) = z8;
    }
    const __half* xi = x + (size_t)img * H * W * C;
    float s4[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int k = 0; k < 4; ++k) s4[mt][k] = 0.f;
    // The band's input rows (R consecutive image rows = one contiguous span of x) go to X with coalesced 16-byte loads,
    // all in flight at once; rows outside the image are zeros.  (Reading the first layer's MFMA operand straight from
    // global memory was a chain of dependent 8-byte loads, one memory latency per 16 pixels: 357 us per launch at 512
    // crops before, see profiles/.)
    for (int i = tid; i < R * W * C8; i += 256) {
        const int r = i / (W * C8), gr = y0 - t + r, pc8 = i - r * W * C8, pix = pc8 / C8, c8 = pc8 - pix * C8;
        const h8 v = (gr >= 0 && gr < H) ? *reinterpret_cast<const h8*>(xi + (size_t)gr * W * C + (size_t)pc8 * 8) : z8;
        if constexpr (MT == 2) {                              // channels 8 c8 .. 8 c8 + 7 = K half c8 >> 1, lanes q = 2 (c8 & 1) and + 1
            _Float16* d = X + ((size_t)r * W + pix) * PS;
            const uint4 u = __builtin_bit_cast(uint4, v);
            *reinterpret_cast<uint2*>(d + lc_slot<MT>(c8 >> 1, 2 * (c8 & 1))) = uint2{ u.x, u.y };
            *reinterpret_cast<uint2*>(d + lc_slot<MT>(c8 >> 1, 2 * (c8 & 1) + 1)) = uint2{ u.z, u.w };
        } else
            *reinterpret_cast<h8*>(X + (size_t)i * 8) = v;
    }
    if constexpr (MT == 2 && C < 32)                          // the slots of channels C .. 31 are never written again: zeros (they meet zero weights)
        for (int i = tid; i < R * W * ((32 - C) / 4); i += 256) {
            const int pix = i / ((32 - C) / 4), j = i - pix * ((32 - C) / 4);
            *reinterpret_cast<uint2*>(X + (size_t)pix * PS + lc_slot<MT>(1, (C - 16) / 4 + j)) = uint2{ 0u, 0u };
        }
    __syncthreads();

    for (int l = 1; l <= t; ++l) {
        const int Lw = lbase + l - 1;
        // this layer's depthwise operands straight from the global table (lane's own entry or the zero entry), in flight during
        // the pointwise phase (an LDS copy of the table cost two of five resident workgroups per CU)
        DwDiag<C> dw;
        dw.load(gtab + (size_t)Lw * DwTab<C>::LAYER, aoff, q);
        // ---- pointwise product of local rows [l-1, R-l] ----
        h4 a[MT][KS];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int oc = mt * 16 + n16, ic0 = ks * 16 + 4 * q;
                a[mt][ks] = (oc < C && ic0 < C) ? *reinterpret_cast<const h4*>(w1 + ((size_t)Lw * C + oc) * C + ic0) : z4;
            }
        const int rlo = l - 1, NT = (R - 2 * l + 2) * W / 16;
        auto load_b = [&](int ti, h4 (&b)[KS]) {
            const int p = ti * 16 + n16, rr = p / W, c = p - rr * W, r = rlo + rr;
            if constexpr (MT == 2) {
                const uint4 u = ti < NT ? *reinterpret_cast<const uint4*>(X + ((size_t)r * W + c) * PS + lc_slot<MT>(0, q)) : uint4{ 0u, 0u, 0u, 0u };
                b[0] = __builtin_bit_cast(h4, uint2{ u.x, u.y }); b[1] = __builtin_bit_cast(h4, uint2{ u.z, u.w });
            } else
                b[0] = ti < NT ? *reinterpret_cast<const h4*>(X + ((size_t)r * W + c) * PS + 4 * q) : z4;
        };
        h4 bcur[KS], bnext[KS];
        load_b(wave, bcur);
        for (int ti = wave; ti < NT; ti += 4) {
            load_b(ti + 4, bnext);
            const int p = ti * 16 + n16, rr = p / W, c = p - rr * W, r = rlo + rr;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f4 d = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x16f16(a[mt][ks], bcur[ks], d, 0, 0, 0);
                h4 o = { (_Float16)d[0], (_Float16)d[1], (_Float16)d[2], (_Float16)d[3] };
                *reinterpret_cast<h4*>(P + ((size_t)(r * WP + c + 1) * PS + lc_slot<MT>(mt, q))) = o;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bcur[ks] = bnext[ks];
        }
        __syncthreads();

        // ---- depthwise 3x3 + bias + ReLU of local rows [l, R-l-1], on the matrix cores (DwDiag: k_lightconv's arithmetic) ----
        // The valid rows are contiguous in X and in the output, so pixel p of the region sits at (l * W + p) there; only P (padded
        // rows) needs the (row, column) split — a shift for the power-of-two widths of OSNet's maps.
        {
            const int npx = (R - 2 * l) * W, ntile = (npx + 15) / 16;
            const bool pow2 = (W & (W - 1)) == 0;
            const int wsh = 31 - __builtin_clz(W);
            const size_t obase = ((size_t)img * H + (y0 - t + l)) * W;              // output pixel index of p = 0 (may be "negative" rows: masked)
            for (int ti = wave; ti < ntile; ti += 4) {
                const int p0 = ti * 16 + n16, p = p0 < npx ? p0 : npx - 1;  // a ragged last tile (8-wide maps) repeats its last pixel
                const int pr = pow2 ? p >> wsh : p / W, px = p - pr * W, py = pr + l, gr = y0 - t + py;
                const bool inside = gr >= 0 && gr < H;
                uint2 v[MT][3][3];
                lc_taps<C>(P + ((size_t)((py - 1) * WP + px) * PS + lc_slot<MT>(0, q)), WP * PS, v);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int c4 = mt * 16 + 4 * q;
                    const bool cok = (C % 16 == 0) || c4 < C;
                    const uint2 o = ss_relu_h4(dw.run(mt, v[mt]));
                    if (!cok || p0 >= npx) continue;
                    if (l < t) {
                        *reinterpret_cast<uint2*>(X + ((size_t)(l * W + p) * PS + lc_slot<MT>(mt, q))) = inside ? o : uint2{ 0u, 0u };
                    } else if (inside) {
                        *reinterpret_cast<uint2*>(out.y[t - 1] + ((obase + p) * C + c4)) = o;
                        const h4 ov = __builtin_bit_cast(h4, o);
#pragma unroll
                        for (int k = 0; k < 4; ++k) s4[mt][k] += (float)ov[k];
                    }
                }
            }
        }
        __syncthreads();
    }

    // band's channel sums of the chain output, fixed order (deterministic): the 16 pixel lanes of a wave, then the four waves
    float* red = reinterpret_cast<float*>(lc_smem);                          // [4][MT * 16]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = s4[mt][k];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (n16 == 0) red[wave * (MT * 16) + mt * 16 + 4 * q + k] = v;
        }
    __syncthreads();
    if (tid < C) psum[(((size_t)(t - 1) * N + img) * bands + band) * C + tid] =
        ((red[tid] + red[MT * 16 + tid]) + red[2 * MT * 16 + tid]) + red[3 * MT * 16 + tid];
}

// ---- the LightConv chains as a register-resident row stream (W = 8, 16 or 32; 8: two images per 16-lane tile) ----
// k_osnet_streams keeps every layer's pointwise output in LDS and reads it back nine times per output (9 x 16 bytes per
// (pixel, 8 channels)): at 512 crops the stage-1 launch is bound by VALU issue + LDS traffic (186 us, rocprofv3).  The MFMA
// output layout makes LDS unnecessary: after v_mfma_f32_16x16x16_f16 (A = weights, B = 16 pixels of one image row) lane
// (q, n) holds output channels 4q..4q+3 of pixel n —
//   * the depthwise 3x3 is per channel, so its column neighbours are the same registers of lanes n-1 / n+1 of the same
//     16-lane DPP row (v_mov_dpp row_shr:1 / row_shl:1; across the two tiles of a 32-wide row: row_ror of the other tile's
//     register as the `old` operand; the image's left / right zero padding is DPP's bound_ctrl zero),
//   * its row neighbours are the previous two rows' registers of the same lane when a wave walks down the rows,
//   * and the depthwise output — 4 channels of pixel n in lane (q, n) — IS the B operand of the next layer's MFMA.
// So a wave streams the rows of its band through all layers of a chain (layer l runs one row behind layer l-1) with no
// LDS, no barrier and no address arithmetic besides the row pointer; arithmetic per output is k_lightconv's, bit for bit
// (same MFMA operands, taps in (ky, kx) order on a float accumulator that starts at the bias, ReLU, one rounding to half).
// Rows outside the image are zero at every layer (the per-layer zero padding), rows of the band's halo that are not
// computed from real data only feed rows that are discarded.
__device__ __forceinline__ unsigned ss_dpp_shr1_z(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned ss_dpp_shl1_z(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned ss_dpp_shr1_o(unsigned old, unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x111, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned ss_dpp_shl1_o(unsigned old, unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x101, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned ss_dpp_ror1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned ss_dpp_ror15(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x12F, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned ss_pk_relu(unsigned v) { unsigned d; asm("v_pk_max_f16 %0, %1, 0" : "=v"(d) : "v"(v)); return d; }
// A depthwise output row is finished over three steps without changing the order of its accumulation (DwDiag: bias, then the
// input rows y-1, y, y+1): when pointwise row r arrives, the accumulator of output row r+1 starts (bias + its ky = 0 MFMAs),
// that of row r continues (ky = 1) and that of row r-1 finishes (ky = 2) — two partial accumulators per layer instead of two
// rows x (left, centre, right) of history.  The (left, centre) operands (12 registers per layer) stay in registers for the life
// of the wave at 16 channels; the (centre, right) operands and the bias — and at 24 / 32 channels all of them — are read from
// the workgroup's LDS copy of DwTab in every step.
// A step runs its T layers first to last, each consuming the row the one before just finished (xs).
// OS_SKEW 1 (A/B build, -DOS_SKEW=1): layer l consumes the row layer l-1 finished in the PREVIOUS step, so the T layers of a step
// are independent of one another — measured slower (150 vs 149 us at 16 channels, 101 vs 94 at 24: T-1 more steps, no better
// overlap), as was asking the scheduler for "one MFMA, three vector instructions" groups (+10 us); profiles/r04_osnet_dw_mfma_ab.txt.
#ifndef OS_SKEW
#define OS_SKEW 0
#endif
template <int C, int NT, int T, bool ALDS, bool PAIR>
struct OsChain {
    static constexpr int MT = (C + 15) / 16, KS = MT, CP = MT * 16;
    uint2 A[T][MT][KS];                  // pointwise weights (MFMA A operand)
    h8 A1[ALDS ? 1 : T][MT][3];          // DwDiag's (left, centre) operands; (centre, right) and the bias come from DwTab per step (ALDS: all three)
    f4 acc0[T][NT][MT], acc1[T][NT][MT];
    uint2 xs[T > 1 ? T - 1 : 1][NT][MT]; // layer l's output row = layer l+1's input
    float s[MT][4];                      // channel sums of the stored outputs
    unsigned mL, mR;                     // PAIR (two 8-wide images in one 16-lane tile): lane 8 has no left, lane 7 no right neighbour

    __device__ __forceinline__ void init(const __half* w1, const char* tab, int aoff, int q, int n)
    {
        mL = (PAIR && n == 8) ? 0u : ~0u; mR = (PAIR && n == 7) ? 0u : ~0u;
        constexpr int lbase = (T * (T - 1)) / 2;
        const uint2 z = { 0u, 0u };
        const f4 z4 = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int l = 0; l < T; ++l) {
            if constexpr (!ALDS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
                        A1[l][mt][ky] = *reinterpret_cast<const h8*>(tab + (size_t)(lbase + l) * DwTab<C>::LAYER + (mt * 6 + 2 * ky) * DwTab<C>::OPB + aoff);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int oc = mt * 16 + n;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int ic0 = ks * 16 + 4 * q;
                    A[l][mt][ks] = (oc < C && ic0 < C) ? *reinterpret_cast<const uint2*>(w1 + ((size_t)(lbase + l) * C + oc) * C + ic0) : z;
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc0[l][j][mt] = acc1[l][j][mt] = z4;
                    if (l + 1 < T) xs[l][j][mt] = z;
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[mt][j] = 0.f;
    }

    // One step: `in` = layer 0's input row `row` (zeros outside the image); layer l finishes its output row (row - l - 1);
    // `out` = the last layer's (row - T).
    __device__ __forceinline__ void step(const uint2 (&in)[NT][KS], int row, int H, const char* tab, int aoff, int q, uint2 (&out)[NT][MT])
    {
        constexpr int lbase = (T * (T - 1)) / 2;
        asm volatile("" : "+v"(aoff));                      // keeps the table reads inside the row loop (hoisted, they cost 24 VGPRs per layer)
#pragma unroll
        for (int li = 0; li < T; ++li) {
            const int l = OS_SKEW ? T - 1 - li : li;         // (OS_SKEW: last to first, a layer reads xs[l-1] before layer l-1 replaces it)
            uint2 x[NT][KS];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) x[j][ks] = l == 0 ? in[j][ks] : xs[l > 0 ? l - 1 : 0][j][ks];
            uint2 pc[NT][MT], pl[NT][MT], pr[NT][MT];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f4 d = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        d = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, A[l][mt][ks]), __builtin_bit_cast(h4, x[j][ks]), d, 0, 0, 0);
                    const h4 o = { (_Float16)d[0], (_Float16)d[1], (_Float16)d[2], (_Float16)d[3] };
                    pc[j][mt] = __builtin_bit_cast(uint2, o);
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const uint2 c = pc[j][mt];
                    if (j == 0) { pl[j][mt].x = ss_dpp_shr1_z(c.x); pl[j][mt].y = ss_dpp_shr1_z(c.y); }
                    else {
                        const uint2 pv = pc[j - 1][mt];
                        pl[j][mt].x = ss_dpp_shr1_o(ss_dpp_ror1(pv.x), c.x); pl[j][mt].y = ss_dpp_shr1_o(ss_dpp_ror1(pv.y), c.y);
                    }
                    if (j == NT - 1) { pr[j][mt].x = ss_dpp_shl1_z(c.x); pr[j][mt].y = ss_dpp_shl1_z(c.y); }
                    else {
                        const uint2 nx = pc[j + 1][mt];
                        pr[j][mt].x = ss_dpp_shl1_o(ss_dpp_ror15(nx.x), c.x); pr[j][mt].y = ss_dpp_shl1_o(ss_dpp_ror15(nx.y), c.y);
                    }
                    if constexpr (PAIR) { pl[j][mt].x &= mL; pl[j][mt].y &= mL; pr[j][mt].x &= mR; pr[j][mt].y &= mR; }
                }
            const int orow = row - (OS_SKEW ? 2 * l : l) - 1;
            const bool inside = orow >= 0 && orow < H;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const uint2 L = pl[j][mt], Cn = pc[j][mt], R = pr[j][mt];
                    h8 a1[3], a2[3];
                    const char* tl = tab + (size_t)(lbase + l) * DwTab<C>::LAYER;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        if constexpr (ALDS) a1[ky] = *reinterpret_cast<const h8*>(tl + (mt * 6 + 2 * ky) * DwTab<C>::OPB + aoff);
                        else a1[ky] = A1[l][mt][ky];
                        a2[ky] = *reinterpret_cast<const h8*>(tl + (mt * 6 + 2 * ky + 1) * DwTab<C>::OPB + aoff);
                    }
                    const f4 bfv = *reinterpret_cast<const f4*>(tl + MT * 6 * DwTab<C>::OPB + mt * 64 + q * 16);
                    // finish row (orow): ky = 2; continue row (orow + 1): ky = 1; start row (orow + 2): bias, ky = 0
                    const f4 f = DwDiag<C>::row(a1[2], a2[2], L, Cn, R, acc1[l][j][mt]);
                    acc1[l][j][mt] = DwDiag<C>::row(a1[1], a2[1], L, Cn, R, acc0[l][j][mt]);
                    acc0[l][j][mt] = DwDiag<C>::row(a1[0], a2[0], L, Cn, R, bfv);
                    uint2 ov = ss_relu_h4(f);
                    if (!inside) ov = uint2{ 0u, 0u };
                    if (l + 1 < T) xs[l][j][mt] = ov; else out[j][mt] = ov;
                }
        }
    }
};

template <int C, int NT, int T, bool ALDS, bool PAIR>
__device__ __forceinline__ void os_chain_run(const __half* __restrict__ xi, const __half* __restrict__ w1, const char* tab,
                                             __half* __restrict__ yo, float* __restrict__ ps, int H,
                                             int y0, int TH, int lane, bool img1ok, int ps_img_stride)
{
    // PAIR: 8-wide maps, the 16-lane tile holds row `row` of TWO images (lanes n < 8: image 0 at xi / yo / ps, n >= 8: the next
    // image, one image further in all three; img1ok: it exists)
    constexpr int MT = (C + 15) / 16, KS = MT, W = PAIR ? 8 : 16 * NT;
    static_assert(!PAIR || NT == 1, "PAIR: one tile of two 8-wide images");
    const int q = lane >> 4, n = lane & 15;
    const int col = PAIR ? (n & 7) : n;
    const size_t lane_img = PAIR ? (size_t)(n >> 3) * H * W * C : 0;
    const bool lane_ok = !PAIR || n < 8 || img1ok;
    OsChain<C, NT, T, ALDS, PAIR> ch;
    const int aoff = ((n >> 2) == q ? n : 16) * 16;
    ch.init(w1, tab, aoff, q, n);
    const uint2 z = { 0u, 0u };
    auto load_row = [&](int row, uint2 (&r)[NT][KS]) {
        const bool ok = row >= 0 && row < H && lane_ok;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int ic0 = ks * 16 + 4 * q;
                r[j][ks] = (ok && ic0 < C) ? *reinterpret_cast<const uint2*>(xi + lane_img + ((size_t)row * W + j * 16 + col) * C + ic0) : z;
            }
    };
    const int yend = (y0 + TH < H ? y0 + TH : H);
    auto emit = [&](int orow, const uint2 (&o)[NT][MT]) {
        if (orow < y0 || orow >= yend) return;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int c4 = mt * 16 + 4 * q;
                if (c4 < C && lane_ok) {
                    *reinterpret_cast<uint2*>(yo + lane_img + ((size_t)orow * W + j * 16 + col) * C + c4) = o[j][mt];   // (a non-temporal store here costs the tail more than it saves: the rows are still in the last-level cache when it reads them)
                    const h4 v = __builtin_bit_cast(h4, o[j][mt]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ch.s[mt][e] += (float)v[e];
                }
            }
    };
    // rows y0 - T .. enter layer 0 (the band's last output row yend - 1 leaves the last layer DELAY steps after it entered; rows
    // past yend - 1 + T only feed discarded rows); three rows are in flight ahead of the one being processed
    constexpr int DELAY = OS_SKEW ? 2 * T - 1 : T;          // steps between a row entering layer 0 and leaving the last layer
    const int r_first = y0 - T, r_last = yend - 1 + DELAY;
    uint2 r0[NT][KS], r1[NT][KS], r2[NT][KS], o[NT][MT];
    load_row(r_first, r0); load_row(r_first + 1, r1); load_row(r_first + 2, r2);
    for (int row = r_first; row <= r_last; row += 3) {
        ch.step(r0, row, H, tab, aoff, q, o); load_row(row + 3, r0); emit(row - DELAY, o);
        if (row + 1 <= r_last) { ch.step(r1, row + 1, H, tab, aoff, q, o); load_row(row + 4, r1); emit(row + 1 - DELAY, o); }
        if (row + 2 <= r_last) { ch.step(r2, row + 2, H, tab, aoff, q, o); load_row(row + 5, r2); emit(row + 2 - DELAY, o); }
    }
    // band sums of this lane's channels: over the pixel lanes of the image (16, or 8 per image of a pair), fixed order
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = ch.s[mt][e];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
            if constexpr (!PAIR) v += __shfl_xor(v, 8);
            if (col == 0 && lane_ok && mt * 16 + 4 * q + e < C) ps[(PAIR ? (n >> 3) * ps_img_stride : 0) + mt * 16 + 4 * q + e] = v;
        }
}

// wave = (image — PAIR: two consecutive 8-wide images —, band of TH rows); blockIdx.y selects the chains the wave runs (bit t-1 of
// nibble blockIdx.y of `masks`)
template <int C, int NT, bool PAIR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_osnet_chains(const __half* __restrict__ x, const __half* __restrict__ w1,
                                                     const char* __restrict__ gtab,
                                                     StreamOut out, float* __restrict__ psum, int N, int H, int TH, int bands,
                                                     unsigned masks, const int* __restrict__ nvalid)
{
    constexpr int W = PAIR ? 8 : 16 * NT;
    constexpr bool ALDS = C > 16;                            // 16 channels: the (left, centre) operands of four layers stay in registers
    __shared__ __attribute__((aligned(16))) char tab[10 * DwTab<C>::LAYER];
    DwTab<C>::copy(tab, gtab, 10, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wave, units = PAIR ? (N + 1) / 2 : N;
    if (item >= units * bands) return;
    const int unit = item / bands, band = item - unit * bands, y0 = band * TH, img = PAIR ? 2 * unit : unit;
    const int nv = nvalid ? min(*nvalid, N) : N;
    if (img >= nv) return;
    const bool img1ok = img + 1 < nv;
    const unsigned m = (masks >> (4 * blockIdx.y)) & 15u;
    const __half* xi = x + (size_t)img * H * W * C;
    const size_t io = (size_t)img * H * W * C;
#define SS_CH(TT)                                                                                                              \
    if (m & (1u << (TT - 1)))                                                                                                  \
        os_chain_run<C, NT, TT, ALDS, PAIR>(xi, w1, tab, out.y[TT - 1] + io, psum + (((size_t)(TT - 1) * N + img) * bands + band) * C, H, y0, \
                                            TH, lane, img1ok, bands * C)
    SS_CH(4); SS_CH(3); SS_CH(2); SS_CH(1);
#undef SS_CH
}

// 2x2 / stride 2 average pooling (OSNet's stage transitions): thread = (output pixel, 8 channels), fp32 sum in the
// library's order ((a00 + a01) + a10) + a11, / 4.  (torch's NHWC kernel runs at 1.1 TB/s on these shapes.)
__global__ __launch_bounds__(256) void k_avgpool2(const __half* __restrict__ x, __half* __restrict__ y, int N, int H, int W, int C8)
{
    const int OH = H / 2, OW = W / 2;
    const size_t total = (size_t)N * OH * OW * C8;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const size_t p = i / C8;
        const int ow = (int)(p % OW), oh = (int)((p / OW) % OH);
        const size_t nimg = p / ((size_t)OW * OH);
        const h8* r0 = reinterpret_cast<const h8*>(x) + ((nimg * H + 2 * oh) * W + 2 * ow) * C8 + c8;
        const h8* r1 = r0 + (size_t)W * C8;
        const h8 a = r0[0], b = r0[C8], c = r1[0], d = r1[C8];
        h8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (_Float16)(((((float)a[k] + (float)b[k]) + (float)c[k]) + (float)d[k]) / 4.0f);
        reinterpret_cast<h8*>(y)[i] = o;
    }
}

// concat(nearest-upsample x2 (lo), hi) along channels in one pass (the detector neck: F.interpolate + torch.cat were two
// launches and an extra round trip of the upsampled tensor).  thread = (output pixel, 8 channels).
__global__ __launch_bounds__(256) void k_upcat(const __half* __restrict__ lo, const __half* __restrict__ hi, __half* __restrict__ out,
                                              int B, int h, int w, int C1_8, int C2_8, int lo_first)
{
    const int H = 2 * h, W = 2 * w, CT = C1_8 + C2_8;
    const size_t total = (size_t)B * H * W * CT;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CT);
        const size_t p = i / CT;
        const int x = (int)(p % W), y = (int)((p / W) % H);
        const size_t b = p / ((size_t)W * H);
        const bool from_lo = lo_first ? cg < C1_8 : cg >= C2_8;
        const int c = lo_first ? (from_lo ? cg : cg - C1_8) : (from_lo ? cg - C2_8 : cg);
        reinterpret_cast<h8*>(out)[i] = from_lo ? reinterpret_cast<const h8*>(lo)[((b * h + (y >> 1)) * w + (x >> 1)) * C1_8 + c]
                                                 : reinterpret_cast<const h8*>(hi)[p * C2_8 + c];
    }
}

// SPPF's pooling pyramid in one launch: out = concat(x, m(x), m(m(x)), m(m(m(x)))) with m = max pool 5x5 / stride 1 / pad 2
// (padding never wins a max, so the cascade is exact).  Workgroup = (image, 8 channels): the map lives in LDS, each pool is
// a horizontal then a vertical pass of 5 taps.  H*W <= 1024.
__global__ __launch_bounds__(256) void k_sppf_pools(const __half* __restrict__ x, __half* __restrict__ out, int H, int W, int C8)
{
    extern __shared__ __attribute__((aligned(16))) char sp_smem[];
    h8* a = reinterpret_cast<h8*>(sp_smem);
    h8* t = a + H * W;
    const int b = blockIdx.x / C8, cg = blockIdx.x - b * C8, HW = H * W;
    const h8* xi = reinterpret_cast<const h8*>(x) + (size_t)b * HW * C8 + cg;
    h8* oi = reinterpret_cast<h8*>(out) + (size_t)b * HW * 4 * C8 + cg;
    for (int p = threadIdx.x; p < HW; p += 256) {
        const h8 v = xi[(size_t)p * C8];
        a[p] = v;
        oi[(size_t)p * 4 * C8] = v;
    }
    __syncthreads();
    for (int lvl = 1; lvl <= 3; ++lvl) {
        for (int p = threadIdx.x; p < HW; p += 256) {               // horizontal
            const int y = p / W, xx = p - y * W;
            h8 m = a[p];
            for (int d = -2; d <= 2; ++d) {
                const int xs = xx + d;
                if (d == 0 || xs < 0 || xs >= W) continue;
                const h8 v = a[y * W + xs];
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = v[k] > m[k] ? v[k] : m[k];
            }
            t[p] = m;
        }
        __syncthreads();
        for (int p = threadIdx.x; p < HW; p += 256) {               // vertical
            const int y = p / W, xx = p - y * W;
            h8 m = t[p];
            for (int d = -2; d <= 2; ++d) {
                const int ys = y + d;
                if (d == 0 || ys < 0 || ys >= H) continue;
                const h8 v = t[ys * W + xx];
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = v[k] > m[k] ? v[k] : m[k];
            }
            oi[((size_t)p * 4 + lvl) * C8] = m;
            // the next level reads `a`; nobody reads it any more in this level
            a[p] = m;
        }
        __syncthreads();
    }
}

// OSNet unified aggregation gate over T <= 4 streams.
//   step 1: mean over H*W of every stream -> means[t][n][C] (f32)
//   step 2: g_t = sigmoid(fc2(relu(fc1(mean_t)))) per sample, out = sum_t x_t * g_t
struct GatePtrs { const __half* x[4]; };

__global__ __launch_bounds__(256) void k_gate_mean(GatePtrs in, float* __restrict__ means, int N, int HW, int C)
{
    // thread = (pixel lane, 8-channel group): 16-byte loads, f32 partial sums, LDS tree over the pixel lanes
    __shared__ float red[256][8];
    const int t = blockIdx.y, n = blockIdx.x;
    const int C8 = C / 8;
    const int pix_par = 256 / C8;
    const int c8 = threadIdx.x % C8, p0 = threadIdx.x / C8;
    const h8* x = reinterpret_cast<const h8*>(in.x[t] + (size_t)n * HW * C);
    float s[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (p0 < pix_par)
        for (int p = p0; p < HW; p += pix_par) {
            h8 v = x[(size_t)p * C8 + c8];
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += (float)v[k];
        }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = s[k];
    __syncthreads();
    if (threadIdx.x < C) {
        const int g = threadIdx.x / 8, k = threadIdx.x % 8;
        float tot = 0.f;
        for (int q = 0; q < pix_par; ++q) tot += red[q * C8 + g][k];
        means[((size_t)t * N + n) * C + threadIdx.x] = tot / (float)HW;
    }
}

// max pooling k x k / stride / pad (-inf padding), NHWC half; thread = (output pixel, 8 channels)
__global__ __launch_bounds__(256) void k_maxpool(const __half* __restrict__ x, __half* __restrict__ y, int N, int H, int W,
                                                int C8, int k, int stride, int pad, int OH, int OW)
{
    const size_t total = (size_t)N * OH * OW * C8;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const size_t p = i / C8;
        const int ow = (int)(p % OW), oh = (int)((p / OW) % OH);
        const size_t n = p / ((size_t)OW * OH);
        float m[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = -INFINITY;
        for (int dy = 0; dy < k; ++dy) {
            const int hh = oh * stride - pad + dy;
            if (hh < 0 || hh >= H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int ww = ow * stride - pad + dx;
                if (ww < 0 || ww >= W) continue;
                h8 v = reinterpret_cast<const h8*>(x)[((n * H + hh) * W + ww) * C8 + c8];
#pragma unroll
                for (int q = 0; q < 8; ++q) m[q] = fmaxf(m[q], (float)v[q]);
            }
        }
        h8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (_Float16)m[q];
        reinterpret_cast<h8*>(y)[i] = o;
    }
}

// means[t][n][parts][C]: `parts` partial sums per (stream, image), scaled by `scale` (parts = 1, scale = 1: plain means)
__global__ __launch_bounds__(256) void k_gate_apply(GatePtrs in, int T, const float* __restrict__ means, int parts,
                                                   float scale, const __half* __restrict__ w1,
                                                   const __half* __restrict__ b1, const __half* __restrict__ w2,
                                                   const __half* __restrict__ b2, __half* __restrict__ out, int N, int HW,
                                                   int C, int Cr)
{
    __shared__ float g[4][256];
    __shared__ float hid[4][16];
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < T * C; i += 256) {          // g doubles as the mean buffer until the gates are known
        const int t = i / C, c = i % C;
        const float* m = means + (((size_t)t * N + n) * parts) * C + c;
        float a = 0.f;
        for (int p = 0; p < parts; ++p) a += m[(size_t)p * C];
        g[t][c] = a * scale;
    }
    __syncthreads();
    if (threadIdx.x < T * Cr) {
        const int t = threadIdx.x / Cr, r = threadIdx.x % Cr;
        float a = __half2float(b1[r]);
        const float* m = g[t];
        for (int c = 0; c < C; ++c) a = fmaf(__half2float(w1[r * C + c]), m[c], a);
        hid[t][r] = a > 0.f ? a : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T * C; i += 256) {
        const int t = i / C, c = i % C;
        float a = __half2float(b2[c]);
        for (int r = 0; r < Cr; ++r) a = fmaf(__half2float(w2[c * Cr + r]), hid[t][r], a);
        g[t][c] = 1.0f / (1.0f + __expf(-a));
    }
    __syncthreads();
    const int C8 = C / 8;
    const size_t nvec = (size_t)HW * C8, base = (size_t)n * nvec;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % C8) * 8;
        float acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int t = 0; t < T; ++t) {
            h8 v = reinterpret_cast<const h8*>(in.x[t])[base + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = fmaf((float)v[k], g[t][c0 + k], acc[k]);
        }
        h8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (_Float16)acc[k];
        reinterpret_cast<h8*>(out)[base + i] = o;
    }
}

// Tail of an OSNet block and the 1x1 convolution that follows it, in one pass over the pixels:
//   x2  = sum_t y_t * gate_t                 (k_gate_apply: gates from the band sums psum, fmaf in t order, rounded to half)
//   o   = relu(conv3(x2) + b3 + idn)         (k_pw: MID -> C2, shortcut before the activation)         -> out (optional)
//   o2  = relu(conv4(o) + b4)                (k_pw: C2 -> N2: the next block's conv1, or the stage's ConvBR) -> out2
//   out2 = avgpool2x2(o2)                    (k_avgpool2, optional)
// The unfused chain moves x2, o (twice or three times) and o2 through HBM: per 512 crops at stage 1 that is 1.0 GB for
// the block before a transition and 0.63 GB for the block before another block; here 0.30 / 0.43 GB.  Workgroup = 128
// consecutive pixels of one image (4 waves x 2 tiles of 16); o goes through a per-wave LDS tile both to vectorise the
// shortcut read / output write and to re-enter the matrix cores as the B operand of the second product.  Operand
// placement, accumulation order and rounding points are those of the separate kernels: outputs are bit-identical.
// gates of every image: gv[img][t][32] = sigmoid(fc2(relu(fc1(mean_hw(y_t))))) from the band sums (k_gate_apply's arithmetic)
__global__ __launch_bounds__(128) void k_gate_vec(const float* __restrict__ psum, int parts, float scale, const __half* __restrict__ gw1,
                                                 const __half* __restrict__ gb1, const __half* __restrict__ gw2,
                                                 const __half* __restrict__ gb2, int Cr, int MID, int Nimg, float* __restrict__ gv,
                                                 const int* __restrict__ nvalid)
{
    if (nvalid && (int)blockIdx.x >= *nvalid) return;
    __shared__ float g[4][32];
    __shared__ float hid[4][16];
    const int tid = threadIdx.x, img = blockIdx.x;
    {
        const int t = tid >> 5, c = tid & 31;
        float a = 0.f;
        if (c < MID) {
            const float* m = psum + (((size_t)t * Nimg + img) * parts) * MID + c;
            for (int p = 0; p < parts; ++p) a += m[(size_t)p * MID];
        }
        g[t][c] = a * scale;
    }
    __syncthreads();
    if (tid < 4 * Cr) {
        const int t = tid / Cr, r = tid - t * Cr;
        float a = __half2float(gb1[r]);
        for (int c = 0; c < MID; ++c) a = fmaf(__half2float(gw1[r * MID + c]), g[t][c], a);
        hid[t][r] = a > 0.f ? a : 0.f;
    }
    __syncthreads();
    {
        const int t = tid >> 5, c = tid & 31;
        float o = 0.f;
        if (c < MID) {
            float a = __half2float(gb2[c]);
            for (int r = 0; r < Cr; ++r) a = fmaf(__half2float(gw2[c * Cr + r]), hid[t][r], a);
            o = 1.0f / (1.0f + __expf(-a));
        }
        gv[((size_t)img * 4 + t) * 32 + c] = o;
    }
}

// C1 > 0: the block's shortcut is its `down` convolution (first block of a stage, C1 -> C2 channels, no activation): instead
// of reading a [pixels][C2] tensor another launch wrote, the tail computes it from the block input `idn` = x [pixels][C1]
// as one more MFMA product (weights wd [C2][C1], bias bd) and applies it in the accumulator layout — the `down` launch
// (56 us at stage 1 / 512 crops) and 0.1 GB of shortcut reads disappear.  Same products, roundings and order as k_pw.
template <int MID, int C2, int N2, int C1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(!SS_TAIL_BUMP ? 1 : MID == 16 ? 4 : MID == 24 ? 3 : 2))) void k_osnet_tail(GatePtrs ys, const float* __restrict__ gv,
                                                   const __half* __restrict__ w3, const __half* __restrict__ b3,
                                                   const __half* __restrict__ idn, const __half* __restrict__ wd,
                                                   const __half* __restrict__ bd, __half* __restrict__ out,
                                                   const __half* __restrict__ w4, const __half* __restrict__ b4,
                                                   __half* __restrict__ out2, int pool, int Nimg, int HW, int W,
                                                   const int* __restrict__ nvalid)
{
    if (nvalid && (long long)blockIdx.x * 128 >= (long long)(*nvalid) * HW) return;
    constexpr int MT = C2 / 16, MT2 = (N2 + 15) / 16, KS2 = C2 / 32, EP = C2 + 8, CG = C2 / 8, CG2 = N2 / 8, K8 = C2 / 8;
    constexpr int IT = C1 > 0 ? 1 : 32 * CG / 64, KS1 = C1 > 0 ? (C1 + 31) / 32 : 1;
    __shared__ __attribute__((aligned(16))) _Float16 Ws4[MT2 * 16 * EP];
    __shared__ __attribute__((aligned(16))) _Float16 Et[128 * EP];
    // the shortcut's weights and the three bias vectors come with the first round trip too: read from global memory where they are
    // used (per 16-channel tile, inside the epilogue loops) each of them was a dependent L2 access in a workgroup that lives ~4 us
    constexpr int WDP = C1 > 0 ? KS1 * 32 + 8 : 8;                            // pitch of a `down` weight row (halfs), zero-padded to whole 32-wide blocks
    __shared__ __attribute__((aligned(16))) _Float16 Wds[C1 > 0 ? C2 * WDP : 8];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[2 * C2 + MT2 * 16];    // b3 | bd | b4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, n = lane & 15;
    const size_t px_wg = (size_t)blockIdx.x * 128, px0 = px_wg + wave * 32;
    const int img = (int)(px_wg / HW);
    const h8 z8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const bool kval = 8 * q < MID;

    // everything this workgroup reads from global memory is requested up front: one memory round trip.  The chain outputs and the
    // shortcut input are read exactly once: non-temporal loads (no L2 allocation) took 189 / 173 / 82 us -> 173 / 162 / 78 us at
    // 1 024 crops
    float4 gq[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float4* gp = reinterpret_cast<const float4*>(gv + ((size_t)img * 4 + t) * 32 + 8 * q);
        gq[t][0] = gp[0]; gq[t][1] = gp[1];
    }
    h8 yv[2][4], a3[MT], rs[IT], xb[2][KS1];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            yv[pt][t] = kval ? __builtin_nontemporal_load(reinterpret_cast<const h8*>(ys.x[t] + (px0 + pt * 16 + n) * MID + 8 * q)) : z8;
    if constexpr (C1 > 0) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int k = ks * 32 + 8 * q;
                xb[pt][ks] = k < C1 ? __builtin_nontemporal_load(reinterpret_cast<const h8*>(idn + (px0 + pt * 16 + n) * C1 + k)) : z8;
            }
    } else {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = it * 64 + lane, row = i / CG, cg = i - row * CG;
            rs[it] = __builtin_nontemporal_load(reinterpret_cast<const h8*>(idn + (px0 + row) * C2 + cg * 8));
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a3[mt] = kval ? *reinterpret_cast<const h8*>(w3 + (size_t)(mt * 16 + n) * MID + 8 * q) : z8;
    // weights and biases -> LDS: every vector of a thread is requested (from a clamped, always valid address) before the first LDS store.
    // (As rolled `load -> store` loops with a predicated load each, these cost one full L2 round trip per 256 vectors - two to eight of
    // them back to back in a workgroup that otherwise lives for one.)
    constexpr int NB = (2 * C2 + MT2 * 16) / 8, NW4 = MT2 * 16 * K8, NWD = C1 > 0 ? C2 * (KS1 * 4) : 0;
    constexpr int IB = (NB + 255) / 256, IW4 = (NW4 + 255) / 256, IWD = (NWD + 255) / 256;
    h8 vb[IB], vw4[IW4], vwd[IWD > 0 ? IWD : 1];
#pragma unroll
    for (int it = 0; it < IB; ++it) {
        const int e = min(tid + 256 * it, NB - 1) * 8;
        const __half* src = e < C2 ? b3 + e : e < 2 * C2 ? (C1 > 0 ? bd + (e - C2) : b3) : b4 + min(e - 2 * C2, N2 - 8);
        vb[it] = *reinterpret_cast<const h8*>(src);
    }
#pragma unroll
    for (int it = 0; it < IWD; ++it) {
        const int i = min(tid + 256 * it, NWD - 1), r = i / (KS1 * 4), c8 = i - r * (KS1 * 4);
        vwd[it] = *reinterpret_cast<const h8*>(wd + (size_t)r * C1 + min(c8 * 8, C1 - 8));
    }
#pragma unroll
    for (int it = 0; it < IW4; ++it) {
        const int i = min(tid + 256 * it, NW4 - 1), r = i / K8, c8 = i - r * K8;
        vw4[it] = *reinterpret_cast<const h8*>(w4 + (size_t)min(r, N2 - 1) * C2 + c8 * 8);
    }
#pragma unroll
    for (int it = 0; it < IB; ++it) {
        const int i = tid + 256 * it, e = i * 8;
        if (i < NB) {
            const bool ok = e < C2 || (e < 2 * C2 ? C1 > 0 : e - 2 * C2 < N2);
            *reinterpret_cast<h8*>(Bs + e) = ok ? vb[it] : z8;
        }
    }
#pragma unroll
    for (int it = 0; it < IWD; ++it) {
        const int i = tid + 256 * it, r = i / (KS1 * 4), c8 = i - r * (KS1 * 4);
        if (i < NWD) *reinterpret_cast<h8*>(Wds + r * WDP + c8 * 8) = c8 * 8 < C1 ? vwd[it] : z8;     // (k >= C1: zeros)
    }
#pragma unroll
    for (int it = 0; it < IW4; ++it) {
        const int i = tid + 256 * it, r = i / K8, c8 = i - r * K8;
        if (i < NW4) *reinterpret_cast<h8*>(Ws4 + r * EP + c8 * 8) = r < N2 ? vw4[it] : z8;
    }
    __syncthreads();

    // x2 as the B operand (lane (q, n): pixel n, channels 8q..8q+7), conv3 on the matrix cores
    f4 acc[MT][2];
    {
        h8 b[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            float s[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
            if (kval) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float gk[8] = { gq[t][0].x, gq[t][0].y, gq[t][0].z, gq[t][0].w, gq[t][1].x, gq[t][1].y, gq[t][1].z, gq[t][1].w };
#pragma unroll
                    for (int k = 0; k < 8; ++k) s[k] = fmaf((float)yv[pt][t][k], gk[k], s[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) b[pt][k] = (_Float16)s[k];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const h4 a0 = { a3[mt][0], a3[mt][1], a3[mt][2], a3[mt][3] }, a1 = { a3[mt][4], a3[mt][5], a3[mt][6], a3[mt][7] };
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const h4 b0 = { b[pt][0], b[pt][1], b[pt][2], b[pt][3] }, b1 = { b[pt][4], b[pt][5], b[pt][6], b[pt][7] };
                f4 d = { 0.f, 0.f, 0.f, 0.f };
                d = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, d, 0, 0, 0);
                acc[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, b1, d, 0, 0, 0);
            }
        }
    }
    _Float16* tile = Et + wave * 32 * EP;
    if constexpr (C1 > 0) {
        // shortcut = down(x): per 16-channel tile one more product, applied in the accumulator layout (lane (q, n): channels
        // mt*16 + 4q .. +3 of pixel n), then the finished tile goes to LDS
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f4 dd[2] = { f4{ 0.f, 0.f, 0.f, 0.f }, f4{ 0.f, 0.f, 0.f, 0.f } };
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int k = ks * 32 + 8 * q;
                const h8 a = *reinterpret_cast<const h8*>(Wds + (mt * 16 + n) * WDP + k);
                const h4 a0 = { a[0], a[1], a[2], a[3] }, a1 = { a[4], a[5], a[6], a[7] };
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const h8 bv = xb[pt][ks];
                    const h4 b0 = { bv[0], bv[1], bv[2], bv[3] }, b1 = { bv[4], bv[5], bv[6], bv[7] };
                    dd[pt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, dd[pt], 0, 0, 0);
                    dd[pt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, b1, dd[pt], 0, 0, 0);
                }
            }
            const h4 bb3 = *reinterpret_cast<const h4*>(Bs + mt * 16 + 4 * q), bbd = *reinterpret_cast<const h4*>(Bs + C2 + mt * 16 + 4 * q);
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                h4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const _Float16 sc = (_Float16)((float)(_Float16)dd[pt][j] + (float)bbd[j]);      // down: conv to half, + bias, to half
                    float f = (float)(_Float16)acc[mt][pt][j] + (float)bb3[j];
                    f += (float)sc;
                    o[j] = (_Float16)(f > 0.f ? f : 0.f);
                }
                *reinterpret_cast<h4*>(tile + (pt * 16 + n) * EP + mt * 16 + 4 * q) = o;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (out) {
#pragma unroll
            for (int it = 0; it < 32 * CG / 64; ++it) {                      // the block output, 16-byte vectors
                const int i = it * 64 + lane, row = i / CG, cg = i - row * CG;
                __builtin_nontemporal_store(*reinterpret_cast<const h8*>(tile + row * EP + cg * 8), reinterpret_cast<h8*>(out + (px0 + row) * C2 + cg * 8));
            }
        }
    } else {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const h4 o = { (_Float16)acc[mt][pt][0], (_Float16)acc[mt][pt][1], (_Float16)acc[mt][pt][2], (_Float16)acc[mt][pt][3] };
                *reinterpret_cast<h4*>(tile + (pt * 16 + n) * EP + mt * 16 + 4 * q) = o;
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 * CG / 64; ++it) {                          // + bias + shortcut, ReLU; 16-byte vectors
            const int i = it * 64 + lane, row = i / CG, cg = i - row * CG;
            const size_t px = px0 + row;
            const h8 v = *reinterpret_cast<const h8*>(tile + row * EP + cg * 8);
            const h8 bb = *reinterpret_cast<const h8*>(Bs + cg * 8);
            const h8 r = rs[it];
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = (float)v[j] + (float)bb[j];
                f += (float)r[j];
                o[j] = (_Float16)(f > 0.f ? f : 0.f);
            }
            if (out) __builtin_nontemporal_store(o, reinterpret_cast<h8*>(out + px * C2 + cg * 8));
            *reinterpret_cast<h8*>(tile + row * EP + cg * 8) = o;
        }
        __builtin_amdgcn_wave_barrier();
    }

    // second product: o (LDS tile) x w4 (LDS)
    f4 acc2[MT2][2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) acc2[mt][pt] = f4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
        h8 bv[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) bv[pt] = *reinterpret_cast<const h8*>(tile + (pt * 16 + n) * EP + ks * 32 + 8 * q);
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
            const h8 a = *reinterpret_cast<const h8*>(Ws4 + (mt * 16 + n) * EP + ks * 32 + 8 * q);
            const h4 a0 = { a[0], a[1], a[2], a[3] }, a1 = { a[4], a[5], a[6], a[7] };
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const h4 b0 = { bv[pt][0], bv[pt][1], bv[pt][2], bv[pt][3] }, b1 = { bv[pt][4], bv[pt][5], bv[pt][6], bv[pt][7] };
                acc2[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, acc2[mt][pt], 0, 0, 0);
                acc2[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, b1, acc2[mt][pt], 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
            if (mt * 16 + 4 * q < N2) {
                const h4 o = { (_Float16)acc2[mt][pt][0], (_Float16)acc2[mt][pt][1], (_Float16)acc2[mt][pt][2], (_Float16)acc2[mt][pt][3] };
                *reinterpret_cast<h4*>(tile + (pt * 16 + n) * EP + mt * 16 + 4 * q) = o;
            }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < (32 * CG2 + 63) / 64; ++it) {
        const int i = it * 64 + lane, row = i / CG2, cg = i - row * CG2;
        if (i >= 32 * CG2) break;
        const size_t px = px0 + row;
        const h8 v = *reinterpret_cast<const h8*>(tile + row * EP + cg * 8);
        const h8 bb = *reinterpret_cast<const h8*>(Bs + 2 * C2 + cg * 8);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (float)v[j] + (float)bb[j];
            o[j] = (_Float16)(f > 0.f ? f : 0.f);
        }
        if (!pool) *reinterpret_cast<h8*>(out2 + px * N2 + cg * 8) = o;
        else *reinterpret_cast<h8*>(tile + row * EP + cg * 8) = o;
    }
    if (!pool) return;
    __syncthreads();
    {                                                                        // 2x2 average of the workgroup's 128 / W rows
        const int OWp = W / 2, r0 = (int)((px_wg - (size_t)img * HW) / W);
        for (int i = tid; i < 32 * CG2; i += 256) {
            const int pp = i / CG2, cg = i - pp * CG2, pr = pp / OWp, pc = pp - pr * OWp, l00 = 2 * pr * W + 2 * pc;
            const h8 a = *reinterpret_cast<const h8*>(Et + (size_t)l00 * EP + cg * 8);
            const h8 b = *reinterpret_cast<const h8*>(Et + (size_t)(l00 + 1) * EP + cg * 8);
            const h8 c = *reinterpret_cast<const h8*>(Et + (size_t)(l00 + W) * EP + cg * 8);
            const h8 d = *reinterpret_cast<const h8*>(Et + (size_t)(l00 + W + 1) * EP + cg * 8);
            h8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (_Float16)(((((float)a[k] + (float)b[k]) + (float)c[k]) + (float)d[k]) / 4.0f);
            const size_t op = (size_t)img * (HW / 4) + (size_t)(r0 / 2 + pr) * OWp + pc;
            *reinterpret_cast<h8*>(out2 + op * N2 + cg * 8) = o;
        }
    }
}

// ---- C2PSA attention (v11: 2-8 heads over the H*W positions of the stride-32 map, key_dim 32, head_dim 64) in one launch ----
// qkv: the fused 1x1's output [B][N][heads * 128] half, per head [q 32 | k 32 | v 64] (nets.Attention); out[b][i][h*64 + c] =
// sum_j v[c][j] * softmax_j(scale * q_i . k_j) (+ pe[b][i][h*64 + c], the depthwise positional term, when given).
// Workgroup = (64 queries, head, image), wave = 16 queries.  Everything stays in the MFMA layouts:
//   * S^T = K Q^T per 16-key tile: A = 16 keys x 32 dims, B = 16 queries x 32 dims, both 16-byte loads straight from qkv;
//     D: lane (q, n) holds keys 16t + 4q .. + 3 of query n — the whole score row of a query lives in the 4 lanes n, n + 16, ..
//     (softmax: two shuffles for the max, two for the sum);
//   * O^T = V^T P^T: two key tiles' probabilities (rounded to half, 2 + 2 registers) ARE the B operand of v_mfma_f32_16x16x32_f16
//     when its k-slots 8q .. 8q + 7 are read as (tile a: keys 4q .. 4q + 3, tile b: the same) — the depthwise-MFMA trick of
//     DwDiag — and the A operand is two 8-byte LDS reads of V^T (staged once per workgroup as [64][N + pad], conflict-free pitch);
//   * O^T's D layout is 4 channels of a query per lane: 8-byte stores into the NHWC output.
// fp32 scores, softmax and accumulation; N <= 256 positions (640 x 384 input: 240).
#define PSA_MAXT 16
__global__ __launch_bounds__(256) void k_psa_attn(const __half* __restrict__ qkv, const __half* __restrict__ pe, __half* __restrict__ out,
                                                 int N, int heads, float scale)
{
    extern __shared__ __attribute__((aligned(16))) char psa_smem[];
    _Float16* VT = reinterpret_cast<_Float16*>(psa_smem);               // [64][VP]
    const int NP = (N + 31) & ~31, VP = NP + 16, NT = NP / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, n = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z, CT = heads * 128, C = heads * 64;
    const __half* base = qkv + (size_t)b * N * CT + h * 128;
    // V^T -> LDS (zeros for the padded positions)
    for (int i = tid; i < NP * 8; i += 256) {
        const int j = i >> 3, c8 = i & 7;
        h8 v = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (j < N) v = *reinterpret_cast<const h8*>(base + (size_t)j * CT + 64 + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) VT[(c8 * 8 + e) * VP + j] = v[e];
    }
    __syncthreads();
    const int i0 = (blockIdx.x * 4 + wave) * 16;
    if (i0 >= N) return;
    const int iq = i0 + n < N ? i0 + n : N - 1;
    const h8 z8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const h8 bq = *reinterpret_cast<const h8*>(base + (size_t)iq * CT + 8 * q);
    f4 S[PSA_MAXT];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < PSA_MAXT; ++t) {
        S[t] = f4{ -INFINITY, -INFINITY, -INFINITY, -INFINITY };
        if (t < NT) {
            const int j = 16 * t + n;
            const h8 ak = j < N ? *reinterpret_cast<const h8*>(base + (size_t)j * CT + 32 + 8 * q) : z8;
            const f4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak, bq, f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                S[t][e] = 16 * t + 4 * q + e < N ? d[e] * scale : -INFINITY;
                m = fmaxf(m, S[t][e]);
            }
        }
    }
    m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < PSA_MAXT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { S[t][e] = __expf(S[t][e] - m); sum += S[t][e]; }     // exp(-inf) = 0 for the masked keys
    sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    uint2 P[PSA_MAXT];
#pragma unroll
    for (int t = 0; t < PSA_MAXT; ++t) {
        const h4 p = { (_Float16)(S[t][0] * inv), (_Float16)(S[t][1] * inv), (_Float16)(S[t][2] * inv), (_Float16)(S[t][3] * inv) };
        P[t] = __builtin_bit_cast(uint2, p);
    }
    f4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int tp = 0; tp < PSA_MAXT / 2; ++tp) {
        if (2 * tp >= NT) break;
        const uint4 bp = { P[2 * tp].x, P[2 * tp].y, P[2 * tp + 1].x, P[2 * tp + 1].y };
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const _Float16* vr = VT + (ct * 16 + n) * VP + 32 * tp + 4 * q;
            const uint2 a0 = *reinterpret_cast<const uint2*>(vr), a1 = *reinterpret_cast<const uint2*>(vr + 16);
            const uint4 av = { a0.x, a0.y, a1.x, a1.y };
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, av), __builtin_bit_cast(h8, bp), acc[ct], 0, 0, 0);
        }
    }
    if (i0 + n < N) {
        const size_t o = ((size_t)b * N + i0 + n) * C + h * 64 + 4 * q;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            f4 v = acc[ct];
            if (pe) {
                const h4 pv = *reinterpret_cast<const h4*>(pe + o + ct * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (float)(_Float16)v[e] + (float)pv[e];     // the product rounded to half, then the sum (as the two torch ops)
            }
            const h4 r = { (_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3] };
            *reinterpret_cast<h4*>(out + o + ct * 16) = r;
        }
    }
}

// OSNet's head: global average over the HW positions of the last feature map, the 128 -> F fully connected layer, + bias, ReLU, in one
// launch (torch: a reduction kernel, a library GEMM and two element-wise passes, ~33 us per 1 024 crops).  Workgroup = 4 crops: all their
// pixels are requested at once (thread = 8 channels x every 16th pixel), the means (rounded to half, as the tensor the GEMM would
// read) go to LDS, every thread owns F / 256 output features of the 4 crops and walks its weight rows once (16-byte loads; fp32
// accumulation in k order).  x [N][HW][C] half, w [F][C] half, out [N][F] half.
template <int C>
__global__ __launch_bounds__(256) void k_osnet_head(const __half* __restrict__ x, const __half* __restrict__ w, const __half* __restrict__ bias,
                                                   __half* __restrict__ out, int N, int HW, int F, const int* __restrict__ nvalid)
{
    constexpr int CR = 4, C8 = C / 8, PG = 256 / C8;
    __shared__ float part[CR][PG][C];                                  // 16 pixel groups x C partial sums per crop (32 KiB)
    __shared__ __attribute__((aligned(16))) _Float16 mean[CR][C];
    const int tid = threadIdx.x, n0 = blockIdx.x * CR;
    const int nv = nvalid ? min(*nvalid, N) : N;
    if (n0 >= nv) return;
    const int c8 = tid % C8, pg = tid / C8;
    float a[CR][8];
#pragma unroll
    for (int cr = 0; cr < CR; ++cr)
#pragma unroll
        for (int k = 0; k < 8; ++k) a[cr][k] = 0.f;
    for (int p = pg; p < HW; p += PG) {
        h8 v[CR];
#pragma unroll
        for (int cr = 0; cr < CR; ++cr)
            v[cr] = n0 + cr < nv ? *reinterpret_cast<const h8*>(x + (((size_t)(n0 + cr) * HW + p) * C + c8 * 8)) : h8{ 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
        for (int cr = 0; cr < CR; ++cr)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[cr][k] += (float)v[cr][k];
    }
#pragma unroll
    for (int cr = 0; cr < CR; ++cr)
#pragma unroll
        for (int k = 0; k < 8; ++k) part[cr][pg][c8 * 8 + k] = a[cr][k];
    __syncthreads();
    for (int i = tid; i < CR * C; i += 256) {
        const int cr = i / C, c = i - cr * C;
        float t = 0.f;
        for (int g = 0; g < PG; ++g) t += part[cr][g][c];
        mean[cr][c] = (_Float16)(t / (float)HW);
    }
    __syncthreads();
    for (int o = tid; o < F; o += 256) {
        float acc[CR];
#pragma unroll
        for (int cr = 0; cr < CR; ++cr) acc[cr] = 0.f;
        const h8* wr = reinterpret_cast<const h8*>(w + (size_t)o * C);
#pragma unroll 4
        for (int k8 = 0; k8 < C8; ++k8) {
            const h8 wv = wr[k8];
#pragma unroll
            for (int cr = 0; cr < CR; ++cr) {
                const h8 mv = *reinterpret_cast<const h8*>(&mean[cr][k8 * 8]);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[cr] = fmaf((float)wv[k], (float)mv[k], acc[cr]);
            }
        }
        const float b = __half2float(bias[o]);
#pragma unroll
        for (int cr = 0; cr < CR; ++cr)
            if (n0 + cr < nv) {
                const float f = (float)(_Float16)acc[cr] + b;          // the product rounded to half, then the bias (as the GEMM + bias epilogue)
                out[(size_t)(n0 + cr) * F + o] = __float2half(f > 0.f ? f : 0.f);
            }
    }
}

static inline int grid_for(size_t n, int block) { size_t g = (n + block - 1) / block; return (int)(g > 4096 ? 4096 : (g ? g : 1)); }

extern "C" int ss_op_bias_act_f16(void* stream, void* x, const void* bias, const void* res, long long n_pix, int C, int act)
{
    if (!x || !bias || C < 1) return SS_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (C % 8 == 0) {
        size_t nv = (size_t)n_pix * (C / 8);
        hipLaunchKernelGGL(k_bias_act8, dim3(grid_for(nv, 256)), dim3(256), 0, st, (__half*)x, (const __half*)bias, (const __half*)res, nv, C / 8, act);
    } else {
        size_t n = (size_t)n_pix * C;
        hipLaunchKernelGGL(k_bias_act1, dim3(grid_for(n, 256)), dim3(256), 0, st, (__half*)x, (const __half*)bias, (const __half*)res, n, C, act);
    }
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_dwconv3x3_f16(void* stream, const void* x, const void* w9, const void* bias, void* y, int N, int H, int W, int C, int act)
{
    if (!x || !w9 || !bias || !y || C % 8) return SS_ERR_INVALID;
    size_t total = (size_t)N * H * W * (C / 8);
    hipLaunchKernelGGL(k_dw3x3, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const __half*)x, (const __half*)w9,
                       (const __half*)bias, (__half*)y, N, H, W, C / 8, act);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_bias_act_place_f16(void* stream, const void* x, const void* bias, const void* res, long long n_pix, int C,
                                        int act, int res_after, void* out, int out_ld, void* out2, int c0, int cn)
{
    if (!x || !bias || !out || C < 8 || C % 8 || out_ld % 8 || out_ld < C || c0 % 8 || cn % 8 || (out2 && (cn < 8 || c0 + cn > C)))
        return SS_ERR_INVALID;
    const size_t nv = (size_t)n_pix * (C / 8);
    hipLaunchKernelGGL(k_bias_act_place, dim3(grid_for(nv, 256)), dim3(256), 0, (hipStream_t)stream, (const __half*)x,
                       (const __half*)bias, (const __half*)res, nv, C / 8, act, res_after, (__half*)out, out_ld / 8,
                       (__half*)out2, c0 / 8, cn / 8);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_v8_decode_ext_f16(void* stream, const void* const* box, const void* const* cls, const void* const* box_bias,
                                       const void* const* cls_bias, const void* const* ext, int n_ext, int ext_ld, int ext_mode,
                                       const int* H, const int* W, const int* strides, int B, int nc, int cls_ld, float* pred)
{
    if (!box || !cls || !box_bias || !cls_bias || !H || !W || !strides || !pred || B < 1 || B > 65535 || nc < 1 || cls_ld < nc) return SS_ERR_INVALID;
    if (n_ext < 0 || (n_ext && (!ext || ext_ld < n_ext || (ext_mode != 0 && ext_mode != 1) || (ext_mode == 1 && n_ext % 3)))) return SS_ERR_INVALID;
    V8Levels L;
    int A = 0;
    L.n_ext = n_ext; L.ext_ld = ext_ld; L.ext_mode = ext_mode; L.cls_ld = cls_ld;
    for (int l = 0; l < 3; ++l) {
        L.box[l] = (const __half*)box[l]; L.cls[l] = (const __half*)cls[l];
        L.box_bias[l] = (const __half*)box_bias[l]; L.cls_bias[l] = (const __half*)cls_bias[l];
        L.ext[l] = n_ext ? (const __half*)ext[l] : nullptr;
        L.H[l] = H[l]; L.W[l] = W[l]; L.stride[l] = strides[l];
        A += H[l] * W[l];
    }
    hipLaunchKernelGGL(k_v8_decode, dim3((A + 127) / 128, B), dim3(128), 0, (hipStream_t)stream, L, B, nc, A, pred);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_v8_decode_f16(void* stream, const void* const* box, const void* const* cls, const void* const* box_bias,
                                   const void* const* cls_bias, const int* H, const int* W, const int* strides, int B, int nc,
                                   float* pred)
{
    return ss_op_v8_decode_ext_f16(stream, box, cls, box_bias, cls_bias, nullptr, 0, 0, 0, H, W, strides, B, nc, nc, pred);
}

static int launch_pw(hipStream_t st, bool conv3, const void* x, const void* w, const void* bias, const void* res, long long M, int K,
                     int N, int act, int res_after, void* out, int out_ld, void* out2, int c0, int cn, ConvGeom g)
{
    // vector epilogue needs 16-byte aligned rows and slices; SS_PW_EPILOGUE=0 forces the 8-byte form (A/B switch)
    const bool vec_allowed = g_opt_pw_epilogue != 0;
    const bool vec = vec_allowed && out_ld % 8 == 0 && c0 % 8 == 0 && cn % 8 == 0 && ((uintptr_t)out % 16) == 0 &&
                     (!out2 || ((uintptr_t)out2 % 16) == 0) && (!res || ((uintptr_t)res % 16) == 0);
    const int nvb = conv3 ? 0 : nv_batch((void*)st);
    const int* nvp = (nvb > 0 && M % nvb == 0) ? nv_for((void*)st, nvb) : nullptr;
    const bool nv = nvp != nullptr;
    const PwArgs A{ (const __half*)x, (const __half*)w, (const __half*)bias, (const __half*)res, (int)M, K, N, act, res_after,
                    (__half*)out, out_ld, (__half*)out2, c0, cn, g, nvp, nv ? (int)(M / nvb) : 0 };
#define SS_PW(BN, PT, CV, VE)                                                                                           \
    hipLaunchKernelGGL((k_pw<BN, PT, CV, VE>), dim3((unsigned)((M + 64 * PT - 1) / (64 * PT)), (N + BN - 1) / BN), dim3(256), 0, st, A)
#define SS_PW2(BN, PT)                                                                                                  \
    do {                                                                                                                \
        if (conv3) { if (vec) SS_PW(BN, PT, true, true); else SS_PW(BN, PT, true, false); }                             \
        else { if (vec) SS_PW(BN, PT, false, true); else SS_PW(BN, PT, false, false); }                                 \
    } while (0)
    // too few pixels for 64-pixel workgroups to fill 256 CUs and a long K walk (3x3 layers of the stride-32 level at batch
    // 16, every 3x3 layer at batch 1): 16-pixel workgroups, K split over the waves.  Measured at batch 16: M = 3840,
    // K = 1152..2304: 17.5 -> 14, 31 -> 15 us; M = 15360 (960 workgroups re-reading the weights from L2): 11 -> 17 us, so
    // those stay on k_pw.  (Needs the 16-byte epilogue; SS_PW_SPLITK=0: A/B switch)
    const bool splitk_allowed = g_opt_pw_splitk != 0;
    if (splitk_allowed && vec && conv3 && K >= 512 && M <= 4096) {
        const dim3 grid((unsigned)((M + 15) / 16), (N + ((N <= 32) ? 31 : 63)) / ((N <= 32) ? 32 : 64));
#define SS_SK(BN, CV) hipLaunchKernelGGL((k_pw_splitk<BN, CV>), grid, dim3(256), 0, st, A)
        if (N <= 32) { if (conv3) SS_SK(32, true); else SS_SK(32, false); }
        else { if (conv3) SS_SK(64, true); else SS_SK(64, false); }
#undef SS_SK
        return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
    }
    const bool big = M >= 32768;                 // enough pixels to fill the chip with 128-pixel workgroups
    if (N <= 32) { if (big) SS_PW2(32, 2); else SS_PW2(32, 1); }
    else if (N == 80) { if (big) SS_PW2(80, 2); else SS_PW2(80, 1); }     // the detector's class branch: no padded channel tiles
    else if (N <= 64 || !big) { if (big) SS_PW2(64, 2); else SS_PW2(64, 1); }
    else SS_PW2(128, 2);
#undef SS_PW2
#undef SS_PW
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_pointwise_f16(void* stream, const void* x, const void* w, const void* bias, const void* res, long long M,
                                   int K, int N, int act, int res_after, void* out, int out_ld, void* out2, int c0, int cn)
{
    if (!x || !w || !bias || !out || M < 1 || M > 0x7fffffffLL || K < 8 || K % 8 || N < 8 || N % 8 || out_ld % 4 || out_ld < N ||
        c0 % 4 || cn % 4 || (out2 && (cn < 4 || c0 + cn > N)))
        return SS_ERR_INVALID;
    return launch_pw((hipStream_t)stream, false, x, w, bias, res, M, K, N, act, res_after, out, out_ld, out2, c0, cn, ConvGeom{});
}

extern "C" int ss_op_conv3x3_f16(void* stream, const void* x, const void* w, const void* bias, const void* res, int B, int H, int W,
                                 int Cin, int N, int conv_stride, int act, int res_after, void* out, int out_ld, void* out2, int c0,
                                 int cn)
{
    if (!x || !w || !bias || !out || B < 1 || H < 1 || W < 1 || Cin < 8 || Cin % 8 || N < 8 || N % 8 || (conv_stride != 1 && conv_stride != 2) ||
        out_ld % 4 || out_ld < N || c0 % 4 || cn % 4 || (out2 && (cn < 4 || c0 + cn > N)))
        return SS_ERR_INVALID;
    ConvGeom g{ H, W, Cin, (H - 1) / conv_stride + 1, (W - 1) / conv_stride + 1, conv_stride };
    const long long M = (long long)B * g.OH * g.OW;
    if (M > 0x7fffffffLL) return SS_ERR_INVALID;
    return launch_pw((hipStream_t)stream, true, x, w, bias, res, M, 9 * Cin, N, act, res_after, out, out_ld, out2, c0, cn, g);
}

extern "C" int ss_op_osnet_stem_f16(void* stream, const void* x, const void* w_prep, const void* bias, void* y, int N, int H, int W,
                                    const void* w1, const void* b1, void* y1)
{
    if (!x || !w_prep || !bias || !y || N < 1 || W != STEM_W || H < 16 || H % 16 || (w1 && (!b1 || !y1))) return SS_ERR_INVALID;
    const int tiles = H / 16;                                   // 4 pooled rows = 16 input rows per tile
    hipLaunchKernelGGL(k_osnet_stem, dim3((unsigned)((size_t)N * tiles)), dim3(256), 0, (hipStream_t)stream, (const __half*)x,
                       (const __half*)w_prep, (const __half*)bias, (__half*)y, H, tiles, nv_for(stream, N),
                       (const __half*)w1, (const __half*)b1, (__half*)y1);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_lightconv_f16(void* stream, const void* x, const void* w1, const void* w9, const void* bias, void* y,
                                   int N, int H, int W, int C)
{
    if (!x || !w1 || !w9 || !bias || !y || N < 1 || H < 1 || W < 8 || W % 8) return SS_ERR_INVALID;
    const size_t lds = (size_t)(LC_TH + 2) * (W + 2) * (((C + 15) / 16) * 16) * 2;   // pixel stride of the pointwise buffer: lc_slot
    if (lds > 65536) return SS_ERR_INVALID;
    const int bands = (H + LC_TH - 1) / LC_TH;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((size_t)N * bands)), block(256);
#define SS_LC(CC) hipLaunchKernelGGL(k_lightconv<CC>, grid, block, lds, st, (const __half*)x, (const __half*)w1, \
                                     (const __half*)w9, (const __half*)bias, (__half*)y, H, W, bands)
    if (C == 16) SS_LC(16);
    else if (C == 24) SS_LC(24);
    else if (C == 32) SS_LC(32);
    else return SS_ERR_INVALID;
#undef SS_LC
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_maxpool_f16(void* stream, const void* x, void* y, int N, int H, int W, int C, int k, int stride, int pad)
{
    if (!x || !y || C % 8 || k < 1 || stride < 1) return SS_ERR_INVALID;
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    size_t total = (size_t)N * OH * OW * (C / 8);
    hipLaunchKernelGGL(k_maxpool, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const __half*)x, (__half*)y,
                       N, H, W, C / 8, k, stride, pad, OH, OW);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_gate_sum_f16(void* stream, const void* const* xs, int T, const void* w1, const void* b1, const void* w2,
                                  const void* b2, float* means_ws, void* out, int N, int HW, int C, int Cr)
{
    if (T < 1 || T > 4 || C % 8 || C > 256 || Cr < 1 || Cr > 16) return SS_ERR_INVALID;
    GatePtrs p;
    for (int t = 0; t < 4; ++t) p.x[t] = (const __half*)xs[t < T ? t : 0];
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_gate_mean, dim3(N, T), dim3(256), 0, st, p, means_ws, N, HW, C);
    const size_t nvec = (size_t)HW * (C / 8);
    hipLaunchKernelGGL(k_gate_apply, dim3(grid_for(nvec, 256) > 64 ? 64 : grid_for(nvec, 256), N), dim3(256), 0, st, p, T, means_ws, 1, 1.0f,
                       (const __half*)w1, (const __half*)b1, (const __half*)w2, (const __half*)b2, (__half*)out, N, HW, C, Cr);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

// register-resident row stream for 16- and 32-wide images (SS_OSNET_CHAINS=0: the LDS form, A/B switch).  A wave of the
// stream form runs its band's rows one after the other (~4 us per row of 5 layers), so it needs enough images to fill the
// chip with waves: below 96 images the LDS form (256 threads per band and chain) is the faster one.
static bool os_chain_form(int N, int W, int C)
{
    const bool chains = g_opt_osnet_chains != 0;
    return chains && N >= 96 && ((W == 32 && C == 16) || ((W == 16 || W == 8) && (C == 16 || C == 24 || C == 32)));
}
// band height of the stream form: as many bands as give one round of waves (32-wide: 3 waves per SIMD = 3072, 16-wide:
// 2048; two chain groups per band), at least 8 rows per band
static int os_band_rows(int N, int H, int W, int C)
{
    if (!os_chain_form(N, W, C)) return LC_TH;
    const int target = W == 32 ? 3072 : 2048;
    const int units = W == 8 ? (N + 1) / 2 : N;             // 8-wide maps: a wave takes two images
    int bands = target / (2 * units);
    if (bands < 1) bands = 1;
    if (bands > (H + 7) / 8) bands = (H + 7) / 8;
    return (H + bands - 1) / bands;
}

extern "C" int ss_op_osnet_streams_bands(int N, int H, int W, int C)
{
    if (H < 1 || N < 1) return SS_ERR_INVALID;
    const int th = os_band_rows(N, H, W, C);
    return (H + th - 1) / th;
}

extern "C" long long ss_op_dwtab_bytes(int layers, int C)
{
    if (layers < 1 || (C != 16 && C != 24 && C != 32)) return SS_ERR_INVALID;
    return (long long)layers * (C == 16 ? DwTab<16>::LAYER : C == 24 ? DwTab<24>::LAYER : DwTab<32>::LAYER);
}

extern "C" int ss_op_dwtab_f16(void* stream, const void* w9, const void* bias, int layers, int C, void* out)
{
    if (!w9 || !bias || !out || ss_op_dwtab_bytes(layers, C) < 0) return SS_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (C == 16) hipLaunchKernelGGL(k_dwtab<16>, dim3(1), dim3(256), 0, st, (const __half*)w9, (const __half*)bias, layers, (char*)out);
    else if (C == 24) hipLaunchKernelGGL(k_dwtab<24>, dim3(1), dim3(256), 0, st, (const __half*)w9, (const __half*)bias, layers, (char*)out);
    else hipLaunchKernelGGL(k_dwtab<32>, dim3(1), dim3(256), 0, st, (const __half*)w9, (const __half*)bias, layers, (char*)out);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_osnet_streams_f16(void* stream, const void* x, const void* w1, const void* dwtab,
                                       void* const* ys, float* psum, int N, int H, int W, int C)
{
    if (!x || !w1 || !dwtab || !ys || !psum || N < 1 || H < 1 || W < 8 || W % 8) return SS_ERR_INVALID;
    const int TH = os_band_rows(N, H, W, C), bands = (H + TH - 1) / TH;
    StreamOut o;
    for (int t = 0; t < 4; ++t) { if (!ys[t]) return SS_ERR_INVALID; o.y[t] = (__half*)ys[t]; }
    hipStream_t st = (hipStream_t)stream;
    const int* nv = nv_for(stream, N);
    if (os_chain_form(N, W, C)) {
        // one wave per (image, band, chain group); groups {4,1} and {3,2}: five layers each
        const unsigned masks = 0x69u;
        const size_t units = W == 8 ? ((size_t)N + 1) / 2 : (size_t)N;
        dim3 grid((unsigned)((units * bands + 3) / 4), 2), block(256);
#define SS_CHN(CC, NT, PR) hipLaunchKernelGGL((k_osnet_chains<CC, NT, PR>), grid, block, 0, st, (const __half*)x, (const __half*)w1, \
                                              (const char*)dwtab, o, psum, N, H, TH, bands, masks, nv)
        if (W == 32) SS_CHN(16, 2, false);                   // (wider channel counts at 32 columns exceed 256 VGPRs: LDS form)
        else if (W == 16) { if (C == 16) SS_CHN(16, 1, false); else if (C == 24) SS_CHN(24, 1, false); else SS_CHN(32, 1, false); }
        else { if (C == 16) SS_CHN(16, 1, true); else if (C == 24) SS_CHN(24, 1, true); else SS_CHN(32, 1, true); }
#undef SS_CHN
        return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
    }
    if (C != 16 && C != 24 && C != 32) return SS_ERR_INVALID;
    const int PS = ((C + 15) / 16) * 16;                     // pixel stride of the pointwise buffer (lc_slot)
    const size_t lds = (size_t)(LC_TH + 2 * OS_TMAX) * (size_t)(2 * W + 2) * PS * 2;
    if (lds > 65536) return SS_ERR_INVALID;
    dim3 grid((unsigned)((size_t)N * bands), 4), block(256);
#define SS_OS(CC) hipLaunchKernelGGL(k_osnet_streams<CC>, grid, block, lds, st, (const __half*)x, (const __half*)w1, \
                                     (const char*)dwtab, o, psum, N, H, W, bands, nv)
    if (C == 16) SS_OS(16);
    else if (C == 24) SS_OS(24);
    else SS_OS(32);
#undef SS_OS
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

// aggregation gate from precomputed partial channel sums (ss_op_osnet_streams_f16's psum): parts per (stream, image)
extern "C" int ss_op_gate_apply_f16(void* stream, const void* const* xs, int T, const void* w1, const void* b1, const void* w2,
                                    const void* b2, const float* sums, int parts, float scale, void* out, int N, int HW, int C,
                                    int Cr)
{
    if (T < 1 || T > 4 || C % 8 || C > 256 || Cr < 1 || Cr > 16 || parts < 1 || !sums || !out) return SS_ERR_INVALID;
    GatePtrs p;
    for (int t = 0; t < 4; ++t) p.x[t] = (const __half*)xs[t < T ? t : 0];
    const size_t nvec = (size_t)HW * (C / 8);
    hipLaunchKernelGGL(k_gate_apply, dim3(grid_for(nvec, 256) > 64 ? 64 : grid_for(nvec, 256), N), dim3(256), 0, (hipStream_t)stream,
                       p, T, sums, parts, scale, (const __half*)w1, (const __half*)b1, (const __half*)w2, (const __half*)b2,
                       (__half*)out, N, HW, C, Cr);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_avgpool2_f16(void* stream, const void* x, void* y, int N, int H, int W, int C)
{
    if (!x || !y || C % 8 || H < 2 || W < 2 || H % 2 || W % 2) return SS_ERR_INVALID;
    size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(k_avgpool2, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const __half*)x, (__half*)y, N, H, W, C / 8);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_osnet_tail_f16(void* stream, const void* const* ys, const float* psum, int parts, float scale, const void* gw1,
                                    const void* gb1, const void* gw2, const void* gb2, int Cr, float* gates_ws, const void* w3,
                                    const void* b3, const void* idn, int C1, const void* wd, const void* bd, void* out, const void* w4,
                                    const void* b4, void* out2, int pool, int N, int H, int W, int MID, int C2, int N2)
{
    if (!ys || !psum || !gw1 || !gb1 || !gw2 || !gb2 || !gates_ws || !w3 || !b3 || !idn || !w4 || !b4 || !out2 || parts < 1 || Cr < 1 ||
        Cr > 16 || N < 1 || H < 1 || W < 2 || (H * W) % 128 || 128 % W || (pool && ((128 / W) % 2 || W % 2)) || MID > 32 ||
        (C1 > 0 && (!wd || !bd)))
        return SS_ERR_INVALID;
    GatePtrs p;
    for (int t = 0; t < 4; ++t) { if (!ys[t]) return SS_ERR_INVALID; p.x[t] = (const __half*)ys[t]; }
    const dim3 grid((unsigned)((size_t)N * H * W / 128)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int* nv = nv_for(stream, N);
    hipLaunchKernelGGL(k_gate_vec, dim3(N), dim3(128), 0, st, psum, parts, scale, (const __half*)gw1, (const __half*)gb1,
                       (const __half*)gw2, (const __half*)gb2, Cr, MID, N, gates_ws, nv);
#define SS_TAIL(A, B, CC, DD)                                                                                                   \
    hipLaunchKernelGGL((k_osnet_tail<A, B, CC, DD>), grid, block, 0, st, p, (const float*)gates_ws, (const __half*)w3,          \
                       (const __half*)b3, (const __half*)idn, (const __half*)wd, (const __half*)bd, (__half*)out, (const __half*)w4, \
                       (const __half*)b4, (__half*)out2, pool, N, H * W, W, nv)
    if (C1 == 0) {
        if (MID == 16 && C2 == 64 && N2 == 16) SS_TAIL(16, 64, 16, 0);
        else if (MID == 16 && C2 == 64 && N2 == 64) SS_TAIL(16, 64, 64, 0);
        else if (MID == 24 && C2 == 96 && N2 == 24) SS_TAIL(24, 96, 24, 0);
        else if (MID == 24 && C2 == 96 && N2 == 96) SS_TAIL(24, 96, 96, 0);
        else if (MID == 32 && C2 == 128 && N2 == 32) SS_TAIL(32, 128, 32, 0);
        else if (MID == 32 && C2 == 128 && N2 == 128) SS_TAIL(32, 128, 128, 0);
        else return SS_ERR_INVALID;
    } else {                                                     // first block of a stage: shortcut = down(x), C1 -> C2
        if (MID == 16 && C2 == 64 && N2 == 16 && C1 == 16) SS_TAIL(16, 64, 16, 16);
        else if (MID == 24 && C2 == 96 && N2 == 24 && C1 == 64) SS_TAIL(24, 96, 24, 64);
        else if (MID == 32 && C2 == 128 && N2 == 32 && C1 == 96) SS_TAIL(32, 128, 32, 96);
        else return SS_ERR_INVALID;
    }
#undef SS_TAIL
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

// Independent convolutions (all 3x3 / pad 1 or all 1x1, stride 1|2, bias + activation, dense outputs) in one launch.
extern "C" int ss_op_conv_group_f16(void* stream, int n, const ss_conv_desc* d)
{
    if (!d || n < 1 || n > PW_GROUP_MAX) return SS_ERR_INVALID;
    const bool splitk_allowed = g_opt_pw_splitk != 0;
    const bool conv3 = d[0].ksize == 3;
    int nmax = 0, order[PW_GROUP_MAX];
    long long cost[PW_GROUP_MAX];
    PwGroup G;
    for (int i = 0; i < n; ++i) {
        const ss_conv_desc& c = d[i];
        if (!c.x || !c.w || !c.bias || !c.out || (c.ksize != 1 && c.ksize != 3) || (c.ksize == 3) != conv3 || c.B < 1 || c.H < 1 || c.W < 1 ||
            c.Cin < 8 || c.Cin % 8 || c.N < 8 || c.N % 8 || c.N > 80 || (c.stride != 1 && c.stride != 2) || (c.ksize == 1 && c.stride != 1) ||
            ((uintptr_t)c.out % 16) || ((uintptr_t)c.x % 16) || ((uintptr_t)c.w % 16))
            return SS_ERR_INVALID;
        if (c.N > nmax) nmax = c.N;
        order[i] = i;
        cost[i] = (long long)c.ksize * c.ksize * c.Cin;               // serial K walk of a workgroup: longest first
    }
    for (int i = 1; i < n; ++i)                                        // insertion sort, stable
        for (int j = i; j > 0 && cost[order[j]] > cost[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    int wgs = 0;
    for (int s = 0; s < n; ++s) {
        const ss_conv_desc& c = d[order[s]];
        ConvGeom g{ c.H, c.W, c.Cin, (c.H - 1) / c.stride + 1, (c.W - 1) / c.stride + 1, c.stride };
        const long long M = (long long)c.B * g.OH * g.OW;
        if (M > 0x7fffffffLL) return SS_ERR_INVALID;
        const int K = c.ksize * c.ksize * c.Cin;
        G.p[s] = PwArgs{ (const __half*)c.x, (const __half*)c.w, (const __half*)c.bias, nullptr, (int)M, K, c.N, c.act, 0, (__half*)c.out,
                         c.N, nullptr, 0, 0, g, nullptr, 0 };
        G.form[s] = (splitk_allowed && conv3 && K >= 512 && M <= 4096) ? 1 : 0;
        G.start[s] = wgs;
        wgs += (int)(G.form[s] == 1 ? (M + 15) / 16 : (M + 63) / 64);
    }
    for (int s = n; s < PW_GROUP_MAX; ++s) { G.p[s] = G.p[0]; G.form[s] = 0; G.start[s] = wgs; }
    G.start[PW_GROUP_MAX] = wgs;
    G.n = n;
    hipStream_t st = (hipStream_t)stream;
    bool any_splitk = false;
    for (int s = 0; s < n; ++s) any_splitk = any_splitk || G.form[s] == 1;
#define SS_GRP(BN) do { if (conv3 && any_splitk) hipLaunchKernelGGL((k_pw_group<BN, true, true>), dim3(wgs), dim3(256), 0, st, G);            \
                        else if (conv3) hipLaunchKernelGGL((k_pw_group<BN, true, false>), dim3(wgs), dim3(256), 0, st, G);                    \
                        else hipLaunchKernelGGL((k_pw_group<BN, false, false>), dim3(wgs), dim3(256), 0, st, G); } while (0)
    if (nmax <= 32) SS_GRP(32);
    else if (nmax <= 64) SS_GRP(64);
    else SS_GRP(80);
#undef SS_GRP
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_upcat_f16(void* stream, const void* lo, const void* hi, void* out, int B, int h, int w, int C1, int C2, int lo_first)
{
    if (!lo || !hi || !out || B < 1 || h < 1 || w < 1 || C1 < 8 || C1 % 8 || C2 < 8 || C2 % 8) return SS_ERR_INVALID;
    const size_t total = (size_t)B * 4 * h * w * ((C1 + C2) / 8);
    hipLaunchKernelGGL(k_upcat, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const __half*)lo, (const __half*)hi,
                       (__half*)out, B, h, w, C1 / 8, C2 / 8, lo_first);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_osnet_head_f16(void* stream, const void* x, const void* w, const void* bias, void* out, int N, int HW, int C, int F)
{
    if (!x || !w || !bias || !out || N < 1 || HW < 1 || C != 128 || F < 1) return SS_ERR_INVALID;
    hipLaunchKernelGGL(k_osnet_head<128>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const __half*)x, (const __half*)w,
                       (const __half*)bias, (__half*)out, N, HW, F, nv_for(stream, N));
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_psa_attention_f16(void* stream, const void* qkv, const void* pe, void* out, int B, int N, int heads, float scale)
{
    if (!qkv || !out || B < 1 || B > 65535 || N < 1 || N > 16 * PSA_MAXT || heads < 1 || heads > 64) return SS_ERR_INVALID;
    const int NP = (N + 31) & ~31;
    hipLaunchKernelGGL(k_psa_attn, dim3((N + 63) / 64, heads, B), dim3(256), (size_t)64 * (NP + 16) * 2, (hipStream_t)stream,
                       (const __half*)qkv, (const __half*)pe, (__half*)out, N, heads, scale);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_sppf_pools_f16(void* stream, const void* x, void* out, int B, int H, int W, int C)
{
    if (!x || !out || B < 1 || H < 1 || W < 1 || C < 8 || C % 8 || H * W > 1024) return SS_ERR_INVALID;
    hipLaunchKernelGGL(k_sppf_pools, dim3((unsigned)(B * (C / 8))), dim3(256), (size_t)2 * H * W * 16, (hipStream_t)stream,
                       (const __half*)x, (__half*)out, H, W, C / 8);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

extern "C" int ss_op_conv0_f16(void* stream, const void* x, const void* w_prep, const void* bias, void* y, int B, int H, int W, int Cout,
                               int act)
{
    if (!x || !w_prep || !bias || !y || B < 1 || H < 2 || W < 128 || W % 128 || (W * 3) % 8 || (Cout != 16 && Cout != 32 && Cout != 48))
        return SS_ERR_INVALID;
    const int OH = (H - 1) / 2 + 1;
    const dim3 grid(W / 128, (OH + C0_ROWS - 1) / C0_ROWS, B), block(256);
    hipStream_t st = (hipStream_t)stream;
#define SS_C0(CO) hipLaunchKernelGGL(k_conv0<CO>, grid, block, 0, st, (const __half*)x, (const __half*)w_prep, (const __half*)bias, (__half*)y, H, W, act)
    if (Cout == 16) SS_C0(16); else if (Cout == 32) SS_C0(32); else SS_C0(48);
#undef SS_C0
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

// C2f bottleneck in one launch (k_bneck).  x dense [B][H][W][C] half, C in {16, 32, 64, 128}; w1 / w2 [C][3][3][C]; out = a channel slice
// of a wider NHWC tensor (out_ld elements per pixel, pointer already at the slice), out2 = dense [B][H][W][C] copy or NULL.
extern "C" int ss_op_bottleneck_f16(void* stream, const void* x, const void* w1, const void* b1, const void* w2, const void* b2, int B, int H,
                                    int W, int C, int add, void* out, int out_ld, void* out2)
{
    if (!x || !w1 || !b1 || !w2 || !b2 || !out || B < 1 || H < 1 || W < 1 || (C != 16 && C != 32 && C != 64 && C != 128) || out_ld < C || out_ld % 4 ||
        ((uintptr_t)out % 8) || ((uintptr_t)x % 16) || ((uintptr_t)w1 % 16) || ((uintptr_t)w2 % 16) || (out2 && ((uintptr_t)out2 % 8)))
        return SS_ERR_INVALID;
    // 128-pixel tiles (8 x 16) while they still give >= 2 workgroups per CU, else 64-pixel tiles (8 x 8)
    const long long M = (long long)B * H * W;
    const bool big = M / 128 >= 512 && C <= 64;                    // (C = 128: the 8 x 16 tile's input + intermediate + weights exceed the LDS)
    BnArgs A{ (const __half*)x, (const __half*)w1, (const __half*)b1, (const __half*)w2, (const __half*)b2, (__half*)out, out_ld, (__half*)out2,
              B, H, W, add, big ? 16 : 8, 8, 0, 0 };
    A.tiles_x = (W + A.TW - 1) / A.TW; A.tiles_y = (H + A.TH - 1) / A.TH;
    const int P = C + 8;
    const size_t lds = ((size_t)(A.TH + 4) * (A.TW + 4) * P + (((size_t)(A.TH + 2) * (A.TW + 2) * P + 7) & ~(size_t)7) + 2 * (size_t)C * 72) * 2 + 128;
    const dim3 grid((unsigned)(B * A.tiles_x * A.tiles_y));
    hipStream_t st = (hipStream_t)stream;
#define SS_BN(CC, P1, P2) do { static unsigned long long attr = 0; int dv_ = 0; (void)hipGetDevice(&dv_); if (!(attr >> (dv_ & 63) & 1)) { (void)hipFuncSetAttribute((const void*)k_bneck<CC, P1, P2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); attr |= 1ull << (dv_ & 63); }   /* the attribute is per DEVICE */ \
                               hipLaunchKernelGGL((k_bneck<CC, P1, P2>), grid, dim3(256), lds, st, A); } while (0)
    if (big) { if (C == 16) SS_BN(16, 3, 2); else if (C == 32) SS_BN(32, 3, 2); else SS_BN(64, 3, 2); }
    else { if (C == 16) SS_BN(16, 2, 1); else if (C == 32) SS_BN(32, 2, 1); else if (C == 64) SS_BN(64, 2, 1); else SS_BN(128, 2, 1); }
#undef SS_BN
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

// One level of the v8 detect head, both branches, three layers each, in one launch (k_head).  x dense [B][H][W][Cin] half,
// Cin in {64, 128, 256}; branch 0: w1 [64][3][3][Cin], w2 [64][3][3][64], w3 [nout0][64] (nout0 <= 64); branch 1: the same with 80
// mid channels (nout1 <= 80); out[b] dense [B][H][W][nout_b].  tile16 != 0: 8 x 16 pixel tiles (Cin = 64 only) instead of 8 x 8.
extern "C" int ss_op_head_f16(void* stream, const void* x, const void* const* w1, const void* const* b1, const void* const* w2,
                              const void* const* b2, const void* const* w3, const void* const* b3, void* const* out, const int* nout,
                              int B, int H, int W, int Cin, int tile16)
{
    if (!x || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !out || !nout || B < 1 || H < 1 || W < 1 || (Cin != 64 && Cin != 128 && Cin != 256) ||
        ((uintptr_t)x % 16))
        return SS_ERR_INVALID;
    HeadArgs A;
    A.x = (const __half*)x;
    for (int b = 0; b < 2; ++b) {
        const int cm = b ? 80 : 64;
        if (!w1[b] || !b1[b] || !w2[b] || !b2[b] || !w3[b] || !b3[b] || !out[b] || nout[b] < 8 || nout[b] % 8 || nout[b] > cm ||
            ((uintptr_t)w1[b] % 16) || ((uintptr_t)w2[b] % 16) || ((uintptr_t)w3[b] % 16) || ((uintptr_t)out[b] % 8))
            return SS_ERR_INVALID;
        A.w1[b] = (const __half*)w1[b]; A.b1[b] = (const __half*)b1[b]; A.w2[b] = (const __half*)w2[b]; A.b2[b] = (const __half*)b2[b];
        A.w3[b] = (const __half*)w3[b]; A.b3[b] = (const __half*)b3[b]; A.out[b] = (__half*)out[b]; A.nout[b] = nout[b];
    }
    const bool big = tile16 != 0 && Cin == 64;
    A.B = B; A.H = H; A.W = W; A.TW = big ? 16 : 8; A.TH = 8;
    A.tiles_x = (W + A.TW - 1) / A.TW; A.tiles_y = (H + A.TH - 1) / A.TH;
    const size_t xs = ((size_t)(A.TH + 4) * (A.TW + 4) * (Cin + 8) + 7) & ~(size_t)7, ts = ((size_t)(A.TH + 2) * (A.TW + 2) * 88 + 7) & ~(size_t)7;
    const size_t lds = (xs + ts + 2 * 80 * 72) * 2 + 128;
    if (lds > 160 * 1024 - 512) return SS_ERR_INVALID;
    const dim3 grid((unsigned)(B * A.tiles_x * A.tiles_y), 2);
    hipStream_t st = (hipStream_t)stream;
#define SS_HD(CC, P1, P2) do { static unsigned long long attr = 0; int dv_ = 0; (void)hipGetDevice(&dv_); if (!(attr >> (dv_ & 63) & 1)) { (void)hipFuncSetAttribute((const void*)k_head<CC, P1, P2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); attr |= 1ull << (dv_ & 63); }   /* the attribute is per DEVICE */ \
                               hipLaunchKernelGGL((k_head<CC, P1, P2>), grid, dim3(256), lds, st, A); } while (0)
    if (big) SS_HD(64, 3, 2);
    else if (Cin == 64) SS_HD(64, 2, 1);
    else if (Cin == 128) SS_HD(128, 2, 1);
    else SS_HD(256, 2, 1);
#undef SS_HD
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ERR_HIP;
}

// The OSNet entry points above (stem, pointwise, streams, tail) launched on `stream` by this host thread compute only the first
// *d_n images of launches whose batch is `batch`, until the next call for that stream; d_n == NULL switches it off.
// Host-side setting, read at launch time (and baked into a HIP graph that captures the launches).
extern "C" int ss_op_set_valid_images(void* stream, const int* d_n, int batch)
{
    NvEntry* slot = nullptr;
    for (NvEntry& e : g_nv) if (e.n && e.stream == stream) slot = &e;
    if (!d_n) { if (slot) *slot = NvEntry{}; return SS_OK; }
    if (batch < 1) return SS_ERR_INVALID;
    if (!slot) for (NvEntry& e : g_nv) if (!e.n) { slot = &e; break; }
    if (!slot) return SS_ERR_CAPACITY;                              // more than 8 streams of one thread with a valid count at once
    *slot = NvEntry{ stream, d_n, batch };
    return SS_OK;
}

// A/B switches of the operators (process-wide, for measurements; all 1 by default):
//   "pw_epilogue"   16-byte vector epilogue of the pointwise / 3x3 kernels (0: the 8-byte form)
//   "pw_splitk"     split-K form for 3x3 layers with few pixels and a long K walk
//   "osnet_chains"  register-resident row-stream form of the OSNet LightConv chains (0: the LDS form)
extern "C" int ss_op_set_option(const char* name, int value)
{
    if (!name) return SS_ERR_INVALID;
    const std::string n(name);
    if (n == "pw_epilogue") g_opt_pw_epilogue = value;
    else if (n == "pw_splitk") g_opt_pw_splitk = value;
    else if (n == "osnet_chains") g_opt_osnet_chains = value;
    else return SS_ERR_INVALID;
    return SS_OK;
}
