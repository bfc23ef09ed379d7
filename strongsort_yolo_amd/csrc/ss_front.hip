// ss_front.hip — frame-side kernels of the hot path: letterbox (a1), NMS (a3), ReID crop (a4).
// Arithmetic follows oracle/csrc/ss_oracle.c (so_letterbox / so_nms / so_scale_boxes / so_crop_norm)
// operation for operation; results are bit-identical (tests/test_gpu_front.py).
#include <hip/hip_fp16.h>
#include "ss_common.h"

// ---- shared bilinear helpers (oracle so_axis / so_bilerp_u8) ---------------------------------------
__device__ inline void ss_axis(int d, float scale, int n_src, int& i0, int& i1, float& frac)
{
    float t = (float)d + 0.5f;
    float s = t * scale;
    float f = s - 0.5f;
    int i = (int)floorf(f);
    float fr = f - (float)i;
    if (i < 0) { i = 0; fr = 0.0f; }
    if (i >= n_src - 1) { i = n_src - 1; fr = 0.0f; i1 = i; } else i1 = i + 1;
    i0 = i; frac = fr;
}

__device__ inline float ss_bilerp_u8(float p00, float p01, float p10, float p11, float fx, float fy)
{
    float a = fmaf(fx, p01 - p00, p00);
    float b = fmaf(fx, p11 - p10, p10);
    float v = fmaf(fy, b - a, a);
    float q = floorf(v + 0.5f);
    return fminf(fmaxf(q, 0.0f), 255.0f);
}

template <typename T> __device__ inline T ss_cvt(float v);
template <> __device__ inline float ss_cvt<float>(float v) { return v; }
template <> __device__ inline __half ss_cvt<__half>(float v) { return __float2half(v); }
// uint8_t output of the crop kernel: the rounded bilinear value itself (0..255), BEFORE the normalisation — the 256 x 3 possible outputs of
// ((q / 255) - mean) / sd are applied by the consumer (k32_stemW's staging) from the same expression, so the crops travel as a quarter of the bytes
template <typename T> struct ss_is_u8 { static constexpr bool value = false; };
template <> struct ss_is_u8<uint8_t> { static constexpr bool value = true; };

// =================================================================================================
// a1 letterbox: one thread per output pixel column pair; writes 3 planes
// =================================================================================================
// blockIdx.z = image of the batch; HWC != 0 writes [out_h][out_w][3] (channels-last) instead of 3 planes
template <typename T, int HWC>
__global__ __launch_bounds__(256) void k_letterbox(const uint8_t* __restrict__ src, long long src_batch_stride,
                                                   int H, int W, int stride, T* __restrict__ dst, int out_h,
                                                   int out_w, int new_h, int new_w, int pad_top, int pad_left,
                                                   float padv)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= out_w) return;
    const size_t plane = (size_t)out_h * out_w;
    src += (size_t)blockIdx.z * src_batch_stride;
    dst += (size_t)blockIdx.z * 3 * plane;
    const size_t o = HWC ? ((size_t)y * out_w + x) * 3 : (size_t)y * out_w + x;
    const size_t cs = HWC ? 1 : plane;
    const int ry = y - pad_top, rx = x - pad_left;
    if (ry < 0 || ry >= new_h || rx < 0 || rx >= new_w) {
        T p = ss_cvt<T>(padv);
        dst[o] = p; dst[cs + o] = p; dst[2 * cs + o] = p;
        return;
    }
    const float sx = (float)W / (float)new_w, sy = (float)H / (float)new_h;
    int y0, y1, x0, x1; float fy, fx;
    ss_axis(ry, sy, H, y0, y1, fy);
    ss_axis(rx, sx, W, x0, x1, fx);
    const uint8_t* r0 = src + (size_t)y0 * stride;
    const uint8_t* r1 = src + (size_t)y1 * stride;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sc = 2 - c;
        float p00 = r0[x0 * 3 + sc], p01 = r0[x1 * 3 + sc], p10 = r1[x0 * 3 + sc], p11 = r1[x1 * 3 + sc];
        dst[c * cs + o] = ss_cvt<T>(ss_bilerp_u8(p00, p01, p10, p11, fx, fy) / 255.0f);
    }
}

// flags: bit 0 = half output, bit 1 = channels-last output
void ss_launch_letterbox(const uint8_t* src, int batch, long long src_batch_stride, int h, int w, int stride, void* dst,
                         int flags, int out_h, int out_w, int new_h, int new_w, int pad_top, int pad_left,
                         int pad_value, hipStream_t st)
{
    if (batch <= 0) return;
    dim3 grid((out_w + 255) / 256, out_h, batch), block(256);
    float padv = (float)pad_value / 255.0f;
#define SS_LB(T, L) hipLaunchKernelGGL((k_letterbox<T, L>), grid, block, 0, st, src, src_batch_stride, h, w, stride, \
                                       (T*)dst, out_h, out_w, new_h, new_w, pad_top, pad_left, padv)
    switch (flags & 3) {
    case 0: SS_LB(float, 0); break;
    case 1: SS_LB(__half, 0); break;
    case 2: SS_LB(float, 1); break;
    default: SS_LB(__half, 1); break;
    }
#undef SS_LB
}

// =================================================================================================
// a4 ReID crop: grid (x tiles, out rows, dets); one thread per output pixel
// =================================================================================================
// blockIdx.z = image * n + detection; HWC != 0 writes [256][128][3] per crop (channels-last)
template <typename T, int HWC>
__global__ __launch_bounds__(128) void k_crop(const uint8_t* __restrict__ src, long long src_batch_stride, int H, int W,
                                              int stride, const float* __restrict__ dets, int det_stride,
                                              long long dets_batch_stride, int n, const int* __restrict__ d_count,
                                              T* __restrict__ dst)
{
    const int out_w = 128, out_h = 256;
    const int img = blockIdx.z / n, d = blockIdx.z - img * n;
    const int cnt = d_count ? d_count[img] : n;
    if (d >= cnt) return;
    src += (size_t)img * src_batch_stride;
    const int x = threadIdx.x, y = blockIdx.y;
    const float* b = dets + (size_t)img * dets_batch_stride + (size_t)d * det_stride;
    int x1 = (int)b[0], y1 = (int)b[1], x2 = (int)b[2], y2 = (int)b[3];
    if (x1 < 0) x1 = 0; if (y1 < 0) y1 = 0;
    if (x2 > W - 1) x2 = W - 1; if (y2 > H - 1) y2 = H - 1;
    if (x1 > W - 1) x1 = W - 1; if (y1 > H - 1) y1 = H - 1;
    int cw = x2 - x1, ch = y2 - y1;
    if (cw < 1) cw = 1; if (ch < 1) ch = 1;
    const float sx = (float)cw / (float)out_w, sy = (float)ch / (float)out_h;
    int yy0, yy1, xx0, xx1; float fy, fx;
    ss_axis(y, sy, ch, yy0, yy1, fy);
    ss_axis(x, sx, cw, xx0, xx1, fx);
    const uint8_t* r0 = src + (size_t)(y1 + yy0) * stride;
    const uint8_t* r1 = src + (size_t)(y1 + yy1) * stride;
    const float mean[3] = { 0.485f, 0.456f, 0.406f }, sd[3] = { 0.229f, 0.224f, 0.225f };
    const size_t plane = (size_t)out_h * out_w;
    T* o = dst + (size_t)blockIdx.z * 3 * plane + (HWC ? ((size_t)y * out_w + x) * 3 : (size_t)y * out_w + x);
    const size_t cs = HWC ? 1 : plane;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sc = 2 - c;
        float p00 = r0[(x1 + xx0) * 3 + sc], p01 = r0[(x1 + xx1) * 3 + sc];
        float p10 = r1[(x1 + xx0) * 3 + sc], p11 = r1[(x1 + xx1) * 3 + sc];
        float q = ss_bilerp_u8(p00, p01, p10, p11, fx, fy) / 255.0f;
        o[c * cs] = ss_cvt<T>((q - mean[c]) / sd[c]);
    }
}

// Channels-last form, vectorised: one thread = 8 consecutive output pixels x 3 channels = 24 values written as 16-byte
// stores (3 for half, 6 for float) — a crop row is 128 x 3 values contiguous, so a 16-thread group writes 768 / 1536
// contiguous bytes.  block = 16 output rows x 16 pixel groups of one crop.  Per-pixel arithmetic is k_crop's, bit for bit:
//  * the bilinear value is rounded to an integer 0..255 before the normalisation, so ((q / 255) - mean) / sd is one of
//    256 x 3 numbers: a table in LDS filled with exactly that expression replaces two correctly-rounded divisions per value
//    (~30 VALU instructions);
//  * the source rows the 16 output rows touch (the box's own columns) are staged in LDS with aligned dword loads and a
//    pixel's three bytes come out of two LDS dwords + a 64-bit shift (the byte-wide global loads — 96 per thread — were the
//    kernel's limit: 82 us per 512 crops).  Boxes whose row span does not fit the LDS budget take the global loads.
#define CROP_LDS_BYTES 30720
#define CROP_RPT 4            // output rows per thread: a workgroup covers a band of 64 rows of a crop (4 bands), so the fixed cost of a
                              // workgroup (box read, table, staging latency) is paid 4x less often than with 16-row bands (113 -> us per ~880 crops)

template <typename T>
__global__ __launch_bounds__(256) void k_crop_hwc8(const uint8_t* __restrict__ src, long long src_batch_stride, int H, int W,
                                                   int stride, const float* __restrict__ dets, int det_stride,
                                                   long long dets_batch_stride, int n, const int* __restrict__ d_count,
                                                   T* __restrict__ dst, const int* __restrict__ d_off)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char crop_smem[];
    T* lut = reinterpret_cast<T*>(crop_smem);                                  // [3][256]
    uint32_t* Ls = reinterpret_cast<uint32_t*>(crop_smem + 768 * sizeof(T));   // staged source rows
    const int out_w = 128, out_h = 256;
    const int img = blockIdx.z / n, d = blockIdx.z - img * n;
    const int cnt = d_count ? d_count[img] : n;
    if (d >= cnt) return;
    src += (size_t)img * src_batch_stride;
    const int tid = threadIdx.x, yb = blockIdx.y * (16 * CROP_RPT), xg = (tid & 15) * 8;
    const float* b = dets + (size_t)img * dets_batch_stride + (size_t)d * det_stride;
    int x1 = (int)b[0], y1 = (int)b[1], x2 = (int)b[2], y2 = (int)b[3];
    if (x1 < 0) x1 = 0; if (y1 < 0) y1 = 0;
    if (x2 > W - 1) x2 = W - 1; if (y2 > H - 1) y2 = H - 1;
    if (x1 > W - 1) x1 = W - 1; if (y1 > H - 1) y1 = H - 1;
    int cw = x2 - x1, ch = y2 - y1;
    if (cw < 1) cw = 1; if (ch < 1) ch = 1;
    const float sx = (float)cw / (float)out_w, sy = (float)ch / (float)out_h;
    const float mean[3] = { 0.485f, 0.456f, 0.406f }, sd[3] = { 0.229f, 0.224f, 0.225f };
    for (int i = tid; i < 768; i += 256) {
        const int c = i >> 8;
        const float q = (float)(i & 255) / 255.0f;
        if constexpr (ss_is_u8<T>::value) lut[i] = (uint8_t)(i & 255);
        else lut[i] = ss_cvt<T>((q - mean[c]) / sd[c]);
    }
    // source rows of this band (uniform): first row's upper tap .. last row's lower tap
    int ra0, ra1, rb0, rb1; float fa, fb;
    ss_axis(yb, sy, ch, ra0, ra1, fa);
    ss_axis(yb + 16 * CROP_RPT - 1, sy, ch, rb0, rb1, fb);
    const int nrows = rb1 - ra0 + 1;
    const size_t rowb = (size_t)x1 * 3;
    const int mis = (int)(rowb & 3), ndw = (mis + cw * 3 + 3) >> 2, pitch = ndw + 1;
    // mode 2: the band's source rows fit the LDS and are staged once; mode 1 (tall / wide boxes): staged per 16 output rows; mode 0
    // (boxes of which not even that fits, unaligned frames): byte loads from global memory
    const bool aligned = (stride & 3) == 0 && ((uintptr_t)src & 3) == 0;
    bool fit_sub = aligned;
#pragma unroll
    for (int rr = 0; rr < CROP_RPT; ++rr) {
        int a0, a1, b0, b1; float f0, f1;
        ss_axis(yb + 16 * rr, sy, ch, a0, a1, f0);
        ss_axis(yb + 16 * rr + 15, sy, ch, b0, b1, f1);
        fit_sub = fit_sub && (size_t)(b1 - a0 + 1) * pitch * 4 <= CROP_LDS_BYTES;
    }
    const int mode = (aligned && (size_t)nrows * pitch * 4 <= CROP_LDS_BYTES) ? 2 : fit_sub ? 1 : 0;
    const bool staged = mode != 0;
    const size_t end = (size_t)H * stride;
    auto stage = [&](const int r_first, const int nr) {
        // eight loads in flight per thread before the first LDS store (a load -> store loop pays one memory latency per
        // iteration: 22 iterations for a 64-row band of a 120-pixel box, the kernel's whole duration)
        const int tot = nr * ndw;
        for (int i0 = tid; i0 < tot; i0 += 256 * 8) {
            uint32_t v[8];
            int la[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u;
                v[u] = 0; la[u] = -1;
                if (i < tot) {
                    const int rr = i / ndw, dw = i - rr * ndw;
                    const size_t off = (size_t)(y1 + r_first + rr) * stride + (rowb - mis) + (size_t)dw * 4;
                    la[u] = rr * pitch + dw;
                    if (off + 4 <= end) v[u] = *reinterpret_cast<const uint32_t*>(src + off);
                    else { for (int k = 0; k < 4 && off + k < end; ++k) v[u] |= (uint32_t)src[off + k] << (8 * k); }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (la[u] >= 0) Ls[la[u]] = v[u];
        }
    };
    if (mode == 2) stage(ra0, nrows);
    __syncthreads();
    const size_t slot = d_off ? (size_t)d_off[img] + d : (size_t)blockIdx.z;     // packed: image i's crops follow image i-1's
    // the column taps of this thread's 8 pixels are the same for every row of the band: byte offsets of the two taps in a staged
    // row and the weight, once per workgroup instead of once per row
    int hob0[8], hob1[8];
    float hfx[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        int xx0, xx1;
        ss_axis(xg + p, sx, cw, xx0, xx1, hfx[p]);
        hob0[p] = mis + 3 * xx0; hob1[p] = mis + 3 * xx1;
    }
#pragma unroll 1
    for (int rr = 0; rr < CROP_RPT; ++rr) {
    const int y = yb + rr * 16 + (tid >> 4);
    int yy0, yy1; float fy;
    ss_axis(y, sy, ch, yy0, yy1, fy);
    __attribute__((aligned(16))) T o[24];
    if (staged) {
        int row0 = ra0;
        if (mode == 1) {                                                      // (uniform) this sub-band's rows replace the previous one's
            int a1, b0, b1; float f0, f1;
            ss_axis(yb + 16 * rr, sy, ch, row0, a1, f0);
            ss_axis(yb + 16 * rr + 15, sy, ch, b0, b1, f1);
            if (rr) __syncthreads();
            stage(row0, b1 - row0 + 1);
            __syncthreads();
        }
        const uint32_t* l0 = Ls + (yy0 - row0) * pitch;
        const uint32_t* l1 = Ls + (yy1 - row0) * pitch;
        auto px = [&](const uint32_t* row, int ob) -> uint32_t {             // bytes 0..2 = B, G, R of the source pixel at byte ob
            const int dw = ob >> 2;
            const uint64_t two = ((uint64_t)row[dw + 1] << 32) | row[dw];
            return (uint32_t)(two >> (8 * (ob & 3)));
        };
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float fx = hfx[p];
            const uint32_t v00 = px(l0, hob0[p]), v01 = px(l0, hob1[p]), v10 = px(l1, hob0[p]), v11 = px(l1, hob1[p]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int sh = 8 * (2 - c);
                const float p00 = (float)((v00 >> sh) & 255u), p01 = (float)((v01 >> sh) & 255u), p10 = (float)((v10 >> sh) & 255u),
                            p11 = (float)((v11 >> sh) & 255u);
                o[p * 3 + c] = lut[c * 256 + (int)ss_bilerp_u8(p00, p01, p10, p11, fx, fy)];
            }
        }
    } else {
        // boxes whose rows do not fit the LDS budget: byte loads from global memory, one pixel at a time (rolled: unrolled, its 96 loads in
        // flight set the kernel's register count - 174 VGPRs, 2 waves per SIMD - for the staged boxes as well)
        const uint8_t* r0 = src + (size_t)(y1 + yy0) * stride + rowb;
        const uint8_t* r1 = src + (size_t)(y1 + yy1) * stride + rowb;
        T* od = dst + (slot * out_h * out_w + (size_t)y * out_w + xg) * 3;
#pragma unroll 1
        for (int p = 0; p < 8; ++p) {
            int xx0, xx1; float fx;
            ss_axis(xg + p, sx, cw, xx0, xx1, fx);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int sc = 2 - c;
                const float p00 = r0[xx0 * 3 + sc], p01 = r0[xx1 * 3 + sc], p10 = r1[xx0 * 3 + sc], p11 = r1[xx1 * 3 + sc];
                od[p * 3 + c] = lut[c * 256 + (int)ss_bilerp_u8(p00, p01, p10, p11, fx, fy)];
            }
        }
        continue;
    }
    if constexpr (ss_is_u8<T>::value) {                          // 24 bytes: three 8-byte stores (a group starts at a multiple of 24 bytes)
        uint2* out = reinterpret_cast<uint2*>(dst + (slot * out_h * out_w + (size_t)y * out_w + xg) * 3);
        const uint2* ov = reinterpret_cast<const uint2*>(o);
#pragma unroll
        for (int v = 0; v < 3; ++v) out[v] = ov[v];
    } else {
        constexpr int NV = 24 * sizeof(T) / 16;
        uint4* out = reinterpret_cast<uint4*>(dst + (slot * out_h * out_w + (size_t)y * out_w + xg) * 3);
        const uint4* ov = reinterpret_cast<const uint4*>(o);
#pragma unroll
        for (int v = 0; v < NV; ++v) out[v] = ov[v];
    }
    }
}

// exclusive prefix of min(count, n) over the images of a batch -> off[batch + 1] (off[batch] = number of crops).
// Wave shuffles + one LDS hop (a single thread walking 1024 partial sums in LDS was ~50 us of the packed crop call, r03).
__global__ __launch_bounds__(1024) void k_crop_offsets(const int* __restrict__ counts, int batch, int n, int* __restrict__ off)
{
    __shared__ int wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, per = (batch + 1023) / 1024, i0 = tid * per;
    int sum = 0;
    for (int i = i0; i < i0 + per && i < batch; ++i) sum += min(counts[i], n);
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = wtot[i]; if (i < wv) base += c; total += c; }
    int a = base + inc - sum;
    if (tid == 0) off[batch] = total;
    for (int i = i0; i < i0 + per && i < batch; ++i) { off[i] = a; a += min(counts[i], n); }
}

// embeddings of a packed batch back to [image][slot]: feats[img][d][:] = emb[off[img] + d][:] for d < min(count, n)
template <typename T>
__global__ __launch_bounds__(128) void k_unpack_feats(const T* __restrict__ emb, const int* __restrict__ off, const int* __restrict__ counts,
                                                     int n, float* __restrict__ feats, long long feats_img_stride)
{
    const int img = blockIdx.y, d = blockIdx.x;
    if (d >= min(counts[img], n)) return;
    const T* src = emb + ((size_t)off[img] + d) * 512;
    float* dst = feats + (size_t)img * feats_img_stride + (size_t)d * 512;
    for (int k = threadIdx.x; k < 512; k += 128) dst[k] = (float)src[k];
}

// One frame's results in ONE buffer the host reads after its synchronisation: dst[0] = n = min(*n_dets, det_cap), dst[1] = m = min(*n_out, out_cap)
// (integers, bit-copied), then n detection rows of det_ld floats at dst + 2, then m track rows of out_ld floats at dst + 2 + det_cap * det_ld.
// dst may be pinned host memory (device-accessible): the per-frame drop-in call used to issue two device copies and three device-to-host
// copies for the same bytes, 89 us of its ~1 ms.
__global__ __launch_bounds__(256) void k_pack_results(const int* __restrict__ n_dets, const float* __restrict__ dets, int det_ld, int det_cap,
                                                      const int* __restrict__ n_out, const float* __restrict__ out, int out_ld, int out_cap,
                                                      float* __restrict__ dst)
{
    const int n = min(max(*n_dets, 0), det_cap), m = (n_out && out) ? min(max(*n_out, 0), out_cap) : 0;
    if (threadIdx.x == 0) { reinterpret_cast<int*>(dst)[0] = n; reinterpret_cast<int*>(dst)[1] = m; }
    float* d0 = dst + 2;
    for (int i = threadIdx.x; i < n * det_ld; i += 256) d0[i] = dets[i];
    float* d1 = dst + 2 + (size_t)det_cap * det_ld;
    for (int i = threadIdx.x; i < m * out_ld; i += 256) d1[i] = out[i];
}

void ss_launch_pack_results(const int* n_dets, const float* dets, int det_ld, int det_cap, const int* n_out, const float* out, int out_ld,
                            int out_cap, float* dst, hipStream_t st)
{ hipLaunchKernelGGL(k_pack_results, dim3(1), dim3(256), 0, st, n_dets, dets, det_ld, det_cap, n_out, out, out_ld, out_cap, dst); }

void ss_launch_crop_offsets(const int* counts, int batch, int n, int* off, hipStream_t st)
{ hipLaunchKernelGGL(k_crop_offsets, dim3(1), dim3(1024), 0, st, counts, batch, n, off); }

void ss_launch_unpack_feats(const void* emb, int half, const int* off, const int* counts, int batch, int n, float* feats,
                            long long feats_img_stride, hipStream_t st)
{
    if (batch <= 0 || n <= 0) return;
    if (half) hipLaunchKernelGGL(k_unpack_feats<__half>, dim3(n, batch), dim3(128), 0, st, (const __half*)emb, off, counts, n, feats, feats_img_stride);
    else hipLaunchKernelGGL(k_unpack_feats<float>, dim3(n, batch), dim3(128), 0, st, (const float*)emb, off, counts, n, feats, feats_img_stride);
}

// flags: bit 0 = half output, bit 1 = channels-last output
void ss_launch_crop(const uint8_t* frame, int batch, long long frame_batch_stride, int h, int w, int stride,
                    const float* dets, int det_stride, long long dets_batch_stride, int n, const int* d_count,
                    void* out, int flags, hipStream_t st, const int* d_off)
{
    if (n <= 0 || batch <= 0) return;
    if (flags & 2) {
        dim3 grid(1, 16 / CROP_RPT, n * batch), block(256);
        if (flags & 4) { hipLaunchKernelGGL(k_crop_hwc8<uint8_t>, grid, block, 768 + CROP_LDS_BYTES + 8, st, frame, frame_batch_stride, h, w, stride, dets, det_stride, dets_batch_stride, n, d_count, (uint8_t*)out, d_off); return; }
        if (flags & 1) hipLaunchKernelGGL(k_crop_hwc8<__half>, grid, block, 768 * sizeof(__half) + CROP_LDS_BYTES + 8, st, frame, frame_batch_stride, h, w, stride, dets, det_stride, dets_batch_stride, n, d_count, (__half*)out, d_off);
        else           hipLaunchKernelGGL(k_crop_hwc8<float>, grid, block, 768 * sizeof(float) + CROP_LDS_BYTES + 8, st, frame, frame_batch_stride, h, w, stride, dets, det_stride, dets_batch_stride, n, d_count, (float*)out, d_off);
        return;
    }
    dim3 grid(1, 256, n * batch), block(128);
#define SS_CR(T, L) hipLaunchKernelGGL((k_crop<T, L>), grid, block, 0, st, frame, frame_batch_stride, h, w, stride, dets, \
                                       det_stride, dets_batch_stride, n, d_count, (T*)out)
    if (flags & 1) SS_CR(__half, 0); else SS_CR(float, 0);
#undef SS_CR
}

// =================================================================================================
// a3 NMS
// =================================================================================================
#define NMS_MAX_ANCHORS 32768
#define NMS_MAX_CAND 8192
#define NMS_WORDS (NMS_MAX_CAND / 64)

struct NmsWs {
    unsigned long long* keys;     // [MAX_ANCHORS] candidate keys (unsorted), then sorted in place [MAX_CAND]
    int* cand_cls;                // [MAX_ANCHORS] class per anchor (indexed by anchor)
    float* box;                   // [MAX_CAND][4] offset boxes, sorted order
    float* area;                  // [MAX_CAND]
    unsigned long long* mask;     // [MAX_CAND][NMS_WORDS]
    int* counters;                // [0] candidates, [1] error
};

size_t ss_nms_workspace_bytes()
{
    return (size_t)NMS_MAX_ANCHORS * 8 + (size_t)NMS_MAX_ANCHORS * 4 + (size_t)NMS_MAX_CAND * 16 +
           (size_t)NMS_MAX_CAND * 4 + (size_t)NMS_MAX_CAND * NMS_WORDS * 8 + 64;
}

__host__ __device__ inline NmsWs carve_nms(void* ws)
{
    NmsWs w; char* p = (char*)ws;
    w.keys = (unsigned long long*)p; p += (size_t)NMS_MAX_ANCHORS * 8;
    w.mask = (unsigned long long*)p; p += (size_t)NMS_MAX_CAND * NMS_WORDS * 8;
    w.box = (float*)p; p += (size_t)NMS_MAX_CAND * 16;
    w.area = (float*)p; p += (size_t)NMS_MAX_CAND * 4;
    w.cand_cls = (int*)p; p += (size_t)NMS_MAX_ANCHORS * 4;
    w.counters = (int*)p;
    return w;
}

// Batched launch: every kernel takes the image index from its grid and finds that image's prediction
// tensor and workspace unit by stride.
struct NmsBatch {
    const float* pred; long long pred_stride;     // floats between images
    char* ws; long long ws_stride;                // bytes between workspace units
};
__device__ inline NmsWs nms_unit(const NmsBatch& nb, int img) { return carve_nms(nb.ws + (size_t)img * nb.ws_stride); }

// candidate filter: best class per anchor, score > conf.  key = (~score_bits, anchor): ascending
// key order == descending score, ties by ascending anchor (stable order of the oracle's sort).
__global__ __launch_bounds__(512) void k_nms_filter(NmsBatch nb, int N, int nc, float conf, unsigned long long cm0,
                                                    unsigned long long cm1)
{
    const float* __restrict__ pred = nb.pred + (size_t)blockIdx.y * nb.pred_stride;
    const NmsWs w = nms_unit(nb, blockIdx.y);
    // 64 anchors per block; wave g scans classes g, g+8, ... (coalesced over anchors), LDS combine
    __shared__ float sbest[8][64];
    __shared__ int scls[8][64];
    const int g = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int a = blockIdx.x * 64 + l;
    float best = -INFINITY; int bc = 0x7fffffff;
    if (a < N)
        for (int k0 = g; k0 < nc; k0 += 80) {                       // ten class rows in flight per thread (80 classes: one round trip, was ten)
            float s[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) s[j] = pred[(size_t)(4 + min(k0 + 8 * j, nc - 1)) * N + a];
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int k = k0 + 8 * j;
                if (k < nc && s[j] > best) { best = s[j]; bc = k; }
            }
        }
    sbest[g][l] = best; scls[g][l] = bc;
    __syncthreads();
    if (g != 0 || a >= N) return;
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        float s = sbest[q][l]; int c = scls[q][l];
        if (s > best || (s == best && c < bc)) { best = s; bc = c; }       // first maximum = lowest class
    }
    // `classes` override (yolo_multi_model.py:22): the anchor's best class must be in the allowed set
    const bool allowed = bc < 64 ? (cm0 >> bc) & 1ull : bc < 128 ? (cm1 >> (bc - 64)) & 1ull : true;
    if (best > conf && allowed) {
        int slot = atomicAdd(&w.counters[0], 1);
        w.keys[slot] = ((unsigned long long)(~__float_as_uint(best)) << 32) | (unsigned)a;
        w.cand_cls[a] = bc;
    }
}

// single-block bitonic sort of the candidate keys in LDS, then offset boxes / areas in sorted order
__global__ __launch_bounds__(1024) void k_nms_sort(NmsBatch nb, int N, int agnostic, float max_wh)
{
    const float* __restrict__ pred = nb.pred + (size_t)blockIdx.x * nb.pred_stride;
    const NmsWs w = nms_unit(nb, blockIdx.x);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* k = (unsigned long long*)smem;
    int n = w.counters[0];
    __syncthreads();
    if (n > NMS_MAX_CAND) {       // too many candidates: flag it and make the mask / scan kernels see an empty image
        if (threadIdx.x == 0) { w.counters[1] = SS_ERR_CAPACITY; w.counters[0] = 0; }
        n = 0;
    }
    int np = 1; while (np < n) np <<= 1;
    for (int i = threadIdx.x; i < np; i += 1024) k[i] = i < n ? w.keys[i] : ~0ull;
    __syncthreads();
    for (int size = 2; size <= np; size <<= 1)
        for (int strd = size >> 1; strd > 0; strd >>= 1) {
            for (int i = threadIdx.x; i < np / 2; i += 1024) {
                int lo = 2 * i - (i & (strd - 1)), hi = lo + strd;
                bool up = (lo & size) == 0;
                unsigned long long a = k[lo], b = k[hi];
                if ((a > b) == up) { k[lo] = b; k[hi] = a; }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += 1024) {
        unsigned long long key = k[i];
        w.keys[i] = key;
        const int a = (int)(key & 0xffffffffu);
        float cx = pred[a], cy = pred[(size_t)N + a], bw = pred[(size_t)2 * N + a], bh = pred[(size_t)3 * N + a];
        float hw = bw / 2.0f, hh = bh / 2.0f;
        float off = agnostic ? 0.0f : (float)w.cand_cls[a] * max_wh;
        float x1 = (cx - hw) + off, y1 = (cy - hh) + off, x2 = (cx + hw) + off, y2 = (cy + hh) + off;
        w.box[i * 4 + 0] = x1; w.box[i * 4 + 1] = y1; w.box[i * 4 + 2] = x2; w.box[i * 4 + 3] = y2;
        w.area[i] = (x2 - x1) * (y2 - y1);
    }
}

// suppression bit matrix: mask[i][jb] bit t set when j = jb*64+t > i and IoU(i,j) > thr.
// grid = (NMS_MASK_WG, images): the workgroups of an image walk the upper-triangle 64x64 blocks (ib <= jb) of ITS
// candidate count with a stride loop — the count is only known on the device, and a grid sized for the worst case
// (128 x 128 blocks per image) was ~100 k empty workgroups per launch (25 us).
#define NMS_MASK_WG 256
__global__ __launch_bounds__(64) void k_nms_mask(NmsBatch nb, float iou_thres)
{
    const NmsWs w = nms_unit(nb, blockIdx.y);
    __shared__ float sb[64 * 4];
    __shared__ float sa[64];
    const int n = min(w.counters[0], NMS_MAX_CAND);
    const int nblk = (n + 63) / 64;
    const int npair = nblk * (nblk + 1) / 2;
    const int t = threadIdx.x;
    for (int pr = blockIdx.x; pr < npair; pr += gridDim.x) {
        // pair index -> (ib, jb), ib <= jb, row-major over the upper triangle
        int ib = 0, rem = pr;
        while (rem >= nblk - ib) { rem -= nblk - ib; ++ib; }
        const int jb = ib + rem;
        const int j = jb * 64 + t;
        __syncthreads();                                   // the previous block's readers are done with sb / sa
        if (j < n) { sb[t * 4] = w.box[j * 4]; sb[t * 4 + 1] = w.box[j * 4 + 1]; sb[t * 4 + 2] = w.box[j * 4 + 2]; sb[t * 4 + 3] = w.box[j * 4 + 3]; sa[t] = w.area[j]; }
        __syncthreads();
        const int i = ib * 64 + t;
        if (i >= n) continue;
        const float x1 = w.box[i * 4], y1 = w.box[i * 4 + 1], x2 = w.box[i * 4 + 2], y2 = w.box[i * 4 + 3], ai = w.area[i];
        unsigned long long bits = 0;
        const int jn = min(64, n - jb * 64);
        for (int q = 0; q < jn; ++q) {
            if (jb * 64 + q <= i) continue;
            float xx1 = fmaxf(x1, sb[q * 4]), yy1 = fmaxf(y1, sb[q * 4 + 1]);
            float xx2 = fminf(x2, sb[q * 4 + 2]), yy2 = fminf(y2, sb[q * 4 + 3]);
            float iw = fmaxf(0.0f, xx2 - xx1), ih = fmaxf(0.0f, yy2 - yy1);
            float inter = iw * ih;
            float iou = inter / (ai + sa[q] - inter);
            if (iou > iou_thres) bits |= 1ull << q;
        }
        w.mask[(size_t)i * NMS_WORDS + jb] = bits;
    }
}

// greedy scan (one wave) + output rows in original-image pixels
// geom != NULL: per-image [gain, pad_x, pad_y, w0, h0] on the device (images of different sizes in one batch)
__global__ __launch_bounds__(64) void k_nms_scan(NmsBatch nb, int N, int nc, int n_extra, int max_det, float gain,
                                                 float pad_x, float pad_y, float w0, float h0,
                                                 const float* __restrict__ geom, float* __restrict__ rows,
                                                 int row_stride, long long rows_batch_stride, int* __restrict__ keep,
                                                 long long keep_batch_stride, int* __restrict__ count)
{
    const int img = blockIdx.x;
    const float* __restrict__ pred = nb.pred + (size_t)img * nb.pred_stride;
    const NmsWs w = nms_unit(nb, img);
    rows += (size_t)img * rows_batch_stride; keep += (size_t)img * keep_batch_stride; count += img;
    if (geom) { const float* g = geom + img * 5; gain = g[0]; pad_x = g[1]; pad_y = g[2]; w0 = g[3]; h0 = g[4]; }
    __shared__ int kept_sorted[1024];
    const int n = min(w.counters[0], NMS_MAX_CAND);
    const int l = threadIdx.x;
    const int nw = (n + 63) / 64;
    unsigned long long rem0 = 0, rem1 = 0;       // removed bits of words l and l+64
    int kept = 0;
    const int cap = min(max_det, 1024);
    for (int ib = 0; ib < nw && kept < cap; ++ib) {
        // removed word of block ib lives in lane ib%64 (rem0 or rem1)
        unsigned long long rw = __shfl((ib < 64) ? rem0 : rem1, ib & 63);
        // diagonal block: resolve dependencies inside the 64-block sequentially on the scalar side
        const int i0 = ib * 64;
        const int cnt = min(64, n - i0);
        unsigned long long diag = (l < cnt) ? w.mask[(size_t)(i0 + l) * NMS_WORDS + ib] : 0ull;
        unsigned long long keptbits = 0;
        for (int q = 0; q < cnt && kept < cap; ++q) {
            if (!((rw >> q) & 1ull)) {
                keptbits |= 1ull << q;
                if (l == 0) kept_sorted[kept] = i0 + q;
                ++kept;
                rw |= __shfl(diag, q);
            }
        }
        // OR the rows of the kept boxes into the removed words of the later blocks; four independent row
        // loads in flight per step (a dependent one-row-per-step chain costs one memory latency per kept box)
        unsigned long long kb = keptbits;
        while (kb) {
            int q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { q[u] = kb ? __builtin_ctzll(kb) : -1; if (kb) kb &= kb - 1; }
            unsigned long long a0[4], a1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0[u] = a1[u] = 0ull;
                if (q[u] >= 0) {
                    const unsigned long long* mr = w.mask + (size_t)(i0 + q[u]) * NMS_WORDS;
                    if (l > ib && l < nw) a0[u] = mr[l];
                    if (l + 64 > ib && l + 64 < nw) a1[u] = mr[l + 64];
                }
            }
            rem0 |= (a0[0] | a0[1]) | (a0[2] | a0[3]);
            rem1 |= (a1[0] | a1[1]) | (a1[2] | a1[3]);
        }
    }
    __syncthreads();
    if (l == 0) { *count = kept; w.counters[0] = 0; }     // re-arm the candidate counter for the next call
    for (int kk = l; kk < kept; kk += 64) {
        const int i = kept_sorted[kk];
        const unsigned long long key = w.keys[i];
        const int a = (int)(key & 0xffffffffu);
        const float score = __uint_as_float(~(unsigned)(key >> 32));
        float cx = pred[a], cy = pred[(size_t)N + a], bw = pred[(size_t)2 * N + a], bh = pred[(size_t)3 * N + a];
        float hw = bw / 2.0f, hh = bh / 2.0f;
        float x1 = ((cx - hw) - pad_x) / gain, y1 = ((cy - hh) - pad_y) / gain;
        float x2 = ((cx + hw) - pad_x) / gain, y2 = ((cy + hh) - pad_y) / gain;
        float* r = rows + (size_t)kk * row_stride;
        r[0] = fminf(fmaxf(x1, 0.0f), w0); r[1] = fminf(fmaxf(y1, 0.0f), h0);
        r[2] = fminf(fmaxf(x2, 0.0f), w0); r[3] = fminf(fmaxf(y2, 0.0f), h0);
        r[4] = score; r[5] = (float)w.cand_cls[a];
        for (int e = 0; e < n_extra; ++e) r[6 + e] = pred[(size_t)(4 + nc + e) * N + a];
        keep[kk] = a;
    }
}

// Sort + suppression matrix + greedy scan + output rows of ONE image in ONE workgroup (1024 threads): the three kernels above
// run back to back for a few dozen candidates per image (each launch costs more than its work: 75 us per 32 images for four
// launches, r03 roofline_front), so the batched entry point launches k_nms_filter and this.  Same arithmetic, same order:
// phase A = k_nms_sort's body, phase B = k_nms_mask's body with one WAVE per 64x64 block pair (per-wave LDS tiles),
// phase C = k_nms_scan's greedy pass on wave 0, then all threads write the rows.
__global__ __launch_bounds__(1024) void k_nms_rest(NmsBatch nb, int N, int nc, int n_extra, int agnostic, float max_wh, float iou_thres,
                                                   int max_det, float gain, float pad_x, float pad_y, float w0, float h0,
                                                   const float* __restrict__ geom, float* __restrict__ rows, int row_stride,
                                                   long long rows_batch_stride, int* __restrict__ keep, long long keep_batch_stride,
                                                   int* __restrict__ count)
{
    const int img = blockIdx.x;
    const float* __restrict__ pred = nb.pred + (size_t)img * nb.pred_stride;
    const NmsWs w = nms_unit(nb, img);
    rows += (size_t)img * rows_batch_stride; keep += (size_t)img * keep_batch_stride; count += img;
    if (geom) { const float* g = geom + img * 5; gain = g[0]; pad_x = g[1]; pad_y = g[2]; w0 = g[3]; h0 = g[4]; }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* k = (unsigned long long*)smem;                         // [NMS_MAX_CAND] sort keys
    float* sbw = (float*)(smem + (size_t)NMS_MAX_CAND * 8);                     // [16 waves][64][4] boxes of a column block
    float* saw = sbw + 16 * 64 * 4;                                            // [16][64] areas
    int* kept_sorted = (int*)(saw + 16 * 64);                                  // [1024]
    __shared__ int s_kept;
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
    // ---- phase A: sort, offset boxes, areas ----
    int n = w.counters[0];
    __syncthreads();
    if (n > NMS_MAX_CAND) {
        if (tid == 0) { w.counters[1] = SS_ERR_CAPACITY; w.counters[0] = 0; }
        n = 0;
    }
    if (n <= 64) {
        // ---- small images (the usual case: ~30 candidates at the reference's thresholds): ONE wave does the three phases in registers ----
        // sort = a 64-lane bitonic network on shuffles (keys are unique: any correct sort gives the order of the LDS sort below), boxes /
        // areas with the same expressions, the IoU row of candidate i against every later one as one 64-bit word per lane, the greedy scan
        // of phase C on that word.  No barrier, no round trip through the global workspace (the general path: 15+ barriers, boxes / areas /
        // bit matrix written and read back; 37 us per call for ~30 candidates per image, latency only).
        if (wv == 0) {
            unsigned long long key = l < n ? w.keys[l] : ~0ull;
            auto shfl64 = [](unsigned long long v, int src) {
                const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src);
                return ((unsigned long long)hi << 32) | lo;
            };
#pragma unroll
            for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
                for (int strd = size >> 1; strd > 0; strd >>= 1) {
                    const unsigned long long other = shfl64(key, l ^ strd);
                    const bool lower = (l & strd) == 0, up = (l & size) == 0;       // ascending blocks where (l & size) == 0 (size 64: the whole wave)
                    const bool take_min = lower == up;
                    key = take_min ? (key < other ? key : other) : (key > other ? key : other);
                }
            float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, ai = 0.f;
            if (l < n) {
                w.keys[l] = key;
                const int a = (int)(key & 0xffffffffu);
                float cx = pred[a], cy = pred[(size_t)N + a], bw = pred[(size_t)2 * N + a], bh = pred[(size_t)3 * N + a];
                float hw = bw / 2.0f, hh = bh / 2.0f;
                float off = agnostic ? 0.0f : (float)w.cand_cls[a] * max_wh;
                x1 = (cx - hw) + off; y1 = (cy - hh) + off; x2 = (cx + hw) + off; y2 = (cy + hh) + off;
                ai = (x2 - x1) * (y2 - y1);
                sbw[l * 4] = x1; sbw[l * 4 + 1] = y1; sbw[l * 4 + 2] = x2; sbw[l * 4 + 3] = y2; saw[l] = ai;
            }
            SS_WAVE_SYNC();
            unsigned long long bits = 0;
            if (l < n) {
                for (int q = 0; q < n; ++q) {
                    if (q <= l) continue;
                    float xx1 = fmaxf(x1, sbw[q * 4]), yy1 = fmaxf(y1, sbw[q * 4 + 1]);
                    float xx2 = fminf(x2, sbw[q * 4 + 2]), yy2 = fminf(y2, sbw[q * 4 + 3]);
                    float iw = fmaxf(0.0f, xx2 - xx1), ih = fmaxf(0.0f, yy2 - yy1);
                    float inter = iw * ih;
                    float iou = inter / (ai + saw[q] - inter);
                    if (iou > iou_thres) bits |= 1ull << q;
                }
            }
            unsigned long long rw = 0;
            int kept = 0;
            const int cap = min(max_det, 1024);
            for (int q = 0; q < n && kept < cap; ++q) {
                if (!((rw >> q) & 1ull)) {
                    if (l == 0) kept_sorted[kept] = q;
                    ++kept;
                    rw |= shfl64(bits, q);
                }
            }
            if (l == 0) { s_kept = kept; *count = kept; w.counters[0] = 0; }
        }
        __threadfence_block();
        __syncthreads();
    } else {
    int np = 1; while (np < n) np <<= 1;
    for (int i = tid; i < np; i += 1024) k[i] = i < n ? w.keys[i] : ~0ull;
    __syncthreads();
    for (int size = 2; size <= np; size <<= 1)
        for (int strd = size >> 1; strd > 0; strd >>= 1) {
            for (int i = tid; i < np / 2; i += 1024) {
                int lo = 2 * i - (i & (strd - 1)), hi = lo + strd;
                bool up = (lo & size) == 0;
                unsigned long long a = k[lo], b = k[hi];
                if ((a > b) == up) { k[lo] = b; k[hi] = a; }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += 1024) {
        unsigned long long key = k[i];
        w.keys[i] = key;
        const int a = (int)(key & 0xffffffffu);
        float cx = pred[a], cy = pred[(size_t)N + a], bw = pred[(size_t)2 * N + a], bh = pred[(size_t)3 * N + a];
        float hw = bw / 2.0f, hh = bh / 2.0f;
        float off = agnostic ? 0.0f : (float)w.cand_cls[a] * max_wh;
        float x1 = (cx - hw) + off, y1 = (cy - hh) + off, x2 = (cx + hw) + off, y2 = (cy + hh) + off;
        w.box[i * 4 + 0] = x1; w.box[i * 4 + 1] = y1; w.box[i * 4 + 2] = x2; w.box[i * 4 + 3] = y2;
        w.area[i] = (x2 - x1) * (y2 - y1);
    }
    __threadfence_block();
    __syncthreads();
    // ---- phase B: suppression bit matrix, one wave per (ib <= jb) block pair ----
    {
        float* sb = sbw + wv * 256;
        float* sa = saw + wv * 64;
        const int nblk = (n + 63) / 64, npair = nblk * (nblk + 1) / 2, t = l;
        for (int pr = wv; pr < npair; pr += 16) {
            int ib = 0, rem = pr;
            while (rem >= nblk - ib) { rem -= nblk - ib; ++ib; }
            const int jb = ib + rem;
            const int j = jb * 64 + t;
            SS_WAVE_SYNC();                                          // the previous pair's reads of sb / sa are done
            if (j < n) { sb[t * 4] = w.box[j * 4]; sb[t * 4 + 1] = w.box[j * 4 + 1]; sb[t * 4 + 2] = w.box[j * 4 + 2]; sb[t * 4 + 3] = w.box[j * 4 + 3]; sa[t] = w.area[j]; }
            SS_WAVE_SYNC();
            const int i = ib * 64 + t;
            if (i >= n) continue;
            const float x1 = w.box[i * 4], y1 = w.box[i * 4 + 1], x2 = w.box[i * 4 + 2], y2 = w.box[i * 4 + 3], ai = w.area[i];
            unsigned long long bits = 0;
            const int jn = min(64, n - jb * 64);
            for (int q = 0; q < jn; ++q) {
                if (jb * 64 + q <= i) continue;
                float xx1 = fmaxf(x1, sb[q * 4]), yy1 = fmaxf(y1, sb[q * 4 + 1]);
                float xx2 = fminf(x2, sb[q * 4 + 2]), yy2 = fminf(y2, sb[q * 4 + 3]);
                float iw = fmaxf(0.0f, xx2 - xx1), ih = fmaxf(0.0f, yy2 - yy1);
                float inter = iw * ih;
                float iou = inter / (ai + sa[q] - inter);
                if (iou > iou_thres) bits |= 1ull << q;
            }
            w.mask[(size_t)i * NMS_WORDS + jb] = bits;
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- phase C: greedy scan on wave 0 ----
    if (wv == 0) {
        const int nw = (n + 63) / 64;
        unsigned long long rem0 = 0, rem1 = 0;
        int kept = 0;
        const int cap = min(max_det, 1024);
        for (int ib = 0; ib < nw && kept < cap; ++ib) {
            unsigned long long rw = __shfl((ib < 64) ? rem0 : rem1, ib & 63);
            const int i0 = ib * 64;
            const int cnt = min(64, n - i0);
            unsigned long long diag = (l < cnt) ? w.mask[(size_t)(i0 + l) * NMS_WORDS + ib] : 0ull;
            unsigned long long keptbits = 0;
            for (int q = 0; q < cnt && kept < cap; ++q) {
                if (!((rw >> q) & 1ull)) {
                    keptbits |= 1ull << q;
                    if (l == 0) kept_sorted[kept] = i0 + q;
                    ++kept;
                    rw |= __shfl(diag, q);
                }
            }
            unsigned long long kb = keptbits;
            while (kb) {
                int q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { q[u] = kb ? __builtin_ctzll(kb) : -1; if (kb) kb &= kb - 1; }
                unsigned long long a0[4], a1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a0[u] = a1[u] = 0ull;
                    if (q[u] >= 0) {
                        const unsigned long long* mr = w.mask + (size_t)(i0 + q[u]) * NMS_WORDS;
                        if (l > ib && l < nw) a0[u] = mr[l];
                        if (l + 64 > ib && l + 64 < nw) a1[u] = mr[l + 64];
                    }
                }
                rem0 |= (a0[0] | a0[1]) | (a0[2] | a0[3]);
                rem1 |= (a1[0] | a1[1]) | (a1[2] | a1[3]);
            }
        }
        if (l == 0) { s_kept = kept; *count = kept; w.counters[0] = 0; }     // re-arm the candidate counter for the next call
    }
    __syncthreads();
    }                                                            // (n > 64)
    const int kept = s_kept;
    for (int kk = tid; kk < kept; kk += 1024) {
        const int i = kept_sorted[kk];
        const unsigned long long key = w.keys[i];
        const int a = (int)(key & 0xffffffffu);
        const float score = __uint_as_float(~(unsigned)(key >> 32));
        float cx = pred[a], cy = pred[(size_t)N + a], bw = pred[(size_t)2 * N + a], bh = pred[(size_t)3 * N + a];
        float hw = bw / 2.0f, hh = bh / 2.0f;
        float x1 = ((cx - hw) - pad_x) / gain, y1 = ((cy - hh) - pad_y) / gain;
        float x2 = ((cx + hw) - pad_x) / gain, y2 = ((cy + hh) - pad_y) / gain;
        float* r = rows + (size_t)kk * row_stride;
        r[0] = fminf(fmaxf(x1, 0.0f), w0); r[1] = fminf(fmaxf(y1, 0.0f), h0);
        r[2] = fminf(fmaxf(x2, 0.0f), w0); r[3] = fminf(fmaxf(y2, 0.0f), h0);
        r[4] = score; r[5] = (float)w.cand_cls[a];
        for (int e = 0; e < n_extra; ++e) r[6 + e] = pred[(size_t)(4 + nc + e) * N + a];
        keep[kk] = a;
    }
}

#define NMS_REST_LDS (NMS_MAX_CAND * 8 + 16 * 64 * 4 * 4 + 16 * 64 * 4 + 1024 * 4)

int ss_front_init()
{
    hipError_t e = hipFuncSetAttribute((const void*)k_nms_sort, hipFuncAttributeMaxDynamicSharedMemorySize, NMS_MAX_CAND * 8);
    hipError_t e2 = hipFuncSetAttribute((const void*)k_nms_rest, hipFuncAttributeMaxDynamicSharedMemorySize, NMS_REST_LDS);
    return (e == hipSuccess && e2 == hipSuccess) ? 0 : 1;
}

// error flags (counters[1]) of the first `units` workspace units, for ss_check_errors
int* ss_nms_error_flag(void* ws, int unit) { return carve_nms((char*)ws + (size_t)unit * ss_nms_workspace_bytes()).counters + 1; }

int ss_nms_fused = 1;      // 1: filter + one workgroup per image for the rest; 0: filter, sort, mask, scan as four launches (ss_set_option "nms_fused")

int ss_launch_nms(const float* pred, int batch, long long pred_stride, int N, int nc, int n_extra, float conf, float iou,
                  int agnostic, float max_wh, int max_det, float gain, float pad_x, float pad_y, float w0, float h0,
                  const float* geom, float* rows, int row_stride, long long rows_batch_stride, int* keep,
                  long long keep_batch_stride, int* count, void* ws, size_t ws_bytes, unsigned long long cm0,
                  unsigned long long cm1, hipStream_t st)
{
    if (batch <= 0) return 0;
    if (N > NMS_MAX_ANCHORS || ws_bytes < ss_nms_workspace_bytes() * (size_t)batch) return SS_ERR_CAPACITY;
    NmsBatch nb{ pred, pred_stride, (char*)ws, (long long)ss_nms_workspace_bytes() };
    // the candidate counter is re-armed by k_nms_scan itself (no memset node: graph-capture safe)
    hipLaunchKernelGGL(k_nms_filter, dim3((N + 63) / 64, batch), dim3(512), 0, st, nb, N, nc, conf, cm0, cm1);
    if (ss_nms_fused) {
        hipLaunchKernelGGL(k_nms_rest, dim3(batch), dim3(1024), NMS_REST_LDS, st, nb, N, nc, n_extra, agnostic, max_wh, iou, max_det, gain,
                           pad_x, pad_y, w0, h0, geom, rows, row_stride, rows_batch_stride, keep, keep_batch_stride, count);
    } else {
        hipLaunchKernelGGL(k_nms_sort, dim3(batch), dim3(1024), NMS_MAX_CAND * 8, st, nb, N, agnostic, max_wh);
        hipLaunchKernelGGL(k_nms_mask, dim3(NMS_MASK_WG, batch), dim3(64), 0, st, nb, iou);
        hipLaunchKernelGGL(k_nms_scan, dim3(batch), dim3(64), 0, st, nb, N, nc, n_extra, max_det, gain, pad_x, pad_y, w0, h0,
                           geom, rows, row_stride, rows_batch_stride, keep, keep_batch_stride, count);
    }
    return 0;
}
