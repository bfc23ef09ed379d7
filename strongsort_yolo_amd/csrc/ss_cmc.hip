// ss_cmc.hip — camera-motion compensation: down-scaled grey frames + ECC alignment of consecutive frames.
// Not in the reference snapshot; stands where upstream StrongSORT calls tracker.camera_update(prev, cur) ahead of
// tracker.predict() inside model.track (/root/reference/yolo_multi_model.py:41).  Arithmetic = oracle so_gray_small /
// so_ecc (oracle/csrc/ss_oracle.c, DECISIONS D-18) operation for operation, float64, same reduction tree -> bit-identical
// warps.  The warps are stateless per frame pair, so a whole frame group is estimated in one launch (one 1024-thread
// workgroup per pair) beside the detector; k_frame applies warp f to the track boxes before predicting frame f.
#include "ss_common.h"

__device__ inline void cm_axis(int d, float scale, int n_src, int& i0, int& i1, float& frac)
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
__device__ inline float cm_bilerp_u8(float p00, float p01, float p10, float p11, float fx, float fy)
{
    float a = fmaf(fx, p01 - p00, p00);
    float b = fmaf(fx, p11 - p10, p10);
    float v = fmaf(fy, b - a, a);
    float q = floorf(v + 0.5f);
    return fminf(fmaxf(q, 0.0f), 255.0f);
}
__device__ inline float cm_grey(const uint8_t* p) { return (float)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14); }

// grid = (ceil(ws*hs / 256), images); image b at src + b*src_stride, output b at dst + b*dst_stride
__global__ __launch_bounds__(256) void k_gray_small(const uint8_t* __restrict__ src, long long src_stride, int H, int W, int stride,
                                                    uint8_t* __restrict__ dst, long long dst_stride, int hs, int ws)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hs * ws) return;
    src += (size_t)blockIdx.y * src_stride;
    const int y = p / ws, x = p - y * ws;
    const float sx = (float)W / (float)ws, sy = (float)H / (float)hs;
    int y0, y1, x0, x1; float fy, fx;
    cm_axis(y, sy, H, y0, y1, fy);
    cm_axis(x, sx, W, x0, x1, fx);
    const uint8_t *r0 = src + (size_t)y0 * stride, *r1 = src + (size_t)y1 * stride;
    dst[(size_t)blockIdx.y * dst_stride + p] =
        (uint8_t)cm_bilerp_u8(cm_grey(r0 + x0 * 3), cm_grey(r0 + x1 * 3), cm_grey(r1 + x0 * 3), cm_grey(r1 + x1 * 3), fx, fy);
}

__device__ inline void cm_sincos(double t, double& s, double& c)
{
    const double t2 = t * t;
    double ps = -1.0 / 1307674368000.0;
    ps = fma(ps, t2, 1.0 / 6227020800.0);
    ps = fma(ps, t2, -1.0 / 39916800.0);
    ps = fma(ps, t2, 1.0 / 362880.0);
    ps = fma(ps, t2, -1.0 / 5040.0);
    ps = fma(ps, t2, 1.0 / 120.0);
    ps = fma(ps, t2, -1.0 / 6.0);
    ps = fma(ps, t2, 1.0);
    double pc = 1.0 / 20922789888000.0;
    pc = fma(pc, t2, -1.0 / 87178291200.0);
    pc = fma(pc, t2, 1.0 / 479001600.0);
    pc = fma(pc, t2, -1.0 / 3628800.0);
    pc = fma(pc, t2, 1.0 / 40320.0);
    pc = fma(pc, t2, -1.0 / 720.0);
    pc = fma(pc, t2, 1.0 / 24.0);
    pc = fma(pc, t2, -0.5);
    pc = fma(pc, t2, 1.0);
    s = t * ps; c = pc;
}

__device__ inline double cm_px(const uint8_t* __restrict__ I, int ws, int hs, int x, int y)
{
    if (x < 0) x = 0; if (x > ws - 1) x = ws - 1;
    if (y < 0) y = 0; if (y > hs - 1) y = hs - 1;
    return (double)I[y * ws + x];
}

__device__ inline bool cm_sample(const uint8_t* __restrict__ I, int ws, int hs, double c, double s, double tx, double ty, int x, int y,
                                 double& iw, double& gx, double& gy)
{
    const double xw = (c * (double)x - s * (double)y) + tx, yw = (s * (double)x + c * (double)y) + ty;
    if (!(xw >= 0.0 && xw <= (double)(ws - 1) && yw >= 0.0 && yw <= (double)(hs - 1))) return false;
    int x0 = (int)xw, y0 = (int)yw;
    if (x0 > ws - 2) x0 = ws - 2; if (y0 > hs - 2) y0 = hs - 2;
    if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0;
    const double fx = xw - (double)x0, fy = yw - (double)y0;
    double v[3][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xi = x0 + (k & 1), yi = y0 + (k >> 1);
        v[0][k] = cm_px(I, ws, hs, xi, yi);
        v[1][k] = (cm_px(I, ws, hs, xi + 1, yi) - cm_px(I, ws, hs, xi - 1, yi)) * 0.5;
        v[2][k] = (cm_px(I, ws, hs, xi, yi + 1) - cm_px(I, ws, hs, xi, yi - 1)) * 0.5;
    }
    double o[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double a = v[q][0] + fx * (v[q][1] - v[q][0]);
        const double b = v[q][2] + fx * (v[q][3] - v[q][2]);
        o[q] = a + fy * (b - a);
    }
    iw = o[0]; gx = o[1]; gy = o[2];
    return true;
}

#define CM_NS 15
// block-wide sums in the oracle's order: 64-lane xor butterflies, then the 16 wave sums added left to right
template <int NS>
__device__ inline void cm_reduce(double (&a)[NS], double (*red)[CM_NS], double (&out)[NS])
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        double p = a[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off);
        a[k] = p;
    }
    __syncthreads();                                     // the previous reduction's readers are done with red
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) red[wv][k] = a[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        double tot = red[0][k];
        for (int g = 1; g < 16; ++g) tot = tot + red[g][k];
        out[k] = tot;
    }
}

// one workgroup per (frame f, stream s): template = small image f of the stream (index 0 = the last frame of the previous
// group), image = small image f + 1.  warps[(f*S+s)*8 + 0..5] = 2x3 matrix (translation in full-frame pixels),
// [6] = iterations run, or -1: no usable alignment / no previous frame (identity stored, k_frame skips it).
__global__ __launch_bounds__(1024) void k_ecc(const uint8_t* __restrict__ smalls, long long img_stride, int S, int hs, int ws,
                                              int max_iter, double eps, double scale_x, double scale_y,
                                              const int* __restrict__ prev_valid, const int* __restrict__ n_valid,
                                              double* __restrict__ warps)
{
    __shared__ double red[16][CM_NS];
    const int f = blockIdx.x / S, s = blockIdx.x - f * S, tid = threadIdx.x;
    const uint8_t* __restrict__ T = smalls + ((size_t)f * S + s) * img_stride;
    const uint8_t* __restrict__ I = smalls + ((size_t)(f + 1) * S + s) * img_stride;
    double* out = warps + (size_t)blockIdx.x * 8;
    const int npx = hs * ws;
    double theta = 0.0, tx = 0.0, ty = 0.0, last_rho = -2.0;
    int it, status = 0;
    if (f == 0 && !prev_valid[s]) status = -1;
    if (n_valid && f >= *n_valid) status = -1;            // frames past the real ones of a partial group
    for (it = 1; status == 0 && it <= max_iter; ++it) {
        double s_, c_;
        cm_sincos(theta, s_, c_);
        double a3[3] = { 0.0, 0.0, 0.0 }, s3[3];
        for (int p = tid; p < npx; p += 1024) {
            const int y = p / ws, x = p - y * ws;
            double iw, gx, gy;
            if (!cm_sample(I, ws, hs, c_, s_, tx, ty, x, y, iw, gx, gy)) continue;
            a3[0] = a3[0] + 1.0; a3[1] = a3[1] + iw; a3[2] = a3[2] + (double)T[p];
        }
        cm_reduce<3>(a3, red, s3);
        if (!(s3[0] >= 64.0)) { status = -1; break; }
        const double mI = s3[1] / s3[0], mT = s3[2] / s3[0];
        double a[CM_NS], sum[CM_NS];
#pragma unroll
        for (int k = 0; k < CM_NS; ++k) a[k] = 0.0;
        for (int p = tid; p < npx; p += 1024) {
            const int y = p / ws, x = p - y * ws;
            double iwv, gx, gy;
            if (!cm_sample(I, ws, hs, c_, s_, tx, ty, x, y, iwv, gx, gy)) continue;
            const double iw = iwv - mI, tz = (double)T[p] - mT;
            const double hx = -((double)x * s_) - (double)y * c_, hy = (double)x * c_ - (double)y * s_;
            const double j0 = gx * hx + gy * hy, j1 = gx, j2 = gy;
            a[0] = a[0] + j0 * j0; a[1] = a[1] + j0 * j1; a[2] = a[2] + j0 * j2;
            a[3] = a[3] + j1 * j1; a[4] = a[4] + j1 * j2; a[5] = a[5] + j2 * j2;
            a[6] = a[6] + j0 * iw; a[7] = a[7] + j1 * iw; a[8] = a[8] + j2 * iw;
            a[9] = a[9] + j0 * tz; a[10] = a[10] + j1 * tz; a[11] = a[11] + j2 * tz;
            a[12] = a[12] + tz * iw; a[13] = a[13] + iw * iw; a[14] = a[14] + tz * tz;
        }
        cm_reduce<CM_NS>(a, red, sum);
        const double h00 = sum[0], h01 = sum[1], h02 = sum[2], h11 = sum[3], h12 = sum[4], h22 = sum[5];
        const double c00 = h11 * h22 - h12 * h12, c01 = h02 * h12 - h01 * h22, c02 = h01 * h12 - h02 * h11;
        const double c11 = h00 * h22 - h02 * h02, c12 = h01 * h02 - h00 * h12, c22 = h00 * h11 - h01 * h01;
        const double det = (h00 * c00 + h01 * c01) + h02 * c02;
        const double ni = sum[13], nt = sum[14], corr = sum[12];
        if (!(det != 0.0) || !(ni > 0.0) || !(nt > 0.0)) { status = -1; break; }
        const double rho = corr / (sqrt(ni) * sqrt(nt));
        if (!(rho == rho)) { status = -1; break; }
        if (it > 1 && fabs(rho - last_rho) < eps) break;
        last_rho = rho;
        const double i00 = c00 / det, i01 = c01 / det, i02 = c02 / det, i11 = c11 / det, i12 = c12 / det, i22 = c22 / det;
        const double ip0 = sum[6], ip1 = sum[7], ip2 = sum[8], tp0 = sum[9], tp1 = sum[10], tp2 = sum[11];
        const double q0 = (i00 * ip0 + i01 * ip1) + i02 * ip2, q1 = (i01 * ip0 + i11 * ip1) + i12 * ip2, q2 = (i02 * ip0 + i12 * ip1) + i22 * ip2;
        const double lam_n = ni - ((ip0 * q0 + ip1 * q1) + ip2 * q2), lam_d = corr - ((tp0 * q0 + tp1 * q1) + tp2 * q2);
        if (!(lam_d > 0.0)) { status = -1; break; }
        const double lam = lam_n / lam_d;
        const double e0 = lam * tp0 - ip0, e1 = lam * tp1 - ip1, e2 = lam * tp2 - ip2;
        theta = theta + ((i00 * e0 + i01 * e1) + i02 * e2);
        tx = tx + ((i01 * e0 + i11 * e1) + i12 * e2);
        ty = ty + ((i02 * e0 + i12 * e1) + i22 * e2);
    }
    if (tid == 0) {
        if (status < 0) {
            out[0] = 1.0; out[1] = 0.0; out[2] = 0.0; out[3] = 0.0; out[4] = 1.0; out[5] = 0.0; out[6] = -1.0; out[7] = 0.0;
        } else {
            double s_, c_;
            cm_sincos(theta, s_, c_);
            out[0] = c_; out[1] = -s_; out[2] = tx * scale_x; out[3] = s_; out[4] = c_; out[5] = ty * scale_y;
            out[6] = (double)(it > max_iter ? max_iter : it); out[7] = 0.0;
        }
    }
}

// the group's last small images become "previous" for the next call (a kernel, not a memcpy node: graph-capture safe)
// (n_valid: device count of real frames in the group, NULL = n_frames; the last REAL frame is the next call's predecessor,
// and a group without real frames leaves the remembered one alone)
__global__ __launch_bounds__(256) void k_cmc_roll(uint8_t* __restrict__ smalls, int n_frames, const int* __restrict__ n_valid, size_t bytes,
                                                  int* __restrict__ prev_valid, int S)
{
    int n = n_valid ? *n_valid : n_frames;
    if (n > n_frames) n = n_frames;
    if (n <= 0) return;
    const size_t last_off = (size_t)n * bytes;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i * 16 < bytes) reinterpret_cast<uint4*>(smalls)[i] = reinterpret_cast<const uint4*>(smalls + last_off)[i];
    if (i < (size_t)S) prev_valid[i] = 1;
}

void ss_launch_cmc(const uint8_t* frames, int n_images, long long frame_stride, int h, int w, int row_stride, uint8_t* smalls,
                   long long img_stride, int S, int n_frames, int hs, int ws, int max_iter, double eps, int* prev_valid,
                   const int* n_valid, double* warps, hipStream_t st)
{
    // small images 1 .. n_frames of the buffer (index 0 holds the previous group's last frame)
    hipLaunchKernelGGL(k_gray_small, dim3((hs * ws + 255) / 256, n_images), dim3(256), 0, st, frames, frame_stride, h, w, row_stride,
                       smalls + (size_t)S * img_stride, img_stride, hs, ws);
    hipLaunchKernelGGL(k_ecc, dim3(n_frames * S), dim3(1024), 0, st, smalls, img_stride, S, hs, ws, max_iter, eps,
                       (double)w / (double)ws, (double)h / (double)hs, prev_valid, n_valid, warps);
    const size_t bytes = (size_t)S * img_stride;                       // img_stride is a multiple of 16
    hipLaunchKernelGGL(k_cmc_roll, dim3((unsigned)((bytes / 16 + 255) / 256 + (S + 255) / 256)), dim3(256), 0, st, smalls,
                       n_frames, n_valid, bytes, prev_valid, S);
}
