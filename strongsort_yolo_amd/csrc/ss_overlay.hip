// ss_overlay.hip — annotation overlay on device frames: boxes, label plates and text, keypoint dots,
// trajectory lines, the blended class-count plate.  Replaces the cv2 drawing of /root/reference/yolo_multi_model.py:58-162
// and :311-331 (per-box rectangle / putText / circle / line calls and the addWeighted count plate), which the reference
// runs on the CPU on a copy of every frame.
//
// A frame's annotation is an ORDERED list of primitives (painter's algorithm: a later primitive overwrites an earlier
// one).  The kernel is pixel-centric so that the order is kept without atomics: one 32x8 pixel tile per workgroup; per
// chunk of 256 primitives every thread tests one primitive's bounding box against the tile, the hits are compacted in
// order into LDS, and every pixel walks the hits in order with exact integer coverage tests (no floating point, so the
// NumPy rasteriser used as the oracle reproduces every pixel).  Pixels no primitive covers are not touched.
#include "ss_common.h"

#define OV_RECT 0      // outline of box (x0,y0)-(x1,y1), thickness a centred on the box edge
#define OV_FILL 1      // filled rectangle, corners in any order
#define OV_CIRCLE 2    // filled circle, centre (x0,y0), radius a
#define OV_LINE 3      // segment (x0,y0)-(x1,y1), thickness a (round caps)
#define OV_TEXT 4      // 5x7 raster text: baseline-left origin (x0,y0), x1 = characters, a = offset into the character buffer, b>>1 = scale
#define OV_POLY 5      // even-odd interior of a closed integer polygon, BLENDED half and half (ties to even) with what is there — the
                       // reference's fillPoly + addWeighted(0.5, 0.5) per mask (yolo_multi_model.py:116-121): (x0,y0)-(x1,y1) = bounding box,
                       // a = byte offset (multiple of 4) of the b>>1 vertices (int32 x, y pairs) in the character buffer
// b bit 0: the primitive belongs to a blended group: consecutive group primitives are composited opaquely among
// themselves and the result is mixed 179:77 (0.7 : 0.3) over what was there before the group (cv2.addWeighted plate)

struct OvPrim { int type, x0, y0, x1, y1, color, a, b; };

__device__ __forceinline__ void ov_bbox(const OvPrim& p, int& bx0, int& by0, int& bx1, int& by1)
{
    switch (p.type) {
    case OV_RECT: { const int o = p.a >> 1; bx0 = min(p.x0, p.x1) - o; by0 = min(p.y0, p.y1) - o; bx1 = max(p.x0, p.x1) + o; by1 = max(p.y0, p.y1) + o; break; }
    case OV_FILL: bx0 = min(p.x0, p.x1); by0 = min(p.y0, p.y1); bx1 = max(p.x0, p.x1); by1 = max(p.y0, p.y1); break;
    case OV_CIRCLE: bx0 = p.x0 - p.a; by0 = p.y0 - p.a; bx1 = p.x0 + p.a; by1 = p.y0 + p.a; break;
    case OV_LINE: { const int o = (p.a + 1) >> 1; bx0 = min(p.x0, p.x1) - o; by0 = min(p.y0, p.y1) - o; bx1 = max(p.x0, p.x1) + o; by1 = max(p.y0, p.y1) + o; break; }
    case OV_POLY: bx0 = p.x0; by0 = p.y0; bx1 = p.x1; by1 = p.y1; break;
    default: { const int sc = max(p.b >> 1, 1); bx0 = p.x0; bx1 = p.x0 + p.x1 * 6 * sc - 1; by1 = p.y0; by0 = p.y0 - 7 * sc + 1; break; }
    }
}

__device__ __forceinline__ bool ov_covers(const OvPrim& p, int x, int y, const uint8_t* __restrict__ chars,
                                          const uint8_t* __restrict__ font)
{
    switch (p.type) {
    case OV_RECT: {
        const int X0 = min(p.x0, p.x1), X1 = max(p.x0, p.x1), Y0 = min(p.y0, p.y1), Y1 = max(p.y0, p.y1);
        const int o = p.a >> 1, in = (p.a + 1) >> 1;
        if (x < X0 - o || x > X1 + o || y < Y0 - o || y > Y1 + o) return false;
        return !(x >= X0 + in && x <= X1 - in && y >= Y0 + in && y <= Y1 - in);
    }
    case OV_FILL:
        return x >= min(p.x0, p.x1) && x <= max(p.x0, p.x1) && y >= min(p.y0, p.y1) && y <= max(p.y0, p.y1);
    case OV_CIRCLE: {
        const int dx = x - p.x0, dy = y - p.y0;
        return dx * dx + dy * dy <= p.a * p.a;
    }
    case OV_LINE: {
        // distance to the segment <= a/2, in integers: 4 d^2 <= a^2
        const long long ax = p.x0, ay = p.y0, dx = p.x1 - p.x0, dy = p.y1 - p.y0, px = x - ax, py = y - ay;
        const long long L = dx * dx + dy * dy, t = px * dx + py * dy, a2 = (long long)p.a * p.a;
        if (t <= 0) return 4 * (px * px + py * py) <= a2;
        if (t >= L) { const long long qx = x - p.x1, qy = y - p.y1; return 4 * (qx * qx + qy * qy) <= a2; }
        const long long cr = px * dy - py * dx;
        return 4 * cr * cr <= a2 * L;
    }
    case OV_POLY: {
        // a pixel is inside when an odd number of edges straddle its row (exactly one end point with y <= the pixel's) strictly to
        // its right: x < ax + (y - ay)(bx - ax) / (by - ay), cross-multiplied in 64-bit integers (oracle polygon_mask_np)
        if (x < p.x0 || x > p.x1 || y < p.y0 || y > p.y1) return false;
        const int n = p.b >> 1;
        if (n < 3) return false;                                   // a degenerate polygon covers nothing (and has no last vertex to read)
        const int* pts = reinterpret_cast<const int*>(chars + p.a);
        int cnt = 0, ax = pts[2 * (n - 1)], ay = pts[2 * (n - 1) + 1];
        for (int i = 0; i < n; ++i) {
            const int bx = pts[2 * i], by = pts[2 * i + 1];
            if ((ay <= y) != (by <= y)) {
                const long long dy = (long long)by - ay, lhs = ((long long)x - ax) * dy, rhs = ((long long)y - ay) * ((long long)bx - ax);
                cnt += (dy > 0) ? (lhs < rhs) : (lhs > rhs);
            }
            ax = bx; ay = by;
        }
        return cnt & 1;
    }
    default: {
        const int sc = max(p.b >> 1, 1);
        const int cx = x - p.x0, cy = y - (p.y0 - 7 * sc + 1);
        if (cx < 0 || cy < 0 || cy >= 7 * sc || cx >= p.x1 * 6 * sc) return false;
        const int ci = cx / (6 * sc), col = (cx - ci * 6 * sc) / sc, row = cy / sc;
        if (col >= 5) return false;
        int ch = chars[p.a + ci];
        if (ch < 32 || ch > 126) ch = '?';
        return (font[(ch - 32) * 5 + col] >> row) & 1;
    }
    }
}

__device__ __forceinline__ int ov_mix(int top, int under)        // 0.7 : 0.3 in 8-bit fixed point (179 : 77), per channel
{
    int out = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int t = (top >> (8 * c)) & 255, u = (under >> (8 * c)) & 255;
        out |= ((179 * t + 77 * u + 128) >> 8) << (8 * c);
    }
    return out;
}

__device__ __forceinline__ int ov_half(int a, int b)             // (a + b) / 2 per channel, ties to even (cv2.addWeighted 0.5 / 0.5 + saturate_cast)
{
    int out = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int t = ((a >> (8 * c)) & 255) + ((b >> (8 * c)) & 255);
        out |= ((t >> 1) + ((t & 1) & ((t >> 1) & 1))) << (8 * c);
    }
    return out;
}

// grid = (ceil(W/32), ceil(H/8), batch); prim_off[b] .. prim_off[b+1] = primitives of frame b
__global__ __launch_bounds__(256) void k_overlay(uint8_t* __restrict__ frames, long long frame_stride, int H, int W, int row_stride,
                                                 const OvPrim* __restrict__ prims, const int* __restrict__ prim_off,
                                                 const uint8_t* __restrict__ chars, const uint8_t* __restrict__ font)
{
    __shared__ OvPrim hits[256];
    __shared__ int wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 8;
    const int x = tx0 + (tid & 31), y = ty0 + (tid >> 5);
    const bool inside = x < W && y < H;
    uint8_t* px = frames + (size_t)blockIdx.z * frame_stride + (size_t)y * row_stride + (size_t)x * 3;
    const int p0 = prim_off[blockIdx.z], p1 = prim_off[blockIdx.z + 1];
    int cur = 0, grp = 0;
    bool touched = false, in_grp = false, loaded = false;
    for (int base = p0; base < p1; base += 256) {
        // ordered compaction of the primitives of this chunk whose bounding box meets the tile
        OvPrim p;
        bool hit = false;
        if (base + tid < p1) {
            p = prims[base + tid];
            int bx0, by0, bx1, by1;
            ov_bbox(p, bx0, by0, bx1, by1);
            hit = bx1 >= tx0 && bx0 < tx0 + 32 && by1 >= ty0 && by0 < ty0 + 8;
        }
        const unsigned long long m = __ballot(hit);
        __syncthreads();                                   // the previous chunk's readers are done with hits / wtot
        if (lane == 0) wtot[wv] = __popcll(m);
        __syncthreads();
        int off = 0, n = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int c = wtot[i]; if (i < wv) off += c; n += c; }
        if (hit) hits[off + __popcll(m & ((1ull << lane) - 1ull))] = p;
        __syncthreads();
        if (inside)
            for (int i = 0; i < n; ++i) {
                const OvPrim& q = hits[i];
                if (!ov_covers(q, x, y, chars, font)) continue;
                if (!loaded) { cur = px[0] | (px[1] << 8) | (px[2] << 16); loaded = true; }
                if (q.type == OV_POLY) { if (in_grp) { cur = ov_mix(grp, cur); in_grp = false; } cur = ov_half(cur, q.color); }
                else if (q.b & 1) { grp = q.color; in_grp = true; }
                else { if (in_grp) { cur = ov_mix(grp, cur); in_grp = false; } cur = q.color; }
                touched = true;
            }
    }
    if (touched) {
        if (in_grp) cur = ov_mix(grp, cur);
        px[0] = (uint8_t)cur; px[1] = (uint8_t)(cur >> 8); px[2] = (uint8_t)(cur >> 16);
    }
}

void ss_launch_overlay(uint8_t* frames, int batch, long long frame_stride, int h, int w, int row_stride, const void* prims,
                       const int* prim_off, const uint8_t* chars, const uint8_t* font, hipStream_t st)
{
    if (batch <= 0 || h <= 0 || w <= 0) return;
    hipLaunchKernelGGL(k_overlay, dim3((w + 31) / 32, (h + 7) / 8, batch), dim3(256), 0, st, frames, frame_stride, h, w, row_stride,
                       (const OvPrim*)prims, prim_off, chars, font);
}
