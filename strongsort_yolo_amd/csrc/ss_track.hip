// ss_track.hip — device-resident StrongSORT tracker update for S independent streams.
//
// Per frame three launches cover rows a6..a10 of SURVEY.md §8(a) for every stream at once:
//   k_pre     one block per stream: Kalman predict of all live tracks (+ Cholesky of the projected
//             covariance for the gate, confirmed-track list); further blocks: L2-normalise the
//             detection embeddings and write them row-major and fragment-major.
//   k_cosine  one 8-wave block per (confirmed track, 32-row gallery tile): streams the gallery
//             tile with coalesced 16-B loads straight into v_mfma_f32_32x32x2_f32 operands
//             (one k-segment per wave), combines the segments in LDS, min over gallery rows
//             (registers -> wave shuffle -> LDS).  This is the HBM-bound association kernel.
//   k_step    one block per stream: gate/blend cost matrix into LDS, LSAP (single wave, SciPy's
//             scan order), IoU stage + LSAP, Kalman/EMA updates, births, deletions, gallery append,
//             output rows.  No host round trip anywhere in the frame.
#include <hip/hip_ext.h>
#include "ss_common.h"

// =================================================================================================
// k_pre
// =================================================================================================
__device__ inline void block_scan256(int flag, int* wtot /*LDS[4]*/, int& pos, int& total)
{
    unsigned long long m = __ballot(flag);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inwave = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();                       // protect wtot from the previous scan's readers
    if (lane == 0) wtot[w] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { int c = wtot[i]; if (i < w) off += c; tot += c; }
    pos = off + inwave;
    total = tot;
}

// Write one 512-float row (LDS or global) as row i of a fragment-major tile (see ss_frag_index).
// lane l owns block q = l/2 (k = 16q..16q+15) and k-slots 2*(l&1), 2*(l&1)+1.
__device__ __forceinline__ void frag_write_row(float4* tile, int i, const float* row, bool zero)
{
    const int l = threadIdx.x & 63, q = l >> 1, ks0 = 2 * (l & 1);
    float r[16];
    if (zero) {
#pragma unroll
        for (int k = 0; k < 16; ++k) r[k] = 0.0f;
    } else {
        const float4* s4 = reinterpret_cast<const float4*>(row + 16 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) { float4 v = s4[j]; r[4 * j] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w; }
    }
    tile[(q * 4 + ks0) * 16 + i] = make_float4(r[ks0], r[4 + ks0], r[8 + ks0], r[12 + ks0]);
    tile[(q * 4 + ks0 + 1) * 16 + i] = make_float4(r[ks0 + 1], r[5 + ks0], r[9 + ks0], r[13 + ks0]);
}

// exclusive prefix sum of small non-negative ints over the 256 threads of a block
__device__ inline void block_scan_sum256(int v, int* wtot /*LDS[4]*/, int& excl, int& total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    __syncthreads();
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { int c = wtot[i]; if (i < w) off += c; tot += c; }
    excl = off + inc - v;
    total = tot;
}

__global__ __launch_bounds__(256) void k_pre(SSDev dev, SSParams prm)
{
    __shared__ int wtot[4];
    __shared__ int tile_base;
    __shared__ __attribute__((aligned(16))) float rowbuf[4][SS_F];
    const int s = blockIdx.x;
    const int tid = threadIdx.x;
    if (blockIdx.y == 0) {
        // ---- predict all live tracks of stream s (thread = position in the track list) ----
        const int nT = dev.n_tracks[s];
        int confirmed = 0, myslot = 0, mycount = 0;
        if (tid < nT) {
            const int slot = dev.order[s * SS_MAXT + tid];
            myslot = slot;
            const size_t g = (size_t)s * SS_MAXT + slot;
            double mean[8], cov[64];
#pragma unroll
            for (int i = 0; i < 8; ++i) mean[i] = dev.mean[g * 8 + i];
#pragma unroll
            for (int i = 0; i < 64; ++i) cov[i] = dev.cov[g * 64 + i];
            ss_kf_predict(mean, cov, prm.wp, prm.wv);
#pragma unroll
            for (int i = 0; i < 8; ++i) dev.mean[g * 8 + i] = mean[i];
#pragma unroll
            for (int i = 0; i < 64; ++i) dev.cov[g * 64 + i] = cov[i];
            dev.age[g] += 1;
            dev.tsu[g] += 1;
            dev.det_idx[g] = -1;
            // gate factorisation (projection with conf = 0) and predicted box, by track index
            double m4[4], S[16], L[16];
            ss_kf_project(mean, cov, 0.0, prm.wp, m4, S);
            ss_chol4(S, L);
            double* ch = dev.chol + ((size_t)s * SS_MAXT + tid) * 16;
            ch[0] = L[0]; ch[1] = L[4]; ch[2] = L[5]; ch[3] = L[8]; ch[4] = L[9]; ch[5] = L[10];
            ch[6] = L[12]; ch[7] = L[13]; ch[8] = L[14]; ch[9] = L[15];
            ch[10] = m4[0]; ch[11] = m4[1]; ch[12] = m4[2]; ch[13] = m4[3];
            double w = mean[2] * mean[3];
            double* tb = dev.ttlwh + ((size_t)s * SS_MAXT + tid) * 4;
            tb[0] = mean[0] - w / 2; tb[1] = mean[1] - mean[3] / 2; tb[2] = w; tb[3] = mean[3];
            confirmed = dev.state[g] == SS_CONFIRMED;
            mycount = dev.gal_count[g];
        }
        int pos, total;
        block_scan256(confirmed, wtot, pos, total);
        if (confirmed) dev.conf_list[s * SS_MAXT + pos] = tid;
        // association work list: one entry per (confirmed track, 32-row gallery tile) that has rows
        const int D = dev.n_dets[s];
        const int ntile = (confirmed && D > 0) ? (mycount + SS_TILE - 1) / SS_TILE : 0;
        const int tlist = (dev.stream_mode && D <= 2 * SS_TILE) ? 0 : 1;
        int toff, ttot;
        block_scan_sum256(ntile, wtot, toff, ttot);
        if (tid == 0) {
            dev.n_conf[s] = total;
            tile_base = ttot ? atomicAdd(dev.tile_count + tlist, ttot) + tlist * dev.S * SS_MAXT * SS_NRT : 0;
            if (total > dev.grid_tracks) dev.err[s] = SS_ERR_CAPACITY;
        }
        __syncthreads();
        for (int rt = 0; rt < ntile; ++rt)
            dev.tiles[tile_base + toff + rt] = make_int4(s, pos, myslot, mycount | (rt << 8) | (D << 16));
        return;
    }
    // ---- detection prep: one wave per detection ----
    const int w = tid >> 6, l = tid & 63;
    const int d = (blockIdx.y - 1) * 4 + w;
    const int D = dev.n_dets[s];
    const int Dpad = (D + SS_TILE - 1) / SS_TILE * SS_TILE;
    if (d >= Dpad) return;
    float4* frag = reinterpret_cast<float4*>(dev.feat_frag + ((size_t)s * SS_NCT + d / SS_TILE) * SS_TILE_FLOATS);
    const int jj = d % SS_TILE;
    if (d < D) {
        const float* raw = dev.feats_raw + ((size_t)s * SS_MAXD + d) * SS_F;
        float v[8], a = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = raw[l + 64 * j]; a = fmaf(v[j], v[j], a); }
        float n = sqrtf(ss_wave_sumsq_reduce(a));
        float* unit = dev.feat_unit + ((size_t)s * SS_MAXD + d) * SS_F;
#pragma unroll
        for (int j = 0; j < 8; ++j) { float u = n > 0.0f ? v[j] / n : 0.0f; unit[l + 64 * j] = u; rowbuf[w][l + 64 * j] = u; }   // all-zero row stays zero (D-17)
        SS_WAVE_SYNC();
        frag_write_row(frag, jj, rowbuf[w], false);
        if (l == 0) {
            const float* b = dev.dets + ((size_t)s * SS_MAXD + d) * 6;
            double x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
            double bw = x2 - x1, bh = y2 - y1;
            double* t = dev.tlwh + ((size_t)s * SS_MAXD + d) * 4;
            double* z = dev.xyah + ((size_t)s * SS_MAXD + d) * 4;
            t[0] = x1; t[1] = y1; t[2] = bw; t[3] = bh;
            z[0] = x1 + bw / 2; z[1] = y1 + bh / 2; z[2] = bw / bh; z[3] = bh;
        }
    } else {
        frag_write_row(frag, jj, nullptr, true);
    }
}

// =================================================================================================
// k_cosine — the association kernel (gallery stream + f32 MFMA + min over gallery rows)
// =================================================================================================
// part_min[s][r][rt][d] = min over the valid rows b of gallery tile rt of (1 - g_b . f_d).
// Shared by the tracker (gallery addressed through conf_list/order) and the KAT entry point
// (tracks addressed directly): slot_of(r) abstracts that.
struct CosineArgs {
    const float* gallery;       // fragment-major tiles, [track][NRT][TILE_FLOATS]
    const int* gal_count;       // per track
    const float* feat_frag;     // [NCT][TILE_FLOATS] of this stream
    float* part_min;            // [r][NRT][MAXD]
    int n_rows;                 // tracks to process
    int D;
};

// A operand of one (track, 16-row tile) for this wave's k-segment: 4 coalesced 16-B loads per lane.
// Lanes whose gallery row is not valid do not touch memory (the ragged last tile costs only its rows).
__device__ __forceinline__ void load_a(const float4* __restrict__ tile, int valid_rows, float4 a[4])
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const bool ok = (l & 15) < valid_rows;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = ok ? tile[(4 * w + j) * 64 + l] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// workgroup barrier that orders LDS traffic only: outstanding global loads (the next tile's prefetch) stay in
// flight across it (__syncthreads() would drain vmcnt to 0 here)
#define SS_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#define SS_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// B operand (detections) of column tiles ct, ct+1 for this wave's k-segment
__device__ __forceinline__ void load_b(const float* __restrict__ feat_frag, int ct, bool two, float4 b0[4], float4 b1[4])
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const float4* fb0 = reinterpret_cast<const float4*>(feat_frag + (size_t)ct * SS_TILE_FLOATS);
    const float4* fb1 = reinterpret_cast<const float4*>(feat_frag + (size_t)(ct + (two ? 1 : 0)) * SS_TILE_FLOATS);
#pragma unroll
    for (int j = 0; j < 4; ++j) { b0[j] = fb0[(4 * w + j) * 64 + l]; b1[j] = fb1[(4 * w + j) * 64 + l]; }
}

// One pair of column tiles: 16 + 16 MFMAs (two independent accumulation chains), then
//   LDS: per-segment partial tiles -> summed left to right (oracle order) -> 1 - dot -> min over rows.
__device__ __forceinline__ void cosine_pair(const float4 a[4], const float4 b0[4], const float4 b1[4], int count, int rt,
                                            int ct, bool two, int D, float* __restrict__ out, float* lds_part,
                                            float* lds_red)
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    f32x4 acc0 = { 0.f, 0.f, 0.f, 0.f }, acc1 = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc0 = SS_MFMA16(a[j].x, b0[j].x, acc0); acc1 = SS_MFMA16(a[j].x, b1[j].x, acc1);
        acc0 = SS_MFMA16(a[j].y, b0[j].y, acc0); acc1 = SS_MFMA16(a[j].y, b1[j].y, acc1);
        acc0 = SS_MFMA16(a[j].z, b0[j].z, acc0); acc1 = SS_MFMA16(a[j].z, b1[j].z, acc1);
        acc0 = SS_MFMA16(a[j].w, b0[j].w, acc0); acc1 = SS_MFMA16(a[j].w, b1[j].w, acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        lds_part[((0 * 8 + w) * 4 + r) * 64 + l] = acc0[r];
        lds_part[((1 * 8 + w) * 4 + r) * 64 + l] = acc1[r];
    }
    SS_LDS_BARRIER();
    // thread (ctl, reg, lane): sum the 8 segment partials of one accumulator element, left to right
    const int ctl = threadIdx.x >> 8, reg = (threadIdx.x >> 6) & 3;
    float tot = lds_part[((ctl * 8 + 0) * 4 + reg) * 64 + l];
#pragma unroll
    for (int sg = 1; sg < SS_NSEG; ++sg) tot = tot + lds_part[((ctl * 8 + sg) * 4 + reg) * 64 + l];
    float m = 1.0f - tot;
    const int row = rt * SS_TILE + 4 * (l >> 4) + reg;             // C/D layout of 16x16x4: row = 4*(lane/16)+reg
    if (row >= count) m = INFINITY;
    m = fminf(m, __shfl_xor(m, 16));
    m = fminf(m, __shfl_xor(m, 32));
    if (l < 16) lds_red[(ctl * 4 + reg) * 16 + l] = m;
    SS_LDS_BARRIER();
    if (threadIdx.x < 32) {
        const int c2 = threadIdx.x >> 4, j = threadIdx.x & 15;
        float f = fminf(fminf(lds_red[(c2 * 4 + 0) * 16 + j], lds_red[(c2 * 4 + 1) * 16 + j]),
                        fminf(lds_red[(c2 * 4 + 2) * 16 + j], lds_red[(c2 * 4 + 3) * 16 + j]));
        const int d = (ct + c2) * SS_TILE + j;
        if ((c2 == 0 || two) && d < D) out[d] = f;
    }
}

// One gallery tile against every detection column tile (generic: B re-read from L2 per pair).
__device__ __forceinline__ void cosine_tile(const float4 a[4], int count, int rt, const float* __restrict__ feat_frag,
                                            int D, float* __restrict__ out, float* lds_part, float* lds_red)
{
    const int nct = (D + SS_TILE - 1) / SS_TILE;
    for (int ct = 0; ct < nct; ct += 2) {
        float4 b0[4], b1[4];
        load_b(feat_frag, ct, ct + 1 < nct, b0, b1);
        cosine_pair(a, b0, b1, count, rt, ct, ct + 1 < nct, D, out, lds_part, lds_red);
    }
}

// ---- association kernel, latency form (work list 1) ------------------------------------------------
// One 8-wave workgroup per gallery tile, one k-segment per wave: the shortest critical path for a
// single stream (a tile's 256 MFMAs are spread over 8 waves).  Persistent over chunks of the list with a
// register prefetch of the next tile; used for small batches and for D > 32.
__global__ __launch_bounds__(512) void k_cosine_wg(SSDev dev)
{
    __shared__ float lds_part[2 * 8 * 4 * 64];
    __shared__ float lds_red[2 * 4 * 16];
    const int4* tiles = dev.tiles + (size_t)dev.S * SS_MAXT * SS_NRT;
    const int G = gridDim.x;
    const int4 spec = tiles[blockIdx.x];             // speculative: right whenever ntiles <= G (chunk == 1)
    const int ntiles = dev.tile_count[1];
    const int chunk = (ntiles + G - 1) / G;
    const int t0 = blockIdx.x * chunk, t1 = min(t0 + chunk, ntiles);
    if (t0 >= t1) return;
    auto tile_ptr = [&](int s, int slot, int rt) {
        return reinterpret_cast<const float4*>(dev.gallery + (((size_t)s * SS_MAXT + slot) * SS_NRT + rt) * SS_TILE_FLOATS);
    };
    int4 v0 = (chunk == 1) ? spec : tiles[t0];
    int s0 = __builtin_amdgcn_readfirstlane(v0.x), r0 = __builtin_amdgcn_readfirstlane(v0.y);
    int sl0 = __builtin_amdgcn_readfirstlane(v0.z), cw0 = __builtin_amdgcn_readfirstlane(v0.w);
    int s1 = s0, r1 = r0, sl1 = sl0, cw1 = cw0;
    if (t0 + 1 < t1) {
        int4 v1 = tiles[t0 + 1];
        s1 = __builtin_amdgcn_readfirstlane(v1.x); r1 = __builtin_amdgcn_readfirstlane(v1.y);
        sl1 = __builtin_amdgcn_readfirstlane(v1.z); cw1 = __builtin_amdgcn_readfirstlane(v1.w);
    }
    float4 a_cur[4], a_nxt[4], b0[4], b1[4];
    int sB = -1;
    load_a(tile_ptr(s0, sl0, (cw0 >> 8) & 7), (cw0 & 0xff) - ((cw0 >> 8) & 7) * SS_TILE, a_cur);
#pragma unroll
    for (int j = 0; j < 4; ++j) { a_nxt[j] = a_cur[j]; b0[j] = a_cur[j]; b1[j] = a_cur[j]; }
    for (int t = t0; t < t1; ++t) {
        int4 v2 = make_int4(s1, r1, sl1, cw1);
        if (t + 2 < t1) v2 = tiles[t + 2];                                   // consumed at the end of the iteration
        const int count = cw0 & 0xff, rt = (cw0 >> 8) & 7, D = cw0 >> 16;
        const int nct = (D + SS_TILE - 1) / SS_TILE;
        const float* ff = dev.feat_frag + (size_t)s0 * SS_NCT * SS_TILE_FLOATS;
        float* out = dev.part_min + (((size_t)s0 * SS_MAXT + r0) * SS_NRT + rt) * SS_MAXD;
        if (nct <= 2) {
            if (sB != s0) { load_b(ff, 0, nct == 2, b0, b1); sB = s0; }      // before the prefetch: vmcnt is in-order
            if (t + 1 < t1)
                load_a(tile_ptr(s1, sl1, (cw1 >> 8) & 7), (cw1 & 0xff) - ((cw1 >> 8) & 7) * SS_TILE, a_nxt);
            cosine_pair(a_cur, b0, b1, count, rt, 0, nct == 2, D, out, lds_part, lds_red);
        } else {
            sB = -1;
            cosine_tile(a_cur, count, rt, ff, D, out, lds_part, lds_red);
            if (t + 1 < t1)
                load_a(tile_ptr(s1, sl1, (cw1 >> 8) & 7), (cw1 & 0xff) - ((cw1 >> 8) & 7) * SS_TILE, a_nxt);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) a_cur[j] = a_nxt[j];
        s0 = s1; r0 = r1; sl0 = sl1; cw0 = cw1;
        s1 = __builtin_amdgcn_readfirstlane(v2.x); r1 = __builtin_amdgcn_readfirstlane(v2.y);
        sl1 = __builtin_amdgcn_readfirstlane(v2.z); cw1 = __builtin_amdgcn_readfirstlane(v2.w);
    }
}

// ---- association kernel, throughput form (work list 0, D <= 32) ------------------------------------
// One WAVE per gallery tile: the wave walks the tile's 8 k-segments itself (8 x 16 MFMAs per column
// tile), adds the segment sums left to right in registers and reduces min-over-rows with two shuffles.
// No LDS combine, no barriers in steady state, waves fully decoupled.  The gallery is one continuous
// stream of 4-KiB segment pieces per wave, prefetched 3 pieces ahead through a 4-deep register ring
// (ordinary loads, so hipcc's counted vmcnt keeps 3 pieces in flight); the stream's detection operand
// B (2 x 32 KiB fragment tiles) sits in LDS, shared by the 8 waves of the workgroup.
#define SS_TS(idx) do { if (dev.ts_enable && blockIdx.x % 32 == 0 && blockIdx.x / 32 < 16 && (threadIdx.x & 63) == 0 && (idx) < 64) \
        dev.ts[((blockIdx.x / 32) * 8 + (threadIdx.x >> 6)) * 64 + (idx)] = wall_clock64(); } while (0)

__global__ __launch_bounds__(512) void k_cosine_stream(SSDev dev)
{
    int tsi = 0;
    SS_TS(tsi++);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* bl = reinterpret_cast<float4*>(smem);                  // [2][32][64] float4 = 64 KiB
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int G = gridDim.x;
    const int ntiles = dev.tile_count[0];
    const int chunk = (ntiles + G - 1) / G;
    const int c0 = blockIdx.x * chunk, c1 = min(c0 + chunk, ntiles);
    if (c0 >= c1) return;
    int4* desc = reinterpret_cast<int4*>(smem + 2 * SS_TILE_FLOATS * 4);     // [512] descriptors of this pass
    for (int p0 = c0; p0 < c1; p0 += 512) {
    const int p1 = min(p0 + 512, c1);
    __syncthreads();
    if (p0 + (int)threadIdx.x < p1) desc[threadIdx.x] = dev.tiles[p0 + threadIdx.x];   // one parallel read, then LDS only
    __syncthreads();
    SS_TS(tsi++);
    const int4* tiles = desc - p0;                                           // tiles[t] for t in [p0, p1)
    int t0 = p0;
    while (t0 < p1) {
        // sub-range [t0, t1) of the pass that belongs to one stream
        const int s = __builtin_amdgcn_readfirstlane(tiles[t0].x);
        const int D = __builtin_amdgcn_readfirstlane(tiles[t0].w) >> 16;
        int t1 = t0 + 1;
        for (;;) {                                                           // 64 descriptors per step
            const int tt = t1 + l;
            const unsigned long long diff = __ballot(tt < p1 && tiles[tt].x != s);
            if (diff) { t1 += __builtin_ctzll(diff); break; }
            if (t1 + 64 >= p1) { t1 = p1; break; }
            t1 += 64;
        }
        const bool two = D > SS_TILE;
        // this wave's tiles: t0 + w, t0 + w + 8, ...
        const int wu = __builtin_amdgcn_readfirstlane(w);
        const int nmine = (t1 - t0 - wu + 7) / 8;                    // may be <= 0
        // descriptors live in SGPRs; a tile is addressed as (uniform byte base) + (per-lane offset)
        auto rd = [&](int t, int& r, int& cw, const char*& base) {
            const int4 v = tiles[t];
            const int sl = __builtin_amdgcn_readfirstlane(v.z);
            r = __builtin_amdgcn_readfirstlane(v.y); cw = __builtin_amdgcn_readfirstlane(v.w);
            base = reinterpret_cast<const char*>(dev.gallery) +
                   ((((size_t)s * SS_MAXT + sl) * SS_NRT + ((cw >> 8) & 7)) * SS_TILE_FLOATS) * 4;
        };
        // lanes of rows past the gallery count re-read row 0 of the tile (same cache lines, no extra HBM
        // traffic, no select); those rows are masked to +inf below
        auto lane_off = [&](int cw) {
            const bool ok = (l & 15) < (cw & 0xff) - ((cw >> 8) & 7) * SS_TILE;
            return (unsigned)((ok ? l : (l & ~15)) * 16);
        };
        auto ld = [&](const char* base, unsigned voff, int sg, float4 a[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const float4*>(base + voff + sg * 4096 + j * 1024);
        };
        int r0 = 0, cw0 = 0, r1 = 0, cw1 = 0;
        const char *base0 = reinterpret_cast<const char*>(dev.gallery), *base1 = base0;
        unsigned vo0 = 0, vo1 = 0;
        float4 ra[4][4];                                              // 4-deep ring of segment pieces
        if (nmine > 0) {
            // gallery pieces 0..2 of the first tile go on the wire BEFORE the B staging below
            rd(t0 + wu, r0, cw0, base0);
            vo0 = lane_off(cw0);
            ld(base0, vo0, 0, ra[0]); ld(base0, vo0, 1, ra[1]); ld(base0, vo0, 2, ra[2]);
        }
        // stage B of stream s in LDS
        __syncthreads();
        {
            const float4* ff = reinterpret_cast<const float4*>(dev.feat_frag + (size_t)s * SS_NCT * SS_TILE_FLOATS);
            // all loads first, then all LDS writes: one memory latency instead of one per 8 KiB slice
            float4 tmp[8];
            const int nu = two ? 8 : 4;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (u < nu) tmp[u] = ff[threadIdx.x + 512 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (u < nu) bl[threadIdx.x + 512 * u] = tmp[u];
        }
        __syncthreads();
        SS_TS(tsi++);
        if (nmine > 0) {
            const char* bls = reinterpret_cast<const char*>(bl) + l * 16;
            const char* bls1 = bls + (two ? 32768 : 0);          // single column tile: read tile 0 twice (result unused)
            for (int k = 0; k < nmine; ++k) {
                const bool more = k + 1 < nmine;
                // the prefetch below is unconditional (static load count -> exact counted vmcnt, no drain at the tile
                // boundary); the last tile of a wave prefetches its own first pieces again, which nobody reads
                if (more) { rd(t0 + wu + 8 * (k + 1), r1, cw1, base1); vo1 = lane_off(cw1); }
                else { base1 = base0; vo1 = vo0; }
                f32x4 tot0 = { 0.f, 0.f, 0.f, 0.f }, tot1 = { 0.f, 0.f, 0.f, 0.f };
                SS_TS(tsi++);
#pragma unroll
                for (int sg = 0; sg < 8; ++sg) {
                    // prefetch piece sg+3 (possibly of the next tile) into ring slot (sg+3)%4
                    if (sg + 3 < 8) ld(base0, vo0, sg + 3, ra[(sg + 3) & 3]);
                    else ld(base1, vo1, sg + 3 - 8, ra[(sg + 3) & 3]);
                    const float4* a = ra[sg & 3];
                    f32x4 acc0 = { 0.f, 0.f, 0.f, 0.f }, acc1 = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 b0 = *reinterpret_cast<const float4*>(bls + (4 * sg + j) * 1024);
                        const float4 b1 = *reinterpret_cast<const float4*>(bls1 + (4 * sg + j) * 1024);
                        acc0 = SS_MFMA16(a[j].x, b0.x, acc0); acc1 = SS_MFMA16(a[j].x, b1.x, acc1);
                        acc0 = SS_MFMA16(a[j].y, b0.y, acc0); acc1 = SS_MFMA16(a[j].y, b1.y, acc1);
                        acc0 = SS_MFMA16(a[j].z, b0.z, acc0); acc1 = SS_MFMA16(a[j].z, b1.z, acc1);
                        acc0 = SS_MFMA16(a[j].w, b0.w, acc0); acc1 = SS_MFMA16(a[j].w, b1.w, acc1);
                    }
                    if (sg == 0) { tot0 = acc0; tot1 = acc1; }
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { tot0[r] = tot0[r] + acc0[r]; tot1[r] = tot1[r] + acc1[r]; }
                    }
                    __builtin_amdgcn_sched_barrier(0);      // keep the B fragments of later segments out of this one
                    if (sg == 0 || sg == 3) SS_TS(tsi++);
                }
                SS_TS(tsi++);
                // 1 - dot, mask rows past the gallery count, min over the tile's 16 rows
                const int count = cw0 & 0xff, rt = (cw0 >> 8) & 7;
                float m0 = INFINITY, m1 = INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool valid = rt * SS_TILE + 4 * (l >> 4) + r < count;
                    m0 = fminf(m0, valid ? 1.0f - tot0[r] : INFINITY);
                    m1 = fminf(m1, valid ? 1.0f - tot1[r] : INFINITY);
                }
                m0 = fminf(m0, __shfl_xor(m0, 16)); m0 = fminf(m0, __shfl_xor(m0, 32));
                m1 = fminf(m1, __shfl_xor(m1, 16)); m1 = fminf(m1, __shfl_xor(m1, 32));
                float* out = dev.part_min + (((size_t)s * SS_MAXT + r0) * SS_NRT + rt) * SS_MAXD;
                if (l < 16) { if (l < D) out[l] = m0; }
                else if (l < 32) { if (l < D) out[l] = m1; }
                r0 = r1; cw0 = cw1; base0 = base1; vo0 = vo1;
            }
        }
        t0 = t1;
    }
    }
}

__global__ __launch_bounds__(512) void k_cosine_kat(CosineArgs a)
{
    __shared__ float lds_part[2 * 8 * 4 * 64];
    __shared__ float lds_red[2 * 4 * 16];
    const int rt = blockIdx.x, r = blockIdx.y;
    const int count = a.gal_count[r];
    if (rt * SS_TILE >= count) return;
    float4 av[4];
    load_a(reinterpret_cast<const float4*>(a.gallery + ((size_t)r * SS_NRT + rt) * SS_TILE_FLOATS), count - rt * SS_TILE, av);
    cosine_tile(av, count, rt, a.feat_frag, a.D, a.part_min + ((size_t)r * SS_NRT + rt) * SS_MAXD, lds_part, lds_red);
}

// =================================================================================================
// LSAP on one wave — shortest augmenting path in SciPy's scan order (oracle so_lsap)
// =================================================================================================
struct LsapLds {
    double *u, *v, *sp;                          // [256] each
    int *path, *row4col, *col4row, *remaining;   // [256] each
    unsigned char *SR, *SC;                      // [256] each
};

// ---- wave-64 reductions on the DPP path (gfx9 row shifts + row broadcasts: an inclusive scan whose lane 63 holds
// the reduction; ~6 VALU steps instead of 6 LDS-crossbar shuffles) ----------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v, unsigned long long identity)
{
    const int lo = __builtin_amdgcn_update_dpp((int)(identity & 0xffffffffu), (int)(v & 0xffffffffu), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(identity >> 32), (int)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long m)
{
    const unsigned long long ID = ~0ull;
    unsigned long long o;
    o = dpp_u64<0x111, 0xf>(m, ID); m = o < m ? o : m;          // row_shr:1
    o = dpp_u64<0x112, 0xf>(m, ID); m = o < m ? o : m;          // row_shr:2
    o = dpp_u64<0x114, 0xf>(m, ID); m = o < m ? o : m;          // row_shr:4
    o = dpp_u64<0x118, 0xf>(m, ID); m = o < m ? o : m;          // row_shr:8
    o = dpp_u64<0x142, 0xa>(m, ID); m = o < m ? o : m;          // row_bcast:15 -> rows 1,3
    o = dpp_u64<0x143, 0xc>(m, ID); m = o < m ? o : m;          // row_bcast:31 -> rows 2,3
    const int lo = __builtin_amdgcn_readlane((int)(m & 0xffffffffu), 63), hi = __builtin_amdgcn_readlane((int)(m >> 32), 63);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

template <bool MAX>
__device__ __forceinline__ int wave_minmax_i32(int p)
{
    const int ID = MAX ? (int)0x80000000 : 0x7fffffff;
    int o;
#define SS_STEP(CTRL, RM) o = __builtin_amdgcn_update_dpp(ID, p, CTRL, RM, 0xf, false); p = MAX ? max(p, o) : min(p, o)
    SS_STEP(0x111, 0xf); SS_STEP(0x112, 0xf); SS_STEP(0x114, 0xf); SS_STEP(0x118, 0xf); SS_STEP(0x142, 0xa); SS_STEP(0x143, 0xc);
#undef SS_STEP
    return __builtin_amdgcn_readlane(p, 63);
}

// ---- register-resident form for nr <= nc <= 64 (the common case: <= 64 tracks x <= 64 detections) --------
// lane j owns column j (v, shortest path cost, path, row4col, position in SciPy's `remaining` list), lane i owns
// row i (u, col4row).  One LDS read (the cost entry) per scan step; the arg-min is a 64-bit unsigned wave-min of an
// order-preserving image of the double, ties resolved with ballots by scan position exactly as the sequential code
// does (last unassigned column among the minima, else the first minimum).
__device__ __forceinline__ unsigned long long ss_f64_key(double v)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(v + 0.0);      // +0.0: -0 and +0 share a key
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__device__ inline int lsap_wave_small(int nr, int nc, const double* cost, const LsapLds& L)
{
    const int l = threadIdx.x & 63;
    double u = 0.0, v = 0.0;
    int c4r = -1, r4c = -1, path = -1;
    const unsigned long long KINF = ss_f64_key(INFINITY);
    for (int cur = 0; cur < nr; ++cur) {
        int pos = nc - 1 - l;
        bool active = l < nc, scj = false, sr = false;
        double sp = INFINITY, minVal = 0.0;
        int num_remaining = nc, sink = -1, i = cur;
        while (sink == -1) {
            if (l == i) sr = true;
            const double ui = __longlong_as_double(__builtin_amdgcn_readlane((int)(__double_as_longlong(u) & 0xffffffff), i) & 0xffffffffll |
                                                   ((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(u) >> 32), i) << 32));
            if (active) {
                const double r = minVal + cost[i * nc + l] - ui - v;
                if (r < sp) { path = i; sp = r; }
            }
            const unsigned long long key = active ? ss_f64_key(sp) : ~0ull;
            const unsigned long long m = wave_min_u64(key);
            if (m >= KINF) return -1;                                              // infeasible (or nothing active)
            const unsigned long long tied = __ballot(active && key == m);
            int w;
            if (__popcll(tied) == 1) w = __builtin_ctzll(tied);
            else {
                const unsigned long long tu = __ballot(active && key == m && r4c == -1);
                const bool wantmax = tu != 0ull;
                const bool insel = active && key == m && (!wantmax || r4c == -1);
                const int p0 = insel ? pos : (wantmax ? -1 : 0x3fffffff);
                const int p = wantmax ? wave_minmax_i32<true>(p0) : wave_minmax_i32<false>(p0);
                w = __builtin_ctzll(__ballot(insel && pos == p));
            }
            w = __builtin_amdgcn_readfirstlane(w);
            {
                const long long bits = __double_as_longlong(sp);
                minVal = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(bits >> 32), w) << 32) |
                                              ((long long)__builtin_amdgcn_readlane((int)(bits & 0xffffffff), w) & 0xffffffffll));
            }
            const int rj = __builtin_amdgcn_readlane(r4c, w);
            const int pj = __builtin_amdgcn_readlane(pos, w);
            if (rj == -1) sink = w; else i = rj;
            --num_remaining;
            if (active && l != w && pos == num_remaining) pos = pj;               // remaining[index] = remaining[--n]
            if (l == w) { active = false; scj = true; }
        }
        // dual variables
        const int src = c4r >= 0 ? c4r : 0;
        const double spc = __shfl(sp, src);
        if (l == cur) u += minVal;
        else if (sr) u += minVal - spc;
        if (scj) v -= minVal - sp;
        // augment along the path (uniform walk; every step is a pair of readlanes)
        int j = sink;
        for (;;) {
            const int r = __builtin_amdgcn_readlane(path, j);
            if (l == j) r4c = r;
            const int t = __builtin_amdgcn_readlane(c4r, r);
            if (l == r) c4r = j;
            j = t;
            if (r == cur) break;
        }
    }
    if (l < nr) L.col4row[l] = c4r;
    SS_WAVE_SYNC();
    return 0;
}

// ---- register-resident form for 64 < nc <= 64*Q: lane l owns columns l, l+64, ... and rows l, l+64, ... -----------
// Same algorithm and tie rule as lsap_wave_small; indexed accesses use a uniform (lane, q) split and statically
// unrolled selects so that the per-column arrays stay in registers.
template <int Q> __device__ __forceinline__ double rl_f64(const double (&a)[Q], int idx)
{
    const int ln = idx & 63, qi = idx >> 6;
    double out = 0.0;
#pragma unroll
    for (int q = 0; q < Q; ++q)
        if (q == qi) {
            const long long b = __double_as_longlong(a[q]);
            out = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(b >> 32), ln) << 32) |
                                       ((long long)__builtin_amdgcn_readlane((int)(b & 0xffffffff), ln) & 0xffffffffll));
        }
    return out;
}
template <int Q> __device__ __forceinline__ int rl_i32(const int (&a)[Q], int idx)
{
    const int ln = idx & 63, qi = idx >> 6;
    int out = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) if (q == qi) out = __builtin_amdgcn_readlane(a[q], ln);
    return out;
}

template <int Q>
__device__ inline int lsap_wave_regs(int nr, int nc, const double* cost, const LsapLds& L)
{
    const int l = threadIdx.x & 63;
    double u[Q], v[Q], sp[Q];
    int c4r[Q], r4c[Q], path[Q], pos[Q];
    bool active[Q], scj[Q], sr[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { u[q] = 0.0; v[q] = 0.0; c4r[q] = -1; r4c[q] = -1; path[q] = -1; }
    const unsigned long long KINF = ss_f64_key(INFINITY);
    for (int cur = 0; cur < nr; ++cur) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int j = l + 64 * q;
            pos[q] = nc - 1 - j; active[q] = j < nc; scj[q] = false; sr[q] = false; sp[q] = INFINITY;
        }
        double minVal = 0.0;
        int num_remaining = nc, sink = -1, i = cur;
        while (sink == -1) {
#pragma unroll
            for (int q = 0; q < Q; ++q) if (l + 64 * q == i) sr[q] = true;
            const double ui = rl_f64<Q>(u, i);
            unsigned long long key[Q], kmin = ~0ull;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (active[q]) {
                    const double r = minVal + cost[i * nc + l + 64 * q] - ui - v[q];
                    if (r < sp[q]) { path[q] = i; sp[q] = r; }
                }
                key[q] = active[q] ? ss_f64_key(sp[q]) : ~0ull;
                kmin = key[q] < kmin ? key[q] : kmin;
            }
            const unsigned long long m = wave_min_u64(kmin);
            if (m >= KINF) return -1;
            unsigned long long tied[Q];
            int ntied = 0;
#pragma unroll
            for (int q = 0; q < Q; ++q) { tied[q] = __ballot(active[q] && key[q] == m); ntied += __popcll(tied[q]); }
            int w = 0;
            if (ntied == 1) {
#pragma unroll
                for (int q = 0; q < Q; ++q) if (tied[q]) w = 64 * q + __builtin_ctzll(tied[q]);
            } else {
                bool wantmax = false;
#pragma unroll
                for (int q = 0; q < Q; ++q) wantmax = wantmax || __ballot(active[q] && key[q] == m && r4c[q] == -1) != 0ull;
                int p0 = wantmax ? -1 : 0x3fffffff;
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const bool insel = active[q] && key[q] == m && (!wantmax || r4c[q] == -1);
                    if (insel) p0 = wantmax ? max(p0, pos[q]) : min(p0, pos[q]);
                }
                const int pbest = wantmax ? wave_minmax_i32<true>(p0) : wave_minmax_i32<false>(p0);
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const bool insel = active[q] && key[q] == m && (!wantmax || r4c[q] == -1);
                    const unsigned long long b = __ballot(insel && pos[q] == pbest);
                    if (b) w = 64 * q + __builtin_ctzll(b);
                }
            }
            w = __builtin_amdgcn_readfirstlane(w);
            minVal = rl_f64<Q>(sp, w);
            const int rj = rl_i32<Q>(r4c, w);
            const int pj = rl_i32<Q>(pos, w);
            if (rj == -1) sink = w; else i = rj;
            --num_remaining;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int j = l + 64 * q;
                if (active[q] && j != w && pos[q] == num_remaining) pos[q] = pj;
                if (j == w) { active[q] = false; scj[q] = true; }
            }
        }
        // dual variables: u[r] += minVal - sp[col4row[r]] (r in SR, r != cur); v[j] -= minVal - sp[j] (j in SC)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int c = c4r[q] >= 0 ? c4r[q] : 0;
            double spc = 0.0;
#pragma unroll
            for (int qq = 0; qq < Q; ++qq) {
                const double t = __shfl(sp[qq], c & 63);
                if ((c >> 6) == qq) spc = t;
            }
            if (l + 64 * q == cur) u[q] += minVal;
            else if (sr[q]) u[q] += minVal - spc;
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) if (scj[q]) v[q] -= minVal - sp[q];
        // augment
        int j = sink;
        for (;;) {
            const int r = rl_i32<Q>(path, j);
#pragma unroll
            for (int q = 0; q < Q; ++q) if (l + 64 * q == j) r4c[q] = r;
            const int t = rl_i32<Q>(c4r, r);
#pragma unroll
            for (int q = 0; q < Q; ++q) if (l + 64 * q == r) c4r[q] = j;
            j = t;
            if (r == cur) break;
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) if (l + 64 * q < nr) L.col4row[l + 64 * q] = c4r[q];
    SS_WAVE_SYNC();
    return 0;
}

// cost: [nr][nc] (nr <= nc <= 256) in LDS or global.  Result col4row[0..nr).  Returns 0 / -1.
__device__ inline int lsap_wave(int nr, int nc, const double* cost, const LsapLds& L)
{
    if (nc <= 64) return lsap_wave_small(nr, nc, cost, L);
    if (nc <= 128) return lsap_wave_regs<2>(nr, nc, cost, L);
    if (nc <= 256) return lsap_wave_regs<4>(nr, nc, cost, L);
    return -1;                       // callers cap both dimensions at 256 (SS_MAX_TRACKS)
}

__device__ inline LsapLds carve_lsap(char*& p)
{
    LsapLds L;
    L.u = (double*)p; p += 256 * 8; L.v = (double*)p; p += 256 * 8; L.sp = (double*)p; p += 256 * 8;
    L.path = (int*)p; p += 256 * 4; L.row4col = (int*)p; p += 256 * 4;
    L.col4row = (int*)p; p += 256 * 4; L.remaining = (int*)p; p += 256 * 4;
    L.SR = (unsigned char*)p; p += 256; L.SC = (unsigned char*)p; p += 256;
    return L;
}

// stand-alone LSAP (KAT entry point): one wave, cost read from global memory
__global__ __launch_bounds__(64) void k_lsap_kat(const double* cost, int nr0, int nc0, int* row_to_col,
                                                  double* scratch_t, int* err)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* p = smem;
    LsapLds L = carve_lsap(p);
    const int l = threadIdx.x;
    const bool tr = nc0 < nr0;
    const int nr = tr ? nc0 : nr0, nc = tr ? nr0 : nc0;
    for (int i = l; i < nr0; i += 64) row_to_col[i] = -1;
    if (nr == 0) return;
    const double* c = cost;
    if (tr) {
        for (int idx = l; idx < nr0 * nc0; idx += 64) { int i = idx / nc0, j = idx % nc0; scratch_t[j * nr0 + i] = cost[idx]; }
        __threadfence_block();
        __syncthreads();
        c = scratch_t;
    }
    int rc = lsap_wave(nr, nc, c, L);
    if (rc) { if (l == 0) *err = SS_ERR_INFEASIBLE; return; }
    for (int i = l; i < nr; i += 64) {
        if (tr) row_to_col[L.col4row[i]] = i; else row_to_col[i] = L.col4row[i];
    }
}

// =================================================================================================
// k_step — per-stream association, assignment and bookkeeping
// =================================================================================================
__device__ inline void ema_wave(const float* smooth_in, const float* feat, float a, float b, float* out)
{
    const int l = threadIdx.x & 63;
    float v[8], acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t1 = a * smooth_in[l + 64 * j];
        float t2 = b * feat[l + 64 * j];
        v[j] = t1 + t2;
        acc = fmaf(v[j], v[j], acc);
    }
    float n = sqrtf(ss_wave_sumsq_reduce(acc));
#pragma unroll
    for (int j = 0; j < 8; ++j) out[l + 64 * j] = n > 0.0f ? v[j] / n : 0.0f;
}

// append row-major unit row `src` (global) as gallery row b of a track (fragment-major tiles)
__device__ inline void gallery_append_wave(float* gal_track, int b, const float* src)
{
    frag_write_row(reinterpret_cast<float4*>(gal_track + (size_t)(b / SS_TILE) * SS_TILE_FLOATS), b % SS_TILE, src, false);
}

#define SS_TS_STEP(idx) do { if (dev.ts_enable && blockIdx.x == 0 && threadIdx.x == 0) dev.ts[(15 * 8 + 7) * 64 + (idx)] = wall_clock64(); } while (0)

__global__ __launch_bounds__(256) void k_step(SSDev dev, SSParams prm)
{
    SS_TS_STEP(0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* p = smem;
    double* cost = (double*)p; p += (size_t)SS_COST_CAP * 8;
    LsapLds L = carve_lsap(p);
    int* matchdet = (int*)p; p += 256 * 4;     // by track index: matched detection or -1
    int* dettrk = (int*)p; p += 256 * 4;       // by detection: matched track index or -1
    int* asg = (int*)p; p += 256 * 4;          // LSAP result by original row
    int* cand = (int*)p; p += 256 * 4;         // IoU-stage rows (track indices)
    int* cols = (int*)p; p += 256 * 4;         // IoU-stage columns (detection indices)
    int* neworder = (int*)p; p += 256 * 4;
    int* freelist = (int*)p; p += 256 * 4;
    int* conf_l = (int*)p; p += 256 * 4;
    int* wtot = (int*)p; p += 16;
    int* flags = (int*)p; p += 16;

    const int s = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const int nT = dev.n_tracks[s], D = dev.n_dets[s], nC = dev.n_conf[s];
    const size_t sb = (size_t)s * SS_MAXT;
    const size_t db = (size_t)s * SS_MAXD;
    matchdet[tid] = -1; dettrk[tid] = -1; asg[tid] = -1;
    if (tid < nC) conf_l[tid] = dev.conf_list[sb + tid];
    if (tid == 0) { flags[0] = 0; }
    int myslot = (tid < nT) ? dev.order[sb + tid] : -1;
    int mystate = (tid < nT) ? dev.state[sb + myslot] : 0;
    int mytsu = (tid < nT) ? dev.tsu[sb + myslot] : 0;
    if (prm.debug) { dev.dbg_lists[(s * 4 + 0) * SS_MAXT + tid] = -1; dev.dbg_lists[(s * 4 + 3) * SS_MAXT + tid] = -1; }
    __syncthreads();

    // ---------------- stage A: appearance + motion cost, LSAP --------------------------------
    if (nC > 0 && D > 0) {
        const bool tr = D < nC;
        const int nr = tr ? D : nC, nc = tr ? nC : D;
        if (nr * nc > SS_COST_CAP) { if (tid == 0) dev.err[s] = SS_ERR_CAPACITY; }
        else {
            for (int idx = tid; idx < nC * D; idx += 256) {
                const int r = idx / D, d = idx % D;
                const int ti = conf_l[r];
                const int slot = dev.order[sb + ti];
                const int count = dev.gal_count[sb + slot];
                const float* pm = dev.part_min + ((sb + r) * SS_NRT) * SS_MAXD + d;
                float c = pm[0];
                for (int rt = 1; rt * SS_TILE < count; ++rt) c = fminf(c, pm[(size_t)rt * SS_MAXD]);
                const double* ch = dev.chol + (sb + ti) * 16;
                double Lm[16] = { ch[0], 0, 0, 0, ch[1], ch[2], 0, 0, ch[3], ch[4], ch[5], 0, ch[6], ch[7], ch[8], ch[9] };
                double m4[4] = { ch[10], ch[11], ch[12], ch[13] };
                const double* zz = dev.xyah + (db + d) * 4;
                double z[4] = { zz[0], zz[1], zz[2], zz[3] };
                double maha = ss_maha(Lm, m4, z);
                int g;
                double v = ss_blend(c, maha, prm, &g);
                cost[tr ? d * nc + r : r * nc + d] = v;
                if (prm.debug) {
                    size_t o = (sb + r) * SS_MAXD + d;
                    dev.dbg_cos[o] = c; dev.dbg_maha[o] = maha; dev.dbg_gated[o] = (uint8_t)g; dev.dbg_cost_a[o] = v;
                }
            }
            __syncthreads();
            SS_TS_STEP(1);
            if (wave == 0) {
                int rc = lsap_wave(nr, nc, cost, L);
                if (rc) { if (tid == 0) dev.err[s] = SS_ERR_INFEASIBLE; }
                else for (int i = tid; i < nr; i += 64) { if (tr) asg[L.col4row[i]] = i; else asg[i] = L.col4row[i]; }
            }
            __syncthreads();
            SS_TS_STEP(2);
            if (tid < nC) {
                int d = asg[tid];
                if (d >= 0) {
                    double v = cost[tr ? d * nc + tid : tid * nc + d];
                    if (!(v > prm.max_dist)) { matchdet[conf_l[tid]] = d; dettrk[d] = conf_l[tid]; }
                    else d = -1;
                }
                if (prm.debug) dev.dbg_lists[(s * 4 + 0) * SS_MAXT + tid] = d;
            }
        }
    }
    __syncthreads();

    // ---------------- stage B: IoU association ----------------------------------------------
    SS_TS_STEP(3);
    int pos, nU, nC1, nCols;
    const int isU = (tid < nT) && (mystate != SS_CONFIRMED);
    block_scan256(isU, wtot, pos, nU);
    if (isU) cand[pos] = tid;
    const int isC1 = (tid < nT) && (mystate == SS_CONFIRMED) && (matchdet[tid] < 0) && (mytsu == 1);
    block_scan256(isC1, wtot, pos, nC1);
    if (isC1) cand[nU + pos] = tid;
    const int nCand = nU + nC1;
    const int isCol = (tid < D) && (dettrk[tid] < 0);
    block_scan256(isCol, wtot, pos, nCols);
    if (isCol) cols[pos] = tid;
    asg[tid] = -1;
    __syncthreads();
    if (prm.debug) {
        if (tid < nCand) dev.dbg_lists[(s * 4 + 1) * SS_MAXT + tid] = cand[tid];
        if (tid < nCols) dev.dbg_lists[(s * 4 + 2) * SS_MAXT + tid] = cols[tid];
        if (tid == 0) { int* c = dev.dbg_counts + s * 4; c[0] = nC; c[1] = nCand; c[2] = nCols; c[3] = D; }
    }
    if (nCand > 0 && nCols > 0) {
        const bool tr = nCols < nCand;
        const int nr = tr ? nCols : nCand, nc = tr ? nCand : nCols;
        if (nr * nc > SS_COST_CAP) { if (tid == 0) dev.err[s] = SS_ERR_CAPACITY; }
        else {
            for (int idx = tid; idx < nCand * nCols; idx += 256) {
                const int r = idx / nCols, c = idx % nCols;
                const int ti = cand[r];
                const double* tb = dev.ttlwh + (sb + ti) * 4;
                const double* cb = dev.tlwh + (db + cols[c]) * 4;
                double t[4] = { tb[0], tb[1], tb[2], tb[3] }, cc[4] = { cb[0], cb[1], cb[2], cb[3] };
                double v = (dev.tsu[sb + dev.order[sb + ti]] > 1) ? prm.max_iou_distance + 1e-5
                                                                : ss_iou_cost(t, cc, prm.max_iou_distance);
                cost[tr ? c * nc + r : r * nc + c] = v;
                if (prm.debug) dev.dbg_cost_b[(sb + r) * SS_MAXD + c] = v;
            }
            __syncthreads();
            if (wave == 0) {
                int rc = lsap_wave(nr, nc, cost, L);
                if (rc) { if (tid == 0) dev.err[s] = SS_ERR_INFEASIBLE; }
                else for (int i = tid; i < nr; i += 64) { if (tr) asg[L.col4row[i]] = i; else asg[i] = L.col4row[i]; }
            }
            __syncthreads();
            if (tid < nCand) {
                int c = asg[tid];
                if (c >= 0) {
                    double v = cost[tr ? c * nc + tid : tid * nc + c];
                    if (!(v > prm.max_iou_distance)) { matchdet[cand[tid]] = cols[c]; dettrk[cols[c]] = cand[tid]; }
                    else c = -1;
                }
                if (prm.debug) dev.dbg_lists[(s * 4 + 3) * SS_MAXT + tid] = c;
            }
        }
    }
    __syncthreads();

    // ---------------- stage C: matched / missed tracks ---------------------------------------
    SS_TS_STEP(4);
    int alive = 0;
    if (tid < nT) {
        const size_t g = sb + myslot;
        const int d = matchdet[tid];
        if (d >= 0) {
            double mean[8], cov[64];
#pragma unroll
            for (int i = 0; i < 8; ++i) mean[i] = dev.mean[g * 8 + i];
#pragma unroll
            for (int i = 0; i < 64; ++i) cov[i] = dev.cov[g * 64 + i];
            const double* zz = dev.xyah + (db + d) * 4;
            double z[4] = { zz[0], zz[1], zz[2], zz[3] };
            const float* det = dev.dets + (db + d) * 6;
            ss_kf_update(mean, cov, z, (double)det[4], prm.wp);
#pragma unroll
            for (int i = 0; i < 8; ++i) dev.mean[g * 8 + i] = mean[i];
#pragma unroll
            for (int i = 0; i < 64; ++i) dev.cov[g * 64 + i] = cov[i];
            dev.conf[g] = det[4];
            dev.class_id[g] = (int)det[5];
            int h = dev.hits[g] + 1;
            dev.hits[g] = h;
            dev.tsu[g] = 0; mytsu = 0;
            dev.det_idx[g] = d;
            if (mystate == SS_TENTATIVE && h >= prm.n_init) mystate = SS_CONFIRMED;
        } else {
            if (mystate == SS_TENTATIVE || mytsu > prm.max_age) mystate = SS_DELETED;
        }
        dev.state[g] = mystate;
        alive = mystate != SS_DELETED;
        if (!alive) { dev.slot_used[g] = 0; dev.gal_count[g] = 0; dev.gal_head[g] = 0; }
    }
    // EMA of matched tracks: one wave per track
    for (int ti = wave; ti < nT; ti += 4) {
        const int d = matchdet[ti];
        if (d < 0) continue;
        float* sm = dev.smooth + (sb + dev.order[sb + ti]) * SS_F;
        ema_wave(sm, dev.feat_unit + (db + d) * SS_F, prm.ema_alpha, prm.ema_one_minus_alpha, sm);
    }
    __threadfence_block();
    // survivors keep their order
    int nSurv;
    block_scan256(alive, wtot, pos, nSurv);
    if (alive) neworder[pos] = myslot;
    // births: unmatched detections in ascending index
    int nNew, nFree;
    const int isNew = (tid < D) && (dettrk[tid] < 0);
    int rank;
    block_scan256(isNew, wtot, rank, nNew);
    const int isFree = !dev.slot_used[sb + tid];      // includes slots freed above (same thread wrote or fenced)
    __syncthreads();
    block_scan256(isFree, wtot, pos, nFree);
    if (isFree) freelist[pos] = tid;
    __syncthreads();
    if (nSurv + nNew > SS_MAXT || nNew > nFree) { if (tid == 0) dev.err[s] = SS_ERR_CAPACITY; nNew = min(nNew, min(nFree, SS_MAXT - nSurv)); }
    const int nid0 = dev.next_id[s];
    if (isNew && rank < nNew) {
        const int slot = freelist[rank];
        const size_t g = sb + slot;
        const double* zz = dev.xyah + (db + tid) * 4;
        double z[4] = { zz[0], zz[1], zz[2], zz[3] };
        double mean[8], cov[64];
        ss_kf_initiate(z, prm.wp, prm.wv, mean, cov);
#pragma unroll
        for (int i = 0; i < 8; ++i) dev.mean[g * 8 + i] = mean[i];
        for (int i = 0; i < 64; ++i) dev.cov[g * 64 + i] = cov[i];
        const float* det = dev.dets + (db + tid) * 6;
        dev.track_id[g] = nid0 + rank;
        dev.state[g] = SS_TENTATIVE; dev.hits[g] = 1; dev.age[g] = 1; dev.tsu[g] = 0;
        dev.class_id[g] = (int)det[5]; dev.conf[g] = det[4]; dev.det_idx[g] = tid;
        dev.gal_count[g] = 0; dev.gal_head[g] = 0; dev.slot_used[g] = 1;
        neworder[nSurv + rank] = slot;
        cols[rank] = tid;                         // detection of the rank-th birth (for the feature copy)
    }
    __syncthreads();
    const int nTot = nSurv + nNew;
    for (int k = wave; k < nNew; k += 4) {          // smooth feature of a new track = its unit feature
        const float* src = dev.feat_unit + (db + cols[k]) * SS_F;
        float* dst = dev.smooth + (sb + neworder[nSurv + k]) * SS_F;
        const int l = tid & 63;
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[l + 64 * j] = src[l + 64 * j];
    }
    if (tid < nTot) dev.order[sb + tid] = neworder[tid];
    if (tid == 0) { dev.n_tracks[s] = nTot; dev.next_id[s] = nid0 + nNew; dev.frame[s] += 1; }
    __threadfence_block();
    __syncthreads();

    // ---------------- stage D: gallery append (every confirmed track) + output rows -----------
    SS_TS_STEP(5);
    for (int k = wave; k < nSurv; k += 4) {
        const size_t g = sb + neworder[k];
        if (dev.state[g] != SS_CONFIRMED) continue;
        int head = dev.gal_head[g], cnt = dev.gal_count[g];
        gallery_append_wave(dev.gallery + g * SS_NRT * SS_TILE_FLOATS, head, dev.smooth + g * SS_F);
        if ((tid & 63) == 0) {
            dev.gal_head[g] = (head + 1 == prm.nn_budget) ? 0 : head + 1;
            dev.gal_count[g] = min(cnt + 1, prm.nn_budget);
        }
    }
    int emit = 0;
    size_t g = 0;
    if (tid < nTot) {
        g = sb + neworder[tid];
        emit = (dev.state[g] == SS_CONFIRMED) && (dev.tsu[g] <= 1);
    }
    int nOut;
    block_scan256(emit, wtot, pos, nOut);
    if (emit) {
        const double* m = dev.mean + g * 8;
        double w = m[2] * m[3];
        double x = m[0] - w / 2, y = m[1] - m[3] / 2;
        const int H = dev.img_hw[s * 2], W = dev.img_hw[s * 2 + 1];
        int x1 = max((int)x, 0), y1 = max((int)y, 0);
        int x2 = min((int)(x + w), W - 1), y2 = min((int)(y + m[3]), H - 1);
        float* o = dev.out_rows + (sb + pos) * 8;
        o[0] = (float)x1; o[1] = (float)y1; o[2] = (float)x2; o[3] = (float)y2;
        o[4] = (float)dev.track_id[g]; o[5] = (float)dev.class_id[g]; o[6] = dev.conf[g];
        o[7] = (float)dev.det_idx[g];
    }
    if (tid == 0) dev.n_out[s] = nOut;
    if (tid == 0 && s == 0) { dev.tile_count[0] = 0; dev.tile_count[1] = 0; }
    SS_TS_STEP(6);          // re-arm the association work list for the next frame
}

// =================================================================================================
// stage KAT kernels (thin wrappers over the same device functions)
// =================================================================================================
__global__ void k_kat_normalize(const float* raw, int n, float* unit)
{
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
    if (w >= n) return;
    float v[8], a = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = raw[(size_t)w * SS_F + l + 64 * j]; a = fmaf(v[j], v[j], a); }
    float nn = sqrtf(ss_wave_sumsq_reduce(a));
#pragma unroll
    for (int j = 0; j < 8; ++j) unit[(size_t)w * SS_F + l + 64 * j] = nn > 0.0f ? v[j] / nn : 0.0f;
}

__global__ void k_kat_ema(const float* smooth, const float* feat, int n, float a, float b, float* out)
{
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= n) return;
    ema_wave(smooth + (size_t)w * SS_F, feat + (size_t)w * SS_F, a, b, out + (size_t)w * SS_F);
}

__global__ void k_kat_kf(int op, double* mean, double* cov, const double* z, const double* conf, int n,
                         double wp, double wv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double m[8], c[64];
    if (op == 2) {
        double zz[4] = { z[i * 4], z[i * 4 + 1], z[i * 4 + 2], z[i * 4 + 3] };
        ss_kf_initiate(zz, wp, wv, m, c);
    } else {
        for (int k = 0; k < 8; ++k) m[k] = mean[(size_t)i * 8 + k];
        for (int k = 0; k < 64; ++k) c[k] = cov[(size_t)i * 64 + k];
        if (op == 0) ss_kf_predict(m, c, wp, wv);
        else {
            double zz[4] = { z[i * 4], z[i * 4 + 1], z[i * 4 + 2], z[i * 4 + 3] };
            ss_kf_update(m, c, zz, conf[i], wp);
        }
    }
    for (int k = 0; k < 8; ++k) mean[(size_t)i * 8 + k] = m[k];
    for (int k = 0; k < 64; ++k) cov[(size_t)i * 64 + k] = c[k];
}

__global__ void k_kat_pack(const float* nat, int T, int B, float* frag)
{
    // nat [T][B][512] -> frag [T][NRT][TILE_FLOATS]; one wave per gallery row
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= T * B) return;
    const int t = w / B, b = w % B;
    gallery_append_wave(frag + (size_t)t * SS_NRT * SS_TILE_FLOATS, b, nat + (size_t)w * SS_F);
}

__global__ void k_kat_featfrag(const float* unit, int D, float* frag)
{
    // unit [D][512] -> fragment-major column tiles (zero padded); one wave per detection
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int Dpad = (D + SS_TILE - 1) / SS_TILE * SS_TILE;
    if (w >= Dpad) return;
    frag_write_row(reinterpret_cast<float4*>(frag + (size_t)(w / SS_TILE) * SS_TILE_FLOATS), w % SS_TILE,
                   unit + (size_t)w * SS_F, w >= D);
}

__global__ void k_kat_gate(const float* part_min, const int* counts, int T, int D, const double* mean,
                           const double* cov, const double* xyah, SSParams prm, double* cost, float* cosd,
                           double* maha, uint8_t* gated)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * D) return;
    const int r = idx / D, d = idx % D;
    const float* pm = part_min + ((size_t)r * SS_NRT) * SS_MAXD + d;
    float c = INFINITY;
    for (int rt = 0; rt * SS_TILE < counts[r]; ++rt) c = fminf(c, pm[(size_t)rt * SS_MAXD]);
    double m[8], cv[64], m4[4], S[16], L[16];
    for (int k = 0; k < 8; ++k) m[k] = mean[(size_t)r * 8 + k];
    for (int k = 0; k < 64; ++k) cv[k] = cov[(size_t)r * 64 + k];
    ss_kf_project(m, cv, 0.0, prm.wp, m4, S);
    ss_chol4(S, L);
    double z[4] = { xyah[d * 4], xyah[d * 4 + 1], xyah[d * 4 + 2], xyah[d * 4 + 3] };
    double mh = ss_maha(L, m4, z);
    int g;
    double v = ss_blend(c, mh, prm, &g);
    cost[idx] = v; cosd[idx] = c; maha[idx] = mh; gated[idx] = (uint8_t)g;
}

__global__ void k_kat_iou(const double* ttlwh, int T, const double* dtlwh, int D, double max_dist, double* cost)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * D) return;
    const int r = idx / D, d = idx % D;
    double t[4] = { ttlwh[r * 4], ttlwh[r * 4 + 1], ttlwh[r * 4 + 2], ttlwh[r * 4 + 3] };
    double c[4] = { dtlwh[d * 4], dtlwh[d * 4 + 1], dtlwh[d * 4 + 2], dtlwh[d * 4 + 3] };
    cost[idx] = ss_iou_cost(t, c, max_dist);
}

// ---- launch helpers used by ss_api.hip -------------------------------------------------------------
size_t ss_step_lds_bytes()
{
    return (size_t)SS_COST_CAP * 8 + (3 * 256 * 8 + 4 * 256 * 4 + 512) + 8 * 256 * 4 + 32;
}
size_t ss_lsap_lds_bytes() { return 3 * 256 * 8 + 4 * 256 * 4 + 512; }
size_t ss_cosine_lds_bytes() { return 2 * SS_TILE_FLOATS * 4 + 512 * 16; }

extern "C" void ss_step_kernel_attr()
{
    // a failure here surfaces as a launch error on first use (checked with hipGetLastError after every launch)
    (void)hipFuncSetAttribute((const void*)k_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ss_step_lds_bytes());
    (void)hipFuncSetAttribute((const void*)k_lsap_kat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ss_lsap_lds_bytes());
    (void)hipFuncSetAttribute((const void*)k_cosine_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ss_cosine_lds_bytes());
}

void ss_launch_frame(const SSDev& dev, const SSParams& prm, int grid_tracks, hipStream_t st,
                     hipEvent_t ev0, hipEvent_t ev1)
{
    hipLaunchKernelGGL(k_pre, dim3(dev.S, 1 + SS_MAXD / 4), dim3(256), 0, st, dev, prm);
    // the association kernel that carries the gallery bytes of this launch gets the start/stop events
    if (dev.stream_mode) {
        if (ev0) hipExtLaunchKernelGGL(k_cosine_stream, dim3(dev.cos_grid), dim3(512), ss_cosine_lds_bytes(), st, ev0, ev1, 0, dev);
        else     hipLaunchKernelGGL(k_cosine_stream, dim3(dev.cos_grid), dim3(512), ss_cosine_lds_bytes(), st, dev);
        hipLaunchKernelGGL(k_cosine_wg, dim3(dev.cos_grid), dim3(512), 0, st, dev);          // D > 32 leftovers (usually empty)
    } else {
        if (ev0) hipExtLaunchKernelGGL(k_cosine_wg, dim3(dev.cos_grid), dim3(512), 0, st, ev0, ev1, 0, dev);
        else     hipLaunchKernelGGL(k_cosine_wg, dim3(dev.cos_grid), dim3(512), 0, st, dev);
    }
    hipLaunchKernelGGL(k_step, dim3(dev.S), dim3(256), ss_step_lds_bytes(), st, dev, prm);
}

void ss_launch_normalize(const float* raw, int n, float* unit, hipStream_t st)
{ if (n) hipLaunchKernelGGL(k_kat_normalize, dim3((n + 3) / 4), dim3(256), 0, st, raw, n, unit); }
void ss_launch_ema(const float* s, const float* f, int n, float a, float b, float* o, hipStream_t st)
{ if (n) hipLaunchKernelGGL(k_kat_ema, dim3((n + 3) / 4), dim3(256), 0, st, s, f, n, a, b, o); }
void ss_launch_kf(int op, double* mean, double* cov, const double* z, const double* conf, int n, double wp, double wv, hipStream_t st)
{ if (n) hipLaunchKernelGGL(k_kat_kf, dim3((n + 63) / 64), dim3(64), 0, st, op, mean, cov, z, conf, n, wp, wv); }
void ss_launch_pack(const float* nat, int T, int B, float* frag, hipStream_t st)
{ if (T * B) hipLaunchKernelGGL(k_kat_pack, dim3((T * B + 3) / 4), dim3(256), 0, st, nat, T, B, frag); }
void ss_launch_assoc(const float* gal_frag, const int* counts, int T, const float* feats, int D,
                     const double* mean, const double* cov, const double* xyah, const SSParams& prm,
                     float* feat_frag_scratch, float* part_min_scratch, double* cost, float* cosd,
                     double* maha, uint8_t* gated, hipStream_t st)
{
    if (T == 0 || D == 0) return;
    const int Dpad = (D + SS_TILE - 1) / SS_TILE * SS_TILE;
    hipLaunchKernelGGL(k_kat_featfrag, dim3((Dpad + 3) / 4), dim3(256), 0, st, feats, D, feat_frag_scratch);
    CosineArgs a{ gal_frag, counts, feat_frag_scratch, part_min_scratch, T, D };
    hipLaunchKernelGGL(k_cosine_kat, dim3(SS_NRT, T), dim3(512), 0, st, a);
    hipLaunchKernelGGL(k_kat_gate, dim3((T * D + 255) / 256), dim3(256), 0, st, part_min_scratch, counts, T, D,
                       mean, cov, xyah, prm, cost, cosd, maha, gated);
}
void ss_launch_iou(const double* t, int T, const double* d, int D, double md, double* cost, hipStream_t st)
{ if (T * D) hipLaunchKernelGGL(k_kat_iou, dim3((T * D + 255) / 256), dim3(256), 0, st, t, T, d, D, md, cost); }
void ss_launch_lsap(const double* cost, int nr, int nc, int* r2c, double* scratch, int* err, hipStream_t st)
{ hipLaunchKernelGGL(k_lsap_kat, dim3(1), dim3(64), ss_lsap_lds_bytes(), st, cost, nr, nc, r2c, scratch, err); }
