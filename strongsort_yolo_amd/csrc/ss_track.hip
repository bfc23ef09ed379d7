// ss_track.hip — device-resident StrongSORT tracker update for S independent streams, a GROUP of F frames per call.
//
// Rows a6..a10 of SURVEY.md §8(a).  Per group (F consecutive frames of every stream; F = 1 is the frame-at-a-time case):
//   k_group_prep  detection embeddings of all F frames L2-normalised and written row-major and fragment-major; the
//                 association work lists (gallery tiles of the confirmed tracks, column-tile pairs, work items).
//   k_assoc       THE association kernel: every gallery tile against the detections of all F frames — coalesced
//                 16-B gallery loads straight into v_mfma_f32_16x16x4_f32 operands, min over the rows that are
//                 still in the ring at each frame — so the gallery is read once per group instead of once per frame.
// and then per frame, in order (the tracker recurrence):
//   k_frame       one workgroup per stream: Kalman predict, gate/blend cost matrix in LDS, LSAP (single wave, SciPy's
//                 scan order), IoU stage + LSAP, track state machine, births, deletions, ring positions.
//   k_post        one wave per track: NSA Kalman update (64 lanes), EMA feature, gallery append, output row.
//   k_newrow      distances of the rows just appended to the detections of the group's later frames (MFMA, k-split).
// No host round trip anywhere in the group.
#include <hip/hip_ext.h>
#include "ss_common.h"

// Memory operations without a return value as inline assembly: the compiler's wait-count pass does not see them, so they do
// not mix a "store" event into the vector-memory counter — which would force s_waitcnt vmcnt(0) (i.e. the completion of the
// store / atomic itself, and the end of every prefetch in flight) before each later use of a loaded value.
// INVARIANT this relies on (gfx9 / CDNA wait counters): loads return in order and these hidden operations are counted in the
// same vmcnt, so an operation the compiler does not know about can only make its `s_waitcnt vmcnt(N)` wait for MORE than it
// asked (any N completions among the outstanding operations include every load older than the youngest N) — never less.  The
// LDS-DMA loads of k_assoc (ss_glds16) are awaited with explicit counts written against the same rule.  A target with separate
// load / store counters or out-of-order load return would need these helpers rewritten: hence the guard below, and the
// bit-exact k_assoc tests (tests/test_gpu_sequence.py) as the gate for toolchain upgrades.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "ss_track.hip: the hidden-vmcnt helpers and LDS-DMA waits are written for gfx950 (CDNA4) wait-counter semantics"
#endif
__device__ __forceinline__ void ss_atomic_min_nr(int* p, int v) { asm volatile("global_atomic_smin %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void ss_atomic_umin64_nr(unsigned long long* p, unsigned long long v) { asm volatile("global_atomic_umin_x2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void ss_atomic_umax64_nr(unsigned long long* p, unsigned long long v) { asm volatile("global_atomic_umax_x2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void ss_store_nr(long long* p, long long v) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }

// =================================================================================================
// block / wave helpers
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

// three independent flag scans for the price of one (k_frame ran ten scans of two barriers each: 20 of its ~30 barriers)
__device__ inline void block_scan256_3(int f0, int f1, int f2, int* wtot /*LDS[12]*/, int& p0, int& p1, int& p2, int& t0, int& t1, int& t2)
{
    const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1), m2 = __ballot(f2);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int i0 = __popcll(m0 & below), i1 = __popcll(m1 & below), i2 = __popcll(m2 & below);
    __syncthreads();                       // protect wtot from the previous scan's readers
    if (lane == 0) { wtot[w] = __popcll(m0); wtot[4 + w] = __popcll(m1); wtot[8 + w] = __popcll(m2); }
    __syncthreads();
    int o0 = 0, o1 = 0, o2 = 0, s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c0 = wtot[i], c1 = wtot[4 + i], c2 = wtot[8 + i];
        if (i < w) { o0 += c0; o1 += c1; o2 += c2; }
        s0 += c0; s1 += c1; s2 += c2;
    }
    p0 = o0 + i0; p1 = o1 + i1; p2 = o2 + i2;
    t0 = s0; t1 = s1; t2 = s2;
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

// =================================================================================================
// f32-MFMA dot products in oracle order, k-split form: one 8-wave workgroup per (16-row tile, pair of 16-column tiles),
// one 64-long k-segment per wave.  Used by the stage KAT (ss_assoc_cost) and by k_newrow.
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

// One pair of column tiles: 16 + 16 MFMAs per wave (two independent accumulation chains over the wave's k-segment),
// per-segment partial tiles to LDS, then thread (ctl, reg, lane) sums the 8 segment partials of accumulator element
// (row 4*(lane/16)+reg, column lane%16) of column tile ctl left to right (oracle order) and returns that dot product.
__device__ __forceinline__ float cosine_dots(const float4 a[4], const float4 b0[4], const float4 b1[4], float* lds_part)
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
    const int ctl = threadIdx.x >> 8, reg = (threadIdx.x >> 6) & 3;
    float tot = lds_part[((ctl * 8 + 0) * 4 + reg) * 64 + l];
#pragma unroll
    for (int sg = 1; sg < SS_NSEG; ++sg) tot = tot + lds_part[((ctl * 8 + sg) * 4 + reg) * 64 + l];
    return tot;
}

// ... -> 1 - dot -> min over the tile's valid rows (registers -> wave shuffle -> LDS).
__device__ __forceinline__ void cosine_pair(const float4 a[4], const float4 b0[4], const float4 b1[4], int count, int rt,
                                            int ct, bool two, int D, float* __restrict__ out, float* lds_part,
                                            float* lds_red)
{
    const int l = threadIdx.x & 63;
    const int ctl = threadIdx.x >> 8, reg = (threadIdx.x >> 6) & 3;
    const float tot = cosine_dots(a, b0, b1, lds_part);
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
    int* col4row;                                // [256] result: column of every row (LDS)
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
            const double ui = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(u) & 0xffffffff), i) & 0xffffffffll) |
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
    L.col4row = (int*)p; p += 256 * 4;
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
// per-track wave helpers
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

// the same from registers (lane l holds elements l + 64 j)
__device__ inline void ema_regs(const float sv[8], const float fv[8], float a, float b, float* out)
{
    const int l = threadIdx.x & 63;
    float v[8], acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t1 = a * sv[j];
        float t2 = b * fv[j];
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


// =================================================================================================
// k_group_prep — once per group: detection features of all F frames, association work lists, M rows
// =================================================================================================
// blockIdx.x = stream.  blockIdx.y = 0: work lists of the stream; 1..MAXT: M row of slot y-1; then 32 blocks per frame
// (one wave per detection): L2-normalise the raw embedding, write it row-major and fragment-major.
#define SS_PREP_FBLK (SS_MAXD / 4)
__global__ __launch_bounds__(256) void k_group_prep(SSDev dev)
{
    __shared__ int wtot[4];
    __shared__ int lcnt[8], lbase[8], npf[SS_FMAX], roff[SS_PLMAX + 1];
    __shared__ int tlw[SS_TLMAX];             // packed gallery tiles of the stream (work-list block only)
    __shared__ int cgw[SS_MAXT * 3 + 32];     // packed 4-row groups of the ragged last tiles
    __shared__ int ngs[SS_PLMAX];             // ordinary records of every pair
    __shared__ int2 plw[SS_PLMAX];
    __shared__ __attribute__((aligned(16))) float rowbuf[4][SS_F];
    const int s = blockIdx.x, tid = threadIdx.x, F = dev.F, S = dev.S;
    const size_t sb = (size_t)s * SS_MAXT;
    if (blockIdx.y == 0) {
        // gallery tiles of the tracks that are confirmed when the group starts (snapshot of count / ring head)
        const int nT = dev.n_tracks[s];
        // A track's last tile usually holds only a few rows (nn_budget 100 = 6 tiles + 4 rows): such rows travel as GROUPS of 4
        // (rows 4*r4 .. 4*r4+3 of the tile), four groups of any tracks make one composite 16-row tile for k_assoc — 188 instead
        // of 210 tiles per frame at 30 tracks x 100 rows.  Tiles with 13..15 rows stay ordinary tiles.
        int ntile = 0, ngrp = 0, slot = 0, count = 0, head = 0;
        if (tid < nT) {
            slot = dev.order[sb + tid];
            if (dev.state[sb + slot] == SS_CONFIRMED) {
                count = dev.gal_count[sb + slot]; head = dev.gal_head[sb + slot];
                const int rem = count % SS_TILE;
                ntile = count / SS_TILE;
                if (rem > dev.comp_rows) ++ntile; else ngrp = (rem + 3) / 4;
            }
        }
        int toff, ttot, goff, gtot;
        block_scan_sum256(ntile, wtot, toff, ttot);
        for (int rt = 0; rt < ntile; ++rt) tlw[toff + rt] = slot | (rt << 8) | (count << 12) | (head << 20);   // 8+3+8+7 bits
        block_scan_sum256(ngrp, wtot, goff, gtot);
        for (int g = 0; g < ngrp; ++g) cgw[goff + g] = slot | ((count / SS_TILE) << 8) | (g << 11) | (count << 13) | (head << 21);   // 8+3+2+8+7 bits
        __syncthreads();
        for (int i = gtot + tid; i < (gtot + 31) / 32 * 32; i += 256) cgw[i] = 0;      // empty groups: count 0, every row masked
        const int ncomp = (gtot + 3) / 4, ncrec = (ncomp + 7) / 8;                    // composite tiles, records of <= 8 of them
        // column-tile pairs of the group, frame by frame (a pair never spans two frames)
        int np = 0, D = 0;
        if (tid < F) { D = min(dev.n_dets[tid * S + s], SS_MAXD); np = ((D + SS_TILE - 1) / SS_TILE + 1) / 2; }
        int poff, ptot;
        block_scan_sum256(np, wtot, poff, ptot);
        if (tid < F) {
            dev.pf[s * (SS_FMAX + 1) + tid] = poff;
            const int nct = (D + SS_TILE - 1) / SS_TILE;
            for (int q = 0; q < np; ++q) {
                const int2 pr = make_int2(tid, (2 * q) | ((2 * q + 1 < nct) ? 256 : 0) | (D << 16));
                dev.pl[(size_t)s * SS_PLMAX + poff + q] = pr;
                plw[poff + q] = pr;
            }
        }
        if (tid == 0) { dev.pf[s * (SS_FMAX + 1) + F] = ptot; dev.n_pl[s] = ptot; }
        // assoc_pack: k_assoc's pairs run over the group's detections PACKED across frames (column g = detections of the frames
        // before + d, written by the detection-prep waves below): ceil(ceil(sum D / 16) / 2) pairs instead of sum ceil(ceil(D / 16) / 2)
        // — 28 instead of 32 at ~28 detections per frame, one step less for every wave of the launch.  k_newrow keeps the per-frame pairs.
        int coff, Dtot;
        block_scan_sum256(tid < F ? D : 0, wtot, coff, Dtot);
        const bool pack = dev.assoc_pack != 0;
        const int nctp = (Dtot + SS_TILE - 1) / SS_TILE, ppk = (nctp + 1) / 2;
        if (pack) {
            __syncthreads();                                             // plw: the per-frame pairs are in global memory now
            if (tid < ppk) plw[tid] = make_int2(0, (2 * tid) | ((2 * tid + 1 < nctp) ? 256 : 0) | (Dtot << 16));
            for (int g = Dtot + tid; g < 32 * ppk + 64; g += 256) dev.colmap[(size_t)s * (SS_FMAX * SS_MAXD + 64) + g] = -1;
            for (int g = Dtot + (tid >> 6); g < nctp * SS_TILE; g += 4)   // zero rows behind the last packed detection
                frag_write_row(reinterpret_cast<float4*>(dev.feat_pack + ((size_t)s * SS_NCTP + g / SS_TILE) * SS_TILE_FLOATS), g % SS_TILE, nullptr, true);
        }
        const int pta = pack ? ppk : ptot;                               // pairs the association records are cut from
        // Work records (one workgroup of k_assoc each) = (column-tile pair of one frame, range of <= SS_RECT gallery tiles).
        // A pair is cut into n_g ranges of (almost) equal length: n_g = the workgroups a pair can have when the launch's
        // cos_grid workgroups are shared evenly by the S*F frames and the frame's pairs, but ranges of at least 8 tiles (one
        // whole tile per wave) and at most SS_RECT (the record carries its tile words).  Inside k_assoc the 8 waves of the
        // workgroup split the range's 8*nt k-segments into 8 equal contiguous runs, so every wave of every workgroup of the
        // launch has the same amount of matrix work (+-1 segment).  Range j of every frame of stream s goes to list
        // (j + s) % 8 (n_g >= 8) and k_assoc's workgroup b serves list b % 8 (the dispatcher places block b on XCD b % 8):
        // the workgroups that need the same part of the gallery run on the same XCD — one pulls it from HBM, the others hit
        // that XCD's L2.
        if (tid < F) npf[tid] = np;
        if (tid < 8) lcnt[tid] = 0;
        __syncthreads();
        const int n_sf = max(1, dev.cos_grid / (S * F)), n_pp = max(1, dev.cos_grid / (S * max(ppk, 1)));
        int ng = 0, nr = 0;
        if (tid < pta) {
            if (ttot > 0) {
                ng = min(max((pack ? n_pp : n_sf / npf[plw[tid].x]) - ncrec, 1), max(1, ttot / 8));
                ng = max(ng, (ttot + SS_RECT - 1) / SS_RECT);
            }
            ngs[tid] = ng;
            nr = ng + ncrec;                                           // + the records of composite tiles (one tile per wave)
        }
        int ro, NR;
        block_scan_sum256(nr, wtot, ro, NR);
        if (tid < pta) roff[tid] = ro;
        if (tid == 0) roff[pta] = NR;
        __syncthreads();
        for (int pass = 0; pass < 2; ++pass) {
            for (int r = tid; r < NR; r += 256) {
                int lo = 0, hi = pta;                                  // pair of record r: roff[lo] <= r < roff[lo + 1]
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (roff[mid] <= r) lo = mid; else hi = mid; }
                const int pp = lo, j = r - roff[pp], nrp = roff[pp + 1] - roff[pp], ngp = ngs[pp];
                // list = XCD: range j of stream s always on the same XCD (its part of the gallery stays in that XCD's L2 for every
                // frame); with fewer than 8 ranges per pair, ceil(8 / ranges) consecutive pairs share the 8 XCDs between them, so a
                // range lives on that many XCDs (not on all 8: at 32 streams x 7 ranges that was 4.1 GB of HBM traffic per launch
                // instead of ~1 GB, L2 hit rate 0.33)
                // (assoc_xcd_map 1: all ranges of a pair on ONE XCD instead — each XCD then stages 1/8 of the detection operands and
                // streams the whole gallery; for launches of few pairs, where the operands' fabric traffic is the start-up burst)
                // (records per pair not a multiple of 8: the ranges past the last full eight rotate over the XCDs from pair to pair — with
                //  17 per pair, j = 16 of EVERY pair landed on list 0: 87 records for 64 workgroups there, 58 on the other lists)
                const int jx = (nrp >= 8 && j >= (nrp & ~7)) ? j + pp * (nrp & 7) : j;
                const int x = dev.xcd_map ? ((pp + s) & 7) : ((jx + s * nrp + (nrp < 8 ? (pp % ((8 + nrp - 1) / nrp)) * nrp : 0)) & 7);
                const int pos = atomicAdd(&lcnt[x], 1);
                if (pass == 0) continue;
                if (lbase[x] + pos >= dev.items_cap) { dev.err[s] = SS_ERR_CAPACITY; continue; }
                int4* rec = dev.items + ((size_t)x * dev.items_cap + lbase[x] + pos) * SS_RECI4;
                int tw[SS_RECT];
                if (j < ngp) {
                    // the cut points carry a per-pair offset: WHICH ranges get the extra tile of a division with remainder rotates from pair
                    // to pair (the same ranges of every pair did, and range j always lives on list j & 7: two lists carried 752 and 744 tiles
                    // against 692 on the others, their workgroups one step longer than the rest of the launch); a range still starts within
                    // one tile of the same place for every pair, so its part of the gallery stays in its XCD's L2
                    const int off = (pp * 5) % ngp;
                    const int t0 = (int)(((long long)j * ttot + off) / ngp), t1 = (int)(((long long)(j + 1) * ttot + off) / ngp), nt = t1 - t0;
#pragma unroll
                    for (int u = 0; u < SS_RECT; ++u) tw[u] = u < nt ? tlw[t0 + u] : 0;
                    rec[0] = make_int4(s, plw[pp].x, plw[pp].y, nt << 16);
                } else {
                    const int c0 = (j - ngp) * 8, nc = min(8, ncomp - c0);             // composite tiles c0 .. c0+nc-1, 4 group words each
#pragma unroll
                    for (int u = 0; u < SS_RECT; ++u) tw[u] = cgw[c0 * 4 + u];
                    rec[0] = make_int4(s, plw[pp].x, plw[pp].y, (nc << 16) | (int)0x80000000);
                }
#pragma unroll
                for (int u = 0; u < SS_RECT / 4; ++u) rec[1 + u] = make_int4(tw[4 * u], tw[4 * u + 1], tw[4 * u + 2], tw[4 * u + 3]);
            }
            __syncthreads();
            if (pass == 0) {
                if (tid < 8) { lbase[tid] = lcnt[tid] ? atomicAdd(dev.n_items + tid, lcnt[tid]) : 0; lcnt[tid] = 0; }
                __syncthreads();
            }
        }
        return;
    }
    if (blockIdx.y <= SS_MAXT) {
        // M[slot][f][d] = +inf for every confirmed track (the association kernel min-combines its tiles into it)
        const int slot = blockIdx.y - 1;
        if (!dev.slot_used[sb + slot] || dev.state[sb + slot] != SS_CONFIRMED) return;
        int4* m = reinterpret_cast<int4*>(dev.M + (sb + slot) * SS_FMAX * SS_MAXD);
        for (int i = tid; i < F * SS_MAXD / 4; i += 256) m[i] = make_int4(SS_KEY_INF, SS_KEY_INF, SS_KEY_INF, SS_KEY_INF);
        // ... and pull the track's gallery towards the chip: inside a frame pipeline the networks' kernels have pushed it out
        // of the memory-side cache since the last group, and every workgroup of k_assoc would start on HBM misses (its launch
        // was 3-7 us longer there than in a tracker-only loop).  One load per 128-byte line.
        if (dev.F > 1) {
            const int rows = dev.gal_count[sb + slot];
            const char* g = reinterpret_cast<const char*>(dev.gallery + (sb + slot) * SS_NRT * SS_TILE_FLOATS);
            const int lines = (rows + SS_TILE - 1) / SS_TILE * (SS_TILE_FLOATS * 4 / 128);
            int acc = 0;
#pragma unroll 8
            for (int i = tid; i < lines; i += 256) acc |= *reinterpret_cast<const int*>(g + (size_t)i * 128);
            asm volatile("" ::"v"(acc));                             // the loads stay, their values go nowhere
        }
        return;
    }
    // ---- detection prep: one wave per detection of frame f ----
    const int yy = blockIdx.y - 1 - SS_MAXT;
    const int f = yy / SS_PREP_FBLK;
    const int w = tid >> 6, l = tid & 63;
    const int d = (yy - f * SS_PREP_FBLK) * 4 + w;
    const size_t fs = (size_t)f * S + s;
    const int D = min(dev.n_dets[fs], SS_MAXD);
    const int Dpad = (D + SS_TILE - 1) / SS_TILE * SS_TILE;
    if (d >= Dpad) return;
    float4* frag = reinterpret_cast<float4*>(dev.feat_frag + (fs * SS_NCT + d / SS_TILE) * SS_TILE_FLOATS);
    const int jj = d % SS_TILE;
    if (d < D) {
        const float* raw = dev.feats_raw + (fs * SS_MAXD + d) * SS_F;
        float v[8], a = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = raw[l + 64 * j]; a = fmaf(v[j], v[j], a); }
        float n = sqrtf(ss_wave_sumsq_reduce(a));
        float* unit = dev.feat_unit + (fs * SS_MAXD + d) * SS_F;
#pragma unroll
        for (int j = 0; j < 8; ++j) { float u = n > 0.0f ? v[j] / n : 0.0f; unit[l + 64 * j] = u; rowbuf[w][l + 64 * j] = u; }   // all-zero row stays zero (D-17)
        SS_WAVE_SYNC();
        frag_write_row(frag, jj, rowbuf[w], false);
        if (dev.assoc_pack) {                                           // the same row as column g of the group's packed detections
            int cb = l < f ? min(dev.n_dets[(size_t)l * S + s], SS_MAXD) : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cb += __shfl_xor(cb, off);
            const int g = cb + d;
            frag_write_row(reinterpret_cast<float4*>(dev.feat_pack + ((size_t)s * SS_NCTP + g / SS_TILE) * SS_TILE_FLOATS), g % SS_TILE, rowbuf[w], false);
            if (l == 0) dev.colmap[(size_t)s * (SS_FMAX * SS_MAXD + 64) + g] = (f << 8) | d;
        }
        if (l == 0) {
            const float* b = dev.dets + (fs * SS_MAXD + d) * 6;
            double x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
            double bw = x2 - x1, bh = y2 - y1;
            double* t = dev.tlwh + (fs * SS_MAXD + d) * 4;
            double* z = dev.xyah + (fs * SS_MAXD + d) * 4;
            t[0] = x1; t[1] = y1; t[2] = bw; t[3] = bh;
            z[0] = x1 + bw / 2; z[1] = y1 + bh / 2; z[2] = bw / bh; z[3] = bh;
        }
    } else {
        frag_write_row(frag, jj, nullptr, true);
    }
}

// =================================================================================================
// k_assoc — the association kernel: gallery stream x detections of the whole group, f32 MFMA, row min
// =================================================================================================
// M[s][slot][f][d] = min over the gallery rows b of track `slot` that are still in its ring when frame f of the group
// is associated of (1 - g_b . f_d).  Row at ring position pos is valid for frame f iff pos < count and
// (pos - head) mod budget >= f: a confirmed track overwrites ring position head + j at the end of frame j of the
// group, so frame f must not see the rows at head .. head + f - 1 (their replacements are added by k_newrow).
// The gallery is therefore read ONCE per group, not once per frame.
//
// Work record = (stream, pair of 16-detection column tiles of one frame, range of nt <= SS_RECT gallery tiles), one
// workgroup of 8 waves per record (k_group_prep).  The pair's detection operand B (2 x 32 KiB fragment tiles) is staged once
// in LDS; the range's 8*nt k-segments (a tile = 8 segments of 64 k, 4 KiB of gallery per segment and wave) are cut into 8
// equal contiguous runs, one per wave, so all waves carry the same number of segments.  A segment = 8 x 16
// v_mfma_f32_16x16x4_f32 per column tile into a fresh accumulator (two independent chains); a tile's segment sums are added
// left to right (oracle order).  Where a run boundary cuts a tile, the wave that owns the tile's first segments computes them
// FIRST and hands its running sum to the next wave through LDS (one 2 KiB slot per wave + a flag); that wave processes the
// rest of the tile LAST, continues the sum in the same order and finishes the tile — bit-identical to one wave walking
// the whole tile.  Finishing = rows not in the ring at frame f masked, min over the tile's 16 rows with two shuffles,
// atomic min on an order-preserving key.  No barriers while a record is processed.  The gallery arrives as 4 KiB pieces
// per wave, prefetched 2 pieces ahead through a register ring (ordinary loads, so the counted vmcnt keeps 2 pieces in flight;
// 3 ahead made every CU pull 64 KB more through its vector L1 before the first MFMA and bought nothing at 4 waves per SIMD); B's global loads are issued together with the first gallery pieces.  nt <= 8: one whole tile per wave.
// Ragged last tile of a track: lanes of rows past the gallery count re-read row 0 (same cache lines, no extra HBM traffic).
// profiling aid (ss_assoc_timeline): wall-clock stamps (100 MHz) of wave 0 of every workgroup's first record
// (a separate instantiation: the stamps cost registers, the production kernel must keep 2 workgroups per CU)
#define SS_TL(i) do { if (TL && threadIdx.x == 0 && first_item) { ss_store_nr(dev.timeline + blockIdx.x * 16 + (i), wall_clock64()); \
                                                                   if (blockIdx.x < 2048) ss_store_nr(dev.timeline + (blockIdx.x + 2048) * 16 + (i), clock64()); } } while (0)

// LDS-DMA of one 1 KiB row of a fragment tile: lane l's 16 bytes land at lds_dst + 16 l (wave-uniform base in M0), no VGPRs hold
// the data.  Inline assembly: the compiler's wait-count pass does not see the load, so the gallery ring keeps its counted
// vmcnt; its completion is awaited with the explicit counts of the staging schedule below (loads return in order, so a
// wait that allows N younger operations guarantees everything issued before them).
__device__ __forceinline__ void ss_glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// ... and of 64 dwords (lane l's dword lands at lds_dst + 4 l): the record's column map
__device__ __forceinline__ void ss_glds4(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
#define SS_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// NP: how the record's detection operand B (2 x 32 KiB) reaches the LDS.  0: through registers, one barrier (round 3).
// 1 / 2 / 4: LDS-DMA in NP pieces of 8 / 4 / 2 k-segments (5: four pieces, only the first requested before the first barrier); a piece is awaited (counted vmcnt + barrier) right before the first
// step that can read it — every wave's run starts at k-segment 0 and advances at most one k-segment per step, and the
// first fragment of a step is fetched during the step before it.  The first MFMAs then wait for 16 + 32 KiB per workgroup
// instead of 64 + 64 KiB, and no staging registers / ds_write pass exist.
// The three leading scalars are what the first record's address needs: with -amdgpu-kernarg-preload-count they are in SGPRs
// when the wave starts (no dependent scalar load of the kernel-argument segment before the record fetch).
template <bool TL, int NP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_assoc(const int* items_all, const int* n_items_all, int items_cap_arg, SSDev dev)
{
    bool first_item = true;
    SS_TL(0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* bl = reinterpret_cast<float4*>(smem);                  // [2][32][64] float4 = 64 KiB: B of the record's pair
    float4* hand = bl + 4096;                                      // [7][2][64] float4 = 14 KiB: running sums wave w -> wave w+1
    int* hflag = reinterpret_cast<int*>(hand + 7 * 128);           // [8] record number whose sum is in the slot
    int* cmap = hflag + 16;                                        // [64] assoc_pack: frame<<8 | detection of the record's 32 packed columns (-1: none)
    const bool pack = dev.assoc_pack != 0;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    if (threadIdx.x < 8) hflag[threadIdx.x] = 0;                    // ordered before its first use by the staging barriers
    if (dev.ts_enable && threadIdx.x == 0) ss_atomic_umin64_nr(dev.tstamp, (unsigned long long)wall_clock64());
    const int xcd = blockIdx.x & 7;                                  // this workgroup's list (see k_group_prep)
    const int* items = items_all + (size_t)xcd * items_cap_arg * SS_RECW;
    int it = blockIdx.x >> 3;
    // The first record comes through the SCALAR cache: at the start of a launch the first vector load of a workgroup takes 0.7 to
    // 6.4 us (median 3.8, staggered by XCD) whatever it asks for, a scalar load 0.85 us (timeline, r03); the vector path's start-up
    // then overlaps the record fetch instead of following it.  Lane i of `rec` holds word i.  The following records are
    // prefetched by an ordinary vector load.
    int rec, n_items;
    {
        const int* rp = items + (size_t)it * SS_RECW;
        const int* np = n_items_all + xcd;
        i32x16 lo, hi;
        i32x4 top;
        asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx4 %2, %4, 0x80\n\ts_load_dword %3, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(lo), "=&s"(hi), "=&s"(top), "=&s"(n_items) : "s"(rp), "s"(np) : "memory");
        rec = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { rec = l == k ? lo[k] : rec; rec = l == 16 + k ? hi[k] : rec; }
#pragma unroll
        for (int k = 0; k < 4; ++k) rec = l == 32 + k ? top[k] : rec;
    }
    n_items = min(n_items, items_cap_arg);                             // k_group_prep drops records past the capacity (SS_ERR_CAPACITY)
    const int budget = dev.budget;
    int seq = 0;
    for (; it < n_items; it += gridDim.x >> 3) {
        ++seq;
        const int cur = rec;                                         // words 4.. = the range's packed tile words
        const int s = __builtin_amdgcn_readlane(cur, 0), f = __builtin_amdgcn_readlane(cur, 1);
        const int pw = __builtin_amdgcn_readlane(cur, 2), w3 = __builtin_amdgcn_readlane(cur, 3), nt = (w3 >> 16) & 0x7fff;
        const bool comp = w3 < 0;                                    // a record of composite tiles (nt <= 8: one per wave)
        const int ct0 = pw & 0xff, D = pw >> 16;
        const bool two = (pw >> 8) & 1;
        {   // next record (consumed at the end of the iteration; stale values past the end are never used)
            const int nx = it + (gridDim.x >> 3);
            if (nx < n_items) rec = items[(size_t)nx * SS_RECW + min(l, SS_RECW - 1)];
        }
        SS_TL(1);                                                    // record in registers
        // this wave's run of segments: [H] first nH segments of tile `last` (sum handed on), [M] nM whole tiles from
        // mfirst, [T] segments a0..7 of tile `first` (sum received)
        bool has = true;
        int nsteps = nt, nH = 0, nM = 1, mfirst = wu, first = wu, a0 = 0, last = wu;
        if (nt <= 8) { has = wu < nt; nsteps = 8; }
        else {
            const int g0 = wu * nt, g1 = g0 + nt;
            first = g0 >> 3; a0 = g0 & 7; last = g1 >> 3; nH = g1 & 7;
            mfirst = first + (a0 ? 1 : 0); nM = last - mfirst;
        }
        const int nHM = nH + 8 * nM;
        if (TL && threadIdx.x == 0 && first_item) { ss_store_nr(dev.timeline + blockIdx.x * 16 + 13, (long long)nt | ((long long)comp << 32)); ss_store_nr(dev.timeline + blockIdx.x * 16 + 14, (long long)n_items); }
        auto where = [&](int i, int& q, int& sg) __attribute__((always_inline)) {                  // segment i of the run -> (tile of the record, k-segment); selects, no branches
            const bool h = i < nH, m = i < nHM;
            q = h ? last : m ? mfirst + ((i - nH) >> 3) : first;
            sg = h ? i : m ? (i - nH) & 7 : a0 + (i - nHM);
        };
        const char* gbase = reinterpret_cast<const char*>(dev.gallery) + (size_t)s * SS_MAXT * SS_NRT * SS_TILE_FLOATS * 4;
        // lane (ks, i) = l reads float4 #(q*64 + l) of an ordinary tile; in a composite tile its row i belongs to group i/4 =
        // rows 4*r4 .. of tile (slot, rt) of that group's track
        auto group_word = [&](int q, int g) __attribute__((always_inline)) {                       // word g (per lane) of composite tile q
            const int b = 4 + 4 * q;
            const int w0 = __builtin_amdgcn_readlane(cur, b), w1 = __builtin_amdgcn_readlane(cur, b + 1);
            const int w2 = __builtin_amdgcn_readlane(cur, b + 2), w3g = __builtin_amdgcn_readlane(cur, b + 3);
            return g == 0 ? w0 : g == 1 ? w1 : g == 2 ? w2 : w3g;
        };
        auto ld = [&](int i, float4 a[4]) __attribute__((always_inline)) {
            int q, sg;
            where(i, q, sg);
            unsigned vo;
            if (comp) {
                const int gw = group_word(q, (l >> 2) & 3);
                vo = (unsigned)(((gw & 0xff) * SS_NRT + ((gw >> 8) & 7)) * (SS_TILE_FLOATS * 4) + ((l & 48) + 4 * ((gw >> 11) & 3) + (l & 3)) * 16);
            } else {
                const int tw = __builtin_amdgcn_readlane(cur, 4 + q);
                const int slot = tw & 0xff, rt = (tw >> 8) & 7, count = (tw >> 12) & 0xff;
                const bool ok = (l & 15) < count - rt * SS_TILE;
                vo = (unsigned)((slot * SS_NRT + rt) * (SS_TILE_FLOATS * 4) + (ok ? l : (l & ~15)) * 16);
            }
            const char* p = gbase + (size_t)sg * 4096;
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const float4*>(p + vo + j * 1024);
        };
        float4 ra[4][4];                                              // 4-deep ring of segment pieces
        // B of (frame f, stream s, column tiles ct0, ct0+1)
        const float4* ff = reinterpret_cast<const float4*>(pack ? dev.feat_pack + ((size_t)s * SS_NCTP + ct0) * SS_TILE_FLOATS
                                                                  : dev.feat_frag + (((size_t)f * dev.S + s) * SS_NCT + ct0) * SS_TILE_FLOATS);
        // the record's column map by LDS-DMA, issued by wave 0 before its first piece of B: loads return in order, so the first
        // awaited piece covers it, and the barrier after that piece publishes it (an uncounted older operation: no wait count changes)
        const int* cmsrc = dev.colmap + (size_t)s * (SS_FMAX * SS_MAXD + 64) + ct0 * SS_TILE + l;
        // (a lone column tile is staged twice: branch-free, and the second copy's results are never stored)
        const float4* ff1 = two ? ff + 4 * 512 : ff;
        const int tx = threadIdx.x;
        // piece m (k-segments 2m, 2m+1 of both column tiles): this wave's two 1 KiB rows of it, straight into bl
        auto dma = [&](int m) __attribute__((always_inline)) {
            ss_glds16(ff + tx + 512 * m, (unsigned)(m * 8192 + wu * 1024));
            ss_glds16(ff1 + tx + 512 * m, (unsigned)(32768 + m * 8192 + wu * 1024));
        };
        if constexpr (NP == 0) {
            if (has) { ld(0, ra[0]); ld(1, ra[1]); }                  // on the wire before the B staging (nsteps >= 8)
            const float4 t00 = ff[tx], t01 = ff[tx + 512], t02 = ff[tx + 1024], t03 = ff[tx + 1536];
            const float4 t10 = ff1[tx], t11 = ff1[tx + 512], t12 = ff1[tx + 1024], t13 = ff1[tx + 1536];
            const int cmv = (pack && tx < 64) ? *cmsrc : 0;
            __syncthreads();                                         // the previous record's readers are done with bl / hand
            if (pack && tx < 64) cmap[tx] = cmv;
            SS_TL(2);                                                // first gallery pieces + B landed
            bl[tx] = t00; bl[tx + 512] = t01; bl[tx + 1024] = t02; bl[tx + 1536] = t03;
            bl[tx + 2048] = t10; bl[tx + 2560] = t11; bl[tx + 3072] = t12; bl[tx + 3584] = t13;
            __syncthreads();
            SS_TL(3);                                                // B staged
        } else {
            if (seq > 1) SS_LDS_BARRIER();                           // the previous record's readers are done with bl / hand
            if (pack && wu == 0) ss_glds4(cmsrc, (unsigned)(reinterpret_cast<char*>(cmap) - smem));
            if (!has) {
                // a wave without a run (records of fewer than 8 tiles) only moves its rows of B and takes part in the barriers —
                // the same number of them as the waves with a run execute (the hardware barrier counts arrivals, not places)
                if constexpr (NP == 5) {
                    dma(0); SS_VMCNT(0); __builtin_amdgcn_s_barrier();
                    dma(1); dma(2); SS_VMCNT(2); __builtin_amdgcn_s_barrier();
                    dma(3); SS_VMCNT(2); __builtin_amdgcn_s_barrier();
                    SS_VMCNT(0); __builtin_amdgcn_s_barrier();
                } else if constexpr (NP == 4) {
                    dma(0); dma(1); SS_VMCNT(2); __builtin_amdgcn_s_barrier();
                    dma(2); SS_VMCNT(2); __builtin_amdgcn_s_barrier();
                    dma(3); SS_VMCNT(2); __builtin_amdgcn_s_barrier();
                    SS_VMCNT(0); __builtin_amdgcn_s_barrier();
                } else if constexpr (NP == 2) {
                    dma(0); dma(1); dma(2); dma(3); SS_VMCNT(4); __builtin_amdgcn_s_barrier();
                    SS_VMCNT(0); __builtin_amdgcn_s_barrier();
                } else {
                    dma(0); dma(1); dma(2); dma(3); SS_VMCNT(0); __builtin_amdgcn_s_barrier();
                }
            }
        }
        if (has) {
            // issue order with LDS-DMA staging: NP 4: m0, ld 0, m1, ld 1 | NP 2: m0 m1, ld 0, ld 1, m2 m3 | NP 1: m0..m3, ld 0, ld 1.
            // Every explicit wait admits exactly the operations issued after the awaited piece.
            if constexpr (NP == 4) { dma(0); ld(0, ra[0]); dma(1); ld(1, ra[1]); SS_VMCNT(10); }
            else if constexpr (NP == 5) { dma(0); ld(0, ra[0]); SS_VMCNT(4); }       // the start-up burst is 16 + 32 KiB per workgroup; the rest follows the barrier
            else if constexpr (NP == 2) { dma(0); dma(1); ld(0, ra[0]); ld(1, ra[1]); dma(2); dma(3); SS_VMCNT(12); }
            else if constexpr (NP == 1) { dma(0); dma(1); dma(2); dma(3); ld(0, ra[0]); ld(1, ra[1]); SS_VMCNT(8); }
            if constexpr (NP != 0) {
                SS_TL(2);                                            // first piece of B landed (this wave's rows)
                __builtin_amdgcn_s_barrier();
                SS_TL(3);                                            // first piece of B staged
                if constexpr (NP == 5) { dma(1); ld(1, ra[1]); }
            }
            const char* bls = reinterpret_cast<const char*>(bl) + l * 16;
            f32x4 tot0 = { 0.f, 0.f, 0.f, 0.f }, tot1 = { 0.f, 0.f, 0.f, 0.f };
            // B fragments are read from LDS one step (8 MFMAs) ahead of their use, so the LDS latency never shows
            int q0, sg0;
            where(0, q0, sg0);
            float4 bq0 = *reinterpret_cast<const float4*>(bls + sg0 * 4096), bq1 = *reinterpret_cast<const float4*>(bls + sg0 * 4096 + 32768);
            // one k-segment: prefetch (PF: piece i+3, clamped to the run's last piece so that the number of loads in flight
            // does not depend on the path — the compiler's wait counts stay exact), 32 MFMAs, running sum, tile end / hand-over
            auto step = [&](int i, float4* a, float4* anew, bool pf) __attribute__((always_inline)) {
                if (pf) ld(min(i + 2, nsteps - 1), anew);
                int q, sg, qn, sgn;
                where(i, q, sg);
                where(min(i + 1, nsteps - 1), qn, sgn);
                const char* bcur = bls + sg * 4096;
                const char* bnxt = bls + sgn * 4096;              // first fragments of the next segment (unused after the last)
                f32x4 acc0 = { 0.f, 0.f, 0.f, 0.f }, acc1 = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 b0 = bq0, b1 = bq1;
                    const char* bn = j < 3 ? bcur + (j + 1) * 1024 : bnxt;
                    bq0 = *reinterpret_cast<const float4*>(bn);
                    bq1 = *reinterpret_cast<const float4*>(bn + 32768);
                    acc0 = SS_MFMA16(a[j].x, b0.x, acc0); acc1 = SS_MFMA16(a[j].x, b1.x, acc1);
                    acc0 = SS_MFMA16(a[j].y, b0.y, acc0); acc1 = SS_MFMA16(a[j].y, b1.y, acc1);
                    acc0 = SS_MFMA16(a[j].z, b0.z, acc0); acc1 = SS_MFMA16(a[j].z, b1.z, acc1);
                    acc0 = SS_MFMA16(a[j].w, b0.w, acc0); acc1 = SS_MFMA16(a[j].w, b1.w, acc1);
                    __builtin_amdgcn_sched_barrier(0);      // keep this order: next fragments requested, then this step's MFMAs
                }
                if (sg == 0) { tot0 = acc0; tot1 = acc1; }
                else {
                    if (i == nHM && a0 != 0) {
                        // the tile's first a0 segment sums, added up in order by the previous wave
                        while (__hip_atomic_load(&hflag[wu - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) __builtin_amdgcn_s_sleep(1);
                        const float4 h0 = hand[(wu - 1) * 128 + l], h1 = hand[(wu - 1) * 128 + 64 + l];
                        tot0[0] = h0.x; tot0[1] = h0.y; tot0[2] = h0.z; tot0[3] = h0.w;
                        tot1[0] = h1.x; tot1[1] = h1.y; tot1[2] = h1.z; tot1[3] = h1.w;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { tot0[r] = tot0[r] + acc0[r]; tot1[r] = tot1[r] + acc1[r]; }
                }
                if (TL && !(i & 1) && i < 16) SS_TL(4 + (i >> 1));      // stamps after segments 0, 2, 4, ..., 14 of the run
                if (i == nH - 1) {
                    // hand the running sum of tile `last` to the wave that owns the rest of it
                    hand[wu * 128 + l] = make_float4(tot0[0], tot0[1], tot0[2], tot0[3]);
                    hand[wu * 128 + 64 + l] = make_float4(tot1[0], tot1[1], tot1[2], tot1[3]);
                    __hip_atomic_store(&hflag[wu], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (sg == 7 && !comp) {
                    // 1 - dot, rows not in the ring at frame f masked to +inf, min over the tile's 16 rows
                    const int tw = __builtin_amdgcn_readlane(cur, 4 + q);
                    const int slot = tw & 0xff, rt = (tw >> 8) & 7, count = (tw >> 12) & 0xff, head = (tw >> 20) & 0x7f;
                    // packed columns: the frame a row must still be in the ring for is the COLUMN's (column l & 15 of either tile)
                    int cm0 = 0, cm1 = 0, f0 = f, f1 = f;
                    if (pack) { cm0 = cmap[l & 15]; cm1 = cmap[16 + (l & 15)]; f0 = cm0 >> 8; f1 = cm1 >> 8; }
                    float m0 = INFINITY, m1 = INFINITY;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int pos = rt * SS_TILE + 4 * (l >> 4) + r;
                        int jrel = pos - head;
                        if (jrel < 0) jrel += budget;
                        const bool in_ring = pos < count;
                        m0 = fminf(m0, (in_ring && jrel >= f0) ? 1.0f - tot0[r] : INFINITY);
                        m1 = fminf(m1, (in_ring && jrel >= f1) ? 1.0f - tot1[r] : INFINITY);
                    }
                    m0 = fminf(m0, __shfl_xor(m0, 16)); m0 = fminf(m0, __shfl_xor(m0, 32));
                    m1 = fminf(m1, __shfl_xor(m1, 16)); m1 = fminf(m1, __shfl_xor(m1, 32));
                    if (!pack) {
                        int* out = dev.M + (((size_t)s * SS_MAXT + slot) * SS_FMAX + f) * SS_MAXD + ct0 * SS_TILE;
                        if (l < 16) { if (ct0 * SS_TILE + l < D) ss_atomic_min_nr(out + l, ss_fkey(m0)); }
                        else if (l < 32) { if (two && ct0 * SS_TILE + l < D) ss_atomic_min_nr(out + l, ss_fkey(m1)); }
                    } else if (l < 32) {                             // lane l < 16: column l of the first tile, else column l - 16 of the second
                        const int cm = l < 16 ? cm0 : cm1;
                        if (cm >= 0 && (l < 16 || two))
                            ss_atomic_min_nr(dev.M + (((size_t)s * SS_MAXT + slot) * SS_FMAX + (cm >> 8)) * SS_MAXD + (cm & 0xff), ss_fkey(l < 16 ? m0 : m1));
                    }
                }
                if (sg == 7 && comp) {
                    // composite tile: this lane's four accumulator rows are the four rows of group l/16 — one track, no shuffles
                    const int gw = group_word(q, l >> 4);
                    const int slot = gw & 0xff, rt = (gw >> 8) & 7, r4 = (gw >> 11) & 3, count = (gw >> 13) & 0xff, head = (gw >> 21) & 0x7f;
                    int cm0 = 0, cm1 = 0, f0 = f, f1 = f;
                    if (pack) { cm0 = cmap[l & 15]; cm1 = cmap[16 + (l & 15)]; f0 = cm0 >> 8; f1 = cm1 >> 8; }
                    float m0 = INFINITY, m1 = INFINITY;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int pos = rt * SS_TILE + 4 * r4 + r;
                        int jrel = pos - head;
                        if (jrel < 0) jrel += budget;
                        const bool in_ring = pos < count;
                        m0 = fminf(m0, (in_ring && jrel >= f0) ? 1.0f - tot0[r] : INFINITY);
                        m1 = fminf(m1, (in_ring && jrel >= f1) ? 1.0f - tot1[r] : INFINITY);
                    }
                    if (!pack) {
                        int* out = dev.M + (((size_t)s * SS_MAXT + slot) * SS_FMAX + f) * SS_MAXD + ct0 * SS_TILE + (l & 15);
                        if (count > 0 && ct0 * SS_TILE + (l & 15) < D) ss_atomic_min_nr(out, ss_fkey(m0));
                        if (count > 0 && two && ct0 * SS_TILE + SS_TILE + (l & 15) < D) ss_atomic_min_nr(out + SS_TILE, ss_fkey(m1));
                    } else {
                        int* mrow = dev.M + ((size_t)s * SS_MAXT + slot) * SS_FMAX * SS_MAXD;
                        if (count > 0 && cm0 >= 0) ss_atomic_min_nr(mrow + (cm0 >> 8) * SS_MAXD + (cm0 & 0xff), ss_fkey(m0));
                        if (count > 0 && two && cm1 >= 0) ss_atomic_min_nr(mrow + (cm1 >> 8) * SS_MAXD + (cm1 & 0xff), ss_fkey(m1));
                    }
                }
            };
            int i = 0;
            if constexpr (NP == 4 || NP == 5 || NP == 2) {
                // The first eight steps (nsteps >= 8 always) with the remaining pieces of B awaited on the way: step i reads
                // k-segments <= i and requests the first fragment of a k-segment <= i + 1.
#define SS_S1(k) step((k), ra[(k) & 3], ra[((k) + 2) & 3], true)
                if constexpr (NP == 4 || NP == 5) {
                    SS_S1(0);
                    dma(2);
                    SS_VMCNT(10); __builtin_amdgcn_s_barrier();     // piece 1 (before step 1); after it: ld 1, ld 2, m2
                    SS_S1(1); SS_S1(2);
                    dma(3);
                    SS_VMCNT(10); __builtin_amdgcn_s_barrier();     // piece 2 (before step 3); after it: ld 3, ld 4, m3
                    SS_S1(3); SS_S1(4);
                    SS_VMCNT(8); __builtin_amdgcn_s_barrier();      // piece 3 (before step 5); after it: ld 5, ld 6
                    SS_S1(5); SS_S1(6); SS_S1(7);
                } else {
                    SS_S1(0); SS_S1(1); SS_S1(2);
                    SS_VMCNT(12); __builtin_amdgcn_s_barrier();     // pieces 2, 3 (before step 3); after them: ld 2, ld 3, ld 4
                    SS_S1(3); SS_S1(4); SS_S1(5); SS_S1(6); SS_S1(7);
                }
#undef SS_S1
                i = 8;
            }
            for (; i + 4 <= nsteps; i += 4) {
                step(i, ra[0], ra[2], true); step(i + 1, ra[1], ra[3], true); step(i + 2, ra[2], ra[0], true); step(i + 3, ra[3], ra[1], true);
            }
            if (i < nsteps) step(i, ra[0], ra[2], true);               // the run's last 0..3 segments; the third one's piece is requested here
            if (i + 1 < nsteps) step(i + 1, ra[1], nullptr, false);
            if (i + 2 < nsteps) step(i + 2, ra[2], nullptr, false);
        }
        SS_TL(12);
        first_item = false;
    }
    if (dev.ts_enable) {
        __builtin_amdgcn_s_waitcnt(0);                               // this wave's memory operations have completed
        __syncthreads();
        if (threadIdx.x == 0) ss_atomic_umax64_nr(dev.tstamp + 1, (unsigned long long)wall_clock64());
    }
}

// =================================================================================================
// k_frame — per stream and frame: predict, cost matrix, LSAP, IoU stage, track bookkeeping (one workgroup)
// =================================================================================================
// Rows a6, a7, the gate/blend half of a8, a9 and the integer half of a10.  Everything that is per-track floating
// point work after the assignment (Kalman update, EMA, gallery append, output boxes) is handed to k_post through
// the `post` list; the appearance distances arrive ready-made in M (k_assoc + k_newrow).
struct FrameLds {
    double* cost;        // [SS_COST_CAP]
    double* chol;        // [MAXT][14] by track index: L (10 entries) + projected mean (4)
    double* ttl;         // [MAXT][4]  predicted tlwh by track index
    double* zs;          // [MAXD][4]  detection xyah
    double* dtl;         // [MAXD][4]  detection tlwh
    LsapLds L;
    int *matchdet, *dettrk, *asg, *cand, *cols, *neworder, *freelist, *conf_l, *slot_l, *tsu_l, *used, *rcnt, *ccnt, *rsel, *wtot;
};
// The f64 work areas are sized by the context (frame_caps: cost entries / tracks / detections kept in LDS); a frame with more
// tracks or detections than that takes the same arrays from the stream's global scratch (same arithmetic, L1 / L2 latency).
// Why it matters: with everything at its maximum the kernel asked for 156 KB — a whole CU's LDS — so its single workgroup could
// only start on a completely empty CU and, on the high-priority stream, held the other stream's workgroups back while it
// waited: the tracker chain ran almost serially with the networks' kernels instead of beside them (r04 trace: two kernels
// in flight 11 % of the time).
__device__ inline FrameLds carve_frame(char* p, const SSDev& dev, int s, int nT, int D)
{
    FrameLds m;
    m.cost = (double*)p; p += (size_t)dev.cap_cost * 8;
    const bool tl = nT <= dev.cap_t, dl = D <= dev.cap_d;
    double* g = dev.frame_scratch + (size_t)s * (SS_MAXT * 18 + SS_MAXD * 8);
    m.chol = tl ? (double*)p : g; p += (size_t)dev.cap_t * 14 * 8;
    m.ttl = tl ? (double*)p : g + SS_MAXT * 14; p += (size_t)dev.cap_t * 4 * 8;
    m.zs = dl ? (double*)p : g + SS_MAXT * 18; p += (size_t)dev.cap_d * 4 * 8;
    m.dtl = dl ? (double*)p : g + SS_MAXT * 18 + SS_MAXD * 4; p += (size_t)dev.cap_d * 4 * 8;
    m.L = carve_lsap(p);
    int** a[] = { &m.matchdet, &m.dettrk, &m.asg, &m.cand, &m.cols, &m.neworder, &m.freelist, &m.conf_l, &m.slot_l, &m.tsu_l, &m.used,
                  &m.rcnt, &m.ccnt, &m.rsel };
    for (auto q : a) { *q = (int*)p; p += 256 * 4; }
    m.wtot = (int*)p; p += 64;
    return m;
}
size_t ss_frame_lds_bytes(int cap_cost, int cap_t, int cap_d) { return (size_t)cap_cost * 8 + (size_t)cap_t * 18 * 8 + 2 * (size_t)cap_d * 4 * 8 + 15 * 256 * 4 + 64; }

// Assignment of one stage.  n_rows x n_cols is the matrix in its natural orientation (rows = tracks); cost is stored
// [nr][nc] after the transposition rule (rows = the smaller side).  Result: m.asg[row] = column or -1.
// Shortcut: rcnt / ccnt count the entries <= threshold per natural row / column (filled while the matrix is built,
// rsel[row] = such a column).  Every other entry equals the replacement value threshold + 1e-5, so if no row and no
// column holds more than one entry <= threshold, the optimal assignment is unique on those entries (swapping any of
// them for a replacement-valued entry costs strictly more) and everything else it contains is rejected by the
// threshold afterwards: the result of the LSAP is known without running it.  Otherwise: LSAP on one wave.
// Returns 1 (shortcut) or 2 (LSAP) for the debug record.
__device__ inline int frame_assign(int n_rows, int n_cols, bool big, const double* cost_lds, const double* cost_glb,
                                   const FrameLds& m, int* err)
{
    const int tid = threadIdx.x;
    const bool tr = n_cols < n_rows;
    const int nr = tr ? n_cols : n_rows, nc = tr ? n_rows : n_cols;
    const int multi = __syncthreads_or((tid < n_rows && m.rcnt[tid] > 1) || (tid < n_cols && m.ccnt[tid] > 1));
    if (!multi) {
        if (tid < n_rows) m.asg[tid] = m.rcnt[tid] == 1 ? m.rsel[tid] : -1;
        __syncthreads();
        return 1;
    }
    if ((tid >> 6) == 0) {
        const int rc = big ? lsap_wave(nr, nc, cost_glb, m.L) : lsap_wave(nr, nc, cost_lds, m.L);
        if (rc) { if (tid == 0) *err = SS_ERR_INFEASIBLE; }
        else for (int i = tid; i < nr; i += 64) { if (tr) m.asg[m.L.col4row[i]] = i; else m.asg[i] = m.L.col4row[i]; }
    }
    __syncthreads();
    return 2;
}

// -DSS_FRAME_STAMPS (a profiling build, tools/frame_phases.py): wall-clock time (100 MHz) of k_frame's phases and of k_postnew's two roles,
// summed per stream-0 launch into rows 3000-3002 of dev.timeline
#ifdef SS_FRAME_STAMPS
#define SS_FS(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const long long t_ = wall_clock64(); dev.timeline[3000 * 16 + (i)] += t_ - fs_t; fs_t = t_; } } while (0)
#define SS_FSV(i, val) do { asm volatile("" :: "v"(val)); if (threadIdx.x == 0 && blockIdx.x == 0) { const long long t_ = wall_clock64(); dev.timeline[3003 * 16 + (i)] += t_ - fs_t0; } } while (0)
#else
#define SS_FS(i) do { } while (0)
#define SS_FSV(i, val) do { } while (0)
#endif
__global__ __launch_bounds__(256) void k_frame(SSDev dev, SSParams prm, int f)
{
#ifdef SS_FRAME_STAMPS
    long long fs_t = wall_clock64();
    const long long fs_t0 = fs_t;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int s = blockIdx.x, tid = threadIdx.x, S = dev.S;
    const size_t sb = (size_t)s * SS_MAXT;
    const size_t fs = (size_t)f * S + s, fb = fs * SS_MAXD;        // (frame, stream) and its detection base
    // what does not depend on the counts is requested with them: every dependent load of a freshly launched workgroup is ~0.5-1 us (the
    // previous kernel's data comes from the memory side of the L2s), and the detection boxes used to be requested last, after the track block
    const int slot_raw = dev.order[sb + tid];
    double zraw[4], draw[4];
    {
        const size_t dz = (fb + (tid < SS_MAXD ? tid : SS_MAXD - 1)) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { zraw[i] = dev.xyah[dz + i]; draw[i] = dev.tlwh[dz + i]; }
    }
    const int nT = dev.n_tracks[s], D = min(dev.n_dets[fs], SS_MAXD);
    SS_FSV(0, nT + D);
    const FrameLds m = carve_frame(smem, dev, s, nT, D);
    if (s == 0 && tid == 0) {
#pragma unroll
        for (int x = 0; x < 8; ++x) dev.n_items[x] = 0;              // re-arm the work lists for the next group
        if (dev.ts_enable && f == 0) {                               // fold the association kernel's in-kernel duration
            const unsigned long long a = dev.tstamp[0], b = dev.tstamp[1];
            if (b > a) { dev.tstamp[2] += b - a; dev.tstamp[3] += 1; }
            dev.tstamp[0] = ~0ull; dev.tstamp[1] = 0;
        }
    }
    m.matchdet[tid] = -1; m.dettrk[tid] = -1; m.asg[tid] = -1; m.rcnt[tid] = 0; m.ccnt[tid] = 0;
    m.used[tid] = dev.slot_used[sb + tid];
    const size_t dbg = fs * SS_MAXT;                                 // debug base (rows of [F][S][MAXT]...)
    if (prm.debug) { dev.dbg_lists[(fs * 4 + 0) * SS_MAXT + tid] = -1; dev.dbg_lists[(fs * 4 + 3) * SS_MAXT + tid] = -1; }

    // ---------------- predict every live track (thread = position in the track list) -----------------------
    int myslot = -1, mystate = 0, mytsu = 0, confirmed = 0;
    int pre_hits = 0, pre_head = 0, pre_cnt = 0;                     // stage C's per-track counters, requested here with the rest of the state
    if (tid < nT) {
        myslot = slot_raw;
        SS_FSV(1, myslot);
        const size_t g = sb + myslot;
        double mean[8], cov[64];
        if (dev.pred_ahead && !dev.cmc) {
            // the prediction was made by post_track of the previous frame (one wave per track, beside the new-row units): only the
            // values the gate needs are read here (mean, the 4x4 block of the covariance), nothing is stored — post_track takes the
            // predicted state from the same arrays.  (This thread-per-track form loads, predicts and stores 72 doubles per track at a
            // 512-byte lane stride: 2-3 us of k_frame's 11.)
#pragma unroll
            for (int i = 0; i < 8; ++i) mean[i] = dev.mean_p[g * 8 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) cov[i * 8 + j] = dev.cov_p[g * 64 + i * 8 + j];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) mean[i] = dev.mean[g * 8 + i];
#pragma unroll
            for (int i = 0; i < 64; ++i) cov[i] = dev.cov[g * 64 + i];
            if (dev.cmc) {                                           // N4: camera motion between frame f-1 and f, before predicting
                const double* wm = dev.cmc + fs * 8;
                if (wm[6] >= 1.0) { const double m6[6] = { wm[0], wm[1], wm[2], wm[3], wm[4], wm[5] }; ss_camera_update(mean, m6); }
            }
            ss_kf_predict(mean, cov, prm.wp, prm.wv);
#pragma unroll
            for (int i = 0; i < 8; ++i) dev.mean[g * 8 + i] = mean[i];
#pragma unroll
            for (int i = 0; i < 64; ++i) dev.cov[g * 64 + i] = cov[i];
        }
        SS_FSV(2, (int)__double2loint(mean[0]) ^ (int)__double2loint(cov[27]));
        pre_hits = dev.hits[g]; pre_head = dev.gal_head[g]; pre_cnt = dev.gal_count[g];
        dev.age[g] += 1;
        mytsu = dev.tsu[g] + 1;
        SS_FSV(3, mytsu);
        dev.tsu[g] = mytsu;
        dev.det_idx[g] = -1;
        // gate factorisation (projection with conf = 0) and predicted box
        double m4[4], Sm[16], L[16];
        ss_kf_project(mean, cov, 0.0, prm.wp, m4, Sm);
        ss_chol4(Sm, L);
        double* ch = m.chol + tid * 14;
        ch[0] = L[0]; ch[1] = L[4]; ch[2] = L[5]; ch[3] = L[8]; ch[4] = L[9]; ch[5] = L[10];
        ch[6] = L[12]; ch[7] = L[13]; ch[8] = L[14]; ch[9] = L[15];
        ch[10] = m4[0]; ch[11] = m4[1]; ch[12] = m4[2]; ch[13] = m4[3];
        const double wd = mean[2] * mean[3];
        double* tb = m.ttl + tid * 4;
        tb[0] = mean[0] - wd / 2; tb[1] = mean[1] - mean[3] / 2; tb[2] = wd; tb[3] = mean[3];
        SS_FSV(4, (int)__double2loint(tb[0]));
        mystate = dev.state[g];
        confirmed = mystate == SS_CONFIRMED;
        SS_FSV(5, mystate);
    }
    m.slot_l[tid] = myslot;
    m.tsu_l[tid] = mytsu;
    if (tid < D) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { m.zs[tid * 4 + i] = zraw[i]; m.dtl[tid * 4 + i] = draw[i]; }
    }
    SS_FS(0);
    int pos, nC;
    block_scan256(confirmed, m.wtot, pos, nC);
    if (confirmed) m.conf_l[pos] = tid;
    __syncthreads();

    SS_FS(1);
    // ---------------- stage A: appearance + motion cost, LSAP --------------------------------
    double* spill = dev.cost_spill + (size_t)s * SS_MAXT * SS_MAXD;
    if (nC > 0 && D > 0) {
        const bool tr = D < nC;
        const int nr = tr ? D : nC, nc = tr ? nC : D;
        const bool big = nr * nc > dev.cap_cost;
        double* cost = big ? spill : m.cost;
        // the appearance distances of a thread's next four entries are requested together: one exposed round trip to the association
        // kernel's output per 1 024 entries instead of one per 256 (at 30 x 30: 4 -> 1; each was ~0.6 us of this ~2.7 us phase)
        const int nE = nC * D;
        for (int base = 0; base < nE; base += 1024) {
        int keys[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = min(base + tid + 256 * j, nE - 1);           // clamped: an unconditional load (a predicated one would wait alone)
            const int r = idx / D, d = idx - r * D;
            keys[j] = dev.M[((sb + m.slot_l[m.conf_l[r]]) * SS_FMAX + f) * SS_MAXD + d];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = base + tid + 256 * j;
            if (idx >= nE) break;
            const int r = idx / D, d = idx % D;
            const int ti = m.conf_l[r];
            const float c = ss_fkey_inv(keys[j]);
            const double* ch = m.chol + ti * 14;
            double Lm[16] = { ch[0], 0, 0, 0, ch[1], ch[2], 0, 0, ch[3], ch[4], ch[5], 0, ch[6], ch[7], ch[8], ch[9] };
            double m4[4] = { ch[10], ch[11], ch[12], ch[13] };
            double z[4] = { m.zs[d * 4], m.zs[d * 4 + 1], m.zs[d * 4 + 2], m.zs[d * 4 + 3] };
            const double maha = ss_maha(Lm, m4, z);
            int gt;
            const double v = ss_blend(c, maha, prm, &gt);
            cost[tr ? d * nc + r : r * nc + d] = v;
            if (!(v > prm.max_dist)) { atomicAdd(&m.rcnt[r], 1); atomicAdd(&m.ccnt[d], 1); m.rsel[r] = d; }
            if (prm.debug) {
                const size_t o = (dbg + r) * SS_MAXD + d;
                dev.dbg_cos[o] = c; dev.dbg_maha[o] = maha; dev.dbg_gated[o] = (uint8_t)gt; dev.dbg_cost_a[o] = v;
            }
        }
        }
        if (big) __threadfence();                        // the spilled matrix is read back by another wave
        __syncthreads();
        SS_FS(2);
        const int path = frame_assign(nC, D, big, m.cost, spill, m, dev.err + s);
        if (prm.debug && tid == 0) dev.dbg_counts[fs * 8 + 4] = path;
        if (tid < nC) {
            int d = m.asg[tid];
            if (d >= 0) {
                const double v = cost[tr ? d * nc + tid : tid * nc + d];
                if (!(v > prm.max_dist)) { m.matchdet[m.conf_l[tid]] = d; m.dettrk[d] = m.conf_l[tid]; }
                else d = -1;
            }
            if (prm.debug) dev.dbg_lists[(fs * 4 + 0) * SS_MAXT + tid] = d;
        }
    }
    __syncthreads();

    SS_FS(3);
    // ---------------- stage B: IoU association ----------------------------------------------
    int nU, nC1, nCols;
    const int isU = (tid < nT) && (mystate != SS_CONFIRMED);
    const int isC1 = (tid < nT) && (mystate == SS_CONFIRMED) && (m.matchdet[tid] < 0) && (mytsu == 1);
    const int isCol = (tid < D) && (m.dettrk[tid] < 0);
    int posC1, posCol;
    block_scan256_3(isU, isC1, isCol, m.wtot, pos, posC1, posCol, nU, nC1, nCols);
    if (isU) m.cand[pos] = tid;
    if (isC1) m.cand[nU + posC1] = tid;
    const int nCand = nU + nC1;
    if (isCol) m.cols[posCol] = tid;
    m.asg[tid] = -1; m.rcnt[tid] = 0; m.ccnt[tid] = 0;
    __syncthreads();
    if (prm.debug) {
        if (tid < nCand) dev.dbg_lists[(fs * 4 + 1) * SS_MAXT + tid] = m.cand[tid];
        if (tid < nCols) dev.dbg_lists[(fs * 4 + 2) * SS_MAXT + tid] = m.cols[tid];
        if (tid == 0) { int* c = dev.dbg_counts + fs * 8; c[0] = nC; c[1] = nCand; c[2] = nCols; c[3] = D; c[5] = 0; if (!(nC > 0 && D > 0)) c[4] = 0; }
    }
    SS_FS(4);
    if (nCand > 0 && nCols > 0) {
        const bool tr = nCols < nCand;
        const int nr = tr ? nCols : nCand, nc = tr ? nCand : nCols;
        const bool big = nr * nc > dev.cap_cost;
        double* cost = big ? spill : m.cost;
        for (int idx = tid; idx < nCand * nCols; idx += 256) {
            const int r = idx / nCols, c = idx % nCols;
            const int ti = m.cand[r];
            const double* tb = m.ttl + ti * 4;
            const double* cb = m.dtl + m.cols[c] * 4;
            double t[4] = { tb[0], tb[1], tb[2], tb[3] }, cc[4] = { cb[0], cb[1], cb[2], cb[3] };
            const double v = (m.tsu_l[ti] > 1) ? prm.max_iou_distance + 1e-5 : ss_iou_cost(t, cc, prm.max_iou_distance);
            cost[tr ? c * nc + r : r * nc + c] = v;
            if (!(v > prm.max_iou_distance)) { atomicAdd(&m.rcnt[r], 1); atomicAdd(&m.ccnt[c], 1); m.rsel[r] = c; }
            if (prm.debug) dev.dbg_cost_b[(dbg + r) * SS_MAXD + c] = v;
        }
        if (big) __threadfence();
        __syncthreads();
        const int path = frame_assign(nCand, nCols, big, m.cost, spill, m, dev.err + s);
        if (prm.debug && tid == 0) dev.dbg_counts[fs * 8 + 5] = path;
        if (tid < nCand) {
            int c = m.asg[tid];
            if (c >= 0) {
                const double v = cost[tr ? c * nc + tid : tid * nc + c];
                if (!(v > prm.max_iou_distance)) { m.matchdet[m.cand[tid]] = m.cols[c]; m.dettrk[m.cols[c]] = m.cand[tid]; }
                else c = -1;
            }
            if (prm.debug) dev.dbg_lists[(fs * 4 + 3) * SS_MAXT + tid] = c;
        }
    }
    __syncthreads();

    SS_FS(5);
    // ---------------- stage C: track states, survivors, gallery ring positions, output slots -----------------
    int alive = 0, md = -1, fl = 0, aux = 0, doapp = 0, emit = 0;
    if (tid < nT) {
        const size_t g = sb + myslot;
        md = m.matchdet[tid];
        if (md >= 0) {
            const float* det = dev.dets + (fb + md) * 6;
            dev.conf[g] = det[4];
            dev.class_id[g] = (int)det[5];
            const int h = pre_hits + 1;
            dev.hits[g] = h;
            dev.tsu[g] = 0; mytsu = 0;
            dev.det_idx[g] = md;
            if (mystate == SS_TENTATIVE && h >= prm.n_init) mystate = SS_CONFIRMED;
            fl = SS_P_MATCHED;
        } else if (mystate == SS_TENTATIVE || mytsu > prm.max_age) mystate = SS_DELETED;
        dev.state[g] = mystate;
        alive = mystate != SS_DELETED;
        if (!alive) { dev.slot_used[g] = 0; m.used[myslot] = 0; dev.gal_count[g] = 0; dev.gal_head[g] = 0; }
        else if (mystate == SS_CONFIRMED) {
            // every confirmed track appends its (possibly just updated) EMA feature to its gallery ring (D-05)
            const int head = pre_head, cnt = pre_cnt;
            dev.gal_head[g] = (head + 1 == prm.nn_budget) ? 0 : head + 1;
            dev.gal_count[g] = min(cnt + 1, prm.nn_budget);
            doapp = 1;
            fl |= SS_P_APPEND | (cnt == 0 ? SS_P_FIRSTROW : 0);
            aux = head;
            emit = mytsu <= 1;
        }
    }
    SS_FS(6);
    int nSurv, nApp, nOut, apos, epos;
    block_scan256_3(alive, doapp, emit, m.wtot, pos, apos, epos, nSurv, nApp, nOut);
    if (alive) {
        if (emit) { fl |= SS_P_EMIT; aux |= epos << 8; }
        m.neworder[pos] = myslot;
        dev.post[sb + pos] = make_int4(myslot, md, fl, aux);
        if (doapp) dev.rowlist[sb + apos] = myslot | ((fl & SS_P_FIRSTROW) ? 1 << 16 : 0) | ((dev.smooth_sel[sb + myslot] & 1) << 17) |
                                           ((fl & SS_P_MATCHED) ? (1 << 18) | (md << 19) : 0);
    }
    SS_FS(7);
    // births: unmatched detections in ascending index take the free slots in ascending order
    int nNew, nFree, rank;
    const int isNew = (tid < D) && (m.dettrk[tid] < 0);
    const int isFree = !m.used[tid];                     // includes the slots freed above (the scan before this one synchronised)
    int pz, nz;
    block_scan256_3(isNew, isFree, 0, m.wtot, rank, pos, pz, nNew, nFree, nz);
    if (isFree) m.freelist[pos] = tid;
    __syncthreads();
    if (nSurv + nNew > SS_MAXT || nNew > nFree) { if (tid == 0) dev.err[s] = SS_ERR_CAPACITY; nNew = min(nNew, min(nFree, SS_MAXT - nSurv)); }
    const int nid0 = dev.next_id[s];
    if (isNew && rank < nNew) {
        const int slot = m.freelist[rank];
        const size_t g = sb + slot;
        const float* det = dev.dets + (fb + tid) * 6;
        dev.track_id[g] = nid0 + rank;
        dev.state[g] = SS_TENTATIVE; dev.hits[g] = 1; dev.age[g] = 1; dev.tsu[g] = 0;
        dev.class_id[g] = (int)det[5]; dev.conf[g] = det[4]; dev.det_idx[g] = tid;
        dev.gal_count[g] = 0; dev.gal_head[g] = 0; dev.slot_used[g] = 1;
        m.neworder[nSurv + rank] = slot;
        dev.post[sb + nSurv + rank] = make_int4(slot, tid, SS_P_BIRTH, 0);
    }
    __syncthreads();
    SS_FS(8);
    const int nTot = nSurv + nNew;
    if (tid < nTot) dev.order[sb + tid] = m.neworder[tid];
    if (tid == 0) {
        dev.n_tracks[s] = nTot; dev.next_id[s] = nid0 + nNew; dev.frame[s] += 1;
        dev.n_post[s] = nTot; dev.n_rows[s] = nApp; dev.n_out[fs] = nOut;
    }
    SS_FS(9);
#ifdef SS_FRAME_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) dev.timeline[3000 * 16 + 15] += 1;
#endif
}

// =================================================================================================
// k_post — per-track floating-point work of the frame, one wave per surviving / new track
// =================================================================================================
// NSA Kalman update across the wave's 64 lanes (one covariance entry each), EMA feature, Kalman initiation of new
// tracks, gallery append (fragment-major) and the output row.  grid = (streams, SS_POST_BLOCKS).
#define SS_POST_BLOCKS 16
// one surviving / new track, by one wave (ws: 72 doubles, rowbuf: 512 floats of this wave)
__device__ __forceinline__ void post_track(const SSDev& dev, const SSParams& prm, int s, int f, int k, double* ws, float* rowbuf)
{
    const int l = threadIdx.x & 63, S = dev.S;
    const size_t sb = (size_t)s * SS_MAXT;
    const size_t fs = (size_t)f * S + s, fb = fs * SS_MAXD;
    const int4 e = dev.post[sb + k];
    const int slot = __builtin_amdgcn_readfirstlane(e.x), d = __builtin_amdgcn_readfirstlane(e.y);
    const int fl = __builtin_amdgcn_readfirstlane(e.z), aux = __builtin_amdgcn_readfirstlane(e.w);
    const size_t g = sb + slot;
    const int sel = __builtin_amdgcn_readfirstlane(dev.smooth_sel[g]) & 1;
    const float* sm = dev.smooth + (g * 2 + sel) * SS_F;          // the EMA row in use
    const float* src = sm;                                        // the row a confirmed track appends
    // the state this frame starts from: what k_frame predicted in place, or the prediction the previous frame's post_track left
    const bool use_pred = dev.pred_ahead && !dev.cmc;
    const double* imean = use_pred ? dev.mean_p + g * 8 : dev.mean + g * 8;
    const double* icov = use_pred ? dev.cov_p + g * 64 : dev.cov + g * 64;
    if (fl & SS_P_MATCHED) {
        const double* zz = dev.xyah + (fb + d) * 4;
        const double z[4] = { zz[0], zz[1], zz[2], zz[3] };
        const float* fu = dev.feat_unit + (fb + d) * SS_F;
        float sv[8], fv[8];                                       // EMA operands on the wire before the Kalman arithmetic
#pragma unroll
        for (int j = 0; j < 8; ++j) { sv[j] = sm[l + 64 * j]; fv[j] = fu[l + 64 * j]; }
        ss_kf_update_wave(imean, icov, dev.mean + g * 8, dev.cov + g * 64, z, (double)dev.dets[(fb + d) * 6 + 4], prm.wp, ws);
        ema_regs(sv, fv, prm.ema_alpha, prm.ema_one_minus_alpha, rowbuf);
        SS_WAVE_SYNC();
        // the new row goes to the OTHER half: new-row units of this launch (k_postnew) may still be reading the old one
        float* sn = dev.smooth + (g * 2 + (sel ^ 1)) * SS_F;
#pragma unroll
        for (int j = 0; j < 8; ++j) sn[l + 64 * j] = rowbuf[l + 64 * j];
        if (l == 0) dev.smooth_sel[g] = sel ^ 1;
        src = rowbuf;
    } else if (fl & SS_P_BIRTH) {
        const double* zz = dev.xyah + (fb + d) * 4;
        const double h = zz[3];
        const int r = l >> 3, c = l & 7;
        // ss_kf_initiate, one covariance entry per lane
        const double sd = (r == 2) ? 1e-2 : (r == 6) ? 1e-5 : (r < 4) ? 2.0 * prm.wp * h : 10.0 * prm.wv * h;
        const double c0 = (r == c) ? sd * sd : 0.0;
        dev.cov[g * 64 + l] = c0; ws[l] = c0;
        if (l < 8) { const double m0 = (l < 4) ? zz[l] : 0.0; dev.mean[g * 8 + l] = m0; ws[64 + l] = m0; }
        const float* fu = dev.feat_unit + (fb + d) * SS_F;
        float* sw = dev.smooth + (g * 2 + sel) * SS_F;
#pragma unroll
        for (int j = 0; j < 8; ++j) sw[l + 64 * j] = fu[l + 64 * j];
    }
    else {                                                        // a track without a detection keeps the predicted state
        const double cv = icov[l];
        ws[l] = cv;
        if (use_pred) dev.cov[g * 64 + l] = cv;
        if (l < 8) { const double mv = imean[l]; ws[64 + l] = mv; if (use_pred) dev.mean[g * 8 + l] = mv; }
    }
    SS_WAVE_SYNC();                                               // ws = the track's state after this frame
    ss_kf_predict_wave(ws, prm.wp, prm.wv, dev.mean_p + g * 8, dev.cov_p + g * 64);
    if (fl & SS_P_APPEND) gallery_append_wave(dev.gallery + g * SS_NRT * SS_TILE_FLOATS, aux & 0xff, src);
    if ((fl & SS_P_EMIT) && l == 0) {
        double mm[4];
        { mm[0] = ws[64]; mm[1] = ws[65]; mm[2] = ws[66]; mm[3] = ws[67]; }
        const double wd = mm[2] * mm[3];
        const double x = mm[0] - wd / 2, y = mm[1] - mm[3] / 2;
        const int H = dev.img_hw[s * 2], W = dev.img_hw[s * 2 + 1];
        const int x1 = max((int)x, 0), y1 = max((int)y, 0);
        const int x2 = min((int)(x + wd), W - 1), y2 = min((int)(y + mm[3]), H - 1);
        float* o = dev.out_rows + (fs * SS_MAXT + (aux >> 8)) * 8;
        o[0] = (float)x1; o[1] = (float)y1; o[2] = (float)x2; o[3] = (float)y2;
        o[4] = (float)dev.track_id[g]; o[5] = (float)dev.class_id[g]; o[6] = dev.conf[g];
        o[7] = (float)dev.det_idx[g];
    }
    SS_WAVE_SYNC();                                               // rowbuf / ws are reused by the next track
}

__global__ __launch_bounds__(256) void k_post(SSDev dev, SSParams prm, int f)
{
    __shared__ double ws[4][72];
    __shared__ __attribute__((aligned(16))) float rowbuf[4][SS_F];
    const int s = blockIdx.x, w = threadIdx.x >> 6;
    const int n = dev.n_post[s];
    for (int k = blockIdx.y * 4 + w; k < n; k += gridDim.y * 4) post_track(dev, prm, s, f, k, ws[w], rowbuf[w]);
}

// =================================================================================================
// k_newrow — distances of the gallery rows appended in frame f to the detections of the later frames of the group
// =================================================================================================
// Unit = (16 appended rows = 16 tracks, pair of column tiles of a frame f2 > f): the k-split MFMA form
// (cosine_dots), A gathered from the 16 rows' EMA features.  Every (track, detection) entry is kept:
// M[slot][f2][d] = min(M, 1 - dot), or just 1 - dot for a gallery's first row.
// OWN_ROWS (k_postnew: the update of the same frame runs beside this unit): the 16 rows are computed HERE, from the row list's
// (smooth half at frame start, matched detection) — the same ema_regs arithmetic on the same operands as post_track — into LDS;
// !OWN_ROWS (k_newrow after k_post): they are read from the half post_track wrote.
template <bool OWN_ROWS>
__device__ __forceinline__ void newrow_units(const SSDev& dev, const SSParams& prm, int s, int f, int u0, int ustep, float* lds_part /*[2*8*4*64]*/,
                                             int* s_slot /*[16]*/, float* rows /*[16][SS_F], OWN_ROWS*/)
{
    const int S = dev.S, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int nrt = (dev.n_rows[s] + 15) / 16;
    const int p0 = dev.pf[s * (SS_FMAX + 1) + f + 1], np = dev.n_pl[s] - p0;
    const size_t sb = (size_t)s * SS_MAXT, fb = ((size_t)f * S + s) * SS_MAXD;
    int have = -1;                                                       // row tile whose rows are in `rows`
    for (int u = u0; u < nrt * np; u += ustep) {
        const int rti = u / np, p = p0 + u % np;                         // row tile major, pair fastest
        const int2 pr = dev.pl[(size_t)s * SS_PLMAX + p];
        const int f2 = pr.x, ct0 = pr.y & 0xff, D2 = pr.y >> 16;
        const bool two = (pr.y >> 8) & 1;
        const int nrow = min(16, dev.n_rows[s] - rti * 16);
        __syncthreads();                                              // the previous unit is done with s_slot / lds_part / rows
        if (threadIdx.x < 16) s_slot[threadIdx.x] = (int)threadIdx.x < nrow ? dev.rowlist[sb + rti * 16 + threadIdx.x] : -1;
        __syncthreads();
        if (OWN_ROWS && have != rti) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {                              // wave w: rows 2w, 2w + 1
                const int i = 2 * w + q, e = s_slot[i];
                float* out = rows + i * SS_F;
                if (e >= 0) {
                    const float* sm = dev.smooth + ((sb + (e & 0xffff)) * 2 + ((e >> 17) & 1)) * SS_F;
                    float sv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) sv[j] = sm[l + 64 * j];
                    if ((e >> 18) & 1) {
                        const float* fu = dev.feat_unit + (fb + ((e >> 19) & 0x7f)) * SS_F;
                        float fv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) fv[j] = fu[l + 64 * j];
                        ema_regs(sv, fv, prm.ema_alpha, prm.ema_one_minus_alpha, out);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) out[l + 64 * j] = sv[j];
                    }
                }
            }
            have = rti;
            __syncthreads();
        }
        // A: element k = 16(4w+j) + 4c + ks of row i, lane = ks*16 + i (the fragment-major float4 #((4w+j)*64 + lane))
        const int i = l & 15, ks = l >> 4;
        const int si = s_slot[i];
        float4 a[4];
        if (si >= 0) {
            const float* row = OWN_ROWS ? rows + i * SS_F + ks
                                        : dev.smooth + ((sb + (si & 0xffff)) * 2 + (dev.smooth_sel[sb + (si & 0xffff)] & 1)) * SS_F + ks;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* q = row + 16 * (4 * w + j);
                a[j] = make_float4(q[0], q[4], q[8], q[12]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 b0[4], b1[4];
        load_b(dev.feat_frag + ((size_t)f2 * S + s) * SS_NCT * SS_TILE_FLOATS, ct0, two, b0, b1);
        const float tot = cosine_dots(a, b0, b1, lds_part);
        const int ctl = threadIdx.x >> 8, reg = (threadIdx.x >> 6) & 3;
        const int row = 4 * (l >> 4) + reg, d = (ct0 + ctl) * SS_TILE + (l & 15);
        if ((ctl == 0 || two) && row < nrow && d < D2) {
            const int sl = s_slot[row];
            int* mp = dev.M + ((sb + (sl & 0xffff)) * SS_FMAX + f2) * SS_MAXD + d;
            const int key = ss_fkey(1.0f - tot);
            *mp = ((sl >> 16) & 1) ? key : min(*mp, key);
        }
    }
}

__global__ __launch_bounds__(512) void k_newrow(SSDev dev, SSParams prm, int f)
{
    __shared__ float lds_part[2 * 8 * 4 * 64];
    __shared__ int s_slot[16];
    newrow_units<false>(dev, prm, blockIdx.y, f, blockIdx.x, gridDim.x, lds_part, s_slot, nullptr);
}

// k_post and k_newrow of a frame in ONE launch (the chain of a group is 2 F dependent launches instead of 3 F - 1): workgroups
// 0 .. SS_PN_POST - 1 of a stream run post_track (8 waves each), the others the new-row units, which do not wait for them:
// they compute the rows they need themselves from the half of `smooth` that this launch does not write.
#define SS_PN_POST 8
#define SS_PN_NEW 64
__global__ __launch_bounds__(512) void k_postnew(SSDev dev, SSParams prm, int f, int with_new)
{
    __shared__ __attribute__((aligned(16))) float pn_lds[2 * 8 * 4 * 64 + 16 * SS_F];      // new-row role: partial sums + the 16 rows; post role: ws + row buffers
    __shared__ int s_slot[16];
    const int s = blockIdx.y;
#ifdef SS_FRAME_STAMPS
    const long long pn_t = wall_clock64();
#endif
    if (blockIdx.x < SS_PN_POST) {
        const int w = threadIdx.x >> 6, n = dev.n_post[s];
        double* ws = reinterpret_cast<double*>(pn_lds) + w * 72;                           // 8 x 72 doubles = 4608 B
        float* rowbuf = pn_lds + 8 * 72 * 2 + w * SS_F;
        for (int k = blockIdx.x * 8 + w; k < n; k += SS_PN_POST * 8) post_track(dev, prm, s, f, k, ws, rowbuf);
#ifdef SS_FRAME_STAMPS
        if (threadIdx.x == 0 && blockIdx.x == 0 && s == 0) { dev.timeline[3001 * 16 + 0] += wall_clock64() - pn_t; dev.timeline[3001 * 16 + 15] += 1; }
#endif
        return;
    }
    if (!with_new) return;
    newrow_units<true>(dev, prm, s, f, blockIdx.x - SS_PN_POST, SS_PN_NEW, pn_lds, s_slot, pn_lds + 2 * 8 * 4 * 64);
#ifdef SS_FRAME_STAMPS
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == SS_PN_POST && s == 0) { dev.timeline[3002 * 16 + 0] += wall_clock64() - pn_t; dev.timeline[3002 * 16 + 15] += 1; }
#endif
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

// a7 on its own: projected mean z = Hx [n][4] and innovation covariance S = H P H^T + R(x, conf) [n][16] (NSA noise)
__global__ void k_kat_project(const double* mean, const double* cov, const double* conf, int n, double wp, double* zmean, double* S)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double m[8], c[64], m4[4], s16[16];
    for (int k = 0; k < 8; ++k) m[k] = mean[(size_t)i * 8 + k];
    for (int k = 0; k < 64; ++k) c[k] = cov[(size_t)i * 64 + k];
    ss_kf_project(m, c, conf ? conf[i] : 0.0, wp, m4, s16);
    for (int k = 0; k < 4; ++k) zmean[(size_t)i * 4 + k] = m4[k];
    for (int k = 0; k < 16; ++k) S[(size_t)i * 16 + k] = s16[k];
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
size_t ss_lsap_lds_bytes() { return 256 * 4; }
size_t ss_assoc_lds_bytes() { return 2 * SS_TILE_FLOATS * 4 + 7 * 2048 + 64 + 256; }       // B pair, hand-over slots, flags, column map

extern "C" void ss_step_kernel_attr()
{
    // a failure here surfaces as a launch error on first use (checked with hipGetLastError after every launch)
    (void)hipFuncSetAttribute((const void*)k_frame, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ss_frame_lds_bytes(SS_COST_CAP, SS_MAXT, SS_MAXD));
    for (const void* k : { (const void*)k_assoc<false, 0>, (const void*)k_assoc<true, 0>, (const void*)k_assoc<false, 1>, (const void*)k_assoc<true, 1>,
                           (const void*)k_assoc<false, 2>, (const void*)k_assoc<true, 2>, (const void*)k_assoc<false, 4>, (const void*)k_assoc<true, 4>,
                           (const void*)k_assoc<false, 5>, (const void*)k_assoc<true, 5> })
        (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ss_assoc_lds_bytes());
}

template <bool TL, int NP>
static void launch_assoc(const SSDev& dev, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1)
{
    const int* items = reinterpret_cast<const int*>(dev.items);
    if (ev0) hipExtLaunchKernelGGL((k_assoc<TL, NP>), dim3(dev.cos_grid), dim3(512), ss_assoc_lds_bytes(), st, ev0, ev1, 0, items, (const int*)dev.n_items, dev.items_cap, dev);
    else hipLaunchKernelGGL((k_assoc<TL, NP>), dim3(dev.cos_grid), dim3(512), ss_assoc_lds_bytes(), st, items, (const int*)dev.n_items, dev.items_cap, dev);
}

// One group of dev.F frames for every stream, first part: feature prep + work lists, the association kernel.  ev0/ev1
// (optional) bracket the association kernel's dispatch.
void ss_launch_group_head(const SSDev& dev, const SSParams& prm, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev_assoc)
{
    hipLaunchKernelGGL(k_group_prep, dim3(dev.S, 1 + SS_MAXT + dev.F * SS_PREP_FBLK), dim3(256), 0, st, dev);
    const bool tl = dev.ts_enable > 1 && dev.cos_grid <= 2048;   // the timeline instantiation (stamps cost registers; its buffer holds 2048 workgroups' shader-clock stamps)
    switch (dev.assoc_stage) {
    case 1: tl ? launch_assoc<true, 1>(dev, st, nullptr, nullptr) : launch_assoc<false, 1>(dev, st, ev0, ev1); break;
    case 2: tl ? launch_assoc<true, 2>(dev, st, nullptr, nullptr) : launch_assoc<false, 2>(dev, st, ev0, ev1); break;
    case 4: tl ? launch_assoc<true, 4>(dev, st, nullptr, nullptr) : launch_assoc<false, 4>(dev, st, ev0, ev1); break;
    case 5: tl ? launch_assoc<true, 5>(dev, st, nullptr, nullptr) : launch_assoc<false, 5>(dev, st, ev0, ev1); break;
    default: tl ? launch_assoc<true, 0>(dev, st, nullptr, nullptr) : launch_assoc<false, 0>(dev, st, ev0, ev1); break;
    }
    if (ev_assoc) (void)hipEventRecord(ev_assoc, st);          // the caller's "association done" event (ss_track_set_assoc_event)
}

// ... and the group's per-frame chain (3 F - 1 dependent launches).  ss_api.hip replays it as ONE captured HIP graph per
// (frames, buffers) when the caller's stream is not itself being captured.
void ss_launch_group_chain(const SSDev& dev, const SSParams& prm, hipStream_t st)
{
    for (int f = 0; f < dev.F; ++f) {
        hipLaunchKernelGGL(k_frame, dim3(dev.S), dim3(256), ss_frame_lds_bytes(dev.cap_cost, dev.cap_t, dev.cap_d), st, dev, prm, f);
        if (dev.chain_merge) {
            const int with_new = f + 1 < dev.F;
            hipLaunchKernelGGL(k_postnew, dim3(SS_PN_POST + (with_new ? SS_PN_NEW : 0), dev.S), dim3(512), 0, st, dev, prm, f, with_new);
        } else {
            hipLaunchKernelGGL(k_post, dim3(dev.S, SS_POST_BLOCKS), dim3(256), 0, st, dev, prm, f);
            if (f + 1 < dev.F) hipLaunchKernelGGL(k_newrow, dim3(64, dev.S), dim3(512), 0, st, dev, prm, f);
        }
    }
}

void ss_launch_normalize(const float* raw, int n, float* unit, hipStream_t st)
{ if (n) hipLaunchKernelGGL(k_kat_normalize, dim3((n + 3) / 4), dim3(256), 0, st, raw, n, unit); }
void ss_launch_ema(const float* s, const float* f, int n, float a, float b, float* o, hipStream_t st)
{ if (n) hipLaunchKernelGGL(k_kat_ema, dim3((n + 3) / 4), dim3(256), 0, st, s, f, n, a, b, o); }
void ss_launch_kf(int op, double* mean, double* cov, const double* z, const double* conf, int n, double wp, double wv, hipStream_t st)
{ if (n) hipLaunchKernelGGL(k_kat_kf, dim3((n + 63) / 64), dim3(64), 0, st, op, mean, cov, z, conf, n, wp, wv); }
void ss_launch_project(const double* mean, const double* cov, const double* conf, int n, double wp, double* zmean, double* S, hipStream_t st)
{ if (n) hipLaunchKernelGGL(k_kat_project, dim3((n + 63) / 64), dim3(64), 0, st, mean, cov, conf, n, wp, zmean, S); }
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
