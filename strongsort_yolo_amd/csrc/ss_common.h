// ss_common.h — shared definitions of the gfx950 StrongSORT hot path (device math + layouts).
//
// Every arithmetic helper here follows the operation order frozen in oracle/DECISIONS.md
// ("exactness contract"); the library is compiled with -ffp-contract=off so only the explicit
// fma()/fmaf()/MFMA calls fuse, exactly as in the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/strongsort_hip.h"

#define SS_F 512              // feature width (OSNet)
#define SS_SEG 64             // dot-product segment (one wave of the cosine kernel per segment)
#define SS_NSEG (SS_F / SS_SEG)
#define SS_TILE 16            // MFMA 16x16 tile edge (v_mfma_f32_16x16x4_f32)
#define SS_NRT 8              // gallery row tiles per track  (capacity 128 rows >= nn_budget)
#define SS_TILE_FLOATS (SS_F * SS_TILE)   // 8192 floats = 32 KiB per (track,row tile) / (col tile)
#define SS_MAXT 256           // track slots per stream
#define SS_MAXD 128           // detections per stream per frame
#define SS_NCT (SS_MAXD / SS_TILE)
#define SS_COST_CAP 12288     // LDS-resident cost entries (f64) of the per-frame kernel; larger matrices spill to HBM
#define SS_FMAX 32            // frames of a stream that one tracker call (group) may carry
#define SS_TLMAX (SS_MAXT * SS_NRT)          // gallery tiles of a stream
#define SS_PLMAX (SS_FMAX * SS_NCT / 2)      // column-tile pairs of a stream's group
#define SS_NCTP (SS_FMAX * SS_NCT)           // packed column tiles of a stream's group at most (every detection of every frame)
#define SS_RECT 32            // gallery tiles per association work record at most (the record carries their tile words)
#define SS_RECW (4 + SS_RECT) // ints per record: {stream, frame, pair word, tiles<<16 | composite<<31} + SS_RECT words (144 bytes)
#define SS_RECI4 (SS_RECW / 4)

#define SS_TENTATIVE 1
#define SS_CONFIRMED 2
#define SS_DELETED 3

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Fragment-major feature layout (16-row gallery tiles and 16-detection column tiles):
//   float index = ((q*4 + ks)*16 + i)*4 + c   holds element k = 16q + 4c + ks of row i of the tile
// (q = k/16, ks = k%4, c = (k%16)/4), so lane l = ks*16+i of a wave reads float4 #(q*64 + l):
// 1 KiB contiguous per wave instruction, and component c of that float4 is the A (or B) operand of
// v_mfma_f32_16x16x4_f32 number 4*(q%4)+c of k-segment q/4 — the instruction consumes k = 16q+4c+{0..3}
// in ascending order, i.e. the fmaf chain of oracle so_dot().
__host__ __device__ inline int ss_frag_index(int i, int k)
{
    return (((k >> 4) * 4 + (k & 3)) * 16 + i) * 4 + ((k & 15) >> 2);
}

struct SSParams {
    double max_dist, max_iou_distance, mc_lambda, gating_threshold, gated_cost, wp, wv;
    float ema_alpha, ema_one_minus_alpha;
    int max_age, n_init, nn_budget;
    int debug;
};

// Device-resident tracker state + per-group scratch for S streams (all pointers device memory).
// One tracker call carries a GROUP of F <= SS_FMAX consecutive frames of every stream; arrays marked [F] are indexed
// by the frame's position in the group, in the caller's layout [F][S][...].
struct SSDev {
    int S, F;
    int budget;                 // nn_budget (ring length of a gallery)
    int cos_grid;               // workgroups of the persistent association kernel
    int comp_rows;              // ragged last gallery tiles of <= comp_rows rows are cut into 4-row groups (composite tiles)
    int xcd_map;                // work-list placement: 0 a gallery range per XCD, 1 a detection column-tile pair per XCD
    int assoc_stage;            // how k_assoc stages a record's detection operand: 0 registers, 1 / 2 / 4 LDS-DMA in that many pieces
    // persistent per stream
    int *n_tracks, *next_id, *frame, *err;
    int* order;                 // [S][MAXT] slot ids in track-list order
    // persistent per slot  [S][MAXT]
    int *slot_used, *track_id, *state, *hits, *age, *tsu, *class_id, *det_idx, *gal_count, *gal_head;
    float* conf;
    double *mean, *cov;         // [S][MAXT][8], [S][MAXT][64]: a track's state after its last frame
    double *mean_p, *cov_p;     // the same PREDICTED one frame ahead (written by post_track for every live track): with pred_ahead and no
                                //   camera motion k_frame only reads the 24 values its gate needs instead of loading, predicting and storing 72
                                //   INVARIANT: post_track of the previous frame is the ONLY writer of mean_p / cov_p and the device chain the only owner of
                                //   smooth_sel.  Nothing imports track state from the host today; a path that ever writes mean / cov / smooth from outside
                                //   (track import, checkpoint restore) must also rebuild mean_p / cov_p (ss_kf_predict) and reset smooth_sel, or switch
                                //   pred_ahead off for the next frame — k_frame would otherwise gate on stale predictions without any error.
    int pred_ahead;             // ss_set_option "pred_ahead" (default 1)
    float* smooth;              // [S][MAXT][2][512] EMA feature, double-buffered: the row in use is [smooth_sel]; an update reads it and
                                //   writes the other half, so the new-row units of the SAME launch (k_postnew) can still read the old one
    int* smooth_sel;            // [S][MAXT] 0 / 1
    float* gallery;             // [S][MAXT][NRT][TILE_FLOATS]  fragment-major ring of nn_budget rows
    // group inputs / outputs (caller's buffers)
    const float* dets;          // [F][S][MAXD][6]
    const int* n_dets;          // [F][S]
    const float* feats_raw;     // [F][S][MAXD][512]
    float* out_rows;            // [F][S][MAXT][8]
    int* n_out;                 // [F][S]
    int* img_hw;                // [S][2]
    // group scratch
    float* feat_unit;           // [FMAX][S][MAXD][512]
    float* feat_frag;           // [FMAX][S][NCT][TILE_FLOATS]
    float* feat_pack;           // [S][NCTP][TILE_FLOATS] the group's detections PACKED across frames (column g = detections of the earlier
                                //   frames + d): k_assoc's B operand when assoc_pack — 28 column-tile pairs instead of 32 at ~28 det/frame
    int* colmap;                // [S][FMAX*MAXD + 64] packed column g -> frame<<8 | detection, -1 behind the last one
    int assoc_pack;             // 1: k_assoc works on packed pairs (ss_set_option "assoc_pack"), 0: pairs of one frame's column tiles
    double *tlwh, *xyah;        // [FMAX][S][MAXD][4]
    int* M;                     // [S][MAXT][FMAX][MAXD] ordered-int keys (ss_fkey) of the appearance distance
                                //   min over the gallery rows of (slot) that are valid in frame f of the group
    int2* pl;                   // [S][PLMAX] column-tile pairs of the group {frame, ct0 | two<<8 | D<<16}, by frame
    int* n_pl;                  // [S]
    int* pf;                    // [S][FMAX+1] first pair of frame f (pf[F] = n_pl)
    int4* items;                // [8][items_cap][SS_RECI4] association work records per XCD: {stream, frame, pair word, tiles<<16 |
                                //   composite<<31} + SS_RECT words: packed tiles (slot | row tile<<8 | count<<12 | ring head<<20) or, in a
                                //   composite record, 4 packed row groups per tile (slot | row tile<<8 | r4<<11 | count<<13 | head<<21);
                                //   snapshot at group start
    int items_cap;
    int* n_items;               // [8] (re-armed by k_frame)
    // per-frame hand-off k_frame -> k_post -> k_newrow
    int4* post;                 // [S][MAXT] surviving tracks in list order {slot, det or -1, flags, aux}
    int* n_post;                // [S]
    int* rowlist;               // [S][MAXT] every gallery row appended this frame: slot | first_row<<16 | smooth_sel at frame start<<17 |
                                //   matched<<18 | matched detection<<19 (the row = EMA(smooth[sel], feature of that detection), or smooth[sel] itself)
    int chain_merge;            // 1: k_post and k_newrow of a frame as ONE launch (k_postnew), 0: two launches (A/B, ss_set_option "chain_merge")
    int* n_rows;                // [S]
    const double* cmc;          // [F][S][8] camera-motion warps of the group (ss_track_set_cmc) or NULL
    double* cost_spill;         // [S][MAXT*MAXD] cost matrices that do not fit the LDS
    double* frame_scratch;      // [S][MAXT*18 + MAXD*8] k_frame's per-track / per-detection f64 work arrays when they exceed cap_t / cap_d
    int cap_cost, cap_t, cap_d; // what k_frame keeps in LDS: cost entries, tracks, detections (ss_set_option "frame_caps")
    unsigned long long* tstamp; // [4] in-kernel timing of the association kernel: min start, max end (100 MHz), sum, count
    int ts_enable;              // 1: first-start / last-end stamps; 2: + per-workgroup timeline
    long long* timeline;        // [4096 workgroups][16] stamps of each workgroup's first item (profiling aid)
    // debug (stage intermediates, per frame of the group)
    float* dbg_cos;             // [FMAX][S][MAXT][MAXD]
    double *dbg_maha, *dbg_cost_a, *dbg_cost_b;
    uint8_t* dbg_gated;
    int* dbg_lists;             // [FMAX][S][4][MAXT]: pairs_a(det per conf row), cand, cols_b, pairs_b
    int* dbg_counts;            // [FMAX][S][4]: n_conf, n_cand, n_cols, n_dets
};

// post[] flags
#define SS_P_MATCHED 1          // Kalman update + EMA with detection .y
#define SS_P_BIRTH 2            // new track from detection .y
#define SS_P_APPEND 4           // append the EMA feature at ring position aux & 0xff
#define SS_P_FIRSTROW 8         // ... and it is the gallery's first row
#define SS_P_EMIT 16            // output row number aux >> 8

// order-preserving int image of a float (signed compare == float compare, also for negative values)
__host__ __device__ inline int ss_fkey(float v)
{
    union { float f; int i; } u; u.f = v;
    return u.i ^ ((u.i >> 31) & 0x7fffffff);
}
__host__ __device__ inline float ss_fkey_inv(int k)
{
    union { float f; int i; } u; u.i = k ^ ((k >> 31) & 0x7fffffff);
    return u.f;
}
#define SS_KEY_INF 0x7f800000

// ---------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------
#define SS_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ inline float ss_wave_sumsq_reduce(float p)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off);
    return p;
}

// ---------------------------------------------------------------------------------------------
// Kalman filter (float64), mirrors oracle so_kf_* line by line
// ---------------------------------------------------------------------------------------------
__device__ inline void ss_kf_initiate(const double z[4], double wp, double wv, double* mean, double* cov)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) { mean[i] = z[i]; mean[4 + i] = 0.0; }
    double h = z[3];
    double sd[8] = { 2.0 * wp * h, 2.0 * wp * h, 1e-2, 2.0 * wp * h,
                     10.0 * wv * h, 10.0 * wv * h, 1e-5, 10.0 * wv * h };
    for (int i = 0; i < 64; ++i) cov[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) cov[i * 8 + i] = sd[i] * sd[i];
}

__device__ inline void ss_kf_predict(double* mean, double* cov, double wp, double wv)
{
    double h = mean[3];
    double sp = wp * h, sv = wv * h;
    double sd[8] = { sp, sp, 1e-2, sp, sv, sv, 1e-5, sv };
    double P[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) P[i] = cov[i];
    // A = P F^T (in place on the left half), B = F A (in place on the top half)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) P[i * 8 + j] = P[i * 8 + j] + P[i * 8 + j + 4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) P[i * 8 + j] = P[i * 8 + j] + P[(i + 4) * 8 + j];
#pragma unroll
    for (int i = 0; i < 8; ++i) P[i * 8 + i] = P[i * 8 + i] + sd[i] * sd[i];
#pragma unroll
    for (int i = 0; i < 64; ++i) cov[i] = P[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) mean[i] = mean[i] + mean[i + 4];
}

__device__ inline void ss_kf_project(const double* mean, const double* cov, double conf, double wp,
                                     double m4[4], double S[16])
{
    double h = mean[3];
    double sd[4] = { wp * h, wp * h, 1e-1, wp * h };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m4[i] = mean[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) S[i * 4 + j] = cov[i * 8 + j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double s = (1.0 - conf) * sd[i];
        S[i * 4 + i] = S[i * 4 + i] + s * s;
    }
}

__device__ inline void ss_chol4(const double S[16], double L[16])
{
#pragma unroll
    for (int i = 0; i < 16; ++i) L[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double sum = S[i * 4 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) sum = fma(-L[i * 4 + k], L[j * 4 + k], sum);
            L[i * 4 + j] = (i == j) ? sqrt(sum) : sum / L[j * 4 + j];
        }
}

// squared Mahalanobis distance given L (lower Cholesky of the projected covariance) and m4
__device__ inline double ss_maha(const double L[16], const double m4[4], const double z[4])
{
    double y[4], acc = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double sum = z[i] - m4[i];
#pragma unroll
        for (int k = 0; k < i; ++k) sum = fma(-L[i * 4 + k], y[k], sum);
        y[i] = sum / L[i * 4 + i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = fma(y[i], y[i], acc);
    return acc;
}

__device__ inline void ss_kf_update(double* mean, double* cov, const double z[4], double conf, double wp)
{
    double m4[4], S[16], L[16], K[32], M[32], y[4];
    ss_kf_project(mean, cov, conf, wp, m4, S);
    ss_chol4(S, L);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        double w[4], x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double sum = cov[r * 8 + i];
#pragma unroll
            for (int k = 0; k < i; ++k) sum = fma(-L[i * 4 + k], w[k], sum);
            w[i] = sum / L[i * 4 + i];
        }
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            double sum = w[i];
#pragma unroll
            for (int k = 3; k > i; --k) sum = fma(-L[k * 4 + i], x[k], sum);
            x[i] = sum / L[i * 4 + i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) K[r * 4 + i] = x[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = z[i] - m4[i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fma(S[i * 4 + k], K[c * 4 + k], acc);
            M[i * 8 + c] = acc;
        }
    double nm[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = fma(y[k], K[r * 4 + k], acc);
        nm[r] = mean[r] + acc;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fma(K[r * 4 + k], M[k * 8 + c], acc);
            cov[r * 8 + c] = cov[r * 8 + c] - acc;
        }
#pragma unroll
    for (int r = 0; r < 8; ++r) mean[r] = nm[r];
}

// Wave-cooperative form of ss_kf_update: lane l owns covariance entry (r, c) = (l >> 3, l & 7) and computes exactly
// the operations the oracle performs for that entry (the gain rows r and c, column c of S K^T), so every result bit
// equals the one-thread form.  ws = 72 doubles of per-wave LDS (covariance, then mean; the new mean is left in
// ws[64..71]).  gmean / gcov are updated in place.
// (imean / icov: the state it starts from — the same arrays, or the predicted copies; the new covariance is ALSO left in ws[0..63])
__device__ inline void ss_kf_update_wave(const double* imean, const double* icov, double* gmean, double* gcov, const double z[4], double conf,
                                         double wp, double* ws)
{
    const int l = threadIdx.x & 63, r = l >> 3, c = l & 7;
    const double p = icov[l];
    ws[l] = p;
    if (l < 8) ws[64 + l] = imean[l];
    SS_WAVE_SYNC();
    const double* cov = ws;
    const double* mean = ws + 64;
    double m4[4], S[16], L[16];
    ss_kf_project(mean, cov, conf, wp, m4, S);
    ss_chol4(S, L);
    double Kr[4], Kc[4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass == 0 ? r : c;
        double w[4], x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double sum = cov[row * 8 + i];
#pragma unroll
            for (int k = 0; k < i; ++k) sum = fma(-L[i * 4 + k], w[k], sum);
            w[i] = sum / L[i * 4 + i];
        }
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            double sum = w[i];
#pragma unroll
            for (int k = 3; k > i; --k) sum = fma(-L[k * 4 + i], x[k], sum);
            x[i] = sum / L[i * 4 + i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { if (pass == 0) Kr[i] = x[i]; else Kc[i] = x[i]; }
    }
    double Mc[4];                                   // column c of M = S K^T
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = fma(S[i * 4 + k], Kc[k], acc);
        Mc[i] = acc;
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc = fma(Kr[k], Mc[k], acc);
    const double mr = mean[r];
    double a2 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) a2 = fma(z[k] - m4[k], Kr[k], a2);
    SS_WAVE_SYNC();                                 // every lane has read the old state
    const double pn = p - acc;
    gcov[l] = pn;
    ws[l] = pn;
    if (c == 0) { const double nm = mr + a2; gmean[r] = nm; ws[64 + r] = nm; }
    SS_WAVE_SYNC();
}

// Kalman prediction of a state held in ws (covariance [64], mean [8]) by the 64 lanes of a wave, lane l = entry (l >> 3, l & 7):
// exactly the operations ss_kf_predict performs for that entry (A = P F^T on the left half, B = F A on the top half, + Q on the
// diagonal; h = the mean's height BEFORE the step).  -> pmean[8], pcov[64] (global).
__device__ inline void ss_kf_predict_wave(const double* ws, double wp, double wv, double* pmean, double* pcov)
{
    const int l = threadIdx.x & 63, i = l >> 3, j = l & 7;
    const double* P = ws;
    const double* mean = ws + 64;
    const double h = mean[3];
    const double sp = wp * h, sv = wv * h;
    double a = P[i * 8 + j];
    if (j < 4) a = a + P[i * 8 + j + 4];
    if (i < 4) {
        double a4 = P[(i + 4) * 8 + j];
        if (j < 4) a4 = a4 + P[(i + 4) * 8 + j + 4];
        a = a + a4;
    }
    if (i == j) {
        const double sd = (i == 2) ? 1e-2 : (i == 6) ? 1e-5 : (i < 4) ? sp : sv;
        a = a + sd * sd;
    }
    pcov[l] = a;
    if (l < 8) pmean[l] = l < 4 ? mean[l] + mean[l + 4] : mean[l];
}

// camera-motion warp m (2x3, full-frame pixels) applied to a track's box (oracle so_camera_update, D-18)
__device__ inline void ss_camera_update(double* mean, const double* m)
{
    const double w = mean[2] * mean[3], h = mean[3];
    const double x1 = mean[0] - w / 2, y1 = mean[1] - h / 2, x2 = x1 + w, y2 = y1 + h;
    const double ax = (m[0] * x1 + m[1] * y1) + m[2], ay = (m[3] * x1 + m[4] * y1) + m[5];
    const double bx = (m[0] * x2 + m[1] * y2) + m[2], by = (m[3] * x2 + m[4] * y2) + m[5];
    const double nw = bx - ax, nh = by - ay;
    mean[0] = ax + nw / 2; mean[1] = ay + nh / 2; mean[2] = nw / nh; mean[3] = nh;
}

// gate + blend + threshold of one entry (oracle so_blend)
__device__ inline double ss_blend(float cosd, double maha, const SSParams& p, int* gated)
{
    double oml = 1.0 - p.mc_lambda, repl = p.max_dist + 1e-5;
    int g = maha > p.gating_threshold;
    double c = g ? p.gated_cost : (double)cosd;
    double t1 = p.mc_lambda * c, t2 = oml * maha;
    double v = t1 + t2;
    if (v > p.max_dist) v = repl;
    *gated = g;
    return v;
}

// IoU cost of one (track tlwh, det tlwh) pair (oracle so_iou_cost)
__device__ inline double ss_iou_cost(const double t[4], const double c[4], double max_dist)
{
    double repl = max_dist + 1e-5;
    double tbr0 = t[0] + t[2], tbr1 = t[1] + t[3], tarea = t[2] * t[3];
    double cbr0 = c[0] + c[2], cbr1 = c[1] + c[3];
    double tl0 = fmax(t[0], c[0]), tl1 = fmax(t[1], c[1]);
    double br0 = fmin(tbr0, cbr0), br1 = fmin(tbr1, cbr1);
    double w = fmax(0.0, br0 - tl0), h = fmax(0.0, br1 - tl1);
    double inter = w * h, carea = c[2] * c[3];
    double iou = inter / (tarea + carea - inter);
    double v = 1.0 - iou;
    if (v > max_dist) v = repl;
    return v;
}
