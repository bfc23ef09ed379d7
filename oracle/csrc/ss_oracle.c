/*
 * ss_oracle.c — CPU restatement of the StrongSORT per-frame hot path with a DEFINED operation
 * order ("exact oracle").  TEST INFRASTRUCTURE ONLY: nothing in the product path may link, load
 * or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED: the reference snapshot (/root/reference) contains no tracker,
 * Kalman, NMS or ReID source — its whole hot path is the opaque call
 * /root/reference/yolo_multi_model.py:41 (model.track) / :173 (model.predict) into the unpinned,
 * un-vendored `ultralytics` package, and yolov5/ yolov7/ (where StrongSORT once lived) are empty.
 * Every function below therefore restates the *published* algorithm named in BASELINE.json's
 * north_star (StrongSORT / DeepSORT / torchvision-style NMS / Ultralytics LetterBox) and cites the
 * reference call site it stands behind plus the DECISIONS.md entry that freezes each constant.
 *
 * Why C and a fixed order: track IDs are decided by threshold compares and LSAP tie-breaks, so
 * "identical IDs" needs bit-identical floats, not close ones.  Each reduction here is written in
 * the exact order the gfx950 kernels use (f32 fmaf chains == v_mfma_f32_32x32x2_f32 accumulation,
 * measured bit-identical: profiles/r01_probe_exact.txt); IEEE sqrt/div are correctly rounded on
 * both sides.  Build with -ffp-contract=off (oracle/Makefile): only explicit fma()/fmaf() fuse.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SO_SEG 64

/* ------------------------------------------------------------------------------------------------
 * E1/E2  appearance: cosine distance to the nearest gallery row
 * stands behind yolo_multi_model.py:41 (tracker association inside model.track); algorithm =
 * DeepSORT NearestNeighborDistanceMetric("cosine") with unit-norm rows (DECISIONS D-03, D-04).
 * dot := 8 segments of 64; inside a segment an fmaf chain from 0 in ascending k; segment sums
 * added left to right.
 * ---------------------------------------------------------------------------------------------- */
float so_dot(const float* g, const float* f, int F)
{
    float total = 0.0f;
    for (int s = 0; s < F / SO_SEG; ++s) {
        float p = 0.0f;
        for (int k = s * SO_SEG; k < (s + 1) * SO_SEG; ++k) p = fmaf(g[k], f[k], p);
        total = (s == 0) ? p : total + p;
    }
    return total;
}

/* out[d] = min_b (1 - dot(gallery[b], feats[d])), b < count.  count == 0 -> +inf. */
void so_cosine_min(const float* gallery, int count, const float* feats, int D, int F, float* out)
{
    /* same arithmetic as so_dot per (b,d); 8 detections advance together only so the independent
     * fmaf chains overlap in the CPU pipeline (no effect on any result bit). */
    for (int d = 0; d < D; ++d) out[d] = INFINITY;
    for (int d0 = 0; d0 < D; d0 += 8) {
        const float* fp[8];
        for (int j = 0; j < 8; ++j) fp[j] = feats + (size_t)((d0 + j < D) ? d0 + j : D - 1) * F;
        for (int b = 0; b < count; ++b) {
            const float* g = gallery + (size_t)b * F;
            float total[8];
            for (int s = 0; s < F / SO_SEG; ++s) {
                float p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0;
                for (int k = s * SO_SEG; k < (s + 1) * SO_SEG; ++k) {
                    float gk = g[k];
                    p0 = fmaf(gk, fp[0][k], p0); p1 = fmaf(gk, fp[1][k], p1);
                    p2 = fmaf(gk, fp[2][k], p2); p3 = fmaf(gk, fp[3][k], p3);
                    p4 = fmaf(gk, fp[4][k], p4); p5 = fmaf(gk, fp[5][k], p5);
                    p6 = fmaf(gk, fp[6][k], p6); p7 = fmaf(gk, fp[7][k], p7);
                }
                float p[8] = { p0, p1, p2, p3, p4, p5, p6, p7 };
                for (int j = 0; j < 8; ++j) total[j] = (s == 0) ? p[j] : total[j] + p[j];
            }
            for (int j = 0; j < 8 && d0 + j < D; ++j) {
                float dist = 1.0f - total[j];
                if (dist < out[d0 + j]) out[d0 + j] = dist;
            }
        }
    }
}

/* E7  sum of squares with the wave-64 shape: lane l chains k = l, l+64, ...; xor-butterfly. */
float so_sumsq(const float* v, int F)
{
    float p[64];
    for (int l = 0; l < 64; ++l) {
        float a = 0.0f;
        for (int k = l; k < F; k += 64) a = fmaf(v[k], v[k], a);
        p[l] = a;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        float q[64];
        for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
        memcpy(p, q, sizeof p);
    }
    return p[0];
}

/* E8  L2 normalise one feature row (Detection.feature / Track feature normalisation, D-03). */
void so_normalize(const float* in, float* out, int F)
{
    float n = sqrtf(so_sumsq(in, F));
    for (int k = 0; k < F; ++k) out[k] = n > 0.0f ? in[k] / n : 0.0f;      /* an all-zero row stays zero (D-17), no NaN */
}

/* E7  EMA feature update + renormalise (StrongSORT Track.update, D-05). */
void so_ema(const float* smooth, const float* feat, float alpha, float one_minus_alpha, float* out, int F)
{
    float tmp[4096];
    for (int k = 0; k < F; ++k) {
        float a = alpha * smooth[k];
        float b = one_minus_alpha * feat[k];
        tmp[k] = a + b;
    }
    so_normalize(tmp, out, F);
}

/* ------------------------------------------------------------------------------------------------
 * E3  Kalman filter, 8-state constant velocity (cx, cy, a, h, v...), all float64.
 * stands behind yolo_multi_model.py:41; algorithm = DeepSORT KalmanFilter with the NSA noise
 * scaling of StrongSORT (D-06..D-08).  cov is row-major 8x8.
 * ---------------------------------------------------------------------------------------------- */
void so_kf_initiate(const double z[4], double wp, double wv, double mean[8], double cov[64])
{
    for (int i = 0; i < 4; ++i) { mean[i] = z[i]; mean[4 + i] = 0.0; }
    double h = z[3];
    double sd[8] = { 2.0 * wp * h, 2.0 * wp * h, 1e-2, 2.0 * wp * h,
                     10.0 * wv * h, 10.0 * wv * h, 1e-5, 10.0 * wv * h };
    memset(cov, 0, 64 * sizeof(double));
    for (int i = 0; i < 8; ++i) cov[i * 8 + i] = sd[i] * sd[i];
}

void so_kf_predict(double mean[8], double cov[64], double wp, double wv)
{
    double h = mean[3];
    double sp = wp * h, sv = wv * h;
    double sd[8] = { sp, sp, 1e-2, sp, sv, sv, 1e-5, sv };
    double A[64], B[64];
    /* A = P F^T : A[i][j] = P[i][j] + P[i][j+4] (j<4) */
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j)
            A[i * 8 + j] = (j < 4) ? cov[i * 8 + j] + cov[i * 8 + j + 4] : cov[i * 8 + j];
    /* B = F A : B[i][j] = A[i][j] + A[i+4][j] (i<4) */
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j)
            B[i * 8 + j] = (i < 4) ? A[i * 8 + j] + A[(i + 4) * 8 + j] : A[i * 8 + j];
    for (int i = 0; i < 8; ++i) B[i * 8 + i] = B[i * 8 + i] + sd[i] * sd[i];
    memcpy(cov, B, sizeof B);
    for (int i = 0; i < 4; ++i) mean[i] = mean[i] + mean[i + 4];
}

/* project to measurement space; NSA: std scaled by (1 - conf) (conf = 0 when gating). */
void so_kf_project(const double mean[8], const double cov[64], double conf, double wp,
                   double m4[4], double S[16])
{
    double h = mean[3];
    double sd[4] = { wp * h, wp * h, 1e-1, wp * h };
    for (int i = 0; i < 4; ++i) {
        m4[i] = mean[i];
        for (int j = 0; j < 4; ++j) S[i * 4 + j] = cov[i * 8 + j];
    }
    for (int i = 0; i < 4; ++i) {
        double s = (1.0 - conf) * sd[i];
        S[i * 4 + i] = S[i * 4 + i] + s * s;
    }
}

/* lower Cholesky, row by row; inner sums are fma chains (sum = fma(-Lik, Ljk, sum)). */
void so_chol4(const double S[16], double L[16])
{
    memset(L, 0, 16 * sizeof(double));
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j <= i; ++j) {
            double sum = S[i * 4 + j];
            for (int k = 0; k < j; ++k) sum = fma(-L[i * 4 + k], L[j * 4 + k], sum);
            L[i * 4 + j] = (i == j) ? sqrt(sum) : sum / L[j * 4 + j];
        }
}

/* squared Mahalanobis distance of every measurement row Z[d] (xyah) to the track (D-07). */
void so_gating(const double mean[8], const double cov[64], const double* Z, int D, double wp,
               double* out)
{
    double m4[4], S[16], L[16];
    so_kf_project(mean, cov, 0.0, wp, m4, S);
    so_chol4(S, L);
    for (int d = 0; d < D; ++d) {
        double y[4], acc = 0.0;
        for (int i = 0; i < 4; ++i) {
            double sum = Z[d * 4 + i] - m4[i];
            for (int k = 0; k < i; ++k) sum = fma(-L[i * 4 + k], y[k], sum);
            y[i] = sum / L[i * 4 + i];
        }
        for (int i = 0; i < 4; ++i) acc = fma(y[i], y[i], acc);
        out[d] = acc;
    }
}

void so_kf_update(double mean[8], double cov[64], const double z[4], double conf, double wp)
{
    double m4[4], S[16], L[16], K[32], M[32], y[4];
    so_kf_project(mean, cov, conf, wp, m4, S);
    so_chol4(S, L);
    /* K[r,:] = solve(S, P[r,0:4]) via L then L^T */
    for (int r = 0; r < 8; ++r) {
        double w[4], x[4];
        for (int i = 0; i < 4; ++i) {
            double sum = cov[r * 8 + i];
            for (int k = 0; k < i; ++k) sum = fma(-L[i * 4 + k], w[k], sum);
            w[i] = sum / L[i * 4 + i];
        }
        for (int i = 3; i >= 0; --i) {
            double sum = w[i];
            for (int k = 3; k > i; --k) sum = fma(-L[k * 4 + i], x[k], sum);
            x[i] = sum / L[i * 4 + i];
        }
        for (int i = 0; i < 4; ++i) K[r * 4 + i] = x[i];
    }
    for (int i = 0; i < 4; ++i) y[i] = z[i] - m4[i];
    /* M = S K^T (4x8);  cov -= K M */
    for (int i = 0; i < 4; ++i)
        for (int c = 0; c < 8; ++c) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc = fma(S[i * 4 + k], K[c * 4 + k], acc);
            M[i * 8 + c] = acc;
        }
    double newcov[64];
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc = fma(K[r * 4 + k], M[k * 8 + c], acc);
            newcov[r * 8 + c] = cov[r * 8 + c] - acc;
        }
    for (int r = 0; r < 8; ++r) {
        double acc = 0.0;
        for (int k = 0; k < 4; ++k) acc = fma(y[k], K[r * 4 + k], acc);
        mean[r] = mean[r] + acc;
    }
    memcpy(cov, newcov, sizeof newcov);
}

/* ------------------------------------------------------------------------------------------------
 * E4  gate + blend + threshold of one cost row (StrongSORT gate_cost_matrix + min_cost_matching
 * thresholding, D-09).  cost stays float64 as in the published code.
 * ---------------------------------------------------------------------------------------------- */
void so_blend(const float* cosd, const double* maha, int D, double lambda, double gate_thr,
              double gated_cost, double max_dist, double* cost, uint8_t* gated)
{
    double oml = 1.0 - lambda, repl = max_dist + 1e-5;
    for (int d = 0; d < D; ++d) {
        int g = maha[d] > gate_thr;
        double c = g ? gated_cost : (double)cosd[d];
        double t1 = lambda * c, t2 = oml * maha[d];
        double v = t1 + t2;
        if (v > max_dist) v = repl;
        cost[d] = v;
        if (gated) gated[d] = (uint8_t)g;
    }
}

/* E5  IoU cost row, float64 tlwh boxes (DeepSORT iou_matching, D-10). */
void so_iou_cost(const double t[4], const double* det_tlwh, int D, double max_dist, double* cost)
{
    double repl = max_dist + 1e-5;
    double tbr0 = t[0] + t[2], tbr1 = t[1] + t[3], tarea = t[2] * t[3];
    for (int d = 0; d < D; ++d) {
        const double* c = det_tlwh + d * 4;
        double cbr0 = c[0] + c[2], cbr1 = c[1] + c[3];
        double tl0 = fmax(t[0], c[0]), tl1 = fmax(t[1], c[1]);
        double br0 = fmin(tbr0, cbr0), br1 = fmin(tbr1, cbr1);
        double w = fmax(0.0, br0 - tl0), h = fmax(0.0, br1 - tl1);
        double inter = w * h, carea = c[2] * c[3];
        double iou = inter / (tarea + carea - inter);
        double v = 1.0 - iou;
        if (v > max_dist) v = repl;
        cost[d] = v;
    }
}

/* ------------------------------------------------------------------------------------------------
 * E6  rectangular linear sum assignment — shortest augmenting path (Crouse 2016), the algorithm of
 * scipy.optimize.linear_sum_assignment, restated with the same scan order and tie-break so the
 * pairs are identical to SciPy's even on ties (tests/test_oracle_lsap.py checks this).
 * row_to_col[nr]: assigned column or -1.  returns 0, or -1 if infeasible.
 * ---------------------------------------------------------------------------------------------- */
int so_lsap(int nr0, int nc0, const double* cost0, int* row_to_col)
{
    for (int i = 0; i < nr0; ++i) row_to_col[i] = -1;
    if (nr0 == 0 || nc0 == 0) return 0;
    int transpose = nc0 < nr0;
    int nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;
    double* cost = (double*)malloc(sizeof(double) * nr * nc);
    if (transpose) {
        for (int i = 0; i < nr0; ++i)
            for (int j = 0; j < nc0; ++j) cost[j * nr0 + i] = cost0[i * nc0 + j];
    } else memcpy(cost, cost0, sizeof(double) * nr * nc);

    double* u = (double*)calloc(nr, sizeof(double));
    double* v = (double*)calloc(nc, sizeof(double));
    double* sp = (double*)malloc(sizeof(double) * nc);
    int* path = (int*)malloc(sizeof(int) * nc);
    int* col4row = (int*)malloc(sizeof(int) * nr);
    int* row4col = (int*)malloc(sizeof(int) * nc);
    int* remaining = (int*)malloc(sizeof(int) * nc);
    unsigned char* SR = (unsigned char*)malloc(nr);
    unsigned char* SC = (unsigned char*)malloc(nc);
    for (int j = 0; j < nc; ++j) { path[j] = -1; row4col[j] = -1; }
    for (int i = 0; i < nr; ++i) col4row[i] = -1;
    int rc = 0;

    for (int cur = 0; cur < nr && rc == 0; ++cur) {
        double minVal = 0.0;
        int num_remaining = nc, sink = -1, i = cur;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        memset(SR, 0, nr); memset(SC, 0, nc);
        for (int j = 0; j < nc; ++j) sp[j] = INFINITY;
        while (sink == -1) {
            int index = -1;
            double lowest = INFINITY;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; ++it) {
                int j = remaining[it];
                double r = minVal + cost[i * nc + j] - u[i] - v[j];
                if (r < sp[j]) { path[j] = i; sp[j] = r; }
                if (sp[j] < lowest || (sp[j] == lowest && row4col[j] == -1)) {
                    lowest = sp[j]; index = it;
                }
            }
            minVal = lowest;
            if (minVal == INFINITY) { rc = -1; break; }
            int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        if (rc) break;
        u[cur] += minVal;
        for (int r = 0; r < nr; ++r)
            if (SR[r] && r != cur) u[r] += minVal - sp[col4row[r]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= minVal - sp[j];
        int j = sink;
        for (;;) {
            int r = path[j];
            row4col[j] = r;
            int t = col4row[r]; col4row[r] = j; j = t;
            if (r == cur) break;
        }
    }
    if (rc == 0) {
        if (transpose) { for (int i = 0; i < nr; ++i) row_to_col[col4row[i]] = i; }
        else           { for (int i = 0; i < nr; ++i) row_to_col[i] = col4row[i]; }
    }
    free(cost); free(u); free(v); free(sp); free(path); free(col4row); free(row4col);
    free(remaining); free(SR); free(SC);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * E9  NMS on a detector head tensor, YOLOv8 layout pred[(4+nc+nm) x N] (xywh, class scores, extra).
 * stands behind yolo_multi_model.py:41/:173 with the overrides pinned at :18-21 (conf .3, iou .4,
 * agnostic False, max_det 1000); algorithm = Ultralytics non_max_suppression (best class per
 * anchor, per-class box offset max_wh) + torchvision-style greedy IoU suppression (D-13).
 * keep_idx[k] = anchor index, rows[k] = x1,y1,x2,y2,conf,cls in letterboxed pixels.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float score; int anchor; int cls; } so_cand;

static int so_cand_cmp(const void* a, const void* b)
{
    const so_cand* x = (const so_cand*)a; const so_cand* y = (const so_cand*)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return (x->anchor > y->anchor) - (x->anchor < y->anchor);
}

int so_nms_classes(const float* pred, int N, int nc, float conf_thres, float iou_thres, int agnostic,
                   float max_wh, int max_nms, int max_det, const unsigned char* class_allow, int* keep_idx, float* rows);

int so_nms(const float* pred, int N, int nc, float conf_thres, float iou_thres, int agnostic,
           float max_wh, int max_nms, int max_det, int* keep_idx, float* rows)
{
    return so_nms_classes(pred, N, nc, conf_thres, iou_thres, agnostic, max_wh, max_nms, max_det, 0, keep_idx, rows);
}

/* class_allow[nc] (NULL: all): Ultralytics `classes` filter, applied to the anchor's best class after the confidence
 * test (the override yolo_multi_model.py:22 sets). */
int so_nms_classes(const float* pred, int N, int nc, float conf_thres, float iou_thres, int agnostic,
                   float max_wh, int max_nms, int max_det, const unsigned char* class_allow, int* keep_idx, float* rows)
{
    so_cand* c = (so_cand*)malloc(sizeof(so_cand) * (N > 0 ? N : 1));
    int n = 0;
    for (int a = 0; a < N; ++a) {
        float best = pred[(size_t)4 * N + a]; int bc = 0;
        for (int k = 1; k < nc; ++k) {
            float s = pred[(size_t)(4 + k) * N + a];
            if (s > best) { best = s; bc = k; }
        }
        if (best > conf_thres && (!class_allow || class_allow[bc])) { c[n].score = best; c[n].anchor = a; c[n].cls = bc; ++n; }
    }
    qsort(c, n, sizeof(so_cand), so_cand_cmp);
    if (n > max_nms) n = max_nms;
    float* bx = (float*)malloc(sizeof(float) * 4 * (n > 0 ? n : 1));   /* offset boxes */
    float* ar = (float*)malloc(sizeof(float) * (n > 0 ? n : 1));
    unsigned char* dead = (unsigned char*)calloc(n > 0 ? n : 1, 1);
    for (int i = 0; i < n; ++i) {
        int a = c[i].anchor;
        float cx = pred[a], cy = pred[(size_t)N + a], w = pred[(size_t)2 * N + a], h = pred[(size_t)3 * N + a];
        float hw = w / 2.0f, hh = h / 2.0f;
        float off = agnostic ? 0.0f : (float)c[i].cls * max_wh;
        bx[i * 4 + 0] = (cx - hw) + off; bx[i * 4 + 1] = (cy - hh) + off;
        bx[i * 4 + 2] = (cx + hw) + off; bx[i * 4 + 3] = (cy + hh) + off;
        ar[i] = (bx[i * 4 + 2] - bx[i * 4 + 0]) * (bx[i * 4 + 3] - bx[i * 4 + 1]);
    }
    int kept = 0;
    for (int i = 0; i < n && kept < max_det; ++i) {
        if (dead[i]) continue;
        int a = c[i].anchor;
        float cx = pred[a], cy = pred[(size_t)N + a], w = pred[(size_t)2 * N + a], h = pred[(size_t)3 * N + a];
        float hw = w / 2.0f, hh = h / 2.0f;
        keep_idx[kept] = a;
        rows[kept * 6 + 0] = cx - hw; rows[kept * 6 + 1] = cy - hh;
        rows[kept * 6 + 2] = cx + hw; rows[kept * 6 + 3] = cy + hh;
        rows[kept * 6 + 4] = c[i].score; rows[kept * 6 + 5] = (float)c[i].cls;
        ++kept;
        for (int j = i + 1; j < n; ++j) {
            if (dead[j]) continue;
            float xx1 = fmaxf(bx[i * 4 + 0], bx[j * 4 + 0]), yy1 = fmaxf(bx[i * 4 + 1], bx[j * 4 + 1]);
            float xx2 = fminf(bx[i * 4 + 2], bx[j * 4 + 2]), yy2 = fminf(bx[i * 4 + 3], bx[j * 4 + 3]);
            float iw = fmaxf(0.0f, xx2 - xx1), ih = fmaxf(0.0f, yy2 - yy1);
            float inter = iw * ih;
            float iou = inter / (ar[i] + ar[j] - inter);
            if (iou > iou_thres) dead[j] = 1;
        }
    }
    free(c); free(bx); free(ar); free(dead);
    return kept;
}

/* map letterboxed boxes back to original pixels (Ultralytics scale_boxes, D-13). in place. */
void so_scale_boxes(float* rows, int n, int row_stride, float gain, float pad_x, float pad_y,
                    float w0, float h0)
{
    for (int i = 0; i < n; ++i) {
        float* r = rows + (size_t)i * row_stride;
        float x1 = (r[0] - pad_x) / gain, y1 = (r[1] - pad_y) / gain;
        float x2 = (r[2] - pad_x) / gain, y2 = (r[3] - pad_y) / gain;
        r[0] = fminf(fmaxf(x1, 0.0f), w0); r[1] = fminf(fmaxf(y1, 0.0f), h0);
        r[2] = fminf(fmaxf(x2, 0.0f), w0); r[3] = fminf(fmaxf(y2, 0.0f), h0);
    }
}

/* ------------------------------------------------------------------------------------------------
 * E10/E11  bilinear resampling shared by letterbox and ReID crop (cv2.INTER_LINEAR geometry:
 * half-pixel centres, edge clamp; result rounded to uint8 as cv2 does before /255) (D-14).
 * ---------------------------------------------------------------------------------------------- */
static inline void so_axis(int d, float scale, int n_src, int* i0, int* i1, float* frac)
{
    float t = (float)d + 0.5f;
    float s = t * scale;
    float f = s - 0.5f;
    int i = (int)floorf(f);
    float fr = f - (float)i;
    if (i < 0) { i = 0; fr = 0.0f; }
    if (i >= n_src - 1) { i = n_src - 1; fr = 0.0f; *i1 = i; } else *i1 = i + 1;
    *i0 = i; *frac = fr;
}

static inline float so_bilerp_u8(float p00, float p01, float p10, float p11, float fx, float fy)
{
    float a = fmaf(fx, p01 - p00, p00);
    float b = fmaf(fx, p11 - p10, p10);
    float v = fmaf(fy, b - a, a);
    float q = floorf(v + 0.5f);
    return fminf(fmaxf(q, 0.0f), 255.0f);
}

/* letterbox: BGR u8 [H,W,3] (row stride in bytes) -> RGB f32 CHW [3,out_h,out_w] / 255, pad 114.
 * resized region new_w x new_h placed at (pad_left, pad_top). */
void so_letterbox(const uint8_t* src, int H, int W, int stride, float* dst, int out_h, int out_w,
                  int new_h, int new_w, int pad_top, int pad_left, int pad_value)
{
    float sx = (float)W / (float)new_w, sy = (float)H / (float)new_h;
    float padv = (float)pad_value / 255.0f;
    size_t plane = (size_t)out_h * out_w;
    for (int y = 0; y < out_h; ++y)
        for (int x = 0; x < out_w; ++x) {
            int ry = y - pad_top, rx = x - pad_left;
            size_t o = (size_t)y * out_w + x;
            if (ry < 0 || ry >= new_h || rx < 0 || rx >= new_w) {
                dst[o] = padv; dst[plane + o] = padv; dst[2 * plane + o] = padv;
                continue;
            }
            int y0, y1, x0, x1; float fy, fx;
            so_axis(ry, sy, H, &y0, &y1, &fy);
            so_axis(rx, sx, W, &x0, &x1, &fx);
            for (int c = 0; c < 3; ++c) {           /* c = output channel (RGB) <- BGR 2-c */
                int sc = 2 - c;
                float p00 = src[(size_t)y0 * stride + x0 * 3 + sc], p01 = src[(size_t)y0 * stride + x1 * 3 + sc];
                float p10 = src[(size_t)y1 * stride + x0 * 3 + sc], p11 = src[(size_t)y1 * stride + x1 * 3 + sc];
                dst[c * plane + o] = so_bilerp_u8(p00, p01, p10, p11, fx, fy) / 255.0f;
            }
        }
}

/* ReID crop: for each det (xyxy f32, original pixels) clip to ints, bilinear to out_h x out_w,
 * BGR->RGB, /255, (x-mean)/std, CHW f32 (StrongSORT _get_features + ReID preprocess, D-15). */
void so_crop_norm(const uint8_t* src, int H, int W, int stride, const float* dets, int det_stride,
                  int D, float* dst, int out_h, int out_w)
{
    const float mean[3] = { 0.485f, 0.456f, 0.406f }, sd[3] = { 0.229f, 0.224f, 0.225f };
    size_t plane = (size_t)out_h * out_w;
    for (int d = 0; d < D; ++d) {
        const float* b = dets + (size_t)d * det_stride;
        int x1 = (int)b[0], y1 = (int)b[1], x2 = (int)b[2], y2 = (int)b[3];
        if (x1 < 0) x1 = 0; if (y1 < 0) y1 = 0;
        if (x2 > W - 1) x2 = W - 1; if (y2 > H - 1) y2 = H - 1;
        if (x1 > W - 1) x1 = W - 1; if (y1 > H - 1) y1 = H - 1;
        int cw = x2 - x1, ch = y2 - y1;
        if (cw < 1) cw = 1; if (ch < 1) ch = 1;
        float sx = (float)cw / (float)out_w, sy = (float)ch / (float)out_h;
        float* o = dst + (size_t)d * 3 * plane;
        for (int y = 0; y < out_h; ++y) {
            int yy0, yy1; float fy;
            so_axis(y, sy, ch, &yy0, &yy1, &fy);
            for (int x = 0; x < out_w; ++x) {
                int xx0, xx1; float fx;
                so_axis(x, sx, cw, &xx0, &xx1, &fx);
                for (int c = 0; c < 3; ++c) {
                    int sc = 2 - c;
                    const uint8_t* r0 = src + (size_t)(y1 + yy0) * stride, *r1 = src + (size_t)(y1 + yy1) * stride;
                    float p00 = r0[(x1 + xx0) * 3 + sc], p01 = r0[(x1 + xx1) * 3 + sc];
                    float p10 = r1[(x1 + xx0) * 3 + sc], p11 = r1[(x1 + xx1) * 3 + sc];
                    float q = so_bilerp_u8(p00, p01, p10, p11, fx, fy) / 255.0f;
                    o[c * plane + (size_t)y * out_w + x] = (q - mean[c]) / sd[c];
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * N4  camera-motion compensation: ECC alignment of consecutive down-scaled grey frames + the warp
 * applied to every track's box before the Kalman prediction.  NOT in the reference snapshot; stands where upstream
 * StrongSORT calls `tracker.camera_update(prev_img, cur_img)` ahead of `tracker.predict()` inside model.track
 * (yolo_multi_model.py:41).  PARITY UNPINNED twice over: the upstream code is recalled, not present (SURVEY App. A.1:
 * cv2.findTransformECC, MOTION_EUCLIDEAN, 100 iterations, eps 1e-5, scale 0.1), and OpenCV itself is not installed, so
 * this is the published forward-additive ECC iteration (Evangelidis & Psarakis 2008) written out with a DEFINED
 * operation order, which the HIP kernel reproduces bit for bit (D-18).  All arithmetic float64.
 * ---------------------------------------------------------------------------------------------- */
#define SO_ECC_THREADS 1024            /* the reduction shape both sides use: 1024 strided partial sums, 64-wide
                                          xor butterflies, then 16 group sums added left to right */
#define SO_ECC_NS 15

/* BGR u8 -> grey (OpenCV's fixed-point weights) -> bilinear down-scale to hs x ws (half-pixel centres, rounded to u8) */
static inline float so_grey(const uint8_t* p) { return (float)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14); }

void so_gray_small(const uint8_t* src, int H, int W, int stride, uint8_t* dst, int hs, int ws)
{
    const float sx = (float)W / (float)ws, sy = (float)H / (float)hs;
    for (int y = 0; y < hs; ++y) {
        int y0, y1; float fy;
        so_axis(y, sy, H, &y0, &y1, &fy);
        for (int x = 0; x < ws; ++x) {
            int x0, x1; float fx;
            so_axis(x, sx, W, &x0, &x1, &fx);
            const uint8_t *r0 = src + (size_t)y0 * stride, *r1 = src + (size_t)y1 * stride;
            dst[y * ws + x] = (uint8_t)so_bilerp_u8(so_grey(r0 + x0 * 3), so_grey(r0 + x1 * 3), so_grey(r1 + x0 * 3), so_grey(r1 + x1 * 3), fx, fy);
        }
    }
}

/* sin / cos by their Taylor polynomials in Horner form with fma (the same expression on both sides; |theta| small) */
void so_sincos(double t, double* s, double* c)
{
    const double t2 = t * t;
    double ps = -1.0 / 1307674368000.0;                       /* -1/15! */
    ps = fma(ps, t2, 1.0 / 6227020800.0);                     /* +1/13! */
    ps = fma(ps, t2, -1.0 / 39916800.0);
    ps = fma(ps, t2, 1.0 / 362880.0);
    ps = fma(ps, t2, -1.0 / 5040.0);
    ps = fma(ps, t2, 1.0 / 120.0);
    ps = fma(ps, t2, -1.0 / 6.0);
    ps = fma(ps, t2, 1.0);
    double pc = 1.0 / 20922789888000.0;                       /* +1/16! */
    pc = fma(pc, t2, -1.0 / 87178291200.0);
    pc = fma(pc, t2, 1.0 / 479001600.0);
    pc = fma(pc, t2, -1.0 / 3628800.0);
    pc = fma(pc, t2, 1.0 / 40320.0);
    pc = fma(pc, t2, -1.0 / 720.0);
    pc = fma(pc, t2, 1.0 / 24.0);
    pc = fma(pc, t2, -0.5);
    pc = fma(pc, t2, 1.0);
    *s = t * ps; *c = pc;
}

static inline double so_px(const uint8_t* I, int ws, int hs, int x, int y)
{
    if (x < 0) x = 0; if (x > ws - 1) x = ws - 1;
    if (y < 0) y = 0; if (y > hs - 1) y = hs - 1;
    return (double)I[y * ws + x];
}

/* warped image value and gradients at template pixel (x, y); returns 0 when the warped point leaves the image */
static inline int so_ecc_sample(const uint8_t* I, int ws, int hs, double c, double s, double tx, double ty, int x, int y,
                                double* iw, double* gx, double* gy)
{
    const double xw = (c * (double)x - s * (double)y) + tx, yw = (s * (double)x + c * (double)y) + ty;
    if (!(xw >= 0.0 && xw <= (double)(ws - 1) && yw >= 0.0 && yw <= (double)(hs - 1))) return 0;
    int x0 = (int)xw, y0 = (int)yw;
    if (x0 > ws - 2) x0 = ws - 2; if (y0 > hs - 2) y0 = hs - 2;
    if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0;
    const double fx = xw - (double)x0, fy = yw - (double)y0;
    double v[3][4];
    for (int k = 0; k < 4; ++k) {
        const int xi = x0 + (k & 1), yi = y0 + (k >> 1);
        v[0][k] = so_px(I, ws, hs, xi, yi);
        v[1][k] = (so_px(I, ws, hs, xi + 1, yi) - so_px(I, ws, hs, xi - 1, yi)) * 0.5;
        v[2][k] = (so_px(I, ws, hs, xi, yi + 1) - so_px(I, ws, hs, xi, yi - 1)) * 0.5;
    }
    double o[3];
    for (int q = 0; q < 3; ++q) {
        const double a = v[q][0] + fx * (v[q][1] - v[q][0]);
        const double b = v[q][2] + fx * (v[q][3] - v[q][2]);
        o[q] = a + fy * (b - a);
    }
    *iw = o[0]; *gx = o[1]; *gy = o[2];
    return 1;
}

/* the reduction tree of the kernel: part[t][k] for t < 1024 -> 16 groups of 64 reduced by xor butterflies -> summed 0..15 */
static void so_ecc_reduce(double (*part)[SO_ECC_NS], int ns, double* out)
{
    for (int k = 0; k < ns; ++k) {
        double tot = 0.0;
        for (int g = 0; g < SO_ECC_THREADS / 64; ++g) {
            double p[64], q[64];
            for (int l = 0; l < 64; ++l) p[l] = part[g * 64 + l][k];
            for (int off = 32; off >= 1; off >>= 1) {
                for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
                memcpy(p, q, sizeof p);
            }
            tot = (g == 0) ? p[0] : tot + p[0];
        }
        out[k] = tot;
    }
}

/* T (previous frame) and I (current frame): grey u8 [hs][ws].  warp[6] = 2x3 matrix mapping T coordinates to I
 * coordinates, translation in small-image pixels.  Returns the iterations run (>= 1), or -1 (no usable alignment:
 * the caller keeps the identity). */
int so_ecc(const uint8_t* T, const uint8_t* I, int hs, int ws, int max_iter, double eps, double* warp)
{
    static double part[SO_ECC_THREADS][SO_ECC_NS];
    double theta = 0.0, tx = 0.0, ty = 0.0, last_rho = -2.0;
    const int npx = hs * ws;
    int it;
    warp[0] = 1; warp[1] = 0; warp[2] = 0; warp[3] = 0; warp[4] = 1; warp[5] = 0;
    for (it = 1; it <= max_iter; ++it) {
        double s, c, sum[SO_ECC_NS];
        so_sincos(theta, &s, &c);
        /* pass 1: valid count, sum of warped image, sum of template */
        for (int t = 0; t < SO_ECC_THREADS; ++t) {
            double a0 = 0, a1 = 0, a2 = 0;
            for (int p = t; p < npx; p += SO_ECC_THREADS) {
                const int y = p / ws, x = p - y * ws;
                double iw, gx, gy;
                if (!so_ecc_sample(I, ws, hs, c, s, tx, ty, x, y, &iw, &gx, &gy)) continue;
                a0 = a0 + 1.0; a1 = a1 + iw; a2 = a2 + (double)T[p];
            }
            part[t][0] = a0; part[t][1] = a1; part[t][2] = a2;
        }
        so_ecc_reduce(part, 3, sum);
        if (!(sum[0] >= 64.0)) return -1;
        const double mI = sum[1] / sum[0], mT = sum[2] / sum[0];
        /* pass 2: Hessian, projections, correlation, norms */
        for (int t = 0; t < SO_ECC_THREADS; ++t) {
            double a[SO_ECC_NS];
            for (int k = 0; k < SO_ECC_NS; ++k) a[k] = 0.0;
            for (int p = t; p < npx; p += SO_ECC_THREADS) {
                const int y = p / ws, x = p - y * ws;
                double iwv, gx, gy;
                if (!so_ecc_sample(I, ws, hs, c, s, tx, ty, x, y, &iwv, &gx, &gy)) continue;
                const double iw = iwv - mI, tz = (double)T[p] - mT;
                const double hx = -((double)x * s) - (double)y * c, hy = (double)x * c - (double)y * s;
                const double j0 = gx * hx + gy * hy, j1 = gx, j2 = gy;
                a[0] = a[0] + j0 * j0; a[1] = a[1] + j0 * j1; a[2] = a[2] + j0 * j2;
                a[3] = a[3] + j1 * j1; a[4] = a[4] + j1 * j2; a[5] = a[5] + j2 * j2;
                a[6] = a[6] + j0 * iw; a[7] = a[7] + j1 * iw; a[8] = a[8] + j2 * iw;
                a[9] = a[9] + j0 * tz; a[10] = a[10] + j1 * tz; a[11] = a[11] + j2 * tz;
                a[12] = a[12] + tz * iw; a[13] = a[13] + iw * iw; a[14] = a[14] + tz * tz;
            }
            for (int k = 0; k < SO_ECC_NS; ++k) part[t][k] = a[k];
        }
        so_ecc_reduce(part, SO_ECC_NS, sum);
        const double h00 = sum[0], h01 = sum[1], h02 = sum[2], h11 = sum[3], h12 = sum[4], h22 = sum[5];
        const double c00 = h11 * h22 - h12 * h12, c01 = h02 * h12 - h01 * h22, c02 = h01 * h12 - h02 * h11;
        const double c11 = h00 * h22 - h02 * h02, c12 = h01 * h02 - h00 * h12, c22 = h00 * h11 - h01 * h01;
        const double det = (h00 * c00 + h01 * c01) + h02 * c02;
        const double ni = sum[13], nt = sum[14], corr = sum[12];
        if (!(det != 0.0) || !(ni > 0.0) || !(nt > 0.0)) return -1;
        const double rho = corr / (sqrt(ni) * sqrt(nt));
        if (!(rho == rho)) return -1;
        if (it > 1 && fabs(rho - last_rho) < eps) break;
        last_rho = rho;
        const double i00 = c00 / det, i01 = c01 / det, i02 = c02 / det, i11 = c11 / det, i12 = c12 / det, i22 = c22 / det;
        const double ip0 = sum[6], ip1 = sum[7], ip2 = sum[8], tp0 = sum[9], tp1 = sum[10], tp2 = sum[11];
        const double q0 = (i00 * ip0 + i01 * ip1) + i02 * ip2, q1 = (i01 * ip0 + i11 * ip1) + i12 * ip2, q2 = (i02 * ip0 + i12 * ip1) + i22 * ip2;
        const double lam_n = ni - ((ip0 * q0 + ip1 * q1) + ip2 * q2), lam_d = corr - ((tp0 * q0 + tp1 * q1) + tp2 * q2);
        if (!(lam_d > 0.0)) return -1;
        const double lam = lam_n / lam_d;
        const double e0 = lam * tp0 - ip0, e1 = lam * tp1 - ip1, e2 = lam * tp2 - ip2;
        theta = theta + ((i00 * e0 + i01 * e1) + i02 * e2);
        tx = tx + ((i01 * e0 + i11 * e1) + i12 * e2);
        ty = ty + ((i02 * e0 + i12 * e1) + i22 * e2);
    }
    double s, c;
    so_sincos(theta, &s, &c);
    warp[0] = c; warp[1] = -s; warp[2] = tx; warp[3] = s; warp[4] = c; warp[5] = ty;
    return it > max_iter ? max_iter : it;
}

/* the warp (translation already in full-frame pixels) applied to one track's box; mean[8] in place (D-18) */
void so_camera_update(double* mean, const double* m)
{
    const double w = mean[2] * mean[3], h = mean[3];
    const double x1 = mean[0] - w / 2, y1 = mean[1] - h / 2, x2 = x1 + w, y2 = y1 + h;
    const double ax = (m[0] * x1 + m[1] * y1) + m[2], ay = (m[3] * x1 + m[4] * y1) + m[5];
    const double bx = (m[0] * x2 + m[1] * y2) + m[2], by = (m[3] * x2 + m[4] * y2) + m[5];
    const double nw = bx - ax, nh = by - ay;
    mean[0] = ax + nw / 2; mean[1] = ay + nh / 2; mean[2] = nw / nh; mean[3] = nh;
}
