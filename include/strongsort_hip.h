/*
 * strongsort_hip.h — C ABI of the MI355X (gfx950) StrongSORT hot path.
 *
 * The reference has no plugin / FFI interface: its only crossings into the
 * hot path are two Python calls into the third-party `ultralytics` package,
 *     /root/reference/yolo_multi_model.py:41    results = model.track(image, ..., persist=True, tracker=...)
 *     /root/reference/yolo_multi_model.py:173   results = model.predict(image, ...)
 * configured by /root/reference/yolo_multi_model.py:18-21 (conf, iou, agnostic_nms, max_det).
 * Every entry point below replaces one stage that runs inside those two calls; the Python host
 * side (strongsort_yolo_amd/yolo.py `YOLO.track/.predict`, strongsort_yolo_amd/tracker.py
 * `StrongSORT.update(dets, frame)`) binds them with ctypes — see INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes; `d_` = device memory owned by the caller; every call is
 * asynchronous on the context's HIP stream unless it says "synchronous"; return 0 or a negative
 * SS_ERR_* code, never a C++ exception; one context is not thread safe.
 */
#ifndef STRONGSORT_HIP_H
#define STRONGSORT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_OK 0
#define SS_ERR_INVALID (-1)      /* bad argument */
#define SS_ERR_CAPACITY (-2)     /* more tracks / detections / candidates than the context holds */
#define SS_ERR_HIP (-3)          /* HIP runtime error (see ss_last_error) */
#define SS_ERR_INFEASIBLE (-4)   /* assignment problem has no finite solution */

#define SS_FEAT_DIM 512
#define SS_MAX_TRACKS 256        /* track slots per stream */
#define SS_MAX_DETS 128          /* detections per stream per frame */
#define SS_OUT_COLS 8            /* x1,y1,x2,y2,track_id,class_id,conf,det_idx */

typedef struct ss_ctx ss_ctx;

/* Tracker constants (oracle/DECISIONS.md D-02..D-11).  Replaces the tracker YAML that
 * yolo_multi_model.py:41 names (`tracker="botsort.yaml"`). */
typedef struct ss_config {
    double max_dist;             /* 0.2    cosine matching threshold            */
    double max_iou_distance;     /* 0.7    IoU-stage threshold on 1-IoU          */
    double mc_lambda;            /* 0.995  appearance / motion blend            */
    double gating_threshold;     /* 9.4877 chi2inv95[4]                          */
    double gated_cost;           /* 1e5                                          */
    double std_weight_position;  /* 1/20                                         */
    double std_weight_velocity;  /* 1/160                                        */
    double ema_alpha;            /* 0.9  (a=(float)alpha, 1-a=(float)(1.0-alpha))  */
    int    max_age;              /* 30                                           */
    int    n_init;               /* 3                                            */
    int    nn_budget;            /* 100 (<= 128)                                 */
    int    n_streams;            /* independent video streams held by this context */
    int    debug;                /* !=0: keep stage intermediates for ss_get_debug */
} ss_config;

/* ---- context ------------------------------------------------------------------------------- */
int  ss_create(const ss_config* cfg, int device, ss_ctx** out);
void ss_destroy(ss_ctx* ctx);
const char* ss_last_error(const ss_ctx* ctx);            /* NULL ctx: last global error */
int  ss_set_hip_stream(ss_ctx* ctx, void* hip_stream);   /* e.g. torch.cuda.current_stream().cuda_stream */
int  ss_reset(ss_ctx* ctx, int stream);                  /* stream < 0: all streams */
int  ss_synchronize(ss_ctx* ctx);

/* Frame source side of model.track(image) (yolo_multi_model.py:272 cap.read() -> :41): copy a host frame (any pageable
 * memory, e.g. the array cv2 returned) into device memory through the context's write-combined pinned staging ring,
 * asynchronously on `hip_stream`.  The host buffer may be reused as soon as the call returns. */
int ss_upload(ss_ctx* ctx, void* hip_stream, void* d_dst, const void* h_src, size_t bytes);
/* A frame GROUP at once (the throughput form of ss_upload, for the reference's `cap.read()` loop at yolo_multi_model.py:270-278 fed
 * `batch` frames at a time): n host buffers of bytes_each bytes are staged by `threads` host threads in one write-combined area and
 * leave in ONE asynchronous copy to d_dst, where the frames lie contiguously; h_srcs may be reused on return. */
int ss_upload_batch(ss_ctx* ctx, void* hip_stream, void* d_dst, const void* const* h_srcs, int n, size_t bytes_each, int threads);

/* ... and back: device -> host through a pinned staging buffer; synchronous. */
int ss_download(ss_ctx* ctx, void* hip_stream, void* h_dst, const void* d_src, size_t bytes);

/* ---- N2  annotation overlay on device frames (the cv2 drawing of yolo_multi_model.py:58-162, :311-331) --------------
 * d_frames: `batch` BGR u8 frames [h][w][3] (row_stride bytes per row, frame b at + b*frame_batch_stride bytes), drawn in
 * place.  d_prims: primitives of all frames, 8 ints each {type, x0, y0, x1, y1, color B|G<<8|R<<16, a, b}; frame b owns
 * d_prim_off[b] .. d_prim_off[b+1] and they are painted in list order:
 *   type 0 rectangle outline, thickness a centred on the edge      type 1 filled rectangle
 *   type 2 filled circle, centre (x0,y0), radius a                 type 3 line, thickness a
 *   type 4 5x7 raster text, baseline-left (x0,y0), x1 characters at d_chars + a, scale b>>1
 *   b bit 0: member of a blended group (composited opaquely inside the group, then 0.7 : 0.3 over the frame).
 * All coverage tests are integer arithmetic (bit-exact against the NumPy rasteriser in oracle/overlay_np.py).
 * ss_overlay_set_font uploads the 95 x 5 column-bitmap font (strongsort_yolo_amd/overlay_font.py) once per context. */
int ss_overlay_set_font(ss_ctx* ctx, const uint8_t* h_font_95x5);
int ss_overlay(ss_ctx* ctx, void* hip_stream, uint8_t* d_frames, int batch, long long frame_batch_stride, int h, int w,
               int row_stride, const int* d_prims, const int* d_prim_off, const uint8_t* d_chars);

/* ---- a1  letterbox / preprocess  (inside model.track/.predict, yolo_multi_model.py:41,:173) ---
 * BGR u8 [h][w][3] (row_stride bytes) -> RGB [3][out_h][out_w], /255, pad 114.
 * dst_flags: bit 0 (SS_DST_F16) writes IEEE half, else float; bit 1 (SS_DST_HWC) writes
 * channels-last [out_h][out_w][3] (what the NHWC convolutions read) instead of three planes. */
#define SS_DST_F16 1
#define SS_DST_HWC 2
#define SS_DST_U8 4   /* ss_crop_norm* with SS_DST_HWC only: uint8 output [n][256][128][3] = the rounded bilinear value per RGB channel BEFORE
                         /255 and mean / std (256 x 3 possible results: the consumer applies them — ss_op32_stem_u8); a quarter of the float bytes */
int ss_letterbox(ss_ctx* ctx, const uint8_t* d_src, int h, int w, int row_stride, void* d_dst,
                 int dst_flags, int out_h, int out_w, int new_h, int new_w, int pad_top, int pad_left,
                 int pad_value);
/* Same for `batch` equally sized images in one launch (the streams of a rank, or consecutive frames of a
 * stream): image b at d_src + b*src_batch_stride bytes, output b at d_dst + b*3*out_h*out_w elements. */
int ss_letterbox_batch(ss_ctx* ctx, const uint8_t* d_src, int batch, long long src_batch_stride, int h,
                       int w, int row_stride, void* d_dst, int dst_flags, int out_h, int out_w, int new_h,
                       int new_w, int pad_top, int pad_left, int pad_value);

/* ---- a3  NMS  (inside model.track/.predict; thresholds = yolo_multi_model.py:18-21) -----------
 * d_pred: YOLOv8 head layout [(4+nc+n_extra)][n_anchors] float (xywh, class scores, extra rows).
 * Writes up to max_det rows [x1,y1,x2,y2,conf,cls, extra...] in ORIGINAL-image pixels
 * (scale_boxes with gain/pad, clipped to w0 x h0), the kept anchor indices and the count. */
int ss_nms(ss_ctx* ctx, const float* d_pred, int n_anchors, int nc, int n_extra, float conf_thres,
           float iou_thres, int agnostic, float max_wh, int max_det, float gain, float pad_x,
           float pad_y, float w0, float h0, float* d_rows, int row_stride, int* d_keep, int* d_count);
/* `batch` images in one set of launches.  Image b: predictions at d_pred + b*pred_batch_stride floats,
 * geometry d_geom[b] = {gain, pad_x, pad_y, w0, h0} (device floats, so images of different sizes can share
 * a batch), rows at d_rows + b*rows_batch_stride floats, anchors at d_keep + b*keep_batch_stride, count
 * d_count[b].  The first call with a larger batch than any before grows the context's workspace and must
 * not be inside a HIP-graph capture.  More than 8192 candidates above conf_thres in an image leave that
 * image's count 0 and raise SS_ERR_CAPACITY at the next ss_check_errors (same for ss_nms). */
int ss_nms_batch(ss_ctx* ctx, const float* d_pred, int batch, long long pred_batch_stride, int n_anchors,
                 int nc, int n_extra, float conf_thres, float iou_thres, int agnostic, float max_wh,
                 int max_det, const float* d_geom, float* d_rows, int row_stride,
                 long long rows_batch_stride, int* d_keep, long long keep_batch_stride, int* d_count);

/* The `classes` override of the reference (yolo_multi_model.py:22, commented out there): keep only anchors whose best
 * class is in classes[0..n) (ids 0..127); n = 0 keeps every class (default).  Host state of the context, read by the
 * following ss_nms / ss_nms_batch calls (and baked into a HIP graph that captures them). */
int ss_nms_set_classes(ss_ctx* ctx, const int* classes, int n);

/* ---- a4  ReID crop-extract  (StrongSORT._get_features, inside model.track) ---------------------
 * For each detection row (x1,y1,x2,y2,... ; det_stride floats) crop + bilinear to 256x128,
 * /255, ImageNet mean/std, RGB [n][3][256][128] (out_flags as ss_letterbox's dst_flags: SS_DST_F16,
 * SS_DST_HWC for [n][256][128][3]).  Rows >= *d_count are left untouched when d_count != NULL. */
int ss_crop_norm(ss_ctx* ctx, const uint8_t* d_frame, int h, int w, int row_stride,
                 const float* d_dets, int det_stride, int n, const int* d_count, void* d_out,
                 int out_flags);
/* `batch` frames in one launch: frame b at d_frames + b*frame_batch_stride bytes, its detections at
 * d_dets + b*dets_batch_stride floats, its count d_counts[b] (NULL: n each), output [batch][n][3][256][128]. */
int ss_crop_norm_batch(ss_ctx* ctx, const uint8_t* d_frames, int batch, long long frame_batch_stride,
                       int h, int w, int row_stride, const float* d_dets, int det_stride,
                       long long dets_batch_stride, int n, const int* d_counts, void* d_out, int out_flags);
/* Packed form of ss_crop_norm_batch (channels-last output only): d_off [batch + 1] receives the exclusive prefix of
 * min(count, n) (d_off[batch] = number of crops) and crop d of image i lands at slot d_off[i] + d, so the valid crops of
 * a frame group are contiguous; ss_op_set_valid_images(stream, d_off + batch, batch * n) then lets the ReID network skip the rest.
 * ss_unpack_feats: d_feats[i][d][0..512) (float, image stride in floats) = d_emb[d_off[i] + d] for d < min(count, n). */
int ss_crop_norm_packed(ss_ctx* ctx, const uint8_t* d_frames, int batch, long long frame_batch_stride, int h, int w, int stride,
                        const float* d_dets, int det_stride, long long dets_batch_stride, int n, const int* d_counts,
                        int* d_off, void* d_out, int out_flags);
int ss_unpack_feats(ss_ctx* ctx, const void* d_emb, int emb_half, const int* d_off, const int* d_counts, int batch, int n,
                    float* d_feats, long long feats_image_stride);

/* One frame's results gathered into ONE buffer on the context's stream (the per-frame call of yolo_multi_model.py:41 / :278 reads boxes and
 * track rows on the host after every frame): dst[0] = n = min(*d_n_dets, det_cap) and dst[1] = m = min(*d_n_out, out_cap) as int32 bit
 * patterns, n rows of det_ld floats from dst + 2, m rows of out_ld floats from dst + 2 + det_cap * det_ld.  dst: 2 + det_cap * det_ld +
 * out_cap * out_ld floats of device memory or of pinned (device-accessible) host memory - then the host reads it after synchronising the
 * stream, with no copy in between.  d_n_out / d_out may both be NULL (detection only: m = 0). */
int ss_pack_results(ss_ctx* ctx, const int* d_n_dets, const float* d_dets, int det_ld, int det_cap, const int* d_n_out, const float* d_out,
                    int out_ld, int out_cap, float* dst);

/* Largest n_frames ss_track_update_group / ss_cmc_estimate accept (compile-time SS_FMAX). */
int ss_max_group_frames(void);
/* hip_event (a hipEvent_t of the caller, NULL: none) is recorded on the tracker's stream right after the association launch
 * of every following ss_track_update_group call: a pipeline can hold its other stream back for those ~40 us — the
 * association kernel needs whole CUs (8 waves x 127 VGPRs + 64 KiB LDS per workgroup) and is starved by co-running network
 * kernels otherwise — and let it run beside the per-frame chain that follows. */
int ss_track_set_assoc_event(ss_ctx* ctx, void* hip_event);
/* ---- a6..a10  tracker update  (tracker.update inside model.track, yolo_multi_model.py:41) -----
 * A GROUP of n_frames (1..ss_max_group_frames() = 32) consecutive frames for EVERY stream of the context in one batch of launches; the
 * frames are associated strictly in order (frame f sees the tracks, galleries and ids left by frame f-1), so the
 * rows are identical to n_frames single-frame calls — what the group buys is that the galleries are read from HBM
 * once per group instead of once per frame (the association kernel handles the detections of all frames at once).
 *   d_dets   [n_frames][n_streams][SS_MAX_DETS][6]   x1,y1,x2,y2,conf,cls (original pixels, float)
 *   d_ndets  [n_frames][n_streams]                   detections per frame and stream (device ints)
 *   d_feats  [n_frames][n_streams][SS_MAX_DETS][512] raw ReID embeddings (normalised on device)
 *   d_img_hw [n_streams][2]                          frame height, width (device ints; output clipping)
 * Results stay on the device: d_out [n_frames][n_streams][SS_MAX_TRACKS][8], d_nout [n_frames][n_streams].
 * Device-side capacity errors: detections of a frame that would need more than SS_MAX_TRACKS slots start no track (the
 * frame is tracked as if they had not been there); the stream's tables stay consistent, SS_ERR_CAPACITY is reported by the
 * next ss_check_errors and stays set until ss_reset. */
int ss_track_update_group(ss_ctx* ctx, int n_frames, const float* d_dets, const int* d_ndets, const float* d_feats,
                          const int* d_img_hw, float* d_out, int* d_nout);
/* One frame: ss_track_update_group with n_frames = 1. */
int ss_track_update(ss_ctx* ctx, const float* d_dets, const int* d_ndets, const float* d_feats,
                    const int* d_img_hw, float* d_out, int* d_nout);

/* ---- N4  camera-motion compensation (upstream StrongSORT: tracker.camera_update(prev, cur) before tracker.predict();
 * not in the reference snapshot; optional) ----------------------------------------------------------
 * ss_cmc_estimate: for the n_frames x n_streams BGR u8 frames at d_frames ([F][S] order, frame_stride bytes apart) build
 * the 0.1x grey images and align each with its predecessor (the stream's last frame of the previous call for f = 0) by
 * ECC, euclidean warp, <= 100 iterations: d_warps[f][s][8] = 2x3 matrix previous -> current in full-frame pixels,
 * [6] = iterations (>= 1) or -1 (no usable alignment / no predecessor: identity), [7] = 0.  Asynchronous on hip_stream;
 * stateless apart from the remembered last frames, so it can run beside the detector.  d_n_valid (device int, may be NULL =
 * n_frames): only the first *d_n_valid frames of the buffer are real (a partial last group behind a captured graph): the
 * others get -1 warps and the last REAL frame is what the next call aligns its first frame with.
 * ss_track_set_cmc: the following tracker calls move every track's box by warp [f][s] before predicting frame f
 * (NULL switches compensation off, the default). */
int ss_cmc_estimate(ss_ctx* ctx, void* hip_stream, const uint8_t* d_frames, int n_frames, long long frame_stride, int h, int w,
                    int row_stride, const int* d_n_valid, double* d_warps);
int ss_track_set_cmc(ss_ctx* ctx, const double* d_warps);

/* Synchronous convenience for one stream with host buffers (used by StrongSORT.update). */
int ss_track_update_host(ss_ctx* ctx, int stream, const float* h_dets, int n, const float* h_feats,
                         int img_h, int img_w, float* h_out, int cap_rows, int* n_out);

/* Tuning switches of a context (host state, read at the next tracker call):
 *   "cos_grid"         persistent workgroups of the association kernel, a multiple of 8 (default 512 = two per CU)
 *   "assoc_comp_rows"  a gallery's last 16-row tile with at most this many rows (0..12, default 12) is cut into 4-row groups
 *                      and four groups of any tracks form one composite tile of the association kernel (0: off)
 *   "assoc_stage"      how the association kernel brings a work record's detection operand (2 x 32 KiB) into the LDS:
 *                      0 through registers behind one barrier; 1 / 2 / 4 by LDS-DMA in that many pieces, each awaited
 *                      right before the first k-segment that reads it; 5 = 4 with only the first piece requested before
 *                      the first barrier (default).  Same bits in every form.
 *   "track_graph"      1 (default): the per-frame chain of a group of >= 2 frames (k_frame / k_post / k_newrow, 3 F - 1 dependent
 *                      launches) is replayed as one captured HIP graph per (frames, caller buffers, options) combination — one
 *                      hipGraphLaunch of host time instead of up to 95 launches; 0: plain launches
 *   "assoc_xcd_map"    0 (default): a gallery range lives on one XCD (every XCD stages all detection operands);
 *                      1: a detection column-tile pair lives on one XCD (every XCD streams the whole gallery)
 *   "frame_caps"       0 (default) or a multiple of 16 in 16..128: the per-frame kernel's LDS work areas are sized for that many
 *                      tracks / detections (larger frames use a global scratch area)
 *   "chain_cus"        0 (default): the per-frame chain runs on the context's stream after the association launch; n (a multiple
 *                      of 8, <= 128): DETACHED — the chain goes to a stream of the library whose queue owns the first n compute
 *                      units of the device's CU mask (n / 8 of every XCD), ordered after the association launch; the call returns
 *                      with the caller's stream free to run its next launches.  Whoever reads d_out / d_nout (or any tracker state)
 *                      afterwards first calls ss_track_join on the stream it reads on; the library's own readers and the next
 *                      ss_track_update_group do so themselves.  The reservation is only real when every other stream of the
 *                      application is created with ss_stream_create(ctx, n, ..), i.e. without those compute units.  -1: detached onto
 *                      a plain high-priority stream (every compute unit, nothing reserved). */
int ss_set_option(ss_ctx* ctx, const char* name, int value);
/* `hip_stream` waits for the detached chain of the last ss_track_update_group (no-op without one). */
int ss_track_join(ss_ctx* ctx, void* hip_stream);
/* A non-blocking HIP stream whose queue may use every compute unit except the first `skip_cus` of the CU mask (bit i = compute
 * unit i / 8 of XCD i % 8; skip_cus == 0: a plain stream) -> *out (hipStream_t); ss_stream_destroy releases it. */
int ss_stream_create(ss_ctx* ctx, int skip_cus, void** out);
int ss_stream_destroy(ss_ctx* ctx, void* hip_stream);

/* Per-stream error flags raised on the device (capacity, infeasible); synchronous. */
int ss_check_errors(ss_ctx* ctx);

/* ---- stage entry points for known-answer tests (device pointers, natural layouts) ------------- */
int ss_feat_normalize(ss_ctx* ctx, const float* d_raw, int n, float* d_unit);
int ss_ema(ss_ctx* ctx, const float* d_smooth, const float* d_feat, int n, float* d_out);
int ss_kf_predict(ss_ctx* ctx, double* d_mean, double* d_cov, int n);
int ss_kf_update(ss_ctx* ctx, double* d_mean, double* d_cov, const double* d_z, const double* d_conf, int n);
/* a7 KalmanFilter.project with the NSA noise: d_zmean[n][4] = H x, d_S[n][4][4] = H P H^T + R(x, conf), R's deviations scaled
 * by (1 - conf); d_conf == NULL: conf = 0 (the form the gating distance uses).  The tracker runs it fused into the gate and
 * the update; this entry point exists for known-answer tests. */
int ss_kf_project(ss_ctx* ctx, const double* d_mean, const double* d_cov, const double* d_conf, int n, double* d_zmean, double* d_S);
int ss_kf_initiate(ss_ctx* ctx, const double* d_z, int n, double* d_mean, double* d_cov);
/* gallery rows natural [T][B][512] -> fragment-major [T][4][16384] used by ss_assoc_cost */
int ss_gallery_pack(ss_ctx* ctx, const float* d_gallery, int T, int B, float* d_frag);
/* fused a7+a8: cosine-gallery-min + Mahalanobis gate + blend + threshold.
 * d_counts [T] valid gallery rows; d_feats [D][512] unit rows; d_mean/d_cov predicted states;
 * d_xyah [D][4].  Outputs [T][D]: cost (f64), cosine (f32), maha (f64), gated (u8). */
int ss_assoc_cost(ss_ctx* ctx, const float* d_gallery_frag, const int* d_counts, int T,
                  const float* d_feats, int D, const double* d_mean, const double* d_cov,
                  const double* d_xyah, double* d_cost, float* d_cos, double* d_maha, uint8_t* d_gated);
int ss_iou_cost(ss_ctx* ctx, const double* d_track_tlwh, int T, const double* d_det_tlwh, int D,
                double* d_cost);
/* a9: rectangular LSAP on a [nr][nc] float64 matrix; d_row_to_col[nr] = column or -1. */
int ss_lsap(ss_ctx* ctx, const double* d_cost, int nr, int nc, int* d_row_to_col);

/* ---- inspection (synchronous; parity tests) ----------------------------------------------------
 * Track table of one stream in track-list order (arrays may be NULL). */
int ss_get_tracks(ss_ctx* ctx, int stream, int cap, int* n_tracks, int* next_id, int* track_id,
                  int* state, int* hits, int* age, int* tsu, int* class_id, float* conf,
                  double* mean, double* cov, float* smooth, int* gal_count);
/* Stage intermediates of frame `frame` of the last group (ctx created with debug != 0).
 * counts[6] = n_conf, n_cand, n_cols_b, n_dets, assignment path of the appearance stage and of the IoU stage (0 stage
 * skipped, 1 unique optimum read off the thresholded matrix, 2 LSAP); matrices are [SS_MAX_TRACKS][SS_MAX_DETS] strided. */
int ss_get_debug(ss_ctx* ctx, int stream, int frame, int* counts, float* cos, double* maha, uint8_t* gated,
                 double* cost_a, double* cost_b, int* lists /* [4][SS_MAX_TRACKS] */);
/* Gallery of one track (by position in the track list) in natural [count][512] order of slots. */
int ss_get_gallery(ss_ctx* ctx, int stream, int track_index, float* rows, int cap_rows, int* count);

/* ---- a2 / a5 glue: stateless fused NHWC-half operators around the PyTorch-ROCm convolutions -------
 * (detector / OSNet forward inside model.track, yolo_multi_model.py:41).  `stream` is a hipStream_t.
 * act: 0 none, 1 relu, 2 silu, 3 sigmoid.  Tensors are channels-last half. */
int ss_op_bias_act_f16(void* stream, void* d_x, const void* d_bias, const void* d_res, long long n_pix, int C, int act);
/* Epilogue with placement (C2f blocks): d_out[pix*out_ld + c] = act(x + bias)[c] (+ res after the activation when
 * res_after), d_out pointing at a channel slice of a wider NHWC tensor (pitch out_ld elements); channels
 * [c0, c0+cn) are also written densely to d_out2 when it is not NULL.  C, out_ld, c0, cn multiples of 8. */
int ss_op_bias_act_place_f16(void* stream, const void* d_x, const void* d_bias, const void* d_res, long long n_pix, int C,
                             int act, int res_after, void* d_out, int out_ld, void* d_out2, int c0, int cn);
/* 1x1 convolution + bias + activation (+ shortcut) in one MFMA kernel: d_out[pix*out_ld + n] =
 * act(sum_k x[pix][k] w[n][k] + bias[n] (+ res[pix][n] before, or after when res_after, the activation)); x [M][K],
 * w [N][K], res [M][N] dense NHWC half; placement arguments as ss_op_bias_act_place_f16 (multiples of 4 here).
 * K % 8 == 0, N % 8 == 0. */
int ss_op_pointwise_f16(void* stream, const void* d_x, const void* d_w, const void* d_bias, const void* d_res, long long M,
                        int K, int N, int act, int res_after, void* d_out, int out_ld, void* d_out2, int c0, int cn);
/* 3x3 / pad 1 / stride 1|2 convolution + bias + activation (+ shortcut) as an implicit GEMM on the same kernel:
 * x [B][H][W][Cin], w [N][3][3][Cin], output [B][OH][OW][N] with OH = (H-1)/stride+1; epilogue and placement
 * arguments as ss_op_pointwise_f16.  Cin % 8 == 0, N % 8 == 0. */
int ss_op_conv3x3_f16(void* stream, const void* d_x, const void* d_w, const void* d_bias, const void* d_res, int B, int H,
                      int W, int Cin, int N, int conv_stride, int act, int res_after, void* d_out, int out_ld,
                      void* d_out2, int c0, int cn);
/* A C2f bottleneck in ONE launch (nets.Bottleneck with 3x3 + 3x3, e = 1.0; reference: ultralytics' Bottleneck as the detector
 * checkpoints the reference loads define it): out = [x +] silu(conv3x3(silu(conv3x3(x, w1) + b1), w2) + b2), the intermediate kept
 * in LDS and rounded to half exactly where the two ss_op_conv3x3_f16 launches round it (the results are bit-identical to them).
 * x dense [B][H][W][C], C in {16, 32, 64, 128}; w1, w2 [C][3][3][C]; d_out points at the block's channel slice of a wider NHWC tensor
 * (out_ld halfs per pixel), d_out2 (or NULL) receives a dense [B][H][W][C] copy. */
int ss_op_bottleneck_f16(void* stream, const void* d_x, const void* d_w1, const void* d_b1, const void* d_w2, const void* d_b2,
                         int B, int H, int W, int C, int add, void* d_out, int out_ld, void* d_out2);
/* Several independent convolutions in ONE launch (the detect head's branches: nets.Detect): every entry is a 3x3 / pad 1
 * (stride 1|2) or 1x1 convolution + bias + activation on NHWC half, dense output [B][OH][OW][N]; all entries of a call have
 * the same ksize; N <= 80, n <= 8.  Weights as ss_op_conv3x3_f16 ([N][3][3][Cin]) / ss_op_pointwise_f16 ([N][Cin]). */
typedef struct ss_conv_desc {
    const void* x; const void* w; const void* bias; void* out;
    int B, H, W, Cin, N, ksize, stride, act;
} ss_conv_desc;
int ss_op_conv_group_f16(void* stream, int n, const ss_conv_desc* descs);
/* One level of the anchor-free (v8) detect head, BOTH branches and all three layers of a branch, in one launch: per branch b
 * 3x3 (Cin -> cm_b) + SiLU -> 3x3 (cm_b -> cm_b) + SiLU -> 1x1 (cm_b -> nout[b]) + bias, cm_0 = 64 (box), cm_1 = 80 (class); the
 * intermediates stay in the LDS.  d_x dense NHWC half [B][H][W][Cin], Cin in {64, 128, 256}; weights as ss_op_conv3x3_f16 /
 * ss_op_pointwise_f16 take them ([cm][3][3][Cin], [cm][3][3][cm], [nout][cm]); d_out[b] dense [B][H][W][nout[b]], nout[b] % 8 == 0,
 * <= cm_b.  Bit-identical to the three separate launches per branch (non-split-K form).  tile16: 8 x 16 tiles for Cin = 64. */
int ss_op_head_f16(void* stream, const void* d_x, const void* const* d_w1, const void* const* d_b1, const void* const* d_w2,
                   const void* const* d_b2, const void* const* d_w3, const void* const* d_b3, void* const* d_out, const int* nout,
                   int B, int H, int W, int Cin, int tile16);
/* YOLOv8 anchor-free head decode: per level l<3 the branch outputs d_box[l] [B][H][W][64] and d_cls[l] [B][H][W][nc]
 * (NHWC half, final 1x1 conv without bias; the biases are added here) -> d_pred [B][4+nc][A] float (xywh in input
 * pixels, class sigmoid), A = sum H[l]*W[l] — the tensor ss_nms reads.  H, W, strides are host int[3]. */
int ss_op_v8_decode_f16(void* stream, const void* const* d_box, const void* const* d_cls, const void* const* d_box_bias,
                        const void* const* d_cls_bias, const int* H, const int* W, const int* strides, int B, int nc,
                        float* d_pred);
/* The same with the head's third branch as extra rows 4 + nc .. 4 + nc + n_ext of the prediction [B][4 + nc + n_ext][A]: d_ext[3]
 * [B][H][W][ext_ld] half with the bias already added (the branch's last 1x1 through ss_op_pointwise_f16); ext_mode 1: keypoint
 * triplets decoded as Ultralytics Pose.kpts_decode ((2 v + cell) * stride for x and y, sigmoid for the visibility), n_ext % 3 == 0;
 * ext_mode 0: raw values (Segment's mask coefficients).  cls_ld >= nc: channels per pixel of the class tensors (a one-class head
 * zero-padded to 8 so that its last 1x1 runs on ss_op_pointwise_f16).  n_ext == 0 and cls_ld == nc: ss_op_v8_decode_f16. */
int ss_op_v8_decode_ext_f16(void* stream, const void* const* d_box, const void* const* d_cls, const void* const* d_box_bias,
                            const void* const* d_cls_bias, const void* const* d_ext, int n_ext, int ext_ld, int ext_mode,
                            const int* H, const int* W, const int* strides, int B, int nc, int cls_ld, float* d_pred);
int ss_op_dwconv3x3_f16(void* stream, const void* d_x, const void* d_w9 /*[9][C]*/, const void* d_bias, void* d_y,
                        int N, int H, int W, int C, int act);
/* OSNet LightConv3x3 in one pass: y = relu(dw3x3(pw1x1(x)) + bias); w1 [C][C] (out, in), w9 [9][C], C in
 * {16,24,32}, W % 8 == 0, 18*(W+2)*PS*2 <= 65536 with PS = C rounded up to 16 (the band lives in LDS; SS_ERR_INVALID otherwise). */
int ss_op_lightconv_f16(void* stream, const void* d_x, const void* d_w1, const void* d_w9, const void* d_bias,
                        void* d_y, int N, int H, int W, int C);
/* The detector's first convolution: 3x3 / stride 2 / pad 1, 3 -> Cout in {16, 32, 48} channels, + bias + activation on
 * [B][H][W][3] half -> [B][(H-1)/2+1][W/2][Cout]; d_w_prep [4][3][Cout][16] = for conv columns c = 4n + r, per (ky, out
 * channel) the taps 3*kx+ch placed (6r + 5) % 8 halfs into a zero-padded 16-wide K window (fused.conv0_weight).  W % 128 == 0. */
int ss_op_conv0_f16(void* stream, const void* d_x, const void* d_w_prep, const void* d_bias, void* d_y, int B, int H, int W, int Cout,
                    int act);
/* Packed ReID batches: until the next call for the same stream, the OSNet-side entry points (ss_op_osnet_stem_f16,
 * ss_op_pointwise_f16, ss_op_osnet_streams_f16, ss_op_osnet_tail_f16) launched on `stream` BY THE CALLING THREAD with `batch`
 * images compute only the first *d_n of them (device int, read by the kernels; the grids stay fixed, so the launches can sit
 * in a captured graph).  d_n == NULL: off for that stream.  Per (host thread, stream): two pipelines in one process do not
 * see each other's setting.  At most 8 streams of a thread hold a setting at once (SS_ERR_CAPACITY). */
int ss_op_set_valid_images(void* stream, const int* d_n, int batch);
/* Process-wide A/B switches of the stateless operators, for measurements (default 1 each): "pw_epilogue" (16-byte vector
 * epilogue), "pw_splitk" (split-K form of small 3x3 layers), "osnet_chains" (register-resident LightConv row streams). */
int ss_op_set_option(const char* name, int value);
/* OSNet stem in one pass: conv 7x7/2 pad 3 (3 -> 16) + bias + ReLU + max pool 3x3/2 pad 1 on crops [N][H][128][3]
 * half -> [N][H/4][32][16]; d_w_prep [4][7][16][32] = for conv columns c = 4n + r, per (ky, out channel) the taps 3*kx+ch
 * placed (6r + 7) % 8 halfs into a zero-padded 32-wide K window (fused.stem_weight).  W == 128, H % 16 == 0.  d_w1 != NULL:
 * also d_y1 [N][H/4][32][16] = relu(conv1x1(d_y, d_w1 [16][16]) + d_b1) — the first OSBlock's conv1, bit-identical to
 * ss_op_pointwise_f16 on d_y. */
int ss_op_osnet_stem_f16(void* stream, const void* d_x, const void* d_w_prep, const void* d_bias, void* d_y, int N, int H,
                         int W, const void* d_w1 /*[16][16] or NULL*/, const void* d_b1, void* d_y1);
/* The four LightConv3x3 chains of an OSNet block (1..4 layers deep, same input) in one launch; the intermediates stay in
 * registers (16- / 32-wide images: a wave streams the rows of a band through the layers) or in LDS:
 * d_w1 [10][C][C] = the pointwise weights of the layers of the 1-, 2-, 3-, 4-deep chain in that order; d_dwtab = their depthwise
 * taps and biases as the kernels' matrix-core operand table (ss_op_dwtab_f16 over the same ten layers, built once per set of
 * weights); d_ys[4] the chain outputs [N][H][W][C]; d_psum [4][N][bands][C] float = per-band channel sums of each output
 * (for ss_op_gate_apply_f16 / ss_op_osnet_tail_f16), bands = ss_op_osnet_streams_bands(N, H, W, C).  C in {16,24,32},
 * W % 8 == 0, 24*(2W+2)*PS*2 <= 65536 with PS = C rounded up to 16. */
int ss_op_osnet_streams_bands(int N, int H, int W, int C);
int ss_op_osnet_streams_f16(void* stream, const void* d_x, const void* d_w1, const void* d_dwtab,
                            void* const* d_ys, float* d_psum, int N, int H, int W, int C);
/* The depthwise 3x3 taps d_w9 [layers][9][C] (tap-major) and biases d_bias [layers][C] of LightConv layers as the operand table of
 * the matrix-core depthwise (csrc/ss_ops.hip DwDiag / DwTab: per layer and 16 channels six diagonal operands of 17 x 16 bytes
 * + the bias in fp32) -> d_out, ss_op_dwtab_bytes(layers, C) bytes.  C in {16, 24, 32}. */
long long ss_op_dwtab_bytes(int layers, int C);
int ss_op_dwtab_f16(void* stream, const void* d_w9, const void* d_bias, int layers, int C, void* d_out);
/* Aggregation gate with the channel means given as `parts` partial sums per (stream, image) times `scale`. */
int ss_op_gate_apply_f16(void* stream, const void* const* d_xs, int T, const void* d_w1, const void* d_b1,
                         const void* d_w2, const void* d_b2, const float* d_sums, int parts, float scale, void* d_out,
                         int N, int HW, int C, int Cr);
/* Tail of an OSNet block fused with the 1x1 convolution that follows it (nets.OSBlock.forward + the next block's conv1 or the
 * stage's ConvBR [+ AvgPool2d(2,2)]): x2 = sum_t ys[t]*gate_t (as ss_op_gate_apply_f16), o = relu(w3 x2 + b3 + idn) -> d_out
 * (may be NULL), o2 = relu(w4 o + b4) -> d_out2, 2x2-averaged first when pool != 0.  ys[4] [N][H][W][MID], idn/out
 * [N][H][W][C2], out2 [N][H][W][N2] or [N][H/2][W/2][N2]; (MID, C2, N2) in {(16,64,16|64), (24,96,24|96), (32,128,32|128)},
 * H*W % 128 == 0, 128 % W == 0; C1 == 0: d_idn is the shortcut tensor [N][H][W][C2]; C1 > 0 ((C1, C2) in {(16,64), (64,96),
 * (96,128)}, N2 == MID): d_idn is the block INPUT [N][H][W][C1] and the shortcut is its 1x1 `down` convolution d_wd [C2][C1] +
 * d_bd (no activation), computed inside; d_gates_ws: scratch for the gates (a small kernel computes them once per image).
 * Bit-identical to the separate calls. */
int ss_op_osnet_tail_f16(void* stream, const void* const* d_ys, const float* d_psum, int parts, float scale, const void* d_gw1,
                         const void* d_gb1, const void* d_gw2, const void* d_gb2, int Cr, float* d_gates_ws /*[N][4][32]*/,
                         const void* d_w3, const void* d_b3, const void* d_idn, int C1, const void* d_wd, const void* d_bd,
                         void* d_out, const void* d_w4, const void* d_b4, void* d_out2, int pool, int N, int H, int W, int MID,
                         int C2, int N2);
/* concat(nearest-neighbour x2 upsampling of d_lo [B][h][w][C1], d_hi [B][2h][2w][C2]) along channels -> d_out
 * [B][2h][2w][C1+C2] (lo_first: the upsampled tensor's channels first), one pass (the detector neck). */
int ss_op_upcat_f16(void* stream, const void* d_lo, const void* d_hi, void* d_out, int B, int h, int w, int C1, int C2, int lo_first);
/* SPPF's pooling pyramid: d_out [B][H][W][4C] = concat(x, m(x), m(m(x)), m(m(m(x)))), m = max pool 5x5 / 1 / pad 2.
 * H*W <= 1024, C % 8 == 0. */
int ss_op_sppf_pools_f16(void* stream, const void* d_x, void* d_out, int B, int H, int W, int C);
/* C2PSA attention of the v11 detectors in one launch: d_qkv [B][N][heads * 128] half, per head [q 32 | k 32 | v 64] (the fused 1x1's
 * output, N = H * W positions <= 256), d_out[b][i][h * 64 + c] = sum_j v[c][j] softmax_j(scale q_i . k_j) (+ d_pe[b][i][h * 64 + c], the
 * depthwise positional term of v, when given); fp32 scores / softmax / accumulation, probabilities rounded to half. */
int ss_op_psa_attention_f16(void* stream, const void* d_qkv, const void* d_pe, void* d_out, int B, int N, int heads, float scale);
/* OSNet's head in one launch: d_out[n][f] = relu(sum_c d_w[f][c] * mean_hw(d_x[n][.][c]) + d_bias[f]); d_x [N][HW][C] half (the last 1x1's
 * output), C == 128; the means are rounded to half before the product (as the tensor a GEMM would read), fp32 accumulation.  Honours
 * ss_op_set_valid_images. */
int ss_op_osnet_head_f16(void* stream, const void* d_x, const void* d_w, const void* d_bias, void* d_out, int N, int HW, int C, int F);
/* 2x2 / stride 2 average pooling, NHWC half (H, W even; C % 8 == 0). */
int ss_op_avgpool2_f16(void* stream, const void* d_x, void* d_y, int N, int H, int W, int C);
/* k x k max pooling (stride, pad with -inf), output floor((H+2p-k)/s)+1. */
int ss_op_maxpool_f16(void* stream, const void* d_x, void* d_y, int N, int H, int W, int C, int k, int stride, int pad);
/* OSNet unified aggregation gate: out = sum_t x_t * sigmoid(fc2(relu(fc1(mean_hw(x_t))))), T <= 4 streams. */
int ss_op_gate_sum_f16(void* stream, const void* const* d_xs, int T, const void* d_w1, const void* d_b1,
                       const void* d_w2, const void* d_b2, float* d_means_ws, void* d_out, int N, int HW, int C, int Cr);

/* ---- ReID network in fp32 (the accuracy mode, csrc/ss_ops32.hip) -------------------------------------
 * OSNet-x0.25 with fp32 activations and weights on v_mfma_f32_16x16x4_f32: what stands behind the ReID forward pass inside
 * model.track (/root/reference/yolo_multi_model.py:41, which passes no half=) when the float distances must agree with a CPU
 * fp32 network to north_star's 1e-4.  All tensors dense NHWC float; weights [out][in] row-major float; d_nvalid (optional,
 * device int): only the first *d_nvalid images are computed (packed ReID batches), the launch grids stay fixed. */
/* relu?(W x + bias (+ res)): d_x [M][K], d_w [N][K], d_out / d_res [M][N].  (K, N) one of the OSNet-x0.25 pairs. */
int ss_op32_pointwise(void* stream, const void* d_x, const void* d_w, const void* d_bias, const void* d_res, void* d_out,
                      long long M, int K, int N, int relu, const int* d_nvalid, int img_px);
/* The four LightConv chains of an OSBlock (layer = 1x1 linear, depthwise 3x3 + bias + ReLU; chains 1, 2, 3, 4 layers deep, ten
 * layers in that order): d_x1 [N][H][W][C] -> d_ys[0..3] (same shape) and d_psum [4][N][bands][C] = channel sums of each output
 * per band, bands = ss_op32_chains_bands(N, H, W, C) (1 for the row-stream kernel, which batches of >= 128 images take).  d_w1 [10][C][C], d_w9 [10][9][C] (tap-major), d_bias [10][C].
 * (C, W) in {(16, 32), (24, 16), (32, 8)}. */
int ss_op32_chains_bands(int N, int H, int W, int C);
int ss_op32_chains(void* stream, const void* d_x1, const void* d_w1, const void* d_w9, const void* d_bias, void* const* d_ys,
                   float* d_psum, int N, int H, int W, int C, const int* d_nvalid);
/* OSBlock tail + the 1x1 ConvBR after it, two launches: gate_t = sigmoid(fc2 relu(fc1 mean_t + b1) + b2) (fc1 [hidden][MID], fc2
 * [MID][hidden]) -> d_gates [4][N][MID] (caller's workspace); then x2 = sum_t gate_t * ys[t], o = relu(w3 x2 + b3 + shortcut) -> d_out
 * (may be NULL), o2 = relu(w4 o + b4) -> d_out2, 2x2-averaged when pool.  shortcut = d_xin [N][H][W][C2] (C1 == 0) or wd d_xin + bd
 * with d_xin [N][H][W][C1]. */
int ss_op32_tail(void* stream, const void* const* d_ys, const float* d_psum, int bands, const void* d_gw1, const void* d_gb1,
                 const void* d_gw2, const void* d_gb2, int hidden, float* d_gates, const void* d_w3, const void* d_b3, const void* d_xin, int C1,
                 const void* d_wd, const void* d_bd, void* d_out, const void* d_w4, const void* d_b4, void* d_out2, int pool, int N,
                 int H, int W, int MID, int C2, int N2, const int* d_nvalid);
/* Process-wide A/B switches of the fp32 operators: "chains_form" 2 (default): k32_chainsR (register-resident row stream) where it
 * applies, 1: k32_chains3 (LDS phases), 0: k32_chains everywhere; "tail_wgs" n: workgroups of the persistent tail grid (0 = default);
 * "chains_probe" (measurement only, bit 0: the row-stream kernel does not store its outputs). */
int ss_op32_set_option(const char* name, int value);
/* conv 7x7 / 2 (3 -> 16) + bias + ReLU + max pool 3x3 / 2: d_x [N][256][128][3] -> d_y [N][64][32][16]; d_w [16][148] with
 * k = (ky * 7 + kx) * 3 + c and a zero in column 147. */
int ss_op32_stem(void* stream, const void* d_x, const void* d_w, const void* d_bias, void* d_y, int N, int H, int W, const int* d_nvalid);
/* The same operator on BYTE crops [N][256][128][3] (ss_crop_norm_batch / _packed with SS_DST_HWC | SS_DST_U8): /255, mean and std are applied
 * while the rows are staged, from a table built with ss_crop_norm's own expression - outputs bit-equal to ss_op32_stem on the float crops. */
int ss_op32_stem_u8(void* stream, const void* d_x, const void* d_w, const void* d_bias, void* d_y, int N, int H, int W, const int* d_nvalid);
/* ... and with the first OSBlock's conv1 (1x1, 16 -> 16, + bias + ReLU) applied to the pooled pixels by the same launch: d_y1 [N][64][32][16] =
 * relu(W1 d_y + b1), bit-equal to ss_op32_pointwise on d_y (the stem's output is not read back).  x_u8 != 0: byte crops. */
int ss_op32_stem_conv1(void* stream, const void* d_x, int x_u8, const void* d_w, const void* d_bias, void* d_y, const void* d_w1, const void* d_b1,
                       void* d_y1, int N, int H, int W, const int* d_nvalid);
/* d_out[n][f] = relu(sum_c d_w[f][c] * mean_hw(d_x[n][.][c]) + d_bias[f]), C == 128. */
int ss_op32_head(void* stream, const void* d_x, const void* d_w, const void* d_bias, void* d_out, int N, int HW, int C, int F,
                 const int* d_nvalid);
/* The detector's convolutions in fp32 (csrc/ss_ops32.hip k32_conv / k32_conv0; the arithmetic of the detector forward behind
 * /root/reference/yolo_multi_model.py:41, :173, which passes no half=).  ss_op32_conv: d_x NHWC [N][H][W][.] with pixel stride xs
 * floats (a channel slice of a wider tensor when xs > Cin), d_w [Cout][ks][ks][Cin], d_out NHWC [N][OH][OW][.] with pixel stride os,
 * d_res (may be NULL; added AFTER the activation) with pixel stride rs; ks 1 | 3, stride 1 | 2 (1x1: 1), pad ks / 2; act 1 = SiLU,
 * 0 = none; Cin, Cout multiples of 16, pointers 16-byte aligned, strides multiples of 4.  ss_op32_conv0: the first convolution,
 * 3 -> 16 channels, 3x3, stride 2, pad 1, d_x dense NHWC [N][H][W][3], d_w [16][3][3][3]. */
int ss_op32_conv(void* stream, const void* d_x, int xs, const void* d_w, const void* d_bias, const void* d_res, int rs, void* d_out,
                 int os, int N, int H, int W, int Cin, int Cout, int ks, int stride, int act);
int ss_op32_conv0(void* stream, const void* d_x, const void* d_w, const void* d_bias, void* d_out, int os, int N, int H, int W, int act);
/* cat(upsample2x_nearest(lo), hi) (lo_first != 0) or cat(hi, upsample2x_nearest(lo)) along channels in one pass (the neck's two
 * upsample + concat pairs), fp32 NHWC: d_lo [N][H/2][W/2][.] pixel stride ls, d_hi [N][H][W][.] pixel stride hs, d_out dense. */
/* YOLOv8 head decode in fp32 (DFL expectation, dist2bbox, stride scale, class sigmoid, level concat; the third branch as keypoint
 * triplets (ext_mode 1) or raw (0)): dense NHWC float branch outputs with their bias -> d_pred [B][4 + nc + n_ext][A]. */
int ss_op32_v8_decode(void* stream, const void* const* d_box, const void* const* d_cls, const void* const* d_ext, int n_ext, int ext_ld,
                      int ext_mode, const int* H, const int* W, const int* strides, int B, int nc, int cls_ld, float* d_pred);
/* SPPF's three chained 5x5 max pools + concat in one pass, fp32: d_x NHWC [N][H][W][.] (pixel stride xs) -> d_out dense [N][H][W][4 C]. */
int ss_op32_sppf_pools(void* stream, const void* d_x, int xs, void* d_out, int N, int H, int W, int C);
int ss_op32_upcat(void* stream, const void* d_lo, int ls, int Cl, const void* d_hi, int hs, int Ch, void* d_out, int N, int H, int W, int lo_first);

/* ---- profiling support ----------------------------------------------------------------------- */
/* Mean duration (ms) of the association (cosine gallery) kernel over the launches since the last
 * call, measured with HIP events on the context stream; also returns the launch count. */
int ss_assoc_timing(ss_ctx* ctx, int enable, float* mean_ms, int* launches);
/* The individual durations (ms, launch order) behind the mean the LAST ss_assoc_timing call returned: up to `cap` values into
 * out_ms, *n = how many there were.  For distributions (p50 / p95 / max) instead of a mean. */
int ss_assoc_timing_values(ss_ctx* ctx, float* out_ms, int cap, int* n);
/* The same kernel timed from inside: first workgroup start -> last workgroup end (100 MHz wall clock, written by the
 * kernel itself), i.e. without the time a dispatch waits for compute units behind other streams' kernels.  Returns
 * the mean over the launches since the last call (microseconds) and re-arms / disarms the stamps. */
int ss_assoc_inkernel_timing(ss_ctx* ctx, int enable, double* mean_us, int* launches);
/* enable = 2 above also records, for every workgroup of the last launch, the stamps of its first work item: [0] kernel
 * entry, [1] work record read, [2] first gallery pieces + detection operand arrived, [3] operand staged in LDS, [4..11]
 * k-segments 0..7 done, [12] results written.  out: [n_workgroups][16] 100 MHz ticks.  Profiling aid. */
int ss_assoc_timeline(ss_ctx* ctx, long long* out, int n_workgroups);

#ifdef __cplusplus
}
#endif
#endif
