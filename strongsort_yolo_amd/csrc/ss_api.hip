// ss_api.hip — C ABI (include/strongsort_hip.h) over the gfx950 kernels.  No torch types, no
// exceptions across the boundary; the context owns all device-resident tracker state.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "ss_common.h"

// launchers implemented in ss_track.hip / ss_front.hip
size_t ss_lsap_lds_bytes();
size_t ss_frame_lds_bytes(int, int, int);
void ss_launch_group_head(const SSDev&, const SSParams&, hipStream_t, hipEvent_t, hipEvent_t, hipEvent_t);
void ss_launch_group_chain(const SSDev&, const SSParams&, hipStream_t);
void ss_launch_normalize(const float*, int, float*, hipStream_t);
void ss_launch_ema(const float*, const float*, int, float, float, float*, hipStream_t);
void ss_launch_kf(int, double*, double*, const double*, const double*, int, double, double, hipStream_t);
void ss_launch_pack(const float*, int, int, float*, hipStream_t);
void ss_launch_assoc(const float*, const int*, int, const float*, int, const double*, const double*,
                     const double*, const SSParams&, float*, float*, double*, float*, double*, uint8_t*, hipStream_t);
void ss_launch_iou(const double*, int, const double*, int, double, double*, hipStream_t);
void ss_launch_lsap(const double*, int, int, int*, double*, int*, hipStream_t);
int  ss_front_init();
void ss_launch_letterbox(const uint8_t*, int, long long, int, int, int, void*, int, int, int, int, int, int, int, int, hipStream_t);
int  ss_launch_nms(const float*, int, long long, int, int, int, float, float, int, float, int, float, float, float, float,
                   float, const float*, float*, int, long long, int*, long long, int*, void*, size_t, unsigned long long,
                   unsigned long long, hipStream_t);
int* ss_nms_error_flag(void*, int);
size_t ss_nms_workspace_bytes();
void ss_launch_crop(const uint8_t*, int, long long, int, int, int, const float*, int, long long, int, const int*, void*, int, hipStream_t, const int*);
void ss_launch_crop_offsets(const int*, int, int, int*, hipStream_t);
void ss_launch_project(const double*, const double*, const double*, int, double, double*, double*, hipStream_t);
extern int ss_nms_fused;
void ss_launch_unpack_feats(const void*, int, const int*, const int*, int, int, float*, long long, hipStream_t);
void ss_launch_pack_results(const int*, const float*, int, int, const int*, const float*, int, int, float*, hipStream_t);
void ss_launch_overlay(uint8_t*, int, long long, int, int, int, const void*, const int*, const uint8_t*, const uint8_t*, hipStream_t);
void ss_launch_cmc(const uint8_t*, int, long long, int, int, int, uint8_t*, long long, int, int, int, int, int, double, int*, const int*, double*, hipStream_t);
extern "C" void ss_step_kernel_attr();

static std::string g_last_error;

struct ss_ctx {
    ss_config cfg;
    SSParams prm;
    SSDev dev;
    int device;
    hipStream_t stream;
    std::vector<void*> allocs;
    std::string err;
    // staging for the host convenience path
    float *d_dets, *d_feats, *d_out;
    int *d_ndets, *d_imghw, *d_nout;
    // KAT scratch
    float *kat_featfrag, *kat_partmin;
    double* kat_lsap_t;
    int* kat_err;
    void* nms_ws;               // nms_units workspace units (grown on demand outside graph capture)
    size_t nms_ws_bytes;
    int nms_units;
    unsigned long long cls_mask[2];   // classes the NMS keeps (ss_nms_set_classes); all ones = every class
    // host -> device upload staging (ss_upload): write-combined pinned buffers, used round robin
    struct Stage { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool busy = false; } stage[4];
    int stage_next = 0;
    Stage bstage[2];            // ss_upload_batch: a whole frame group per staging area
    int bstage_next = 0;
    uint8_t* font;              // [95][5] overlay font (ss_overlay_set_font)
    // N4 camera-motion compensation
    uint8_t* cmc_small;         // [FMAX+1][S][cmc_stride] down-scaled grey frames (index 0 = last frame of the previous group)
    size_t cmc_stride;          // bytes per small image (multiple of 16), 0 until the first ss_cmc_estimate
    int cmc_hw[2];              // frame size the buffer was made for
    int* cmc_prev_valid;        // [S]
    const double* cmc_warps;    // what ss_track_set_cmc installed
    hipEvent_t assoc_event;     // what ss_track_set_assoc_event installed (recorded after every association launch)
    struct Back { void* p = nullptr; size_t cap = 0; } back;      // device -> host staging (ss_download)
    int cos_grid;               // persistent workgroups of the association kernel
    int comp_rows;              // ragged last tiles with at most this many rows travel as 4-row groups of composite tiles (0: never)
    int assoc_stage;            // staging of the association kernel's detection operand (SSDev.assoc_stage)
    int xcd_map;                // SSDev.xcd_map
    int inkernel;               // in-kernel timing of the association kernel: 0 off, 1 duration, 2 + timeline
    // association-kernel timing
    bool timing;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t ev_used;
    std::vector<float> ev_ms;   // the durations behind the last ss_assoc_timing mean
    // the per-frame chain of a group (3 F - 1 launches) as captured HIP graphs, one per distinct kernel-argument set
    struct Chain { std::vector<char> key; hipGraph_t graph; hipGraphExec_t exec; unsigned long long used; hipStream_t last_st; };   // last_st: the stream of its last replay
    std::vector<Chain> chains;
    hipStream_t cap_stream;     // capture-only stream (nothing ever executes on it)
    int track_graph;            // option "track_graph": 1 (default) replay the chain as a graph for groups of >= 2 frames, 0 plain launches
    unsigned long long chain_clock;
    std::vector<unsigned long long> chain_seen;   // hashes of argument sets launched plainly once
    // option "chain_cus": the per-frame chain runs on a stream of its own that owns the first `chain_cus` compute units of the
    // CU mask (see ss_stream_create) instead of on the caller's stream
    int chain_cus;
    hipStream_t chain_stream;
    hipEvent_t ev_head, ev_chain;
    bool chain_pending;
};

static int fail(ss_ctx* c, int code, const std::string& msg)
{
    g_last_error = msg;
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(c, x)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess)                                                               \
            return fail(c, SS_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));     \
    } while (0)

template <typename T>
static int dalloc(ss_ctx* c, T** p, size_t n, bool zero = true)
{
    void* q = nullptr;
    HIPCHK(c, hipMalloc(&q, n * sizeof(T)));
    if (zero) HIPCHK(c, hipMemsetAsync(q, 0, n * sizeof(T), c->stream));
    c->allocs.push_back(q);
    *p = (T*)q;
    return SS_OK;
}

// CU masks: bit i of the mask = compute unit i / n_xcd of XCD i % n_xcd (the driver deals the bits of a queue's mask out to the
// XCDs in turn), so the first 8 k bits are k compute units of every XCD.
static int cu_mask_stream(ss_ctx* c, int first, int count, hipStream_t* out)
{
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c ? c->device : 0));
    const int total = prop.multiProcessorCount;
    if (first < 0 || count < 1 || first + count > total) return fail(c, SS_ERR_INVALID, "CU mask: range outside the device's compute units");
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    for (int i = first; i < first + count; ++i) mask[i >> 5] |= 1u << (i & 31);
    HIPCHK(c, hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data()));
    return SS_OK;
}

// a later launch on `s` sees what the detached chain wrote (no-op when nothing is detached)
static int chain_join(ss_ctx* c, hipStream_t s)
{
    if (c->chain_pending) HIPCHK(c, hipStreamWaitEvent(s, c->ev_chain, 0));
    return SS_OK;
}

extern "C" const char* ss_last_error(const ss_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

extern "C" int ss_create(const ss_config* cfg, int device, ss_ctx** out)
{
    if (!cfg || !out) return fail(nullptr, SS_ERR_INVALID, "ss_create: null argument");
    if (cfg->n_streams < 1 || cfg->nn_budget < 1 || cfg->nn_budget > SS_NRT * SS_TILE)
        return fail(nullptr, SS_ERR_INVALID, "ss_create: n_streams >= 1 and 1 <= nn_budget <= 128 required");
    ss_ctx* c = new ss_ctx();
    c->cfg = *cfg;
    c->device = device;
    c->stream = nullptr;
    c->timing = false;
    c->ev_used = 0;
    c->cos_grid = 512;           // persistent workgroups of the association kernel: two per CU
    c->comp_rows = 12;
    c->assoc_stage = 5;          // LDS-DMA in four pieces, first piece + first gallery piece only before the first barrier (r04: 37.8 -> 36.4 us per 1 x 32-frame launch)
    c->xcd_map = 0;
    c->cap_stream = nullptr; c->track_graph = 1; c->chain_clock = 0;
    c->chain_cus = 0; c->chain_stream = nullptr; c->ev_head = c->ev_chain = nullptr; c->chain_pending = false;
    c->inkernel = 0;
    c->cls_mask[0] = c->cls_mask[1] = ~0ull;
    c->cmc_small = nullptr; c->cmc_stride = 0; c->cmc_hw[0] = c->cmc_hw[1] = 0; c->cmc_warps = nullptr; c->assoc_event = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { int r = fail(nullptr, SS_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e)); delete c; return r; }
    SSParams& p = c->prm;
    p.max_dist = cfg->max_dist; p.max_iou_distance = cfg->max_iou_distance; p.mc_lambda = cfg->mc_lambda;
    p.gating_threshold = cfg->gating_threshold; p.gated_cost = cfg->gated_cost;
    p.wp = cfg->std_weight_position; p.wv = cfg->std_weight_velocity;
    p.ema_alpha = (float)cfg->ema_alpha; p.ema_one_minus_alpha = (float)(1.0 - cfg->ema_alpha);
    p.max_age = cfg->max_age; p.n_init = cfg->n_init; p.nn_budget = cfg->nn_budget; p.debug = cfg->debug;
    const size_t S = cfg->n_streams, T = SS_MAXT, D = SS_MAXD;
    SSDev& d = c->dev;
    memset(&d, 0, sizeof d);
    d.S = (int)S;
    d.budget = cfg->nn_budget;
    d.cap_cost = SS_COST_CAP; d.cap_t = SS_MAXT; d.cap_d = SS_MAXD;
    d.chain_merge = 1;
    d.pred_ahead = 1;
    d.assoc_pack = 1;
    int rc = SS_OK;
#define A(field, n) if (rc == SS_OK) rc = dalloc(c, &d.field, (n))
    A(n_tracks, S); A(next_id, S); A(frame, S); A(err, S); A(order, S * T);
    A(slot_used, S * T); A(track_id, S * T); A(state, S * T); A(hits, S * T); A(age, S * T); A(tsu, S * T);
    A(class_id, S * T); A(det_idx, S * T); A(gal_count, S * T); A(gal_head, S * T); A(conf, S * T);
    A(mean, S * T * 8); A(cov, S * T * 64); A(mean_p, S * T * 8); A(cov_p, S * T * 64); A(smooth, S * T * 2 * SS_F); A(smooth_sel, S * T);
    A(gallery, S * T * SS_NRT * SS_TILE_FLOATS);
    const size_t FM = SS_FMAX;
    A(feat_unit, FM * S * D * SS_F); A(feat_frag, FM * S * SS_NCT * SS_TILE_FLOATS);
    A(feat_pack, S * SS_NCTP * SS_TILE_FLOATS); A(colmap, S * (FM * D + 64));
    A(tlwh, FM * S * D * 4); A(xyah, FM * S * D * 4);
    A(M, S * T * FM * D); A(pl, S * SS_PLMAX); A(n_pl, S); A(pf, S * (FM + 1));
    d.items_cap = (int)(S * 4096);        // records per XCD list (<= F * max(cos_grid / (S F), pairs * ceil(tiles / SS_RECT)) per stream, spread over 8 lists)
    A(items, 8 * (size_t)d.items_cap * SS_RECI4); A(n_items, 8);
    A(post, S * T); A(n_post, S); A(rowlist, S * T); A(n_rows, S); A(cost_spill, S * T * D); A(frame_scratch, S * (T * 18 + D * 8)); A(tstamp, 4); A(timeline, 4096 * 16);
    if (cfg->debug) {
        A(dbg_cos, FM * S * T * D); A(dbg_maha, FM * S * T * D); A(dbg_cost_a, FM * S * T * D); A(dbg_cost_b, FM * S * T * D);
        A(dbg_gated, FM * S * T * D); A(dbg_lists, FM * S * 4 * T); A(dbg_counts, FM * S * 8);
    }
#undef A
    if (rc == SS_OK) rc = dalloc(c, &c->d_dets, S * D * 6);
    if (rc == SS_OK) rc = dalloc(c, &c->d_feats, S * D * SS_F);
    if (rc == SS_OK) rc = dalloc(c, &c->d_out, S * T * 8);
    if (rc == SS_OK) rc = dalloc(c, &c->d_ndets, S);
    if (rc == SS_OK) rc = dalloc(c, &c->d_imghw, S * 2);
    if (rc == SS_OK) rc = dalloc(c, &c->d_nout, S);
    if (rc == SS_OK) rc = dalloc(c, &c->kat_featfrag, (size_t)SS_NCT * SS_TILE_FLOATS);
    if (rc == SS_OK) rc = dalloc(c, &c->kat_partmin, T * SS_NRT * D);
    if (rc == SS_OK) rc = dalloc(c, &c->kat_lsap_t, (size_t)256 * 256);
    if (rc == SS_OK) rc = dalloc(c, &c->kat_err, 4);
    if (rc == SS_OK) rc = dalloc(c, &c->font, 95 * 5);
    if (rc == SS_OK) rc = dalloc(c, &c->cmc_prev_valid, S);
    c->nms_ws_bytes = ss_nms_workspace_bytes();
    c->nms_units = 1;
    if (rc == SS_OK) { char* w; rc = dalloc(c, &w, c->nms_ws_bytes); c->nms_ws = w; }
    if (rc == SS_OK) { ss_step_kernel_attr(); if (ss_front_init() != 0) rc = fail(c, SS_ERR_HIP, "kernel attribute setup failed"); }
    if (rc == SS_OK) { hipError_t e2 = hipStreamSynchronize(c->stream); if (e2 != hipSuccess) rc = fail(c, SS_ERR_HIP, hipGetErrorString(e2)); }
    if (rc != SS_OK) { std::string m = c->err; ss_destroy(c); g_last_error = m; return rc; }
    // next_id starts at 1
    std::vector<int> ones(S, 1);
    hipError_t e3 = hipMemcpy(d.next_id, ones.data(), S * sizeof(int), hipMemcpyHostToDevice);
    if (e3 != hipSuccess) { int r = fail(nullptr, SS_ERR_HIP, std::string("ss_create: ") + hipGetErrorString(e3)); ss_destroy(c); return r; }
    *out = c;
    return SS_OK;
}

extern "C" void ss_destroy(ss_ctx* c)
{
    if (!c) return;
    // teardown: nothing useful can be done with an error here
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& e : c->ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& g : c->chains) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
    if (c->chain_stream) { (void)hipStreamSynchronize(c->chain_stream); (void)hipStreamDestroy(c->chain_stream); }
    if (c->ev_head) (void)hipEventDestroy(c->ev_head);
    if (c->ev_chain) (void)hipEventDestroy(c->ev_chain);
    for (void* p : c->allocs) (void)hipFree(p);
    for (auto& st : c->stage) { if (st.ev) (void)hipEventDestroy(st.ev); if (st.p) (void)hipHostFree(st.p); }
    for (auto& st : c->bstage) { if (st.ev) (void)hipEventDestroy(st.ev); if (st.p) (void)hipHostFree(st.p); }
    if (c->back.p) (void)hipHostFree(c->back.p);
    delete c;
}

extern "C" int ss_set_hip_stream(ss_ctx* c, void* s)
{
    if (!c) return SS_ERR_INVALID;
    c->stream = (hipStream_t)s;
    return SS_OK;
}

// Frame upload: the caller's pageable buffer -> write-combined pinned staging (streaming CPU stores, no cache snooping
// on the DMA read) -> device, asynchronous on `hip_stream`.  Measured on the MI355X box for a 1280x720x3 frame: a
// cacheable pinned buffer costs 3.6 ms to fill + 2.8 ms to DMA right after the CPU wrote it; pageable hipMemcpy 1.7 ms.
extern "C" int ss_upload(ss_ctx* c, void* hip_stream, void* d_dst, const void* h_src, size_t bytes)
{
    if (!c || !d_dst || !h_src) return fail(c, SS_ERR_INVALID, "ss_upload: null argument");
    if (bytes == 0) return SS_OK;
    ss_ctx::Stage& st = c->stage[c->stage_next];
    c->stage_next = (c->stage_next + 1) & 3;
    if (st.busy) { HIPCHK(c, hipEventSynchronize(st.ev)); st.busy = false; }        // its previous upload has left the buffer
    if (st.cap < bytes) {
        if (st.p) { HIPCHK(c, hipHostFree(st.p)); st.p = nullptr; st.cap = 0; }
        HIPCHK(c, hipHostMalloc(&st.p, bytes, hipHostMallocWriteCombined));
        st.cap = bytes;
    }
    if (!st.ev) HIPCHK(c, hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    memcpy(st.p, h_src, bytes);
    HIPCHK(c, hipMemcpyAsync(d_dst, st.p, bytes, hipMemcpyHostToDevice, (hipStream_t)hip_stream));
    HIPCHK(c, hipEventRecord(st.ev, (hipStream_t)hip_stream));
    st.busy = true;
    return SS_OK;
}

// A group of frames at once: the n host buffers (bytes_each each) are copied into ONE write-combined staging area by `threads`
// host threads (a single thread fills write-combined memory at ~30 GB/s: 0.09 ms per 720p frame, 2.9 ms for a group of 32 — more
// than the GPU needs for the group), then leave in one asynchronous copy to d_dst (the frames contiguous there).  Two staging
// areas alternate; h_srcs may be reused on return.
extern "C" int ss_upload_batch(ss_ctx* c, void* hip_stream, void* d_dst, const void* const* h_srcs, int n, size_t bytes_each, int threads)
{
    if (!c || !d_dst || !h_srcs || n < 0 || threads < 1) return fail(c, SS_ERR_INVALID, "ss_upload_batch: bad argument");
    if (n == 0 || bytes_each == 0) return SS_OK;
    for (int i = 0; i < n; ++i) if (!h_srcs[i]) return fail(c, SS_ERR_INVALID, "ss_upload_batch: null frame");
    ss_ctx::Stage& st = c->bstage[c->bstage_next];                  // (the area is taken for good only when the copy has been enqueued)
    if (st.busy) { HIPCHK(c, hipEventSynchronize(st.ev)); st.busy = false; }
    const size_t bytes = (size_t)n * bytes_each;
    if (st.cap < bytes) {
        if (st.p) { HIPCHK(c, hipHostFree(st.p)); st.p = nullptr; st.cap = 0; }
        HIPCHK(c, hipHostMalloc(&st.p, bytes, hipHostMallocWriteCombined));
        st.cap = bytes;
    }
    if (!st.ev) HIPCHK(c, hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    // small groups are not worth the threads' start + join (tens of microseconds); nothing the standard library throws
    // (std::system_error when no thread can be started, bad_alloc) may cross the C boundary
    int T = threads < n ? threads : n;
    if (bytes < (size_t)4 << 20) T = 1;
    char* base = (char*)st.p;
    auto work = [=](int t, int nt) { for (int i = t; i < n; i += nt) memcpy(base + (size_t)i * bytes_each, h_srcs[i], bytes_each); };
    try {
        std::vector<std::thread> pool;
        pool.reserve(T > 1 ? T - 1 : 0);
        int started = 1;
        try {
            for (int t = 1; t < T; ++t) { pool.emplace_back(work, t, T); ++started; }
        } catch (...) {
            for (auto& th : pool) th.join();                       // the frames of the threads that never started are copied here
            for (int t = started; t < T; ++t) work(t, T);
            pool.clear();
        }
        work(0, T);
        for (auto& th : pool) th.join();
    } catch (...) {
        return fail(c, SS_ERR_INVALID, "ss_upload_batch: host staging failed");
    }
    HIPCHK(c, hipMemcpyAsync(d_dst, st.p, bytes, hipMemcpyHostToDevice, (hipStream_t)hip_stream));
    HIPCHK(c, hipEventRecord(st.ev, (hipStream_t)hip_stream));
    st.busy = true;
    c->bstage_next ^= 1;
    return SS_OK;
}

// Device -> host through a pinned staging buffer; synchronous (returns when h_dst holds the bytes).
extern "C" int ss_download(ss_ctx* c, void* hip_stream, void* h_dst, const void* d_src, size_t bytes)
{
    if (!c || !h_dst || !d_src) return fail(c, SS_ERR_INVALID, "ss_download: null argument");
    if (bytes == 0) return SS_OK;
    if (c->back.cap < bytes) {
        if (c->back.p) { HIPCHK(c, hipHostFree(c->back.p)); c->back.p = nullptr; c->back.cap = 0; }
        HIPCHK(c, hipHostMalloc(&c->back.p, bytes, hipHostMallocDefault));
        c->back.cap = bytes;
    }
    HIPCHK(c, hipMemcpyAsync(c->back.p, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    HIPCHK(c, hipStreamSynchronize((hipStream_t)hip_stream));
    memcpy(h_dst, c->back.p, bytes);
    return SS_OK;
}

// ---- N4 camera-motion compensation ------------------------------------------------------------------------
extern "C" int ss_cmc_estimate(ss_ctx* c, void* hip_stream, const uint8_t* d_frames, int n_frames, long long frame_stride, int h,
                               int w, int row_stride, const int* d_n_valid, double* d_warps)
{
    if (!c || !d_frames || !d_warps || n_frames < 1 || n_frames > SS_FMAX || h < 20 || w < 20 || row_stride < 3 * w)
        return fail(c, SS_ERR_INVALID, "ss_cmc_estimate: bad argument");
    const int hs = (int)(h * 0.1), ws = (int)(w * 0.1);
    const size_t stride = ((size_t)hs * ws + 15) / 16 * 16;
    if (c->cmc_hw[0] != h || c->cmc_hw[1] != w) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hip_stream) HIPCHK(c, hipStreamIsCapturing((hipStream_t)hip_stream, &cs));
        if (cs != hipStreamCaptureStatusNone)
            return fail(c, SS_ERR_INVALID, "ss_cmc_estimate: first call for a frame size must not be inside a graph capture");
        uint8_t* buf = nullptr;
        int rc = dalloc(c, &buf, (size_t)(SS_FMAX + 1) * c->dev.S * stride);
        if (rc) return rc;
        c->cmc_small = buf; c->cmc_stride = stride; c->cmc_hw[0] = h; c->cmc_hw[1] = w;
        HIPCHK(c, hipMemsetAsync(c->cmc_prev_valid, 0, (size_t)c->dev.S * 4, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    ss_launch_cmc(d_frames, n_frames * c->dev.S, frame_stride, h, w, row_stride, c->cmc_small, (long long)c->cmc_stride, c->dev.S,
                  n_frames, hs, ws, 100, 1e-5, c->cmc_prev_valid, d_n_valid, d_warps, (hipStream_t)hip_stream);
    HIPCHK(c, hipGetLastError());
    return SS_OK;
}

extern "C" int ss_track_set_cmc(ss_ctx* c, const double* d_warps)
{
    if (!c) return SS_ERR_INVALID;
    c->cmc_warps = d_warps;
    return SS_OK;
}

// ---- N2 overlay -----------------------------------------------------------------------------------------
extern "C" int ss_overlay_set_font(ss_ctx* c, const uint8_t* h_font_95x5)
{
    if (!c || !h_font_95x5) return fail(c, SS_ERR_INVALID, "ss_overlay_set_font: null argument");
    HIPCHK(c, hipMemcpy(c->font, h_font_95x5, 95 * 5, hipMemcpyHostToDevice));
    return SS_OK;
}

extern "C" int ss_overlay(ss_ctx* c, void* hip_stream, uint8_t* d_frames, int batch, long long frame_batch_stride, int h, int w,
                          int row_stride, const int* d_prims, const int* d_prim_off, const uint8_t* d_chars)
{
    if (!c || !d_frames || !d_prims || !d_prim_off || !d_chars || batch < 0 || batch > 65535 || row_stride < 3 * w)
        return fail(c, SS_ERR_INVALID, "ss_overlay: bad argument");
    ss_launch_overlay(d_frames, batch, frame_batch_stride, h, w, row_stride, d_prims, d_prim_off, d_chars, c->font, (hipStream_t)hip_stream);
    HIPCHK(c, hipGetLastError());
    return SS_OK;
}

extern "C" int ss_synchronize(ss_ctx* c)
{
    if (!c) return SS_ERR_INVALID;
    if (int rcj = chain_join(c, c->stream)) return rcj;               // a detached chain wrote what this call reads
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SS_OK;
}

extern "C" int ss_reset(ss_ctx* c, int stream)
{
    if (!c || stream >= c->dev.S) return fail(c, SS_ERR_INVALID, "ss_reset: bad stream");
    if (int rcj = chain_join(c, c->stream)) return rcj;               // a detached chain wrote what this call reads
    SSDev& d = c->dev;
    const int s0 = stream < 0 ? 0 : stream, s1 = stream < 0 ? d.S : stream + 1;
    const size_t T = SS_MAXT;
    for (int s = s0; s < s1; ++s) {
        int one = 1;
        HIPCHK(c, hipMemsetAsync(d.n_tracks + s, 0, 4, c->stream));
        HIPCHK(c, hipMemsetAsync(d.frame + s, 0, 4, c->stream));
        HIPCHK(c, hipMemsetAsync(d.err + s, 0, 4, c->stream));
        HIPCHK(c, hipMemsetAsync(d.slot_used + s * T, 0, T * 4, c->stream));
        HIPCHK(c, hipMemsetAsync(d.gal_count + s * T, 0, T * 4, c->stream));
        HIPCHK(c, hipMemsetAsync(d.gal_head + s * T, 0, T * 4, c->stream));
        HIPCHK(c, hipMemcpyAsync(d.next_id + s, &one, 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    HIPCHK(c, hipMemsetAsync(d.n_items, 0, 32, c->stream));
    for (int s = s0; s < s1; ++s) HIPCHK(c, hipMemsetAsync(c->cmc_prev_valid + s, 0, 4, c->stream));
    if (stream < 0)
        for (int u = 0; u < c->nms_units; ++u) HIPCHK(c, hipMemsetAsync(ss_nms_error_flag(c->nms_ws, u), 0, 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SS_OK;
}

// ---- tracker -----------------------------------------------------------------------------------
extern "C" int ss_track_update_group(ss_ctx* c, int n_frames, const float* d_dets, const int* d_ndets, const float* d_feats,
                                     const int* d_img_hw, float* d_out, int* d_nout)
{
    if (!c || !d_dets || !d_ndets || !d_feats || !d_img_hw || !d_out || !d_nout)
        return fail(c, SS_ERR_INVALID, "ss_track_update_group: null argument");
    if (n_frames < 1 || n_frames > SS_FMAX) return fail(c, SS_ERR_INVALID, "ss_track_update_group: 1 <= n_frames <= SS_FMAX");
    // a gallery ring position must be overwritten at most once per group (k_assoc's per-frame row mask, k_newrow)
    if (n_frames > c->cfg.nn_budget) return fail(c, SS_ERR_INVALID, "ss_track_update_group: n_frames <= nn_budget required");
    SSDev dev;
    memcpy(&dev, &c->dev, sizeof dev);                          // byte copy: the padding stays zero (the chain graphs are keyed by the struct's bytes)
    dev.F = n_frames;
    dev.dets = d_dets; dev.n_dets = d_ndets; dev.feats_raw = d_feats; dev.img_hw = (int*)d_img_hw;
    dev.out_rows = d_out; dev.n_out = d_nout;
    // Every launch dimension is fixed by (streams, n_frames): track and detection counts are device-side values read
    // from the work lists, so nothing here needs a host round trip and the sequence can be captured into a HIP graph.
    dev.cos_grid = c->cos_grid; dev.comp_rows = c->comp_rows; dev.assoc_stage = c->assoc_stage; dev.xcd_map = c->xcd_map;
    dev.ts_enable = c->inkernel;
    dev.cmc = c->cmc_warps;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        if (c->ev_used == c->ev.size()) {
            hipEvent_t a, b;
            HIPCHK(c, hipEventCreate(&a)); HIPCHK(c, hipEventCreate(&b));
            c->ev.emplace_back(a, b);
        }
        e0 = c->ev[c->ev_used].first; e1 = c->ev[c->ev_used].second; ++c->ev_used;
    }
    if (int rc = chain_join(c, c->stream)) return rc;          // the previous group's detached chain wrote the track tables this call reads
    ss_launch_group_head(dev, c->prm, c->stream, e0, e1, c->assoc_event);
    HIPCHK(c, hipGetLastError());
    // The chain: k_frame / k_post / k_newrow per frame, strictly dependent.  For a group of several frames that is up to 95
    // launches (~3.5 us of host time each); replayed as one graph it costs one hipGraphLaunch.  The kernels' arguments are
    // values (SSDev, SSParams, f): a graph is valid for exactly one (frames, caller buffers, options) combination, which is
    // what the key compares.  Not inside somebody else's stream capture (torch capturing the frame-at-a-time pipeline).
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (c->stream) HIPCHK(c, hipStreamIsCapturing(c->stream, &cs));
    // Detached chain (option "chain_cus"): the chain's one- to 64-workgroup kernels go to a stream whose queue owns a few compute
    // units, ordered after the association launch by an event; the caller's stream goes on with its next launches (the networks
    // of the following group) and ss_track_join makes whoever consumes the rows wait for the chain.
    hipStream_t chain_st = c->stream;
    struct Done { ss_ctx* c; hipStream_t st; bool on; ~Done() { if (on && hipEventRecord(c->ev_chain, st) == hipSuccess) c->chain_pending = true; } };
    const bool detach = c->chain_stream && cs == hipStreamCaptureStatusNone;
    if (detach) {
        HIPCHK(c, hipEventRecord(c->ev_head, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->chain_stream, c->ev_head, 0));
        chain_st = c->chain_stream;
    }
    Done done{ c, chain_st, detach };                          // recorded after whichever form of the chain is enqueued below
    if (!c->track_graph || n_frames < 2 || cs != hipStreamCaptureStatusNone) {
        ss_launch_group_chain(dev, c->prm, chain_st);
        HIPCHK(c, hipGetLastError());
        return SS_OK;
    }
    std::vector<char> key(sizeof(SSDev) + sizeof(SSParams));
    memcpy(key.data(), &dev, sizeof(SSDev));
    memcpy(key.data() + sizeof(SSDev), &c->prm, sizeof(SSParams));
    ss_ctx::Chain* hit = nullptr;
    for (auto& g : c->chains) if (g.key == key) { hit = &g; break; }
    if (!hit) {
        // a combination is captured the SECOND time it is seen: a caller that walks through fresh buffers (every call another
        // slice of a long array) would otherwise pay a capture + instantiation per call and never replay anything
        unsigned long long h = 1469598103934665603ull;
        for (char ch : key) h = (h ^ (unsigned char)ch) * 1099511628211ull;
        bool seen = false;
        for (unsigned long long v : c->chain_seen) seen = seen || v == h;
        if (!seen) {
            if (c->chain_seen.size() >= 64) c->chain_seen.erase(c->chain_seen.begin());
            c->chain_seen.push_back(h);
            ss_launch_group_chain(dev, c->prm, chain_st);
            HIPCHK(c, hipGetLastError());
            return SS_OK;
        }
        // From here on the group's head (k_group_prep + k_assoc) is enqueued: whatever fails below, the chain still goes out (as
        // plain launches) before the error is reported — a group is never left half-applied.
        auto plain_then = [&](hipError_t e) -> int {
            (void)hipGetLastError();
            ss_launch_group_chain(dev, c->prm, chain_st);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, e);
            return SS_OK;
        };
        if (!c->cap_stream) { const hipError_t e = hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking); if (e != hipSuccess) { c->cap_stream = nullptr; return plain_then(e); } }
        if (c->chains.size() >= 16) {                            // least recently used entry goes
            size_t lru = 0;
            for (size_t i = 1; i < c->chains.size(); ++i) if (c->chains[i].used < c->chains[lru].used) lru = i;
            // its last replay may still be running (16 other launches is no guarantee on a detached, slow chain), and on ANOTHER stream
            // than today's if chain_cus / ss_set_hip_stream changed in between: the stream it was last launched on is drained before
            // the executable graph goes
            const hipError_t e = hipStreamSynchronize(c->chains[lru].last_st);
            if (e != hipSuccess) return plain_then(e);
            (void)hipGraphExecDestroy(c->chains[lru].exec); (void)hipGraphDestroy(c->chains[lru].graph);
            c->chains.erase(c->chains.begin() + lru);
        }
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        // A failed capture / instantiation must not leave the group half-applied (its head, k_group_prep + k_assoc, is already
        // enqueued): the chain then goes out as plain launches, the captured graph (if any) is released.
        hipError_t be = hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal), le = hipSuccess, ee = hipSuccess, ie = hipSuccess;
        if (be == hipSuccess) {
            ss_launch_group_chain(dev, c->prm, c->cap_stream);
            le = hipGetLastError();
            ee = hipStreamEndCapture(c->cap_stream, &graph);
            if (le == hipSuccess && ee == hipSuccess && graph) ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        }
        if (be != hipSuccess || le != hipSuccess || ee != hipSuccess || !graph || ie != hipSuccess || !exec) {
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            ss_launch_group_chain(dev, c->prm, chain_st);
            HIPCHK(c, hipGetLastError());
            return SS_OK;
        }
        c->chains.push_back({ key, graph, exec, 0, chain_st });
        hit = &c->chains.back();
    }
    hit->used = ++c->chain_clock;
    hit->last_st = chain_st;
    if (hipGraphLaunch(hit->exec, chain_st) != hipSuccess) {     // (the replay did not start: the same kernels as plain launches)
        (void)hipGetLastError();
        ss_launch_group_chain(dev, c->prm, chain_st);
        HIPCHK(c, hipGetLastError());
    }
    return SS_OK;
}

extern "C" int ss_track_join(ss_ctx* c, void* hip_stream)
{
    if (!c) return SS_ERR_INVALID;
    return chain_join(c, (hipStream_t)hip_stream);
}

extern "C" int ss_stream_create(ss_ctx* c, int skip_cus, void** out)
{
    if (!c || !out || skip_cus < 0) return fail(c, SS_ERR_INVALID, "ss_stream_create: bad argument");
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    hipStream_t st = nullptr;
    if (skip_cus == 0) HIPCHK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    else if (int rc = cu_mask_stream(c, skip_cus, prop.multiProcessorCount - skip_cus, &st)) return rc;
    *out = (void*)st;
    return SS_OK;
}

extern "C" int ss_stream_destroy(ss_ctx* c, void* hip_stream)
{
    if (!c || !hip_stream) return SS_ERR_INVALID;
    HIPCHK(c, hipStreamDestroy((hipStream_t)hip_stream));
    return SS_OK;
}

extern "C" int ss_max_group_frames(void) { return SS_FMAX; }

extern "C" int ss_track_set_assoc_event(ss_ctx* c, void* hip_event)
{
    if (!c) return SS_ERR_INVALID;
    c->assoc_event = (hipEvent_t)hip_event;
    return SS_OK;
}

extern "C" int ss_track_update(ss_ctx* c, const float* d_dets, const int* d_ndets, const float* d_feats,
                               const int* d_img_hw, float* d_out, int* d_nout)
{
    return ss_track_update_group(c, 1, d_dets, d_ndets, d_feats, d_img_hw, d_out, d_nout);
}

extern "C" int ss_track_update_host(ss_ctx* c, int stream, const float* h_dets, int n, const float* h_feats,
                                    int img_h, int img_w, float* h_out, int cap_rows, int* n_out)
{
    if (!c || stream < 0 || stream >= c->dev.S || n < 0 || !n_out) return fail(c, SS_ERR_INVALID, "ss_track_update_host: bad argument");
    if (n > SS_MAXD) return fail(c, SS_ERR_CAPACITY, "ss_track_update_host: more than SS_MAX_DETS detections");
    const int S = c->dev.S;
    // other streams see 0 detections this call only if they are not driven: to keep streams
    // independent the host path requires a single-stream context.
    if (S != 1) return fail(c, SS_ERR_INVALID, "ss_track_update_host: context must hold exactly one stream");
    int hw[2] = { img_h, img_w };
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->d_dets, h_dets, (size_t)n * 6 * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_feats, h_feats, (size_t)n * SS_F * 4, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(c->d_ndets, &n, 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_imghw, hw, 8, hipMemcpyHostToDevice, c->stream));
    int rc = ss_track_update(c, c->d_dets, c->d_ndets, c->d_feats, c->d_imghw, c->d_out, c->d_nout);
    if (rc) return rc;
    if (int rcj = chain_join(c, c->stream)) return rcj;
    int cnt[2] = { 0, 0 }, err = 0;
    HIPCHK(c, hipMemcpyAsync(&cnt[0], c->d_nout, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&err, c->dev.err, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (err) return fail(c, err, err == SS_ERR_CAPACITY ? "tracker capacity exceeded on device" : "assignment infeasible on device");
    *n_out = cnt[0];
    if (cnt[0] > cap_rows) return fail(c, SS_ERR_CAPACITY, "ss_track_update_host: output buffer too small");
    if (cnt[0]) HIPCHK(c, hipMemcpy(h_out, c->d_out, (size_t)cnt[0] * 8 * 4, hipMemcpyDeviceToHost));
    return SS_OK;
}

extern "C" int ss_set_option(ss_ctx* c, const char* name, int value)
{
    if (!c || !name) return fail(c, SS_ERR_INVALID, "ss_set_option: null argument");
    const std::string n(name);
    if (n == "cos_grid") { if (value < 8 || value > 4096 || value % 8) return fail(c, SS_ERR_INVALID, "cos_grid: a multiple of 8 in 8..4096"); c->cos_grid = value; }
    else if (n == "nms_fused") ss_nms_fused = value != 0;       // process-wide: one workgroup per image after the filter (1, default) or sort / mask / scan launches
    else if (n == "assoc_comp_rows") { if (value < 0 || value > 12) return fail(c, SS_ERR_INVALID, "assoc_comp_rows: 0..12"); c->comp_rows = value; }
    else if (n == "assoc_stage") { if (value != 0 && value != 1 && value != 2 && value != 4 && value != 5) return fail(c, SS_ERR_INVALID, "assoc_stage: 0, 1, 2, 4 or 5"); c->assoc_stage = value; }
    else if (n == "assoc_pack") c->dev.assoc_pack = value != 0;         // association on detection columns packed across the group's frames (default) / per-frame pairs
    else if (n == "pred_ahead") c->dev.pred_ahead = value != 0;         // k_frame takes the predicted state post_track left (default) / predicts itself
    else if (n == "chain_merge") c->dev.chain_merge = value != 0;       // k_post + k_newrow of a frame as one launch (default) / two launches
    else if (n == "frame_caps") {
        // value = tracks and detections k_frame keeps in LDS (cost entries = value^2): 0 restores the maxima (256 tracks, 128 detections, 12288 entries)
        // (112 is the largest square that fits: 128 x 128 cost entries + the per-track areas would be 173 KB of LDS, past the CU's 160)
        if (value != 0 && (value < 16 || value > 112 || value % 16)) return fail(c, SS_ERR_INVALID, "frame_caps: 0 or a multiple of 16 in 16..112");
        c->dev.cap_t = value ? value : SS_MAXT; c->dev.cap_d = value ? value : SS_MAXD;
        c->dev.cap_cost = value ? (value * value < SS_COST_CAP ? value * value : SS_COST_CAP) : SS_COST_CAP;
    }
    else if (n == "chain_cus") {
        if (value < -1 || value > 128 || (value > 0 && value % 8)) return fail(c, SS_ERR_INVALID, "chain_cus: 0 (off), -1 (detached, no reservation) or a multiple of 8 up to 128");
        if (c->chain_stream) { HIPCHK(c, hipStreamSynchronize(c->chain_stream)); HIPCHK(c, hipStreamDestroy(c->chain_stream)); c->chain_stream = nullptr; }
        c->chain_pending = false;
        if (value) {
            if (value < 0) {                                     // detached onto a plain high-priority stream: every compute unit, no reservation
                int lo = 0, hi = 0;
                HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
                HIPCHK(c, hipStreamCreateWithPriority(&c->chain_stream, hipStreamNonBlocking, hi));
            } else if (int rc = cu_mask_stream(c, 0, value, &c->chain_stream)) return rc;
            if (!c->ev_head) { HIPCHK(c, hipEventCreateWithFlags(&c->ev_head, hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&c->ev_chain, hipEventDisableTiming)); }
        }
        c->chain_cus = value;
    }
    else if (n == "track_graph") { if (value != 0 && value != 1) return fail(c, SS_ERR_INVALID, "track_graph: 0 or 1"); c->track_graph = value; }
    else if (n == "assoc_xcd_map") { if (value != 0 && value != 1) return fail(c, SS_ERR_INVALID, "assoc_xcd_map: 0 or 1"); c->xcd_map = value; }
    else return fail(c, SS_ERR_INVALID, "ss_set_option: unknown option '" + n + "'");
    return SS_OK;
}

extern "C" int ss_check_errors(ss_ctx* c)
{
    if (!c) return SS_ERR_INVALID;
    if (int rcj = chain_join(c, c->stream)) return rcj;               // a detached chain wrote what this call reads
    std::vector<int> e(c->dev.S);
    HIPCHK(c, hipMemcpyAsync(e.data(), c->dev.err, e.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t s = 0; s < e.size(); ++s)
        if (e[s]) return fail(c, e[s], "device error flag on stream " + std::to_string(s));
    for (int u = 0; u < c->nms_units; ++u) {                 // NMS candidate overflow (flag stays set until ss_reset)
        int f = 0;
        HIPCHK(c, hipMemcpy(&f, ss_nms_error_flag(c->nms_ws, u), 4, hipMemcpyDeviceToHost));
        if (f) return fail(c, f, "ss_nms: more than 8192 candidates above conf_thres (image " + std::to_string(u) + " of the batch)");
    }
    return SS_OK;
}

// ---- stage entry points ---------------------------------------------------------------------------
extern "C" int ss_feat_normalize(ss_ctx* c, const float* raw, int n, float* unit)
{ if (!c) return SS_ERR_INVALID; ss_launch_normalize(raw, n, unit, c->stream); HIPCHK(c, hipGetLastError()); return SS_OK; }

extern "C" int ss_ema(ss_ctx* c, const float* s, const float* f, int n, float* o)
{ if (!c) return SS_ERR_INVALID; ss_launch_ema(s, f, n, c->prm.ema_alpha, c->prm.ema_one_minus_alpha, o, c->stream); HIPCHK(c, hipGetLastError()); return SS_OK; }

extern "C" int ss_kf_predict(ss_ctx* c, double* mean, double* cov, int n)
{ if (!c) return SS_ERR_INVALID; ss_launch_kf(0, mean, cov, nullptr, nullptr, n, c->prm.wp, c->prm.wv, c->stream); HIPCHK(c, hipGetLastError()); return SS_OK; }

extern "C" int ss_kf_update(ss_ctx* c, double* mean, double* cov, const double* z, const double* conf, int n)
{ if (!c) return SS_ERR_INVALID; ss_launch_kf(1, mean, cov, z, conf, n, c->prm.wp, c->prm.wv, c->stream); HIPCHK(c, hipGetLastError()); return SS_OK; }

extern "C" int ss_kf_project(ss_ctx* c, const double* mean, const double* cov, const double* conf, int n, double* zmean, double* S)
{
    if (!c || !mean || !cov || !zmean || !S || n < 0) return fail(c, SS_ERR_INVALID, "ss_kf_project: bad argument");
    ss_launch_project(mean, cov, conf, n, c->prm.wp, zmean, S, c->stream); HIPCHK(c, hipGetLastError()); return SS_OK;
}

extern "C" int ss_kf_initiate(ss_ctx* c, const double* z, int n, double* mean, double* cov)
{ if (!c) return SS_ERR_INVALID; ss_launch_kf(2, mean, cov, z, nullptr, n, c->prm.wp, c->prm.wv, c->stream); HIPCHK(c, hipGetLastError()); return SS_OK; }

extern "C" int ss_gallery_pack(ss_ctx* c, const float* nat, int T, int B, float* frag)
{
    if (!c || B > SS_NRT * SS_TILE) return fail(c, SS_ERR_INVALID, "ss_gallery_pack: B > 128");
    ss_launch_pack(nat, T, B, frag, c->stream); HIPCHK(c, hipGetLastError()); return SS_OK;
}

extern "C" int ss_assoc_cost(ss_ctx* c, const float* gal, const int* counts, int T, const float* feats, int D,
                             const double* mean, const double* cov, const double* xyah, double* cost,
                             float* cosd, double* maha, uint8_t* gated)
{
    if (!c) return SS_ERR_INVALID;
    if (T > SS_MAXT || D > SS_MAXD) return fail(c, SS_ERR_CAPACITY, "ss_assoc_cost: T <= 256, D <= 128");
    ss_launch_assoc(gal, counts, T, feats, D, mean, cov, xyah, c->prm, c->kat_featfrag, c->kat_partmin,
                    cost, cosd, maha, gated, c->stream);
    HIPCHK(c, hipGetLastError());
    return SS_OK;
}

extern "C" int ss_iou_cost(ss_ctx* c, const double* t, int T, const double* d, int D, double* cost)
{ if (!c) return SS_ERR_INVALID; ss_launch_iou(t, T, d, D, c->prm.max_iou_distance, cost, c->stream); HIPCHK(c, hipGetLastError()); return SS_OK; }

extern "C" int ss_lsap(ss_ctx* c, const double* cost, int nr, int nc, int* r2c)
{
    if (!c) return SS_ERR_INVALID;
    if (nr > 256 || nc > 256) return fail(c, SS_ERR_CAPACITY, "ss_lsap: at most 256 x 256");
    if (nr == 0) return SS_OK;
    HIPCHK(c, hipMemsetAsync(c->kat_err, 0, 4, c->stream));
    ss_launch_lsap(cost, nr, nc, r2c, c->kat_lsap_t, c->kat_err, c->stream);
    HIPCHK(c, hipGetLastError());
    int e = 0;
    HIPCHK(c, hipMemcpyAsync(&e, c->kat_err, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (e) return fail(c, e, "ss_lsap: infeasible cost matrix");
    return SS_OK;
}

// ---- front end ----------------------------------------------------------------------------------------
extern "C" int ss_letterbox_batch(ss_ctx* c, const uint8_t* src, int batch, long long src_batch_stride, int h, int w,
                                  int stride, void* dst, int dst_flags, int out_h, int out_w, int new_h, int new_w,
                                  int pad_top, int pad_left, int pad_value)
{
    if (!c || !src || !dst || new_h < 1 || new_w < 1 || batch < 0 || batch > 65535 || out_h > 65535)
        return fail(c, SS_ERR_INVALID, "ss_letterbox: bad argument");
    ss_launch_letterbox(src, batch, src_batch_stride, h, w, stride, dst, dst_flags, out_h, out_w, new_h, new_w, pad_top,
                        pad_left, pad_value, c->stream);
    HIPCHK(c, hipGetLastError());
    return SS_OK;
}

extern "C" int ss_letterbox(ss_ctx* c, const uint8_t* src, int h, int w, int stride, void* dst, int dst_flags,
                            int out_h, int out_w, int new_h, int new_w, int pad_top, int pad_left, int pad_value)
{
    return ss_letterbox_batch(c, src, 1, 0, h, w, stride, dst, dst_flags, out_h, out_w, new_h, new_w, pad_top, pad_left,
                              pad_value);
}

// workspace for `batch` images; growing it allocates, which a stream capture does not allow
static int nms_reserve(ss_ctx* c, int batch)
{
    if (batch <= c->nms_units) return SS_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (c->stream) HIPCHK(c, hipStreamIsCapturing(c->stream, &cs));
    if (cs != hipStreamCaptureStatusNone)
        return fail(c, SS_ERR_INVALID, "ss_nms_batch: first call with this batch size must not be inside a graph capture");
    char* w = nullptr;
    int rc = dalloc(c, &w, c->nms_ws_bytes * (size_t)batch);
    if (rc) return rc;
    c->nms_ws = w;               // the smaller workspace stays owned by the context until ss_destroy
    c->nms_units = batch;
    return SS_OK;
}

extern "C" int ss_nms_batch(ss_ctx* c, const float* pred, int batch, long long pred_batch_stride, int n_anchors, int nc,
                            int n_extra, float conf_thres, float iou_thres, int agnostic, float max_wh, int max_det,
                            const float* d_geom, float* rows, int row_stride, long long rows_batch_stride, int* keep,
                            long long keep_batch_stride, int* count)
{
    if (!c || !pred || !d_geom || !rows || !keep || !count || row_stride < 6 + n_extra || batch < 0 || batch > 65535)
        return fail(c, SS_ERR_INVALID, "ss_nms_batch: bad argument");
    int rc = nms_reserve(c, batch);
    if (rc) return rc;
    rc = ss_launch_nms(pred, batch, pred_batch_stride, n_anchors, nc, n_extra, conf_thres, iou_thres, agnostic, max_wh,
                       max_det, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, d_geom, rows, row_stride, rows_batch_stride, keep,
                       keep_batch_stride, count, c->nms_ws, c->nms_ws_bytes * (size_t)c->nms_units, c->cls_mask[0], c->cls_mask[1],
                       c->stream);
    if (rc) return fail(c, rc, "ss_nms_batch: anchors exceed the workspace (n_anchors <= 32768)");
    HIPCHK(c, hipGetLastError());
    return SS_OK;
}

extern "C" int ss_nms_set_classes(ss_ctx* c, const int* classes, int n)
{
    if (!c || n < 0 || (n > 0 && !classes)) return fail(c, SS_ERR_INVALID, "ss_nms_set_classes: bad argument");
    if (n == 0) { c->cls_mask[0] = c->cls_mask[1] = ~0ull; return SS_OK; }
    unsigned long long m[2] = { 0, 0 };
    for (int i = 0; i < n; ++i) {
        if (classes[i] < 0 || classes[i] >= 128) return fail(c, SS_ERR_INVALID, "ss_nms_set_classes: class ids 0..127");
        m[classes[i] >> 6] |= 1ull << (classes[i] & 63);
    }
    c->cls_mask[0] = m[0]; c->cls_mask[1] = m[1];
    return SS_OK;
}

extern "C" int ss_nms(ss_ctx* c, const float* pred, int n_anchors, int nc, int n_extra, float conf_thres,
                      float iou_thres, int agnostic, float max_wh, int max_det, float gain, float pad_x,
                      float pad_y, float w0, float h0, float* rows, int row_stride, int* keep, int* count)
{
    if (!c || !pred || !rows || !keep || !count || row_stride < 6 + n_extra)
        return fail(c, SS_ERR_INVALID, "ss_nms: bad argument");
    int rc = ss_launch_nms(pred, 1, 0, n_anchors, nc, n_extra, conf_thres, iou_thres, agnostic, max_wh, max_det, gain,
                           pad_x, pad_y, w0, h0, nullptr, rows, row_stride, 0, keep, 0, count, c->nms_ws,
                           c->nms_ws_bytes * (size_t)c->nms_units, c->cls_mask[0], c->cls_mask[1], c->stream);
    if (rc) return fail(c, rc, "ss_nms: anchors exceed the workspace (n_anchors <= 32768)");
    HIPCHK(c, hipGetLastError());
    return SS_OK;
}

extern "C" int ss_crop_norm_batch(ss_ctx* c, const uint8_t* frames, int batch, long long frame_batch_stride, int h,
                                  int w, int stride, const float* dets, int det_stride, long long dets_batch_stride,
                                  int n, const int* d_counts, void* out, int out_flags)
{
    if (!c || !frames || !dets || !out || batch < 0 || n < 0 || (long long)batch * n > 65535)
        return fail(c, SS_ERR_INVALID, "ss_crop_norm: bad argument (batch * n <= 65535)");
    if ((out_flags & 4) && !(out_flags & 2)) return fail(c, SS_ERR_INVALID, "ss_crop_norm: SS_DST_U8 needs SS_DST_HWC");
    ss_launch_crop(frames, batch, frame_batch_stride, h, w, stride, dets, det_stride, dets_batch_stride, n, d_counts, out,
                   out_flags, c->stream, nullptr);
    HIPCHK(c, hipGetLastError());
    return SS_OK;
}

// Packed form: d_off[batch + 1] <- exclusive prefix of min(count, n) (d_off[batch] = crops in total), crop d of image i is
// written at slot d_off[i] + d.  The network behind it then computes *(d_off + batch) crops (ss_op_set_valid_images) and
// ss_unpack_feats puts the embeddings back at [image][d].
extern "C" int ss_crop_norm_packed(ss_ctx* c, const uint8_t* frames, int batch, long long frame_batch_stride, int h, int w,
                                   int stride, const float* dets, int det_stride, long long dets_batch_stride, int n,
                                   const int* d_counts, int* d_off, void* out, int out_flags)
{
    if (!c || !frames || !dets || !out || !d_counts || !d_off || batch < 1 || n < 1 || (long long)batch * n > 65535 || !(out_flags & 2))
        return fail(c, SS_ERR_INVALID, "ss_crop_norm_packed: bad argument (channels-last output, batch * n <= 65535)");
    ss_launch_crop_offsets(d_counts, batch, n, d_off, c->stream);
    ss_launch_crop(frames, batch, frame_batch_stride, h, w, stride, dets, det_stride, dets_batch_stride, n, d_counts, out,
                   out_flags, c->stream, d_off);
    HIPCHK(c, hipGetLastError());
    return SS_OK;
}

extern "C" int ss_unpack_feats(ss_ctx* c, const void* d_emb, int emb_half, const int* d_off, const int* d_counts, int batch, int n,
                               float* d_feats, long long feats_image_stride)
{
    if (!c || !d_emb || !d_off || !d_counts || !d_feats || batch < 1 || n < 1 || n > 65535 || batch > 65535)
        return fail(c, SS_ERR_INVALID, "ss_unpack_feats: bad argument");
    ss_launch_unpack_feats(d_emb, emb_half, d_off, d_counts, batch, n, d_feats, feats_image_stride, c->stream);
    HIPCHK(c, hipGetLastError());
    return SS_OK;
}

extern "C" int ss_pack_results(ss_ctx* c, const int* d_n_dets, const float* d_dets, int det_ld, int det_cap, const int* d_n_out,
                               const float* d_out, int out_ld, int out_cap, float* dst)
{
    if (!c || !d_n_dets || !d_dets || !dst || det_ld < 1 || det_cap < 1 || ((d_n_out || d_out) && (!d_n_out || !d_out || out_ld < 1 || out_cap < 1)))
        return fail(c, SS_ERR_INVALID, "ss_pack_results: bad argument");
    ss_launch_pack_results(d_n_dets, d_dets, det_ld, det_cap, d_n_out, d_out, out_ld, out_cap, dst, c->stream);
    HIPCHK(c, hipGetLastError());
    return SS_OK;
}

extern "C" int ss_crop_norm(ss_ctx* c, const uint8_t* frame, int h, int w, int stride, const float* dets,
                            int det_stride, int n, const int* d_count, void* out, int out_flags)
{
    return ss_crop_norm_batch(c, frame, 1, 0, h, w, stride, dets, det_stride, 0, n, d_count, out, out_flags);
}

// ---- inspection ----------------------------------------------------------------------------------------
extern "C" int ss_get_tracks(ss_ctx* c, int s, int cap, int* n_tracks, int* next_id, int* track_id, int* state,
                             int* hits, int* age, int* tsu, int* class_id, float* conf, double* mean,
                             double* cov, float* smooth, int* gal_count)
{
    if (!c || s < 0 || s >= c->dev.S) return fail(c, SS_ERR_INVALID, "ss_get_tracks: bad stream");
    if (int rcj = chain_join(c, c->stream)) return rcj;               // a detached chain wrote what this call reads
    HIPCHK(c, hipStreamSynchronize(c->stream));
    SSDev& d = c->dev;
    int nt = 0, nid = 0;
    HIPCHK(c, hipMemcpy(&nt, d.n_tracks + s, 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&nid, d.next_id + s, 4, hipMemcpyDeviceToHost));
    if (n_tracks) *n_tracks = nt;
    if (next_id) *next_id = nid;
    if (nt > cap) return fail(c, SS_ERR_CAPACITY, "ss_get_tracks: cap too small");
    const size_t T = SS_MAXT, sb = (size_t)s * T;
    std::vector<int> order(T), tmp(T);
    HIPCHK(c, hipMemcpy(order.data(), d.order + sb, T * 4, hipMemcpyDeviceToHost));
    auto gather_i = [&](const int* src, int* dst) -> int {
        if (!dst) return 0;
        if (hipMemcpy(tmp.data(), src + sb, T * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        for (int i = 0; i < nt; ++i) dst[i] = tmp[order[i]];
        return 0;
    };
    if (gather_i(d.track_id, track_id) || gather_i(d.state, state) || gather_i(d.hits, hits) || gather_i(d.age, age) ||
        gather_i(d.tsu, tsu) || gather_i(d.class_id, class_id) || gather_i(d.gal_count, gal_count))
        return fail(c, SS_ERR_HIP, "ss_get_tracks: copy failed");
    if (conf) {
        std::vector<float> t(T);
        HIPCHK(c, hipMemcpy(t.data(), d.conf + sb, T * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < nt; ++i) conf[i] = t[order[i]];
    }
    for (int i = 0; i < nt; ++i) {
        const size_t g = sb + order[i];
        if (mean) HIPCHK(c, hipMemcpy(mean + (size_t)i * 8, d.mean + g * 8, 64, hipMemcpyDeviceToHost));
        if (cov) HIPCHK(c, hipMemcpy(cov + (size_t)i * 64, d.cov + g * 64, 512, hipMemcpyDeviceToHost));
        if (smooth) {
            int sel = 0;
            HIPCHK(c, hipMemcpy(&sel, d.smooth_sel + g, 4, hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(smooth + (size_t)i * SS_F, d.smooth + (g * 2 + (sel & 1)) * SS_F, SS_F * 4, hipMemcpyDeviceToHost));
        }
    }
    return SS_OK;
}

extern "C" int ss_get_debug(ss_ctx* c, int s, int frame, int* counts, float* cosd, double* maha, uint8_t* gated,
                            double* cost_a, double* cost_b, int* lists)
{
    if (!c || s < 0 || s >= c->dev.S || frame < 0 || frame >= SS_FMAX) return fail(c, SS_ERR_INVALID, "ss_get_debug: bad stream / frame");
    if (int rcj = chain_join(c, c->stream)) return rcj;               // a detached chain wrote what this call reads
    if (!c->cfg.debug) return fail(c, SS_ERR_INVALID, "ss_get_debug: context created without debug");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    SSDev& d = c->dev;
    const size_t fs = (size_t)frame * d.S + s;                     // [F][S] layout of the group
    const size_t n = (size_t)SS_MAXT * SS_MAXD, o = fs * n;
    if (counts) HIPCHK(c, hipMemcpy(counts, d.dbg_counts + fs * 8, 24, hipMemcpyDeviceToHost));
    if (cosd) HIPCHK(c, hipMemcpy(cosd, d.dbg_cos + o, n * 4, hipMemcpyDeviceToHost));
    if (maha) HIPCHK(c, hipMemcpy(maha, d.dbg_maha + o, n * 8, hipMemcpyDeviceToHost));
    if (gated) HIPCHK(c, hipMemcpy(gated, d.dbg_gated + o, n, hipMemcpyDeviceToHost));
    if (cost_a) HIPCHK(c, hipMemcpy(cost_a, d.dbg_cost_a + o, n * 8, hipMemcpyDeviceToHost));
    if (cost_b) HIPCHK(c, hipMemcpy(cost_b, d.dbg_cost_b + o, n * 8, hipMemcpyDeviceToHost));
    if (lists) HIPCHK(c, hipMemcpy(lists, d.dbg_lists + fs * 4 * SS_MAXT, 4 * SS_MAXT * 4, hipMemcpyDeviceToHost));
    return SS_OK;
}

extern "C" int ss_get_gallery(ss_ctx* c, int s, int track_index, float* rows, int cap_rows, int* count)
{
    if (!c || s < 0 || s >= c->dev.S || track_index < 0 || track_index >= SS_MAXT) return fail(c, SS_ERR_INVALID, "ss_get_gallery: bad argument");
    if (int rcj = chain_join(c, c->stream)) return rcj;               // a detached chain wrote what this call reads
    HIPCHK(c, hipStreamSynchronize(c->stream));
    SSDev& d = c->dev;
    int slot = 0, cnt = 0;
    HIPCHK(c, hipMemcpy(&slot, d.order + (size_t)s * SS_MAXT + track_index, 4, hipMemcpyDeviceToHost));
    const size_t g = (size_t)s * SS_MAXT + slot;
    HIPCHK(c, hipMemcpy(&cnt, d.gal_count + g, 4, hipMemcpyDeviceToHost));
    if (count) *count = cnt;
    if (!rows) return SS_OK;
    if (cnt > cap_rows) return fail(c, SS_ERR_CAPACITY, "ss_get_gallery: cap too small");
    std::vector<float> frag((size_t)SS_NRT * SS_TILE_FLOATS);
    HIPCHK(c, hipMemcpy(frag.data(), d.gallery + g * SS_NRT * SS_TILE_FLOATS, frag.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < cnt; ++b)
        for (int k = 0; k < SS_F; ++k)
            rows[(size_t)b * SS_F + k] = frag[(size_t)(b / SS_TILE) * SS_TILE_FLOATS + ss_frag_index(b % SS_TILE, k)];
    return SS_OK;
}

extern "C" int ss_assoc_inkernel_timing(ss_ctx* c, int enable, double* mean_us, int* launches)
{
    if (!c) return SS_ERR_INVALID;
    if (int rcj = chain_join(c, c->stream)) return rcj;               // a detached chain wrote what this call reads
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long t[4] = { 0, 0, 0, 0 };
    HIPCHK(c, hipMemcpy(t, c->dev.tstamp, sizeof t, hipMemcpyDeviceToHost));
    if (c->inkernel && t[1] > t[0] && t[0] != ~0ull) { t[2] += t[1] - t[0]; t[3] += 1; }      // the last launch is not folded yet
    if (mean_us) *mean_us = t[3] ? (double)t[2] / (double)t[3] / 100.0 : 0.0;                   // 100 MHz ticks
    if (launches) *launches = (int)t[3];
    const unsigned long long arm[4] = { ~0ull, 0, 0, 0 };
    HIPCHK(c, hipMemcpy(c->dev.tstamp, arm, sizeof arm, hipMemcpyHostToDevice));
    c->inkernel = enable < 0 ? 0 : enable > 2 ? 2 : enable;
    return SS_OK;
}

extern "C" int ss_assoc_timeline(ss_ctx* c, long long* out, int n_workgroups)
{
    if (!c || !out || n_workgroups < 1 || n_workgroups > 4096) return SS_ERR_INVALID;
    if (int rcj = chain_join(c, c->stream)) return rcj;               // a detached chain wrote what this call reads
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->dev.timeline, (size_t)n_workgroups * 16 * 8, hipMemcpyDeviceToHost));
    return SS_OK;
}

extern "C" int ss_assoc_timing(ss_ctx* c, int enable, float* mean_ms, int* launches)
{
    if (!c) return SS_ERR_INVALID;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double tot = 0;
    c->ev_ms.clear();
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[i].first, c->ev[i].second));
        tot += ms;
        c->ev_ms.push_back(ms);
    }
    if (mean_ms) *mean_ms = c->ev_used ? (float)(tot / c->ev_used) : 0.f;
    if (launches) *launches = (int)c->ev_used;
    c->ev_used = 0;
    c->timing = enable != 0;
    return SS_OK;
}

extern "C" int ss_assoc_timing_values(ss_ctx* c, float* out_ms, int cap, int* n)
{
    if (!c || !n || cap < 0 || (cap > 0 && !out_ms)) return SS_ERR_INVALID;
    *n = (int)c->ev_ms.size();
    for (int i = 0; i < cap && i < *n; ++i) out_ms[i] = c->ev_ms[i];
    return SS_OK;
}
