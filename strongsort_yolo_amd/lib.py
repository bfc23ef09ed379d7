"""ctypes binding of libstrongsort_hip.so (include/strongsort_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this module
raises.  Device memory is managed with torch tensors; their data_ptr() values cross the C ABI as
plain pointers.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("SS_LIB_PATH") or os.path.join(_HERE, "libstrongsort_hip.so")     # SS_LIB_PATH: a differently built library (kernel experiments)
CSRC = os.path.join(_HERE, "csrc")

SS_OK, SS_ERR_INVALID, SS_ERR_CAPACITY, SS_ERR_HIP, SS_ERR_INFEASIBLE = 0, -1, -2, -3, -4
MAX_TRACKS, MAX_DETS, FEAT_DIM, OUT_COLS = 256, 128, 512, 8
DST_F16, DST_HWC, DST_U8 = 1, 2, 4

EXPORTS = [
    "ss_create", "ss_destroy", "ss_last_error", "ss_set_hip_stream", "ss_reset", "ss_synchronize",
    "ss_upload", "ss_upload_batch", "ss_download", "ss_overlay_set_font", "ss_overlay", "ss_letterbox", "ss_letterbox_batch", "ss_nms", "ss_nms_batch", "ss_nms_set_classes", "ss_crop_norm", "ss_crop_norm_batch",
    "ss_track_update", "ss_track_update_group", "ss_cmc_estimate", "ss_track_set_cmc", "ss_crop_norm_packed", "ss_unpack_feats", "ss_pack_results", "ss_op_set_valid_images", "ss_op_set_option", "ss_track_set_assoc_event", "ss_track_update_host",
    "ss_check_errors", "ss_set_option", "ss_feat_normalize", "ss_ema", "ss_kf_predict", "ss_kf_update", "ss_kf_project", "ss_kf_initiate",
    "ss_gallery_pack", "ss_assoc_cost", "ss_iou_cost", "ss_lsap", "ss_get_tracks", "ss_get_debug",
    "ss_get_gallery", "ss_max_group_frames", "ss_track_join", "ss_stream_create", "ss_stream_destroy", "ss_assoc_timing", "ss_assoc_timing_values", "ss_assoc_inkernel_timing", "ss_assoc_timeline", "ss_op_bias_act_f16", "ss_op_bias_act_place_f16", "ss_op_pointwise_f16", "ss_op_conv3x3_f16", "ss_op_bottleneck_f16", "ss_op_conv_group_f16", "ss_op_head_f16", "ss_op_v8_decode_f16", "ss_op_v8_decode_ext_f16", "ss_op_dwconv3x3_f16", "ss_op_lightconv_f16", "ss_op_osnet_stem_f16", "ss_op_conv0_f16", "ss_op_osnet_streams_f16", "ss_op_osnet_streams_bands", "ss_op_dwtab_bytes", "ss_op_dwtab_f16", "ss_op_gate_apply_f16", "ss_op_osnet_tail_f16", "ss_op_gate_sum_f16", "ss_op_avgpool2_f16", "ss_op_upcat_f16", "ss_op_sppf_pools_f16", "ss_op_psa_attention_f16", "ss_op_osnet_head_f16", "ss_op_maxpool_f16",
    "ss_op32_pointwise", "ss_op32_chains_bands", "ss_op32_chains", "ss_op32_tail", "ss_op32_stem", "ss_op32_stem_u8", "ss_op32_stem_conv1", "ss_op32_head", "ss_op32_set_option", "ss_op32_conv", "ss_op32_conv0", "ss_op32_upcat", "ss_op32_v8_decode", "ss_op32_sppf_pools",
]


class SSError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"strongsort_hip error {code}: {msg}")
        self.code = code


class ss_config(C.Structure):
    _fields_ = [
        ("max_dist", C.c_double), ("max_iou_distance", C.c_double), ("mc_lambda", C.c_double),
        ("gating_threshold", C.c_double), ("gated_cost", C.c_double),
        ("std_weight_position", C.c_double), ("std_weight_velocity", C.c_double),
        ("ema_alpha", C.c_double), ("max_age", C.c_int), ("n_init", C.c_int), ("nn_budget", C.c_int),
        ("n_streams", C.c_int), ("debug", C.c_int),
    ]


class ss_conv_desc(C.Structure):                 # mirrors `typedef struct ss_conv_desc`
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p), ("B", C.c_int), ("H", C.c_int),
                ("W", C.c_int), ("Cin", C.c_int), ("N", C.c_int), ("ksize", C.c_int), ("stride", C.c_int), ("act", C.c_int)]


def build(force: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "strongsort_hip.h"))
    stale = (not os.path.exists(SO_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"] + (["-B"] if force else []))
    return SO_PATH


_lib = None


def load():
    """Load the library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime: it must be the first one mapped into the process, or the
    # library would bind the system libamdhip64 and see no device.
    import torch  # noqa: F401
    if not os.path.exists(SO_PATH):
        raise SSError(SS_ERR_INVALID, f"{SO_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(SO_PATH)
    vp, ip, fp, dp, u8 = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p   # device pointers
    i, f, d = C.c_int, C.c_float, C.c_double
    L.ss_create.argtypes = [C.POINTER(ss_config), i, C.POINTER(vp)]
    L.ss_destroy.argtypes = [vp]; L.ss_destroy.restype = None
    L.ss_last_error.argtypes = [vp]; L.ss_last_error.restype = C.c_char_p
    L.ss_set_hip_stream.argtypes = [vp, vp]
    L.ss_reset.argtypes = [vp, i]
    L.ss_synchronize.argtypes = [vp]
    L.ss_upload.argtypes = [vp, vp, vp, vp, C.c_size_t]
    L.ss_upload_batch.argtypes = [vp, vp, vp, C.POINTER(vp), i, C.c_size_t, i]
    L.ss_download.argtypes = [vp, vp, vp, vp, C.c_size_t]
    L.ss_overlay_set_font.argtypes = [vp, vp]
    L.ss_overlay.argtypes = [vp, vp, vp, i, C.c_longlong, i, i, i, vp, vp, vp]
    L.ss_letterbox.argtypes = [vp, u8, i, i, i, vp, i, i, i, i, i, i, i, i]
    L.ss_nms.argtypes = [vp, fp, i, i, i, f, f, i, f, i, f, f, f, f, f, fp, i, ip, ip]
    L.ss_nms_set_classes.argtypes = [vp, C.POINTER(C.c_int), i]
    L.ss_crop_norm.argtypes = [vp, u8, i, i, i, fp, i, i, ip, vp, i]
    ll = C.c_longlong
    L.ss_letterbox_batch.argtypes = [vp, u8, i, ll, i, i, i, vp, i, i, i, i, i, i, i, i]
    L.ss_nms_batch.argtypes = [vp, fp, i, ll, i, i, i, f, f, i, f, i, fp, fp, i, ll, ip, ll, ip]
    L.ss_crop_norm_batch.argtypes = [vp, u8, i, ll, i, i, i, fp, i, ll, i, ip, vp, i]
    L.ss_crop_norm_packed.argtypes = [vp, u8, i, ll, i, i, i, fp, i, ll, i, ip, ip, vp, i]
    L.ss_unpack_feats.argtypes = [vp, vp, i, ip, ip, i, i, fp, ll]
    L.ss_pack_results.argtypes = [vp, ip, fp, i, i, ip, fp, i, i, fp]
    L.ss_op_set_valid_images.argtypes = [vp, vp, i]
    L.ss_op_set_option.argtypes = [C.c_char_p, i]
    L.ss_track_update.argtypes = [vp, fp, ip, fp, ip, fp, ip]
    L.ss_track_update_group.argtypes = [vp, i, fp, ip, fp, ip, fp, ip]
    L.ss_cmc_estimate.argtypes = [vp, vp, vp, i, C.c_longlong, i, i, i, ip, vp]
    L.ss_track_set_cmc.argtypes = [vp, vp]
    L.ss_track_set_assoc_event.argtypes = [vp, vp]
    L.ss_track_join.argtypes = [vp, vp]
    L.ss_stream_create.argtypes = [vp, i, C.POINTER(vp)]
    L.ss_stream_destroy.argtypes = [vp, vp]
    hf, hi = C.POINTER(C.c_float), C.POINTER(C.c_int)
    L.ss_track_update_host.argtypes = [vp, i, hf, i, hf, i, i, hf, i, hi]
    L.ss_check_errors.argtypes = [vp]
    L.ss_set_option.argtypes = [vp, C.c_char_p, i]
    L.ss_feat_normalize.argtypes = [vp, fp, i, fp]
    L.ss_ema.argtypes = [vp, fp, fp, i, fp]
    L.ss_kf_predict.argtypes = [vp, dp, dp, i]
    L.ss_kf_update.argtypes = [vp, dp, dp, dp, dp, i]
    L.ss_kf_initiate.argtypes = [vp, dp, i, dp, dp]
    L.ss_kf_project.argtypes = [vp, dp, dp, dp, i, dp, dp]
    L.ss_gallery_pack.argtypes = [vp, fp, i, i, fp]
    L.ss_assoc_cost.argtypes = [vp, fp, ip, i, fp, i, dp, dp, dp, dp, fp, dp, u8]
    L.ss_iou_cost.argtypes = [vp, dp, i, dp, i, dp]
    L.ss_lsap.argtypes = [vp, dp, i, i, ip]
    hd, hu8 = C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    L.ss_get_tracks.argtypes = [vp, i, i, hi, hi, hi, hi, hi, hi, hi, hi, hf, hd, hd, hf, hi]
    L.ss_get_debug.argtypes = [vp, i, i, hi, hf, hd, hu8, hd, hd, hi]
    L.ss_get_gallery.argtypes = [vp, i, i, hf, i, hi]
    L.ss_assoc_timing.argtypes = [vp, i, hf, hi]
    L.ss_assoc_timing_values.argtypes = [vp, hf, i, hi]
    L.ss_assoc_inkernel_timing.argtypes = [vp, i, C.POINTER(C.c_double), hi]
    L.ss_assoc_timeline.argtypes = [vp, C.POINTER(C.c_longlong), i]
    L.ss_op_bias_act_f16.argtypes = [vp, vp, vp, vp, C.c_longlong, i, i]
    L.ss_op_dwconv3x3_f16.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i]
    L.ss_op_bias_act_place_f16.argtypes = [vp, vp, vp, vp, C.c_longlong, i, i, i, vp, i, vp, i, i]
    L.ss_op_pointwise_f16.argtypes = [vp, vp, vp, vp, vp, C.c_longlong, i, i, i, i, vp, i, vp, i, i]
    L.ss_op_conv3x3_f16.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, vp, i, vp, i, i]
    L.ss_op_bottleneck_f16.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp, i, vp]
    L.ss_op_conv_group_f16.argtypes = [vp, i, C.POINTER(ss_conv_desc)]
    pv = C.POINTER(vp)
    L.ss_op_head_f16.argtypes = [vp, vp, pv, pv, pv, pv, pv, pv, pv, C.POINTER(C.c_int), i, i, i, i, i]
    L.ss_op_v8_decode_f16.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), hi, hi, hi, i, i, vp]
    L.ss_op_v8_decode_ext_f16.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i, i, i, hi, hi, hi, i, i, i, vp]
    L.ss_op_lightconv_f16.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i]
    L.ss_op_conv0_f16.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i]
    L.ss_op_osnet_stem_f16.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp, vp, vp]
    L.ss_op_osnet_streams_f16.argtypes = [vp, vp, vp, vp, C.POINTER(vp), vp, i, i, i, i]
    L.ss_op_dwtab_bytes.argtypes = [i, i]
    L.ss_op_dwtab_bytes.restype = C.c_longlong
    L.ss_op_dwtab_f16.argtypes = [vp, vp, vp, i, i, vp]
    L.ss_op_osnet_streams_bands.argtypes = [i, i, i, i]
    L.ss_op_gate_apply_f16.argtypes = [vp, C.POINTER(vp), i, vp, vp, vp, vp, vp, i, f, vp, i, i, i, i]
    L.ss_op_osnet_tail_f16.argtypes = [vp, C.POINTER(vp), vp, i, f, vp, vp, vp, vp, i, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i]
    L.ss_op_upcat_f16.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i]
    L.ss_op_sppf_pools_f16.argtypes = [vp, vp, vp, i, i, i, i]
    L.ss_op_psa_attention_f16.argtypes = [vp, vp, vp, vp, i, i, i, C.c_float]
    L.ss_op_osnet_head_f16.argtypes = [vp, vp, vp, vp, vp, i, i, i, i]
    L.ss_op_avgpool2_f16.argtypes = [vp, vp, vp, i, i, i, i]
    L.ss_op_maxpool_f16.argtypes = [vp, vp, vp, i, i, i, i, i, i, i]
    L.ss_op_gate_sum_f16.argtypes = [vp, C.POINTER(vp), i, vp, vp, vp, vp, vp, vp, i, i, i, i]
    L.ss_op32_pointwise.argtypes = [vp, vp, vp, vp, vp, vp, ll, i, i, i, vp, i]
    L.ss_op32_chains_bands.argtypes = [i, i, i, i]
    L.ss_op32_chains.argtypes = [vp, vp, vp, vp, vp, C.POINTER(vp), vp, i, i, i, i, vp]
    L.ss_op32_tail.argtypes = [vp, C.POINTER(vp), vp, i, vp, vp, vp, vp, i, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, vp]
    L.ss_op32_stem.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp]
    L.ss_op32_stem_u8.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp]
    L.ss_op32_stem_conv1.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, vp, i, i, i, vp]
    L.ss_op32_head.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp]
    L.ss_op32_set_option.argtypes = [C.c_char_p, i]
    L.ss_op32_conv.argtypes = [vp, vp, i, vp, vp, vp, i, vp, i, i, i, i, i, i, i, i, i]
    L.ss_op32_conv0.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i]
    L.ss_op32_upcat.argtypes = [vp, vp, i, i, vp, i, i, vp, i, i, i, i]
    L.ss_op32_sppf_pools.argtypes = [vp, vp, i, vp, i, i, i, i]
    L.ss_op32_v8_decode.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i, i, i, C.POINTER(i), C.POINTER(i), C.POINTER(i), i, i, i, vp]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("ss_destroy", "ss_last_error"):
            fn.restype = C.c_int
    _lib = L
    return L


def check(ctx, rc):
    if rc != SS_OK:
        msg = load().ss_last_error(ctx)
        raise SSError(rc, msg.decode() if msg else "?")


def make_config(cfg, n_streams=1, debug=False) -> ss_config:
    return ss_config(cfg.max_dist, cfg.max_iou_distance, cfg.mc_lambda, cfg.gating_threshold, cfg.gated_cost,
                     cfg.std_weight_position, cfg.std_weight_velocity, cfg.ema_alpha, cfg.max_age, cfg.n_init,
                     cfg.nn_budget, n_streams, int(debug))
