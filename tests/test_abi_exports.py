"""The C-ABI library loads (no GPU needed) and exports every symbol include/strongsort_hip.h declares."""
import ctypes
import os
import re

from strongsort_yolo_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "strongsort_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ss_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib.build()
    L = ctypes.CDLL(lib.SO_PATH)
    names = _declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(names) == sorted(lib.EXPORTS)          # the ctypes binding covers the whole header


def test_config_struct_matches_header():
    src = open(os.path.join(ROOT, "include", "strongsort_hip.h")).read()
    body = src[src.index("typedef struct ss_config {"):src.index("} ss_config;")]
    fields = re.findall(r"^\s*(double|float|int)\s+(\w+);", body, flags=re.M)
    ctype = {"double": ctypes.c_double, "float": ctypes.c_float, "int": ctypes.c_int}
    assert [(n, ctype[t]) for t, n in fields] == [(n, t) for n, t in lib.ss_config._fields_]


def test_error_without_gpu_is_loud():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from strongsort_yolo_amd.engine import TrackerEngine
    with pytest.raises(lib.SSError):
        TrackerEngine()


def test_host_side_entry_points_without_a_gpu():
    """Entry points that only compute on the host or validate arguments: callable on a CPU-only box."""
    lib.build()
    L = lib.load()
    assert L.ss_max_group_frames() == 32
    # band count of the LightConv chain launches: LDS form (16-row bands) below 96 images, the row-stream form sizes its bands
    # for one round of waves (32-wide: 3072 waves = images x bands x 2 chain groups), at least 8 rows per band
    assert L.ss_op_osnet_streams_bands(32, 64, 32, 16) == 4
    assert L.ss_op_osnet_streams_bands(512, 64, 32, 16) == 3
    assert L.ss_op_osnet_streams_bands(1024, 64, 32, 16) == 1
    assert L.ss_op_osnet_streams_bands(128, 64, 32, 16) == 8
    assert L.ss_op_osnet_streams_bands(512, 32, 16, 24) == 2
    assert L.ss_op_osnet_streams_bands(512, 16, 8, 32) == 2            # 8-wide maps: two images per wave, 256 units
    assert L.ss_op_osnet_streams_bands(32, 16, 8, 32) == 1             # below 96 images: the LDS form, whole 16-row bands
    assert L.ss_op_osnet_streams_bands(0, 64, 32, 16) < 0
    assert L.ss_op_set_valid_images(None, None, 0) == 0 and L.ss_op_set_valid_images(None, None, 5) == 0       # NULL count: off
    assert L.ss_op_set_option(b"pw_splitk", 1) == 0 and L.ss_op_set_option(b"no_such_switch", 1) == lib.SS_ERR_INVALID
    assert L.ss_op_conv_group_f16(None, 0, None) < 0                    # n out of range / no descriptors
    d = (lib.ss_conv_desc * 1)()
    assert L.ss_op_conv_group_f16(None, 1, d) < 0                       # null tensors
    assert L.ss_op_upcat_f16(None, None, None, None, 1, 4, 4, 8, 8, 1) < 0
    assert L.ss_op_sppf_pools_f16(None, None, None, 1, 40, 40, 8) < 0   # H*W > 1024 (and null tensors)
    assert L.ss_op_conv0_f16(None, None, None, None, None, 1, 8, 100, 16, 2) < 0
