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
