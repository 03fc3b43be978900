"""The C-ABI library loads on a machine without a GPU, exports every symbol that
include/ycnr_als.h declares, and fails loudly (never silently falls back) when no HIP
device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import ycnr_als
from ycnr_als import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ycnr_als.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ycnr_[A-Za-z0-9_]+)\s*\(", src)))


def test_header_and_binding_list_agree():
    assert declared_functions() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    for name in declared_functions():
        assert hasattr(L, name), f"{name} is declared in include/ycnr_als.h but not exported"
    assert L.ycnr_version() == _lib.ABI_VERSION == 4


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.Options) == 56
    assert C.sizeof(_lib.StepInfo) == 136
    src = open(HEADER).read()
    for field, _ in _lib.Options._fields_:
        assert re.search(r"\b%s;" % field, src), field
    for field, _ in _lib.StepInfo._fields_:
        assert re.search(r"\b%s;" % field, src), field


def test_no_silent_cpu_fallback():
    """Without a GPU every compute entry point must fail with an error, not compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu suite")
    L = _lib.load()
    assert L.ycnr_device_count() == _lib.ERR_HIP
    assert b"hip" in L.ycnr_last_error().lower()
    with pytest.raises(ycnr_als.YcnrError) as e:
        ycnr_als.AlsDevice(8, 4, 4)
    assert e.value.code == _lib.ERR_HIP
    rows = np.array([1, 0, 1], np.int32)
    U = np.zeros((1, 4), np.float32)
    with pytest.raises(ycnr_als.YcnrError):
        ycnr_als.als_calc_portion(0.05, 4, rows, np.zeros(1, np.int32), np.ones(1, np.float32),
                                  np.ones((1, 4), np.float32), U)
    assert not U.any()


def test_argument_validation_happens_before_the_device_is_touched():
    rows = np.array([1, 7, 1], np.int32)  # rowId 7 outside a 1-row matrix
    with pytest.raises(ycnr_als.YcnrError) as e:
        ycnr_als.als_calc_portion(0.05, 4, rows, np.zeros(1, np.int32), np.ones(1, np.float32),
                                  np.ones((1, 4), np.float32), np.zeros((1, 4), np.float32))
    assert e.value.code == _lib.ERR_INVALID
    with pytest.raises(TypeError, match="invalid type!"):
        ycnr_als.als_calc_portion(0.05, 4, rows, np.zeros(1, np.int32), np.ones(1, np.int16),
                                  np.ones((1, 4), np.float32), np.zeros((1, 4), np.float32))
    with pytest.raises(ycnr_als.YcnrError) as e:
        # (any factorsCount up to 4096 is served; beyond that the right-hand side would no longer fit a CU's LDS)
        ycnr_als.als_calc_portion(0.05, 5000, np.array([1, 0, 1], np.int32), np.zeros(1, np.int32),
                                  np.ones(1, np.float32), np.ones((1, 5000), np.float32), np.zeros((1, 5000), np.float32))
    assert e.value.code == _lib.ERR_UNSUPPORTED
