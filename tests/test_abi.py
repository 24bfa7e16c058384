"""
The C-ABI library loads without a GPU and exports every symbol include/fcdiff_hip.h declares (no compute
calls here).  Also the host-only entry points (index maps, error strings, state sizes).
"""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from fcdiff_amd import _lib

HEADER = os.path.join(ROOT, "include", "fcdiff_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fcd_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_binding_binds():
    names = declared_functions()
    assert len(names) >= 25
    assert set(names) == set(_lib.SIGNATURES.keys())


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(raw, name), "libfcdiff_hip.so does not export %s" % name
    lib = _lib.load()
    assert lib.fcd_abi_version() == _lib.ABI_VERSION == 4


def test_host_index_maps_match_util():
    """fcdiff/util.py through the C ABI (test_fcdiff/test_util.py)."""
    lib = _lib.load()
    from fcdiff_amd import util
    for N in range(2, 10):
        Cn = lib.fcd_N_to_C(N)
        assert Cn == util.N_to_C(N) and lib.fcd_C_to_N(Cn) == N
    c = 0
    for n in range(1, 10):
        for m in range(0, n):
            assert lib.fcd_nm_to_c(n, m) == c == util.nm_to_c(n, m)
            (a, b) = (C.c_int64(), C.c_int64())
            assert lib.fcd_c_to_nm(c, C.byref(a), C.byref(b)) == 0
            assert (a.value, b.value) == (n, m) == util.c_to_nm(c)
            c += 1
    for bad in (2, 4, 5, 7, 19901):
        assert lib.fcd_C_to_N(bad) == _lib.FCD_ERR_SHAPE          # fit.py:62-65
    # large sizes: the float sqrt inverse needs its integer correction
    for N in (400, 4097, 46340):
        Cn = lib.fcd_N_to_C(N)
        assert lib.fcd_C_to_N(Cn) == N
        (a, b) = (C.c_int64(), C.c_int64())
        lib.fcd_c_to_nm(Cn - 1, C.byref(a), C.byref(b))
        assert (a.value, b.value) == (N - 1, N - 2)


def test_error_strings_and_state_size():
    lib = _lib.load()
    assert b"triangular" in lib.fcd_strerror(_lib.FCD_ERR_SHAPE)
    assert lib.fcd_strerror(0) == b"ok"
    (fb, rb) = (C.c_size_t(), C.c_size_t())
    assert lib.fcd_gibbs_state_size(200, 50, 1024, C.byref(fb), C.byref(rb)) == 0
    assert fb.value == 16 * 19900 * 64 and rb.value == 16 * 200 * 50 * 8
    assert lib.fcd_gibbs_state_size(200, 50, 1000, C.byref(fb), C.byref(rb)) == 0
    assert fb.value == 16 * 19900 * 64
    assert lib.fcd_gibbs_state_size(1, 50, 64, C.byref(fb), C.byref(rb)) == _lib.FCD_ERR_ARG


def test_error_codes_of_header_and_binding_agree():
    """Every FCD_ERR_* of include/fcdiff_hip.h has the same value in the binding and a message of its own."""
    import os
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fcdiff_hip.h")).read()
    codes = dict(re.findall(r"#define (FCD_ERR_[A-Z]+) \((-\d+)\)", text))
    assert set(codes) == {"FCD_ERR_ARG", "FCD_ERR_SHAPE", "FCD_ERR_UNSUPPORTED", "FCD_ERR_INDEX", "FCD_ERR_DEVICE", "FCD_ERR_COMM"}
    lib = _lib.load()
    seen = set()
    for (name, val) in codes.items():
        assert getattr(_lib, name) == int(val)
        msg = lib.fcd_strerror(int(val))
        assert msg and msg != b"unknown fcdiff_hip error" and msg not in seen
        seen.add(msg)


def test_fitter_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    import fcdiff_amd
    fit = fcdiff_amd.fit.UnsharedRegionFit()
    fit.model = fcdiff_amd.UnsharedRegionModel()
    fit.b, fit.bt = np.zeros((3, 2)), np.zeros((3, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fit.run()
