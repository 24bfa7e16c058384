"""Port of test_fcdiff/test_util.py:5-36 plus the golden fixture G1."""
import numpy as np

import fcdiff_amd
from conftest import load_golden


def test_N_to_C_to_N():
    for N in range(2, 10):
        C = fcdiff_amd.util.N_to_C(N)
        assert fcdiff_amd.util.C_to_N(C) == N


def test_nm_to_c_and_back_lower_triangular_order():
    N = 10
    c = 0
    for n in range(1, N):
        for m in range(0, n):
            assert fcdiff_amd.nm_to_c(n, m) == c
            assert fcdiff_amd.c_to_nm(c) == (n, m)
            c += 1
    assert c == fcdiff_amd.N_to_C(N)


def test_against_reference_fixture():
    g = load_golden("G1_index_maps")
    for N, C, Nb in zip(g["Ns"], g["Cs"], g["N_back"]):
        assert fcdiff_amd.N_to_C(N) == C and fcdiff_amd.util.C_to_N(C) == Nb
    for c in range(45):
        assert fcdiff_amd.c_to_nm(c) == tuple(g["c_to_nm_N10"][c])
    assert isinstance(fcdiff_amd.util.C_to_N(7), float) and fcdiff_amd.util.C_to_N(7) % 1 != 0


def test_upper_to_lower_permutation():
    N = 7
    p = fcdiff_amd.util.upper_to_lower_edge_order(N)
    upper = [(n, m) for n in range(N) for m in range(n + 1, N)]
    for c in range(fcdiff_amd.N_to_C(N)):
        (n, m) = fcdiff_amd.c_to_nm(c)
        assert upper[p[c]] == (m, n)
    assert sorted(p) == list(range(len(upper)))
