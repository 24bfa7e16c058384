#!/usr/bin/env python3
"""K_corr alone at cfg3's shape (S=100, Nreg=200, T=1200), N calls: for rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fcdiff_amd  # noqa: E402
from fcdiff_amd.corr import correlations  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    (S, N, T) = (100, 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1200)
    from fcdiff_amd import _lib
    ctx = _lib.Context()
    ts = torch.randn((S, N, T), dtype=torch.float64, device="cuda")
    correlations(ts, ctx=ctx, as_numpy=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        correlations(ts, ctx=ctx, as_numpy=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    flops = 2.0 * S * (N * (N + 1) / 2) * T
    print("K_corr (S=%d, Nreg=%d, T=%d): %.1f us per call, %.2f TFLOP/s of the lower triangle" % (S, N, T, ms * 1e3, flops / (ms * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
