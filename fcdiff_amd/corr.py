"""
Front-end named by BASELINE.json's north_star: region x time series -> the edge-major correlation arrays the
fitter takes.  Not part of the reference (fcdiff/fit.py:20-23 starts from correlations); oracle = numpy.corrcoef.
"""
import numpy as np

from . import _lib
from . import util


def correlations(ts, fisher_z=False, ctx=None, as_numpy=True):
    """
    ts : (S, Nreg, T) float64 time series of S subjects.
    Returns (C, S) float64, row c = n(n-1)/2 + m (n > m, util.c_to_nm order), column = subject:
    `b = out[:, healthy]`, `bt = out[:, patients]` can go straight into UnsharedRegionFit.
    fisher_z applies atanh (default off: the model's defaults are on raw correlations, model.py:213, 236).
    """
    import torch
    ctx = ctx if ctx is not None else _lib.Context()
    t = ts if isinstance(ts, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(ts, dtype=np.float64), device=ctx.device)
    t = t.to(device=ctx.device, dtype=torch.float64).contiguous()
    if t.dim() != 3:
        raise ValueError("ts must have shape (S, Nreg, T)")
    (S, Nreg, T) = (int(t.shape[0]), int(t.shape[1]), int(t.shape[2]))
    out = torch.empty((util.N_to_C(Nreg), S), dtype=torch.float64, device=ctx.device)
    ctx.call("fcd_corr_edges", _lib.dptr(t), S, Nreg, T, 1 if fisher_z else 0, _lib.dptr(out), _lib.stream_ptr())
    return out.cpu().numpy() if as_numpy else out
