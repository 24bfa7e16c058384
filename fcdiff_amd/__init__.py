"""
fcdiff_amd -- the fcdiff fit path on MI355X (gfx950).

Same public names as the reference package (fcdiff/__init__.py:1-5):
    UnsharedRegionModel, fit (module), N_to_C, nm_to_c, c_to_nm
The fitter runs hand-written HIP kernels through the C ABI of include/fcdiff_hip.h; there is no CPU
fallback: using the fitter without libfcdiff_hip.so or without a GPU raises.
"""
from .model import UnsharedRegionModel
from . import fit
from . import util
from .util import N_to_C, nm_to_c, c_to_nm

__all__ = ["UnsharedRegionModel", "fit", "util", "N_to_C", "nm_to_c", "c_to_nm"]
