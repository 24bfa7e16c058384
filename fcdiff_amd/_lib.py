"""
ctypes binding of libfcdiff_hip.so (C ABI: include/fcdiff_hip.h).

The library is built in-tree by `__graft_entry__.build()` (or `make -C fcdiff_amd/csrc`).  There is no
CPU fallback anywhere in this package: a missing library, a missing GPU or a non-zero return code raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FCDIFF_HIP_LIB: load another build of the same ABI (the ablation variant of csrc/Makefile, for kernel timing only)
LIB_PATH = os.environ.get("FCDIFF_HIP_LIB") or os.path.join(_HERE, "libfcdiff_hip.so")

FCD_OK = 0
FCD_ERR_ARG = -1
FCD_ERR_SHAPE = -2
FCD_ERR_UNSUPPORTED = -3
FCD_ERR_INDEX = -4
FCD_ERR_DEVICE = -5
FCD_ERR_COMM = -6
FCD_COMM_ID_BYTES = 128
EDGE_REFERENCE = 0
EDGE_SYMMETRIC = 1
EDGE_MODES = {"reference": EDGE_REFERENCE, "symmetric": EDGE_SYMMETRIC}
ABI_VERSION = 4

_p = C.c_void_p
_i64 = C.c_int64
_u64 = C.c_uint64
_int = C.c_int
_dbl = C.c_double

# name -> (restype, argtypes); one entry per declaration of include/fcdiff_hip.h
SIGNATURES = {
    "fcd_abi_version": (_int, []),
    "fcd_strerror": (C.c_char_p, [_int]),
    "fcd_last_message": (C.c_char_p, [_p]),
    "fcd_ctx_create": (_int, [C.POINTER(_p)]),
    "fcd_ctx_destroy": (_int, [_p]),
    "fcd_ctx_reserve": (_int, [_p, _i64, _i64, _i64]),
    "fcd_ctx_set_knob": (_int, [_p, C.c_char_p, _dbl]),
    "fcd_ctx_stat": (_int, [_p, C.c_char_p, C.POINTER(_i64)]),
    "fcd_ctx_check": (_int, [_p]),
    "fcd_ctx_clear_error": (_int, [_p]),
    "fcd_comm_load": (_int, [_p, C.c_char_p]),
    "fcd_comm_unique_id": (_int, [_p, _p]),
    "fcd_comm_init": (_int, [_p, _p, _int, _int]),
    "fcd_comm_destroy": (_int, [_p]),
    "fcd_allreduce_stats": (_int, [_p, _p, _p]),
    "fcd_prof_enable": (_int, [_p, _int]),
    "fcd_prof_collect": (_int, [_p, _int, C.POINTER(_dbl), C.POINTER(_i64)]),
    "fcd_N_to_C": (_i64, [_i64]),
    "fcd_C_to_N": (_i64, [_i64]),
    "fcd_nm_to_c": (_i64, [_i64, _i64]),
    "fcd_c_to_nm": (_int, [_i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "fcd_hyper_set": (_int, [_p, _p, C.POINTER(_dbl), C.POINTER(_dbl), _p]),
    "fcd_lik_tables": (_int, [_p, _p, _p, _i64, _i64, _i64, C.POINTER(_dbl), _p, _p, _p, _p, _p]),
    "fcd_model_sample": (_int, [_p, C.POINTER(_dbl), _i64, _i64, _i64, _u64, _p, _p, _p, _p, _p, _p, _p]),
    "fcd_corr_edges": (_int, [_p, _p, _i64, _i64, _i64, _int, _p, _p]),
    "fcd_vb_update_qF": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _p, _p]),
    "fcd_vb_update_qR": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p, _p]),
    "fcd_vb_energy": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p]),
    "fcd_vb_theta_step": (_int, [_p, _p, _p, _i64, _i64, _p, _p, _p]),
    "fcd_theta_sub_weights_vb": (_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    "fcd_gibbs_pair_counts": (_int, [_p, _p, _p, _i64, _i64, _i64, _int, _p, _p]),
    "fcd_theta_sub_objective": (_int, [_p, _p, _p, _i64, _i64, C.POINTER(_dbl), _p, _p]),
    "fcd_theta_full_objective": (_int, [_p, _p, _p, _p, _i64, _i64, _i64, C.POINTER(_dbl), _p, _p]),
    "fcd_gibbs_state_size": (_int, [_i64, _i64, _i64, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "fcd_gibbs_init": (_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _u64, _dbl, _p]),
    "fcd_gibbs_edge_tables": (_int, [_p, _p, _i64, _i64, _p, _p]),
    "fcd_gibbs_f_step": (_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _u64, _i64, _p]),
    "fcd_gibbs_region_tables": (_int, [_p, _p, _i64, _i64, _int, _p, _p]),
    "fcd_gibbs_r_step": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _u64, _i64, _int, _p]),
    "fcd_gibbs_sweeps": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _u64, _i64, _i64, _int, _p, _p]),
    "fcd_gibbs_stats": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p]),
    "fcd_gibbs_mstep": (_int, [_p, _p, _i64, _i64, _p, _p]),
    "fcd_gibbs_accumulate": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "fcd_gibbs_tally": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p]),
    "fcd_gibbs_run": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _u64, _i64, _i64, _int, _i64, _i64, _p, _p,
                            _p, _p]),
    "fcd_gibbs_logjoint": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p]),
    "fcd_gibbs_chain_rsum": (_int, [_p, _p, _i64, _i64, _i64, _p, _p]),
    "fcd_gibbs_conditionals": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _int, _p, _p, _p]),
    "fcd_gibbs_export_state": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "fcd_gibbs_import_state": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "fcd_philox_uniforms": (_int, [_p, _p, _i64, _u64, _p, _p]),
}

_lib = None


class FcdiffHipError(RuntimeError):
    pass


def load():
    """dlopen the library and bind every symbol of the header (raises when it is not built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "fcdiff_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C fcdiff_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    # torch first: its wheel carries the HIP runtime (libamdhip64.so.7) every tensor and stream of this
    # process lives in; loaded afterwards, our library binds to that same runtime by soname.  The other
    # order leaves two runtimes in the process and ours then sees no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    v = lib.fcd_abi_version()
    if v != ABI_VERSION:
        raise ImportError("fcdiff_amd: libfcdiff_hip.so has ABI %d, this package expects %d" % (v, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, ctx=None):
    """Map a return code to the exception the reference's NumPy path would have raised."""
    if rc == FCD_OK:
        return
    lib = load()
    msg = lib.fcd_last_message(ctx).decode() if ctx else ""
    base = lib.fcd_strerror(rc).decode()
    text = msg if msg else base
    if rc == FCD_ERR_SHAPE or rc == FCD_ERR_ARG:
        raise ValueError(text)
    if rc == FCD_ERR_INDEX:
        raise IndexError(text)
    if rc == FCD_ERR_UNSUPPORTED:
        raise NotImplementedError(text)
    if rc == FCD_ERR_DEVICE or rc == FCD_ERR_COMM:
        raise FcdiffHipError(text)
    raise FcdiffHipError("HIP error %d: %s %s" % (rc, base, msg))


class Context(object):
    """Owns one fcd_ctx on the current torch CUDA (HIP) device."""

    def __init__(self):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("fcdiff_amd needs an MI355X: torch.cuda.is_available() is False "
                               "(the fit path has no CPU fallback)")
        self.lib = load()
        self.handle = _p()
        check(self.lib.fcd_ctx_create(C.byref(self.handle)))
        self.device = torch.device("cuda", torch.cuda.current_device())

    def close(self):
        if self.handle:
            self.lib.fcd_ctx_destroy(self.handle)
            self.handle = _p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, *args):
        check(getattr(self.lib, name)(self.handle, *args), self.handle)

    def set_knob(self, name, value):
        """Tuning / test knob of include/fcdiff_hip.h (fcd_ctx_set_knob); 0 restores the default."""
        self.call("fcd_ctx_set_knob", name.encode(), float(value))

    def stat(self, name):
        v = _i64()
        self.call("fcd_ctx_stat", name.encode(), C.byref(v))
        return v.value

    def check_device(self):
        """
        Raise FcdiffHipError if a kernel of this context gave up a device-side wait (pipelined r pass).  The word is
        written by the device, so call this after something that synchronises with the sweeps (a .cpu() read, a
        torch.cuda.synchronize()): GibbsEngine does after every host read of chain state or counts.
        """
        self.call("fcd_ctx_check")

    def clear_error(self):
        self.call("fcd_ctx_clear_error")

    def attach_comm(self, group=None):
        """
        Give this context an RCCL communicator over the ranks of the torch.distributed `group` (default: the world):
        from then on fcd_gibbs_run pools the counts of every M-step over those ranks with ncclAllReduce on the stream of
        the sweep kernels (include/fcdiff_hip.h, fcd_comm_*).  Collective: every rank of the group calls it, before any
        sweep.  The unique id travels through the group itself (a 128-byte broadcast from its rank 0); RCCL's entry
        points come from the librccl torch has loaded.  A group of one rank is legal.
        """
        import os
        import numpy as np
        import torch
        import torch.distributed as dist
        if self.stat("comm_world"):
            return
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("attach_comm needs an initialised torch.distributed process group")
        (world, rank) = (dist.get_world_size(group), dist.get_rank(group))
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.call("fcd_comm_load", path.encode() if os.path.exists(path) else b"")
        ident = np.zeros(FCD_COMM_ID_BYTES, dtype=np.uint8)
        if rank == 0:
            self.call("fcd_comm_unique_id", ident.ctypes.data_as(_p))
        on_gpu = dist.get_backend(group) == "nccl"
        t = torch.from_numpy(ident).to(self.device) if on_gpu else torch.from_numpy(ident)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(t, src=src, group=group)
        ident = np.ascontiguousarray(t.cpu().numpy())
        torch.cuda.synchronize()
        self.call("fcd_comm_init", ident.ctypes.data_as(_p), int(world), int(rank))

    def detach_comm(self):
        self.call("fcd_comm_destroy")

    PROF_SLOTS = {"lik_kernel": 0, "gibbs_f_pair_kernel": 1, "gibbs_r_step_kernel": 2, "pack_f_kernel": 3}

    def prof_enable(self, on=True):
        self.call("fcd_prof_enable", 1 if on else 0)

    def prof_collect(self):
        """{kernel: (total_ms, launches)} of the event pairs recorded since the last collect."""
        out = {}
        for name, slot in self.PROF_SLOTS.items():
            (ms, n) = (_dbl(), _i64())
            self.call("fcd_prof_collect", slot, C.byref(ms), C.byref(n))
            out[name] = (ms.value, n.value)
        return out


def stream_ptr():
    import torch
    return _p(torch.cuda.current_stream().cuda_stream)


def dptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return _p(0) if t is None else _p(t.data_ptr())


def dbl_array(values):
    import numpy as np
    a = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
    return (a.ctypes.data_as(C.POINTER(_dbl)), a)
