"""ctypes binding of libetamd.so (C ABI: include/eigentraj.h).

PyTorch is used here only for device memory and streams: every call passes raw
device pointers (``tensor.data_ptr()``) and the current HIP stream across the C
ABI.  There is no CPU fallback: if the shared library is missing, or no HIP
device is present, the operations raise.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
#: ET_LIBETAMD selects another build of the library (kernel-variant A/B runs, tools/build_variant.sh)
LIB_PATH = os.environ.get("ET_LIBETAMD") or os.path.join(_HERE, "libetamd.so")

ET_OK = 0
ET_ERR_BAD_DATA = 5
ABI_VERSION = 3  # include/eigentraj.h ET_ABI_VERSION: the struct mirrors below are for this version
MODE_STATIC, MODE_MOVING, MODE_SPLIT, MODE_IDENTITY = 0, 1, 2, 3
MAX_T, MAX_K, KMEANS_MAX_D, KMEANS_MAX_CLUSTERS = 32, 32, 32, 255
SCENE_MAX_N = 16384  # ET_SCENE_MAX_N

#: every symbol include/eigentraj.h declares (tests check the library exports all of them)
SYMBOLS = [
    "et_abi_version", "et_status_string", "et_compiled_arch", "et_set_option", "et_get_option",
    "et_norm_params", "et_norm_params_from_nrm", "et_normalize", "et_denormalize",
    "et_norm_project", "et_norm_project_pose", "et_scene_project", "et_scene_project_train", "et_wrapper_losses_fwd", "et_wrapper_losses_bwd",
    "et_anchor_reconstruct_fwd", "et_anchor_reconstruct_bwd", "et_anchor_reconstruct_metrics",
    "et_anchor_reconstruct_metrics_pose",
    "et_fit_gram_workspace_bytes", "et_fit_gram", "et_eigh_topk", "et_eigh_topk_batch",
    "et_fit_descriptor_workspace_bytes", "et_fit_descriptor",
    "et_euc_sim", "et_euc_sim_batch", "et_kmeans_partials_len", "et_kmeans_workspace_bytes", "et_kmeans_scan", "et_kmeans_begin",
    "et_kmeans_init_step", "et_kmeans_init_set", "et_kmeans_init_select", "et_kmeans_gather_point", "et_kmeans_init_farthest",
    "et_kmeans_assign_accumulate", "et_kmeans_update", "et_kmeans_joint_done", "et_kmeans_labels_i64", "et_kmeans_fit", "et_kmeans_batch_workspace_bytes", "et_kmeans_fit_batch", "et_kmeans_predict", "et_kmeans_predict_batch",
    "et_kmeans_reforder_workspace_bytes", "et_euc_sim_reforder", "et_kmeans_init_farthest_reforder",
    "et_kmeans_predict_reforder", "et_kmeans_fit_reforder", "et_kmeans_reforder_batch_workspace_bytes",
    "et_kmeans_fit_reforder_batch",
    "et_center_columns", "et_kmeanspp_workspace_bytes", "et_kmeanspp_seed", "et_kmeanspp_batch_workspace_bytes", "et_kmeanspp_seed_batch",
    "et_comm_load", "et_comm_unique_id", "et_comm_init_rank", "et_comm_destroy", "et_comm_info",
    "et_fit_gram_sharded", "et_kmeans_sharded_workspace_bytes", "et_kmeans_init_farthest_sharded", "et_kmeans_fit_sharded",
    "et_kmeans_reforder_shard_block", "et_kmeans_reforder_sharded_workspace_bytes", "et_kmeans_fit_reforder_sharded",
]


class KMeansState(C.Structure):
    """Mirror of ``et_kmeans_state`` (include/eigentraj.h)."""
    _fields_ = [("max_abs_x", C.c_double), ("max_abs_c", C.c_double), ("n_total", C.c_int64), ("frac", C.c_int64),
                ("sim_frac", C.c_int64), ("iter", C.c_int64), ("done", C.c_int64), ("bad_input", C.c_int64),
                ("error", C.c_double), ("inertia", C.c_double), ("fast_ok", C.c_int64),
                ("min_nz_x_bits", C.c_int64)]


class KMeansTiming(C.Structure):
    """Mirror of ``et_kmeans_timing``."""
    _fields_ = [("assign_ms", C.c_double), ("assign_launches", C.c_int64), ("first_assign_ms", C.c_double),
                ("iterations", C.c_int64)]


STATE_BYTES = C.sizeof(KMeansState)
_lib = None


class ETLibraryError(RuntimeError):
    pass


def lib():
    """Load libetamd.so (built in-tree by ``__graft_entry__.build()`` / ``make -C eigentrajectory_amd/csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ETLibraryError(
                f"{LIB_PATH} is missing: build the HIP kernels first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C eigentrajectory_amd/csrc).  eigentrajectory_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        if l.et_abi_version() != ABI_VERSION:
            raise ETLibraryError(f"{LIB_PATH} has ABI version {l.et_abi_version()}, this binding is for {ABI_VERSION}: "
                                 "rebuild the library (make -C eigentrajectory_amd/csrc)")
        l.et_status_string.restype = C.c_char_p
        l.et_compiled_arch.restype = C.c_char_p
        for name in ("et_fit_gram_workspace_bytes", "et_fit_descriptor_workspace_bytes", "et_kmeans_partials_len", "et_kmeans_workspace_bytes",
                     "et_kmeanspp_workspace_bytes", "et_kmeans_sharded_workspace_bytes", "et_kmeans_batch_workspace_bytes",
                     "et_kmeanspp_batch_workspace_bytes", "et_kmeans_reforder_workspace_bytes",
                     "et_kmeans_reforder_batch_workspace_bytes", "et_kmeans_reforder_sharded_workspace_bytes"):
            getattr(l, name).restype = C.c_size_t
        l.et_kmeans_reforder_shard_block.restype = C.c_int64
        _lib = l
        # the library reads nothing from the environment; ET_OPT_<KEY>=value is forwarded once, here (A/B scripts under tools/)
        # every variable is applied; the ones the library rejects are named in ONE warning (never an exception out of
        # whichever unrelated call happened to load the library)
        rejected = []
        for name, value in sorted(os.environ.items()):
            if name.startswith("ET_OPT_") and l.et_set_option(name[len("ET_OPT_"):].lower().encode(), value.encode()) != ET_OK:
                rejected.append(f"{name}={value}")
        if rejected:
            import warnings
            warnings.warn("eigentrajectory_amd: et_set_option rejected " + ", ".join(rejected) +
                          " (unknown key or value; the other ET_OPT_ variables were applied)", RuntimeWarning, stacklevel=2)
    return _lib


def set_option(key: str, value) -> None:
    """et_set_option (include/eigentraj.h, "tuning switches"): measurement aids / test levers; never needed for results."""
    rc = lib().et_set_option(str(key).encode(), str(value).encode())
    if rc != ET_OK:
        raise ValueError(f"et_set_option({key!r}, {value!r}): unknown key or value")


def get_option(key: str) -> str:
    buf = C.create_string_buffer(64)
    if lib().et_get_option(str(key).encode(), buf, C.c_size_t(64)) != ET_OK:
        raise ValueError(f"et_get_option({key!r}): unknown key")
    return buf.value.decode()


class option:
    """``with option("kmeans_packed", 0): ...`` -- set a switch for a block and put the old value back."""

    def __init__(self, key, value):
        self.key, self.value = key, value

    def __enter__(self):
        self.old = get_option(self.key)
        set_option(self.key, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.key, self.old)
        return False


def check(rc: int, what: str):
    if rc != ET_OK:
        msg = lib().et_status_string(rc).decode()
        if rc in (1, ET_ERR_BAD_DATA):
            raise ValueError(f"{what}: {msg}")
        raise ETLibraryError(f"{what}: {msg} (status {rc})")


def require_device(*tensors):
    """Return the HIP device the call runs on; raise when there is none (no CPU path)."""
    for t in tensors:
        if t is not None and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise ETLibraryError("eigentrajectory_amd needs a HIP device (MI355X); no CPU fallback is provided")
    return torch.device("cuda", torch.cuda.current_device())


def on_device(t, device, dtype=torch.float32):
    """Contiguous ``dtype`` copy/view of ``t`` on ``device`` (None passes through)."""
    if t is None:
        return None
    t = t.detach()
    if t.device != device or t.dtype != dtype:
        t = t.to(device=device, dtype=dtype)
    return t.contiguous()


def ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def i64(v):
    return C.c_int64(int(v))


def f32(v):
    return C.c_float(float(v))


# ---------------------------------------------------------------------------------------------------------------
# Lean call path for the scene-size regime (N <= a few dozen pedestrians per forward: everything is launch- and
# host-bound there).  The entry points below get ctypes ``argtypes`` once, so that plain Python ints (``data_ptr()``,
# sizes, the raw stream handle) cross the boundary without per-call wrapper objects.
_P, _I64, _I, _F = C.c_void_p, C.c_int64, C.c_int, C.c_float
_FAST_SIGNATURES = {
    "et_scene_project": [_P, _I64, _I, _I, _P, _P, _I, _F, _P, _P, _P, _P, _P],
    "et_anchor_reconstruct_fwd": [_P, _I64, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P],
    "et_anchor_reconstruct_metrics": [_P, _I64, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P, _P],
    "et_scene_project_train": [_P, _P, _I64, _I, _I, _I, _P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _P, _P],
    "et_wrapper_losses_fwd": [_P, _I64, _I, _I, _I, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _P, _P, _P],
    "et_wrapper_losses_bwd": [_P, _P, _P, _P, _I64, _I, _I, _I, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _P, _P],
    "et_fit_descriptor": [_P, _P, _I64, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P],
}
_fast = {}


def fast(name):
    """The C entry point ``name`` with its argument types declared (cached)."""
    fn = _fast.get(name)
    if fn is None:
        fn = getattr(C.CDLL(LIB_PATH), name)  # a private handle: declaring argtypes must not affect lib()'s users
        fn.argtypes = _FAST_SIGNATURES[name]
        fn.restype = C.c_int
        _fast[name] = fn
    return fn


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream  # the current stream's hipStream_t as an int, ~0.2 us
except AttributeError:  # pragma: no cover - older / newer torch without the private accessor
    def _raw_stream(device_index):
        return torch.cuda.current_stream(device_index).cuda_stream


def raw_stream(device_index):
    return _raw_stream(device_index)
