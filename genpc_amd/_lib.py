"""ctypes binding of libgenpc_hip.so (C ABI: include/genpc_hip.h).

The library is the product; there is NO fallback.  If it is missing, was built
for another ABI version or lacks a symbol, importing this module raises.
"""
import ctypes
import os

import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first; ours binds to it)

_HERE = os.path.dirname(os.path.abspath(__file__))
# GENPC_LIB: an alternative build of the same library (kernel experiments only)
LIB_PATH = os.environ.get("GENPC_LIB") or os.path.join(_HERE, "lib", "libgenpc_hip.so")
ABI_VERSION = 16

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_d = ctypes.c_double

# name -> (restype, argtypes); must list every symbol include/genpc_hip.h declares
# (tests/test_abi.py parses the header and checks this table and the .so).
SIGNATURES = {
    "genpc_abi_version": (_i, []),
    "genpc_last_error": (ctypes.c_char_p, []),
    "genpc_set_arith": (_i, [_i]),
    "genpc_set_arith_thread": (_i, [_i]),
    "genpc_get_arith": (_i, []),
    "genpc_thread_state_export": (_i, [_vp]),
    "genpc_thread_state_import": (_i, [_vp]),
    "genpc_release_workspace": (_i, []),
    "genpc_tune_table": (_i, [_vp, _i]),
    "genpc_nn_tune": (_i, [_i, _i]),
    "genpc_nn_stats": (_i, [_vp, _i, _vp]),
    "genpc_nn_duplicate_mask": (_i, [_i, _i, _vp, _vp, _vp]),
    "genpc_nn_profile": (ctypes.c_float, [_i]),
    "genpc_chamfer_forward": (_i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "genpc_nm_distance": (_i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "genpc_nm_distance_within": (_i, [_i, _i, _vp, _i, _vp, _f, _vp, _vp, _vp]),
    "genpc_chamfer_backward": (_i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "genpc_emd_forward": (_i, [_i, _i, _i] + [_vp] * 14 + [_f, _i, _vp]),
    "genpc_emd_tune": (_i, [_i, _i]),
    "genpc_emd_stats": (_i, [_vp, _i, _vp]),
    "genpc_emd_status": (_i, [_i, _vp]),
    "genpc_emd_contended": (_i, []),
    "genpc_emd_calc_dist": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "genpc_emd_backward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "genpc_get_uvs": (_i, [_i, _i, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp]),
    "genpc_uv_to_pixels": (_i, [_i, _vp, _f, _i, _vp, _vp]),
    "genpc_paint_pixels": (_i, [_i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "genpc_gather_colors": (_i, [_i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "genpc_hpr_visibility": (_i, [_i, _i, _vp, _vp, _d, _vp, _vp, _vp, _vp]),
    "genpc_hpr_best_view_counts": (_i, [_i, _i, _vp, _vp, _d, _vp, _vp, _vp, _vp, _vp]),
    "genpc_zbuffer_visibility": (_i, [_i, _i, _vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    "genpc_pose_transform": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "genpc_pose_cd_grad": (_i, [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp]),
    "genpc_splat_image": (_i, [_i, _vp, _vp, _f, _i, _vp, _vp]),
    "genpc_mask_loss": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "genpc_pose_loss_grad": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i, _vp, _vp, _vp]),
    "genpc_pose_tune": (_i, [_i]),
    "genpc_pose_dual": (_i, [_i]),
    "genpc_render_tune": (_i, [_i]),
    "genpc_pose_optimize_batch": (_i, [_i, _i, _vp, _vp, _i, _vp, _vp, _f, _i, _i, _f, _i, _f, _vp, _vp, _vp, _vp]),
    "genpc_pose_optimize_cd": (_i, [_i, _vp, _i, _vp, _f, _i, _i, _vp, _vp, _vp, _vp]),
    "genpc_pose_optimize_cd_batch": (_i, [_i, _i, _vp, _i, _vp, _f, _i, _i, _vp, _vp, _vp, _vp]),
    "genpc_icp_batch": (_i, [_i, _i, _vp, _i, _vp, _d, _vp, _i, _d, _d, _vp, _vp, _vp]),
    "genpc_scale_search_scores": (_i, [_i, _i, _vp, _i, _vp, _vp, _f, _vp, _vp]),
    "genpc_voxel_down_sample": (_i, [_i, _vp, _vp, _d, _vp, _vp, _vp, _vp]),
    "genpc_mfma_f16_probe": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "genpc_list_code_probe": (_i, [ctypes.c_longlong, _vp, _vp, _vp, _vp]),
    "genpc_fastdiv_probe": (_i, [ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "genpc_fps_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "genpc_fps_defer": (_i, [_i]),
    "genpc_streams_prepare": (_i, [_vp]),
    "genpc_fps_deferred_check": (_i, [_vp]),
    "genpc_fps_tune": (_i, [_i]),
    "genpc_fps_stats": (_i, [_i, _vp, _vp]),
    "genpc_fps": (_i, [_i, _i, _vp, _i, _vp, _vp]),
    "genpc_knn_mean_distance": (_i, [_i, _vp, _i, _vp, _vp]),
}


class GenpcLibraryError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise GenpcLibraryError(
            "libgenpc_hip.so not found at %s -- run `python -m genpc_amd.build` "
            "(hipcc, gfx950).  There is no CPU or PyTorch fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise GenpcLibraryError("libgenpc_hip.so lacks symbol %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    got = lib.genpc_abi_version()
    if got != ABI_VERSION:
        raise GenpcLibraryError("libgenpc_hip.so ABI %d != expected %d (rebuild)" % (got, ABI_VERSION))
    return lib


lib = _load()


def tune_table():
    """The library's tuning switches consulted so far, with their values and defaults (GENPC_* environment variables)."""
    n = lib.genpc_tune_table(None, 0)
    buf = ctypes.create_string_buffer(n)
    lib.genpc_tune_table(ctypes.cast(buf, ctypes.c_void_p), n)
    return buf.value.decode("utf-8", "replace")


def last_error():
    return lib.genpc_last_error().decode("utf-8", "replace")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return _vp(0) if t is None else _vp(t.data_ptr())


def stream_of(t):
    """hipStream_t of torch's current stream on t's device, as void*."""
    return _vp(torch.cuda.current_stream(t.device).cuda_stream)


def on_device_of(t, fn, *args):
    """Call fn(*args, stream) with t's device current and torch's current stream of
    that device as the trailing void* argument.  The device switch is skipped when
    t's device already is the current one (the common case: saves ~10 us/call)."""
    idx = t.device.index
    if idx is None or idx == torch.cuda.current_device():
        return fn(*args, _vp(torch.cuda.current_stream().cuda_stream))
    with torch.cuda.device(idx):
        return fn(*args, _vp(torch.cuda.current_stream().cuda_stream))


def check_tensors(f32=(), i32=()):
    """GPU + dtype + contiguity checks for the raw-pointer boundary (the reference's
    C++ does none and reads garbage instead)."""
    for name, t in f32:
        if not t.is_cuda:
            raise RuntimeError("genpc_amd: GPU tensors only (got a %s tensor for %s); the "
                               "HIP path has no CPU fallback" % (t.device, name))
        if t.dtype != torch.float32:
            raise TypeError("genpc_amd: %s must be torch.float32, got %s" % (name, t.dtype))
        if not t.is_contiguous():
            raise ValueError("genpc_amd: %s must be contiguous" % name)
    for name, t in i32:
        if not t.is_cuda:
            raise RuntimeError("genpc_amd: GPU tensors only (got a %s tensor for %s); the "
                               "HIP path has no CPU fallback" % (t.device, name))
        if t.dtype != torch.int32:
            raise TypeError("genpc_amd: %s must be torch.int32, got %s" % (name, t.dtype))
        if not t.is_contiguous():
            raise ValueError("genpc_amd: %s must be contiguous" % name)


def require_gpu(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("genpc_amd: GPU tensors only (got a %s tensor); the "
                               "HIP path has no CPU fallback" % t.device)


def require(t, dtype, name):
    if t.dtype != dtype:
        raise TypeError("genpc_amd: %s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("genpc_amd: %s must be contiguous" % name)


def thread_state():
    """The calling thread's library modes (arithmetic override, nn / emd / pose / fps tunes) plus torch's grad mode, to be
    handed to worker threads (apply_thread_state): they are thread-local, and a new thread starts with the defaults."""
    buf = (ctypes.c_int * 8)()
    lib.genpc_thread_state_export(ctypes.cast(buf, ctypes.c_void_p))
    return (tuple(buf), torch.is_grad_enabled())


def apply_thread_state(state):
    """Installs a thread_state() snapshot in the calling thread; returns the state it replaces."""
    prev = thread_state()
    buf = (ctypes.c_int * 8)(*state[0])
    lib.genpc_thread_state_import(ctypes.cast(buf, ctypes.c_void_p))
    torch.set_grad_enabled(state[1])
    return prev
