"""ctypes binding of libdt_hip.so (C ABI: include/dt_hip.h).  No fallback: a missing or
unloadable library raises, and every non-zero status from the library raises."""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_longlong, c_size_t, c_uint8,
                    c_void_p)

_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libdt_hip.so")

COND_NONE, COND_ZERO, COND_ONE = 0, 1, 2
RULE_ENGINE, RULE_PSAMPLE, RULE_MANAGER = 0, 1, 2
PREC_FP32, PREC_SPLIT_BF16, PREC_AUTO = 0, 1, 2
BT_COUNT, GT_COUNT, N_BLOCKS = 16, 9, 8
ABI_VERSION = 4


class HipLibraryError(RuntimeError):
    pass


class UNetDesc(ctypes.Structure):
    _fields_ = [("channels", c_int32), ("dims", c_int32 * 4), ("temb_dim", c_int32)]


# name -> (restype, argtypes); must list every symbol include/dt_hip.h declares
SIGNATURES = {
    "dt_abi_version": (c_int, []),
    "dt_status_string": (c_char_p, [c_int]),
    "dt_unet_create": (c_int, [POINTER(UNetDesc), POINTER(c_void_p), POINTER(c_void_p), c_void_p, POINTER(c_void_p)]),
    "dt_unet_destroy": (None, [c_void_p]),
    "dt_unet_time_bias_stride": (c_int, [c_void_p]),
    "dt_unet_time_bias": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "dt_unet_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "dt_unet_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "dt_unet_autotune": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "dt_unet_conv_choice": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int),
                                    POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "dt_unet_set_conv_choice": (c_int, [c_void_p] + [c_int] * 10),
    "dt_unet_declare_shape": (c_int, [c_void_p] + [c_int] * 5),
    "dt_unet_set_precision": (c_int, [c_void_p, c_int]),
    "dt_unet_set_head_fusion": (c_int, [c_void_p, c_int]),
    "dt_unet_set_fused": (c_int, [c_void_p, c_int]),
    "dt_unet_fused_active": (c_int, [c_void_p, c_int, c_int]),
    "dt_unet_time_conv": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p, c_size_t, c_void_p, POINTER(c_float), POINTER(c_double)]),
    "dt_unet_debug_activation": (c_int, [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_size_t), POINTER(c_int),
                                         POINTER(c_int), POINTER(c_int)]),
    "dt_cfg_update": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_float), c_int,
                              c_void_p, c_float, c_void_p, c_int, c_int, c_void_p]),
    "dt_sample_trajectory": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, POINTER(c_float),
                                     POINTER(c_int32), c_void_p, c_void_p, POINTER(c_int64), c_void_p, c_float,
                                     c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dt_unet_forward_mixed": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "dt_sample_trajectory_mixed": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                           POINTER(c_float), POINTER(c_int32), c_void_p, c_void_p, POINTER(c_int64),
                                           c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dt_traj_metrics": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dt_traj_wasserstein": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                    c_void_p]),
    "dt_traj_pair_metrics": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dt_traj_resampled_distance": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dt_resize_bilinear": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dt_pair_stats": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dt_traj_sample_mean": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dt_profile_marker": (c_int, [c_int, c_void_p]),
    "dt_profile_begin": (c_int, []),
    "dt_profile_end": (c_int, []),
    "dt_profile_class_count": (c_int, []),
    "dt_profile_read": (c_int, [c_int, POINTER(c_char_p), POINTER(c_longlong), POINTER(c_double), POINTER(c_double),
                                POINTER(c_double)]),
}


def load(path=None):
    """Load (once) and return the library with argtypes set.  Raises HipLibraryError if absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise HipLibraryError(
            f"{path} is missing: the HIP extension has not been built (run "
            "`python -m distillation_trajectories_amd.csrc.build`). There is no CPU fallback.")
    import torch  # noqa: F401  -- first, so libamdhip64.so.7 resolves to the runtime torch already loaded
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise HipLibraryError(f"cannot load {path}: {e}. There is no CPU fallback.") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{path} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.dt_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"ABI mismatch: library {lib.dt_abi_version()} vs binding {ABI_VERSION}")
    _LIB = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().dt_status_string(status)
        raise HipLibraryError(f"{what} failed with status {status}: {msg.decode() if msg else '?'}")


def ptr(t):
    """device (or host) address of a torch tensor, None -> NULL"""
    return None if t is None else c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def profile_marker(marker_id):
    """Empty, recognisably named kernel on the current stream (brackets a region for external profilers)."""
    check(load().dt_profile_marker(int(marker_id), stream_ptr()), "dt_profile_marker")


def profile_begin():
    check(load().dt_profile_begin(), "dt_profile_begin")


def profile_end():
    """Stop recording and return {kernel class: dict(launches, ms, flops, bytes)} (stream must be synchronised)."""
    lib = load()
    check(lib.dt_profile_end(), "dt_profile_end")
    out = {}
    for cls in range(lib.dt_profile_class_count()):
        name, n, ms, fl, by = c_char_p(), c_longlong(), c_double(), c_double(), c_double()
        check(lib.dt_profile_read(cls, ctypes.byref(name), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl),
                                  ctypes.byref(by)), "dt_profile_read")
        if n.value:
            out[name.value.decode()] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value)
    return out
