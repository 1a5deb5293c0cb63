"""ctypes binding of ``libunet_tm_hip.so`` (declared in ``include/unet_tm_hip.h``).

There is no fallback: if the shared library is missing or an entry point fails, an
exception is raised.  Build it with ``python __graft_entry__.py build`` (or ``make -C
csrc``)."""
from __future__ import annotations

import ctypes as C
import os

MI_F32, MI_BF16 = 0, 1
MI_FWD_REPACK, MI_FWD_TRAIN = 1, 2
MI_OPT_GRAPH, MI_OPT_SERIAL_BACKWARD, MI_OPT_DEBUG_STREAMS, MI_OPT_PACK_ONE_STREAM, MI_OPT_NO_KSPLIT = 1, 2, 3, 4, 5

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libunet_tm_hip.so")


class UNetDesc(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("n_levels", C.c_int),
                ("features", C.c_int * 8), ("batch", C.c_int), ("height", C.c_int), ("width", C.c_int),
                ("out_h", C.c_int), ("out_w", C.c_int), ("dtype", C.c_int), ("need_input_grad", C.c_int)]


_i, _f, _p, _sz, _i64 = C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_int64
_VIEW = [_p, _i, _i]  # pointer, channel stride, channel offset

# name -> (restype, argtypes); must list every symbol the header declares.
SIGNATURES = {
    "mi_version": (_i, []),
    "mi_last_error": (C.c_char_p, []),
    "mi_pad_channels": (_i, [_i]),
    "mi_nchw_to_nhwc": (_i, [_p] + _VIEW + [_i, _i, _i, _i, _i, _p]),
    "mi_nhwc_to_nchw": (_i, _VIEW + [_p, _i, _i, _i, _i, _i, _p]),
    "mi_pack_conv3x3": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "mi_pack_convt2x2": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "mi_packed_conv3x3_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "mi_packed_convt2x2_bytes": (_sz, [_i, _i, _i, _i]),
    "mi_conv3x3_relu_fwd": (_i, _VIEW + [_p, _p] + _VIEW + [_i] * 8 + [_p]),
    "mi_conv3x3_dgrad": (_i, _VIEW + [_p] + _VIEW + _VIEW + [_i] * 7 + [_p]),
    "mi_conv3x3_wgrad": (_i, _VIEW + _VIEW + [_p, _p, _f, _f, _p, _sz] + [_i] * 7 + [_p]),
    "mi_conv3x3_wgrad_ws_bytes": (_sz, [_i] * 7),
    "mi_conv3x3_wgrad_static": (_i, _VIEW + _VIEW + [_p, _p, _f, _f] + [_i] * 7 + [_p]),
    "mi_conv3x3_wgrad_static_max_bytes": (_sz, []),
    "mi_maxpool2_fwd": (_i, _VIEW + _VIEW + [_i] * 5 + [_p]),
    "mi_maxpool2_bwd": (_i, _VIEW + _VIEW + _VIEW + _VIEW + [_i] * 6 + [_p]),
    "mi_convt2x2_fwd": (_i, _VIEW + [_p, _p] + _VIEW + [_i] * 6 + [_p]),
    "mi_convt2x2_dgrad": (_i, _VIEW + [_p] + _VIEW + _VIEW + [_i] * 6 + [_p]),
    "mi_convt2x2_wgrad": (_i, _VIEW + _VIEW + [_p, _p, _f, _f, _p, _sz] + [_i] * 6 + [_p]),
    "mi_convt2x2_wgrad_ws_bytes": (_sz, [_i] * 6),
    "mi_conv1x1_head_fwd": (_i, _VIEW + [_p, _p, _p] + [_i] * 6 + [_p]),
    "mi_conv1x1_head_bwd": (_i, _VIEW + [_p, _p] + _VIEW + [_p, _p, _f, _f, _p, _sz] + [_i] * 7 + [_p]),
    "mi_conv1x1_head_ws_bytes": (_sz, [_i, _i]),
    "mi_mse_fwd_bwd": (_i, [_p, _p, _i64, _p, _p, _f, _p, _p, _sz, _p]),
    "mi_mse_ws_bytes": (_sz, [_i64]),
    "mi_bilinear_fwd": (_i, [_p] + [_i64] * 4 + [_i, _p] + [_i64] * 4 + [_i] + [_i] * 6 + [_p]),
    "mi_bilinear_bwd": (_i, [_p] + [_i64] * 4 + [_i, _p] + [_i64] * 4 + [_i] + [_i] * 6 + [_p]),
    "mi_adam_step": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _i, _f, _p]),
    "mi_series_minmax_ws_bytes": (_sz, []),
    "mi_series_minmax": (_i, [_p, _i64, _p, _p, _sz, _p]),
    "mi_window_count": (_i64, [_i64, _i, _i, _i, _i, _i, _i]),
    "mi_window_batch": (_i, [_p, _i64, _p, _i, _i, _i, _i, _i, _i, _i, _i, C.c_double, C.c_double, _p, _p, _p]),
    "mi_series_reconstruct": (_i, [_p, _p, _p, _i64, _i, _i, _i, _i, _i64, _p, _p, _p, _p]),
    "mi_unet_plan_create": (_i, [C.POINTER(UNetDesc), C.POINTER(_p)]),
    "mi_unet_plan_destroy": (None, [_p]),
    "mi_unet_plan_workspace_bytes": (_sz, [_p]),
    "mi_unet_plan_flat_elems": (_i64, [_p]),
    "mi_unet_plan_num_params": (_i, [_p]),
    "mi_unet_plan_param_info": (_i, [_p, _i, C.c_char_p, _i, C.POINTER(_i64), C.POINTER(_i64 * 4), C.POINTER(_i)]),
    "mi_unet_plan_num_tensors": (_i, [_p]),
    "mi_unet_plan_tensor_info": (_i, [_p, _i, C.c_char_p, _i, C.POINTER(_i64), C.POINTER(_i * 4), C.POINTER(_i),
                                      C.POINTER(_i), C.POINTER(_i)]),
    "mi_unet_plan_num_segments": (_i, [_p]),
    "mi_unet_plan_segment_range": (_i, [_p, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "mi_unet_forward": (_i, [_p, _p, _p, _p, _sz, _p, _i, _p]),
    "mi_unet_backward": (_i, [_p, _p, _p, _p, _sz, _p, _f, _f, _i, _i, _p, _p]),
    "mi_unet_plan_set_option": (_i, [_p, _i, _i]),
    "mi_unet_plan_ksplit_region": (_i, [_p, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    "mi_unet_plan_defer_join": (_i, [_p, _i]),
    "mi_unet_plan_side_stream": (_p, [_p, _p]),
    "mi_unet_plan_last_call_deferred": (_i, [_p]),
    "mi_unet_plan_prepare": (_i, [_p, _p]),
    "mi_unet_plan_adam_pack_supported": (_i, [_p]),
    "mi_unet_adam_pack": (_i, [_p, _p, _p, _p, _p, _p, _sz, _f, _f, _f, _f, _i, _f, _p]),
    "mi_unet_profile_enable": (_i, [_p, _i]),
    "mi_unet_profile_count": (_i, [_p]),
    "mi_unet_profile_read": (_i, [_p, _i, C.c_char_p, _i, C.c_char_p, _i, C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(_f)]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} is missing: the HIP extension has not been built and there is no "
                "CPU/PyTorch fallback. Run `python __graft_entry__.py build`.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    """Turn a non-zero status into a RuntimeError carrying the library's message."""
    if rc != 0:
        msg = lib().mi_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libunet_tm_hip {what} failed (status {rc}): {msg}")
