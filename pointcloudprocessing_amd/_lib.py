"""ctypes binding of libpointnet_hip.so (include/pointnet_hip.h).

There is no CPU fallback: if the library cannot be loaded every entry point raises.  The library is
built in-tree by ``__graft_entry__.build()`` / ``make -C pointcloudprocessing_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpointnet_hip.so")
CSRC = os.path.join(_HERE, "csrc")

PN_PREC_BF16 = 1
PN_PREC_BF16X3 = 3
PN_STORE_BF16 = 0x100      # or-ed into prec: the per-point layer-boundary tensors are stored as bf16
PN_IO_KEEP_ACTIVATIONS = 1  # pn_model_io.flags: no vertical fusion of a frozen segmentation head (its activations stay inspectable)
ABI_VERSION = 6            # PN_ABI_VERSION of include/pointnet_hip.h this binding was written against
PN_NUM_BLOCKS = 15
# "bf16": bf16 MFMA operands AND bf16 storage of the layer-boundary tensors (half the HBM traffic of a step);
# "bf16_f32act": bf16 operands, fp32 storage; "bf16x3": split operands (fp32-grade products), fp32 storage
PREC = {"bf16": PN_PREC_BF16 | PN_STORE_BF16, "bf16_f32act": PN_PREC_BF16, "bf16x3": PN_PREC_BF16X3}

BLOCK_NAMES = ["input_transform", "mlp_1_1", "mlp_1_2", "feature_transform", "mlp_2_1", "mlp_2_2", "mlp_2_3",
               "mlp_cls_1", "mlp_cls_2", "mlp_cls_3", "mlp_seg_1", "mlp_seg_2", "mlp_seg_3", "mlp_seg_4",
               "mlp_seg_5"]


class PointNetHipError(RuntimeError):
    pass


class pn_operand(C.Structure):
    _fields_ = [("s1", C.c_void_p), ("s2", C.c_void_p), ("ca", C.c_void_p), ("cb", C.c_void_p),
                ("cc", C.c_void_p), ("ld", C.c_int64), ("lo", C.c_float), ("h16", C.c_int32)]


class pn_dense_tail(C.Structure):
    _fields_ = [("z", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("keep", C.c_void_p), ("keep_scale", C.c_float), ("bn_mode", C.c_int32), ("act", C.c_int32),
                ("dz", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dbias", C.c_void_p)]


class pn_dense_wgrad_job(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int32), ("dz", C.c_void_p), ("R", C.c_int32), ("K", C.c_int32), ("C", C.c_int32),
                ("dw", C.c_void_p), ("db", C.c_void_p)]


class pn_model_desc(C.Structure):
    _fields_ = [("ccls", C.c_int32), ("cseg", C.c_int32), ("vanilla", C.c_int32), ("reg_in", C.c_int32),
                ("reg_feat", C.c_int32), ("prec", C.c_int32), ("dropout_rate", C.c_float),
                ("bn_momentum", C.c_float), ("bn_eps", C.c_float), ("sync_world", C.c_int32)]


class pn_slot_info(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("offset", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("kind", C.c_int32), ("block", C.c_int32)]


class pn_model_io(C.Structure):
    _fields_ = [("pc", C.c_void_p), ("B", C.c_int32), ("N", C.c_int32), ("params", C.c_void_p),
                ("grads", C.c_void_p), ("trainable", C.c_void_p), ("training", C.c_int32), ("zero_grads_in_forward", C.c_int32),
                ("keep1", C.c_void_p), ("keep2", C.c_void_p), ("labels_cls", C.c_void_p),
                ("labels_seg", C.c_void_p), ("se3", C.c_void_p), ("loss_weights", C.c_float * 3),
                ("pad2_", C.c_float), ("out_cls", C.c_void_p), ("out_seg", C.c_void_p), ("out_R", C.c_void_p),
                ("scalars", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("prof_events", C.POINTER(C.c_void_p)), ("aux_stream", C.c_void_p),
                ("bwd_phase", C.c_int32), ("flags", C.c_int32), ("dropout_seed", C.c_uint64), ("dropout_step", C.c_void_p),
                ("sync_rank", C.c_int32), ("pad3_", C.c_int32), ("sync_hook", C.c_void_p), ("sync_ctx", C.c_void_p)]


# pn_model_io.sync_hook: int (*)(void* ctx, int op, const void* src, void* dst, int64_t n, int dtype, void* stream)
SYNC_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)


# every symbol include/pointnet_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_I = C.c_int
_I64 = C.c_int64
_F = C.c_float
_D = C.c_double
_OP = C.POINTER(pn_operand)
_DESC = C.POINTER(pn_model_desc)
_IO = C.POINTER(pn_model_io)
SIGNATURES = {
    "pn_abi_version": (_I, []),
    "pn_last_error": (C.c_char_p, []),
    "pn_normalize": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "pn_conv3_fwd": (_I, [_P, _P, _I64, _I, _I, _I, _P, _P, _P]),
    "pn_conv3_wgrad": (_I, [_P, _OP, _I, _I, _I, _P, _P]),
    "pn_conv_fwd": (_I, [_OP, _P, _I64, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "pn_conv_fwd_max": (_I, [_OP, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    "pn_weights_prep": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "pn_panel_slots_per_cloud": (_I, [_I, _I]),
    "pn_conv_fwd_max_panel": (_I, [_OP, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    "pn_panel_finalize": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pn_chain_fwd_max": (_I, [_OP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "pn_weights_copy16": (_I, [_P, _I, _I, _P, _P, _P]),
    "pn_max_resolve": (_I, [_OP, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "pn_maxbwd_scatter": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "pn_conv_bwd_data": (_I, [_OP, _P, _I64, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "pn_conv_wgrad": (_I, [_OP, _OP, _I, _I, _I, _I, _I, _P, _I, _P]),
    "pn_slab_reduce": (_I, [_P, _I, _I, _I64, _P, _P]),
    "pn_bn_finalize": (_I, [_P, _I, _I, _I64, _P, _P, _P, _P, _F, _F, _I, _I, _P, _P, _P, _P, _P]),
    "pn_bn_bwd_finalize": (_I, [_P, _I, _I, _I64, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "pn_sign": (_I, [_P, _I, _P, _P]),
    "pn_max_finalize": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "pn_dense_workspace_floats": (C.c_size_t, [_I, _I, _I]),
    "pn_dense_layer": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _P, _F, _P, _P, _P, _P, _P]),
    "pn_dense_bwd_step": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "pn_dense_wgrad_batch": (_I, [_P, _I, _P]),
    "pn_dense_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _F, _P, _P, _P, _P, _P, _P]),
    "pn_softmax_xent": (_I, [_P, _I, _I, _P, _F, _P, _P, _P, _P, _P]),
    "pn_argmax_rows": (_I, [_P, _I64, _I, _P, _P]),
    "pn_seg_out_part_stride": (_I, []),
    "pn_seg_out_part_rows": (_I, []),
    "pn_seg_out_fwd": (_I, [_OP, _P, _P, _I64, _I, _I, _P, _F, _P, _P, _P, _P]),
    "pn_bmm": (_I, [_P, _P, _I, _I, _I, _P, _I, _P]),
    "pn_dropout_masks": (_I, [_P, _I64, _P, _I64, _F, C.c_uint64, _P, _P]),
    "pn_count_nonfinite": (_I, [_P, _I64, _I, _P, _P]),
    "pn_fps_workspace_bytes": (C.c_size_t, [_I, _I]),
    "pn_fps": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, C.c_size_t, _P]),
    "pn_voxel_workspace_bytes": (C.c_size_t, [_I]),
    "pn_voxel_downsample": (_I, [_P, _P, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _I, _P, _P, _P, _P, _P,
                                 C.c_size_t, _P]),
    "pn_model_num_slots": (_I, [_DESC]),
    "pn_model_param_floats": (_I64, [_DESC]),
    "pn_model_slot_info": (_I, [_DESC, _I, C.POINTER(pn_slot_info)]),
    "pn_model_workspace_bytes": (C.c_size_t, [_DESC, _I, _I, _I]),
    "pn_model_ws_lookup": (_I, [_DESC, _I, _I, _I, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pn_model_ws_entry": (_I, [_DESC, _I, _I, _I, _I, C.c_char_p, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pn_model_forward": (_I, [_DESC, _IO, _P]),
    "pn_model_backward": (_I, [_DESC, _IO, _P, _P, _P, _P]),
    "pn_adam_prepare": (_I, [_P, _P, _D, _D, _D, _D, _D, _P]),
    "pn_adam_step": (_I, [_P, _P, _P, _P, _I64, _P, _P, _D, _D, _D, _D, _D, _D, _F, _P]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into libpointnet_hip.so (in-tree)."""
    res = subprocess.run(["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise PointNetHipError("building libpointnet_hip.so failed")
    return LIB_PATH


def lib():
    """The loaded library with argtypes set; raises PointNetHipError if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PointNetHipError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C {CSRC}`).  There is no CPU fallback for the PointNet hot path.")
        # PyTorch first: its wheel bundles its own libamdhip64 and every stream / device pointer handed to the library comes from that
        # runtime.  Loaded before torch, the library binds /opt/rocm's copy instead and the process ends up with two HIP runtimes
        # (launches then fail with "no ROCm-capable device is detected").
        import torch  # noqa: F401
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:
            raise PointNetHipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.pn_abi_version() != ABI_VERSION:
            raise PointNetHipError("libpointnet_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().pn_last_error()
        raise PointNetHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu_tensor(t, name, dtype=None):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise PointNetHipError(f"{name} must be a CUDA/HIP tensor: the PointNet hot path has no CPU fallback")
    if not t.is_contiguous():
        raise PointNetHipError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise PointNetHipError(f"{name} must have dtype {dtype}, got {t.dtype}")
    return t


def operand(s1, ca=None, cc=None, s2=None, cb=None, relu=False, ld=None):
    o = pn_operand()
    o.s1 = s1.data_ptr()
    o.s2 = s2.data_ptr() if s2 is not None else None
    o.ca = ca.data_ptr() if ca is not None else None
    o.cb = cb.data_ptr() if cb is not None else None
    o.cc = cc.data_ptr() if cc is not None else None
    o.ld = ld if ld is not None else s1.shape[-1]
    o.lo = 0.0 if relu else float("-inf")
    import torch
    o.h16 = 1 if s1.dtype == torch.bfloat16 else 0       # bf16 sources (s2 must match s1)
    if s2 is not None and s2.dtype != s1.dtype:
        raise PointNetHipError("operand: s1 and s2 must have the same dtype")
    o._keepalive = (s1, s2, ca, cb, cc)   # the struct holds raw device pointers: keep their owners alive
    return o
