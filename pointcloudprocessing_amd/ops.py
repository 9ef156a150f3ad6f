"""Thin torch-tensor wrappers over the op-level C ABI (include/pointnet_hip.h).

Every function takes contiguous fp32 HIP tensors, allocates its outputs with torch (device memory plumbing
only) and enqueues the HIP kernels on torch's current stream.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import check, current_stream, lib, operand, ptr, require_gpu_tensor

F32 = torch.float32


def _tiles(B, N):
    return B * ((N + 127) // 128)


def normalize(xyz: torch.Tensor):
    """PointCloudNormalization.call (reference pointnet/PointNet.py:691-706) -> (normalized, (centroid, scale))."""
    require_gpu_tensor(xyz, "xyz", F32)
    B, N, _ = xyz.shape
    out = torch.empty_like(xyz)
    cen = torch.empty(B, 1, 3, device=xyz.device, dtype=F32)
    scl = torch.empty(B, 1, 1, device=xyz.device, dtype=F32)
    check(lib().pn_normalize(ptr(xyz), B, N, ptr(out), ptr(cen), ptr(scl), current_stream()), "pn_normalize")
    return out, (cen, scl)


def conv3_fwd(x3, w, B, N, per_cloud=False, want_stats=True):
    C_ = w.shape[-1]
    z = torch.empty(B * N, C_, device=x3.device, dtype=F32)
    part = torch.empty(_tiles(B, N), 2, C_, device=x3.device, dtype=F32) if want_stats else None
    check(lib().pn_conv3_fwd(ptr(x3), ptr(w), 3 * C_ if per_cloud else 0, B, N, C_, ptr(z), ptr(part), current_stream()),
          "pn_conv3_fwd")
    return z, part


def conv3_wgrad(x3, dz_op, B, N, C_):
    slabs = torch.empty(_tiles(B, N), 3, C_, device=x3.device, dtype=F32)
    check(lib().pn_conv3_wgrad(ptr(x3), C.byref(dz_op), B, N, C_, ptr(slabs), current_stream()), "pn_conv3_wgrad")
    return slabs


def conv_fwd(x_op, w, B, N, K, C_, prec, w_cloud_stride=0, cloud_bias=None, store=True, want_stats=True):
    """prec | _lib.PN_STORE_BF16: z comes back as a bf16 tensor"""
    dev = w.device
    z = torch.empty(B * N, C_, device=dev, dtype=torch.bfloat16 if prec & _lib.PN_STORE_BF16 else F32) if store else None
    part = torch.empty(_tiles(B, N), 2, C_, device=dev, dtype=F32) if want_stats else None
    check(lib().pn_conv_fwd(C.byref(x_op), ptr(w), w_cloud_stride, B, N, K, C_, ptr(cloud_bias), ptr(z), ptr(part), prec,
                            current_stream()), "pn_conv_fwd")
    return z, part


def conv_fwd_max(x_op, w, B, N, K, C_, sgn, prec):
    dev = w.device
    T = _tiles(B, N)
    pmax = torch.empty(T, C_, device=dev, dtype=F32)
    pidx = torch.empty(T, C_, device=dev, dtype=torch.int32)
    part = torch.empty(T, 2, C_, device=dev, dtype=F32)
    check(lib().pn_conv_fwd_max(C.byref(x_op), ptr(w), B, N, K, C_, ptr(sgn), ptr(pmax), ptr(pidx), ptr(part), prec,
                                current_stream()), "pn_conv_fwd_max")
    return pmax, pidx, part


def weights_prep(w, sgn=None):
    """fragment-ordered bf16 copies (hi, lo) of a Keras kernel (K, C), columns pre-multiplied by sign(sgn) (include/pointnet_hip.h)"""
    K, C_ = w.shape
    hi = torch.empty(C_ * K, device=w.device, dtype=torch.bfloat16)
    lo = torch.empty(C_ * K, device=w.device, dtype=torch.bfloat16)
    check(lib().pn_weights_prep(ptr(w), ptr(sgn), K, C_, ptr(hi), ptr(lo), current_stream()), "pn_weights_prep")
    return hi, lo


def conv_fwd_max_panel(x_op, wf, B, N, K, C_, prec, want_stats=True):
    """the row-panel kernel: per slot (run of 64-row panels) and channel max of sgn*z, the 32-row block holding it and (want_stats)
    sum z^2, and per cloud the column sums of the staged operand rows in 2^-24 fixed point (panel_finalize turns those into the channel sums of z).
    wf = weights_prep(w, gamma)."""
    dev = wf[0].device
    T = B * lib().pn_panel_slots_per_cloud(B, N)
    pmax = torch.empty(T, C_, device=dev, dtype=F32)
    pblk = torch.empty(T, C_, device=dev, dtype=torch.int32)
    sumsq = torch.empty(T, C_, device=dev, dtype=F32) if want_stats else None
    colsum = torch.zeros(B, (2 if (prec & 3) == 3 else 1) * K, device=dev, dtype=torch.int64) if want_stats else None      # accumulators: zero on entry
    check(lib().pn_conv_fwd_max_panel(C.byref(x_op), ptr(wf[0]), ptr(wf[1]), B, N, K, C_, ptr(pmax), ptr(pblk), ptr(sumsq), ptr(colsum), prec,
                                      current_stream()), "pn_conv_fwd_max_panel")
    return pmax, pblk, sumsq, colsum


def weights_copy16(w):
    """bf16 copies of a Keras kernel (K, C): (as it is (K, C), transposed (C, K)) -- what the model plan's row GEMMs stage"""
    K, C_ = w.shape
    nat = torch.empty(K, C_, device=w.device, dtype=torch.bfloat16)
    tr = torch.empty(C_, K, device=w.device, dtype=torch.bfloat16)
    check(lib().pn_weights_copy16(ptr(w), K, C_, ptr(nat), ptr(tr), current_stream()), "pn_weights_copy16")
    return nat, tr


def chain_fwd_max(x_op, xyz, w1, w1t, sc1, sh1, w2t, sc2, sh2, wf_hi, B, N):
    """inference: ConvLayer(3 | 64 -> 64) -> ConvLayer(64 -> 128) -> ConvLayer(128 -> 1024) -> reduce_max in one launch (moving
    statistics); exactly one of x_op (64-channel bf16 lazy operand, with w1t) and xyz ((B*N, 3), with w1).  Returns (pmax, pblock) per
    slot, as conv_fwd_max_panel leaves them."""
    dev = wf_hi.device
    T = B * lib().pn_panel_slots_per_cloud(B, N)
    pmax = torch.empty(T, 1024, device=dev, dtype=F32)
    pblk = torch.empty(T, 1024, device=dev, dtype=torch.int32)
    check(lib().pn_chain_fwd_max(C.byref(x_op) if x_op is not None else None, ptr(xyz), ptr(w1), ptr(w1t), ptr(sc1), ptr(sh1), ptr(w2t),
                                 ptr(sc2), ptr(sh2), ptr(wf_hi), B, N, ptr(pmax), ptr(pblk), current_stream()), "pn_chain_fwd_max")
    return pmax, pblk


def panel_finalize(pmax, pblk, sumsq, colsum, wf, prec, B, N, K, gamma, beta, moving_mean, moving_var, training=True, momentum=0.99, eps=1e-3):
    """BN coefficients of the layer + reduce_max over each cloud's tiles -> (mean, invstd, scale, shift, g, zstar, arg_block);
    wf, prec, K: what the panel launch was given"""
    C_ = pmax.shape[1]
    dev = pmax.device
    mean, invstd, scale, shift = (torch.empty(C_, device=dev, dtype=F32) for _ in range(4))
    g = torch.empty(B, C_, device=dev, dtype=F32)
    zstar = torch.empty(B, C_, device=dev, dtype=F32)
    argb = torch.empty(B, C_, device=dev, dtype=torch.int32)
    check(lib().pn_panel_finalize(ptr(pmax), ptr(pblk), ptr(sumsq), ptr(colsum), ptr(wf[0]), ptr(wf[1]), prec, B, N, K, C_,
                                  ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var), momentum, eps, int(training), int(training),
                                  ptr(mean), ptr(invstd), ptr(scale), ptr(shift), ptr(g), ptr(zstar), ptr(argb), current_stream()),
          "pn_panel_finalize")
    return mean, invstd, scale, shift, g, zstar, argb


def max_resolve(x_op, wf, argb, B, N, K, C_, prec):
    """the row of each (cloud, channel) maximum inside its 32-row block"""
    arg = torch.empty(B, C_, device=argb.device, dtype=torch.int32)
    check(lib().pn_max_resolve(C.byref(x_op), ptr(wf[0]), ptr(wf[1]), ptr(argb), B, N, K, C_, ptr(arg), prec, current_stream()),
          "pn_max_resolve")
    return arg


def maxbwd_scatter(arg, hs, wt, q, B, N, K, C_, store16=False):
    """D[b][n][:] = q + sum over the channels whose maximum sits at row n of hs[b][c] * wt[c][:]   (fp32, or bf16 with store16)"""
    D = torch.empty(B * N, K, device=hs.device, dtype=torch.bfloat16 if store16 else torch.float32)
    check(lib().pn_maxbwd_scatter(ptr(arg), ptr(hs), ptr(wt), ptr(q), B, N, K, C_, ptr(D), int(store16), current_stream()), "pn_maxbwd_scatter")
    return D


def conv_bwd_data(dz_op, w, B, N, K, C_, prec, w_cloud_stride=0, addend=None, zmask=None, msc=None, msh=None,
                  want_stats=True):
    """prec | _lib.PN_STORE_BF16: out comes back as a bf16 tensor and addend / zmask must be bf16 tensors"""
    dev = w.device
    s16 = bool(prec & _lib.PN_STORE_BF16)
    for t, name in ((addend, "addend"), (zmask, "zmask")):
        if t is not None and t.dtype != (torch.bfloat16 if s16 else F32):
            raise _lib.PointNetHipError(f"conv_bwd_data: {name} must be {'bf16' if s16 else 'fp32'} for this prec")
    out = torch.empty(B * N, C_, device=dev, dtype=torch.bfloat16 if s16 else F32)
    part = torch.empty(_tiles(B, N), 2, C_, device=dev, dtype=F32) if want_stats else None
    check(lib().pn_conv_bwd_data(C.byref(dz_op), ptr(w), w_cloud_stride, B, N, K, C_, ptr(addend), ptr(zmask), ptr(msc),
                                 ptr(msh), ptr(out), ptr(part), prec, current_stream()), "pn_conv_bwd_data")
    return out, part


def conv_wgrad(a_op, b_op, B, N, Ci, Cj, prec, slab_rows=256, per_cloud=False):
    dev = torch.device("cuda")
    spc = (N + slab_rows - 1) // slab_rows
    slabs = torch.empty(B * spc, Ci, Cj, device=dev, dtype=F32)
    check(lib().pn_conv_wgrad(C.byref(a_op), C.byref(b_op), B, N, Ci, Cj, slab_rows, ptr(slabs), prec, current_stream()),
          "pn_conv_wgrad")
    groups = B if per_cloud else 1
    out = torch.empty(groups, Ci, Cj, device=dev, dtype=F32)
    check(lib().pn_slab_reduce(ptr(slabs), B * spc, spc if per_cloud else B * spc, Ci * Cj, ptr(out), current_stream()),
          "pn_slab_reduce")
    return out if per_cloud else out[0]


def slab_reduce(slabs, per_group):
    n = slabs.shape[0]
    elems = slabs[0].numel()
    out = torch.empty(n // per_group, *slabs.shape[1:], device=slabs.device, dtype=F32)
    check(lib().pn_slab_reduce(ptr(slabs), n, per_group, elems, ptr(out), current_stream()), "pn_slab_reduce")
    return out


def bn_finalize(part, count, gamma, beta, moving_mean, moving_var, use_batch_stats=True, update_moving=True,
                momentum=0.99, eps=1e-3):
    C_ = gamma.numel()
    dev = gamma.device
    mean, invstd, scale, shift = (torch.empty(C_, device=dev, dtype=F32) for _ in range(4))
    nt = part.shape[0] if part is not None else 0
    check(lib().pn_bn_finalize(ptr(part), nt, C_, count, ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var), momentum,
                               eps, int(use_batch_stats), int(update_moving), ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                               current_stream()), "pn_bn_finalize")
    return mean, invstd, scale, shift


def bn_bwd_finalize(part, count, gamma, mean, invstd, batch_stats=True):
    C_ = gamma.numel()
    dev = gamma.device
    dgamma, dbeta, ca, cb, cc = (torch.zeros(C_, device=dev, dtype=F32) for _ in range(5))
    nt = part.shape[0] if part is not None else 0
    check(lib().pn_bn_bwd_finalize(ptr(part), nt, C_, count, ptr(gamma), ptr(mean), ptr(invstd), int(batch_stats), ptr(dgamma),
                                   ptr(dbeta), ptr(ca), ptr(cb), ptr(cc), current_stream()), "pn_bn_bwd_finalize")
    return dgamma, dbeta, ca, cb, cc


def sign(gamma):
    s = torch.empty_like(gamma)
    check(lib().pn_sign(ptr(gamma), gamma.numel(), ptr(s), current_stream()), "pn_sign")
    return s


def max_finalize(pmax, pidx, B, sgn, scale, shift):
    T, C_ = pmax.shape
    dev = pmax.device
    g = torch.empty(B, C_, device=dev, dtype=F32)
    zstar = torch.empty(B, C_, device=dev, dtype=F32)
    arg = torch.empty(B, C_, device=dev, dtype=torch.int32)
    check(lib().pn_max_finalize(ptr(pmax), ptr(pidx), B, T // B, C_, ptr(sgn), ptr(scale), ptr(shift), ptr(g), ptr(zstar),
                                ptr(arg), current_stream()), "pn_max_finalize")
    return g, zstar, arg


def argmax_rows(values: torch.Tensor) -> torch.Tensor:
    """index of the first maximum along the last axis (np.argmax order), int32; any leading shape"""
    require_gpu_tensor(values, "values", F32)
    v = values.contiguous()
    C_ = v.shape[-1]
    out = torch.empty(v.shape[:-1], dtype=torch.int32, device=v.device)
    check(lib().pn_argmax_rows(ptr(v), v.numel() // C_, C_, ptr(out), current_stream()), "pn_argmax_rows")
    return out


def farthest_point_sample(xyz: torch.Tensor, m: int, start_idx: int = 0, return_mindist: bool = False):
    """xyz (B,N,3) -> idx (B,m) int32 in selection order (spec: include/pointnet_hip.h, pn_fps)."""
    require_gpu_tensor(xyz, "xyz", F32)
    B, N, _ = xyz.shape
    idx = torch.empty(B, m, device=xyz.device, dtype=torch.int32)
    md = torch.empty(B, N, device=xyz.device, dtype=F32)
    nbytes = lib().pn_fps_workspace_bytes(B, N)
    ws = torch.empty(nbytes, device=xyz.device, dtype=torch.uint8)
    check(lib().pn_fps(ptr(xyz), B, N, m, start_idx, ptr(idx), ptr(md), ptr(ws), nbytes, current_stream()), "pn_fps")
    if int(ws[:4].view(torch.int32).item()) != 0:
        raise _lib.PointNetHipError("pn_fps: a block timed out waiting for a peer block")
    return (idx, md) if return_mindist else idx


def voxel_downsample(xyz: torch.Tensor, leaf, origin, labels: Optional[torch.Tensor] = None, n_labels: int = 0):
    """xyz (N,3) -> (centroids (V,3), counts (V,), majority (V,) or None) ordered by ascending (kz,ky,kx)."""
    require_gpu_tensor(xyz, "xyz", F32)
    N = xyz.shape[0]
    dev = xyz.device
    cent = torch.empty(N, 3, device=dev, dtype=F32)
    cnt = torch.empty(N, device=dev, dtype=torch.int32)
    maj = torch.empty(N, device=dev, dtype=torch.int32)
    nout = torch.zeros(1, device=dev, dtype=torch.int32)
    nbytes = lib().pn_voxel_workspace_bytes(N)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    leaf_c = (C.c_float * 3)(*[float(v) for v in leaf])
    org_c = (C.c_float * 3)(*[float(v) for v in origin])
    if labels is not None:
        require_gpu_tensor(labels, "labels", torch.int32)
    check(lib().pn_voxel_downsample(ptr(xyz), ptr(labels), N, leaf_c, org_c, n_labels, ptr(cent), ptr(cnt), ptr(maj), ptr(nout),
                                    ptr(ws), nbytes, current_stream()), "pn_voxel_downsample")
    v = int(nout.item())
    if int(ws[:4].view(torch.int32).item()) != 0:
        raise _lib.PointNetHipError("pn_voxel_downsample: a voxel key fell outside [0, 2^21)")
    return cent[:v], cnt[:v], (maj[:v] if labels is not None else None)


def dense_layer(x, w, trans=False, bias=None, gamma=None, beta=None, moving_mean=None, moving_var=None, bn_mode=0, act=0, keep=None,
                rate=0.0, momentum=0.99, eps=1e-3, counters=None):
    """DenseLayer forward in one launch: returns (z, a, mean, invstd); moving statistics are updated in place (bn_mode 1)."""
    R, K = x.shape
    C_ = w.shape[0] if trans else w.shape[1]
    dev = x.device
    ws = torch.empty(max(1, lib().pn_dense_workspace_floats(R, K, C_)), device=dev, dtype=F32)
    if counters is None:
        counters = torch.zeros(256, device=dev, dtype=torch.int32)
    z = torch.empty(R, C_, device=dev, dtype=F32)
    a = torch.empty(R, C_, device=dev, dtype=F32)
    mean = torch.empty(C_, device=dev, dtype=F32)
    invstd = torch.empty(C_, device=dev, dtype=F32)
    check(lib().pn_dense_layer(ptr(x), x.stride(0), ptr(w), w.stride(0), int(trans), R, K, C_, ptr(ws), ptr(counters), ptr(bias), ptr(gamma),
                               ptr(beta), ptr(moving_mean), ptr(moving_var), momentum, eps, bn_mode, act, ptr(keep),
                               1.0 / (1.0 - rate), ptr(z), ptr(a), ptr(mean), ptr(invstd), current_stream()), "pn_dense_layer")
    return z, a, mean, invstd


def dense_bwd_step(dz_above, w_above, z=None, gamma=None, beta=None, mean=None, invstd=None, bn_mode=0, act=0, keep=None, rate=0.0):
    """one launch of a backward chain: dx = dz_above . W_above^T (w_above: the (C, K) kernel of the layer above) and, with z given,
    the layer below taken backward in the same launch: returns (dx, dz, dgamma, dbeta, dbias)"""
    R, K = dz_above.shape
    C_ = w_above.shape[0]
    dev = dz_above.device
    ws = torch.empty(max(1, lib().pn_dense_workspace_floats(R, K, C_)), device=dev, dtype=F32)
    counters = torch.zeros(256, device=dev, dtype=torch.int32)
    dx = torch.empty(R, C_, device=dev, dtype=F32)
    dz = dg = db = dbias = None
    tail = None
    if z is not None:
        dz = torch.empty(R, C_, device=dev, dtype=F32)
        dg = torch.zeros(C_, device=dev, dtype=F32); db = torch.zeros(C_, device=dev, dtype=F32); dbias = torch.zeros(C_, device=dev, dtype=F32)
        tail = _lib.pn_dense_tail()
        for k, v in dict(z=z, gamma=gamma, beta=beta, mean=mean, invstd=invstd, keep=keep, dz=dz, dgamma=dg, dbeta=db, dbias=dbias).items():
            setattr(tail, k, None if v is None else v.data_ptr())
        tail.keep_scale = 1.0 / (1.0 - rate)
        tail.bn_mode, tail.act = bn_mode, act
    check(lib().pn_dense_bwd_step(ptr(dz_above), dz_above.stride(0), ptr(w_above), w_above.stride(0), R, K, C_, ptr(ws), ptr(counters), ptr(dx),
                                  C.byref(tail) if tail is not None else None, current_stream()), "pn_dense_bwd_step")
    return dx, dz, dg, db, dbias


def dense_wgrad_batch(jobs):
    """jobs: [(x (R, K), dz (R, C), want_db)] -> [(dw (K, C), db (C) or None)] in one launch"""
    arr = (_lib.pn_dense_wgrad_job * len(jobs))()
    outs = []
    for j, (x, dz, want_db) in zip(arr, jobs):
        R, K = x.shape
        C_ = dz.shape[1]
        dw = torch.empty(K, C_, device=x.device, dtype=F32)
        db = torch.empty(C_, device=x.device, dtype=F32) if want_db else None
        j.x, j.ldx, j.dz, j.R, j.K, j.C, j.dw, j.db = x.data_ptr(), x.stride(0), dz.data_ptr(), R, K, C_, dw.data_ptr(), (db.data_ptr() if want_db else None)
        outs.append((dw, db))
    check(lib().pn_dense_wgrad_batch(arr, len(jobs), current_stream()), "pn_dense_wgrad_batch")
    return outs


def dense_bwd(da, z, x, gamma=None, beta=None, mean=None, invstd=None, bn_mode=0, act=0, keep=None, rate=0.0, want_dw=True):
    """backward of the layer tail + parameters: returns (dz, dgamma, dbeta, dbias, dw)"""
    R, C_ = da.shape
    K = x.shape[1]
    dev = da.device
    dz = torch.empty(R, C_, device=dev, dtype=F32)
    dg = torch.zeros(C_, device=dev, dtype=F32); db = torch.zeros(C_, device=dev, dtype=F32); dbias = torch.zeros(C_, device=dev, dtype=F32)
    dw = torch.empty(K, C_, device=dev, dtype=F32) if want_dw else None
    check(lib().pn_dense_bwd(ptr(da), ptr(z), ptr(x), x.stride(0), R, K, C_, ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), bn_mode, act,
                             ptr(keep), 1.0 / (1.0 - rate), ptr(dz), ptr(dg), ptr(db), ptr(dbias), ptr(dw), current_stream()), "pn_dense_bwd")
    return dz, dg, db, dbias, dw
