"""GPU parity tests of the op-level C ABI against CPU restatements (numpy / torch-CPU).

Every test calls libpointnet_hip.so through pointcloudprocessing_amd.ops (ctypes) and compares with a
CPU computation of the same formula.  Integer-valued inputs make the bf16 MFMA path EXACT, so layout bugs
(row/column swaps, k permutations) show up as hard mismatches, not tolerance noise.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pointnet_oracle as O          # noqa: E402  (checker only)
from oracle import sampling_oracle as SO         # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ops():
    from pointcloudprocessing_amd import ops
    return ops


def _lib():
    from pointcloudprocessing_amd import _lib
    return _lib


def ints(gen, shape, lo=-3, hi=4):
    return torch.randint(lo, hi, shape, generator=gen).to(torch.float32)


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def lazy_ref(s1, ca=None, cc=None, s2=None, cb=None, relu=False):
    v = s1.double()
    if ca is not None:
        v = v * ca.double()
    if s2 is not None:
        v = v + cb.double() * s2.double()
    if cc is not None:
        v = v + cc.double()
    if relu:
        v = torch.clamp(v, min=0)
    return v


# ------------------------------------------------------------------------------------------------
def test_normalize_matches_reference_formula(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    for B, N in [(3, 1000), (2, 1), (1, 5000), (4, 257)]:
        s = torch.rand(B, 1, 1, generator=g) * 49 + 1
        o = (torch.rand(B, 1, 3, generator=g) * 2 - 1) * 100
        pc = o + s * (torch.rand(B, N, 3, generator=g) * 2 - 1)
        ref, (rc, rs) = O.normalize(pc.double())
        out, (cen, scl) = ops.normalize(pc.to(dev))
        assert torch.allclose(out.cpu().double(), ref, atol=2e-6, rtol=0), (B, N)
        assert torch.allclose(cen.cpu().double(), rc, atol=1e-4)
        assert torch.allclose(scl.cpu().double(), rs, rtol=1e-6)
    # degenerate cloud: all points equal -> scale clamps at 1e-7, output 0 (PointNet.py:701)
    pc = torch.ones(1, 64, 3) * 3.5
    out, (_, scl) = ops.normalize(pc.to(dev))
    assert float(out.abs().max()) == 0.0 and abs(float(scl) - 1e-7) < 1e-12


@pytest.mark.parametrize("B,N,per_cloud", [(2, 300, False), (3, 128, True), (1, 1, False)])
def test_conv3_fwd_and_wgrad(dev, B, N, per_cloud):
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B * N, 3, generator=g)
    w = torch.randn((B, 3, 64) if per_cloud else (3, 64), generator=g)
    z, part = ops.conv3_fwd(x.to(dev), w.to(dev), B, N, per_cloud=per_cloud)
    xr = x.view(B, N, 3).double()
    ref = xr @ (w.double() if per_cloud else w.double().unsqueeze(0))
    assert torch.allclose(z.cpu().double().view(B, N, 64), ref, atol=1e-5)
    s = part.cpu().double().sum(0)
    assert torch.allclose(s[0], ref.reshape(-1, 64).sum(0), atol=1e-3)
    assert torch.allclose(s[1], (ref.reshape(-1, 64) ** 2).sum(0), atol=1e-2, rtol=1e-5)
    # weight gradient with a two-source lazy dz
    dy = torch.randn(B * N, 64, generator=g)
    zz = torch.randn(B * N, 64, generator=g)
    ca, cb, cc = (torch.randn(64, generator=g) for _ in range(3))
    op = _lib().operand(dy.to(dev), ca=ca.to(dev), cc=cc.to(dev), s2=zz.to(dev), cb=cb.to(dev))
    slabs = ops.conv3_wgrad(x.to(dev), op, B, N, 64)
    dz = lazy_ref(dy, ca, cc, zz, cb)
    ref_w = torch.einsum("bnk,bnc->bkc", xr, dz.view(B, N, 64))
    tpc = (N + 127) // 128
    got = ops.slab_reduce(slabs, tpc).cpu().double()
    assert torch.allclose(got, ref_w, atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("prec", [1, 3])
@pytest.mark.parametrize("B,N,K,C", [(2, 256, 64, 64), (1, 200, 128, 128), (3, 130, 64, 128), (2, 64, 512, 256),
                                     (1, 384, 128, 1024)])
def test_conv_fwd_exact_on_integers(dev, prec, B, N, K, C):
    """small-integer operands are exact in bf16 and in the fp32 accumulator: result must match bit for bit."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = ints(g, (B * N, K))
    w = ints(g, (K, C))
    bias = ints(g, (B, C))
    op = _lib().operand(x.to(dev))
    z, part = ops.conv_fwd(op, w.to(dev), B, N, K, C, prec, cloud_bias=bias.to(dev))
    ref = (x.double() @ w.double()).view(B, N, C) + bias.double().unsqueeze(1)
    assert torch.equal(z.cpu().double().view(B, N, C), ref), "MFMA layout / indexing error"
    s = part.cpu().double().sum(0)
    assert torch.equal(s[0], ref.reshape(-1, C).sum(0))
    assert torch.equal(s[1], (ref.reshape(-1, C) ** 2).sum(0))


@pytest.mark.parametrize("prec", [1, 3])
def test_conv_fwd_lazy_bn_relu_and_per_cloud_weights(dev, prec):
    """tf.matmul(X, R_64) (PointNet.py:228) with the previous layer's BN+ReLU applied on load."""
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    B, N, K, C = 3, 200, 64, 64
    zprev = torch.randn(B * N, K, generator=g)
    sc = torch.rand(K, generator=g) + 0.5
    sh = torch.randn(K, generator=g) * 0.3
    R = torch.randn(B, K, C, generator=g) / 8
    op = _lib().operand(zprev.to(dev), ca=sc.to(dev), cc=sh.to(dev), relu=True)
    z, _ = ops.conv_fwd(op, R.to(dev), B, N, K, C, prec, w_cloud_stride=K * C, want_stats=False)
    a = lazy_ref(zprev, sc, sh, relu=True).float()
    if prec == 1:
        ref = (bf16r(a).view(B, N, K).double() @ bf16r(R).double())
        tol = 1e-4          # only the accumulation order differs
    else:
        ref = a.view(B, N, K).double() @ R.double()
        tol = 2e-4          # split-bf16: ~2^-17 per product
    err = (z.cpu().double().view(B, N, C) - ref).abs().max()
    assert err < tol * max(1.0, float(ref.abs().max())), float(err)


@pytest.mark.parametrize("panel", [0, 1])     # 0: the tiled kernel of the generic engine; 1: the kernel-stationary panel kernel
@pytest.mark.parametrize("prec", [1, 3])
@pytest.mark.parametrize("B,N", [(2, 256), (3, 200), (1, 1000), (2, 33), (16, 136), (40, 520), (300, 70)])
def test_conv_fwd_max_matches_reduce_max(dev, prec, B, N, panel):
    """ConvLayer(128->1024)+BN+ReLU+reduce_max (PointNet.py:242-248) without the (B,N,1024) tensor."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    K, C = 128, 1024
    x = ints(g, (B * N, K), -2, 3)
    w = ints(g, (K, C), -2, 3)
    gamma = torch.randn(C, generator=g)          # both signs: exercises the min side of the monotone trick
    beta = torch.randn(C, generator=g)
    mm = torch.zeros(C)
    mv = torch.ones(C)
    sgn = ops.sign(gamma.to(dev))
    op = _lib().operand(x.to(dev))
    if panel:
        # the row-panel kernel (the one the model plan uses): max / 32-row block / sum of squares per tile, column sums of the panel,
        # one finaliser for the BN coefficients and the reduce_max, and the row resolved among the block's 32 candidates
        wf = ops.weights_prep(w.to(dev), gamma.to(dev))
        pmax, pblk, sumsq, sumz = ops.conv_fwd_max_panel(op, wf, B, N, K, C, prec)
        mmd, mvd = mm.to(dev), mv.to(dev)
        mean, invstd, scale, shift, gfeat, zstar, argb = ops.panel_finalize(pmax, pblk, sumsq, sumz, wf, prec, B, N, K, gamma.to(dev), beta.to(dev), mmd, mvd,
                                                                           training=True)
        arg = ops.max_resolve(op, wf, argb, B, N, K, C, prec)
        assert int(argb.min()) >= 0 and int(argb.max()) <= (N - 1) // 32
    else:
        pmax, pidx, part = ops.conv_fwd_max(op, w.to(dev), B, N, K, C, sgn, prec)
        mean, invstd, scale, shift = ops.bn_finalize(part, B * N, gamma.to(dev), beta.to(dev), mm.to(dev), mv.to(dev))
        gfeat, zstar, arg = ops.max_finalize(pmax, pidx, B, sgn, scale, shift)
    z = (x.double() @ w.double()).view(B, N, C)
    m = z.reshape(-1, C).mean(0)
    v = ((z.reshape(-1, C) - m) ** 2).mean(0)
    assert torch.allclose(mean.cpu().double(), m, atol=1e-4)
    inv = torch.rsqrt(v + 1e-3) * gamma.double()
    y = torch.relu(z * inv + (beta.double() - m * inv))
    ref = y.max(dim=1).values
    assert torch.allclose(gfeat.cpu().double(), ref, atol=2e-4, rtol=1e-4)
    # argmax: the value at the reported row must be the extremum; reported row must be the lowest such row
    s = torch.where(gamma >= 0, 1.0, -1.0).double()
    t = z * s
    tv = t.max(dim=1).values
    first = torch.where(t == tv.unsqueeze(1), torch.arange(N).view(1, N, 1).expand_as(t), N).min(dim=1).values
    assert torch.equal(arg.cpu().long(), first)
    assert torch.equal(zstar.cpu().double(), tv * s)
    if panel:      # the panel path also hands back the statistics: biased variance, and the moving statistics moved by 1 - 0.99
        assert torch.allclose(invstd.cpu().double(), torch.rsqrt(v + 1e-3), rtol=1e-4)
        assert torch.allclose(mmd.cpu().double(), 0.01 * m, atol=1e-5) and torch.allclose(mvd.cpu().double(), 0.99 + 0.01 * v, rtol=1e-4)


@pytest.mark.parametrize("prec", [1, 3])
def test_panel_path_on_real_valued_clouds_with_duplicated_points(dev, prec):
    """the panel kernel + finaliser + row resolution on real-valued operands: zstar is the exact maximum of the MFMA products, the
    resolved row reaches that maximum within fp32 rounding of a 128-term dot product, and duplicated points (identical rows: the
    reference pads clouds that way, PointCloudSet.py:459-463) resolve to the LOWEST index, as the oracle's reduce_max does."""
    ops = _ops()
    g = torch.Generator().manual_seed(15)
    B, N, K, C = 3, 700, 128, 1024
    x = torch.randn(B * N, K, generator=g)
    xv = x.view(B, N, K)
    xv[:, 300:500] = xv[:, 100:300]                 # every row 100..299 of a cloud appears again 200 rows later (another panel)
    xv[:, 1] = xv[:, 0]                             # and a duplicate inside one 32-row block
    sc = torch.rand(K, generator=g) + 0.5
    sh = torch.randn(K, generator=g) * 0.3
    w = torch.randn(K, C, generator=g) / 11
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    op = _lib().operand(x.to(dev), ca=sc.to(dev), cc=sh.to(dev), relu=True)
    wf = ops.weights_prep(w.to(dev), gamma.to(dev))
    pmax, pblk, sumsq, sumz = ops.conv_fwd_max_panel(op, wf, B, N, K, C, prec)
    mmd, mvd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd, scale, shift, gfeat, zstar, argb = ops.panel_finalize(pmax, pblk, sumsq, sumz, wf, prec, B, N, K, gamma.to(dev), beta.to(dev), mmd, mvd,
                                                                       training=True)
    arg = ops.max_resolve(op, wf, argb, B, N, K, C, prec).cpu().long()
    a = lazy_ref(x, sc, sh, relu=True).float()
    aq, wq = (bf16r(a), bf16r(w)) if prec == 1 else (a, w)
    z = (aq.double().view(B, N, K) @ wq.double())                                     # (B, N, C)
    s = torch.where(gamma >= 0, 1.0, -1.0).double()
    t = z * s
    tv = t.max(dim=1).values
    tol = (1e-5 if prec == 1 else 2e-4) * float(t.abs().max())
    assert float((zstar.cpu().double() * s - tv).abs().max()) < tol
    at_arg = t.gather(1, arg.unsqueeze(1)).squeeze(1)
    assert float((tv - at_arg).abs().max()) < tol                                      # the resolved row attains the maximum
    first = torch.where(t >= (tv - tol).unsqueeze(1), torch.arange(N).view(1, N, 1).expand_as(t), N).min(dim=1).values
    dup_hi = (arg >= 300) & (arg < 500)
    assert not bool(dup_hi.any()), "a duplicated row was resolved to its later copy"    # ties go to the lowest index
    assert not bool((arg == 1).any())
    assert float((arg == first).double().mean()) > 0.999                               # elsewhere: the oracle's own first maximum
    m, v = z.reshape(-1, C).mean(0), z.reshape(-1, C).var(0, unbiased=False)
    assert torch.allclose(mean.cpu().double(), m, atol=2e-4 * float(z.abs().max()))
    assert torch.allclose(invstd.cpu().double(), torch.rsqrt(v + 1e-3), rtol=2e-3)


@pytest.mark.parametrize("prec", [1, 3])
def test_conv_bwd_data_mask_addend_stats(dev, prec):
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    B, N, K, C = 2, 200, 128, 64          # layer 64 -> 128; dz has K=128 columns, output C=64
    dy = ints(g, (B * N, K), -2, 3)
    zz = ints(g, (B * N, K), -2, 3)
    ca = ints(g, (K,), 1, 3)
    cb = ints(g, (K,), -1, 2)
    cc = ints(g, (K,), -1, 2)
    w = ints(g, (C, K), -2, 3)             # Keras kernel (Cin=64, Cout=128)
    addend = ints(g, (B * N, C))
    zprev = torch.randn(B * N, C, generator=g)
    msc = torch.randn(C, generator=g)
    msh = torch.randn(C, generator=g) * 0.2
    op = _lib().operand(dy.to(dev), ca=ca.to(dev), cc=cc.to(dev), s2=zz.to(dev), cb=cb.to(dev))
    out, part = ops.conv_bwd_data(op, w.to(dev), B, N, K, C, prec, addend=addend.to(dev), zmask=zprev.to(dev),
                                  msc=msc.to(dev), msh=msh.to(dev))
    dz = lazy_ref(dy, ca, cc, zz, cb)
    da = dz @ w.double().t() + addend.double()
    mask = (torch.addcmul(msh, msc, zprev) > 0)      # fmaf(msc, z, msh) > 0
    ref = torch.where(mask, da, torch.zeros_like(da))
    got = out.cpu().double()
    bad = (got != ref)
    # the mask is evaluated with an fma on the GPU; allow disagreement only where msc*z+msh is ~0
    near0 = (msc * zprev + msh).abs() < 1e-5
    assert not (bad & ~near0).any(), int((bad & ~near0).sum())
    s = part.cpu().double().sum(0)
    assert torch.allclose(s[0], got.sum(0), atol=1e-6)
    assert torch.allclose(s[1], (got * zprev.double()).sum(0), atol=1e-2, rtol=1e-5)


@pytest.mark.parametrize("prec", [1, 3])
@pytest.mark.parametrize("Ci,Cj,per_cloud", [(64, 64, True), (64, 128, False), (128, 128, False), (64, 512, False),
                                             (512, 256, False)])
def test_conv_wgrad_exact_on_integers(dev, prec, Ci, Cj, per_cloud):
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    B, N = 2, 300
    a = ints(g, (B * N, Ci), -2, 3)
    dy = ints(g, (B * N, Cj), -2, 3)
    zz = ints(g, (B * N, Cj), -1, 2)
    cb = ints(g, (Cj,), -1, 2)
    cc = ints(g, (Cj,), -1, 2)
    a_op = _lib().operand(a.to(dev), relu=True)
    b_op = _lib().operand(dy.to(dev), cc=cc.to(dev), s2=zz.to(dev), cb=cb.to(dev))
    got = ops.conv_wgrad(a_op, b_op, B, N, Ci, Cj, prec, slab_rows=128, per_cloud=per_cloud).cpu().double()
    ar = torch.clamp(a.double(), min=0).view(B, N, Ci)
    dz = lazy_ref(dy, None, cc, zz, cb).view(B, N, Cj)
    ref = torch.einsum("bni,bnj->bij", ar, dz)
    if not per_cloud:
        ref = ref.sum(0)
    assert torch.equal(got, ref), float((got - ref).abs().max())


def test_bn_finalize_and_backward_coefficients(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    M, C = 1000, 128
    z = torch.randn(M, C, generator=g) * 2 + 0.5
    gamma = torch.randn(C, generator=g)
    beta = torch.randn(C, generator=g)
    mm = torch.randn(C, generator=g)
    mv = torch.rand(C, generator=g) + 0.5
    part = torch.stack([z.sum(0), (z * z).sum(0)]).unsqueeze(0)
    mm_d, mv_d = mm.to(dev).clone(), mv.to(dev).clone()
    mean, invstd, scale, shift = ops.bn_finalize(part.to(dev), M, gamma.to(dev), beta.to(dev), mm_d, mv_d)
    zm = z.double().mean(0)
    zv = ((z.double() - zm) ** 2).mean(0)
    assert torch.allclose(mean.cpu().double(), zm, atol=1e-5)
    assert torch.allclose(invstd.cpu().double(), torch.rsqrt(zv + 1e-3), rtol=1e-4)
    assert torch.allclose(mm_d.cpu().double(), 0.99 * mm.double() + 0.01 * zm, atol=1e-5)
    assert torch.allclose(mv_d.cpu().double(), 0.99 * mv.double() + 0.01 * zv, atol=1e-4)
    # frozen / inference: coefficients from the moving statistics, moving statistics untouched
    mean2, invstd2, scale2, shift2 = ops.bn_finalize(None, M, gamma.to(dev), beta.to(dev), mm.to(dev), mv.to(dev),
                                                     use_batch_stats=False, update_moving=False)
    assert torch.allclose(scale2.cpu().double(), gamma.double() * torch.rsqrt(mv.double() + 1e-3), rtol=1e-5)
    assert torch.allclose(shift2.cpu().double(), beta.double() - mm.double() * gamma.double() * torch.rsqrt(mv.double() + 1e-3),
                          atol=1e-5)
    # backward coefficients: autograd through batch-norm as the reference
    zt = z.double().requires_grad_(True)
    gt = gamma.double().requires_grad_(True)
    bt = beta.double().requires_grad_(True)
    m_ = zt.mean(0)
    v_ = ((zt - m_) ** 2).mean(0)
    y = (zt - m_) * torch.rsqrt(v_ + 1e-3) * gt + bt
    dy = torch.randn(M, C, generator=g).double()
    y.backward(dy)
    bpart = torch.stack([dy.sum(0), (dy * z.double()).sum(0)]).unsqueeze(0).float()
    dgamma, dbeta, ca, cb, cc = ops.bn_bwd_finalize(bpart.to(dev), M, gamma.to(dev), mean, invstd)
    dz = ca.cpu().double() * dy + cb.cpu().double() * z.double() + cc.cpu().double()
    assert torch.allclose(dz, zt.grad, atol=2e-4, rtol=1e-3)
    assert torch.allclose(dgamma.cpu().double(), gt.grad, atol=2e-3, rtol=1e-3)
    assert torch.allclose(dbeta.cpu().double(), bt.grad, atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("B,N,M", [(2, 1000, 64), (1, 5000, 512), (3, 17, 17), (1, 40000, 256), (2, 20000, 100), (1, 21504, 40), (1, 21505, 40), (1, 16384, 40), (1, 16385, 40),
                                   (2, 1024, 300), (2, 1025, 100), (2, 4096, 200), (1, 4097, 100), (1, 131072, 64)])
def test_fps_bit_exact_indices(dev, B, N, M):
    ops = _ops()
    rng = np.random.default_rng(9)
    xyz = rng.uniform(-10, 10, size=(B, N, 3)).astype(np.float32)
    if N >= 1000:
        xyz[:, 500:520] = xyz[:, 100:120]           # duplicated points: distance ties -> lowest index
    if N >= 5000:                                   # ties inside one lane (i and i + block width), inside one wave, across waves
        xyz[:, 140 + 768] = xyz[:, 140]
        xyz[:, 141 + 1024] = xyz[:, 141]
        xyz[:, 150] = xyz[:, 151]
        xyz[:, 4000] = xyz[:, 152]
    idx, md = ops.farthest_point_sample(torch.from_numpy(xyz).to(dev), M, start_idx=0, return_mindist=True)
    for b in range(B):
        ri, rmd = SO.fps(xyz[b], M, 0)
        assert np.array_equal(idx[b].cpu().numpy(), ri), (b, np.flatnonzero(idx[b].cpu().numpy() != ri)[:5])
        assert np.array_equal(md[b].cpu().numpy(), rmd)


@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("N,M", [(8000, 2500), (20480, 1200), (4097, 4097), (12345, 300)])
def test_fps_pruned_kernel_on_spatially_ordered_clouds(dev, monkeypatch, N, M, mode):
    """the pruned FPS kernel (pn_sample.hip: fps_pruned_kernel, clouds of 4097 .. 20480 points; an experiment behind PN_FPS_PRUNE -- on the
    C5 scan its bounding-box test skips too little to pay, DESIGN.md section 7 -- mode 1 with the test, mode 2 its one-barrier round
    without it) in the regime it is built for: points in
    (z, y, x) order as the voxel grid leaves them, so that most 64-point groups are skipped in most rounds -- indices and final minimum
    distances bit-exact against the NumPy oracle, duplicated points (distance ties -> lowest index) inside one group, across groups of
    one wave and across waves, M = N (every point drawn once, then ties at distance 0)."""
    ops = _ops()
    monkeypatch.setenv("PN_FPS_PRUNE", mode)
    rng = np.random.default_rng(N + M)
    side = int(np.ceil(N ** (1.0 / 3.0))) + 1
    gi = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 3)[:N]     # z, y, x ascending
    xyz = (gi[:, ::-1] * 0.25 + rng.uniform(-0.1, 0.1, size=(N, 3))).astype(np.float32)
    xyz[70:75] = xyz[10:15]                         # ties inside a group / between groups of one wave
    xyz[3000:3004] = xyz[200:204]                   # ... between waves
    xyz[N - 3:] = xyz[N - 6:N - 3]
    idx, md = ops.farthest_point_sample(torch.from_numpy(xyz[None]).to(dev), M, start_idx=N // 2, return_mindist=True)
    ri, rmd = SO.fps(xyz, M, N // 2)
    assert np.array_equal(idx[0].cpu().numpy(), ri), np.flatnonzero(idx[0].cpu().numpy() != ri)[:5]
    assert np.array_equal(md[0].cpu().numpy(), rmd)


def test_scan_pipeline_c5_matches_oracle(dev):
    """BASELINE config 5 at full size (tools/bench_scan.py's synthetic scan): N = 131072 -> voxel grid 0.25 m -> FPS M = 8192, every
    stage bit-exact against the NumPy oracle; the sampled cloud then goes through PointNet.predict."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_scan", os.path.join(ROOT, "tools", "bench_scan.py"))
    bs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bs)
    ops = _ops()
    xyz, origin = bs.make_scan(131072)
    leaf = (0.25, 0.25, 0.25)
    cent, cnt, _ = ops.voxel_downsample(torch.from_numpy(xyz).to(dev), leaf, origin)
    rc, rn, _ = SO.voxel_downsample(xyz, leaf, origin)
    assert np.array_equal(cent.cpu().numpy(), rc) and np.array_equal(cnt.cpu().numpy(), rn)
    M = 8192
    assert rc.shape[0] > M
    idx = ops.farthest_point_sample(cent.unsqueeze(0).contiguous(), M)
    ri, _ = SO.fps(rc, M, 0)
    assert np.array_equal(idx[0].cpu().numpy(), ri)
    assert len(np.unique(ri)) == M                              # a property at full size: no point is drawn twice
    # ... and the sampled cloud through PointNet.predict against the oracle on the same (seeded) weights: class index bit-exact, part
    # index bit-exact wherever the oracle's top-2 margin exceeds the bf16 mode's tolerance
    from oracle import pointnet_oracle as O            # checker only
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    params = O.init_params(23, 12, seed=29, vanilla=True, randomize_bn=True)
    model = PointNet(23, 12, 0.3, 42, vanilla=True, precision="bf16", device=dev)
    model.set_weights(params)
    sampled = cent[idx[0].long()].unsqueeze(0).contiguous()
    ci, pi, _ = model.predict(sampled)
    assert tuple(ci.shape) == (1,) and tuple(pi.shape) == (1, M)
    ref = O.forward({k: v.double() for k, v in params.items()}, sampled.cpu().double(), training=False, vanilla=True)
    assert int(ci[0]) == int(ref[0].argmax(-1)[0])
    top2 = ref[1].topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 4e-3
    assert float(safe.double().mean()) > 0.9
    assert torch.equal(pi.cpu().long()[safe], ref[1].argmax(-1)[safe])


@pytest.mark.parametrize("B,N,K,C_", [(3, 1000, 128, 1024), (2, 200, 64, 1024), (2, 2048, 128, 2048), (1, 31, 128, 256)])
@pytest.mark.parametrize("store16", [False, True])
def test_maxbwd_scatter_is_exact_on_integer_operands(dev, B, N, K, C_, store16):
    """pn_maxbwd_scatter (the sparse arg-max term of the max-pooled layers' data gradient, PointNet.py:242-248 under tf.GradientTape):
    D[b][n] = q + sum of hs[b][c] * wt[c] over the channels whose maximum sits at row n.  Small-integer operands are exact in the split
    bf16 products and in the fp32 accumulator, so the result must equal the integer sum BIT FOR BIT whatever the kernel's order --
    ragged last tile, two 1024-channel chunks, a row that holds a third of a cloud's maxima, rows with none, both storage types."""
    from pointcloudprocessing_amd import ops
    g = torch.Generator().manual_seed(B * N + K + C_ + int(store16))
    arg = torch.randint(0, N, (B, C_), generator=g, dtype=torch.int32)
    arg[0, : C_ // 3] = N - 1                       # one heavy row in the (ragged) last tile of cloud 0
    arg[B - 1, C_ // 2:] = arg[B - 1, C_ // 2]      # ... and one in the last cloud: half of its channels on one row
    hs = torch.randint(-7, 8, (B, C_), generator=g).float()
    wt = torch.randint(-9, 10, (C_, K), generator=g).float()
    q = torch.randint(-5, 6, (K,), generator=g).float()
    D = ops.maxbwd_scatter(arg.to(dev), hs.to(dev), wt.to(dev), q.to(dev), B, N, K, C_, store16=store16)
    ref = q.double().expand(B * N, K).clone()
    contrib = hs.double()[:, :, None] * wt.double()[None, :, :]                     # (B, C, K)
    rows = (torch.arange(B)[:, None] * N + arg.long()).reshape(-1)
    ref.index_add_(0, rows, contrib.reshape(-1, K))
    want = ref.float().to(torch.bfloat16) if store16 else ref.float()
    assert float(ref.abs().max()) < 2.0 ** 24
    assert torch.equal(D.cpu(), want)


@pytest.mark.parametrize("front", ["xyz", "x64_plain", "x64_lazy"])
@pytest.mark.parametrize("B,N", [(3, 200), (32, 1024), (2, 4099)])
def test_chain_kernel_equals_the_layered_launches(dev, front, B, N):
    """pn_chain_fwd_max (inference: ConvLayer(3 | 64 -> 64) -> ConvLayer(64 -> 128) -> ConvLayer(128 -> 1024) -> reduce_max in ONE
    launch, PointNet.py:236-248 / 421-429 with moving statistics) against the three launches it replaces in the bf16-storage mode
    (pn_conv3_fwd | pn_conv_fwd, pn_conv_fwd, pn_conv_fwd_max_panel): per-slot maxima and 32-row blocks BIT FOR BIT -- same rounding
    points (the 16-bit stores of the two narrow layers, the bf16 MFMA operands), same contraction order.  Ragged clouds included."""
    ops = _ops()
    from pointcloudprocessing_amd import _lib
    prec = _lib.PREC["bf16"]
    g = torch.Generator().manual_seed(B * N + len(front))
    rnd = lambda *sh: torch.randn(*sh, generator=g)                  # noqa: E731
    w2, w3 = (rnd(64, 128) / 8).to(dev), (rnd(128, 1024) / 11).to(dev)
    gamma3 = rnd(1024).to(dev)                                        # both signs: the panel copy carries sign(gamma)
    sc1, sh1 = (torch.rand(64, generator=g) + 0.5).to(dev), (rnd(64) * 0.3).to(dev)
    sc2, sh2 = (torch.rand(128, generator=g) + 0.5).to(dev), (rnd(128) * 0.3).to(dev)
    wf = ops.weights_prep(w3, gamma3)
    _, w2t = ops.weights_copy16(w2)
    if front == "xyz":
        x3 = rnd(B * N, 3).to(dev)
        w1 = (rnd(3, 64) / 2).to(dev)
        z1, _ = ops.conv3_fwd(x3, w1, B, N, want_stats=False)
        z1 = z1.to(torch.bfloat16)                                    # the plan's 16-bit store (round to nearest even)
        fused = lambda: ops.chain_fwd_max(None, x3, w1, None, sc1, sh1, w2t, sc2, sh2, wf[0], B, N)     # noqa: E731
    else:
        x = (rnd(B * N, 64) * 3).to(dev).to(torch.bfloat16)
        w1 = (rnd(64, 64) / 8).to(dev)
        if front == "x64_lazy":
            ca, cc = (torch.rand(64, generator=g) + 0.5).to(dev), (rnd(64) * 0.3).to(dev)
            op0 = _lib.operand(x, ca=ca, cc=cc, relu=True)
        else:
            op0 = _lib.operand(x)
        z1, _ = ops.conv_fwd(op0, w1, B, N, 64, 64, prec, want_stats=False)
        assert z1.dtype == torch.bfloat16
        _, w1t = ops.weights_copy16(w1)
        fused = lambda: ops.chain_fwd_max(op0, None, None, w1t, sc1, sh1, w2t, sc2, sh2, wf[0], B, N)   # noqa: E731
    z2, _ = ops.conv_fwd(_lib.operand(z1, ca=sc1, cc=sh1, relu=True), w2, B, N, 64, 128, prec, want_stats=False)
    pmax_ref, pblk_ref, _, _ = ops.conv_fwd_max_panel(_lib.operand(z2, ca=sc2, cc=sh2, relu=True), wf, B, N, 128, 1024, prec, want_stats=False)
    pmax, pblk = fused()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(pmax).all()) and float(pmax.abs().max()) > 0
    assert torch.equal(pmax, pmax_ref), float((pmax - pmax_ref).abs().max())
    assert torch.equal(pblk, pblk_ref)


def test_voxel_downsample_matches_oracle(dev):
    ops = _ops()
    rng = np.random.default_rng(10)
    N = 20000
    xyz = rng.uniform(0, 10, size=(N, 3)).astype(np.float32)
    xyz[1000:1100] = xyz[0:100]                      # exact duplicates share a voxel
    labels = rng.integers(0, 12, size=N).astype(np.int32)
    leaf, origin = (0.5, 0.5, 0.25), (-1.0, -1.0, -1.0)
    cent, cnt, maj = ops.voxel_downsample(torch.from_numpy(xyz).to(dev), leaf, origin, torch.from_numpy(labels).to(dev), 12)
    rc, rn, rm = SO.voxel_downsample(xyz, leaf, origin, labels, 12)
    assert cent.shape[0] == rc.shape[0]
    assert np.array_equal(cnt.cpu().numpy(), rn)
    assert np.array_equal(maj.cpu().numpy(), rm)
    assert np.array_equal(cent.cpu().numpy(), rc)
    assert int(cnt.sum()) == N
    # idempotence: downsampling the centroids with the same grid keeps one point per voxel
    c2, n2, _ = ops.voxel_downsample(cent.contiguous(), leaf, origin)
    assert c2.shape[0] == cent.shape[0] and int(n2.max()) == 1


def test_invalid_arguments_raise(dev):
    ops = _ops()
    L = _lib()
    x = torch.zeros(128, 48, device=dev)
    w = torch.zeros(48, 64, device=dev)
    with pytest.raises(L.PointNetHipError):
        ops.conv_fwd(L.operand(x), w, 1, 128, 48, 64, 1)          # K not a multiple of 64
    with pytest.raises(L.PointNetHipError):
        ops.farthest_point_sample(torch.zeros(1, 10, 3, device=dev), 4, start_idx=10)
    with pytest.raises(L.PointNetHipError):
        ops.normalize(torch.zeros(1, 10, 3))                      # CPU tensor: no CPU fallback


# ---------------------------------------------------------------------------------------------------------------------
# DenseLayer in one launch (split-K blocks meeting in-launch) and its fused backward
# ---------------------------------------------------------------------------------------------------------------------
def _dense_ref(x, w, bias, gamma, beta, mm, mv, bn_mode, act, keep, rate, momentum=0.99, eps=1e-3):
    """fp64 restatement of keras Dense -> BatchNormalization -> ReLU -> Dropout (pointnet/PointNet.py:597-679)"""
    z = x.double() @ w.double()
    if bias is not None:
        z = z + bias.double()
    mean = invstd = None
    y = z
    if bn_mode == 1:
        mean = z.mean(0); var = z.var(0, unbiased=False)
        invstd = 1.0 / torch.sqrt(var + eps)
        y = (z - mean) * invstd * gamma.double() + beta.double()
        mm_new = mm.double() * momentum + mean * (1 - momentum); mv_new = mv.double() * momentum + var * (1 - momentum)
    elif bn_mode == 2:
        mean = mm.double(); invstd = 1.0 / torch.sqrt(mv.double() + eps)
        y = (z - mean) * invstd * gamma.double() + beta.double()
        mm_new, mv_new = mm.double(), mv.double()
    else:
        mm_new = mv_new = None
    if act:
        y = torch.relu(y)
    if keep is not None:
        y = torch.where(keep.bool(), y / (1.0 - rate), torch.zeros_like(y))
    return z, y, mean, invstd, mm_new, mv_new


@pytest.mark.parametrize("R,K,C_,trans,bn_mode,act,drop", [
    (32, 1024, 512, False, 1, 1, True), (32, 512, 256, False, 1, 1, False), (32, 256, 23, False, 0, 0, False),
    (32, 256, 4096, False, 0, 0, False), (5, 256, 9, False, 0, 0, False), (7, 300, 70, False, 2, 1, True),
    (32, 512, 1024, True, 0, 0, False), (32, 23, 256, True, 0, 0, False), (3, 9, 256, True, 0, 0, False),
    (45, 1024, 96, False, 1, 1, True), (33, 130, 40, True, 0, 0, False),
])
def test_dense_layer_matches_fp64_reference(dev, R, K, C_, trans, bn_mode, act, drop):
    g = torch.Generator().manual_seed(R * 7919 + K * 31 + C_)
    x = torch.randn(R, K, generator=g).to(dev)
    w = (torch.randn(K, C_, generator=g) * 0.05).to(dev)
    bias = None if bn_mode else torch.randn(C_, generator=g).to(dev)
    gamma = (torch.rand(C_, generator=g) + 0.5).to(dev); beta = torch.randn(C_, generator=g).to(dev)
    mm = torch.randn(C_, generator=g).to(dev); mv = (torch.rand(C_, generator=g) + 0.5).to(dev)
    keep = (torch.rand(R, C_, generator=g) > 0.3).to(torch.uint8).to(dev) if drop else None
    zr, yr, mr, ir, mmr, mvr = _dense_ref(x, w, bias, gamma, beta, mm, mv, bn_mode, act, keep, 0.3)
    wdev = w.t().contiguous() if trans else w
    cnt = torch.zeros(256, device=dev, dtype=torch.int32)
    mm_k, mv_k = mm.clone(), mv.clone()
    for rep in range(2):           # the second call checks that the arrival counters were left at zero
        if rep:
            mm_k, mv_k = mm.clone(), mv.clone()
        z, a, mean, invstd = _ops().dense_layer(x, wdev, trans=trans, bias=bias, gamma=gamma if bn_mode else None, beta=beta if bn_mode else None,
                                             moving_mean=mm_k if bn_mode else None, moving_var=mv_k if bn_mode else None, bn_mode=bn_mode,
                                             act=act, keep=keep, rate=0.3, counters=cnt)
        torch.cuda.synchronize()
        assert int(cnt.abs().sum()) == 0
        # products run as bf16 hi/lo splits on the matrix cores (3 of the 4 cross terms): ~2^-16 relative per product
        assert torch.allclose(z.double(), zr, rtol=1e-4, atol=1e-4), float((z.double() - zr).abs().max())
        assert torch.allclose(a.double(), yr, rtol=3e-4, atol=3e-4), float((a.double() - yr).abs().max())
        if bn_mode:
            assert torch.allclose(mean.double(), mr, rtol=1e-4, atol=1e-5) and torch.allclose(invstd.double(), ir, rtol=1e-4, atol=1e-5)
            assert torch.allclose(mm_k.double(), mmr, rtol=1e-5, atol=1e-6) and torch.allclose(mv_k.double(), mvr, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("R,K,C_,bn_mode,act,drop", [(32, 1024, 512, 1, 1, True), (32, 256, 23, 0, 0, False), (6, 130, 300, 1, 1, False),
                                                     (17, 64, 70, 2, 1, True)])
def test_dense_bwd_matches_autograd(dev, R, K, C_, bn_mode, act, drop):
    g = torch.Generator().manual_seed(R + K + C_)
    x = torch.randn(R, K, generator=g, dtype=torch.float64)
    w = (torch.randn(K, C_, generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    bias = torch.randn(C_, generator=g, dtype=torch.float64).requires_grad_(True)
    gamma = (torch.rand(C_, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = torch.randn(C_, generator=g, dtype=torch.float64).requires_grad_(True)
    mm = torch.randn(C_, generator=g, dtype=torch.float64); mv = torch.rand(C_, generator=g, dtype=torch.float64) + 0.5
    keep = (torch.rand(R, C_, generator=g) > 0.3).to(torch.uint8) if drop else None
    da = torch.randn(R, C_, generator=g, dtype=torch.float64)
    z, y, mean, invstd, _, _ = _dense_ref(x, w, None if bn_mode else bias, gamma, beta, mm, mv, bn_mode, act, keep, 0.3)
    zf = z.detach().float()
    yf = (zf.double() - (mean if mean is not None else 0)) * (invstd * gamma if invstd is not None else 1) + (beta if bn_mode else 0)
    if act and bool(((yf.abs() < 1e-4)).any()):
        pytest.skip("a pre-activation sits on the ReLU boundary")
    y.backward(da)
    f = lambda t: None if t is None else t.detach().float().to(dev).contiguous()
    dz, dg, db, dbias, dw = _ops().dense_bwd(f(da), f(z), f(x), gamma=f(gamma) if bn_mode else None, beta=f(beta) if bn_mode else None,
                                          mean=f(mean), invstd=f(invstd), bn_mode=bn_mode, act=act, keep=None if keep is None else keep.to(dev), rate=0.3)
    assert torch.allclose(dw.double().cpu(), w.grad, rtol=1e-4, atol=1e-4), float((dw.double().cpu() - w.grad).abs().max())
    if bn_mode == 1:
        assert torch.allclose(dg.double().cpu(), gamma.grad, rtol=1e-4, atol=1e-4) and torch.allclose(db.double().cpu(), beta.grad, rtol=1e-4, atol=1e-4)
    if bn_mode == 0:
        assert torch.allclose(dbias.double().cpu(), bias.grad, rtol=1e-4, atol=1e-4)
    # dz: check through dx = dz . W^T against autograd's input gradient
    x2 = x.clone().requires_grad_(True)
    _, y2, _, _, _, _ = _dense_ref(x2, w.detach(), None if bn_mode else bias.detach(), gamma.detach(), beta.detach(), mm, mv, bn_mode, act, keep, 0.3)
    y2.backward(da)
    dx = dz.double().cpu() @ w.detach().t()
    assert torch.allclose(dx, x2.grad, rtol=1e-4, atol=1e-4), float((dx - x2.grad).abs().max())


@pytest.mark.parametrize("R,Kin,C_,Cup,bn_mode,act,drop", [(32, 256, 512, 256, 1, 1, True), (7, 64, 96, 23, 1, 1, False), (32, 128, 256, 9, 2, 1, False),
                                                        (16, 64, 64, 40, 0, 0, False)])
def test_dense_backward_chain_step_equals_the_two_launch_form(dev, R, Kin, C_, Cup, bn_mode, act, drop):
    """The model plan takes a chain of dense layers backward with ONE launch per layer (pn_dense_bwd_step: dx = dz_above . W_above^T and
    the layer below's dropout / ReLU / BatchNormalization backward in its finishing workgroups) and one batched weight-gradient launch
    (pn_dense_wgrad_batch).  Against the two-launch form (pn_dense_layer(trans) then pn_dense_bwd, themselves checked against fp64
    autograd above): dx is the same kernel -> bit-identical; dz, dgamma, dbeta / dbias differ only by the order of the column sums;
    dw is an exact fp32 fma chain where pn_dense_bwd multiplies split bf16 operands."""
    ops = _ops()
    g = torch.Generator().manual_seed(R * 7 + C_)
    x = torch.randn(R, Kin, generator=g).to(dev)                       # input of the layer below
    z = torch.randn(R, C_, generator=g).to(dev)                        # its stored pre-BN output
    gamma = (torch.rand(C_, generator=g) + 0.5).to(dev); beta = torch.randn(C_, generator=g).to(dev)
    mean = z.mean(0) if bn_mode == 1 else torch.randn(C_, generator=g).to(dev) * 0.1
    invstd = torch.rsqrt(z.var(0, unbiased=False) + 1e-3) if bn_mode == 1 else (torch.rand(C_, generator=g).to(dev) + 0.5)
    keep = (torch.rand(R, C_, generator=g) > 0.3).to(torch.uint8).to(dev) if drop else None
    w_up = (torch.randn(C_, Cup, generator=g) * 0.1).to(dev)           # kernel of the layer above: (cin = C_, cout = Cup)
    dz_up = torch.randn(R, Cup, generator=g).to(dev)
    bn = dict(gamma=gamma if bn_mode else None, beta=beta if bn_mode else None, mean=mean if bn_mode else None, invstd=invstd if bn_mode else None)
    # two-launch form
    da, _, _, _ = ops.dense_layer(dz_up, w_up, trans=True)
    dz_ref, dg_ref, db_ref, dbias_ref, dw_ref = ops.dense_bwd(da, z, x, bn_mode=bn_mode, act=act, keep=keep, rate=0.3, **bn)
    # chain form
    dx, dz, dg, db, dbias = ops.dense_bwd_step(dz_up, w_up, z=z, bn_mode=bn_mode, act=act, keep=keep, rate=0.3, **bn)
    (dw, dbcol), (dw_up, db_up) = ops.dense_wgrad_batch([(x, dz, True), (torch.relu(z), dz_up, True)])
    assert torch.equal(dx, da)
    tol = dict(rtol=2e-5, atol=2e-5 * float(dz_ref.abs().max()))
    assert torch.allclose(dz, dz_ref, **tol), float((dz - dz_ref).abs().max())
    if bn_mode == 1:
        assert torch.allclose(dg, dg_ref, rtol=1e-5, atol=1e-4) and torch.allclose(db, db_ref, rtol=1e-5, atol=1e-4)
    if bn_mode == 0:
        assert torch.allclose(dbias, dbias_ref, rtol=1e-5, atol=1e-4)
    ref_dw = x.double().t() @ dz.double()
    assert torch.allclose(dw.double(), ref_dw, rtol=1e-5, atol=1e-5 * float(ref_dw.abs().max())), float((dw.double() - ref_dw).abs().max())
    assert torch.allclose(dw, dw_ref, rtol=2e-3, atol=2e-4 * float(ref_dw.abs().max()))
    assert torch.allclose(dbcol.double(), dz.double().sum(0), rtol=1e-5, atol=1e-5)
    ref_up = torch.relu(z).double().t() @ dz_up.double()
    assert torch.allclose(dw_up.double(), ref_up, rtol=1e-5, atol=1e-5 * float(ref_up.abs().max()))
    assert torch.allclose(db_up.double(), dz_up.double().sum(0), rtol=1e-5, atol=1e-5)
    # no tail: the plain product
    dx2, none_dz, _, _, _ = ops.dense_bwd_step(dz_up, w_up)
    assert none_dz is None and torch.equal(dx2, da)


def test_dropout_masks_counter_based(dev):
    """keep-probability, independence of the two layers, fresh masks per call, reproducible from (seed, step)"""
    import ctypes as C
    L = _lib()
    B = 64
    k1 = torch.zeros(B, 512, dtype=torch.uint8, device=dev); k2 = torch.zeros(B, 256, dtype=torch.uint8, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)

    def draw(seed):
        L.check(L.lib().pn_dropout_masks(L.ptr(k1), k1.numel(), L.ptr(k2), k2.numel(), 0.3, seed, L.ptr(step), L.current_stream()), "pn_dropout_masks")
        torch.cuda.synchronize()
        return k1.clone(), k2.clone()
    a1, a2 = draw(1234)
    b1, b2 = draw(1234)
    assert int(step) == 2 and set(a1.unique().tolist()) <= {0, 1}
    assert abs(float(a1.float().mean()) - 0.7) < 0.02 and abs(float(a2.float().mean()) - 0.7) < 0.03
    assert 0.35 < float((a1 != b1).float().mean()) < 0.49          # 2 * 0.3 * 0.7 = 0.42 when independent
    assert 0.35 < float((a1[:, :256] != a2).float().mean()) < 0.49
    step.zero_()
    c1, c2 = draw(1234)
    assert torch.equal(a1, c1) and torch.equal(a2, c2)
    step.zero_()
    d1, _ = draw(99)
    assert not torch.equal(a1, d1)


# ---------------------------------------------------------------------------------------------------------------------
# classification loss, segmentation output layer, per-cloud matmul: the remaining SURVEY 8(b) exports
@pytest.mark.parametrize("shape", [(32, 23), (4, 1000, 12), (1, 1), (70000, 3), (5, 300)])
def test_argmax_rows_is_first_maximum(dev, shape):
    ops = _ops()
    g = torch.Generator().manual_seed(sum(shape))
    v = torch.randint(0, 6, shape, generator=g).float()        # few distinct values: most rows hold ties
    v[..., -1:] += (torch.rand(shape[:-1] + (1,), generator=g) > 0.5).float() * 7   # and some rows peak in the last column
    got = ops.argmax_rows(v.to(dev))
    assert got.dtype == torch.int32 and tuple(got.shape) == tuple(shape[:-1])
    assert np.array_equal(got.cpu().numpy(), np.argmax(v.numpy(), axis=-1))
    with pytest.raises(_lib().PointNetHipError):
        ops.argmax_rows(v)                                      # CPU tensor: no CPU fallback


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,C_", [(32, 23), (5, 2), (70, 40), (32, 300)])
def test_softmax_xent_matches_keras_restatement(dev, R, C_):
    L = _lib()
    g = torch.Generator().manual_seed(R * 100 + C_)
    logits = (torch.randn(R, C_, generator=g) * 3).double()
    logits[0, 0] = 60.0                                    # one saturated row: exercises the clip to [1e-7, 1-1e-7]
    labels = torch.randint(0, C_, (R,), generator=g)
    lg = logits.clone().requires_grad_(True)
    probs = torch.softmax(lg, -1)
    loss = O.keras_sparse_cce(probs, labels) * R           # sum over rows
    loss.backward()
    ld = logits.float().to(dev)
    lab_d = labels.int().to(dev)                           # named: a temporary would be freed before the kernel runs
    p = torch.empty(R, C_, device=dev); d = torch.empty(R, C_, device=dev); ls = torch.zeros(1, device=dev); cr = torch.zeros(1, device=dev)
    L.check(L.lib().pn_softmax_xent(L.ptr(ld), R, C_, L.ptr(lab_d), 0.5, L.ptr(p), L.ptr(d), L.ptr(ls), L.ptr(cr),
                                    L.current_stream()), "pn_softmax_xent")
    torch.cuda.synchronize()
    assert torch.allclose(p.double().cpu(), probs.detach(), rtol=1e-5, atol=1e-7)
    assert torch.equal(p.argmax(-1).cpu(), probs.argmax(-1))
    assert abs(float(ls) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    assert float(cr) == float((probs.argmax(-1) == labels).sum())
    assert torch.allclose(d.double().cpu(), 0.5 * lg.grad, rtol=2e-4, atol=2e-6), float((d.double().cpu() - 0.5 * lg.grad).abs().max())


@pytest.mark.parametrize("B,N,C_", [(2, 300, 12), (3, 257, 5)])
def test_seg_out_fwd_matches_restatement(dev, B, N, C_):
    L = _lib()
    K, M = 128, B * N
    g = torch.Generator().manual_seed(B * N + C_)
    z = torch.randn(M, K, generator=g); ca = torch.rand(K, generator=g) + 0.5; cc = torch.randn(K, generator=g) * 0.3
    w = torch.randn(K, C_, generator=g) * 0.2; bias = torch.randn(C_, generator=g)
    labels = torch.randint(0, C_, (M,), generator=g)
    x = torch.relu(z.double() * ca.double() + cc.double())
    lg = (x @ w.double() + bias.double()).requires_grad_(True)
    probs = torch.softmax(lg, -1)
    nll_sum = O.keras_sparse_cce(probs, labels) * M
    nll_sum.backward()
    zd, cad, ccd = z.to(dev), ca.to(dev), cc.to(dev)
    wd, bd, lab_d = w.to(dev), bias.to(dev), labels.int().to(dev)
    op = L.operand(zd, ca=cad, cc=ccd, ld=K, relu=True)
    stride = L.lib().pn_seg_out_part_stride()
    rpb = L.lib().pn_seg_out_part_rows()
    nparts = (M + rpb - 1) // rpb
    p = torch.empty(M, C_, device=dev); d = torch.empty(M, C_, device=dev); part = torch.zeros(nparts * stride, device=dev)
    L.check(L.lib().pn_seg_out_fwd(C.byref(op), L.ptr(wd), L.ptr(bd), M, K, C_, L.ptr(lab_d), 1.0 / M,
                                   L.ptr(p), L.ptr(d), L.ptr(part), L.current_stream()), "pn_seg_out_fwd")
    torch.cuda.synchronize()
    assert torch.allclose(p.double().cpu(), probs.detach(), rtol=1e-4, atol=1e-6)
    pr = part.view(nparts, stride).double().cpu()
    assert abs(float(pr[:, 0].sum()) - float(nll_sum)) <= 1e-4 * float(nll_sum)
    safe = (probs.detach().topk(2, -1).values.diff(dim=-1).abs().squeeze(-1) > 1e-5)
    assert abs(float(pr[:, 1].sum()) - float((probs.argmax(-1) == labels).sum())) <= float((~safe).sum())
    assert torch.allclose(d.double().cpu(), lg.grad / M, rtol=1e-3, atol=1e-7)


@pytest.mark.parametrize("K", [3, 64])
def test_bmm_per_cloud_matrices(dev, K):
    L = _lib()
    B, N = 3, 200
    g = torch.Generator().manual_seed(K)
    x = ints(g, (B, N, K)); R = ints(g, (B, K, K), -2, 3)       # integer-valued: exact in bf16 and in fp32
    out = torch.empty(B * N, K, device=dev)
    xd, Rd = x.to(dev), R.to(dev)
    L.check(L.lib().pn_bmm(L.ptr(xd), L.ptr(Rd), B, N, K, L.ptr(out), 1, L.current_stream()), "pn_bmm")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu().view(B, N, K), torch.bmm(x, R))


# ------------------------------------------------------------------------------------------------
# bf16 storage of the per-point tensors (pn_operand.h16, PN_STORE_BF16): fed bf16-representable values, every kernel must give
# bit for bit what it gives from the same values stored as fp32 (the 16-bit path changes loads and stores, never the arithmetic),
# and a bf16 output is the round-to-nearest-even of the fp32 output.
def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("prec", [1, 3])
@pytest.mark.parametrize("B,N,K,C", [(2, 300, 64, 128), (1, 130, 128, 64), (3, 64, 512, 256)])
def test_bf16_storage_conv_fwd(dev, prec, B, N, K, C):
    ops, L = _ops(), _lib()
    g = torch.Generator().manual_seed(31)
    x = _bf(torch.randn(B * N, K, generator=g)).to(dev)
    ca = (torch.rand(K, generator=g) + 0.5).to(dev); cc = (torch.randn(K, generator=g) * 0.3).to(dev)
    w = (torch.randn(K, C, generator=g) * 0.2).to(dev)
    bias = torch.randn(B, C, generator=g).to(dev)
    z32, p32 = ops.conv_fwd(L.operand(x.float(), ca=ca, cc=cc, relu=True), w, B, N, K, C, prec, cloud_bias=bias)
    z16, p16 = ops.conv_fwd(L.operand(x, ca=ca, cc=cc, relu=True), w, B, N, K, C, prec | L.PN_STORE_BF16, cloud_bias=bias)
    assert z16.dtype == torch.bfloat16
    assert torch.equal(z16, _bf(z32))
    assert torch.equal(p16, p32)                 # statistics come from the fp32 values before rounding
    # mixed: 16-bit source, fp32 store
    z_mixed, _ = ops.conv_fwd(L.operand(x, ca=ca, cc=cc, relu=True), w, B, N, K, C, prec, cloud_bias=bias)
    assert torch.equal(z_mixed, z32)


@pytest.mark.parametrize("prec", [1, 3])
def test_bf16_storage_conv_bwd_data(dev, prec):
    ops, L = _ops(), _lib()
    g = torch.Generator().manual_seed(32)
    B, N, K, C = 2, 200, 128, 64
    dy = _bf(torch.randn(B * N, K, generator=g)).to(dev); zz = _bf(torch.randn(B * N, K, generator=g)).to(dev)
    ca, cb, cc = ((torch.randn(K, generator=g) * 0.5).to(dev) for _ in range(3))
    w = (torch.randn(C, K, generator=g) * 0.2).to(dev)
    addend = _bf(torch.randn(B * N, C, generator=g)).to(dev); zprev = _bf(torch.randn(B * N, C, generator=g)).to(dev)
    msc = torch.randn(C, generator=g).to(dev); msh = (torch.randn(C, generator=g) * 0.2).to(dev)
    o32, p32 = ops.conv_bwd_data(L.operand(dy.float(), ca=ca, cc=cc, s2=zz.float(), cb=cb), w, B, N, K, C, prec, addend=addend.float(),
                                 zmask=zprev.float(), msc=msc, msh=msh)
    o16, p16 = ops.conv_bwd_data(L.operand(dy, ca=ca, cc=cc, s2=zz, cb=cb), w, B, N, K, C, prec | L.PN_STORE_BF16, addend=addend,
                                 zmask=zprev, msc=msc, msh=msh)
    assert o16.dtype == torch.bfloat16 and torch.equal(o16, _bf(o32)) and torch.equal(p16, p32)
    with pytest.raises(L.PointNetHipError):      # storage types must match the flag
        ops.conv_bwd_data(L.operand(dy, ca=ca, cc=cc, s2=zz, cb=cb), w, B, N, K, C, prec | L.PN_STORE_BF16, addend=addend.float())


@pytest.mark.parametrize("prec", [1, 3])
@pytest.mark.parametrize("Ci,Cj", [(64, 64), (64, 128), (128, 128), (64, 512), (512, 256)])
def test_bf16_storage_conv_wgrad(dev, prec, Ci, Cj):
    ops, L = _ops(), _lib()
    g = torch.Generator().manual_seed(33)
    B, N = 2, 300
    a = _bf(torch.randn(B * N, Ci, generator=g)).to(dev)
    dy = _bf(torch.randn(B * N, Cj, generator=g)).to(dev); zz = _bf(torch.randn(B * N, Cj, generator=g)).to(dev)
    sa, sh = (torch.rand(Ci, generator=g) + 0.5).to(dev), (torch.randn(Ci, generator=g) * 0.2).to(dev)
    ca, cb, cc = ((torch.randn(Cj, generator=g) * 0.5).to(dev) for _ in range(3))
    for slab_rows in (128, 64):
        ref = ops.conv_wgrad(L.operand(a.float(), ca=sa, cc=sh, relu=True), L.operand(dy.float(), ca=ca, cc=cc, s2=zz.float(), cb=cb), B, N, Ci, Cj,
                             prec, slab_rows=slab_rows)
        got = ops.conv_wgrad(L.operand(a, ca=sa, cc=sh, relu=True), L.operand(dy, ca=ca, cc=cc, s2=zz, cb=cb), B, N, Ci, Cj, prec,
                             slab_rows=slab_rows)
        assert torch.equal(got, ref), float((got - ref).abs().max())
    # one 16-bit operand, one fp32 operand
    got = ops.conv_wgrad(L.operand(a, ca=sa, cc=sh, relu=True), L.operand(dy.float(), ca=ca, cc=cc, s2=zz.float(), cb=cb), B, N, Ci, Cj, prec,
                         slab_rows=64)
    assert torch.equal(got, ref)


def test_bf16_storage_conv3_wgrad_and_seg_out(dev):
    ops, L = _ops(), _lib()
    g = torch.Generator().manual_seed(34)
    B, N, C_ = 3, 257, 64
    x3 = torch.randn(B * N, 3, generator=g).to(dev)
    dy = _bf(torch.randn(B * N, C_, generator=g)).to(dev); zz = _bf(torch.randn(B * N, C_, generator=g)).to(dev)
    ca, cb, cc = ((torch.randn(C_, generator=g) * 0.5).to(dev) for _ in range(3))
    ref = ops.conv3_wgrad(x3, L.operand(dy.float(), ca=ca, cc=cc, s2=zz.float(), cb=cb), B, N, C_)
    got = ops.conv3_wgrad(x3, L.operand(dy, ca=ca, cc=cc, s2=zz, cb=cb), B, N, C_)
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-4)      # same values, another summation order (row pairs across the half-waves)
    # segmentation output layer (K = 128: the MFMA form)
    K, M, CS = 128, B * N, 12
    z = _bf(torch.randn(M, K, generator=g)).to(dev)
    sa, sh = (torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.3).to(dev)
    w = (torch.randn(K, CS, generator=g) * 0.2).to(dev); bias = torch.randn(CS, generator=g).to(dev)
    lab = torch.randint(0, CS, (M,), generator=g).int().to(dev)
    stride, rpb = L.lib().pn_seg_out_part_stride(), L.lib().pn_seg_out_part_rows()
    nparts = (M + rpb - 1) // rpb
    res = []
    for src in (z.float(), z):
        op = L.operand(src, ca=sa, cc=sh, ld=K, relu=True)
        p = torch.empty(M, CS, device=dev); d = torch.empty(M, CS, device=dev); part = torch.zeros(nparts * stride, device=dev)
        L.check(L.lib().pn_seg_out_fwd(C.byref(op), L.ptr(w), L.ptr(bias), M, K, CS, L.ptr(lab), 1.0 / M, L.ptr(p), L.ptr(d), L.ptr(part),
                                       L.current_stream()), "pn_seg_out_fwd")
        res.append((p, d, part))
    for a_, b_ in zip(res[0], res[1]):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("B,N", [(2, 256), (3, 200), (16, 136), (40, 520)])
def test_bf16_storage_panel_kernel_and_resolve(dev, B, N):
    ops, L = _ops(), _lib()
    g = torch.Generator().manual_seed(35 + N)
    K, C_ = 128, 1024
    x = _bf(torch.randn(B * N, K, generator=g)).to(dev)
    ca = (torch.rand(K, generator=g) + 0.5).to(dev); cc = (torch.randn(K, generator=g) * 0.3).to(dev)
    w = (torch.randn(K, C_, generator=g) * 0.1).to(dev); gamma = torch.randn(C_, generator=g).to(dev)
    wf = ops.weights_prep(w, gamma)
    res = []
    for src in (x.float(), x):
        op = L.operand(src, ca=ca, cc=cc, relu=True)
        pmax, pblk, sumsq, sumz = ops.conv_fwd_max_panel(op, wf, B, N, K, C_, 1)
        beta = torch.zeros(C_, device=dev); mm = torch.zeros(C_, device=dev); mv = torch.ones(C_, device=dev)
        fin = ops.panel_finalize(pmax, pblk, sumsq, sumz, wf, 1, B, N, K, gamma, beta, mm, mv, training=True)
        arg = ops.max_resolve(op, wf, fin[-1], B, N, K, C_, 1)
        res.append((pmax, pblk, sumsq, sumz, arg))
    for a_, b_ in zip(res[0], res[1]):
        assert torch.equal(a_, b_)
    with pytest.raises(L.PointNetHipError):      # the panel kernel takes 16-bit sources in the bf16 mode only
        ops.conv_fwd_max_panel(L.operand(x, ca=ca, cc=cc, relu=True), wf, B, N, K, C_, 3)
