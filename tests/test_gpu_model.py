"""GPU parity of the whole PointNet path (forward, losses, backward, moving statistics) against the CPU
oracle on identical injected weights and inputs.

PARITY UNPINNED against the reference itself (no TF here, no golden vectors in the reference): the
checker is oracle/pointnet_oracle.py, a line-by-line restatement of pointnet/PointNet.py.

Tolerances (stated per test): 'bf16x3' mode carries 16 significant bits per MFMA operand -> probabilities
within 2e-4 of the fp64 oracle, argmax identical; 'bf16' mode is compared with the oracle run with the
same bf16 operand rounding (tight) and with the fp32 oracle (loose, 3e-2).
"""
import os

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pointnet_oracle as O   # noqa: E402  (checker only)

from parity_harness import (CCLS, CSEG, PROFILES, build_model, check_training_step, make_inputs, rel_err,   # noqa: E402
                            report as _report)


@pytest.mark.parametrize("vanilla", [False, True])
@pytest.mark.parametrize("B,N", [(4, 200), (2, 1024)])
def test_inference_forward_matches_oracle(dev, vanilla, B, N):
    params = O.init_params(CCLS, CSEG, seed=11, vanilla=vanilla, randomize_bn=True)
    pc, *_ = make_inputs(B, N, 5)
    ref = O.forward({k: v.double() for k, v in params.items()}, pc.double(), training=False, vanilla=vanilla)
    m = build_model(dev, params, vanilla)
    cls, seg, R = m(pc.to(dev), training=False)
    e = [float((cls.cpu().double() - ref[0]).abs().max()), float((seg.cpu().double() - ref[1]).abs().max()),
         float((R.cpu().double() - ref[2]).abs().max())]
    _report(f"inference vanilla={vanilla} B={B} N={N}: max abs err cls={e[0]:.3e} seg={e[1]:.3e} R={e[2]:.3e}")
    assert e[0] < 2e-4 and e[1] < 2e-4 and e[2] < 2e-4, e
    assert torch.equal(cls.cpu().argmax(-1), ref[0].argmax(-1))
    # per-point part indices: identical wherever the oracle's own top-2 margin exceeds the error bound
    top2 = ref[1].topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3
    assert torch.equal(seg.cpu().argmax(-1)[safe], ref[1].argmax(-1)[safe])
    assert float(safe.double().mean()) > 0.5


@pytest.mark.parametrize("profile", ["all", "classification_pretrain", "final", "heads_only"])
@pytest.mark.parametrize("vanilla", [False, True])
def test_training_step_gradients_match_oracle(dev, profile, vanilla):
    # batch-statistics BatchNormalization over the B rows of the T-Net dense layers amplifies rounding differences by
    # ~1/sqrt(var+eps): B=4 is ill-conditioned (the fp32 oracle itself is 2e-4 away from fp64), so the T-Net cases use B=16
    B, N = (4, 200) if vanilla else (16, 136)
    check_training_step(dev, B, N, profile, vanilla=vanilla, precision="bf16x3", tag=f"train[{profile},vanilla={vanilla}]")


def test_autograd_path_equals_fused_loss_path(dev):
    """torch losses on the three outputs + loss.backward() must give the same gradients as the fused native loss."""
    B, N = 2, 256
    params = O.init_params(CCLS, CSEG, seed=13, randomize_bn=True)
    pc, y_cls, y_seg, se3, keep = make_inputs(B, N, 7)
    kp = (keep["dropout_1"].to(torch.uint8).to(dev), keep["dropout_2"].to(torch.uint8).to(dev))
    m1 = build_model(dev, params)
    m1.fused_loss_step(pc.to(dev), y_cls.to(torch.int32).to(dev), y_seg.to(torch.int32).to(dev), se3.to(dev), (1.0, 1.0, 1.0), keep=kp)
    g1 = m1.grads_flat.clone()
    m2 = build_model(dev, params)
    m2._dropout_rate = 0.3
    m2._fixed_keep = kp
    orig_io = m2._io

    def io_with_keep(pc_, training, fused):
        return orig_io(pc_, training, dict(keep=kp) if training else fused)
    m2._io = io_with_keep
    cls, seg, R = m2(pc.to(dev), training=True)
    yc, ys = y_cls.to(dev), y_seg.to(dev)
    loss = (O.keras_sparse_cce(cls, yc) + O.keras_sparse_cce(seg, ys) + O.keras_mse(R, se3.to(dev)))
    loss.backward()
    g2 = m2.params_flat.grad
    assert g2 is not None
    e = float((g1 - g2).abs().max() / (g1.abs().max() + 1e-12))
    assert e < 1e-4, e


def test_bf16_mode_against_bf16_emulating_oracle(dev):
    B, N = 4, 256
    params = O.init_params(CCLS, CSEG, seed=14, randomize_bn=True)
    pc, *_ = make_inputs(B, N, 8)
    ref_q = O.forward(params, pc, training=False, quant=O.bf16_round)
    ref_f = O.forward({k: v.double() for k, v in params.items()}, pc.double(), training=False)
    m = build_model(dev, params, precision="bf16")
    cls, seg, R = m(pc.to(dev), training=False)
    e_q = [float((cls.cpu() - ref_q[0]).abs().max()), float((seg.cpu() - ref_q[1]).abs().max()), float((R.cpu() - ref_q[2]).abs().max())]
    e_f = [float((cls.cpu().double() - ref_f[0]).abs().max()), float((seg.cpu().double() - ref_f[1]).abs().max()),
           float((R.cpu().double() - ref_f[2]).abs().max())]
    _report(f"bf16 mode: vs bf16-emulating oracle {e_q}; vs fp64 oracle {e_f}")
    assert max(e_q) < 5e-3, e_q        # same rounding points; residual = accumulation order + rare 1-ulp flips
    assert max(e_f) < 3e-2, e_f        # stated bf16 tolerance vs the exact model


def test_no_cpu_fallback(dev):
    from pointcloudprocessing_amd._lib import PointNetHipError
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    m = PointNet(CCLS, CSEG, 0.3, 42, device="cpu")
    with pytest.raises(PointNetHipError):
        m(torch.zeros(1, 64, 3), training=False)


# ---------------------------------------------------------------------------------------------------------------------
# memory safety without a GPU sanitizer: guard bands around every workspace entry and every boundary buffer
# ---------------------------------------------------------------------------------------------------------------------
GUARD_PAT = 0xA5


def _ws_entries(m, B, N, training):
    import ctypes as C
    from pointcloudprocessing_amd._lib import lib
    name = C.create_string_buffer(128); off = C.c_int64(); nb = C.c_int64()
    out, i = [], 0
    while lib().pn_model_ws_entry(C.byref(m._desc), B, N, int(training), i, name, 128, C.byref(off), C.byref(nb)) == 0:
        out.append((name.value.decode(), off.value, nb.value)); i += 1
    return out


@pytest.mark.parametrize("vanilla", [False, True])
@pytest.mark.parametrize("B,N", [(3, 100), (16, 136), (5, 2048)])
def test_workspace_entries_are_never_overrun(dev, monkeypatch, vanilla, B, N):
    """PN_WS_GUARD plans the workspace with an untouched gap after every entry; after a training step and an inference
    call every gap byte must still hold the fill pattern (ragged shapes: partial tiles, odd batch)."""
    monkeypatch.setenv("PN_WS_GUARD", "32768")
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    m = PointNet(23, 12, 0.3, 42, vanilla=vanilla, precision="bf16", device=dev)
    g = torch.Generator().manual_seed(B * N)
    pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
    y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
    y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
    se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
    keep = (torch.ones(B, 512, dtype=torch.uint8, device=dev), torch.ones(B, 256, dtype=torch.uint8, device=dev))
    for training in (True, False):
        m._workspace(B, N, training).fill_(GUARD_PAT)
    m.fused_loss_step(pc, y_cls, y_seg, se3, (1.0, 1.0, 1.0), keep=keep)
    m(pc, training=False)
    torch.cuda.synchronize()
    for training in (True, False):
        ws = m._workspace(B, N, training)
        ent = _ws_entries(m, B, N, training)
        assert len(ent) > 50
        for i, (n, o, nb) in enumerate(ent):
            end = ent[i + 1][1] if i + 1 < len(ent) else ws.numel()
            assert end - (o + nb) >= 32768
            assert bool((ws[o + nb: end] == GUARD_PAT).all()), f"guard after workspace entry {n} was overwritten (training={training})"


@pytest.mark.parametrize("vanilla", [False, True])
def test_boundary_buffers_are_never_overrun(dev, vanilla):
    """every buffer handed over the C ABI (inputs, labels, masks, parameters, gradients, Adam state, scalars, outputs) is
    carved from one patterned arena: forward + backward + Adam may not touch a gap byte or modify a read-only input."""
    import ctypes as C
    from pointcloudprocessing_amd._lib import check, current_stream, lib
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    B, N, GAP = 5, 200, 8192
    m = PointNet(23, 12, 0.3, 42, vanilla=vanilla, precision="bf16", device=dev)
    P = m.params_flat.numel()
    arena = torch.full((16 * 2**20 + 4 * P * 4,), GUARD_PAT, dtype=torch.uint8, device=dev)
    cur, carved = [GAP], {}

    def carve(name, src=None, nbytes=None):
        nb = src.numel() * src.element_size() if src is not None else nbytes
        o = cur[0]
        t = arena[o:o + nb]
        if src is not None:
            t.copy_(src.contiguous().view(torch.uint8).reshape(-1))
        carved[name] = (o, nb)
        cur[0] = (o + nb + GAP + 255) & ~255
        return t

    g = torch.Generator().manual_seed(5)
    ins = dict(pc=torch.rand(B, N, 3, generator=g) * 10, y_cls=torch.randint(0, 23, (B,), generator=g, dtype=torch.int32),
               y_seg=torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32), se3=torch.eye(3).expand(B, 3, 3).contiguous(),
               keep1=(torch.rand(B, 512, generator=g) > 0.3).to(torch.uint8), keep2=(torch.rand(B, 256, generator=g) > 0.3).to(torch.uint8))
    tin = {k: carve(k, v.to(dev)) for k, v in ins.items()}
    t_params = carve("params", m.params_flat.data)
    t_grads = carve("grads", nbytes=P * 4)
    t_m, t_v = carve("adam_m", torch.zeros(P)), carve("adam_v", torch.zeros(P))
    t_it, t_al = carve("iterations", torch.zeros(1, dtype=torch.int32)), carve("alpha", torch.zeros(4))
    t_sc = carve("scalars", torch.zeros(16))
    t_cls, t_seg, t_R = carve("out_cls", nbytes=B * 23 * 4), carve("out_seg", nbytes=B * N * 12 * 4), carve("out_R", nbytes=B * 9 * 4)
    snap = {k: tin[k].clone() for k in tin}
    fused = dict(labels_cls=tin["y_cls"].view(torch.int32), labels_seg=tin["y_seg"].view(torch.int32), se3=tin["se3"].view(torch.float32),
                 loss_weights=(1.0, 1.0, 1.0), keep=(tin["keep1"], tin["keep2"]))
    io, _ = m._io(tin["pc"].view(torch.float32).view(B, N, 3), True, fused)
    io.params, io.grads, io.scalars = t_params.data_ptr(), t_grads.data_ptr(), t_sc.data_ptr()
    io.out_cls, io.out_seg, io.out_R = t_cls.data_ptr(), t_seg.data_ptr(), t_R.data_ptr()
    check(lib().pn_adam_prepare(t_it.data_ptr(), t_al.data_ptr(), 1e-3, 0.7, 7000.0, 0.9, 0.999, current_stream()), "pn_adam_prepare")
    for _ in range(2):
        check(lib().pn_model_forward(C.byref(m._desc), C.byref(io), current_stream()), "pn_model_forward")
        check(lib().pn_model_backward(C.byref(m._desc), C.byref(io), None, None, None, current_stream()), "pn_model_backward")
        check(lib().pn_adam_step(t_params.data_ptr(), t_grads.data_ptr(), t_m.data_ptr(), t_v.data_ptr(), P, t_it.data_ptr(),
                                 t_al.data_ptr(), 1e-3, 0.7, 7000.0, 0.9, 0.999, 1e-7, 1.0, current_stream()), "pn_adam_step")
    torch.cuda.synchronize()
    mask = torch.ones_like(arena, dtype=torch.bool)
    for k, (o, nb) in carved.items():
        mask[o:o + nb] = False
    assert int(((arena != GUARD_PAT) & mask).sum()) == 0
    for k in tin:
        assert torch.equal(tin[k], snap[k]), f"read-only input {k} was modified"
    assert bool(torch.isfinite(t_grads.view(torch.float32)).all()) and int(t_it.view(torch.int32)[0]) == 2


def test_inference_against_committed_golden_vectors(dev):
    """the HIP path on the inputs of tests/golden/oracle_b2_n64.npz against the outputs stored there (fp64 oracle, made by
    tests/golden/make_oracle_vectors.py): probabilities within 1e-5 (bf16x3 operands), arg-max class and per-point part bit-exact
    wherever the stored top-2 margin exceeds that tolerance."""
    gv = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_b2_n64.npz"))
    params = O.init_params(23, 12, seed=7, vanilla=False, dtype=torch.float64, randomize_bn=True)     # as in the generator
    m = build_model(dev, params, precision="bf16x3")
    pc = torch.from_numpy(gv["pc"]).float().to(dev)
    cls, seg, R = m(pc, training=False)
    cls, seg, R = cls.double().cpu().numpy(), seg.double().cpu().numpy(), R.double().cpu().numpy()
    assert np.abs(cls - gv["inf_cls"]).max() < 1e-5 and np.abs(seg - gv["inf_seg"]).max() < 1e-5 and np.abs(R - gv["inf_R"]).max() < 1e-4
    top2 = np.sort(gv["inf_seg"], -1)
    safe = (top2[..., -1] - top2[..., -2]) > 1e-4
    assert safe.mean() > 0.9
    assert np.array_equal(seg.argmax(-1)[safe], gv["inf_seg"].argmax(-1)[safe])
    assert np.array_equal(cls.argmax(-1), gv["inf_cls"].argmax(-1))
    # the inference entry (indices taken on the device by pn_argmax_rows): the same indices, int32
    ci, si, R2 = m.predict(pc)
    assert ci.dtype == torch.int32 and si.dtype == torch.int32 and tuple(si.shape) == seg.shape[:2]
    assert np.array_equal(ci.cpu().numpy(), cls.argmax(-1)) and np.array_equal(si.cpu().numpy(), seg.argmax(-1))
    assert np.array_equal(si.cpu().numpy()[safe], gv["inf_seg"].argmax(-1)[safe])
    assert np.array_equal(R2.double().cpu().numpy(), R)


# ---------------------------------------------------------------------------------------------------------------------
# the reference's layer classes used on their own (PointNet.py:379-679): forward through the op-level C ABI
# ---------------------------------------------------------------------------------------------------------------------
def _bn_ref(z, gamma, beta, mm, mv, training, eps=1e-3):
    z = z.double()
    if training:
        mean, var = z.mean(0), z.var(0, unbiased=False)
    else:
        mean, var = mm.double(), mv.double()
    return (z - mean) / torch.sqrt(var + eps) * gamma.double() + beta.double()


@pytest.mark.parametrize("cin,filters,training", [(3, 64, False), (64, 128, True), (128, 1024, False)])
def test_free_standing_conv_layer(dev, cin, filters, training):
    from pointcloudprocessing_amd.pointnet.PointNet import ConvLayer
    B, N = 3, 200
    g = torch.Generator().manual_seed(cin + filters)
    x = torch.randn(B, N, 1, cin, generator=g).to(dev)
    layer = ConvLayer(filters=filters, name="t", activation="relu", random_seed=3)
    y = layer(x, training=training)
    assert y.shape == (B, N, 1, filters) and layer.name == "t_convolution_layer"
    W = layer.kernel.detach().clone()
    z = x.reshape(-1, cin).double().cpu() @ W.double().cpu()
    ref = torch.relu(_bn_ref(z, torch.ones(filters), torch.zeros(filters), torch.zeros(filters), torch.ones(filters), training))
    assert torch.allclose(y.reshape(-1, filters).double().cpu(), ref, rtol=2e-4, atol=2e-4)
    if training:      # moving statistics moved towards the batch statistics with momentum 0.99
        assert torch.allclose(layer.bn.moving_mean.double().cpu(), 0.01 * z.mean(0), rtol=1e-3, atol=1e-5)
    layer.freeze()
    assert not layer.is_trainable() and not layer.bn.trainable


def test_free_standing_dense_layer_and_tnet(dev):
    from pointcloudprocessing_amd.pointnet.PointNet import DenseLayer, TNet
    g = torch.Generator().manual_seed(11)
    x = torch.randn(8, 1024, generator=g).to(dev)
    d = DenseLayer(units=512, name="d", activation="relu", apply_bn=True, random_seed=5)
    y = d(x, training=True)
    z = x.double().cpu() @ d.kernel.double().cpu()
    ref = torch.relu(_bn_ref(z, torch.ones(512), torch.zeros(512), None, None, True))
    assert y.shape == (8, 512) and torch.allclose(y.double().cpu(), ref, rtol=3e-4, atol=3e-4)
    # a fresh T-Net predicts the identity (w = 0, b = I) whatever the input: PointNet.py:412-416
    for K in (3, 64):
        t = TNet(name=f"tn{K}", random_seed=7)
        pc = torch.randn(2, 130, K, generator=g).to(dev)
        R = t(pc, training=False)
        assert R.shape == (2, K, K) and torch.allclose(R.cpu(), torch.eye(K).expand(2, K, K), atol=1e-6)
        assert t.get_last_predicted_transformation() is R
        # with a non-zero w the chain conv -> max -> dense -> @w + b must match a plain restatement
        t._own["w"].copy_(torch.randn(256, K * K, generator=g) * 0.01)
        R2 = t(pc, training=False).double().cpu()
        a = pc.double().cpu().reshape(-1, K)
        inv = 1.0 / np.sqrt(1.0 + 1e-3)
        for l in (t.conv_layer_1, t.conv_layer_2, t.conv_layer_3):
            a = torch.relu(a @ l.kernel.double().cpu() * inv)
        gmax = a.reshape(2, 130, 1024).amax(1)
        for l in (t.dense_layer_1, t.dense_layer_2):
            gmax = torch.relu(gmax @ l.kernel.double().cpu() * inv)
        ref = (gmax @ t.w.double().cpu() + t.b.double().cpu().reshape(-1)).reshape(2, K, K)
        assert torch.allclose(R2, ref, rtol=1e-3, atol=1e-4), float((R2 - ref).abs().max())


def test_forward_prologue_draws_masks_and_clears_gradients(dev):
    """pn_model_io.dropout_step / zero_grads_in_forward: the forward pass's first launch draws the masks pn_dropout_masks would
    (counter increment included) and clears the gradient buffer; a step run that way equals a step with those masks as inputs."""
    from pointcloudprocessing_amd import _lib as L
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    B, N, seed = 5, 300, 0x1234567811
    g = torch.Generator().manual_seed(77)
    pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
    y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
    y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
    se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
    # what pn_dropout_masks writes for (seed, step 5)
    step = torch.full((1,), 5, dtype=torch.int32, device=dev)
    k1 = torch.empty(B, 512, dtype=torch.uint8, device=dev); k2 = torch.empty(B, 256, dtype=torch.uint8, device=dev)
    L.check(L.lib().pn_dropout_masks(L.ptr(k1), k1.numel(), L.ptr(k2), k2.numel(), 0.3, seed, L.ptr(step), L.current_stream()), "pn_dropout_masks")
    assert int(step.item()) == 6 and 0.6 < float(k1.float().mean()) < 0.8
    params = O.init_params(23, 12, seed=3, randomize_bn=True)
    res = []
    for mode in ("inputs", "drawn"):
        m = PointNet(23, 12, 0.3, 42, precision="bf16", device=dev)
        m.set_weights(params)
        m.grads_flat.fill_(1e30)                       # stale gradients: the forward pass must clear them
        if mode == "inputs":
            m.fused_loss_step(pc, y_cls, y_seg, se3, (1.0, 1.0, 1.0), keep=(k1, k2))
        else:
            step2 = torch.full((1,), 5, dtype=torch.int32, device=dev)
            b1 = torch.zeros_like(k1); b2 = torch.zeros_like(k2)
            m.fused_loss_step(pc, y_cls, y_seg, se3, (1.0, 1.0, 1.0), keep=(b1, b2), dropout_rng=(seed, step2))
            assert torch.equal(b1, k1) and torch.equal(b2, k2) and int(step2.item()) == 6
        torch.cuda.synchronize()
        res.append((m.grads_flat.clone(), m.scalars.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert float(res[0][0].abs().max()) < 1e6        # nothing of the stale fill survived


@pytest.mark.parametrize("vanilla", [False, True])
# (32, 1024), (40, 840), (64, 576): enough tiles for the 128-row form of the kernel (pn_segout.hip: seg_head_fused) -- whole tiles, a ragged
# last tile with rows in both 64-row halves, a last tile whose second half is empty
@pytest.mark.parametrize("B,N", [(4, 300), (32, 1024), (40, 840), (64, 576)])
def test_fused_frozen_segmentation_head_equals_layer_by_layer(dev, vanilla, B, N):
    """A segmentation head that normalises with moving statistics and gets no gradient (inference; `classification_pretrain`) runs as
    ONE launch with its activations kept on chip (pn_segout.hip: seg_head_fused).  It must reproduce the layer-by-layer plan bit for
    bit: same probabilities in inference, same outputs and gradients in a training step; the loss / accuracy sums are grouped per
    64 instead of 128 rows, so they agree to rounding."""
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    params = O.init_params(CCLS, CSEG, seed=21, vanilla=vanilla, randomize_bn=True)
    pc, y_cls, y_seg, se3, keep = make_inputs(B, N, 9)
    kp = (keep["dropout_1"].to(torch.uint8).to(dev), keep["dropout_2"].to(torch.uint8).to(dev))
    res = []
    for keep_act in (False, True):
        m = PointNet(CCLS, CSEG, 0.3, 42, vanilla=vanilla, precision="bf16", device=dev)
        m.set_weights(params)
        m.keep_activations = keep_act
        inf = [t.clone() for t in m(pc.to(dev), training=False)]
        m.freeze_segmentation_head()
        tr = m.fused_loss_step(pc.to(dev), y_cls.to(torch.int32).to(dev), y_seg.to(torch.int32).to(dev), se3.to(dev), (1.0, 0.0, 0.0), keep=kp)
        torch.cuda.synchronize()
        res.append((inf, [t.clone() for t in tr], m.grads_flat.clone(), m.scalars.clone()))
    (inf_f, tr_f, g_f, sc_f), (inf_l, tr_l, g_l, sc_l) = res
    for a_, b_ in zip(inf_f + tr_f, inf_l + tr_l):
        assert torch.equal(a_, b_)
    assert torch.equal(g_f, g_l)
    assert torch.allclose(sc_f, sc_l, rtol=1e-5, atol=1e-5), (sc_f.tolist(), sc_l.tolist())
    assert float(inf_f[1].sum(-1).sub(1).abs().max()) < 1e-4          # the fused head's softmax rows sum to one
    # a gradient through a head that kept no activations is refused, not computed from stale buffers
    m = PointNet(CCLS, CSEG, 0.3, 42, vanilla=vanilla, precision="bf16", device=dev)
    m.set_weights(params)
    m.freeze_segmentation_head()
    m.fused_loss_step(pc.to(dev), y_cls.to(torch.int32).to(dev), y_seg.to(torch.int32).to(dev), se3.to(dev), (1.0, 0.0, 0.0), keep=kp)
    from pointcloudprocessing_amd._lib import PointNetHipError
    with pytest.raises(PointNetHipError):
        m._run_backward(None, torch.zeros(B, N, CSEG, device=dev), None)
