"""Shared harness of the GPU parity tests: one PointNet training step (forward + keras losses + backward, moving statistics)
on the HIP path against the CPU oracle on identical injected weights, inputs and dropout masks.

TEST INFRASTRUCTURE.  PARITY UNPINNED against the reference itself (TensorFlow is absent, the reference ships no numeric
vectors): the checker is oracle/pointnet_oracle.py, a restatement of pointnet/PointNet.py:197-292 and pointnet_train.py:310-351.

How the comparison is made: PointNet is piecewise smooth -- ReLU signs and reduce_max rows are discrete
decisions, and an fp64 oracle and an fp32 / MFMA implementation legitimately disagree on them where a pre-activation is within
rounding of 0 or two points tie within rounding.  (1) The GPU's decisions must equal the free oracle's except where the
pre-activation is below `near_zero`; (2) the GPU's decisions are imposed on the oracle and every loss, output, moving statistic,
stored activation gradient and parameter gradient is compared.

Arithmetic modes and tolerances.  The GPU is compared with oracle B = the fp64 oracle whose per-point matmul operands (K >= 64) are
rounded the way the GPU's mode rounds its MFMA operands (`quant=O.bf16_round` for 'bf16' and 'bf16_f32act', BASELINE config C2;
`O.bf16x3_round` for 'bf16x3'); for 'bf16', which also keeps the layer-boundary tensors in bf16, B rounds the stored pre-BN outputs,
X_64 and the gradients of the BN outputs as well (`store_quant`).

B cannot be met exactly, and by how much not is MEASURED in the same test: oracle B32 = the SAME oracle with the SAME rounding points
run in fp32 instead of fp64.  B32 and B differ only by arithmetic below fp32 precision, yet a value that differs by 1e-7 can land on
the other side of a bf16 rounding boundary, where it differs by 4e-3 -- a rounding stage turns a relative difference e into a sparse
one of RMS sqrt(e * ulp), and twenty stacked layers drive any two implementations of one 16-bit mode apart until the difference is of
the order of the ulp itself (measured at B=32, N=1024, bf16: B32 vs B 8e-3 max / 1e-3 RMS in class probabilities, 4e-2 max / 3e-3 RMS in
part probabilities).  `floor` = |B32 - B| per quantity is therefore what ANY correct implementation of the mode
shows against B, and a quantity passes when  |gpu - B| <= max(absolute tolerance, floor_factor * floor), floor_factor = 4 (the maximum
over 1e5..1e7 elements of a heavy-tailed difference, taken twice, scatters by about 2x).  Until round 3 the widening term was the
SENSITIVITY of the quantity to the rounding (|oracle without rounding - B|), 10-30 times larger: a formula error several times the
rounding noise passed.  Outputs are also bounded in RMS (floor of the RMS, same factor), where the limit stays below 1e-2 in
probability for every BASELINE configuration.

The discriminating part is tests/teacher_forced.py (called from here): every layer of the same GPU step recomputed in fp64 from the
GPU's OWN stored input of that layer, forward and backward, with fixed tolerances of a few units in the last place of the storage
type -- nothing cascades there, so a wrong index, coefficient, mask or summation range in one kernel fails its own line whatever the
conditioning of the network.  The ratio err / limit of every line is written to gpurun_out/model_report.txt.
"""
import os

import torch

from oracle import pointnet_oracle as O   # checker only

CCLS, CSEG = 23, 12
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "model_report.txt")


def report(line):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def make_inputs(B, N, seed, kind="survey"):
    """kind 'survey': SURVEY.md 8d clouds (what bench.py times): per cloud scale ~U(1,50) m, offset ~U(-100,100)^3, points = offset +
    scale*U(-1,1)^3 -- every cloud is the same uniform cube up to noise, so the batch statistics of the per-cloud dense layers are those
    of near-constant features.  kind 'shapes': every cloud a different shape (a mixture of 1-4 anisotropic Gaussian blobs), same scales
    and offsets: the clouds of a batch differ as the reference's aircraft at different poses do."""
    g = torch.Generator().manual_seed(seed)
    s = torch.rand(B, 1, 1, generator=g) * 49 + 1
    o = (torch.rand(B, 1, 3, generator=g) * 2 - 1) * 100
    pc = (o + s * (torch.rand(B, N, 3, generator=g) * 2 - 1)).float()
    y_cls = torch.randint(0, CCLS, (B,), generator=g)
    y_seg = torch.randint(0, CSEG, (B, N), generator=g)
    q, r = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
    se3 = q.float().contiguous()
    keep = {"dropout_1": (torch.rand(B, 512, generator=g) >= 0.3), "dropout_2": (torch.rand(B, 256, generator=g) >= 0.3)}
    if kind == "shapes":
        clouds = []
        for b in range(B):
            k = int(torch.randint(1, 5, (1,), generator=g))
            parts = []
            for j in range(k):
                n = N // k + (N % k if j == 0 else 0)
                A_ = torch.randn(3, 3, generator=g) * torch.rand(3, generator=g).pow(2)       # anisotropic: lines, sheets, blobs
                parts.append(torch.randn(n, 3, generator=g) @ A_ + torch.randn(3, generator=g) * 2)
            clouds.append(torch.cat(parts)[torch.randperm(N, generator=g)])
        pc = (o + s * torch.stack(clouds)).float()
    return pc, y_cls, y_seg, se3, keep


def build_model(dev, params, vanilla=False, precision="bf16x3", reg=False, debugging=False):
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    m = PointNet(CCLS, CSEG, 0.3, 42, vanilla=vanilla, regularize_input_transform=reg, regularize_feature_transform=reg,
                 precision=precision, debugging=debugging, device=dev)
    m.set_weights(params)
    return m


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


PROFILES = {
    "all": (dict(), (1.0, 1.0, 1.0)),
    # the two live training profiles (f15_lidar_config.json:43-95)
    "classification_pretrain": (dict(seg=False), (1.0, 0.0, 0.0)),
    "final": (dict(cls=False), (0.0, 1.0, 0.0)),
    "heads_only": (dict(shared=False, it=False), (1.0, 1.0, 0.0)),
    # nothing but the add_loss terms of the two T-Nets drives the gradients (with reg=True)
    "reg_only": (dict(cls=False, seg=False), (0.0, 0.0, 0.0)),
}


def oracle_trainable(spec):
    t = {}
    if not spec.get("shared", True):
        for b in O.GROUPS["shared_network"]:
            t[b] = False
    if "it" in spec:
        t["input_transform"] = spec["it"]
    if not spec.get("cls", True):
        for b in O.GROUPS["classification_head"]:
            t[b] = False
    if not spec.get("seg", True):
        for b in O.GROUPS["segmentation_head"]:
            t[b] = False
    return t


def apply_profile(m, spec):
    # same call order as pointnet_train.py:322-332
    (m.thaw_shared_network if spec.get("shared", True) else m.freeze_shared_network)()
    (m.thaw_input_transform if spec.get("it", spec.get("shared", True)) else m.freeze_input_transform)()
    (m.thaw_classification_head if spec.get("cls", True) else m.freeze_classification_head)()
    (m.thaw_segmentation_head if spec.get("seg", True) else m.freeze_segmentation_head)()


def check_training_step(dev, B, N, profile, vanilla=False, precision="bf16x3", reg=False, seed_params=12, seed_inputs=6,
                        tol_grad=5e-3, tol_fwd=3e-4, tol_loss=2e-3, tol_stats=2e-3, near_zero=1e-2, floor_factor=4.0, tag=None,
                        inputs="survey", forced=True, end_to_end=True, damp_tnet=1.0):
    """runs one fused_loss_step on the GPU; `forced`: every layer against its teacher-forced fp64 recomputation (tests/teacher_forced.py);
    `end_to_end`: the whole step against the oracle.  Returns (worst relative gradient error, model); raises AssertionError with the
    list of failed quantities."""
    tag = tag or f"train[{profile},vanilla={vanilla},{precision},B={B},N={N},reg={reg}]"
    spec, lw = PROFILES[profile]
    if vanilla and ("it" in spec):
        spec = {k: v for k, v in spec.items() if k != "it"}
    params = O.init_params(CCLS, CSEG, seed=seed_params, vanilla=vanilla, randomize_bn=True)
    if damp_tnet != 1.0 and not vanilla:      # T-Net tails closer to their identity bias, as a regularised trained model has them
        for k in ("input_transform.w", "feature_transform.w"):
            params[k] = params[k] * damp_tnet
    pc, y_cls, y_seg, se3, keep = make_inputs(B, N, seed_inputs, inputs)
    tr = oracle_trainable(spec)
    if "it" not in spec and not spec.get("shared", True):
        tr["input_transform"] = False
    quant_b = O.bf16x3_round if precision == "bf16x3" else O.bf16_round
    store_b = O.bf16_round if precision == "bf16" else None          # 'bf16' also stores the layer-boundary tensors as bf16
    okw = dict(training=True, trainable=tr, vanilla=vanilla, dropout_masks=keep, regularize_input_transform=reg and not vanilla,
               regularize_feature_transform=reg and not vanilla)
    m = build_model(dev, params, vanilla, precision=precision, reg=reg)
    apply_profile(m, spec)
    kp = (keep["dropout_1"].to(torch.uint8).to(dev), keep["dropout_2"].to(torch.uint8).to(dev))
    max_ws = {"mlp_2_3": ("mm23", "m22")} if vanilla else {"mlp_2_3": ("mm23", "m22"), "input_transform": ("iT.m3", "iT.c2"),
                                                           "feature_transform": ("fT.m3", "fT.c2")}
    # every workspace byte is poisoned first (0xFF: NaN as fp32 and as bf16, -1 as an index): an entry this step does not write cannot be
    # read back as a plausible value left by another test -- a check that touches one fails on the NaN
    m._workspace(B, N, True).fill_(255)
    for wn, _ in max_ws.values():      # the rows of the maxima are resolved by the backward pass: mark them unwritten first
        m.workspace_tensor(wn + ".arg", B, N, True, torch.int32).fill_(-1)
    outs_g = m.fused_loss_step(pc.to(dev), y_cls.to(torch.int32).to(dev), y_seg.to(torch.int32).to(dev), se3.to(dev), lw, keep=kp)
    torch.cuda.synchronize()
    # ... and independently through the op-level entry (pn_max_resolve) from what the forward pass left in the workspace: where the
    # backward pass of a layer ran, both must agree bit for bit; where it did not (nothing upstream of it trains), only this one exists
    from pointcloudprocessing_amd import _lib, ops
    prec_id = _lib.PREC[precision]
    for on, (wn, src) in max_ws.items():
        op = _lib.operand(m.workspace_tensor(src + ".Z", B, N, True, m.activation_dtype).view(B * N, 128), ca=m.workspace_tensor(src + ".scale", B, N, True),
                          cc=m.workspace_tensor(src + ".shift", B, N, True), relu=True)
        wf = (m.workspace_tensor(wn + ".wb_hi", B, N, True, torch.bfloat16), m.workspace_tensor(wn + ".wb_lo", B, N, True, torch.bfloat16))
        argq = m.workspace_tensor(wn + ".argq", B, N, True, torch.int32).view(B, 1024)
        a_op = ops.max_resolve(op, wf, argq, B, N, 128, 1024, prec_id)
        a_bw = m.workspace_tensor(wn + ".arg", B, N, True, torch.int32).view(B, 1024)
        if bool((a_bw == -1).all()):
            a_bw.copy_(a_op)
            report(f"{tag} arg-max rows of {on}: backward pass did not run for this layer; rows from pn_max_resolve")
        else:
            assert torch.equal(a_bw, a_op), f"{on}: rows resolved inside the backward scatter differ from pn_max_resolve"

    act = m.activation_dtype

    def ws(name, dtype=torch.float32):
        return m.workspace_tensor(name, B, N, True, dtype).cpu()
    conv_map = {"mlp_1_1": "m11", "mlp_1_2": "m12", "mlp_2_1": "m21", "mlp_2_2": "m22", "mlp_seg_1": "s1", "mlp_seg_2": "s2",
                "mlp_seg_3": "s3", "mlp_seg_4": "s4"}
    dense_map = {"mlp_cls_1": "c1", "mlp_cls_2": "c2"}
    max_map = {"mlp_2_3": "mm23"}
    if not vanilla:
        conv_map.update({"input_transform.conv1": "iT.c1", "input_transform.conv2": "iT.c2", "feature_transform.conv1": "fT.c1",
                         "feature_transform.conv2": "fT.c2"})
        dense_map.update({"input_transform.dense1": "iT.d1", "input_transform.dense2": "iT.d2", "feature_transform.dense1": "fT.d1",
                          "feature_transform.dense2": "fT.d2"})
        max_map.update({"input_transform": "iT.m3", "feature_transform": "fT.m3"})
    # A frozen segmentation head that gets no gradient runs as ONE launch in the bf16 mode and keeps its layers' outputs on chip
    # (pn_segout.hip: seg_head_fused) -- the workspace entries s1..s4 of `m` are then never written.  Their decisions come from a
    # second model that keeps its activations (the layer-by-layer plan): tests/test_gpu_model.py shows the two plans agree bit for
    # bit, and the outputs compared below are still those of `m`, the plan the product runs.
    seg_src = m
    if precision == "bf16" and not spec.get("seg", True) and lw[1] == 0:
        seg_src = build_model(dev, params, vanilla, precision=precision, reg=reg)
        apply_profile(seg_src, spec)
        seg_src.keep_activations = True
        seg_src._workspace(B, N, True).fill_(255)
        outs_k = seg_src.fused_loss_step(pc.to(dev), y_cls.to(torch.int32).to(dev), y_seg.to(torch.int32).to(dev), se3.to(dev), lw, keep=kp)
        torch.cuda.synchronize()
        assert all(torch.equal(a_, b_) for a_, b_ in zip(outs_g, outs_k)), "fused frozen head differs from the layer-by-layer plan"

    fails = []
    if forced:
        import teacher_forced
        fails += teacher_forced.check_layers(m, outs_g, params, pc, y_cls, y_seg, se3, keep, tr, lw, precision, vanilla, tag, report,
                                             seg_fused=seg_src is not m, seg_src=seg_src if seg_src is not m else None, reg=reg)
    if not end_to_end:
        assert not fails, fails
        return 0.0, m

    def ws_of(on):
        mm_ = seg_src if on.startswith("mlp_seg_") else m
        return lambda name, dtype=torch.float32: mm_.workspace_tensor(name, B, N, True, dtype).cpu()
    decisions = {}
    for on, wn in conv_map.items():
        wsl = ws_of(on)
        z = wsl(wn + ".Z", act).float()
        C_ = wsl(wn + ".scale").numel()
        decisions[on + ".relu"] = (torch.addcmul(wsl(wn + ".shift"), wsl(wn + ".scale"), z.view(-1, C_)) > 0).view(B, N, C_)
    for on, wn in dense_map.items():
        a = ws(wn + ".a")
        decisions[on + ".relu"] = (a > 0).view(B, -1)
    for on, wn in max_map.items():
        arg = ws(wn + ".arg", torch.int32).view(B, 1024)
        assert int(arg.min()) >= 0 and int(arg.max()) < N, (on, int(arg.min()), int(arg.max()))
        decisions[on + ".argmax"] = arg
        # the ReLU behind the max: relu(max) = max(relu), so only the arg-max row of each (cloud, channel) matters; impose the GPU's
        # sign there (a BN output within rounding of 0 at that row may legitimately flip)
        mask = torch.ones(B, N, 1024, dtype=torch.bool)
        mask.scatter_(1, arg.long().unsqueeze(1), (ws(wn + ".g").view(B, 1024) > 0).unsqueeze(1))
        decisions[(on if on.startswith("mlp") else on + ".conv3") + ".relu"] = mask

    targets = {"classification_output": y_cls, "segmentation_output": y_seg, "se3": se3.double()}

    def oracle(dtype):
        """the oracle with the GPU mode's roundings and the GPU's decisions imposed: outputs, losses, moving statistics, every gradient"""
        pp = {k: v.to(dtype).requires_grad_(O.is_trainable_name(k) and tr.get(O.block_of(k), True)) for k, v in params.items()}
        outs, ctx = O.forward(pp, pc.to(dtype), return_ctx=True, decisions=decisions, quant=quant_b, store_quant=store_b, **okw)
        tg = {"classification_output": y_cls, "segmentation_output": y_seg, "se3": se3.to(dtype)}
        loss, parts = O.total_loss(outs, tg, dict(classification=lw[0], segmentation=lw[1], rotation=lw[2]), ctx.reg_losses)
        names = [k for k, t in pp.items() if t.requires_grad]
        tapn = [k for k, t in ctx.taps.items() if (k.endswith(".y") or k.endswith(".z")) and t.requires_grad]
        allg = torch.autograd.grad(loss, [pp[k] for k in names] + [ctx.taps[k] for k in tapn], allow_unused=True)
        dd = lambda t: None if t is None else t.detach().double()      # noqa: E731
        return dict(outs=[dd(o) for o in outs], ctx=ctx, parts={k: float(v.detach()) for k, v in parts.items()},
                    grads={k: dd(g_) for k, g_ in zip(names, allg[:len(names)])},
                    tapg={k: dd(g_) for k, g_ in zip(tapn, allg[len(names):])}, reg=[float(r.detach()) for r in ctx.reg_losses],
                    stats={k: dd(v) for k, v in ctx.new_stats.items()})

    Bq = oracle(torch.float64)       # oracle B: what the GPU is compared with
    ctx = Bq["ctx"]
    F = oracle(torch.float32)        # oracle B32: same roundings, fp32 arithmetic -> |B32 - B| = the floor of the mode on these inputs
    tap_names = ["pcn", "global"] + ([] if vanilla else ["input_transform.global", "feature_transform.global", "x64", "R64"])
    y_floor, tap_floor = {}, {}
    for on in list(conv_map) + [(o if o.startswith("mlp") else o + ".conv3") for o in max_map]:
        y_floor[on] = float((F["ctx"].taps[on + ".y"].detach().double() - ctx.taps[on + ".y"].detach()).abs().max())
    for tn in tap_names:
        tap_floor[tn] = float((F["ctx"].taps[tn].detach().double() - ctx.taps[tn].detach()).abs().max())
    F.pop("ctx")

    def judge(name, err, floor, tol_abs, scale=1.0):
        """err, floor absolute; tol_abs relative to `scale`"""
        lim = max(tol_abs * scale, floor_factor * floor)
        ok = err <= lim
        report(f"{tag} {name:44s} err {err:.3e}  floor(B32,B) {floor:.3e}  limit {lim:.3e}  err/limit {err / (lim + 1e-300):.2f}{'' if ok else '   <-- FAIL'}")
        if not ok:
            fails.append((name, err, lim))

    # (1) the GPU's decisions are valid ones: against the pre-activations of oracle B (earlier decisions imposed, so nothing cascades)
    #     a ReLU sign may differ only where |y| is within `near_zero`, and the arg-max row's value must reach the true maximum within it
    for on in conv_map:
        y = ctx.taps[on + ".y"].detach()
        diff = (y > 0) != decisions[on + ".relu"]
        worst_flip = float(y[diff].abs().max()) if diff.any() else 0.0
        lim = max(near_zero, floor_factor * y_floor[on])
        report(f"{tag} relu decisions of {on}: {int(diff.sum())} of {diff.numel()} differ from oracle B, largest |y| among them {worst_flip:.3e} (limit {lim:.3e})")
        if not worst_flip < lim:
            fails.append((on + ".relu decision", worst_flip, lim))
    for on, wn in max_map.items():
        pref = on if on.startswith("mlp") else on + ".conv3"
        yb = torch.relu(ctx.taps[pref + ".y"].detach())              # (B, N, 1024)
        arg = decisions[on + ".argmax"].long()
        gap = float((yb.amax(1) - yb.gather(1, arg.unsqueeze(1)).squeeze(1)).abs().max())
        lim = max(near_zero, floor_factor * y_floor[pref])
        report(f"{tag} arg-max rows of {on}: {int((yb.argmax(1) != arg).sum())} of {arg.numel()} differ from oracle B, worst value gap {gap:.3e} (limit {lim:.3e})")
        if not gap < lim:
            fails.append((on + ".argmax decision", gap, lim))
        del yb
    # (2) continuous quantities against oracle B, tolerance widened by the measured floor |B32 - B|
    for i, nm in enumerate(["cls", "seg", "R"]):
        d = outs_g[i].cpu().double() - Bq["outs"][i]
        fl = F["outs"][i] - Bq["outs"][i]
        judge(f"forward {nm} (max)", float(d.abs().max()), float(fl.abs().max()), tol_fwd)
        judge(f"forward {nm} (RMS)", float(d.pow(2).mean().sqrt()), float(fl.pow(2).mean().sqrt()), tol_fwd / 4)
    taps = {"pcn": "pcn", "global": "mm23.g"}
    if not vanilla:
        taps.update({"input_transform.global": "iT.m3.g", "feature_transform.global": "fT.m3.g", "x64": "X64", "R64": "fT.R"})
    for tname, wname in taps.items():
        gt = m.workspace_tensor(wname, B, N, True, act if wname == "X64" else torch.float32).cpu().double()
        rt = ctx.taps[tname].detach().reshape(-1)
        judge(f"tap {tname}", float((gt[: rt.numel()] - rt).abs().max()), tap_floor[tname], tol_fwd, scale=float(rt.abs().max()))
    # activation gradients layer by layer: dL/dy_hat (stored) and dL/dz (rebuilt from the lazy coefficients)
    lay = {"mlp_seg_4": "s4", "mlp_seg_3": "s3", "mlp_seg_2": "s2", "mlp_seg_1": "s1", "mlp_2_2": "m22", "mlp_2_1": "m21",
           "mlp_1_2": "m12", "mlp_1_1": "m11"}
    if not vanilla:
        lay.update({"feature_transform.conv2": "fT.c2", "feature_transform.conv1": "fT.c1", "input_transform.conv2": "iT.c2",
                    "input_transform.conv1": "iT.c1"})
    tapg, tapf = Bq["tapg"], F["tapg"]
    for oname, wname in lay.items():
        gy, gz = tapg.get(oname + ".y"), tapg.get(oname + ".z")
        if gy is None or gz is None or float(gy.abs().max()) == 0.0:
            continue                                    # no loss reaches this layer in this profile: its backward did not run
        C_ = gy.shape[-1]
        dy = m.workspace_tensor(wname + ".dy", B, N, True, act).cpu().double().view(-1, C_)
        zz = m.workspace_tensor(wname + ".Z", B, N, True, act).cpu().double().view(-1, C_)
        ca, cb, cc = (m.workspace_tensor(f"{wname}.{t}", B, N, True).cpu().double() for t in ("ca", "cb", "cc"))
        dz = ca * dy + cb * zz + cc
        judge(f"act-grad {oname} d(BN output)", float((dy - gy.reshape(-1, C_)).abs().max()),
              float((tapf[oname + ".y"] - gy).abs().max()), tol_grad, scale=float(gy.abs().max()))
        judge(f"act-grad {oname} d(pre-BN output)", float((dz - gz.reshape(-1, C_)).abs().max()),
              float((tapf[oname + ".z"] - gz).abs().max()), tol_grad, scale=float(gz.abs().max()))
    sc = m.scalars.cpu().double()
    pf, pb = F["parts"], Bq["parts"]
    # a loss is a function of its head's output: its floor is at least that output's RMS floor (|B32 - B| of two scalars alone can be
    # small by coincidence while the probabilities behind them moved)
    out_floor = [float((F["outs"][i] - Bq["outs"][i]).pow(2).mean().sqrt()) for i in range(3)]
    judge("cls loss", abs(float(sc[0] / B) - pb["classification_output_loss"]),
          max(abs(pf["classification_output_loss"] - pb["classification_output_loss"]), out_floor[0]), tol_loss)
    judge("seg loss", abs(float(sc[2] / (B * N)) - pb["segmentation_output_loss"]),
          max(abs(pf["segmentation_output_loss"] - pb["segmentation_output_loss"]), out_floor[1]), tol_loss)
    judge("se3 loss", abs(float(sc[4] / (B * 9)) - pb["se3_loss"]), max(abs(pf["se3_loss"] - pb["se3_loss"]), out_floor[2]), tol_loss)
    if reg and not vanilla:          # the two orthogonality regularisers, PointNet.py:447-451 (add_loss terms)
        for i, nm in ((0, "input_transform reg"), (1, "feature_transform reg")):
            judge(nm, abs(float(sc[5 + i]) - Bq["reg"][i]), abs(F["reg"][i] - Bq["reg"][i]), 2e-3, scale=max(abs(Bq["reg"][i]), 1e-6))
    # moving statistics
    nw = m.named_weights()
    for k, v in Bq["stats"].items():
        sc_ = float(v.abs().max()) + 1e-12
        judge(k, float((nw[k].double().cpu() - v).abs().max()), float((F["stats"][k] - v).abs().max()), tol_stats, scale=sc_)
    for k in params:
        if ("moving" in k) and k not in Bq["stats"]:
            assert torch.equal(nw[k].cpu(), params[k]), f"frozen statistic {k} changed"
    # gradients
    ng = m.named_grads()
    worst = 0.0
    layer_scale = {}                 # largest reference gradient among the tensors of one layer (kernel, gamma, beta / bias)
    for k, r in Bq["grads"].items():
        if r is not None:
            pre = k.rsplit(".", 2)[0] if ".bn." in k else k.rsplit(".", 1)[0]
            layer_scale[pre] = max(layer_scale.get(pre, 0.0), float(r.abs().max()))
    for k in params:
        if not O.is_trainable_name(k):
            continue
        g = ng[k].cpu().double()
        if k in Bq["grads"] and Bq["grads"][k] is not None:
            gf = F["grads"][k]
            pieces = [(k, g, Bq["grads"][k], gf)]
            if k == "mlp_seg_1.kernel":
                pieces = [(k + "[:64]", g[:64], Bq["grads"][k][:64], gf[:64]), (k + "[64:]", g[64:], Bq["grads"][k][64:], gf[64:])]
            for nm, gg, r, rf in pieces:
                scale = float(r.abs().max())
                e = float((gg - r).abs().max())
                if scale > 1e-6:
                    worst = max(worst, e / scale)
                    judge("grad " + nm, e, float((rf - r).abs().max()), tol_grad, scale=scale)
                else:
                    # a gradient that vanishes identically (e.g. d/dbeta of a layer whose only consumer is a batch-statistics
                    # BatchNormalization: a per-channel constant is annihilated) is a cancellation on the GPU: judge its residue
                    # against the layer's other gradients
                    pre = k.rsplit(".", 2)[0] if ".bn." in k else k.rsplit(".", 1)[0]
                    lim0 = max(1e-5, 1e-4 * layer_scale.get(pre, 0.0))
                    report(f"{tag} grad {nm:40s} reference ~0 ({scale:.1e}): gpu residue {e:.3e} limit {lim0:.3e}")
                    if not e < lim0:
                        fails.append((nm, e, lim0))
        else:
            assert float(g.abs().max()) == 0.0, f"frozen / unused parameter {k} received a gradient"
    report(f"{tag} worst relative gradient error {worst:.3e}")
    assert not fails, fails
    return worst, m
