"""End-to-end trainer run on the GPU: the reference's entry point (pointnet_train.py config -> datasets -> two chained
profiles -> artefacts) on the HIP engine, and a short convergence check of the native train step."""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_cpu_train import write_config      # noqa: E402


def test_trainer_end_to_end_on_gpu(dev, tmp_path):
    from pointcloudprocessing_amd import pointnet_train as T
    cfg, d = write_config(tmp_path, vanilla=False, epochs=2)
    assert T.train_pointnet([cfg], max_steps_per_epoch=3)
    for prof in ("classification_pretrain", "final"):
        pd = d + f"models/unit/{prof}/"
        h = json.load(open(pd + f"unit_{prof}_history.json"))
        assert set(h.keys()) == set(T.HISTORY_KEYS) | {"val_" + k for k in T.HISTORY_KEYS}
        assert all(len(v) == 2 and all(np.isfinite(v)) for v in h.values())
        ck = torch.load(pd + f"unit_{prof}.pt", weights_only=True)
        assert ck["config"]["classification_output_width"] == 23 and "mlp_2_3.kernel" in ck["weights"]
    log = open(glob.glob(d + "models/unit/log_*.log")[0]).read()
    assert "Continuing training on model unit/classification_pretrain/unit_classification_pretrain.pt" in log


def test_trainer_at_baseline_config_c1_size(dev, tmp_path):
    """BASELINE config C1 (plumbing): PointNet-cls N=1024 points, batch 4, through the reference's entry point
    (pointnet_train.py config -> datasets -> profiles -> artefacts); 3 steps per epoch.  The 490- / 313-point reference clouds are
    padded to 1024 by _adjust_to_input_width (PointCloudSet.py:443-470)."""
    from pointcloudprocessing_amd import pointnet_train as T
    cfg, d = write_config(tmp_path, vanilla=False, epochs=1, input_width=1024, batch_size=4, n_frames=24)
    assert T.train_pointnet([cfg], max_steps_per_epoch=3, precision="bf16")
    h = json.load(open(d + "models/unit/classification_pretrain/unit_classification_pretrain_history.json"))
    assert all(len(v) == 1 and np.isfinite(v[0]) for v in h.values())
    assert 0.0 <= h["classification_output_sparse_categorical_accuracy"][0] <= 1.0 and h["loss"][0] > 0
    ck = torch.load(d + "models/unit/final/unit_final.pt", weights_only=True)
    assert ck["config"]["precision"] == "bf16" and tuple(ck["weights"]["mlp_2_3.kernel"].shape) == (128, 1024)


def test_gpu_trained_model_exports_to_onnx_and_matches_hip_inference(dev, tmp_path):
    """N3 on the GPU: 3 training steps per epoch through the trainer entry point (pointnet_train.py:238-248 writes <name>_<prof>.onnx
    from the restored best weights), then the written file -- parsed and evaluated by the independent NumPy interpreter of
    tests/test_cpu_onnx.py -- against HIP inference (PointNet.__call__, training=False) on the very weights the file restores:
    probabilities within 1e-4 (bf16x3 inference vs the file evaluated in fp64), class / part indices identical where the margin allows.
    The file also restores as a checkpoint with the model's own constructor arguments (metadata_props)."""
    from pointcloudprocessing_amd import onnx_export as X
    from pointcloudprocessing_amd import pointnet_train as T
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    from test_cpu_onnx import _run_onnx
    cfg, d = write_config(tmp_path, vanilla=False, epochs=1, input_width=256, batch_size=4, n_frames=24)
    assert T.train_pointnet([cfg], max_steps_per_epoch=3)
    path = d + "models/unit/final/unit_final.onnx"
    model = X.parse_model(open(path, "rb").read())
    w = X.read_onnx_weights(path)
    conf = X.read_onnx_config(path)
    assert conf["dropout_rate"] == 0.3 and conf["vanilla"] is False and conf["classification_output_width"] == 23
    ck = torch.load(d + "models/unit/final/unit_final.pt", weights_only=True)
    for k, v in ck["weights"].items():                            # the file holds the trained weights bit for bit
        assert np.array_equal(w[k], v.numpy()), k
    assert not np.array_equal(w["mlp_2_3.kernel"], PointNet(23, 12, 0.3, ck["config"]["random_seed"], device="cpu").named_weights()["mlp_2_3.kernel"].numpy())
    m = PointNet(**{**conf, "precision": "bf16x3", "device": dev})
    m.set_weights({k: torch.from_numpy(v) for k, v in w.items()})
    g = torch.Generator().manual_seed(11)
    pc = (torch.rand(3, 256, 3, generator=g) * 40 - 20).float()
    cls, seg, R = m(pc.to(dev), training=False)
    out = _run_onnx(model, {"pointnet_input": pc.numpy().astype(np.float64)})
    e = [float(np.abs(cls.cpu().double().numpy() - out[0]).max()), float(np.abs(seg.cpu().double().numpy() - out[1]).max()),
         float(np.abs(R.cpu().double().numpy() - out[2]).max())]
    assert max(e) < 1e-4, e
    assert np.array_equal(cls.cpu().numpy().argmax(-1), out[0].argmax(-1))
    top2 = np.sort(out[1], -1)
    safe = (top2[..., -1] - top2[..., -2]) > 1e-3
    assert np.array_equal(seg.cpu().numpy().argmax(-1)[safe], out[1].argmax(-1)[safe])


def test_native_train_step_learns_and_graph_matches_eager(dev):
    """a few hundred Adam steps on one fixed batch must drive the classification loss down; the hipGraph replay of the
    step must produce exactly the same weights as eager launches (same kernels, same order)."""
    from pointcloudprocessing_amd.engine import TrainStep
    from pointcloudprocessing_amd.optim import KerasAdam
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    B, N = 8, 256
    g = torch.Generator().manual_seed(0)
    pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
    y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
    y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
    se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
    finals, w0 = [], None
    # split: the data-parallel step layout (backward in two phases around the bucketed all-reduce, Adam in its own graph)
    for use_graph, aux, split in ((False, True, False), (True, True, False), (False, False, False), (True, False, False),
                                  (False, False, True), (True, False, True)):
        m = PointNet(23, 12, 0.0, 42, precision="bf16x3", device=dev)     # dropout 0: deterministic step
        if w0 is None:
            # the reference leaves the classification DenseLayers unseeded (PointNet.py:130-134) and so does the model: a fresh draw per
            # process.  The comparisons below are between precision modes, so the test fixes the draw (one run in ~20 landed 4 % apart at
            # step 0 where the usual draw is within 1 %: eight clouds under batch-statistics BatchNormalization)
            from pointcloudprocessing_amd.pointnet.PointNet import _glorot_uniform
            with torch.no_grad():
                for i, nme in enumerate(("mlp_cls_1.kernel", "mlp_cls_2.kernel", "mlp_cls_3.kernel")):
                    v = m._weights.view(nme)
                    v.copy_(_glorot_uniform(tuple(v.shape), 1000 + i).to(v.device))
            w0 = m.params_flat.data.clone()
        else:
            m.params_flat.data.copy_(w0)       # the classification head is unseeded (PointNet.py:186-206): share the start
        opt = KerasAdam(m.params_flat.data, 1e-3, 7000, 0.7)
        ts = TrainStep(m, opt, B, N, (1.0, 1.0, 1.0), use_graph=use_graph, aux=aux, split_optimizer=split)   # aux: parameter gradients on a 2nd stream
        losses = []
        for i in range(62):
            ts(pc, y_cls, y_seg, se3)
            losses.append(float(m.scalars[0]) / B)
        # The fixed 8-cloud batch is fitted to a loss of ~0.005 within 40 steps and then, at this learning rate, spikes now and then
        # (batch-statistics BatchNormalization over 8 rows): the fp32 CPU oracle's own trajectory on this problem does the same
        # (O.train_step, lr 1e-3: 3.38 -> 0.005 with excursions to 0.97 among its last 30 steps).  So "it learns" is asserted on the
        # best and the median of the tail, each far below the round-2 form (last < 0.5 x first) in what it demands of the typical step.
        tail = sorted(losses[-16:])
        assert min(losses[-8:]) < 0.1 * losses[0] and tail[len(tail) // 2] < 0.25 * losses[0], (losses[0], losses[-16:])
        finals.append(m.params_flat.data.clone())
        assert int(opt.iterations) == 62
        assert ts.mode == ("hipgraph" if use_graph else "eager")
    for f in finals[1:]:
        assert torch.equal(finals[0], f)
    # the mode bench.py and the trainer default to on the GPU box ('bf16': bf16 MFMA operands, bf16 layer-boundary tensors) against the
    # fp32-grade mode from the same start: the first steps, before the trajectories decorrelate, stay within a band, and it learns too
    traj = {}
    for prec in ("bf16x3", "bf16"):
        m = PointNet(23, 12, 0.0, 42, precision=prec, device=dev)
        m.params_flat.data.copy_(w0)
        opt = KerasAdam(m.params_flat.data, 1e-3, 7000, 0.7)
        ts = TrainStep(m, opt, B, N, (1.0, 1.0, 1.0))
        traj[prec] = []
        for i in range(62):
            ts(pc, y_cls, y_seg, se3)
            traj[prec].append(float(m.scalars[0]) / B)
    from parity_harness import report
    report("fixed-batch loss, first 12 steps  bf16x3: " + " ".join(f"{v:.3f}" for v in traj["bf16x3"][:12]))
    report("fixed-batch loss, first 12 steps  bf16:   " + " ".join(f"{v:.3f}" for v in traj["bf16"][:12]))
    # Adam's first updates are ~lr * sign(gradient): rounding-level gradient differences move whole parameters by +-lr, so the two
    # trajectories are within 1 % at step 0 and a few tenths apart by step 2 (measured: 3.52 3.63 2.82 1.61 vs 3.49 3.34 2.07 1.37) while
    # descending at the same pace: the band is 0.35 x the initial loss over the first 10 steps, and both halve the loss within 2 steps
    # of each other
    for i in range(10):
        assert abs(traj["bf16"][i] - traj["bf16x3"][i]) < 0.35 * traj["bf16x3"][0], (i, traj["bf16"][:12], traj["bf16x3"][:12])
    assert abs(traj["bf16"][0] - traj["bf16x3"][0]) < 0.03 * traj["bf16x3"][0]
    half = {k: next(i for i, v in enumerate(t) if v < 0.5 * t[0]) for k, t in traj.items()}
    assert abs(half["bf16"] - half["bf16x3"]) <= 2, half
    tail = sorted(traj["bf16"][-16:])
    assert min(traj["bf16"][-8:]) < 0.1 * traj["bf16"][0] and tail[len(tail) // 2] < 0.25 * traj["bf16"][0], traj["bf16"][-16:]


def test_interleaved_models_graph_replay_is_exact(dev):
    """regression: a graph-replayed model must stay bit-identical to its eagerly stepped twin while an unrelated model
    (different weights) trains in the same process between the replays, and the gradient slots nobody writes (moving
    statistics) must stay exactly zero.  (A captured hipMemsetAsync node used to replay with a garbage pattern here.)"""
    from pointcloudprocessing_amd.engine import TrainStep
    from pointcloudprocessing_amd.optim import KerasAdam
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    B, N = 8, 256
    g = torch.Generator().manual_seed(1)
    pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
    y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
    y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
    se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
    arms = []
    for tag, use_graph in (("other", False), ("eager", False), ("graph", True), ("other2", True)):
        m = PointNet(23, 12, 0.0, 42, precision="bf16x3", device=dev)
        if tag == "graph":
            m.params_flat.data.copy_(arms[1][0].params_flat.data)
        opt = KerasAdam(m.params_flat.data, 1e-3, 7000, 0.7)
        arms.append((m, opt, TrainStep(m, opt, B, N, (1.0, 1.0, 1.0), use_graph=use_graph)))
    moving = [s for n, s in arms[0][0]._weights.slots.items() if n.endswith("moving_mean") or n.endswith("moving_var")]
    for step in range(10):
        for m, opt, ts in arms:
            ts(pc, y_cls, y_seg, se3)
        torch.cuda.synchronize()
        for m, opt, ts in arms:
            assert bool(torch.isfinite(m.grads_flat).all()), (step, ts.mode)
            for s in moving:
                assert not bool(m.grads_flat[s["offset"]: s["offset"] + s["rows"] * s["cols"]].any()), (step, ts.mode)
        assert torch.equal(arms[1][0].grads_flat, arms[2][0].grads_flat), step
        assert torch.equal(arms[1][0].params_flat.data, arms[2][0].params_flat.data), step
    assert arms[2][2].mode == "hipgraph" and arms[3][2].mode == "hipgraph"


_DIGEST_SCRIPT = r"""
import hashlib, sys, torch
sys.path.insert(0, {root!r})
from pointcloudprocessing_amd.pointnet.PointNet import PointNet
dev = torch.device("cuda:0")
B, N = 4, 384
g = torch.Generator().manual_seed(5)
pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
m = PointNet(23, 12, 0.3, 42, precision="bf16", device=dev)
with torch.no_grad():                                 # the classification DenseLayers are unseeded, as in the reference: pin them here
    for n, v in m.named_weights().items():
        if n.startswith("mlp_cls") and n.endswith("kernel"):
            v.copy_(((torch.rand(v.shape, generator=g) * 2 - 1) * 0.05).to(dev))
keep = ((torch.rand(B, 512, generator=g) >= 0.3).to(torch.uint8).to(dev), (torch.rand(B, 256, generator=g) >= 0.3).to(torch.uint8).to(dev))
m.fused_loss_step(pc, y_cls, y_seg, se3, (1.0, 1.0, 1.0), keep=keep)      # explicit masks: the default draws them from the device RNG
torch.cuda.synchronize()
import json
gf = m.grads_flat.cpu().numpy()
print("DIGEST", json.dumps({{n: hashlib.sha256(gf[int(sl["offset"]):int(sl["offset"]) + int(sl["rows"]) * int(sl["cols"])].tobytes()).hexdigest()[:16]
                            for n, sl in m._weights.slots.items()}}), float(abs(gf).sum()))
"""


def test_batched_backward_launches_do_not_change_a_bit(dev):
    """the deferred / batched launches of the backward pass (slab reductions, weight-gradient GEMMs, G W products, tile shapes of the
    few-slab jobs) are a scheduling matter only: every gradient keeps its bits when each switch restores the one-launch-per-layer form.
    The switches are read once per process, hence the child processes (one at a time)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _DIGEST_SCRIPT.format(root=root)

    def digest(**env):
        e = dict(os.environ)
        e.update({k: str(v) for k, v in env.items()})
        out = subprocess.run([sys.executable, "-c", script], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][0]
        body, total = line[len("DIGEST "):].rsplit(" ", 1)
        assert float(total) > 0
        return json.loads(body)

    def differing(a, b):
        return [k for k in a if a[k] != b[k]]

    base = digest()
    assert differing(digest(), base) == []                       # the step itself is reproducible from process to process
    for switch in ("PN_WGRAD_BATCH", "PN_SLAB_DEFER", "PN_GW_BATCH", "PN_PM_SMALL"):
        assert differing(digest(**{switch: 0}), base) == [], switch


_RCCL_SCRIPT = r"""
import json, os, sys, torch
sys.path.insert(0, {root!r})
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
import torch.distributed as dist
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
from pointcloudprocessing_amd.engine import TrainStep
from pointcloudprocessing_amd.optim import KerasAdam
from pointcloudprocessing_amd.pointnet.PointNet import PointNet
B, N = 8, 256
g = torch.Generator().manual_seed(0)
pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
finals, modes, w0, extents = [], [], None, []
# arms 0-2: every block trains; arms 3-5: the `classification_pretrain` stage (segmentation head frozen, loss weights 1/0/0) -- both
# buckets and the optimizer are clipped to the extent of the trainable blocks (PointNet.grad_extent)
for arm, split in enumerate((False, True, True, False, True, True)):
    os.environ["PN_DDP_OVERLAP"] = "0" if arm % 3 == 2 else "1"
    frozen = arm >= 3
    m = PointNet(23, 12, 0.0, 42, precision="bf16", device=dev)
    if w0 is None:
        w0 = m.params_flat.data.clone()
    else:
        m.params_flat.data.copy_(w0)
    if frozen:
        m.freeze_segmentation_head()
    extents.append(list(m.grad_extent()) + [m.grads_flat.numel()])
    opt = KerasAdam(m.params_flat.data, 1e-3, 7000, 0.7)
    ts = TrainStep(m, opt, B, N, (1.0, 0.0, 0.0) if frozen else (1.0, 1.0, 1.0), use_graph=True, split_optimizer=split)
    assert ts.reduce == split
    for i in range(12):
        ts(pc, y_cls, y_seg, se3)
    torch.cuda.synchronize()
    finals.append(m.params_flat.data.clone()); modes.append(ts.mode)
print("RESULT", json.dumps(dict(modes=modes, same_overlapped=bool(torch.equal(finals[0], finals[1])),
                                same_single=bool(torch.equal(finals[0], finals[2])), finite=bool(torch.isfinite(finals[1]).all()),
                                moved=float((finals[0] - w0).abs().max()),
                                frozen_same_overlapped=bool(torch.equal(finals[3], finals[4])), frozen_same_single=bool(torch.equal(finals[3], finals[5])),
                                frozen_moved=float((finals[3] - w0).abs().max()), extents=extents)))
dist.barrier()
dist.destroy_process_group()
"""


def test_rccl_world_size_1_split_graph_step_equals_fused_step(dev):
    """the data-parallel layout of a step with a REAL RCCL process group (world_size 1, one fresh child process): graph 1 (forward +
    backward phase 1) -> asynchronous all-reduce of the bucket [cut, end) -> graph 1b (backward phase 2) -> all-reduce of [0, cut) ->
    wait -> graph 2 (Adam), captured with RCCL's watchdog thread alive (capture_error_mode="thread_local", engine.py), against the
    fused single-graph step: the weights after 12 steps must be bit-identical, for the overlapped and the single-collective form --
    with every block training, and in the `classification_pretrain` stage, where the buckets and Adam cover only the trainable extent."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT.format(root=root)], capture_output=True, text=True, timeout=420,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT")][0][len("RESULT "):])
    assert res["modes"] == ["hipgraph"] * 6 and res["finite"] and res["moved"] > 0 and res["frozen_moved"] > 0
    assert res["same_overlapped"] and res["same_single"], res
    assert res["frozen_same_overlapped"] and res["frozen_same_single"], res
    lo, hi, n = res["extents"][3]
    assert res["extents"][0] == [0, n, n] and lo == 0 and hi < n, res["extents"]       # the frozen head is the tail of the flat buffer


@pytest.mark.gpu
def test_optimizer_over_the_trainable_extent_equals_the_full_range(dev):
    """With the segmentation head frozen (the `classification_pretrain` stage) its slots lie outside PointNet.grad_extent(): the step
    hands Adam only the extent.  Gradients and moments of the rest are zero, so the weights after a few steps must equal, bit for
    bit, those of an optimizer stepping the whole flat buffer -- and the frozen slots must not have moved at all."""
    from pointcloudprocessing_amd.engine import TrainStep
    from pointcloudprocessing_amd.optim import KerasAdam
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    B, N = 8, 256
    g = torch.Generator().manual_seed(2)
    pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
    y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
    y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
    se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
    finals, w0 = [], None
    for restricted in (True, False):
        m = PointNet(23, 12, 0.0, 42, precision="bf16", device=dev)
        if w0 is None:
            # the reference leaves the classification DenseLayers unseeded (PointNet.py:130-134) and so does the model: a fresh draw per
            # process.  The comparisons below are between precision modes, so the test fixes the draw (one run in ~20 landed 4 % apart at
            # step 0 where the usual draw is within 1 %: eight clouds under batch-statistics BatchNormalization)
            from pointcloudprocessing_amd.pointnet.PointNet import _glorot_uniform
            with torch.no_grad():
                for i, nme in enumerate(("mlp_cls_1.kernel", "mlp_cls_2.kernel", "mlp_cls_3.kernel")):
                    v = m._weights.view(nme)
                    v.copy_(_glorot_uniform(tuple(v.shape), 1000 + i).to(v.device))
            w0 = m.params_flat.data.clone()
        else:
            m.params_flat.data.copy_(w0)
        m.freeze_segmentation_head()
        lo, hi = m.grad_extent()
        n = m.grads_flat.numel()
        assert 0 <= lo < hi < n, (lo, hi, n)                       # the frozen head is the tail of the flat buffer
        if not restricted:
            m.grad_extent = lambda n=n: (0, n)                     # this arm steps everything
        opt = KerasAdam(m.params_flat.data, 1e-3, 7000, 0.7)
        ts = TrainStep(m, opt, B, N, (1.0, 0.0, 0.0), use_graph=restricted)
        for _ in range(6):
            ts(pc, y_cls, y_seg, se3)
        torch.cuda.synchronize()
        assert torch.equal(m.params_flat.data[hi:], w0[hi:])       # nothing outside the extent moved
        finals.append(m.params_flat.data.clone())
    assert torch.equal(finals[0], finals[1])
