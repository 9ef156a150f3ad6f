#!/usr/bin/env python3
"""PointNet training-step benchmark on MI355X:  python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): points/sec of the PointNet-cls training step (forward + keras losses + backward +
gradient all-reduce + Adam) at N=1024 points, batch 32 per GPU, synthetic random clouds -- the reference's
`classification_pretrain` profile (f15_lidar_config.json:43-69: segmentation head frozen, loss weights 1/0/0;
both heads are still evaluated in the forward pass, as PointNet.call always does, PointNet.py:250-292).
For N > 1 GPUs the driver launches this file under torch.distributed.run, one rank per GPU (weak scaling:
32 clouds per rank), gradients summed with one RCCL all-reduce of the flat buffer.

One JSON line is printed by rank 0.  Extra objects:
  roofline     the dominant kernel = fused ConvLayer(128->1024)+BN-stats+reduce_max (3 launches per step), timed live with
               HIP events on the launch stream (60 back-to-back launches, the three layers in rotation over copies of the step's
               own operands: a working set beyond the L2s and inside the Infinity Cache, which is where a launch finds its rows in
               the step);
               achieved = 2*128*1024 FLOP/point * points per launch / mean launch time, against the dense bf16 MFMA peak
               (2.5 PFLOP/s).  bound = "mfma": with 256 B/point of compulsory input (128 channels stored as bf16; 512 B with
               fp32 storage) this kernel is compute bound (1024 FLOP/B against a ridge of 312).
               traffic = HBM bytes per launch from the rocprofv3 PMC passes under profiles/round3 (FETCH_SIZE x 2 + WRITE_SIZE,
               MI355X_MICROARCH.md), reported only while pn_panel.hip is the source they were collected on, else null.
  cpu_baseline the CPU oracle (torch-CPU restatement of the reference model; the TF reference itself cannot run
               here) timed on this host for a bounded number of steps of the same workload.
  fp32_grade   the same step in the 'bf16x3' mode (split MFMA operands: fp32-grade products, fp32 layer-boundary tensors) -- the
               like-for-like figure against the reference's fp32 arithmetic -- timed after the main loop (N = 1 only; not `value`).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CCLS, CSEG = 23, 12     # f15_lidar_config.json:4-42
MFMA_BF16_PEAK = 2.5e15  # dense bf16 MFMA peak, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


def synth_batch(B, N, seed, device):
    """per cloud scale ~U(1,50) m, offset ~U(-100,100)^3, points = offset + scale*U(-1,1)^3."""
    g = torch.Generator().manual_seed(seed)
    s = torch.rand(B, 1, 1, generator=g) * 49 + 1
    o = (torch.rand(B, 1, 3, generator=g) * 2 - 1) * 100
    pc = (o + s * (torch.rand(B, N, 3, generator=g) * 2 - 1)).float().contiguous()
    y_cls = torch.randint(0, CCLS, (B,), generator=g, dtype=torch.int32)
    y_seg = torch.randint(0, CSEG, (B, N), generator=g, dtype=torch.int32)
    q, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
    se3 = q.float().contiguous()
    return [t.to(device) for t in (pc, y_cls, y_seg, se3)]


def cpu_baseline(B, N, steps=6, warmup=1):
    from oracle import pointnet_oracle as O          # reported baseline only
    # the GPU box gives one GPU a 16-CPU share even though os.cpu_count() reports the whole host
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    p = O.init_params(CCLS, CSEG, seed=1)
    pc, y_cls, y_seg, se3 = synth_batch(B, N, 20260001, "cpu")
    tg = {"classification_output": y_cls.long(), "segmentation_output": y_seg.long(), "se3": se3}
    tr = {b: False for b in O.GROUPS["segmentation_head"]}
    st = {}
    lr = {"rate": 1e-4, "decay_steps": 7000, "decay_rate": 0.7}
    g = torch.Generator().manual_seed(3)
    keep = {"dropout_1": torch.rand(B, 512, generator=g) >= 0.3, "dropout_2": torch.rand(B, 256, generator=g) >= 0.3}
    lw = dict(classification=1.0, segmentation=0.0, rotation=0.0)
    for i in range(warmup):
        O.train_step(p, pc, tg, lw, tr, st, lr, i, dropout_masks=keep)
    t0 = time.perf_counter()
    for i in range(steps):
        O.train_step(p, pc, tg, lw, tr, st, lr, warmup + i, dropout_masks=keep)
    dt = time.perf_counter() - t0
    return {"value": B * N * steps / dt, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} training steps of the same workload (B={B}, N={N}, fp32) on the torch-CPU oracle, {warmup} warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16_f32act", "bf16x3"])
    ap.add_argument("--profile", default="classification_pretrain", choices=["classification_pretrain", "final", "all"])
    ap.add_argument("--no-graph", action="store_true", help="do not replay the step from a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--aux", action="store_true", help="A/B: parameter-gradient kernels on a second stream")
    ap.add_argument("--rehearse-ddp", action="store_true",
                    help="one process: initialise RCCL with world_size 1 and run the data-parallel step layout (graph, all-reduce, graph)")
    ap.add_argument("--caller-stream", action="store_true", help="A/B: run the step on torch's current (null) stream instead of its own")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the PointNet hot path has no CPU compute path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)
    elif args.rehearse_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
    if args.gpus != world and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from pointcloudprocessing_amd.optim import KerasAdam
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    B, N = args.batch, args.points
    model = PointNet(CCLS, CSEG, 0.3, 42, precision=args.precision, device=dev)
    if world > 1:
        dist.broadcast(model.params_flat.data, src=0)
    lw = {"classification_pretrain": (1.0, 0.0, 0.0), "final": (0.0, 1.0, 0.0), "all": (1.0, 1.0, 1.0)}[args.profile]
    model.thaw_shared_network(); model.thaw_input_transform()
    (model.freeze_classification_head if args.profile == "final" else model.thaw_classification_head)()
    (model.freeze_segmentation_head if args.profile == "classification_pretrain" else model.thaw_segmentation_head)()
    opt = KerasAdam(model.params_flat.data, 1e-4, 7000, 0.7)
    pc, y_cls, y_seg, se3 = synth_batch(B, N, 20260001 + rank, dev)
    from pointcloudprocessing_amd.engine import TrainStep
    ts = TrainStep(model, opt, B, N, lw, use_graph=not args.no_graph,
                   stream=torch.cuda.current_stream() if args.caller_stream else None, aux=args.aux,
                   split_optimizer=True if args.rehearse_ddp else None)      # hipGraph replay of the whole step
    ts.load(pc, y_cls, y_seg, se3)
    step, step_eager = ts.run, ts.run_eager
    torch.cuda.set_stream(ts.stream)     # the loop lives on the step's stream: no cross-stream fences (engine.TrainStep)
    for _ in range(3):
        step()            # two eager steps, then the step is captured into a hipGraph
    graph_mode = ts.mode

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- dominant kernel: the fused ConvLayer(128->1024) + BN statistics + reduce_max panel kernel (3 launches per step) ----
    # Timed live with HIP events on the launch stream: after one eager step has left the three layers' real inputs in the workspace,
    # the three layers' launches -- same entry point, same operands, same outputs as inside the step -- are repeated in rotation, 20
    # rounds back to back between ONE event pair (an event bracket around a single ~15 us launch reads several us high).  The mean
    # includes the boundaries between the launches; rocprofv3's per-kernel average of the same command (profiles/) is the cross-check.
    import ctypes as C
    from pointcloudprocessing_amd import _lib
    step_eager()
    torch.cuda.synchronize()
    K_, C_ = 128, 1024
    prec_id = _lib.PREC[args.precision]
    layers = [("iT.c2", "iT.m3"), ("fT.c2", "fT.m3"), ("m22", "mm23")]

    def wsf(name, dtype=torch.float32):
        return model.workspace_tensor(name, B, N, True, dtype)
    REPS = 20
    calls = []
    # the launches' column-sum accumulators (values unused here): NT * K words per cloud, NT = 2 in the split-operand mode (pointnet_hip.h)
    colacc = torch.zeros(B, (2 if args.precision == "bf16x3" else 1) * K_, device=model.params_flat.device, dtype=torch.int64)
    # In the step a launch reads rows another kernel wrote a moment earlier: they come from the Infinity Cache, not from the reading
    # XCD's own L2.  A rotation over only the three layers' inputs (3 x 8 MB at C2, an eighth of each per XCD) would sit in the 4 MB L2s
    # and read ~1.5 us fast (measured: 11.6 us against ~13.1 us in the step).  So every layer's input is cloned until the rotation's
    # working set is ~128 MB: beyond the L2s, inside the 256 MB Infinity Cache -- as in the step.
    keep_alive = []
    in_bytes = B * N * K_ * (2 if model.activation_dtype == torch.bfloat16 else 4)
    n_clones = max(1, min(8, int(128e6 // (3 * in_bytes))))
    for src, ml in layers:
        z0 = wsf(src + ".Z", model.activation_dtype).view(B * N, K_)
        for q in range(n_clones):
            zq = z0 if q == 0 else z0.clone()
            keep_alive.append(zq)
            op = _lib.operand(zq, ca=wsf(src + ".scale"), cc=wsf(src + ".shift"), relu=True)
            calls.append((op, (C.byref(op), _lib.ptr(wsf(ml + ".wb_hi", torch.bfloat16)), _lib.ptr(wsf(ml + ".wb_lo", torch.bfloat16)), B, N, K_, C_,
                               _lib.ptr(wsf(ml + ".pmax")), _lib.ptr(wsf(ml + ".pq", torch.int32)), _lib.ptr(wsf(ml + ".sumsq")),
                               _lib.ptr(colacc), prec_id, _lib.current_stream())))
    # layer-major -> interleave the layers: consecutive launches belong to different layers, as in the step
    calls = [calls[l * n_clones + q] for q in range(n_clones) for l in range(len(layers))]
    REPS = max(2, 60 // len(calls))
    for _ in range(3):
        for _, a in calls:
            _lib.check(_lib.lib().pn_conv_fwd_max_panel(*a), "pn_conv_fwd_max_panel")
    # the layers' launches in rotation (as in the step, a launch never finds its own operands of the previous launch in the L2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        for _, a in calls:
            _lib.check(_lib.lib().pn_conv_fwd_max_panel(*a), "pn_conv_fwd_max_panel")
    e1.record()
    torch.cuda.synchronize()
    kt = [e0.elapsed_time(e1) * 1e-3 / (REPS * len(calls))] * len(calls)
    k_mean = sum(kt) / len(kt)
    flop_per_launch = 2.0 * 128 * 1024 * B * N
    # algorithmic bytes: pre-BN input rows read once (in their storage type) + the bf16 kernel copy + per-(cloud, channel) max / block /
    # sum of squares
    act_bytes = 2 if model.activation_dtype == torch.bfloat16 else 4
    bytes_per_launch = 128 * act_bytes * B * N + 128 * 1024 * 2 * (2 if args.precision == "bf16x3" else 1) + B * 1024 * 12
    # HBM bytes per launch from rocprofv3 PMC passes of this same command (profiles/, corrected per MI355X_MICROARCH.md: FETCH_SIZE x 2):
    # reported only while the kernel source is the one the counters were collected on
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "round3", "panel_pmc.json")
    if os.path.exists(pmc_file) and (B, N, args.precision) == (32, 1024, "bf16"):
        import hashlib
        with open(pmc_file) as f:
            pmc = json.load(f)
        src_sha = hashlib.sha256(open(os.path.join(ROOT, "pointcloudprocessing_amd", "csrc", "pn_panel.hip"), "rb").read()).hexdigest()
        if pmc.get("pn_panel_hip_sha256") == src_sha:
            traffic = pmc["traffic_bytes_per_launch"]
    achieved = flop_per_launch / k_mean

    out = {
        "metric": "points/sec PointNet fwd+bwd N=1024 B=32 at 1/2/4/8 MI355X vs CPU ref",
        "value": world * B * N * args.steps / dt,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16x3" if args.precision == "bf16x3" else "bf16",
        "data": "synthetic",
        "config": {"workload": f"PointNet-cls training step (fwd + losses + bwd + grad all-reduce + Adam), N={N} points, "
                               f"batch {B} per GPU, profile {args.profile}, {CCLS} classes / {CSEG} parts, random-init weights",
                   "global_batch": world * B, "points_per_cloud": N, "parallelism": f"dp{world}", "launch": graph_mode},
        "roofline": {"bound": "mfma", "kernel": "panel_max_kernel<NS,128> (ConvLayer 128->1024 + BN sums + reduce_max, 3 launches per step)",
                     "achieved": achieved / 1e12, "peak": MFMA_BF16_PEAK / 1e12 / (3 if args.precision == "bf16x3" else 1),
                     "unit": "TFLOP/s", "frac": achieved / (MFMA_BF16_PEAK / (3 if args.precision == "bf16x3" else 1)),
                     "traffic": traffic, "launch_us": k_mean * 1e6, "launches_timed": REPS * len(kt),
                     "timing": f"one HIP event pair around {REPS * len(kt)} back-to-back launches on the launch stream (the step's three layers in rotation over {n_clones} copies of each input: a working set beyond the L2s, inside the Infinity Cache, as in the step); includes the launch boundaries",
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "hbm_frac_if_bandwidth_bound": bytes_per_launch / k_mean / HBM_PEAK},
    }
    # whole-step roofline: cls trunk fwd+bwd 2.61 MFLOP and 18.4 KB of compulsory layer-boundary traffic per point
    # (7.04 MFLOP with the segmentation head trained); the slower of the two roofs bounds the step
    f_alg = (7.04e6 if lw[1] != 0.0 else 2.61e6) * B * N
    b_alg = 18.4e3 * B * N + 16.8e6
    t_roof = max(f_alg / MFMA_BF16_PEAK, b_alg / HBM_PEAK)
    out["step_roofline"] = {"algorithmic_flop": f_alg, "algorithmic_bytes": b_alg, "t_mfma_us": f_alg / MFMA_BF16_PEAK * 1e6,
                            "t_hbm_us": b_alg / HBM_PEAK * 1e6, "bound": "hbm" if b_alg / HBM_PEAK > f_alg / MFMA_BF16_PEAK else "mfma",
                            "frac": t_roof / (dt / args.steps)}
    if world == 1 and args.precision == "bf16" and not args.no_cpu_baseline:
        # the fp32-grade mode beside the headline mode: same workload, same launch form, 60 steps after 10
        m3 = PointNet(CCLS, CSEG, 0.3, 42, precision="bf16x3", device=dev)
        m3.thaw_shared_network(); m3.thaw_input_transform()
        (m3.freeze_classification_head if args.profile == "final" else m3.thaw_classification_head)()
        (m3.freeze_segmentation_head if args.profile == "classification_pretrain" else m3.thaw_segmentation_head)()
        ts3 = TrainStep(m3, KerasAdam(m3.params_flat.data, 1e-4, 7000, 0.7), B, N, lw, use_graph=not args.no_graph)
        ts3.load(pc, y_cls, y_seg, se3)
        torch.cuda.set_stream(ts3.stream)
        for _ in range(13):
            ts3.run()
        torch.cuda.synchronize()
        batches = []                       # the median of three batches of 20 steps: a batch now and then runs 15-20 % slow on this pool
        for _ in range(3):
            t1 = time.perf_counter()
            for _ in range(20):
                ts3.run()
            torch.cuda.synchronize()
            batches.append((time.perf_counter() - t1) / 20)
        dt3 = sorted(batches)[1]
        out["fp32_grade"] = {"precision": "bf16x3", "ms_per_step": dt3 * 1e3, "value": B * N / dt3, "unit": "points/s", "launch": ts3.mode,
                             "note": "split bf16 MFMA operands (3 products, 16 significant bits), fp32 layer-boundary tensors; inference parity vs the fp64 oracle 7e-6"}
        # how far the headline mode's TRAINING-mode forward is from the fp32-grade mode's on the same weights, clouds and dropout masks
        # (batch-statistics BatchNormalization amplifies operand rounding; inference with moving statistics: 1e-4 .. 9e-4)
        with torch.no_grad():
            m3.params_flat.data.copy_(model.params_flat.data)
            gk = torch.Generator().manual_seed(11)
            keep = ((torch.rand(B, 512, generator=gk) >= 0.3).to(torch.uint8).to(dev), (torch.rand(B, 256, generator=gk) >= 0.3).to(torch.uint8).to(dev))
            o16 = model._run_forward(pc, True, {"keep": keep})
            o32 = m3._run_forward(pc, True, {"keep": keep})
            torch.cuda.synchronize()
            out["fp32_grade"]["headline_mode_vs_this_mode_training_forward"] = {
                "classification_probability_max_abs_diff": float((o16[0] - o32[0]).abs().max()),
                "segmentation_probability_max_abs_diff": float((o16[1] - o32[1]).abs().max()),
                "classification_probability_rms_diff": float((o16[0] - o32[0]).pow(2).mean().sqrt()),
                "segmentation_probability_rms_diff": float((o16[1] - o32[1]).pow(2).mean().sqrt()),
                "class_argmax_agreement": float((o16[0].argmax(-1) == o32[0].argmax(-1)).float().mean()),
                "part_argmax_agreement": float((o16[1].argmax(-1) == o32[1].argmax(-1)).float().mean())}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(B, N)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
