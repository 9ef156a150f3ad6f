"""(Historical reproducer -- at commit c5040b3 the backward pass cleared the gradient buffer with hipMemsetAsync and modes eg / ege / gg
failed within a few steps; the library has used a zero-fill kernel since, so every mode is clean now.  DESIGN.md section 8a.)
Find what makes the hipGraph arm produce garbage gradients when other (differently initialised) models run between replays.
usage: graph_hunt.py MODE ; detector = gradient slots of the moving statistics must stay exactly 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudprocessing_amd.pointnet.PointNet import PointNet
from pointcloudprocessing_amd.engine import TrainStep
from pointcloudprocessing_amd.optim import KerasAdam

mode = sys.argv[1]
variant = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
B, N = 8, 256
g = torch.Generator().manual_seed(0)
pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)

plan = {"g": [True], "eg": [False, True], "ge": [True, False], "ege": [False, True, False], "gg": [True, True],
        "ee": [False, False], "eg_noadam": [False, True], "g_torch": [True]}[mode]
arms = []
for use_graph in plan:
    m = PointNet(23, 12, 0.0, 42, precision="bf16x3", device=dev)
    opt = KerasAdam(m.params_flat.data, 1e-3, 7000, 0.7)
    ts = TrainStep(m, opt, B, N, (1.0, 1.0, 1.0), use_graph=use_graph)
    if mode == "eg_noadam":
        ts.world = 2                      # keeps Adam out of graph 1 (graph 2 = Adam); no process group: all_reduce is skipped below
        ts.dist = type("D", (), {"all_reduce": staticmethod(lambda t: None)})
    arms.append((m, opt, ts))
junk = None
side = torch.cuda.Stream()
if variant == "side":
    torch.cuda.set_stream(side)
pre = {}
for step in range(1, 21):
    for ai, (m, opt, ts) in enumerate(arms):
        pre[ai] = m.params_flat.data.clone()
        ts(pc, y_cls, y_seg, se3)
        if variant == "sync":
            torch.cuda.synchronize()
        if variant == "null":            # the failing arrangement: replay the captured graph straight into the null stream
            pass
        if mode == "g_torch":             # unrelated torch work between replays
            junk = torch.randn(1 << 22, device=dev).sort().values
    torch.cuda.synchronize()
    for i, (m, opt, ts) in enumerate(arms):
        mv = [float(m.grads_flat[s["offset"]: s["offset"] + s["rows"] * s["cols"]].abs().max()) for n, s in m._weights.slots.items()
              if n.endswith("moving_mean") or n.endswith("moving_var")]
        gmax = float(m.grads_flat.abs().max())
        if max(mv) != 0 or not (gmax < 1e6):
            print(f"MODE {mode} {variant}: arm {i} ({ts.mode}) step {step}: max|G[moving]| {max(mv):.4g}  max|G| {gmax:.4g}  loss {float(m.scalars[0]):.4g}")
            if variant == "diag":
                import ctypes as C
                from pointcloudprocessing_amd._lib import lib
                x = PointNet(23, 12, 0.0, 42, precision="bf16x3", device=dev)
                x.params_flat.data.copy_(pre[i])
                x.fused_loss_step(pc, y_cls, y_seg, se3, (1.0, 1.0, 1.0))
                torch.cuda.synchronize()
                print("  recomputed eagerly from the pre-step weights: max|G|", float(x.grads_flat.abs().max()), "loss", float(x.scalars[0]))
                name = C.create_string_buffer(128); off = C.c_int64(); nb = C.c_int64(); k = 0; shown = 0
                while lib().pn_model_ws_entry(C.byref(m._desc), B, N, 1, k, name, 128, C.byref(off), C.byref(nb)) == 0:
                    a = m._workspace(B, N, True)[off.value: off.value + nb.value]
                    b = x._workspace(B, N, True)[off.value: off.value + nb.value]
                    if not torch.equal(a, b) and shown < 60:
                        nf = nb.value // 4
                        fa, fb = a[:nf * 4].view(torch.float32), b[:nf * 4].view(torch.float32)
                        ndiff = int((a != b).sum())
                        print(f"  ws {k:3d} {name.value.decode():14s} bytes {nb.value:9d} differing bytes {ndiff:9d}  max|graph| {float(fa.abs().nan_to_num(1e38).max()):.4g}  max|eager| {float(fb.abs().max()):.4g}")
                        shown += 1
                    k += 1
            sys.exit(0)
print(f"MODE {mode} {variant}: clean after 20 steps; losses {[round(float(a[0].scalars[0]), 3) for a in arms]}")
