import os, sys
sys.path.insert(0, '/root/repo')
import torch
torch.manual_seed(1234)
from pointcloudprocessing_amd.engine import TrainStep
from pointcloudprocessing_amd.optim import KerasAdam
from pointcloudprocessing_amd.pointnet.PointNet import PointNet
dev = torch.device("cuda:0")
B, N = 32, 1024
g = torch.Generator().manual_seed(B * N)
pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
m = PointNet(23, 12, 0.3, 42, precision="bf16", device=dev)
wg = torch.Generator().manual_seed(7)
for n, s in m._weights.slots.items():          # seed the (unseeded) classification head too
    if n.startswith("mlp_cls") and n.endswith("kernel"):
        v = m._weights.view(n); lim = (6.0 / (v.shape[0] + v.shape[1])) ** 0.5
        v.copy_(((torch.rand(v.shape, generator=wg) * 2 - 1) * lim).to(dev))
opt = KerasAdam(m.params_flat.data, 1e-3, 7000, 0.7)
ts = TrainStep(m, opt, B, N, (1.0, 1.0, 1.0))
ts._mask_seed = 99
with torch.cuda.stream(ts.stream):
    for i in range(3001):
        ts(pc, y_cls, y_seg, se3)
        if i % 500 == 0:
            torch.cuda.synchronize(); sc = m.scalars
            print(f"rows={os.environ.get('PN_PANEL_ROWS','128')} step {i}: cls {float(sc[0])/B:.4f} acc {float(sc[1])/B:.2f}  seg {float(sc[2])/(B*N):.4f} acc {float(sc[3])/(B*N):.3f}  mse {float(sc[4])/(B*9):.5f}")
