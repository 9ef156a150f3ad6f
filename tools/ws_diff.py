"""Dump every workspace entry + gradients after one fused step (mode dump), or compare against such a dump (mode cmp): finds the first
buffers that differ between two builds / settings of the library."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudprocessing_amd.pointnet.PointNet import PointNet
from pointcloudprocessing_amd._lib import lib

mode, path = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
B, N = 16, 136
g = torch.Generator().manual_seed(0)
pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
m = PointNet(23, 12, 0.0, 42, precision="bf16x3", device=dev)
wts = torch.Generator().manual_seed(5)
m.params_flat.data.copy_(torch.randn(m.params_flat.numel(), generator=wts).to(dev) * 0.05)
for n, s in m._weights.slots.items():
    if n.endswith("moving_var") or n.endswith("gamma"):
        m.params_flat.data[s["offset"]: s["offset"] + s["rows"] * s["cols"]].abs_().add_(0.5)
m._workspace(B, N, True).zero_()
m.fused_loss_step(pc, y_cls, y_seg, se3, (1.0, 1.0, 1.0))
torch.cuda.synchronize()
name = C.create_string_buffer(128); off = C.c_int64(); nb = C.c_int64()
out, i = {}, 0
while lib().pn_model_ws_entry(C.byref(m._desc), B, N, 1, i, name, 128, C.byref(off), C.byref(nb)) == 0:
    out[name.value.decode()] = m._workspace(B, N, True)[off.value: off.value + nb.value].clone().cpu()
    i += 1
out["__grads__"] = m.grads_flat.clone().cpu().view(torch.uint8)
if mode == "dump":
    torch.save(out, path)
    print("dumped", len(out), "entries")
else:
    ref = torch.load(path)
    for k in out:
        if k not in ref or ref[k].numel() != out[k].numel():
            print("layout differs:", k); continue
        if not torch.equal(ref[k], out[k]):
            nf = out[k].numel() // 4
            a, b = ref[k][:nf * 4].view(torch.float32), out[k][:nf * 4].view(torch.float32)
            d = (a - b).abs().nan_to_num(1e30)
            print(f"{k:16s} differs: max |d| {float(d.max()):.4g}  max |ref| {float(a.abs().nan_to_num(0).max()):.4g}  count {(d > 0).sum().item()} / {nf}")
