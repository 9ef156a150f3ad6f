#!/usr/bin/env python3
"""the C5 inference leg alone (B=1, N=8192, vanilla, bf16), 50 eager forwards: for rocprofv3 --kernel-trace --stats"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd.pointnet.PointNet import PointNet   # noqa: E402

dev = torch.device("cuda:0")
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 8192)
m = PointNet(23, 12, 0.3, 42, vanilla=(B == 1), precision="bf16", device=dev)
pc = (torch.rand(B, N, 3, generator=torch.Generator().manual_seed(1)) * 20 - 10).to(dev)
for _ in range(50):
    m._run_forward(pc, False, None)
torch.cuda.synchronize()
