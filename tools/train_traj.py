#!/usr/bin/env python3
"""Loss trajectory of the fixed-batch training loop of tests/test_gpu_train.py (diagnostic): per step-layout the classification loss at
steps 0, 20, 40, 61, 123."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd.engine import TrainStep          # noqa: E402
from pointcloudprocessing_amd.optim import KerasAdam           # noqa: E402
from pointcloudprocessing_amd.pointnet.PointNet import PointNet  # noqa: E402

dev = torch.device("cuda:0")
B, N = 8, 256
g = torch.Generator().manual_seed(0)
pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
w0 = None
for prec in ("bf16x3", "bf16"):
    for use_graph, aux, split in ((False, True, False), (False, False, False), (True, False, True)):
        m = PointNet(23, 12, 0.0, 42, precision=prec, device=dev)
        if w0 is None:
            w0 = m.params_flat.data.clone()
        else:
            m.params_flat.data.copy_(w0)
        opt = KerasAdam(m.params_flat.data, 1e-3, 7000, 0.7)
        ts = TrainStep(m, opt, B, N, (1.0, 1.0, 1.0), use_graph=use_graph, aux=aux, split_optimizer=split)
        losses = []
        for i in range(124):
            ts(pc, y_cls, y_seg, se3)
            losses.append(float(m.scalars[0]) / B)
        print(json.dumps({"prec": prec, "graph": use_graph, "aux": aux, "split": split,
                          "loss": [round(losses[i], 4) for i in (0, 1, 5, 10, 20, 40, 61, 90, 123)]}), flush=True)
