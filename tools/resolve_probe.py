#!/usr/bin/env python3
"""Times pn_max_resolve alone (the staging + row resolution that the backward scatter kernel embeds), uniform random blocks."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import _lib, ops   # noqa: E402
dev = torch.device("cuda:0")
K, C_ = 128, 1024
for B, N in ((32, 1024), (32, 4096)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B * N, K, generator=g).to(dev)
    w = (torch.randn(K, C_, generator=g) / 11).to(dev)
    gamma = torch.randn(C_, generator=g).to(dev)
    sc, sh = (torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.3).to(dev)
    op = _lib.operand(x, ca=sc, cc=sh, relu=True)
    wf = ops.weights_prep(w, gamma)
    for skew in (0, 1):
        argq = torch.randint(0, N // 32, (B, C_), generator=g, dtype=torch.int32)
        if skew:
            argq[:, :512] = 3                      # half of every cloud's channels peak in one 32-row block
        argq = argq.to(dev)
        for prec in (1, 3):
            for _ in range(3):
                ops.max_resolve(op, wf, argq, B, N, K, C_, prec)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50):
                ops.max_resolve(op, wf, argq, B, N, K, C_, prec)
            e1.record(); torch.cuda.synchronize()
            print(json.dumps({"B": B, "N": N, "prec": prec, "skewed": skew, "us": round(e0.elapsed_time(e1) * 1e3 / 50, 2)}), flush=True)
