#!/usr/bin/env python3
"""Launches the product panel kernel (bf16 operands, bf16 source, statistics) 20 times at B=32, N=4096 and at N=1024 -- the target of
rocprofv3 --pmc passes that look at where its cycles go (SQ counters)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import _lib, ops   # noqa: E402

dev = torch.device("cuda:0")
K, C_ = 128, 1024
for B, N in ((32, 4096), (32, 1024)):
    g = torch.Generator().manual_seed(B * N)
    x = torch.randn(B * N, K, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(K, C_, generator=g) / 11).to(dev)
    gamma = torch.randn(C_, generator=g).to(dev)
    sc, sh = (torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.3).to(dev)
    op = _lib.operand(x, ca=sc, cc=sh, relu=True)
    wf = ops.weights_prep(w, gamma)
    for _ in range(20):
        ops.conv_fwd_max_panel(op, wf, B, N, K, C_, 1, want_stats=True)
    torch.cuda.synchronize()
print("done")
