#!/usr/bin/env python3
"""Times pn_panel_finalize alone (50 back-to-back launches) with its phases switched off one at a time (PN_FIN_DBG)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import _lib, ops   # noqa: E402

dev = torch.device("cuda:0")
K, C_ = 128, 1024
for B, N in ((32, 1024), (32, 4096)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B * N, K, generator=g).to(dev)
    w = (torch.randn(K, C_, generator=g) / 11).to(dev)
    gamma, beta = torch.randn(C_, generator=g).to(dev), torch.randn(C_, generator=g).to(dev)
    op = _lib.operand(x, relu=True)
    wf = ops.weights_prep(w, gamma)
    for prec in (1, 3):
        pmax, pblk, sumsq, sumz = ops.conv_fwd_max_panel(op, wf, B, N, K, C_, prec)
        mm, mv = torch.zeros(C_, device=dev), torch.ones(C_, device=dev)
        for dbg in (0, 1, 2, 4, 8, 15):
            os.environ["PN_FIN_DBG"] = str(dbg)
            for _ in range(3):
                ops.panel_finalize(pmax, pblk, sumsq, sumz, wf, prec, B, N, K, gamma, beta, mm, mv, training=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                ops.panel_finalize(pmax, pblk, sumsq, sumz, wf, prec, B, N, K, gamma, beta, mm, mv, training=True)
            e1.record()
            torch.cuda.synchronize()
            print(json.dumps({"B": B, "N": N, "prec": prec, "dbg": dbg, "us": round(e0.elapsed_time(e1) * 1e3 / 50, 2)}), flush=True)
os.environ.pop("PN_FIN_DBG", None)
