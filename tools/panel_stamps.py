#!/usr/bin/env python3
"""Where a launch of the panel kernel spends its time: runs the PN_PANEL_DBG=16 variant (the product kernel + s_memtime stamps of
wave 0 at the phase boundaries, left in the slot's pq row) and prints, per shape, the median over the slots of every phase in shader
cycles.  Phases: 0 entry -> 1 prologue requests issued -> 2 coefficient table visible -> 3 first panel staged -> per panel: chains 0..3 issued, first wave group's conversion done, chains ..5 issued,
second group's conversion done, last chain + epilogue done, barrier passed -> maxima written -> end.  Printed for wave 0 and wave 4."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import _lib, ops   # noqa: E402

dev = torch.device("cuda:0")
K, C_ = 128, 1024
for B, N in ((32, 1024), (32, 4096)):
    g = torch.Generator().manual_seed(B * N)
    x = torch.randn(B * N, K, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(K, C_, generator=g) / 11).to(dev)
    gamma = torch.randn(C_, generator=g).to(dev)
    sc, sh = (torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.3).to(dev)
    op = _lib.operand(x, ca=sc, cc=sh, relu=True)
    wf = ops.weights_prep(w, gamma)
    os.environ["PN_PANEL_DBG"] = "16"
    for _ in range(3):
        outs = ops.conv_fwd_max_panel(op, wf, B, N, K, C_, 1, want_stats=True)
    torch.cuda.synchronize()
    pq = outs[1].cpu().numpy().astype(np.int64) & 0xffffffff
    n = int(pq[0, 0])
    for w, nm in ((0, "wave0"), (1, "wave4")):
        st = pq[:, 1 + 64 * w:1 + 64 * w + n]
        d = (st[:, 1:] - st[:, :-1]) & 0xffffffff
        span = ((st[:, n - 1] - st[:, 0]) & 0xffffffff)
        print(json.dumps({"B": B, "N": N, "wave": nm, "slots": int(st.shape[0]), "stamps": n, "phase_cycles_median": [int(v) for v in np.median(d, axis=0)],
                          "wg_span_median": int(np.median(span)), "wg_span_max": int(span.max())}), flush=True)
os.environ.pop("PN_PANEL_DBG", None)
