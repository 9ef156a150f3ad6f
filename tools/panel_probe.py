#!/usr/bin/env python3
"""Times the fused ConvLayer(128->1024) + BN statistics + reduce_max panel kernel (pn_conv_fwd_max_panel) alone, over shapes,
operand precisions and panel heights: 50 back-to-back launches between one HIP event pair, random (sign-mixed) operands.
Prints one JSON line per case with us per launch, TFLOP/s and the fraction of the dense bf16 MFMA peak."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import _lib, ops   # noqa: E402

dev = torch.device("cuda:0")
K, C_ = 128, 1024
ABLATE = os.environ.get("PROBE_ABLATE", "0") == "1"      # time the PN_PANEL_DBG variants (bf16, statistics) instead of the product kernel
for B, N in (((32, 1024), (32, 4096)) if ABLATE else ((32, 1024), (32, 4096), (8, 4096), (32, 2048))):
    g = torch.Generator().manual_seed(B * N)
    x = torch.randn(B * N, K, generator=g).to(dev)
    w = (torch.randn(K, C_, generator=g) / 11).to(dev)
    gamma = torch.randn(C_, generator=g).to(dev)
    sc, sh = (torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.3).to(dev)
    wf = ops.weights_prep(w, gamma)
    for prec, h16 in (((1, 0),) if ABLATE else ((1, 1), (1, 0), (3, 0))):     # h16: the activations are stored as bf16 (the step's 'bf16' mode)
      op = _lib.operand(x.to(torch.bfloat16) if h16 else x, ca=sc, cc=sh, relu=True)
      if True:
        for rows in (64,):
            for stats, dbg in ([(True, d) for d in (0, 1, 4, 8, 9, 13)] if ABLATE else [(True, 0), (False, 0)]):
                os.environ["PN_PANEL_DBG"] = str(dbg)
                for _ in range(3):
                    outs = ops.conv_fwd_max_panel(op, wf, B, N, K, C_, prec, want_stats=stats)
                T = outs[0].shape[0]
                pmax, pblk = outs[0], outs[1]
                sumsq, a1 = outs[2], outs[3]      # (sumz)
                args = (_lib.C.byref(op), _lib.ptr(wf[0]), _lib.ptr(wf[1]), B, N, K, C_, _lib.ptr(pmax), _lib.ptr(pblk), _lib.ptr(sumsq), _lib.ptr(a1),
                        prec, _lib.current_stream())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(50):
                    _lib.check(_lib.lib().pn_conv_fwd_max_panel(*args), "pn_conv_fwd_max_panel")
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 50
                tf = 2.0 * K * C_ * B * N / (us * 1e-6) / 1e12
                print(json.dumps({"B": B, "N": N, "prec": prec, "bf16_source": h16, "panel_rows": rows, "stats": stats, "dbg": dbg, "tiles": T, "us": round(us, 2),
                                  "TFLOPs": round(tf, 1), "frac_of_2.5PF": round(tf / 2500 / (3 if prec == 3 else 1), 4)}), flush=True)
os.environ.pop("PN_PANEL_DBG", None)
