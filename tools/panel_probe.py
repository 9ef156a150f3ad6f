"""Time the fused ConvLayer(128->1024)+reduce_max launch inside a hipGraph, with and without the BatchNorm sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudprocessing_amd import ops, _lib

dev = torch.device("cuda:0")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for B, N in ((32, 1024), (32, 4096)):
        K, C_ = 128, 1024
        x = torch.randn(B * N, K, device=dev)
        ca = torch.rand(K, device=dev) + 0.5; cc = torch.randn(K, device=dev)
        w = torch.randn(K, C_, device=dev) * 0.1
        sgn = torch.ones(C_, device=dev)
        op = _lib.operand(x, ca=ca, cc=cc, ld=K, relu=True)
        for stats in (True, False):
            for prec, pn in ((1, "bf16"), (3, "bf16x3")):
                ops.conv_fwd_max_panel(op, w, B, N, K, C_, sgn, prec, want_stats=stats); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    outs = [ops.conv_fwd_max_panel(op, w, B, N, K, C_, sgn, prec, want_stats=stats) for _ in range(10)]
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    g.replay()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 100          # includes one weights_prep launch (~5 us) per call
                fl = 2.0 * K * C_ * B * N
                print(f"B={B} N={N} {pn:7s} stats={int(stats)}  {us:7.2f} us per (prep + panel) ; panel alone ~{us - 5.0:6.2f} us -> {fl / ((us - 5.0) * 1e-6) / 1e12:6.1f} TFLOP/s")
                del outs, g
