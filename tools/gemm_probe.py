"""Time single conv_fwd launches inside a hipGraph (the way the step runs them).  PN_GEMM_DBG=bitmask ablates parts of
the kernel (1 no output stores, 2 no statistics, 4 no activation loads, 8 no weight loads); PN_GEMM_NARROW=1 forces 128x64 tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudprocessing_amd import ops, _lib

dev = torch.device("cuda:0")
B, N = 32, 1024
reps = 20
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for K, C_ in ((64, 64), (64, 128), (128, 128), (64, 512), (512, 256)):
        x = torch.randn(B * N, K, device=dev)
        ca = torch.rand(K, device=dev) + 0.5; cc = torch.randn(K, device=dev)
        w = torch.randn(K, C_, device=dev) * 0.1
        op = _lib.operand(x, ca=ca, cc=cc, ld=K, relu=True)
        for prec, pn in ((1, "bf16"), (3, "bf16x3")):
            ops.conv_fwd(op, w, B, N, K, C_, prec)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                outs = [ops.conv_fwd(op, w, B, N, K, C_, prec) for _ in range(reps)]
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (10 * reps)
            mb = (B * N * (K + C_) * 4 + K * C_ * 4) / 1e6
            print(f"K={K:4d} C={C_:4d} {pn:7s} {us:7.2f} us/launch   {mb:6.1f} MB  -> {mb / us * 1e-3 * 1e3:6.0f} GB/s  dbg={os.environ.get('PN_GEMM_DBG', '0')}")
            del outs, g
