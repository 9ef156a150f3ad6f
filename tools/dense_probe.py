"""Time the dense-layer launches of the T-Net / classification head inside a hipGraph (as the step runs them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudprocessing_amd import ops

dev = torch.device("cuda:0")
B, reps = 32, 20
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for K, C_, trans in ((1024, 512, False), (512, 256, False), (256, 4096, False), (256, 9, False), (256, 23, False),
                         (512, 1024, True), (256, 512, True), (4096, 256, True), (23, 256, True)):
        x = torch.randn(B, K, device=dev)
        w = torch.randn((C_, K) if trans else (K, C_), device=dev) * 0.05
        gamma = torch.rand(C_, device=dev) + 0.5; beta = torch.randn(C_, device=dev)
        mm = torch.zeros(C_, device=dev); mv = torch.ones(C_, device=dev)
        cnt = torch.zeros(256, device=dev, dtype=torch.int32)
        kw = dict(trans=trans, counters=cnt) if trans else dict(gamma=gamma, beta=beta, moving_mean=mm, moving_var=mv, bn_mode=1, act=1, counters=cnt)
        ops.dense_layer(x, w, **kw); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            outs = [ops.dense_layer(x, w, **kw) for _ in range(reps)]
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"dense_layer K={K:5d} C={C_:5d} trans={int(trans)}  {e0.elapsed_time(e1) * 1e3 / (10 * reps):7.2f} us/launch")
        del outs, g
    for K, C_ in ((1024, 512), (512, 256), (256, 23)):
        da = torch.randn(B, C_, device=dev); z = torch.randn(B, C_, device=dev); x = torch.randn(B, K, device=dev)
        gamma = torch.rand(C_, device=dev) + 0.5; beta = torch.randn(C_, device=dev); mean = torch.zeros(C_, device=dev); inv = torch.ones(C_, device=dev)
        kw = dict(gamma=gamma, beta=beta, mean=mean, invstd=inv, bn_mode=1, act=1)
        ops.dense_bwd(da, z, x, **kw); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            outs = [ops.dense_bwd(da, z, x, **kw) for _ in range(reps)]
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        # each ops.dense_bwd also runs 3 torch.zeros fills: subtract nothing, just report
        print(f"dense_bwd   K={K:5d} C={C_:5d}          {e0.elapsed_time(e1) * 1e3 / (10 * reps):7.2f} us/call (incl. 3 tiny fills)")
        del outs, g
