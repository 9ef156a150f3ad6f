"""Measure the per-kernel cost of a chain of dependent trivial kernels, eager vs hipGraph replay (MI355X)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudprocessing_amd import _lib
from pointcloudprocessing_amd._lib import lib, ptr, current_stream

dev = torch.device("cuda:0")
g = torch.randn(1024, device=dev)
s = torch.empty_like(g)
N = 200
def chain():
    for _ in range(N):
        lib().pn_sign(ptr(g), 1024, ptr(s), current_stream())
for _ in range(3):
    chain()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    chain()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 20 / N * 1e6:.2f} us per kernel")
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    chain()
gr.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    gr.replay()
torch.cuda.synchronize()
print(f"hipGraph: {(time.perf_counter() - t0) / 50 / N * 1e6:.2f} us per kernel")
# two independent chains on two streams inside one graph
s2 = torch.empty_like(g)
side = torch.cuda.Stream()
gr2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr2):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(N):
            lib().pn_sign(ptr(g), 1024, ptr(s2), current_stream())
    chain()
    main.wait_stream(side)
gr2.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    gr2.replay()
torch.cuda.synchronize()
print(f"hipGraph, 2 parallel chains of {N}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us total ({(time.perf_counter() - t0) / 50 / N * 1e6:.2f} us per kernel-pair)")
