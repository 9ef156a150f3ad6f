"""Does a captured fork/join run its branches concurrently on this runtime?  Two chains of n small dependent kernels, captured
(a) back to back on one stream, (b) on two streams forked and joined with events.  Prints the replay time of each."""
import json
import time

import torch

dev = torch.device("cuda:0")
n = 60


def chain(x):
    for _ in range(n):
        x.mul_(1.0001)


def capture(two_streams, size):
    a = torch.ones(size, device=dev)
    b = torch.ones(size, device=dev)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(a); chain(b)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            if two_streams:
                ev = torch.cuda.Event(); ev.record(s)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    chain(b)
                    ev2 = torch.cuda.Event(); ev2.record(side)
                chain(a)
                s.wait_event(ev2)
            else:
                chain(a); chain(b)
    return g, (a, b)


for size in (4096, 1 << 20):
    for two in (False, True):
        g, keep = capture(two, size)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 50 * 1e6
        print(json.dumps({"elements": size, "two_streams": two, "kernels": 2 * n, "replay_us": round(us, 1), "us_per_kernel": round(us / (2 * n), 2)}), flush=True)
