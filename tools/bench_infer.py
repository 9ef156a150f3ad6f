#!/usr/bin/env python3
"""Inference timing (training=False: moving statistics everywhere): PointNet forward at the BASELINE shapes, replayed from a hipGraph
(device time: the eager loop is bound by the host's launches).  One JSON line per (shape, plan): the fused-chain plan (each max-pooled
chain conv -> conv -> conv + reduce_max in ONE launch, pn_panel.hip: chain_max_kernel) and, in a child process with PN_CHAIN_FUSE=0,
the layer-by-layer plan.  usage: bench_infer.py [--child]"""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    dev = torch.device("cuda:0")
    fused = os.environ.get("PN_CHAIN_FUSE", "1") != "0"
    for B, N, vanilla in ((32, 1024, False), (32, 2048, False), (8, 4096, False), (32, 4096, False), (1, 8192, True)):
        m = PointNet(23, 12, 0.3, 42, vanilla=vanilla, precision="bf16", device=dev)
        g = torch.Generator().manual_seed(B * N)
        pc = (torch.rand(B, N, 3, generator=g) * 20 - 10).to(dev)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(3):
                m._run_forward(pc, False, None)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                m._run_forward(pc, False, None)
            for _ in range(5):
                gr.replay()
            # seven batches of 40 replays, the median batch: a batch now and then runs 3-5x slower on this pool (seen on the fifth shape of
            # one run and the fourth of the next, both plans), whatever the kernels
            batches = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(40):
                    gr.replay()
                e1.record()
                torch.cuda.synchronize()
                batches.append(e0.elapsed_time(e1) * 25.0)
        us = sorted(batches)[len(batches) // 2]
        print(json.dumps({"what": "PointNet inference forward (hipGraph replay)", "B": B, "N": N, "vanilla": vanilla, "precision": "bf16",
                          "plan": "fused chains" if fused else "layer by layer", "us_per_forward": round(us, 1), "us_fastest_batch": round(min(batches), 1), "us_slowest_batch": round(max(batches), 1),
                          "points_per_s": round(B * N / us * 1e6)}), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        run()
    else:
        run()
        env = dict(os.environ, PN_CHAIN_FUSE="0")
        sys.stdout.flush()
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False)
