#!/usr/bin/env python3
"""Time pn_fps over a few cloud shapes (HIP events, median of 5).  --lib PATH loads another build of the library, to compare kernels."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(32, 1024, 256), (8, 4096, 1024), (1, 16384, 4096), (1, 20254, 8192), (1, 21504, 2048), (1, 65536, 2048), (1, 131072, 1024)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    from pointcloudprocessing_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from pointcloudprocessing_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    out = {}
    for B, N, M in SHAPES:
        x = torch.from_numpy(rng.uniform(-10, 10, size=(B, N, 3)).astype(np.float32)).to(dev)
        ts = []
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            idx = ops.farthest_point_sample(x, M)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts))
        out[f"B{B}_N{N}_M{M}"] = {"ms": round(ms, 3), "us_per_round": round(1e3 * ms / M, 3), "checksum": int(idx.long().sum())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
