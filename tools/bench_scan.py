#!/usr/bin/env python3
"""BASELINE config 5: dense LiDAR scan N=131072 -> voxel-grid (0.25 m) -> FPS (M=8192) -> PointNet inference, one MI355X.

Synthetic scan: 95 % of the points on the hull of the reference's kc-46 cloud (each reference
point replicated with N(0, 0.15 m) noise), 5 % uniform outliers in the bounding box.  Prints one JSON line with the
time of every stage (HIP events on the launch stream).  The same pipeline is checked bit for bit against the NumPy oracle by
tests/test_gpu_ops.py::test_scan_pipeline_c5_matches_oracle (the oracle is test infrastructure: nothing here imports it)."""
import argparse
import json
import os
import re
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_scan(n, seed=20260005):
    rng = np.random.default_rng(seed)
    pts = []
    for line in open(os.path.join(ROOT, "tests", "golden", "kc-46.txt")):
        m = re.match(r"\(([^)]*)\)", line.strip())
        pts.append([float(v) for v in m.group(1).split(",")])
    ref = np.asarray(pts, dtype=np.float32)
    n_hull = int(0.95 * n)
    hull = ref[rng.integers(0, len(ref), n_hull)] + rng.normal(0, 0.15, size=(n_hull, 3)).astype(np.float32)
    lo, hi = ref.min(0) - 1, ref.max(0) + 1
    out = rng.uniform(lo, hi, size=(n - n_hull, 3)).astype(np.float32)
    xyz = np.concatenate([hull, out]).astype(np.float32)
    rng.shuffle(xyz)
    return xyz, lo.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=131072)
    ap.add_argument("--leaf", type=float, default=0.25)
    ap.add_argument("--samples", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from pointcloudprocessing_amd import ops
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    dev = torch.device("cuda:0")
    xyz, origin = make_scan(args.points)
    x = torch.from_numpy(xyz).to(dev)
    model = PointNet(23, 12, 0.3, 42, vanilla=True, precision="bf16", device=dev)   # kc46_lidar_config.json: vanilla
    leaf = (args.leaf,) * 3
    times = {"voxel_ms": [], "fps_ms": [], "inference_ms": []}
    for rep in range(args.reps + 1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        cent, cnt, _ = ops.voxel_downsample(x, leaf, origin)
        ev[1].record()
        V = cent.shape[0]
        M = min(args.samples, V)
        idx = ops.farthest_point_sample(cent.unsqueeze(0).contiguous(), M)
        ev[2].record()
        cloud = cent[idx[0].long()].unsqueeze(0).contiguous()
        cls_idx, part_idx, R = model.predict(cloud)          # class index, per-point part indices (device-side arg-max), pose
        ev[3].record()
        torch.cuda.synchronize()
        if rep:
            times["voxel_ms"].append(ev[0].elapsed_time(ev[1]))
            times["fps_ms"].append(ev[1].elapsed_time(ev[2]))
            times["inference_ms"].append(ev[2].elapsed_time(ev[3]))
    out = {"workload": f"scan N={args.points} -> voxel {args.leaf} m ({V} voxels) -> FPS M={M} -> PointNet(vanilla) inference",
           **{k: float(np.median(v)) for k, v in times.items()},
           "fps_distance_updates_per_s": float(M * V / (np.median(times["fps_ms"]) * 1e-3)),
           "class": int(cls_idx[0]), "part_histogram": torch.bincount(part_idx[0].long(), minlength=12).tolist()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
