"""NumPy oracles for farthest point sampling and voxel-grid downsampling -- TEST INFRASTRUCTURE.

The reference has no FPS and no voxel grid (the only resize-to-N logic is truncate-first-N /
random duplicate padding, point_cloud_analysis/pointcloud/PointCloudSet.py:443-470, restated here as
`adjust_to_input_width`).  The specification of the two samplers is build-defined (include/pointnet_hip.h)
and this file is its executable statement: PARITY UNPINNED against the reference by construction.
"""
from __future__ import annotations

import numpy as np


def fps(xyz: np.ndarray, m: int, start_idx: int = 0):
    """xyz (N,3) float32.  Returns (idx (m,) int32 in selection order, mindist (N,) float32).
    d = (dx*dx + dy*dy) + dz*dz in float32 without fused multiply-add; ties -> lowest index."""
    xyz = np.asarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    md = np.full(n, np.inf, dtype=np.float32)
    idx = np.zeros(m, dtype=np.int32)
    cur = int(start_idx)
    for it in range(m):
        idx[it] = cur
        if it == m - 1:
            break
        d = xyz - xyz[cur]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        md = np.minimum(md, d2.astype(np.float32))
        cur = int(np.argmax(md))          # np.argmax returns the first maximum
    return idx, md


def voxel_downsample(xyz: np.ndarray, leaf, origin, labels=None, n_labels: int = 0):
    """Returns (centroids (V,3) f32, counts (V,) i32, majority (V,) i32 or None), voxels ordered by
    ascending (kz, ky, kx); centroid accumulated in float64 in point-index order; majority ties -> lowest
    label."""
    xyz = np.asarray(xyz, dtype=np.float32)
    leaf = np.asarray(leaf, dtype=np.float32)
    origin = np.asarray(origin, dtype=np.float32)
    k = np.floor((xyz - origin) / leaf).astype(np.int64)
    assert (k >= 0).all() and (k < (1 << 21)).all(), "voxel key out of range"
    key = (k[:, 2] << 42) | (k[:, 1] << 21) | k[:, 0]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    heads = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    ends = np.r_[heads[1:], len(ks)]
    cent = np.zeros((len(heads), 3), dtype=np.float32)
    cnt = np.zeros(len(heads), dtype=np.int32)
    maj = np.zeros(len(heads), dtype=np.int32) if labels is not None else None
    x64 = xyz.astype(np.float64)
    for v, (s, e) in enumerate(zip(heads, ends)):
        ids = order[s:e]
        acc = np.zeros(3, dtype=np.float64)
        for i in ids:                      # point-index order, sequential float64 adds
            acc += x64[i]
        cent[v] = (acc * (1.0 / (e - s))).astype(np.float32)
        cnt[v] = e - s
        if labels is not None:
            h = np.bincount(labels[ids], minlength=n_labels)[:n_labels]
            maj[v] = int(np.argmax(h))
    return cent, cnt, maj


def adjust_to_input_width(observations: np.ndarray, part_labels: np.ndarray, width: int, rng=None):
    """PointCloudSet._adjust_to_input_width (pointcloud/PointCloudSet.py:443-470): keep the first `width`
    rows, or append rows drawn with np.random.uniform(0, n) -> int (labels kept aligned)."""
    n = observations.shape[0]
    if n > width:
        return observations[:width], part_labels[:width]
    rng = rng or np.random
    rep = rng.uniform(0, n, width - n).astype(np.int_)
    return (np.concatenate((observations, observations[rep]), axis=0),
            np.concatenate((part_labels, part_labels[rep]), axis=0))
