"""Shared test helpers: synthetic Aftr collects in the reference's on-disk format."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# f15_lidar_config.json:4-42
F15_CLASSES = ["f-15_model", "a-10", "b-1b", "b-2", "c-5", "c-12", "c-17a", "c-32", "c-130j", "e-3", "f-15e", "f-16", "f-18e", "f-22",
               "g-iii", "kc-46", "kc-135", "lj-25", "mig-29", "mq-20", "su-27", "vc-25a", "x-47b"]
F15_PARTS = ["wing", "fuselage", "engine", "hstab", "vstab", "landing_gear", "armament", "boom_wing", "boom_hull", "boom_hose", "dish",
             "probe"]


def make_collect(tmp, name, n_frames, width_hint=None, seed=0):
    """A synthetic Aftr collect in the reference's on-disk format, built from the two labelled reference clouds."""
    rng = np.random.default_rng(seed)
    d = os.path.join(tmp, name)
    os.makedirs(os.path.join(d, "Lidar"))
    lines = {fn: open(os.path.join(GOLD, fn)).read().strip().split("\n") for fn in ("kc-46.txt", "f-15_model.txt")}
    with open(os.path.join(d, f"_palindrome_state__{name}.log"), "w") as f:
        f.write("Time   Frame   Sensor Pose   Tanker Pose\n")
        for i in range(n_frames):
            def pose():
                q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
                m = np.eye(4); m[:3, :3] = q; m[:3, 3] = rng.normal(size=3) * 10
                return m
            sp, tp = pose(), pose()
            vals = [f"{v:.9f}" for v in sp.T.reshape(-1)] + [f"{v:.9f}" for v in tp.T.reshape(-1)]   # column-major
            f.write(f"{i * 0.1:.3f} {i} " + " ".join(vals) + "\n")
    for i in range(n_frames):
        src = lines["kc-46.txt" if i % 2 == 0 else "f-15_model.txt"]
        with open(os.path.join(d, "Lidar", f"frame_{i}.txt"), "w") as f:
            for ln in src:
                m = re.match(r"\(([^)]*)\)(.*)", ln)
                xyz = np.array([float(v) for v in m.group(1).split(",")]) + rng.normal(size=3) * 0.01
                f.write(f"({xyz[0]:.3f}, {xyz[1]:.3f}, {xyz[2]:.3f}){m.group(2)}\n")
    return d


