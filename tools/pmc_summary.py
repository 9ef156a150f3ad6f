#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one counter per pass: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950) into
per-kernel averages and the panel kernel's HBM traffic per launch, corrected as MI355X_MICROARCH.md (HBM section) prescribes:
FETCH_SIZE counts 128-byte requests at 64 bytes on gfx950 -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.

usage: pmc_summary.py <fetch pass dir> <write pass dir> <out dir> <tag> [<workload text>]
writes <out dir>/<tag>_pmc_fetch_by_kernel.csv, <tag>_pmc_write_by_kernel.csv and panel_pmc.json (stamped with the sha256 of
pn_panel.hip: bench.py reports the traffic figure only while the kernel source is the one the counters were collected on)."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def by_kernel(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    acc = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            acc.setdefault(r["Kernel_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])      # one row per XCD / instance: summed per dispatch
    return {k: list(v.values()) for k, v in acc.items()}


def write_csv(path, data, col):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", f"avg_{col}_KiB", "min", "max"])
        for k, v in sorted(data.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), round(sum(v) / len(v), 1), round(min(v), 1), round(max(v), 1)])


def main():
    fdir, wdir, out, tag = sys.argv[1:5]
    workload = sys.argv[5] if len(sys.argv) > 5 else "B=32 N=1024 bf16 (bench.py default)"
    fetch, write = by_kernel(fdir, "FETCH_SIZE"), by_kernel(wdir, "WRITE_SIZE")
    os.makedirs(out, exist_ok=True)
    write_csv(os.path.join(out, f"{tag}_pmc_fetch_by_kernel.csv"), fetch, "FETCH_SIZE")
    write_csv(os.path.join(out, f"{tag}_pmc_write_by_kernel.csv"), write, "WRITE_SIZE")
    pk = [k for k in fetch if "panel_max_kernel" in k]
    if not pk:
        raise SystemExit("no panel_max_kernel dispatches in the fetch pass")
    k = max(pk, key=lambda n: len(fetch[n]))
    f_kib = sum(fetch[k]) / len(fetch[k])
    w_kib = sum(write[k]) / len(write[k]) if k in write else 0.0
    src = open(os.path.join(ROOT, "pointcloudprocessing_amd", "csrc", "pn_panel.hip"), "rb").read()
    rec = {"kernel": k, "workload": workload, "dispatches": len(fetch[k]), "FETCH_SIZE_KiB_raw": f_kib, "FETCH_bytes_corrected": 2 * f_kib * 1024,
           "WRITE_SIZE_KiB": w_kib, "WRITE_bytes": w_kib * 1024, "traffic_bytes_per_launch": 2 * f_kib * 1024 + w_kib * 1024,
           "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
           "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph  (second pass: --pmc WRITE_SIZE)",
           "pn_panel_hip_sha256": hashlib.sha256(src).hexdigest()}
    json.dump(rec, open(os.path.join(out, "panel_pmc.json"), "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
