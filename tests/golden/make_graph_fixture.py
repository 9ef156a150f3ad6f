#!/usr/bin/env python3
"""Builds tests/golden/ref_tf2onnx_graph_f15.json from the tf2onnx DEBUG dump the reference's own training run logged
(/root/reference/point_cloud_analysis/models/f15_scale_lidar/log_20260126_16*0916.log, the export of pointnet_train.py:238-248).

Run in the build container only (the reference never travels to the GPU box); the output is DATA: for every TensorFlow node of
the inference graph its op type, name and the shapes the converter printed, plus the converter's own op counter line.  It pins the
STRUCTURE of the reference model (op histogram, kernel shapes, placeholder and output shapes) -- the only model artefact the
reference still holds, since every .keras / .onnx blob is stripped.  Numeric parity stays unpinned.
"""
import ast
import glob
import json
import os
import re
import sys

REF = "/root/reference/point_cloud_analysis/models/f15_scale_lidar"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_tf2onnx_graph_f15.json")


def main():
    logs = sorted(glob.glob(os.path.join(REF, "log_20260126_16*0916.log")))
    if not logs:
        sys.exit("reference log not found")
    lines = open(logs[0], encoding="utf-8", errors="replace").read().split("\n")
    # the first export in the log (profile classification_pretrain): from the converter banner to its op counter
    start = next(i for i, l in enumerate(lines) if "tf2onnx.tfonnx - INFO - Using tensorflow=" in l)
    end = next(i for i, l in enumerate(lines) if i > start and "tensorflow ops: Counter(" in l)
    banner = lines[start].split(" - INFO - ")[-1]
    counter = ast.literal_eval(re.search(r"Counter\((\{.*\})\)", lines[end]).group(1))
    nodes, i = [], start
    shape_re = re.compile(r"^\t(.*?)=(?:(\w+), )?(\[[^\]]*\]), (\d+)$")
    while i < end:
        if "tf2onnx.tfonnx - DEBUG - Process node: " in lines[i]:
            node = {"name": lines[i].split("Process node: ")[1], "op": None, "inputs": [], "outputs": []}
            i += 1
            section = None
            while i < end and not re.match(r"^\d{4}-\d\d-\d\d ", lines[i]):
                l = lines[i]
                if l.startswith("OP="):
                    node["op"] = l[3:]
                elif l.startswith("Inputs:"):
                    section = "inputs"
                elif l.startswith("Outpus:"):
                    section = "outputs"
                elif l.startswith("\t") and section:
                    m = shape_re.match(l)
                    if m:
                        ent = {"tensor": m.group(1), "shape": json.loads(m.group(3))}
                        if m.group(2):
                            ent["producer_op"] = m.group(2)
                        node[section].append(ent)
                i += 1
            nodes.append(node)
        else:
            i += 1
    hist = {}
    for n in nodes:
        hist[n["op"]] = hist.get(n["op"], 0) + 1
    json.dump({"source": "models/f15_scale_lidar/log_20260126_16*0916.log:" + f"{start + 1}-{end + 1}", "converter": banner,
               "tensorflow_ops_counter": counter, "op_histogram_of_dumped_nodes": hist, "nodes": nodes}, open(OUT, "w"), indent=0)
    print(f"{len(nodes)} nodes -> {OUT}; ops {hist}")


if __name__ == "__main__":
    main()
