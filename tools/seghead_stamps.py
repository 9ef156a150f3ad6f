#!/usr/bin/env python3
"""Where a launch of the fused frozen segmentation head spends its time: runs the PN_SEGHEAD_DBG=16 variant (the product kernel +
s_memtime stamps of wave 0 at its phase boundaries, left in the tile's loss-partial rows) inside a training step with the head frozen
and prints the median over the tiles of every phase.  Stamps: 0 entry, 1 input tile converted and stored, 2 barrier passed, 3 first
fragments requested (loop entry), 4 chunk 1 starts, 5 its seg_l1 MFMAs issued + next fragments requested, 6 its epilogue stored, 7 barrier,
8 its seg_l2 MFMAs issued, 9 barrier, 10 loop done, 11 seg_l2 epilogue stored, 12 barrier, 13 seg_l3 done and stored, 14 barrier,
15 seg_l4 done and stored, 16 barrier, 17 output kernel image built, 18 barrier, 19 logits, 20 barrier, 21 softmax / loss tail done;
22 prologue requests issued, 23 input tile converted and stored (both between 0 and 1).  PN_SEGHEAD_MB=2 / 4 picks the tile height."""
import json
import os
import sys

import numpy as np
import torch

os.environ["PN_SEGHEAD_DBG"] = "16"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd.pointnet.PointNet import PointNet   # noqa: E402

dev = torch.device("cuda:0")
for B, N in ((32, 1024), (32, 4096)):
    g = torch.Generator().manual_seed(1)
    m = PointNet(23, 12, 0.3, 42, precision="bf16", device=dev)
    m.freeze_segmentation_head()
    pc = (torch.rand(B, N, 3, generator=g) * 10).to(dev)
    y_cls = torch.randint(0, 23, (B,), generator=g, dtype=torch.int32).to(dev)
    y_seg = torch.randint(0, 12, (B, N), generator=g, dtype=torch.int32).to(dev)
    se3 = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
    for _ in range(3):
        outs = m.fused_loss_step(pc, y_cls, y_seg, se3, (1.0, 0.0, 0.0))
    torch.cuda.synchronize()
    rows_per_tile = 64 if os.environ.get("PN_SEGHEAD_MB", "2") == "2" else 128
    raw = outs[1].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    rows = raw.reshape(B * (N // rows_per_tile), rows_per_tile * 12)[:, :24]
    pro = np.stack([(rows[:, 22] - rows[:, 0]) & 0xffffffff, (rows[:, 23] - rows[:, 22]) & 0xffffffff, (rows[:, 1] - rows[:, 23]) & 0xffffffff], 1)
    rows = rows[:, :22]
    d = (rows[:, 1:] - rows[:, :-1]) & 0xffffffff
    span = (rows[:, 21] - rows[:, 0]) & 0xffffffff
    t0 = rows[:, 0].min()
    print(json.dumps({"B": B, "N": N, "rows_per_tile": rows_per_tile, "tiles": int(rows.shape[0]), "prologue_issue_convert_stage": [int(v) for v in np.median(pro, axis=0)], "phase_median": [int(v) for v in np.median(d, axis=0)], "span_median": int(np.median(span)),
                      "span_max": int(span.max()), "first_entry_to_last_exit": int(((rows[:, 21] - t0) & 0xffffffff).max()),
                      "entry_spread": int(((rows[:, 0] - t0) & 0xffffffff).max())}), flush=True)
