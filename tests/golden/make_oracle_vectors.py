#!/usr/bin/env python3
"""Writes tests/golden/oracle_b2_n64.npz: inputs and fp64 outputs of oracle/pointnet_oracle.py on a tiny seeded case
(B=2, N=64).  The reference cannot run here (TensorFlow absent), so these vectors pin the ORACLE against
drift between rounds -- they are not reference outputs ("parity unpinned").  Regenerate only on purpose:
    python tests/golden/make_oracle_vectors.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pointnet_oracle as O   # noqa: E402


def build():
    B, N, CCLS, CSEG = 2, 64, 23, 12
    p = O.init_params(CCLS, CSEG, seed=7, vanilla=False, dtype=torch.float64, randomize_bn=True)
    g = torch.Generator().manual_seed(20260003)
    pc = (torch.rand(B, N, 3, generator=g, dtype=torch.float64) * 2 - 1) * 20 + 5
    y_cls = torch.randint(0, CCLS, (B,), generator=g)
    y_seg = torch.randint(0, CSEG, (B, N), generator=g)
    se3 = torch.linalg.qr(torch.randn(B, 3, 3, generator=g, dtype=torch.float64))[0].contiguous()
    out = {"pc": pc.numpy(), "y_cls": y_cls.numpy(), "y_seg": y_seg.numpy(), "se3": se3.numpy()}
    with torch.no_grad():
        cls, seg, R = O.forward(p, pc, training=False)
    out.update(inf_cls=cls.numpy(), inf_seg=seg.numpy(), inf_R=R.numpy())
    # training-mode forward + the three keras losses + gradients of every trainable tensor
    q = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not (k.endswith("moving_mean") or k.endswith("moving_var"))) for k, v in p.items()}
    cls, seg, R = O.forward(q, pc, training=True)
    tot, parts = O.total_loss((cls, seg, R), {"classification_output": y_cls, "segmentation_output": y_seg, "se3": se3},
                              {"classification": 1.0, "segmentation": 1.0, "rotation": 1.0})
    tot.backward()
    out.update(train_cls=cls.detach().numpy(), train_seg=seg.detach().numpy(), train_R=R.detach().numpy(), loss=np.array(float(tot.detach())),
               loss_cls=np.array(float(parts["classification_output_loss"])), loss_seg=np.array(float(parts["segmentation_output_loss"])),
               loss_se3=np.array(float(parts["se3_loss"])))
    for k, v in q.items():
        if v.requires_grad and v.grad is not None:
            gq = v.grad.reshape(-1)          # a digest per tensor keeps the fixture small: sum, L2 norm, 6 fixed probes
            idx = torch.linspace(0, gq.numel() - 1, 6).long()
            out["grad/" + k] = torch.cat([gq.sum().reshape(1), gq.norm().reshape(1), gq[idx]]).numpy()
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_b2_n64.npz")
    np.savez_compressed(path, **build())
    print("wrote", path, os.path.getsize(path), "bytes")
