import sys; sys.path.insert(0, '/root/repo')
import torch
from pointcloudprocessing_amd import ops, _lib
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
for B, N in ((16, 136), (4, 256), (2, 1000)):
    K, C = 128, 1024
    x = torch.randn(B * N, K, generator=g); w = torch.randn(K, C, generator=g) * 0.1
    gamma = torch.randn(C, generator=g)
    sgn = ops.sign(gamma.to(dev))
    z = (x.double() @ w.double()).view(B, N, C)
    s = torch.where(gamma >= 0, 1.0, -1.0).double()
    tv = (z * s).max(1).values
    xd, wd = x.to(dev), w.to(dev)
    op = _lib.operand(xd)
    for rows in (64, 128):
        for prec in (1, 3):
            pmax, pidx, part = ops.conv_fwd_max_panel(op, wd, B, N, K, C, sgn, prec, panel_rows=rows)
            T = pmax.shape[0] // B
            mx = pmax.view(B, T, C).max(1).values.double().cpu()
            s1 = part[:, 0].sum(0).double().cpu(); s2 = part[:, 1].sum(0).double().cpu()
            e_max = float((mx - tv).abs().max() / tv.abs().max())
            e_s1 = float((s1 - z.reshape(-1, C).sum(0)).abs().max() / z.reshape(-1, C).sum(0).abs().max())
            e_s2 = float((s2 - (z.reshape(-1, C) ** 2).sum(0)).abs().max() / (z.reshape(-1, C) ** 2).sum(0).abs().max())
            print(f"B={B} N={N} rows={rows} prec={prec}: max rel err {e_max:.2e}  sum {e_s1:.2e}  sumsq {e_s2:.2e}")
