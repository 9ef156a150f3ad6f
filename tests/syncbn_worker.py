"""one rank of the synchronised-BatchNormalization test (tests/test_gpu_syncbn.py): gloo process group, the rank's share of the batch on
cuda:0, one fused training step through PointNet(sync_bn_world=W), gradients summed as engine.TrainStep does; results to <out>/rank<r>.pt"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class MergedRun:
    """the W ranks' workspaces as the workspace of ONE step on the whole batch: what tests/teacher_forced.py reads from a model.
    A named range that has the whole-batch size on every rank (rows of all ranks: gathered / computed redundantly, or a per-channel
    vector formed from the all-reduced sums) is assembled from each rank's OWN share of it; one that has 1/W of it is the
    concatenation of the ranks' ranges.  Gradients: the sum engine.TrainStep forms; loss sums: added up."""

    def __init__(self, m, plain, raws, grads, Bg, B, N, W):
        self.m, self.plain, self.raws, self.grads, self.Bg, self.B, self.N, self.W = m, plain, raws, grads, Bg, B, N, W
        self.activation_dtype = m.activation_dtype
        self.scalars = sum(r["scalars"].double() for r in raws).float()

    @staticmethod
    def _lookup(model, name, B, N):
        import ctypes as C
        from pointcloudprocessing_amd._lib import lib
        off, nb = C.c_int64(), C.c_int64()
        rc = lib().pn_model_ws_lookup(C.byref(model._desc), B, N, 1, name.encode(), C.byref(off), C.byref(nb))
        return (off.value, nb.value) if rc == 0 else None

    def workspace_tensor(self, name, B_, N_, training, dtype=torch.float32):
        assert B_ == self.Bg and N_ == self.N and training
        single = self._lookup(self.plain, name, self.Bg, self.N)
        assert single is not None, name
        loc = self._lookup(self.m, name + "_all", self.B, self.N) or self._lookup(self.m, name, self.B, self.N)
        assert loc is not None, name
        off, nb = loc
        if nb == single[1]:
            assert nb % self.W == 0, (name, nb)
            sh = nb // self.W
            parts = [self.raws[r]["ws"][off + r * sh: off + (r + 1) * sh] for r in range(self.W)]
        else:
            assert nb * self.W == single[1], (name, nb, single[1])
            parts = [self.raws[r]["ws"][off: off + nb] for r in range(self.W)]
        return torch.cat(parts).view(dtype)

    def named_grads(self):
        w = self.m._weights
        return {n: w.view(n, self.grads) for n in w.slots}

    def named_weights(self):
        return {k: v.cpu() for k, v in self.m.named_weights().items()}


def engine_steps(m, rank, world, out, dev, pc, y_cls, y_seg, se3, lw, steps):
    """engine.TrainStep on this rank's clouds (the trainer's step: masks drawn on the device from one seed for all ranks' rows, gradients
    summed, Adam): loss sums per step and the final parameters to <out>/engine<r>.pt"""
    from pointcloudprocessing_amd.engine import TrainStep
    from pointcloudprocessing_amd.optim import KerasAdam
    B, N = pc.shape[0], pc.shape[1]
    opt = KerasAdam(m.params_flat.data, 1e-3, 7000, 0.7)
    ts = TrainStep(m, opt, B, N, lw)
    assert ts.sync_bn and ts.mode == "eager"
    ts._mask_seed = 1234                     # the parent's whole-batch run draws the same masks
    pc, y_cls, y_seg, se3 = pc.contiguous().to(dev), y_cls.to(torch.int32).to(dev), y_seg.contiguous().to(torch.int32).to(dev), se3.contiguous().to(dev)
    sums = []
    for _ in range(steps):
        ts(pc, y_cls, y_seg, se3)
        sums.append(m.scalars.cpu().double()[:2].clone())
    torch.cuda.synchronize()
    torch.save({"sums": torch.stack(sums), "params": m.params_flat.data.cpu(), "iterations": int(opt.iterations)}, os.path.join(out, f"engine{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def main():
    rank, world, port, out, precision, profile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pointnet_oracle as O       # weights / inputs only
    import parity_harness as H
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    dev = torch.device("cuda:0")
    Bg, N = 16, 200
    B = Bg // world
    spec, lw = H.PROFILES[profile]
    params = O.init_params(H.CCLS, H.CSEG, seed=17, randomize_bn=True)
    pc, y_cls, y_seg, se3, keep = H.make_inputs(Bg, N, 33, "shapes")
    sl = slice(rank * B, (rank + 1) * B)
    m = PointNet(H.CCLS, H.CSEG, 0.3, 42, precision=precision, device=dev, sync_bn_world=world, sync_bn_rank=rank,
                 regularize_input_transform=True, regularize_feature_transform=True)
    m.set_weights(params)
    H.apply_profile(m, spec)
    if len(sys.argv) > 7 and sys.argv[7] == "engine":
        return engine_steps(m, rank, world, out, dev, pc[sl], y_cls[sl], y_seg[sl], se3[sl], lw, int(sys.argv[8]))
    kp = (keep["dropout_1"].to(torch.uint8).to(dev), keep["dropout_2"].to(torch.uint8).to(dev))        # the rows of ALL ranks
    m._workspace(B, N, True)
    for wn in ("iT.m3", "fT.m3", "mm23"):          # rows of the maxima: resolved by the backward pass where it runs (mark them unwritten)
        m.workspace_tensor(wn + ".arg", B, N, True, torch.int32).fill_(-1)
    outs = m.fused_loss_step(pc[sl].contiguous().to(dev), y_cls[sl].to(torch.int32).to(dev), y_seg[sl].contiguous().to(torch.int32).to(dev),
                             se3[sl].contiguous().to(dev), lw, keep=kp)
    torch.cuda.synchronize()
    # per-layer intermediates of this rank's clouds (diagnostics: the parent compares them with the whole-batch run's rows)
    dump = {}
    names = []
    for wn, C_ in (("iT.c1", 64), ("iT.c2", 128), ("m11", 64), ("m12", 64), ("fT.c1", 64), ("fT.c2", 128), ("m21", 64), ("m22", 128), ("s1", 512), ("s2", 256),
                   ("s3", 128), ("s4", 128)):
        for suffix in ("Z", "dy"):
            dump[f"{wn}.{suffix}"] = m.workspace_tensor(f"{wn}.{suffix}", B, N, True, m.activation_dtype).float().cpu().view(-1, C_)
        for suffix in ("scale", "shift", "ca", "cb", "cc", "mean", "invstd"):
            dump[f"{wn}.{suffix}"] = m.workspace_tensor(f"{wn}.{suffix}", B, N, True).cpu()
    for wn in ("iT.m3", "fT.m3", "mm23"):
        for suffix, rows in (("g_all", Bg), ("zstar_all", Bg), ("hs", Bg), ("dG", Bg)):
            dump[f"{wn}.{suffix}"] = m.workspace_tensor(f"{wn}.{suffix}", B, N, True).cpu().view(rows, 1024)
        for suffix in ("e", "f", "q"):
            dump[f"{wn}.{suffix}"] = m.workspace_tensor(f"{wn}.{suffix}", B, N, True).cpu()
        dump[f"{wn}.arg"] = m.workspace_tensor(f"{wn}.arg", B, N, True, torch.int32).cpu().view(B, 1024)
        dump[f"{wn}.D"] = m.workspace_tensor(f"{wn}.D", B, N, True, m.activation_dtype).float().cpu().view(-1, 128)
    for wn, per in (("iT.R", 9), ("fT.R", 4096), ("iT.dR", 9), ("fT.dR", 4096), ("dGcls", 1024), ("dGseg", 1024), ("cls_dlogits", H.CCLS)):
        dump[wn] = m.workspace_tensor(wn, B, N, True).cpu().view(Bg, per)
    dump["X64"] = m.workspace_tensor("X64", B, N, True, m.activation_dtype).float().cpu().view(-1, 64)
    dump["dX64"] = m.workspace_tensor("dX64", B, N, True, m.activation_dtype).float().cpu().view(-1, 64)
    from pointcloudprocessing_amd import _lib, ops
    for wn, src in (("mm23", "m22"), ("iT.m3", "iT.c2"), ("fT.m3", "fT.c2")):      # as tests/parity_harness.py: a layer whose backward did not run
        a_bw = m.workspace_tensor(wn + ".arg", B, N, True, torch.int32).view(B, 1024)
        if bool((a_bw == -1).all()):
            op = _lib.operand(m.workspace_tensor(src + ".Z", B, N, True, m.activation_dtype).view(B * N, 128), ca=m.workspace_tensor(src + ".scale", B, N, True),
                              cc=m.workspace_tensor(src + ".shift", B, N, True), relu=True)
            wf = (m.workspace_tensor(wn + ".wb_hi", B, N, True, torch.bfloat16), m.workspace_tensor(wn + ".wb_lo", B, N, True, torch.bfloat16))
            a_bw.copy_(ops.max_resolve(op, wf, m.workspace_tensor(wn + ".argq", B, N, True, torch.int32).view(B, 1024), B, N, 128, 1024, _lib.PREC[precision]))
    torch.cuda.synchronize()
    g = m.grads_flat
    if rank != 0:
        g.mul_(m.replicated_grad_mask())
    dist.all_reduce(g)
    torch.cuda.synchronize()
    # every layer of the step against its fp64 recomputation from the stored inputs (tests/teacher_forced.py), on the MERGED run: rank 0
    # reads all ranks' workspaces and presents them to the checker as the workspace of one whole-batch step
    torch.save({"ws": m._workspace(B, N, True).cpu(), "outs": [o.cpu() for o in outs], "scalars": m.scalars.cpu()}, os.path.join(out, f"ws{rank}.pt"))
    dist.barrier()
    fails = []
    if rank == 0:
        import teacher_forced
        raws = [torch.load(os.path.join(out, f"ws{r}.pt"), weights_only=True) for r in range(world)]
        plain = PointNet(H.CCLS, H.CSEG, 0.3, 42, precision=precision, device=dev, regularize_input_transform=True, regularize_feature_transform=True)
        merged = MergedRun(m, plain, raws, g.cpu(), Bg, B, N, world)
        outs_m = [torch.cat([raws[r]["outs"][i] for r in range(world)]) for i in range(3)]
        fails = teacher_forced.check_layers(merged, outs_m, params, pc, y_cls, y_seg, se3, keep, H.oracle_trainable(spec), lw, precision, False,
                                            f"sync-BN [{profile}] merged ranks", H.report, reg=True)
        fails = [(str(a), float(b), float(c)) for a, b, c in fails]
    torch.save({"grads": g.cpu(), "forced_fails": fails, "weights": {k: v.cpu().clone() for k, v in m.named_weights().items()}, "outs": [o.cpu() for o in outs],
                "scalars": m.scalars.cpu(), "dump": dump}, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
