"""Teacher-forced, layer-by-layer parity of one PointNet training step on the HIP path.

TEST INFRASTRUCTURE.  PARITY UNPINNED against the reference itself (TensorFlow is absent); every formula below restates the layer the
reference builds from Keras ops: ConvLayer.call pointnet/PointNet.py:554-566 (1x1 Conv2D -> BatchNormalization -> ReLU), DenseLayer.call
:642-654, TNet.call :418-454, PointNet.call :197-292, loss assembly pointnet_train.py:334-345, with the Keras semantics listed in
oracle/pointnet_oracle.py's header.

Why this exists.  End to end, PointNet in training mode amplifies the rounding of a 16-bit arithmetic mode (batch-statistics
BatchNormalization over the B rows of the T-Net and classification dense layers, then ~20 more layers): at B = 32 two correct bf16
implementations differ by 0.1-0.2 in probability, so an end-to-end tolerance has to be that wide and a formula error of a few percent
hides below it.  Here every layer is checked ON ITS OWN: its input is what the GPU itself stored for the layer below (workspace
entries *.Z / *.scale / *.shift, X64, *.g, *.a; on the way back *.dy, *.dz, *.din, dG), the layer is recomputed in fp64 with the
mode's operand roundings, and the GPU's output of THAT layer (values, BatchNormalization statistics and coefficients, arg-max rows,
activation gradients, parameter gradients) must agree within a small fixed tolerance: a few units in the last place of the storage
type, relative to the tensor's largest element.  Nothing cascades, so the tolerances do not depend on how well conditioned the
network is, and a wrong index, coefficient, mask or summation range in any one kernel fails its own line.

Tolerances:
  tensors stored as bf16 ('bf16' mode: pre-BN outputs, X_64, activation gradients): ELEMENT BY ELEMENT, |gpu - ref| <= half a bf16 ulp
                             of that element + 2^-10 (forward) / 2^-9 (backward) of the tensor maximum (an MFMA operand whose fp32
                             affine lands on the other side of a bf16 rounding boundary than the fp64 one moves a product by 2^-8);
  fp32-stored tensors (max |gpu - ref| <= tol * max |ref|): 1e-3 forward / 2e-3 backward with bf16 operands, 1e-4 / 2e-4 with split
                             operands ('bf16x3');
  per-cloud dense layers, statistics, coefficients, pooled features: fp32-grade (1e-5 .. 1e-3, stated at each check);
  the Gram-form backward of the three max-pooled layers is algebraically, not operation by operation, the canonical backward it is
  compared with: 1.5 x 2^-7 of the tensor maximum in 'bf16' (measured <= 1.05 x 2^-7), 2e-3 otherwise.
The measured ratio err / limit of every line goes to gpurun_out/model_report.txt.
"""
import math

import torch

from oracle import pointnet_oracle as O   # checker only

EPS = O.BN_EPS
MOM = O.BN_MOMENTUM


def _amax(t):
    return float(t.abs().max()) if t.numel() else 0.0


class Forced:
    def __init__(self, m, params, pc, y_cls, y_seg, se3, keep, trainable, lw, precision, vanilla, tag, report, seg_src=None, reg=False):
        """m: the model after fused_loss_step on these inputs; params: the weights BEFORE the step (name -> fp32 tensor);
        trainable: block -> bool (missing = True); keep: {'dropout_1','dropout_2'} keep masks (bool, CPU);
        seg_src: a model that kept the segmentation head's activations when `m` ran the head fused (or None: head checked on m)."""
        self.m, self.P = m, {k: v.double() for k, v in params.items()}
        self.B, self.N = pc.shape[0], pc.shape[1]
        self.M = self.B * self.N
        self.pc, self.y_cls, self.y_seg, self.se3 = pc, y_cls, y_seg, se3
        self.keep, self.tr, self.lw, self.vanilla, self.tag, self.report = keep, trainable, lw, vanilla, tag, report
        self.seg_src = seg_src
        self.reg = reg
        self.precision = precision
        self.s16 = precision == "bf16"
        self.act = torch.bfloat16 if self.s16 else torch.float32
        self.qa = O.bf16x3_round if precision == "bf16x3" else O.bf16_round
        # forward: stored per-point tensors; backward: stored activation gradients; fp32-grade small tensors
        self.t_z = 2.0 ** -8 if self.s16 else (1e-4 if precision == "bf16x3" else 1e-3)
        self.t_dy = 2.0 ** -7 if self.s16 else (2e-4 if precision == "bf16x3" else 2e-3)
        self.t_wg = 1e-4 if precision == "bf16x3" else 2e-3       # weight gradients: fp32 accumulation of products of rounded operands
        self.t_gram = 1.5 * 2.0 ** -7 if self.s16 else 2e-3      # (measured: at most 1.05 x 2^-7 over the BASELINE configurations)
        self.t_pool = 1e-4 if precision == "bf16x3" else 1e-3     # pooled features, fp32, from fp32 accumulators
        self.fails = []
        self.G = m.named_grads()
        self.NW = m.named_weights()

    # ---------------------------------------------------------------- plumbing
    def ws(self, name, dtype=torch.float32, model=None):
        mm = model or self.m
        return mm.workspace_tensor(name, self.B, self.N, True, dtype).cpu().double()

    def ws_act(self, name, C_, model=None):
        return self.ws(name, self.act, model).view(-1, C_)

    def judge(self, what, got, ref, tol, floor=0.0):
        got, ref = got.detach().cpu().double().reshape(-1), ref.detach().cpu().double().reshape(-1)
        assert got.numel() == ref.numel(), (what, got.numel(), ref.numel())
        skip = torch.isnan(ref)                                    # elements whose ReLU decision is within rounding (lazy())
        if bool(skip.any()):
            if float(skip.double().mean()) >= 1e-3:
                self.fails.append((what + ": too many undecided ReLU elements", float(skip.double().mean()), 1e-3))
            got, ref = got.masked_fill(skip, 0.0), ref.masked_fill(skip, 0.0)
        scale = max(_amax(ref), floor)
        err = _amax(got - ref)
        lim = tol * scale + 1e-30
        ok = err <= lim and bool(torch.isfinite(got).all())
        self.report(f"{self.tag} TF {what:52s} err {err:.3e}  max|ref| {scale:.3e}  limit {lim:.3e}  err/limit {err / lim:.2f}{'' if ok else '   <-- FAIL'}")
        if not ok:
            self.fails.append((what, err, lim))

    def judge_stored(self, what, got, ref, slack):
        """a tensor the GPU stored as bf16: element by element  |gpu - ref| <= half a bf16 ulp OF THAT ELEMENT + slack * max |ref|
        (slack: an MFMA operand that rounded the other way moves a product by 2^-8; fp32 accumulation order)"""
        got, ref = got.detach().cpu().double().reshape(-1), ref.detach().cpu().double().reshape(-1)
        assert got.numel() == ref.numel(), (what, got.numel(), ref.numel())
        skip = torch.isnan(ref)
        if bool(skip.any()):
            if float(skip.double().mean()) >= 1e-3:
                self.fails.append((what + ": too many undecided ReLU elements", float(skip.double().mean()), 1e-3))
            got, ref = got.masked_fill(skip, 0.0), ref.masked_fill(skip, 0.0)
        scale = _amax(ref)
        half_ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-38))) - 8.0)
        lim = half_ulp * 1.0001 + slack * scale + 1e-30
        ratio = float(((got - ref).abs() / lim).max())
        ok = ratio <= 1.0 and bool(torch.isfinite(got).all())
        self.report(f"{self.tag} TF {what:52s} worst |err| / (half bf16 ulp of the element + {slack:.1e} max|ref|) = {ratio:.2f}  max|ref| {scale:.3e}"
                    f"{'' if ok else '   <-- FAIL'}")
        if not ok:
            self.fails.append((what, ratio, 1.0))

    def stored(self, what, got, ref, bwd=False):
        if self.s16:
            self.judge_stored(what, got, ref, 2.0 ** -9 if bwd else 2.0 ** -10)
        else:
            self.judge(what, got, ref, self.t_dy if bwd else self.t_z)

    def bn_batch(self, block):
        return self.tr.get(block, True)

    def lazy(self, wn, C_, model=None):
        """relu(bn(z)) as the consumer of a stored layer forms it: fp32 affine of the stored z, lower clamp, operand rounding"""
        z = self.ws_act(wn + ".Z", C_, model)
        sc, sh = self.ws(wn + ".scale", model=model), self.ws(wn + ".shift", model=model)
        y = z * sc + sh
        # the ReLU decision of an element whose fp32 affine is within rounding of zero is the GPU's to make: such elements (a handful in
        # millions) are left out of the comparisons that apply this mask (second value: NaN there, which judge() skips)
        sure = y.abs() > 1e-6 * ((z * sc).abs() + sh.abs())       # (fp32 rounding of the affine is 6e-8 of its terms)
        return self.qa(torch.relu(y)), torch.where(sure, (y > 0).double(), torch.full_like(y, float("nan")))

    # ---------------------------------------------------------------- forward pieces
    def fwd_stats(self, pref, wn, block, z_ref, count, model=None, tol=1e-4):
        """BatchNormalization of a layer whose exact pre-BN output is z_ref (rows x C) or given as (sum, sumsq): mean / invstd / scale /
        shift and the moving statistics against the Keras formulas"""
        P = self.P
        g, b, mm, mv = (P[f"{pref}.bn.{k}"] for k in ("gamma", "beta", "moving_mean", "moving_var"))
        mean_g, inv_g = self.ws(wn + ".mean", model=model), self.ws(wn + ".invstd", model=model)
        if self.bn_batch(block):
            if isinstance(z_ref, tuple):
                s1, s2 = z_ref
            else:
                s1, s2 = z_ref.sum(0), (z_ref * z_ref).sum(0)
            mean = s1 / count
            ez2 = s2 / count
            var = (ez2 - mean * mean).clamp_min(0)
            rms = float(ez2.sqrt().max())
            self.judge(f"{pref} batch mean", mean_g, mean, tol, floor=rms)
            var_g = 1.0 / (inv_g * inv_g) - EPS
            self.judge(f"{pref} batch variance (from invstd)", var_g, var, tol, floor=float(ez2.max()))
            if model is None:
                self.judge(f"{pref} moving_mean update", self.NW[f"{pref}.bn.moving_mean"], mm * MOM + mean * (1 - MOM), 1e-5, floor=1e-3)
                self.judge(f"{pref} moving_var update", self.NW[f"{pref}.bn.moving_var"], mv * MOM + var * (1 - MOM), 1e-5, floor=1e-3)
        else:
            self.judge(f"{pref} frozen mean", mean_g, mm, 1e-6, floor=1e-3)
            self.judge(f"{pref} frozen invstd", inv_g, torch.rsqrt(mv + EPS), 1e-5)
            if model is None:
                assert torch.equal(self.NW[f"{pref}.bn.moving_mean"].cpu().double(), mm) and torch.equal(self.NW[f"{pref}.bn.moving_var"].cpu().double(), mv), \
                    f"{pref}: moving statistics of a frozen layer changed"
        sc = g * inv_g
        self.judge(f"{pref} scale = gamma * invstd", self.ws(wn + ".scale", model=model), sc, 1e-5)
        self.judge(f"{pref} shift = beta - mean * scale", self.ws(wn + ".shift", model=model), b - mean_g * sc, 1e-5, floor=float(sc.abs().max()) * 1e-2)

    def fwd_conv(self, pref, wn, block, a, W, cloud_bias=None, model=None, quant_w=True):
        """a stored ConvLayer: z = a . W (+ per-cloud bias) -> stored Z, BatchNormalization statistics / coefficients"""
        C_ = W.shape[-1]
        Wq = self.qa(W) if quant_w else W
        if W.dim() == 3:        # per-cloud kernel
            z = torch.einsum("bnk,bkc->bnc", a.view(self.B, self.N, -1), Wq).reshape(self.M, C_)
        else:
            z = a @ Wq
        if cloud_bias is not None:
            z = (z.view(self.B, self.N, C_) + cloud_bias.view(self.B, 1, C_)).reshape(self.M, C_)
        if quant_w:
            self.stored(f"{pref} pre-BN output Z", self.ws_act(wn + ".Z", C_, model), z)
        elif self.s16:
            self.judge_stored(f"{pref} pre-BN output Z", self.ws_act(wn + ".Z", C_, model), z, 1e-6)
        else:
            self.judge(f"{pref} pre-BN output Z", self.ws_act(wn + ".Z", C_, model), z, 1e-5)
        if f"{pref}.bn.gamma" in self.P:
            self.fwd_stats(pref, wn, block, z, self.M, model)
        return z

    def fwd_max(self, pref, cwn, mwn, block, a):
        """ConvLayer(128 -> 1024) + BatchNormalization + ReLU + reduce_max over the points (PointNet.py:242-248, 425-429): nothing of
        (B, N, 1024) is stored; statistics, pooled maxima, the pre-BN value at the maximum and the arg-max row are checked"""
        P, B, N = self.P, self.B, self.N
        Wq = self.qa(P[f"{pref}.kernel"])
        s1 = torch.zeros(1024, dtype=torch.float64); s2 = torch.zeros(1024, dtype=torch.float64)
        zs = []
        for b in range(B):
            z = a[b * N:(b + 1) * N] @ Wq
            s1 += z.sum(0); s2 += (z * z).sum(0)
            zs.append(z)
        self.fwd_stats(pref, cwn, block, (s1, s2), self.M)
        sc, sh = self.ws(cwn + ".scale"), self.ws(cwn + ".shift")
        g_gpu, zst_gpu = self.ws(mwn + ".g").view(B, 1024), self.ws(mwn + ".zstar").view(B, 1024)
        arg = self.ws(mwn + ".arg", torch.int32).long().view(B, 1024)
        g_ref = torch.empty(B, 1024, dtype=torch.float64); gap = 0.0; ymax = 0.0; zat = torch.empty(B, 1024, dtype=torch.float64)
        for b in range(B):
            y = zs[b] * sc + sh
            top = y.max(0).values
            g_ref[b] = torch.relu(top)
            at = y.gather(0, arg[b].clamp(0, N - 1).unsqueeze(0)).squeeze(0)
            gap = max(gap, float((top - at).max()))
            ymax = max(ymax, _amax(y))
            zat[b] = zs[b].gather(0, arg[b].clamp(0, N - 1).unsqueeze(0)).squeeze(0)
        self.judge(f"{pref} pooled feature g = max relu(bn(z))", g_gpu, g_ref, self.t_pool)
        assert int(arg.min()) >= 0 and int(arg.max()) < N, (pref, int(arg.min()), int(arg.max()))
        lim = self.t_pool * ymax
        ok = gap <= lim
        self.report(f"{self.tag} TF {pref + ' arg-max row reaches the maximum':52s} worst gap {gap:.3e} limit {lim:.3e}{'' if ok else '   <-- FAIL'}")
        if not ok:
            self.fails.append((pref + " argmax gap", gap, lim))
        self.judge(f"{pref} pre-BN value at the arg-max row (zstar)", zst_gpu, zat, self.t_pool)
        return zs

    def fwd_dense(self, pref, wn, block, x, keep=None, act=True):
        P = self.P
        z = x @ P[f"{pref}.kernel"]
        if f"{pref}.bn.gamma" not in P:
            return z + P[f"{pref}.bias"]
        self.judge(f"{pref} dense pre-BN z", self.ws(wn + ".z"), z, 1e-4)
        zg = self.ws(wn + ".z").view(self.B, -1)                   # teacher: the statistics of the GPU's own z
        g, b, mm, mv = (P[f"{pref}.bn.{k}"] for k in ("gamma", "beta", "moving_mean", "moving_var"))
        mean_g, inv_g = self.ws(wn + ".mean"), self.ws(wn + ".invstd")
        if self.bn_batch(block):
            mean = zg.mean(0); var = ((zg - mean) ** 2).mean(0)
            self.judge(f"{pref} dense batch mean", mean_g, mean, 1e-5, floor=float(zg.abs().max()))
            self.judge(f"{pref} dense batch invstd", inv_g, torch.rsqrt(var + EPS), 1e-4)
            self.judge(f"{pref} moving_mean update", self.NW[f"{pref}.bn.moving_mean"], mm * MOM + mean * (1 - MOM), 1e-5, floor=1e-3)
            self.judge(f"{pref} moving_var update", self.NW[f"{pref}.bn.moving_var"], mv * MOM + var * (1 - MOM), 1e-5, floor=1e-3)
        else:
            self.judge(f"{pref} dense frozen mean", mean_g, mm, 1e-6, floor=1e-3)
            self.judge(f"{pref} dense frozen invstd", inv_g, torch.rsqrt(mv + EPS), 1e-5)
        y = (zg - mean_g) * inv_g * g + b
        a = torch.relu(y) if act else y
        if keep is not None:
            a = a * keep.double() / (1.0 - 0.3)
        self.judge(f"{pref} dense output (BN, ReLU, dropout)", self.ws(wn + ".a"), a, 1e-4, floor=1e-2)
        return a

    # ---------------------------------------------------------------- backward pieces
    def bn_bwd_coeffs(self, pref, wn, block, model=None):
        """BatchNormalization backward as the lazy operand dz = ca * dy + cb * z + cc (and dgamma, dbeta), from the GPU's own stored dy, z"""
        P = self.P
        C_ = P[f"{pref}.bn.gamma"].numel()
        g = P[f"{pref}.bn.gamma"]
        mean, inv = self.ws(wn + ".mean"), self.ws(wn + ".invstd")
        dy = self.ws_act(wn + ".dy", C_)
        z = self.ws_act(wn + ".Z", C_)
        ca, cb, cc = self.ws(wn + ".ca"), self.ws(wn + ".cb"), self.ws(wn + ".cc")
        s = g * inv
        self.judge(f"{pref} bwd ca = gamma * invstd", ca, s, 1e-5)
        if self.bn_batch(block):
            zh = (z - mean) * inv
            m1 = dy.mean(0); m2 = (dy * zh).mean(0)
            # the GPU's sums are taken from the fp32 values before dy is stored: a stored bf16 dy differs by <= 2^-9 |dy| per element
            sl = 2.0 ** -9 if self.s16 else 0.0
            a1 = dy.abs().mean(0); a2 = (dy * zh).abs().mean(0)
            cb_ref = -s * inv * m2
            cc_ref = -s * m1 + s * inv * mean * m2
            lim_cb = (s.abs() * inv * (sl * a2 + 1e-4 * a2)).max()
            e_cb = _amax(cb - cb_ref)
            self._lim(f"{pref} bwd cb = -gamma invstd^2 mean(dy zhat)", e_cb, float(lim_cb) + 1e-4 * _amax(cb_ref))
            lim_cc = (s.abs() * ((sl + 1e-4) * a1 + inv * mean.abs() * (sl + 1e-4) * a2)).max()
            self._lim(f"{pref} bwd cc", _amax(cc - cc_ref), float(lim_cc) + 1e-4 * _amax(cc_ref))
            if self.tr.get(block, True) and model is None:
                M = dy.shape[0]
                self._lim(f"grad {pref}.bn.gamma = sum dy zhat", _amax(self.G[f"{pref}.bn.gamma"].cpu().double() - m2 * M),
                          float(((sl + 1e-4) * a2 * M).max()) + 1e-4 * _amax(m2 * M))
                self._lim(f"grad {pref}.bn.beta = sum dy", _amax(self.G[f"{pref}.bn.beta"].cpu().double() - m1 * M),
                          float(((sl + 1e-4) * a1 * M).max()) + 1e-4 * _amax(m1 * M))
        else:
            self.judge(f"{pref} bwd cb (moving statistics: 0)", cb, torch.zeros_like(cb), 1.0, floor=1e-12)
            self.judge(f"{pref} bwd cc (moving statistics: 0)", cc, torch.zeros_like(cc), 1.0, floor=1e-12)
        dz = ca * dy + cb * z + cc
        return self.qa(dz), dz

    def _lim(self, what, err, lim):
        ok = err <= lim
        self.report(f"{self.tag} TF {what:52s} err {err:.3e}  limit {lim:.3e}  err/limit {err / (lim + 1e-300):.2f}{'' if ok else '   <-- FAIL'}")
        if not ok:
            self.fails.append((what, err, lim))

    def dense_bwd(self, pref, wn, block, da, x, keep=None):
        """dropout -> ReLU -> BatchNormalization backward of a dense layer over the B rows, from the GPU's own da and z; returns dz (GPU)"""
        P, B = self.P, self.B
        g, b = P[f"{pref}.bn.gamma"], P[f"{pref}.bn.beta"]
        z = self.ws(wn + ".z").view(B, -1)
        mean, inv = self.ws(wn + ".mean"), self.ws(wn + ".invstd")
        zh = (z - mean) * inv
        y = zh * g + b
        d = da.view(B, -1)
        if keep is not None:
            d = d * keep.double() / (1.0 - 0.3)
        dy = d * (y > 0)
        if self.bn_batch(block):
            dz = g * inv * (dy - dy.mean(0) - zh * (dy * zh).mean(0))
        else:
            dz = g * inv * dy
        dz_g = self.ws(wn + ".dz").view(B, -1)
        self.judge(f"{pref} dense dz (dropout, ReLU, BN backward)", dz_g, dz, 1e-3)
        if self.tr.get(block, True):
            if self.bn_batch(block):
                self.judge(f"grad {pref}.bn.gamma", self.G[f"{pref}.bn.gamma"], (dy * zh).sum(0), 1e-3)
                self.judge(f"grad {pref}.bn.beta", self.G[f"{pref}.bn.beta"], dy.sum(0), 1e-3)
            self.judge(f"grad {pref}.kernel = x^T dz", self.G[f"{pref}.kernel"], x.view(B, -1).t() @ dz_g, 2e-4)
        return dz_g

    def max_bwd(self, pref, cwn, mwn, block, a, amask, zs, dG, prev_wn, prev_C, addend=None):
        """canonical backward of ConvLayer(128->1024)+BN+ReLU+reduce_max from the GPU's dG, arg-max rows and statistics: parameter
        gradients and the gradient of the layer below's BN output (prev.dy), which the GPU forms in Gram-matrix form (pn_maxbwd.hip)"""
        P, B, N, M = self.P, self.B, self.N, self.M
        W = P[f"{pref}.kernel"]; Wq = self.qa(W)
        g = P[f"{pref}.bn.gamma"]
        mean, inv = self.ws(cwn + ".mean"), self.ws(cwn + ".invstd")
        s = g * inv
        arg = self.ws(mwn + ".arg", torch.int32).long().view(B, 1024)
        gfeat = self.ws(mwn + ".g").view(B, 1024)
        hs = dG.view(B, 1024) * (gfeat > 0)
        self.judge(f"{pref} hs = scale dG [g > 0]", self.ws(mwn + ".hs"), hs * self.ws(cwn + ".scale"), 1e-5)
        batch = self.bn_batch(block)
        zst = torch.stack([zs[b].gather(0, arg[b].unsqueeze(0)).squeeze(0) for b in range(B)])
        zh_st = (zst - mean) * inv
        sum_dy = hs.sum(0); sum_dyz = (hs * zh_st).sum(0)
        if self.tr.get(block, True) and batch:
            self.judge(f"grad {pref}.bn.beta = sum dy", self.G[f"{pref}.bn.beta"], sum_dy, 1e-3, floor=_amax(hs))
            self.judge(f"grad {pref}.bn.gamma = sum dy zhat", self.G[f"{pref}.bn.gamma"], sum_dyz, 1e-3, floor=_amax(hs))
        dW = torch.zeros(128, 1024, dtype=torch.float64)
        dA = torch.empty(M, 128, dtype=torch.float64)
        for b in range(B):
            dz = torch.zeros(N, 1024, dtype=torch.float64)
            dz.scatter_(0, arg[b].unsqueeze(0), hs[b].unsqueeze(0))
            if batch:
                zh = (zs[b] - mean) * inv
                dz = s * (dz - sum_dy / M - zh * (sum_dyz / M))
            else:
                dz = s * dz
            Ab = a[b * N:(b + 1) * N]
            dW += Ab.t() @ dz
            dA[b * N:(b + 1) * N] = dz @ Wq.t()
        if self.tr.get(block, True):
            self.judge(f"grad {pref}.kernel (Gram form on the GPU)", self.G[f"{pref}.kernel"], dW, self.t_gram)
        if addend is not None:
            dA = dA + addend
        dprev = dA * amask
        self.judge(f"{pref} -> d(BN output) of the layer below ({prev_wn}.dy)", self.ws_act(prev_wn + ".dy", prev_C), dprev, self.t_gram)

    def conv_bwd(self, pref, wn, block, x_act, prev=None, model=None):
        """interior per-point layer: BN-backward coefficients, kernel gradient x^T dz, and the data gradient into the layer below
        prev = (prev_wn, prev_C, mask, addend or None)"""
        P = self.P
        dzq, _ = self.bn_bwd_coeffs(pref, wn, block, model)
        W = P[f"{pref}.kernel"]
        if self.tr.get(block, True):
            self.judge(f"grad {pref}.kernel = x^T dz", self.G[f"{pref}.kernel"], x_act.t() @ dzq, self.t_wg)
        if prev is not None:
            pwn, pC, mask, addend = prev
            d = dzq @ self.qa(W).t()
            if addend is not None:
                d = d + addend
            if mask is not None:
                d = d * mask
            self.stored(f"{pref} -> d(BN output) of the layer below ({pwn}.dy)", self.ws_act(pwn + ".dy", pC), d, bwd=True)
        return dzq

    # ---------------------------------------------------------------- the whole step
    def run(self):
        P, B, N, M, m = self.P, self.B, self.N, self.M, self.m
        van = self.vanilla
        # ---- forward ----
        pcn_ref, _ = O.normalize(self.pc.double())
        pcn = self.ws("pcn").view(M, 3)
        self.judge("normalised cloud (PointCloudNormalization)", pcn, pcn_ref, 2e-5)       # fp32 sums over N points; values <= 1
        zs = {}
        if not van:
            self.fwd_conv("input_transform.conv1", "iT.c1", "input_transform", pcn, P["input_transform.conv1.kernel"], quant_w=False)
            a, _ = self.lazy("iT.c1", 64)
            self.fwd_conv("input_transform.conv2", "iT.c2", "input_transform", a, P["input_transform.conv2.kernel"])
            a, _ = self.lazy("iT.c2", 128)
            zs["iT"] = self.fwd_max("input_transform.conv3", "iT.c3", "iT.m3", "input_transform", a)
            h = self.fwd_dense("input_transform.dense1", "iT.d1", "input_transform", self.ws("iT.m3.g").view(B, 1024))
            h = self.fwd_dense("input_transform.dense2", "iT.d2", "input_transform", self.ws("iT.d1.a").view(B, 512))
            R = (self.ws("iT.d2.a").view(B, 256) @ P["input_transform.w"]).view(B, 3, 3) + P["input_transform.b"]
            self.judge("input transform R = a2 . w + b", self.ws("iT.R"), R, 1e-4)
            Rg = self.ws("iT.R").view(B, 3, 3)
            Weff = Rg @ P["mlp_1_1.kernel"]
            self.judge("W_eff = R . W(mlp_1_1): tf.matmul(pc, R) folded", self.ws("Weff1"), Weff, 1e-5)
            self.fwd_conv("mlp_1_1", "m11", "mlp_1_1", pcn, self.ws("Weff1").view(B, 3, 64), quant_w=False)
        else:
            self.fwd_conv("mlp_1_1", "m11", "mlp_1_1", pcn, P["mlp_1_1.kernel"], quant_w=False)
        a11, _ = self.lazy("m11", 64)
        self.fwd_conv("mlp_1_2", "m12", "mlp_1_2", a11, P["mlp_1_2.kernel"])
        a12, mask12 = self.lazy("m12", 64)
        if not van:
            self.fwd_conv("feature_transform.conv1", "fT.c1", "feature_transform", a12, P["feature_transform.conv1.kernel"])
            a, _ = self.lazy("fT.c1", 64)
            self.fwd_conv("feature_transform.conv2", "fT.c2", "feature_transform", a, P["feature_transform.conv2.kernel"])
            a, _ = self.lazy("fT.c2", 128)
            zs["fT"] = self.fwd_max("feature_transform.conv3", "fT.c3", "fT.m3", "feature_transform", a)
            self.fwd_dense("feature_transform.dense1", "fT.d1", "feature_transform", self.ws("fT.m3.g").view(B, 1024))
            self.fwd_dense("feature_transform.dense2", "fT.d2", "feature_transform", self.ws("fT.d1.a").view(B, 512))
            R64 = (self.ws("fT.d2.a").view(B, 256) @ P["feature_transform.w"]).view(B, 64, 64) + P["feature_transform.b"]
            self.judge("feature transform R_64", self.ws("fT.R"), R64, 1e-4)
            R64g = self.ws("fT.R").view(B, 64, 64)
            x64_ref = torch.einsum("bnk,bkc->bnc", a12.view(B, N, 64), self.qa(R64g)).reshape(M, 64)
            self.stored("X_64 = relu(bn(mlp_1_2)) . R_64", self.ws_act("X64", 64), x64_ref)
            x64 = self.qa(self.ws_act("X64", 64))
        else:
            x64 = a12
        self.fwd_conv("mlp_2_1", "m21", "mlp_2_1", x64, P["mlp_2_1.kernel"])
        a21, mask21 = self.lazy("m21", 64)
        self.fwd_conv("mlp_2_2", "m22", "mlp_2_2", a21, P["mlp_2_2.kernel"])
        a22, mask22 = self.lazy("m22", 128)
        zs["m23"] = self.fwd_max("mlp_2_3", "m23", "mm23", "mlp_2_3", a22)
        G_ = self.ws("mm23.g").view(B, 1024)
        # classification head
        self.fwd_dense("mlp_cls_1", "c1", "mlp_cls_1", G_, self.keep["dropout_1"])
        self.fwd_dense("mlp_cls_2", "c2", "mlp_cls_2", self.ws("c1.a").view(B, 512), self.keep["dropout_2"])
        logits = self.ws("c2.a").view(B, 256) @ P["mlp_cls_3.kernel"] + P["mlp_cls_3.bias"]
        self.judge("classification logits", self.ws("cls_logits"), logits, 1e-4)
        lg = self.ws("cls_logits").view(B, -1)
        self.judge("classification probabilities (softmax)", self.outs[0], torch.softmax(lg, -1), 1e-5)
        # segmentation head (layer by layer when its activations exist)
        seg_model = None
        have_seg_acts = True
        if self.seg_fused:
            seg_model = self.seg_src
            have_seg_acts = seg_model is not None
        if have_seg_acts:
            w1 = P["mlp_seg_1.kernel"]
            gb = G_ @ w1[64:]
            self.judge("seg_l1 global half: g . W1[64:] (per-cloud bias)", self.ws("gb", model=seg_model), gb, 1e-4)
            self.fwd_conv("mlp_seg_1", "s1", "mlp_seg_1", x64, w1[:64], cloud_bias=self.ws("gb", model=seg_model).view(B, 512), model=seg_model)
            a1, mask_s1 = self.lazy("s1", 512, seg_model)
            self.fwd_conv("mlp_seg_2", "s2", "mlp_seg_2", a1, P["mlp_seg_2.kernel"], model=seg_model)
            a2, mask_s2 = self.lazy("s2", 256, seg_model)
            self.fwd_conv("mlp_seg_3", "s3", "mlp_seg_3", a2, P["mlp_seg_3.kernel"], model=seg_model)
            a3, mask_s3 = self.lazy("s3", 128, seg_model)
            self.fwd_conv("mlp_seg_4", "s4", "mlp_seg_4", a3, P["mlp_seg_4.kernel"], model=seg_model)
            z4 = self.ws_act("s4.Z", 128, seg_model)
            y4 = torch.relu(z4 * self.ws("s4.scale", model=seg_model) + self.ws("s4.shift", model=seg_model))       # seg_l5 is an fp32 layer: no operand rounding
            seg_logits = y4 @ P["mlp_seg_5.kernel"] + P["mlp_seg_5.bias"]
            seg_p = torch.softmax(seg_logits, -1)
            self.judge("segmentation probabilities (seg_l5 + softmax)", self.outs[1], seg_p, 1e-4)
        if not van:
            self.judge("third output = R", self.outs[2], self.ws("iT.R"), 0.0, floor=1.0)
        # losses from the GPU's own outputs
        sc = m.scalars.cpu().double()
        l_cls = O.keras_sparse_cce(self.outs[0].double(), self.y_cls)
        self._lim("classification loss (Keras SCCE of the GPU's probabilities)", abs(float(sc[0] / B) - float(l_cls)), 1e-5 * max(1.0, float(l_cls)))
        l_seg = O.keras_sparse_cce(self.outs[1].double().view(B, N, -1), self.y_seg)
        self._lim("segmentation loss", abs(float(sc[2] / M) - float(l_seg)), 1e-5 * max(1.0, float(l_seg)))
        l_se3 = O.keras_mse(self.outs[2].double(), self.se3.double())
        self._lim("se3 loss (MSE)", abs(float(sc[4] / (B * 9)) - float(l_se3)), 1e-5 * max(1.0, float(l_se3)))
        acc = float((self.outs[0].argmax(-1) == self.y_cls).double().sum())
        self._lim("classification accuracy count", abs(float(sc[1]) - acc), 0.5)

        # ---- backward ----
        lw = self.lw
        has_cls = lw[0] != 0.0
        has_seg = lw[1] != 0.0
        dGc = dGs = None
        if has_cls:
            lgt = lg.clone().requires_grad_(True)
            (O.keras_sparse_cce(torch.softmax(lgt, -1), self.y_cls) * lw[0]).backward()
            self.judge("d(loss)/d(classification logits)", self.ws("cls_dlogits"), lgt.grad, 1e-4)
            dl = self.ws("cls_dlogits").view(B, -1)
            if self.tr.get("mlp_cls_3", True):
                self.judge("grad mlp_cls_3.kernel", self.G["mlp_cls_3.kernel"], self.ws("c2.a").view(B, 256).t() @ dl, 2e-4)
                self.judge("grad mlp_cls_3.bias", self.G["mlp_cls_3.bias"], dl.sum(0), 2e-4)
            da2 = dl @ P["mlp_cls_3.kernel"].t()
            self.judge("d(mlp_cls_2 output)", self.ws("c3.din"), da2, 2e-4)
            dz2 = self.dense_bwd("mlp_cls_2", "c2", "mlp_cls_2", self.ws("c3.din"), self.ws("c1.a"), self.keep["dropout_2"])
            self.judge("d(mlp_cls_1 output)", self.ws("c2.din"), dz2 @ P["mlp_cls_2.kernel"].t(), 2e-4)
            dz1 = self.dense_bwd("mlp_cls_1", "c1", "mlp_cls_1", self.ws("c2.din"), G_, self.keep["dropout_1"])
            self.judge("d(global feature) from the classification head", self.ws("dGcls"), dz1 @ P["mlp_cls_1.kernel"].t(), 2e-4)
            dGc = self.ws("dGcls").view(B, 1024)
        dx64_seg = None
        if has_seg:
            slg = seg_logits.clone().requires_grad_(True)
            (O.keras_sparse_cce(torch.softmax(slg, -1).view(B, N, -1), self.y_seg) * lw[1]).backward()
            self.judge("d(loss)/d(segmentation logits)", self.ws("seg_dlogits"), slg.grad, 1e-3)
            dls = self.ws("seg_dlogits").view(M, -1)
            if self.tr.get("mlp_seg_5", True):
                self.judge("grad mlp_seg_5.kernel", self.G["mlp_seg_5.kernel"], y4.t() @ dls, 1e-3)
                self.judge("grad mlp_seg_5.bias", self.G["mlp_seg_5.bias"], dls.sum(0), 1e-3)
            _, mask_s4 = self.lazy("s4", 128)
            d4 = (dls @ P["mlp_seg_5.kernel"].t()) * mask_s4
            self.stored("seg_l5 -> d(BN output) of seg_l4 (s4.dy)", self.ws_act("s4.dy", 128), d4, bwd=True)
            self.conv_bwd("mlp_seg_4", "s4", "mlp_seg_4", a3, ("s3", 128, mask_s3, None))
            self.conv_bwd("mlp_seg_3", "s3", "mlp_seg_3", a2, ("s2", 256, mask_s2, None))
            self.conv_bwd("mlp_seg_2", "s2", "mlp_seg_2", a1, ("s1", 512, mask_s1, None))
            dz1q, dz1 = self.bn_bwd_coeffs("mlp_seg_1", "s1", "mlp_seg_1")
            if self.tr.get("mlp_seg_1", True):
                gk = self.G["mlp_seg_1.kernel"].cpu().double()
                self.judge("grad mlp_seg_1.kernel[:64] = x64^T dz", gk[:64], x64.t() @ dz1q, self.t_wg)
            dgb = dz1.view(B, N, 512).sum(1)            # the per-cloud bias sees the unrounded dz of every point of its cloud
            self.judge("d(per-cloud bias of seg_l1) = sum over the cloud's points of dz", self.ws("dgb"), dgb, 2e-3 if not self.s16 else 2.0 ** -7)
            dgbg = self.ws("dgb").view(B, 512)
            if self.tr.get("mlp_seg_1", True):
                self.judge("grad mlp_seg_1.kernel[64:] = g^T dgb", gk[64:], G_.t() @ dgbg, 2e-4)
            self.judge("d(global feature) from the segmentation head", self.ws("dGseg"), dgbg @ w1[64:].t(), 2e-4)
            dGs = self.ws("dGseg").view(B, 1024)
            dx64_seg = dz1q @ self.qa(w1[:64]).t()
        if not (has_cls or has_seg):
            return self.fails
        dG = (dGc if dGc is not None else 0) + (dGs if dGs is not None else 0)
        self.max_bwd("mlp_2_3", "m23", "mm23", "mlp_2_3", a22, mask22, zs["m23"], dG, "m22", 128)
        self.conv_bwd("mlp_2_2", "m22", "mlp_2_2", a21, ("m21", 64, mask21, None))
        dz21q, _ = self.bn_bwd_coeffs("mlp_2_1", "m21", "mlp_2_1")
        if self.tr.get("mlp_2_1", True):
            self.judge("grad mlp_2_1.kernel = x64^T dz", self.G["mlp_2_1.kernel"], x64.t() @ dz21q, self.t_wg)
        dx = dz21q @ self.qa(P["mlp_2_1.kernel"]).t()
        if dx64_seg is not None:
            dx = dx + dx64_seg
        if van:
            self.stored("d(BN output) of mlp_1_2 (m12.dy)", self.ws_act("m12.dy", 64), dx * mask12, bwd=True)
        else:
            self.stored("d(X_64) = mlp_2_1 part + segmentation part", self.ws_act("dX64", 64), dx, bwd=True)
            dxg = self.qa(self.ws_act("dX64", 64))
            dR = torch.einsum("bnk,bnc->bkc", a12.view(B, N, 64), dxg.view(B, N, 64))
            # feature transform: X_64 = A_12 . R_64 -> dR_64 = A_12^T dX_64 per cloud, dA_12 = dX_64 . R_64^T
            if self.reg:
                Rm = R64g
                dR = dR + 1e-3 * 2 * ((Rm @ Rm.transpose(1, 2) - torch.eye(64, dtype=torch.float64)) @ Rm)
            self.judge("d(R_64) = A_12^T dX_64 per cloud (+ regulariser)", self.ws("fT.dR"), dR, self.t_wg)
            tmp = torch.einsum("bnc,bkc->bnk", dxg.view(B, N, 64), self.qa(R64g)).reshape(M, 64)
            self.stored("dX_64 . R_64^T (tmpA12)", self.ws_act("tmpA12", 64), tmp, bwd=True)
            self.tnet_bwd("feature_transform", "fT", 64, zs["fT"], a12)
            dzf1q, _ = self.bn_bwd_coeffs("feature_transform.conv1", "fT.c1", "feature_transform")
            if self.tr.get("feature_transform", True):
                self.judge("grad feature_transform.conv1.kernel", self.G["feature_transform.conv1.kernel"], a12.t() @ dzf1q, self.t_wg)
            d12 = (dzf1q @ self.qa(P["feature_transform.conv1.kernel"]).t() + self.ws_act("tmpA12", 64)) * mask12
            self.stored("d(BN output) of mlp_1_2 (m12.dy): T-Net path + transform path", self.ws_act("m12.dy", 64), d12, bwd=True)
        _, a11m = self.lazy("m11", 64)
        self.conv_bwd("mlp_1_2", "m12", "mlp_1_2", a11, ("m11", 64, a11m, None))
        _, dz11 = self.bn_bwd_coeffs("mlp_1_1", "m11", "mlp_1_1")
        dWeff = torch.einsum("bnk,bnc->bkc", pcn.view(B, N, 3), dz11.view(B, N, 64))
        if van:
            if self.tr.get("mlp_1_1", True):
                self.judge("grad mlp_1_1.kernel", self.G["mlp_1_1.kernel"], dWeff.sum(0), 1e-3)
            return self.fails
        if self.tr.get("mlp_1_1", True):
            self.judge("grad mlp_1_1.kernel = sum_b R_b^T dW_eff,b", self.G["mlp_1_1.kernel"], torch.einsum("bji,bjc->ic", Rg, dWeff), 1e-3)
        dRin = torch.einsum("bic,jc->bij", dWeff, P["mlp_1_1.kernel"])
        if lw[2] != 0.0:
            dRin = dRin + 2.0 * lw[2] / (B * 9) * (Rg - self.se3.double())
        if self.reg:
            dRin = dRin + 1e-3 * 2 * ((Rg @ Rg.transpose(1, 2) - torch.eye(3, dtype=torch.float64)) @ Rg)
        self.judge("d(R) = dW_eff . W^T (+ se3 loss, regulariser)", self.ws("iT.dR"), dRin, 1e-3)
        self.tnet_bwd("input_transform", "iT", 3, zs["iT"], None)
        _, dzi1 = self.bn_bwd_coeffs("input_transform.conv1", "iT.c1", "input_transform")
        if self.tr.get("input_transform", True):
            self.judge("grad input_transform.conv1.kernel", self.G["input_transform.conv1.kernel"], pcn.t() @ dzi1, 1e-3)
        return self.fails

    def tnet_bwd(self, name, wn, K, zs, x_in):
        P, B = self.P, self.B
        dR = self.ws(f"{wn}.dR").view(B, K * K)
        tr = self.tr.get(name, True)
        a2 = self.ws(f"{wn}.d2.a").view(B, 256)
        if tr:
            self.judge(f"grad {name}.w = a2^T dR", self.G[f"{name}.w"], a2.t() @ dR, 2e-4)
            self.judge(f"grad {name}.b = sum_b dR", self.G[f"{name}.b"], dR.sum(0), 2e-4)
        self.judge(f"{name} d(dense2 output)", self.ws(f"{wn}.da2"), dR @ P[f"{name}.w"].t(), 2e-4)
        dz2 = self.dense_bwd(f"{name}.dense2", f"{wn}.d2", name, self.ws(f"{wn}.da2"), self.ws(f"{wn}.d1.a"))
        self.judge(f"{name} d(dense1 output)", self.ws(f"{wn}.d2.din"), dz2 @ P[f"{name}.dense2.kernel"].t(), 2e-4)
        dz1 = self.dense_bwd(f"{name}.dense1", f"{wn}.d1", name, self.ws(f"{wn}.d2.din"), self.ws(f"{wn}.m3.g"))
        self.judge(f"{name} d(pooled feature)", self.ws(f"{wn}.m3.dG"), dz1 @ P[f"{name}.dense1.kernel"].t(), 2e-4)
        a2c, mask2 = self.lazy(f"{wn}.c2", 128)
        self.max_bwd(f"{name}.conv3", f"{wn}.c3", f"{wn}.m3", name, a2c, mask2, zs, self.ws(f"{wn}.m3.dG"), f"{wn}.c2", 128)
        a1c, mask1 = self.lazy(f"{wn}.c1", 64)
        self.conv_bwd(f"{name}.conv2", f"{wn}.c2", name, a1c, (f"{wn}.c1", 64, mask1, None))


def check_layers(m, outs, params, pc, y_cls, y_seg, se3, keep, trainable, lw, precision, vanilla, tag, report, seg_fused=False, seg_src=None,
                 reg=False):
    """returns the list of failed lines [(what, err, limit)]"""
    f = Forced(m, params, pc, y_cls, y_seg, se3, keep, trainable, lw, precision, vanilla, tag, report, seg_src, reg)
    f.outs = [o.detach().cpu().double() for o in outs]
    f.seg_fused = seg_fused
    return f.run()
