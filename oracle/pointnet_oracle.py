"""CPU oracle for the PointNet path -- TEST INFRASTRUCTURE, never imported by the product.

A plain torch-CPU restatement (fp32, or fp64 for tolerance studies) of the reference model, written
from the reference source, function by function:

  PointCloudNormalization.call   point_cloud_analysis/pointnet/PointNet.py:691-706
  ConvLayer.call                 point_cloud_analysis/pointnet/PointNet.py:554-566  (1x1 Conv2D, no bias when BN)
  DenseLayer.call                point_cloud_analysis/pointnet/PointNet.py:642-654
  TNet.call                      point_cloud_analysis/pointnet/PointNet.py:418-454
  PointNet.call                  point_cloud_analysis/pointnet/PointNet.py:197-292
  freeze/thaw semantics          point_cloud_analysis/pointnet/PointNet.py:294-342,469-490,585-594,670-679
  loss / optimizer assembly      point_cloud_analysis/pointnet_train.py:310-351

The arithmetic itself lives in TensorFlow 2.20 / Keras 3.10 (requirements.txt:113,46), which is absent
from this image; the Keras semantics restated here are the library's documented defaults:
BatchNormalization(momentum=0.99, epsilon=1e-3, biased batch variance, moving <- 0.99*moving+0.01*batch,
inference statistics whenever training=False OR the layer is not trainable), Dropout scale 1/(1-rate),
SparseCategoricalCrossentropy(from_logits=False) = -log_softmax(log(clip(p,1e-7,1-1e-7)))[label],
MeanSquaredError = mean over all elements, tf.nn.l2_loss = sum(x^2)/2, Adam(beta 0.9/0.999, eps 1e-7
outside the sqrt, bias correction folded into the step size), ExponentialDecay(staircase=False).

PARITY UNPINNED (see oracle/__init__.py): no reference golden vector exists for these outputs.

Gradients come from torch autograd over these elementary ops; nothing here calls a fused torch.nn
layer, so every formula is visible.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Callable, Dict, Optional

import torch

BN_EPS = 1e-3          # keras.layers.BatchNormalization default epsilon
BN_MOMENTUM = 0.99     # PointNet.py:502,603 (bn_momentum default)
NORM_MIN_SCALE = 1e-7  # PointNet.py:701
KERAS_EPS = 1e-7       # keras.backend.epsilon()

# layer_trainability names, in the order the reference logs them
# (models/f15_scale_lidar/log_20260126_16*0916.log:203-218; PointNet.py:144-159)
TRAINABILITY_NAMES = [
    "input_normalization", "input_transform",
    "s1_l1_64_convolution_layer", "s1_l2_64_convolution_layer", "feature_transform",
    "s2_l1_64_convolution_layer", "s2_l2_128_convolution_layer", "s2_l3_1024_convolution_layer",
    "s3_l1_512_dense_layer", "s3_l2_256_dense_layer", "output_dense_layer",
    "seg_l1_512_convolution_layer", "seg_l2_256_convolution_layer", "seg_l3_128_convolution_layer",
    "seg_l4_128_convolution_layer", "seg_l5_output_convolution_layer",
]

# trainability groups as set by the four freeze_/thaw_ pairs (PointNet.py:294-342)
GROUPS = {
    "input_transform": ["input_transform"],
    "shared_network": ["input_transform", "mlp_1_1", "mlp_1_2", "feature_transform",
                       "mlp_2_1", "mlp_2_2", "mlp_2_3"],
    "classification_head": ["mlp_cls_1", "mlp_cls_2", "mlp_cls_3"],
    "segmentation_head": ["mlp_seg_1", "mlp_seg_2", "mlp_seg_3", "mlp_seg_4", "mlp_seg_5"],
}
ALL_BLOCKS = ["input_transform", "mlp_1_1", "mlp_1_2", "feature_transform", "mlp_2_1", "mlp_2_2",
              "mlp_2_3", "mlp_cls_1", "mlp_cls_2", "mlp_cls_3", "mlp_seg_1", "mlp_seg_2", "mlp_seg_3",
              "mlp_seg_4", "mlp_seg_5"]


# ----------------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------------
def layer_table(ccls: int, cseg: int, vanilla: bool = False):
    """(prefix, kind, cin, cout, has_bn) for every ConvLayer/DenseLayer, in PointNet.py:116-141 order."""
    t = []

    def tnet(prefix, k):
        t.append((f"{prefix}.conv1", "conv", k, 64, True))      # PointNet.py:406
        t.append((f"{prefix}.conv2", "conv", 64, 128, True))    # :407
        t.append((f"{prefix}.conv3", "conv", 128, 1024, True))  # :408
        t.append((f"{prefix}.dense1", "dense", 1024, 512, True))  # :409
        t.append((f"{prefix}.dense2", "dense", 512, 256, True))   # :410

    if not vanilla:
        tnet("input_transform", 3)
    t.append(("mlp_1_1", "conv", 3, 64, True))
    t.append(("mlp_1_2", "conv", 64, 64, True))
    if not vanilla:
        tnet("feature_transform", 64)
    t.append(("mlp_2_1", "conv", 64, 64, True))
    t.append(("mlp_2_2", "conv", 64, 128, True))
    t.append(("mlp_2_3", "conv", 128, 1024, True))
    t.append(("mlp_cls_1", "dense", 1024, 512, True))
    t.append(("mlp_cls_2", "dense", 512, 256, True))
    t.append(("mlp_cls_3", "dense", 256, ccls, False))   # apply_bn default False -> use_bias (PointNet.py:134,630)
    t.append(("mlp_seg_1", "conv", 1088, 512, True))
    t.append(("mlp_seg_2", "conv", 512, 256, True))
    t.append(("mlp_seg_3", "conv", 256, 128, True))
    t.append(("mlp_seg_4", "conv", 128, 128, True))
    t.append(("mlp_seg_5", "conv", 128, cseg, False))    # apply_bn=False -> bias (PointNet.py:141,540)
    return t


def glorot_uniform(shape, gen, dtype):
    """keras.initializers.GlorotUniform: U(-l, l), l = sqrt(6 / (fan_in + fan_out)).  The TF random
    stream itself is not reproducible here: parity tests always inject weights."""
    fan_in, fan_out = shape[0], shape[1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return ((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(dtype)


def init_params(ccls: int, cseg: int, seed: int = 42, vanilla: bool = False,
                dtype=torch.float32, randomize_bn: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """All weights of the model keyed by canonical name.  `randomize_bn=True` perturbs gamma/beta and the
    moving statistics away from their (1,0,0,1) initial values so tests exercise every term."""
    gen = torch.Generator().manual_seed(seed)
    p: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for prefix, kind, cin, cout, has_bn in layer_table(ccls, cseg, vanilla):
        p[f"{prefix}.kernel"] = glorot_uniform((cin, cout), gen, dtype)
        if has_bn:
            if randomize_bn:
                p[f"{prefix}.bn.gamma"] = (1 + 0.3 * torch.randn(cout, generator=gen, dtype=torch.float64)).to(dtype)
                p[f"{prefix}.bn.beta"] = (0.2 * torch.randn(cout, generator=gen, dtype=torch.float64)).to(dtype)
                p[f"{prefix}.bn.moving_mean"] = (0.1 * torch.randn(cout, generator=gen, dtype=torch.float64)).to(dtype)
                p[f"{prefix}.bn.moving_var"] = (0.5 + torch.rand(cout, generator=gen, dtype=torch.float64)).to(dtype)
            else:
                p[f"{prefix}.bn.gamma"] = torch.ones(cout, dtype=dtype)
                p[f"{prefix}.bn.beta"] = torch.zeros(cout, dtype=dtype)
                p[f"{prefix}.bn.moving_mean"] = torch.zeros(cout, dtype=dtype)
                p[f"{prefix}.bn.moving_var"] = torch.ones(cout, dtype=dtype)
        else:
            if randomize_bn:
                p[f"{prefix}.bias"] = (0.1 * torch.randn(cout, generator=gen, dtype=torch.float64)).to(dtype)
            else:
                p[f"{prefix}.bias"] = torch.zeros(cout, dtype=dtype)
        if prefix.endswith(".dense2"):   # T-Net tail: w (256, K^2) glorot, b (K, K) identity (PointNet.py:415-416)
            tn = prefix.split(".")[0]
            k = 3 if tn == "input_transform" else 64
            p[f"{tn}.w"] = glorot_uniform((256, k * k), gen, dtype)
            p[f"{tn}.b"] = torch.eye(k, dtype=dtype)
            if randomize_bn:
                p[f"{tn}.b"] = p[f"{tn}.b"] + (0.05 * torch.randn(k, k, generator=gen, dtype=torch.float64)).to(dtype)
    return p


def is_trainable_name(name: str) -> bool:
    return not (name.endswith("moving_mean") or name.endswith("moving_var"))


def block_of(name: str) -> str:
    return name.split(".")[0]


def census(params) -> tuple:
    tr = sum(v.numel() for k, v in params.items() if is_trainable_name(k))
    nt = sum(v.numel() for k, v in params.items() if not is_trainable_name(k))
    return tr, nt


# ----------------------------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------------------------
def normalize(pc: torch.Tensor):
    """PointCloudNormalization.call, PointNet.py:691-706."""
    centroid = pc.mean(dim=1, keepdim=True)                       # :694
    centered = pc - centroid                                      # :695
    dist = torch.sqrt((centered * centered).sum(dim=-1))          # :696
    max_dist = dist.max(dim=1, keepdim=True).values.unsqueeze(-1)  # :697-698
    scale = torch.clamp(max_dist, min=NORM_MIN_SCALE)             # :701
    return centered / scale, (centroid, scale)                    # :704


class _Ctx:
    """Per-call state: which blocks are trainable, new moving statistics, regularisation losses."""

    def __init__(self, training, trainable, quant, dropout_masks, dropout_rate, tie_split, decisions=None, store_quant=None):
        self.training = training
        self.trainable = trainable
        self.quant = quant
        self.store_quant = store_quant
        self.dropout_masks = dropout_masks or {}
        self.dropout_rate = dropout_rate
        self.new_stats: Dict[str, torch.Tensor] = {}
        self.reg_losses = []
        self.tie_split = tie_split
        self.taps: Dict[str, torch.Tensor] = {}
        self.decisions = decisions or {}


class _RoundGrad(torch.autograd.Function):
    """identity whose gradient is rounded by `fn` on the way back (emulation of a 16-bit store of an activation gradient)"""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return ctx.fn(g), None


def _bn(ctx: _Ctx, p, prefix: str, z: torch.Tensor, z_stored=None):
    """keras BatchNormalization over every axis but the last (PointNet.py:528,559,623,647).  z_stored (storage emulation of the
    implementation under test): the statistics come from z, the affine is applied to z_stored."""
    gamma, beta = p[f"{prefix}.bn.gamma"], p[f"{prefix}.bn.beta"]
    block_trainable = ctx.trainable.get(block_of(prefix), True)
    red = tuple(range(z.dim() - 1))
    if ctx.training and block_trainable:
        mean = z.mean(dim=red)
        var = ((z - mean) ** 2).mean(dim=red)                     # biased, as tf.nn.moments
        mm, mv = p[f"{prefix}.bn.moving_mean"], p[f"{prefix}.bn.moving_var"]
        ctx.new_stats[f"{prefix}.bn.moving_mean"] = (mm * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM))
        ctx.new_stats[f"{prefix}.bn.moving_var"] = (mv * BN_MOMENTUM + var.detach() * (1 - BN_MOMENTUM))
    else:                                                         # inference mode incl. frozen layers
        mean, var = p[f"{prefix}.bn.moving_mean"], p[f"{prefix}.bn.moving_var"]
    inv = torch.rsqrt(var + BN_EPS) * gamma                       # tf.nn.batch_normalization
    return (z if z_stored is None else z_stored) * inv + (beta - mean * inv)


def _relu(ctx: _Ctx, prefix: str, y):
    """ReLU; with an injected decision mask (tests: the discrete choices of the implementation under test are
    imposed so that the remaining comparison is between continuous functions)."""
    m = ctx.decisions.get(f"{prefix}.relu")
    if m is not None:
        return y * m.to(y.dtype).reshape(y.shape)
    return torch.relu(y)


def _mm(ctx: _Ctx, a, w, quantize: bool):
    if quantize and ctx.quant is not None:
        return ctx.quant(a) @ ctx.quant(w)
    return a @ w


def conv_layer(ctx, p, prefix, x, act="relu", mfma=True, kernel=None, stored=True):
    """ConvLayer.call PointNet.py:554-566: 1x1 conv == per-point matmul with kernel (Cin, Cout).
    stored / ctx.store_quant: emulation of an implementation that keeps this layer's pre-BN output and the gradient of its BN
    output in a narrower type (tests only; the max-pooled layers are never stored)."""
    w = p[f"{prefix}.kernel"] if kernel is None else kernel
    z = _mm(ctx, x, w, mfma)
    ctx.taps[f"{prefix}.z"] = z
    if f"{prefix}.bn.gamma" in p:
        sq = ctx.store_quant if stored else None
        z = _bn(ctx, p, prefix, z, None if sq is None else sq(z))
        ctx.taps[f"{prefix}.y"] = z
        if sq is not None:
            z = _RoundGrad.apply(z, sq)
    else:
        z = z + p[f"{prefix}.bias"]
    if act == "relu":
        return _relu(ctx, prefix, z)
    if act == "softmax":
        return torch.softmax(z, dim=-1)
    return z


def dense_layer(ctx, p, prefix, x, act="relu"):
    """DenseLayer.call PointNet.py:642-654."""
    z = x @ p[f"{prefix}.kernel"]
    if f"{prefix}.bn.gamma" in p:
        z = _bn(ctx, p, prefix, z)
    else:
        z = z + p[f"{prefix}.bias"]
    if act == "relu":
        return _relu(ctx, prefix, z)
    if act == "softmax":
        return torch.softmax(z, dim=-1)
    return z


class _MaxTiesToFirst(torch.autograd.Function):
    """reduce_max over axis 1 whose gradient goes to the lowest-index maximum.  TF's reduce_max splits the
    gradient equally among ties; the build sends it to the first.  Parameter gradients agree
    whenever ties are duplicated points (identical rows at every layer)."""

    @staticmethod
    def forward(ctx, x):
        v, idx = x.max(dim=1)
        # torch.max returns *an* index of the maximum; force the lowest one
        eq = (x == v.unsqueeze(1))
        n = x.shape[1]
        ar = torch.arange(n, device=x.device).view(1, n, 1).expand_as(x)
        idx = torch.where(eq, ar, torch.full_like(ar, n)).min(dim=1).values
        ctx.save_for_backward(idx)
        ctx.n = n
        return v

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        out = torch.zeros(g.shape[0], ctx.n, g.shape[1], dtype=g.dtype)
        out.scatter_(1, idx.unsqueeze(1), g.unsqueeze(1))
        return out


class _MaxTiesSplit(torch.autograd.Function):
    """reduce_max with TF's gradient rule: dy / (number of tied maxima) to each tied position."""

    @staticmethod
    def forward(ctx, x):
        v = x.max(dim=1).values
        eq = (x == v.unsqueeze(1)).to(x.dtype)
        ctx.save_for_backward(eq)
        return v

    @staticmethod
    def backward(ctx, g):
        (eq,) = ctx.saved_tensors
        return eq / eq.sum(dim=1, keepdim=True) * g.unsqueeze(1)


def reduce_max_points(ctx: _Ctx, x, name: str = ""):
    idx = ctx.decisions.get(f"{name}.argmax")
    if idx is not None:      # injected argmax rows (B, C): value and gradient both go through that row
        return x.gather(1, idx.long().unsqueeze(1)).squeeze(1)
    return _MaxTiesSplit.apply(x) if ctx.tie_split else _MaxTiesToFirst.apply(x)


def tnet(ctx, p, name, x, regularize: bool):
    """TNet.call PointNet.py:418-454; x is (B, N, K); returns (B, K, K)."""
    k = x.shape[-1]
    h = conv_layer(ctx, p, f"{name}.conv1", x, mfma=(k >= 64))    # K=3 first layer stays fp32 on the GPU
    h = conv_layer(ctx, p, f"{name}.conv2", h)
    h = conv_layer(ctx, p, f"{name}.conv3", h, stored=False)
    g = reduce_max_points(ctx, h, name)                           # :429
    ctx.taps[f"{name}.global"] = g
    h = dense_layer(ctx, p, f"{name}.dense1", g)                  # :432
    h = dense_layer(ctx, p, f"{name}.dense2", h)                  # :433
    t = (h @ p[f"{name}.w"]).reshape(-1, k, k) + p[f"{name}.b"]   # :436-442
    if regularize:                                                # :447-451
        eye = torch.eye(k, dtype=t.dtype)
        d = eye - t @ t.transpose(1, 2)
        ctx.reg_losses.append(1e-3 * (d * d).sum() / 2)           # tf.nn.l2_loss over the whole batch
    return t


def forward(p, pc, training: bool = False, trainable: Optional[Dict[str, bool]] = None,
            vanilla: bool = False, regularize_input_transform: bool = False,
            regularize_feature_transform: bool = False, dropout_rate: float = 0.3,
            dropout_masks: Optional[Dict[str, torch.Tensor]] = None,
            quant: Optional[Callable] = None, tie_split: bool = False, return_ctx: bool = False,
            decisions: Optional[Dict[str, torch.Tensor]] = None, store_quant: Optional[Callable] = None):
    """PointNet.call PointNet.py:197-292.  Returns [cls (B,Ccls), seg (B,N,Cseg), R (B,3,3)] and, in
    training mode, the updated moving statistics (ctx.new_stats) plus add_loss terms (ctx.reg_losses).

    `dropout_masks`: {"dropout_1": (B,512) 0/1, "dropout_2": (B,256) 0/1} keep-masks; if absent in
    training mode dropout is skipped (the reference's dropout is unseeded, so it has no reproducible
    stream to match).  `quant`: optional rounding applied to both operands of every per-point matmul
    with K >= 64 (emulates the GPU's bf16 MFMA operand rounding for the bf16 configs).  `store_quant`: optional rounding of the
    per-point layer-boundary tensors an implementation stores between kernels (pre-BN outputs, X_64, and the gradients of the BN
    outputs on the way back) -- emulates bf16 storage; statistics are taken before it."""
    ctx = _Ctx(training, trainable or {}, quant, dropout_masks, dropout_rate, tie_split, decisions, store_quant)
    pcn, _ = normalize(pc)                                        # :202
    ctx.taps["pcn"] = pcn
    if not vanilla:
        R = tnet(ctx, p, "input_transform", pcn, regularize_input_transform)   # :206
        x = pcn @ R                                               # :207
    else:
        R = torch.eye(3, dtype=pc.dtype).expand(pc.shape[0], 3, 3)  # :211
        x = pcn
    x = conv_layer(ctx, p, "mlp_1_1", x, mfma=False)              # :217
    x = conv_layer(ctx, p, "mlp_1_2", x)                          # :220
    if not vanilla:
        R64 = tnet(ctx, p, "feature_transform", x, regularize_feature_transform)  # :227
        ctx.taps["R64"] = R64
        x64 = _mm(ctx, x, R64, True)                              # :228
        if store_quant is not None:
            x64 = _RoundGrad.apply(store_quant(x64), store_quant)
    else:
        x64 = x
    ctx.taps["x64"] = x64
    h = conv_layer(ctx, p, "mlp_2_1", x64)                        # :236
    h = conv_layer(ctx, p, "mlp_2_2", h)                          # :239
    h = conv_layer(ctx, p, "mlp_2_3", h, stored=False)            # :242
    g = reduce_max_points(ctx, h, "mlp_2_3")                      # :248
    ctx.taps["global"] = g

    c = dense_layer(ctx, p, "mlp_cls_1", g)                       # :252
    c = _dropout(ctx, "dropout_1", c)                             # :255
    c = dense_layer(ctx, p, "mlp_cls_2", c)                       # :257
    c = _dropout(ctx, "dropout_2", c)                             # :260
    cls = dense_layer(ctx, p, "mlp_cls_3", c, act="softmax")      # :262

    n = pc.shape[1]
    # tile + concat (:268-270).  Written as the algebraically identical split of seg_l1's kernel so that
    # the optional operand rounding applies to the per-point 64-wide part only, as on the GPU.
    w1 = p["mlp_seg_1.kernel"]
    z1 = _mm(ctx, x64, w1[:64], True) + (g @ w1[64:]).unsqueeze(1)
    ctx.taps["mlp_seg_1.z"] = z1
    z1 = _bn(ctx, p, "mlp_seg_1", z1, None if store_quant is None else store_quant(z1))
    ctx.taps["mlp_seg_1.y"] = z1
    if store_quant is not None:
        z1 = _RoundGrad.apply(z1, store_quant)
    s = _relu(ctx, "mlp_seg_1", z1)                               # :275
    s = conv_layer(ctx, p, "mlp_seg_2", s)                        # :278
    s = conv_layer(ctx, p, "mlp_seg_3", s)                        # :281
    s = conv_layer(ctx, p, "mlp_seg_4", s)                        # :284
    seg = conv_layer(ctx, p, "mlp_seg_5", s, act="softmax", mfma=False)  # :287
    assert seg.shape[1] == n
    if return_ctx:
        return [cls, seg, R], ctx
    return [cls, seg, R]


def forward_concat_form(p, pc, **kw):
    """Same model with seg_l1 fed by the literal tile+concat of PointNet.py:268-270 -- used by a test to
    show the split-kernel form above is the same function."""
    ctx = _Ctx(kw.get("training", False), kw.get("trainable", {}), None, None, 0.0, False)
    pcn, _ = normalize(pc)
    R = tnet(ctx, p, "input_transform", pcn, False)
    x = pcn @ R
    x = conv_layer(ctx, p, "mlp_1_1", x)
    x = conv_layer(ctx, p, "mlp_1_2", x)
    R64 = tnet(ctx, p, "feature_transform", x, False)
    x64 = x @ R64
    h = conv_layer(ctx, p, "mlp_2_1", x64)
    h = conv_layer(ctx, p, "mlp_2_2", h)
    h = conv_layer(ctx, p, "mlp_2_3", h)
    g = h.max(dim=1).values
    xs = torch.cat([x64, g.unsqueeze(1).expand(-1, pc.shape[1], -1)], dim=-1)
    s = conv_layer(ctx, p, "mlp_seg_1", xs)
    s = conv_layer(ctx, p, "mlp_seg_2", s)
    s = conv_layer(ctx, p, "mlp_seg_3", s)
    s = conv_layer(ctx, p, "mlp_seg_4", s)
    return conv_layer(ctx, p, "mlp_seg_5", s, act="softmax")


def _dropout(ctx: _Ctx, name, x):
    if not ctx.training or name not in ctx.dropout_masks:
        return x
    keep = ctx.dropout_masks[name].to(x.dtype)
    return x * keep / (1.0 - ctx.dropout_rate)                    # keras Dropout: scale kept units by 1/(1-rate)


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bfloat16 and back (the MFMA operand rounding of the bf16 configs)."""
    return x.to(torch.bfloat16).to(x.dtype)


def bf16x3_round(x: torch.Tensor) -> torch.Tensor:
    """hi + lo split used by the 'bf16x3' mode: x ~ bf16(x) + bf16(x - bf16(x)) (16 significant bits)."""
    hi = x.to(torch.bfloat16).to(x.dtype)
    lo = (x - hi).to(torch.bfloat16).to(x.dtype)
    return hi + lo


# ----------------------------------------------------------------------------------------------
# losses / metrics / optimizer  (pointnet_train.py:310-351)
# ----------------------------------------------------------------------------------------------
def keras_sparse_cce(probs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """keras.losses.SparseCategoricalCrossentropy(from_logits=False), reduction = mean over all labels.
    Keras 3 clips the probabilities to [eps, 1-eps], takes the log and feeds that as logits to
    tf.nn.sparse_softmax_cross_entropy_with_logits."""
    q = torch.log(torch.clamp(probs, KERAS_EPS, 1 - KERAS_EPS))
    logp = q - torch.logsumexp(q, dim=-1, keepdim=True)
    nll = -logp.gather(-1, labels.long().unsqueeze(-1)).squeeze(-1)
    return nll.mean()


def keras_mse(pred, target):
    return ((pred - target) ** 2).mean()


def total_loss(outputs, targets, loss_weights, reg_losses=()):
    """compile(loss={...}, loss_weights={...}) + model.losses (pointnet_train.py:334-345)."""
    cls, seg, R = outputs
    l_cls = keras_sparse_cce(cls, targets["classification_output"])
    l_seg = keras_sparse_cce(seg, targets["segmentation_output"])
    l_se3 = keras_mse(R, targets["se3"])
    tot = (loss_weights["classification"] * l_cls + loss_weights["segmentation"] * l_seg
           + loss_weights["rotation"] * l_se3)
    for r in reg_losses:
        tot = tot + r
    return tot, {"classification_output_loss": l_cls, "segmentation_output_loss": l_seg, "se3_loss": l_se3}


def sparse_categorical_accuracy(probs, labels):
    return (probs.argmax(dim=-1) == labels.long()).to(torch.float64).mean()


def exponential_decay_lr(lr0: float, step: int, decay_steps: int, decay_rate: float) -> float:
    """keras ExponentialDecay(staircase=False): lr0 * rate ** (step / decay_steps)."""
    return lr0 * decay_rate ** (step / decay_steps)


def keras_adam_step(param, grad, m, v, step_index: int, lr: float,
                    beta1: float = 0.9, beta2: float = 0.999, eps: float = KERAS_EPS):
    """One keras.optimizers.Adam update (in place).  `step_index` is optimizer.iterations before the
    update (0 for the first step); lr is the schedule evaluated at that index."""
    t = step_index + 1
    alpha = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m.add_((grad - m) * (1 - beta1))
    v.add_((grad * grad - v) * (1 - beta2))
    param.sub_(m * alpha / (torch.sqrt(v) + eps))


def train_step(p, pc, targets, loss_weights, trainable, opt_state, lr_cfg, step_index,
               dropout_masks=None, vanilla=False, reg_in=False, reg_feat=False, quant=None):
    """forward + 3-term loss + backward + Adam + moving-stat update, in place on `p` / `opt_state`.
    This is the 'step' bench.py's cpu_baseline leg times."""
    leaves = {}
    for k, t in p.items():
        tr = is_trainable_name(k) and trainable.get(block_of(k), True)
        leaves[k] = t.detach().clone().requires_grad_(tr)
    outs, ctx = forward(leaves, pc, training=True, trainable=trainable, vanilla=vanilla,
                        regularize_input_transform=reg_in, regularize_feature_transform=reg_feat,
                        dropout_masks=dropout_masks, quant=quant, return_ctx=True)
    loss, parts = total_loss(outs, targets, loss_weights, ctx.reg_losses)
    names = [k for k, t in leaves.items() if t.requires_grad]
    grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    lr = exponential_decay_lr(lr_cfg["rate"], step_index, lr_cfg["decay_steps"], lr_cfg["decay_rate"])
    with torch.no_grad():
        for k, g in zip(names, grads):
            if g is None:
                g = torch.zeros_like(p[k])
            if k not in opt_state:
                opt_state[k] = (torch.zeros_like(p[k]), torch.zeros_like(p[k]))
            keras_adam_step(p[k], g, opt_state[k][0], opt_state[k][1], step_index, lr)
        for k, vnew in ctx.new_stats.items():
            p[k].copy_(vnew)
    return float(loss.detach()), {k: float(v.detach()) for k, v in parts.items()}, outs
