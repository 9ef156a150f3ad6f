"""MI355X-native drop-in for the reference's ``pointnet/PointNet.py``.

Same public names, constructor arguments, method names and layer names as
``/root/reference/point_cloud_analysis/pointnet/PointNet.py`` (PointNet :84-376, TNet :379-490,
ConvLayer :493-594, DenseLayer :597-679, PointCloudNormalization :681-712), but the arithmetic runs in
``libpointnet_hip.so`` (hand-written HIP for gfx950) on PyTorch-ROCm tensors.

There is NO CPU compute path: calling a model whose parameters are not on a HIP device raises
``PointNetHipError``.  Construction, freeze/thaw, ``get_layer_trainability``, ``get_config`` and weight
access are host logic and work anywhere.

Deviations from the reference, on purpose ("tolerate, not replicate"):
  * ``get_config`` includes ``vanilla`` (the reference omits it, PointNet.py:354-362, so a reloaded
    vanilla model silently became non-vanilla);
  * ``freeze_shared_network`` / ``thaw_shared_network`` work for ``vanilla=True`` (the reference
    dereferences ``None``, PointNet.py:302-318);
  * ``get_last_predicted_dcm`` returns the last predicted input transform (the reference calls a method
    its TNet does not define, PointNet.py:375-376);
  * reduce_max sends the gradient to the lowest-index maximum instead of splitting it among ties
    (identical parameter gradients when ties are duplicated points).
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import _lib
from .._lib import PointNetHipError, check, current_stream, lib, pn_model_desc, pn_model_io, pn_slot_info, ptr

BN_MOMENTUM = 0.99
BN_EPS = 1e-3


def _glorot_uniform(shape, seed: Optional[int]) -> torch.Tensor:
    """keras.initializers.GlorotUniform(seed).  As in the reference every seeded layer builds its own
    initializer from the SAME seed (PointNet.py:532,627,415), so same-shape layers start identical."""
    gen = torch.Generator()
    if isinstance(seed, int):
        gen.manual_seed(seed)
    else:
        gen.seed()
    fan_in, fan_out = shape[0], shape[1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2 - 1) * lim


class _Weights:
    """Named views into the model's flat parameter buffer (layout defined by the C library)."""

    def __init__(self, desc: pn_model_desc, device):
        l = lib()
        n = l.pn_model_num_slots(C.byref(desc))
        if n <= 0:
            raise PointNetHipError("bad model descriptor: " + l.pn_last_error().decode())
        total = l.pn_model_param_floats(C.byref(desc))
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.slots: "OrderedDict[str, dict]" = OrderedDict()
        info = pn_slot_info()
        for i in range(n):
            check(l.pn_model_slot_info(C.byref(desc), i, C.byref(info)), "pn_model_slot_info")
            self.slots[info.name.decode()] = dict(offset=info.offset, rows=info.rows, cols=info.cols, kind=info.kind,
                                                  block=info.block)

    def view(self, name: str, flat: Optional[torch.Tensor] = None) -> torch.Tensor:
        s = self.slots[name]
        f = self.flat if flat is None else flat
        v = f[s["offset"]: s["offset"] + s["rows"] * s["cols"]]
        if s["kind"] in (0, 6, 7):
            return v.view(s["rows"], s["cols"])
        return v.view(s["cols"])


class _LayerBase:
    """Common freeze/thaw/name plumbing of the reference's custom layers."""

    def __init__(self, name: str):
        self.name = name
        self.trainable = True
        self._owner = None     # PointNet that owns the weights (None for a free-standing layer)
        self._prefix = None

    def is_trainable(self):
        return self.trainable

    def freeze(self):
        self.trainable = False

    def thaw(self):
        self.trainable = True

    def _w(self, suffix):
        if self._owner is None:
            own = getattr(self, "_own", None)
            if own is None or suffix not in own:
                raise PointNetHipError(f"layer {self.name} is not built")
            return own[suffix]
        return self._owner._weights.view(f"{self._prefix}.{suffix}")

    # ---- free-standing use (a layer that is not part of a PointNet owns its tensors) ----
    def _own_bn(self, width, device):
        self._own.update({"bn.gamma": torch.ones(width, device=device), "bn.beta": torch.zeros(width, device=device),
                          "bn.moving_mean": torch.zeros(width, device=device), "bn.moving_var": torch.ones(width, device=device)})

    @staticmethod
    def _device_of(x):
        _lib.require_gpu_tensor(x, "input", torch.float32)
        return x.device


class _BN:
    def __init__(self, layer, momentum):
        self._layer = layer
        self.momentum = momentum
        self.epsilon = BN_EPS
        self.trainable = True

    gamma = property(lambda self: self._layer._w("bn.gamma"))
    beta = property(lambda self: self._layer._w("bn.beta"))
    moving_mean = property(lambda self: self._layer._w("bn.moving_mean"))
    moving_variance = property(lambda self: self._layer._w("bn.moving_var"))


class ConvLayer(_LayerBase):
    """1x1 Conv2D (+BatchNormalization) (+activation) -- reference PointNet.py:493-594.  Inside a PointNet the
    layer is executed by the fused plan; free-standing use goes through ``pointcloudprocessing_amd.ops``."""

    def __init__(self, filters: int, name: str, kernel_size: tuple = (1, 1), strides: tuple = (1, 1), padding: str = 'same',
                 activation=None, apply_bn: bool = True, bn_momentum: float = 0.99, random_seed=None, **kwargs):
        super().__init__(f"{name}_convolution_layer")
        if tuple(kernel_size) != (1, 1) or tuple(strides) != (1, 1):
            raise PointNetHipError("only the 1x1 / stride-1 convolution the reference model uses is implemented")
        self.filters = filters
        self.kernel_size = tuple(kernel_size)
        self.strides = tuple(strides)
        self.padding = padding
        self.activation = activation
        self.apply_bn = apply_bn
        self.bn_momentum = bn_momentum
        self.seed = random_seed
        self.bn = _BN(self, bn_momentum) if apply_bn else None

    kernel = property(lambda self: self._w("kernel"))
    bias = property(lambda self: None if self.apply_bn else self._w("bias"))

    def freeze(self):
        self.trainable = False
        if self.apply_bn:
            self.bn.trainable = False

    def thaw(self):
        self.trainable = True
        if self.apply_bn:
            self.bn.trainable = True

    def build(self, input_shape, device="cuda"):
        """free-standing layer: Keras-style lazy weights (GlorotUniform kernel; BN gamma 1, beta 0, moving 0 / 1) -- PointNet.py:526-542"""
        cin = int(input_shape[-1])
        self._own = {"kernel": _glorot_uniform((cin, self.filters), self.seed).to(device)}
        if self.apply_bn:
            self._own_bn(self.filters, device)
        else:
            self._own["bias"] = torch.zeros(self.filters, device=device)

    def __call__(self, inputs, training: bool = False):
        """(B, N, 1, Cin) or (B, N, Cin) -> same rank with ``filters`` channels: Conv2D 1x1 -> BatchNormalization -> activation
        (PointNet.py:554-566), forward only, through the op-level C ABI (pn_conv_fwd / pn_conv3_fwd, pn_bn_finalize)."""
        from .. import ops
        if self._owner is not None:
            raise PointNetHipError("a layer inside a PointNet is executed by the model's plan; call the model")
        dev = self._device_of(inputs)
        if getattr(self, "_own", None) is None:
            self.build(inputs.shape, dev)
        shape = inputs.shape
        B, N, cin = shape[0], shape[1], shape[-1]
        x = inputs.reshape(B * N, cin).contiguous()
        w = self.kernel
        if cin == 3:
            z, part = ops.conv3_fwd(x, w, B, N)
        elif cin % 64 == 0 and self.filters % 64 == 0:
            z, part = ops.conv_fwd(_lib.operand(x, ld=cin), w, B, N, cin, self.filters, _lib.PN_PREC_BF16X3)
        else:
            raise PointNetHipError(f"ConvLayer: channel widths must be 3 or multiples of 64 (got {cin} -> {self.filters})")
        if self.apply_bn:
            batch = bool(training) and self.bn.trainable          # a frozen BN runs in inference mode (PointNet.py:585-591)
            _, _, scale, shift = ops.bn_finalize(part if batch else None, B * N, self.bn.gamma, self.bn.beta, self.bn.moving_mean,
                                                 self.bn.moving_variance, use_batch_stats=batch, update_moving=batch,
                                                 momentum=self.bn_momentum, eps=self.bn.epsilon)
            y = z * scale + shift
        else:
            y = z + self.bias
        if self.activation in ("relu", torch.relu) or getattr(self.activation, "__name__", "") == "relu":
            y = torch.relu(y)
        elif self.activation is not None:
            raise PointNetHipError(f"ConvLayer: activation {self.activation!r} is not implemented (the reference uses relu / None)")
        return y.reshape(*shape[:-1], self.filters)

    call = __call__

    def get_config(self):
        return {'filters': self.filters, 'name': self.name, 'kernel_size': self.kernel_size, 'strides': self.strides,
                'padding': self.padding, 'activation': self.activation, 'apply_bn': self.apply_bn,
                'bn_momentum': self.bn_momentum, 'random_seed': self.seed}


class DenseLayer(_LayerBase):
    """Dense (+BatchNormalization) (+activation) -- reference PointNet.py:597-679."""

    def __init__(self, units: int, name: str, activation=None, apply_bn: bool = False, bn_momentum: float = 0.99,
                 random_seed=None, **kwargs):
        super().__init__(f"{name}_dense_layer")
        self.units = units
        self.activation = activation
        self.apply_bn = apply_bn
        self.bn_momentum = bn_momentum
        self.seed = random_seed
        self.bn = _BN(self, bn_momentum) if apply_bn else None

    kernel = property(lambda self: self._w("kernel"))
    bias = property(lambda self: None if self.apply_bn else self._w("bias"))

    def freeze(self):
        self.trainable = False
        if self.apply_bn:
            self.bn.trainable = False

    def thaw(self):
        self.trainable = True
        if self.apply_bn:
            self.bn.trainable = True

    def build(self, input_shape, device="cuda"):
        cin = int(input_shape[-1])
        self._own = {"kernel": _glorot_uniform((cin, self.units), self.seed).to(device)}
        if self.apply_bn:
            self._own_bn(self.units, device)
        else:
            self._own["bias"] = torch.zeros(self.units, device=device)

    def __call__(self, inputs, training: bool = False):
        """(B, Cin) -> (B, units): Dense -> BatchNormalization -> activation (PointNet.py:642-654) in one native launch (pn_dense_layer)"""
        from .. import ops
        if self._owner is not None:
            raise PointNetHipError("a layer inside a PointNet is executed by the model's plan; call the model")
        dev = self._device_of(inputs)
        if getattr(self, "_own", None) is None:
            self.build(inputs.shape, dev)
        relu = self.activation in ("relu", torch.relu) or getattr(self.activation, "__name__", "") == "relu"
        if self.activation is not None and not relu:
            raise PointNetHipError(f"DenseLayer: activation {self.activation!r} is not implemented (the reference uses relu / None)")
        x = inputs.reshape(inputs.shape[0], -1).contiguous()
        if self.apply_bn:
            mode = 1 if (training and self.bn.trainable) else 2
            _, a, _, _ = ops.dense_layer(x, self.kernel, gamma=self.bn.gamma, beta=self.bn.beta, moving_mean=self.bn.moving_mean,
                                         moving_var=self.bn.moving_variance, bn_mode=mode, act=int(relu), momentum=self.bn_momentum,
                                         eps=self.bn.epsilon)
        else:
            _, a, _, _ = ops.dense_layer(x, self.kernel, bias=self.bias, bn_mode=0, act=int(relu))
        return a

    call = __call__

    def get_config(self):
        return {'units': self.units, 'name': self.name, 'activation': self.activation, 'apply_bn': self.apply_bn,
                'bn_momentum': self.bn_momentum, 'random_seed': self.seed}


class TNet(_LayerBase):
    """Transform network -- reference PointNet.py:379-490: three ConvLayers, max over points, two DenseLayers,
    ``@ w + b`` reshaped to (K, K)."""

    def __init__(self, name: str, add_regularization: bool = False, bn_momentum: float = 0.99,
                 layer_widths: list = [64, 128, 1024, 512, 256], random_seed=None, **kwargs):
        super().__init__(name)
        if list(layer_widths) != [64, 128, 1024, 512, 256]:
            raise PointNetHipError("only the reference's layer widths [64,128,1024,512,256] are implemented")
        self.add_regularization = add_regularization
        self.bn_momentum = bn_momentum
        self.layer_widths = list(layer_widths)
        self.seed = random_seed
        relu = "relu"
        self.conv_layer_1 = ConvLayer(filters=64, activation=relu, bn_momentum=bn_momentum, name=f"{name}_convolution_layer_1",
                                      random_seed=self.seed)
        self.conv_layer_2 = ConvLayer(filters=128, activation=relu, bn_momentum=bn_momentum, name=f"{name}_convolution_layer_2",
                                      random_seed=self.seed)
        self.conv_layer_3 = ConvLayer(filters=1024, activation=relu, bn_momentum=bn_momentum, name=f"{name}_convolution_layer_3",
                                      random_seed=self.seed)
        self.dense_layer_1 = DenseLayer(units=512, activation=relu, apply_bn=True, bn_momentum=bn_momentum,
                                        name=f"{name}_dense_layer_1", random_seed=self.seed)
        self.dense_layer_2 = DenseLayer(units=256, activation=relu, apply_bn=True, bn_momentum=bn_momentum,
                                        name=f"{name}_dense_layer_2", random_seed=self.seed)
        self._last_predicted = None

    w = property(lambda self: self._w("w"))
    b = property(lambda self: self._w("b"))

    def _sub(self):
        return [self.conv_layer_1, self.conv_layer_2, self.conv_layer_3, self.dense_layer_1, self.dense_layer_2]

    def freeze(self):
        self.trainable = False
        for l in self._sub():
            l.freeze()

    def thaw(self):
        self.trainable = True
        for l in self._sub():
            l.thaw()

    def is_trainable(self):
        return all(l.is_trainable() for l in self._sub())

    def get_last_predicted_transformation(self):
        return self._last_predicted

    def build(self, input_shape, device="cuda"):
        """free-standing T-Net: w = zeros (256, K^2), b = identity (K, K) -- PointNet.py:412-416"""
        K = int(input_shape[-1])
        self._own = {"w": torch.zeros(256, K * K, device=device), "b": torch.eye(K, device=device)}

    def __call__(self, inputs, training: bool = False):
        """(B, N, K), K in {3, 64} -> (B, K, K): three ConvLayers, max over the points, two DenseLayers, ``@ w + b`` (PointNet.py:418-454).
        Forward only, layer by layer through the op-level C ABI; inside a PointNet the fused plan runs instead."""
        from .. import ops
        if self._owner is not None:
            raise PointNetHipError("a T-Net inside a PointNet is executed by the model's plan; call the model")
        dev = self._device_of(inputs)
        if getattr(self, "_own", None) is None:
            self.build(inputs.shape, dev)
        B, N, K = inputs.shape
        x = self.conv_layer_1(inputs, training)
        x = self.conv_layer_2(x, training)
        x = self.conv_layer_3(x, training)
        g = torch.amax(x, dim=1)                                   # tf.reduce_max(X, axis=1)
        g = self.dense_layer_1(g, training)
        g = self.dense_layer_2(g, training)
        _, out, _, _ = ops.dense_layer(g, self.w, bias=self.b.reshape(-1).contiguous(), bn_mode=0, act=0)
        R = out.reshape(B, K, K)
        self._last_predicted = R
        return R

    call = __call__

    def get_config(self):
        return {'name': self.name, 'add_regularization': self.add_regularization, 'bn_momentum': self.bn_momentum,
                'layer_widths': self.layer_widths, 'random_seed': self.seed}


class PointCloudNormalization(_LayerBase):
    """Centre on the centroid, scale by the max radius (clamped at 1e-7) -- reference PointNet.py:681-712.
    Returns ``(normalized, (centroid, scale))``."""

    def __init__(self, name: str = "point_cloud_normalization", **kwargs):
        super().__init__(name)

    def __call__(self, input):
        from .. import ops
        return ops.normalize(input)

    call = __call__

    def is_trainable(self):
        return False

    def get_config(self):
        return {'name': self.name}


class _PointNetFn(torch.autograd.Function):
    """autograd bridge: forward = pn_model_forward, backward = pn_model_backward (whole model, native plan)."""

    @staticmethod
    def forward(ctx, flat, model, pc, training, fused):
        outs = model._run_forward(pc, training, fused)
        ctx.model = model
        ctx.training = training
        return outs

    @staticmethod
    def backward(ctx, d_cls, d_seg, d_R):
        model = ctx.model
        if not ctx.training:
            raise PointNetHipError("backward through a PointNet call made with training=False")
        g = model._run_backward(d_cls, d_seg, d_R)
        return g, None, None, None, None


class PointNet(torch.nn.Module):
    """PointNet with classification, per-point segmentation and input-transform outputs
    (reference PointNet.py:84-376).  ``model(pc, training=...)`` returns ``[cls (B,Ccls), seg (B,N,Cseg), R (B,3,3)]``."""

    def __init__(self, classification_output_width: int, segmentation_output_width: int, dropout_rate: float,
                 random_seed: int, debugging: bool = False, vanilla: bool = False,
                 regularize_input_transform: bool = False, regularize_feature_transform: bool = False,
                 precision: str = "bf16x3", device=None, sync_bn_world: int = 1, sync_bn_rank: int = 0, sync_bn_group=None, **kwargs):
        """``sync_bn_world`` > 1 (data parallel, numerics-parity mode): every training-mode BatchNormalization takes its statistics over
        the clouds of ALL ranks, as the reference does on its one device (PointNet.py:528,559,623,647) -- a step on W ranks of B clouds
        is then the reference's step on the B*W clouds.  The collectives go through ``torch.distributed`` (``sync_bn_group`` or the
        default group); see include/pointnet_hip.h (pn_model_io.sync_hook) for what is exchanged and engine.TrainStep for the step."""
        super().__init__()
        self._classification_output_width = classification_output_width
        self._segmentation_output_width = segmentation_output_width
        self._dropout_rate = dropout_rate
        self._random_seed = random_seed
        self._debugging = debugging
        self._vanilla = vanilla
        self._regularize_input_transform = regularize_input_transform
        self._regularize_feature_transform = regularize_feature_transform
        if precision not in _lib.PREC:
            raise PointNetHipError(f"precision must be one of {list(_lib.PREC)}")
        self._precision = precision
        self._sync_world = int(sync_bn_world) if sync_bn_world and int(sync_bn_world) > 1 else 1
        self._sync_rank = int(sync_bn_rank) if self._sync_world > 1 else 0
        self._sync_group = sync_bn_group
        self._sync_error = None
        self._custom_layers = []
        self.input_names = ['pointnet_input']
        self.output_names = ['classification_output', 'segmentation_output', 'se3']

        relu, softmax = "relu", "softmax"
        seed = self._random_seed
        self.normalize_input = PointCloudNormalization(name="input_normalization")
        self.input_transform = TNet(name='input_transform', add_regularization=regularize_input_transform,
                                    random_seed=seed) if not vanilla else None
        self.mlp_1_1 = ConvLayer(filters=64, name='s1_l1_64', activation=relu, apply_bn=True, random_seed=seed)
        self.mlp_1_2 = ConvLayer(filters=64, name='s1_l2_64', activation=relu, apply_bn=True, random_seed=seed)
        self.feature_transform = TNet(name='feature_transform', add_regularization=regularize_feature_transform,
                                      random_seed=seed) if not vanilla else None
        self.mlp_2_1 = ConvLayer(filters=64, name='s2_l1_64', activation=relu, apply_bn=True, random_seed=seed)
        self.mlp_2_2 = ConvLayer(filters=128, name='s2_l2_128', activation=relu, apply_bn=True, random_seed=seed)
        self.mlp_2_3 = ConvLayer(filters=1024, name='s2_l3_1024', activation=relu, apply_bn=True, random_seed=seed)
        self.mlp_cls_1 = DenseLayer(units=512, name='s3_l1_512', activation=relu, apply_bn=True)
        self.mlp_cls_2 = DenseLayer(units=256, name='s3_l2_256', activation=relu, apply_bn=True)
        self.mlp_cls_3 = DenseLayer(units=classification_output_width, name='output', activation=softmax)
        self.mlp_seg_1 = ConvLayer(filters=512, name='seg_l1_512', activation=relu, apply_bn=True, random_seed=seed)
        self.mlp_seg_2 = ConvLayer(filters=256, name='seg_l2_256', activation=relu, apply_bn=True, random_seed=seed)
        self.mlp_seg_3 = ConvLayer(filters=128, name='seg_l3_128', activation=relu, apply_bn=True, random_seed=seed)
        self.mlp_seg_4 = ConvLayer(filters=128, name='seg_l4_128', activation=relu, apply_bn=True, random_seed=seed)
        self.mlp_seg_5 = ConvLayer(filters=segmentation_output_width, name='seg_l5_output', activation=softmax, apply_bn=False,
                                   random_seed=seed)

        self._custom_layers.append(self.normalize_input)
        if not vanilla:
            self._custom_layers.append(self.input_transform)
        self._custom_layers += [self.mlp_1_1, self.mlp_1_2]
        if not vanilla:
            self._custom_layers.append(self.feature_transform)
        self._custom_layers += [self.mlp_2_1, self.mlp_2_2, self.mlp_2_3, self.mlp_cls_1, self.mlp_cls_2, self.mlp_cls_3,
                                self.mlp_seg_1, self.mlp_seg_2, self.mlp_seg_3, self.mlp_seg_4, self.mlp_seg_5]

        # ---- weights: one flat buffer whose layout the C library defines ----
        self._desc = pn_model_desc(ccls=classification_output_width, cseg=segmentation_output_width, vanilla=int(vanilla),
                                   reg_in=int(regularize_input_transform), reg_feat=int(regularize_feature_transform),
                                   prec=_lib.PREC[precision], dropout_rate=float(dropout_rate), bn_momentum=BN_MOMENTUM,
                                   bn_eps=BN_EPS, sync_world=self._sync_world if self._sync_world > 1 else 0)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self._weights = _Weights(self._desc, torch.device(device))
        self.params_flat = torch.nn.Parameter(self._weights.flat)
        self._weights.flat = self.params_flat.data
        self.grads_flat = torch.zeros_like(self._weights.flat)
        self._bind_layers()
        self._initialize_weights()
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._last = None
        self._built = False
        self.scalars = torch.zeros(16, dtype=torch.float32, device=self._weights.flat.device)
        self._fused_targets = None

    # ------------------------------------------------------------------ construction helpers
    def _bind_layers(self):
        def bind(layer, prefix):
            layer._owner, layer._prefix = self, prefix
        if not self._vanilla:
            for tn, pref in ((self.input_transform, "input_transform"), (self.feature_transform, "feature_transform")):
                bind(tn, pref)
                bind(tn.conv_layer_1, f"{pref}.conv1")
                bind(tn.conv_layer_2, f"{pref}.conv2")
                bind(tn.conv_layer_3, f"{pref}.conv3")
                bind(tn.dense_layer_1, f"{pref}.dense1")
                bind(tn.dense_layer_2, f"{pref}.dense2")
        for attr in ("mlp_1_1", "mlp_1_2", "mlp_2_1", "mlp_2_2", "mlp_2_3", "mlp_cls_1", "mlp_cls_2", "mlp_cls_3", "mlp_seg_1",
                     "mlp_seg_2", "mlp_seg_3", "mlp_seg_4", "mlp_seg_5"):
            bind(getattr(self, attr), attr)

    def _initialize_weights(self):
        W = self._weights
        seeded = {"mlp_cls_1", "mlp_cls_2", "mlp_cls_3"}       # the classification DenseLayers are unseeded (PointNet.py:130-134)
        with torch.no_grad():
            for name, s in W.slots.items():
                v = W.view(name)
                k = s["kind"]
                if k == 0 or k == 6:
                    seed = None if name.split(".")[0] in seeded else self._random_seed
                    v.copy_(_glorot_uniform((s["rows"], s["cols"]), seed))
                elif k == 1 or k == 4:
                    v.fill_(1.0)                                # gamma, moving_variance
                elif k == 7:
                    v.copy_(torch.eye(s["rows"]))               # T-Net b: 'identity' (PointNet.py:416)
                else:
                    v.zero_()                                   # beta, moving_mean, bias

    def build(self, input_shape):
        """Keras-style build; weights already exist, this only validates the shape (PointNet.py:161-195)."""
        if len(input_shape) != 3 or input_shape[2] != 3:
            raise PointNetHipError(f"PointNet expects (None, N, 3) inputs, got {input_shape}")
        self._built = True

    # ------------------------------------------------------------------ weights access
    def named_weights(self) -> "OrderedDict[str, torch.Tensor]":
        """name -> view into the flat buffer.  Names: '<block>[.<sub>].kernel|bn.gamma|bn.beta|bn.moving_mean|
        bn.moving_var|bias', '<tnet>.w', '<tnet>.b'; kernels are (Cin, Cout) like Keras."""
        return OrderedDict((n, self._weights.view(n)) for n in self._weights.slots)

    def named_grads(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((n, self._weights.view(n, self.grads_flat)) for n in self._weights.slots)

    def set_weights(self, weights: Dict[str, torch.Tensor]):
        with torch.no_grad():
            for n, t in weights.items():
                self._weights.view(n).copy_(torch.as_tensor(t, dtype=torch.float32).reshape(self._weights.view(n).shape))

    def count_params(self):
        tr = sum(s["rows"] * s["cols"] for s in self._weights.slots.values() if s["kind"] not in (3, 4))
        nt = sum(s["rows"] * s["cols"] for s in self._weights.slots.values() if s["kind"] in (3, 4))
        return tr, nt

    def trainable_mask(self) -> torch.Tensor:
        """1.0 where a flat-buffer element belongs to a trainable weight of a trainable block."""
        flags = self._block_flags()
        m = torch.zeros_like(self._weights.flat)
        for s in self._weights.slots.values():
            if s["kind"] not in (3, 4) and flags[s["block"]]:
                m[s["offset"]: s["offset"] + s["rows"] * s["cols"]] = 1.0
        return m

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._weights.flat = self.params_flat.data
        self.grads_flat = fn(self.grads_flat)
        self.scalars = fn(self.scalars)
        self._ws = {}
        return r

    # ------------------------------------------------------------------ trainability (PointNet.py:294-349)
    def _block_layers(self):
        return [self.input_transform, self.mlp_1_1, self.mlp_1_2, self.feature_transform, self.mlp_2_1, self.mlp_2_2, self.mlp_2_3,
                self.mlp_cls_1, self.mlp_cls_2, self.mlp_cls_3, self.mlp_seg_1, self.mlp_seg_2, self.mlp_seg_3, self.mlp_seg_4,
                self.mlp_seg_5]

    def _block_flags(self) -> List[int]:
        return [int(l.is_trainable()) if l is not None else 0 for l in self._block_layers()]

    def freeze_input_transform(self) -> None:
        if self._vanilla:
            print("PointNet:  No input transorm available to freeze.")
        else:
            self.input_transform.freeze()

    def thaw_input_transform(self) -> None:
        if self._vanilla:
            print("PointNet:  No input transorm available to thaw.")
        else:
            self.input_transform.thaw()

    def _shared(self):
        return [l for l in (self.input_transform, self.mlp_1_1, self.mlp_1_2, self.feature_transform, self.mlp_2_1, self.mlp_2_2,
                            self.mlp_2_3) if l is not None]

    def freeze_shared_network(self) -> None:
        for l in self._shared():
            l.freeze()

    def thaw_shared_network(self) -> None:
        for l in self._shared():
            l.thaw()

    def freeze_segmentation_head(self) -> None:
        for l in (self.mlp_seg_1, self.mlp_seg_2, self.mlp_seg_3, self.mlp_seg_4, self.mlp_seg_5):
            l.freeze()

    def thaw_segmentation_head(self) -> None:
        for l in (self.mlp_seg_1, self.mlp_seg_2, self.mlp_seg_3, self.mlp_seg_4, self.mlp_seg_5):
            l.thaw()

    def freeze_classification_head(self) -> None:
        for l in (self.mlp_cls_1, self.mlp_cls_2, self.mlp_cls_3):
            l.freeze()

    def thaw_classification_head(self) -> None:
        for l in (self.mlp_cls_1, self.mlp_cls_2, self.mlp_cls_3):
            l.thaw()

    def get_layer_trainability(self) -> dict:
        return {layer_.name: layer_.is_trainable() for layer_ in self._custom_layers}

    def get_config(self):
        return {'classification_output_width': self._classification_output_width,
                'segmentation_output_width': self._segmentation_output_width, 'dropout_rate': self._dropout_rate,
                'random_seed': self._random_seed, 'debugging': self._debugging, 'vanilla': self._vanilla,
                'regularize_input_transform': self._regularize_input_transform,
                'regularize_feature_transform': self._regularize_feature_transform, 'precision': self._precision}

    @classmethod
    def from_config(cls, config):
        return cls(**config)

    def get_last_predicted_dcm(self):
        return None if self._last is None else self._last[2]

    # ------------------------------------------------------------------ execution
    def _workspace(self, B, N, training):
        key = (B, N, bool(training))
        ws = self._ws.get(key)
        if ws is None:
            nbytes = lib().pn_model_workspace_bytes(C.byref(self._desc), B, N, int(training))
            if nbytes == 0:
                raise PointNetHipError("pn_model_workspace_bytes failed: " + lib().pn_last_error().decode())
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self._weights.flat.device)
            self._ws[key] = ws
        return ws

    @property
    def activation_dtype(self):
        """storage type of the per-point layer-boundary tensors of the plan (workspace entries *.Z, *.dy, *.D, X64, dX64, tmpA12)"""
        return torch.bfloat16 if self._desc.prec & _lib.PN_STORE_BF16 else torch.float32

    def workspace_tensor(self, name: str, B: int, N: int, training: bool, dtype=torch.float32) -> torch.Tensor:
        """View of a named intermediate of the last call of that shape (test introspection)."""
        off, nb = C.c_int64(), C.c_int64()
        check(lib().pn_model_ws_lookup(C.byref(self._desc), B, N, int(training), name.encode(), C.byref(off), C.byref(nb)),
              "pn_model_ws_lookup")
        return self._workspace(B, N, training)[off.value: off.value + nb.value].view(dtype)

    def _io(self, pc, training, fused):
        B, N, _ = pc.shape
        dev = pc.device
        ws = self._workspace(B, N, training)
        io = pn_model_io()
        io.pc = pc.data_ptr()
        io.B, io.N = B, N
        io.params = self._weights.flat.data_ptr()
        io.grads = self.grads_flat.data_ptr() if training else None
        self._flags_c = (C.c_uint8 * _lib.PN_NUM_BLOCKS)(*self._block_flags())
        io.trainable = C.cast(self._flags_c, C.c_void_p)
        io.training = int(training)
        io.workspace = ws.data_ptr()
        io.workspace_bytes = ws.numel()
        if self._debugging or getattr(self, "keep_activations", False):      # check_numerics reads every layer's stored output
            io.flags = _lib.PN_IO_KEEP_ACTIVATIONS
        io.scalars = self.scalars.data_ptr()
        aux = getattr(self, "_aux_stream", None)       # engine.TrainStep: parameter gradients on a second stream
        io.aux_stream = aux.cuda_stream if (aux is not None and training) else None
        keep = None
        W = self._sync_world if training else 1
        if W > 1:
            io.sync_rank = self._sync_rank
            io.sync_hook = C.cast(self._sync_hook_c(), C.c_void_p)
            self._sync_ws = ws
        if training and self._dropout_rate > 0:
            if fused is not None and fused.get("keep") is not None:
                keep = fused["keep"]
                if keep[0].numel() != B * W * 512 or keep[1].numel() != B * W * 256:
                    raise PointNetHipError(f"dropout keep masks need {B * W} rows (synchronised BatchNormalization: the rows of all ranks)")
            else:
                if W > 1:
                    raise PointNetHipError("synchronised BatchNormalization needs the dropout keep masks of all ranks' rows (engine.TrainStep draws them)")
                keep = ((torch.rand(B, 512, device=dev) >= self._dropout_rate).to(torch.uint8),
                        (torch.rand(B, 256, device=dev) >= self._dropout_rate).to(torch.uint8))
            io.keep1, io.keep2 = keep[0].data_ptr(), keep[1].data_ptr()
            rng = None if fused is None else fused.get("dropout_rng")
            if rng is not None:        # (seed, device uint32 step counter): the forward's first launch draws the masks into `keep`
                io.dropout_seed, io.dropout_step = int(rng[0]), rng[1].data_ptr()
        if fused is not None and fused.get("zero_grads_in_forward"):
            io.zero_grads_in_forward = 1
        if fused is not None and fused.get("labels_cls") is not None:
            _lib.require_gpu_tensor(fused["labels_cls"], "labels_cls", torch.int32)
            _lib.require_gpu_tensor(fused["labels_seg"], "labels_seg", torch.int32)
            _lib.require_gpu_tensor(fused["se3"], "se3", torch.float32)
            if fused["labels_cls"].numel() != B or fused["labels_seg"].numel() != B * N or fused["se3"].numel() != B * 9:
                raise PointNetHipError("fused loss targets have the wrong number of elements")
            io.labels_cls = fused["labels_cls"].data_ptr()
            io.labels_seg = fused["labels_seg"].data_ptr()
            io.se3 = fused["se3"].data_ptr()
            lw = fused["loss_weights"]
            # synchronised BatchNormalization: every rank seeds the gradient of the GLOBAL mean loss (pn_model_io.sync_hook)
            io.loss_weights[0], io.loss_weights[1], io.loss_weights[2] = float(lw[0]) / W, float(lw[1]) / W, float(lw[2]) / W
        return io, keep

    # ------------------------------------------------------------------ synchronised BatchNormalization
    def _sync_hook_c(self):
        """the C callback the native plan calls between two of its launches (pn_model_io.sync_hook): op 0 all-reduce (sum), op 1 all-gather,
        on ranges of the workspace, through torch.distributed on the current stream"""
        if getattr(self, "_sync_cb", None) is None:
            import torch.distributed as dist

            def hook(ctx, op, src, dst, n, dtype, stream):
                try:
                    ws = self._sync_ws
                    base, es = ws.data_ptr(), (4 if dtype == 0 else 8)
                    tdt = torch.float32 if dtype == 0 else torch.int64

                    def view(ptr_, count):
                        off = ptr_ - base
                        if off < 0 or off + count * es > ws.numel():
                            raise PointNetHipError("sync_hook: range outside the workspace")
                        return ws[off: off + count * es].view(tdt)
                    W = self._sync_world
                    if op == 0:
                        d = view(dst, n)
                        if src != dst:
                            d.copy_(view(src, n))
                        dist.all_reduce(d, group=self._sync_group)
                    else:
                        out, mine = view(dst, n * W), view(src, n).clone()
                        dist.all_gather([out[i * n:(i + 1) * n] for i in range(W)], mine, group=self._sync_group)
                    return 0
                except Exception as e:      # never let an exception cross the C frame
                    self._sync_error = f"{type(e).__name__}: {e}"
                    return 1
            self._sync_cb = _lib.SYNC_HOOK(hook)
        return self._sync_cb

    def replicated_grad_mask(self) -> torch.Tensor:
        """synchronised BatchNormalization: 0.0 where every rank computes a gradient slot IN FULL (all bn.gamma / bn.beta -- formed from
        the sums over all ranks; the per-cloud dense layers' kernels and bias and the T-Nets' w / b -- run on all ranks' rows), 1.0 where
        a rank holds its clouds' share.  Every rank but one multiplies its gradients by this before they are summed."""
        m = torch.ones_like(self._weights.flat)
        dense_blocks = ("mlp_cls_1", "mlp_cls_2", "mlp_cls_3")
        for name, s in self._weights.slots.items():
            k = s["kind"]
            rep = k in (1, 2, 6, 7) or (k in (0, 5) and (name.split(".")[0] in dense_blocks or ".dense" in name))
            if rep:
                m[s["offset"]: s["offset"] + s["rows"] * s["cols"]] = 0.0
        return m

    def _run_forward(self, pc, training, fused):
        _lib.require_gpu_tensor(pc, "pc", torch.float32)
        if not self._weights.flat.is_cuda:
            raise PointNetHipError("PointNet parameters are not on a HIP device: there is no CPU compute path")
        if pc.dim() != 3 or pc.shape[2] != 3:
            raise PointNetHipError(f"PointNet expects (B, N, 3) inputs, got {tuple(pc.shape)}")
        B, N, _ = pc.shape
        dev = pc.device
        io, keep = self._io(pc, training, fused)
        cls = torch.empty(B, self._classification_output_width, device=dev, dtype=torch.float32)
        seg = torch.empty(B, N, self._segmentation_output_width, device=dev, dtype=torch.float32)
        R = torch.empty(B, 3, 3, device=dev, dtype=torch.float32)
        io.out_cls, io.out_seg, io.out_R = cls.data_ptr(), seg.data_ptr(), R.data_ptr()
        rc = lib().pn_model_forward(C.byref(self._desc), C.byref(io), current_stream())
        if rc != 0 and self._sync_error:
            raise PointNetHipError(f"pn_model_forward: collective failed inside sync_hook: {self._sync_error}")
        check(rc, "pn_model_forward")
        if self._debugging:
            self._check_numerics(pc, cls, seg, training)
        self._last = (cls, seg, R)
        self._last_call = dict(pc=pc, training=training, fused=fused, keep=keep, io=io)
        if not self._vanilla:
            self.input_transform._last_predicted = R
        return cls, seg, R

    def _check_numerics(self, pc, cls, seg, training):
        """``debugging: true`` (reference pointnet_train.py:112): PointNet.call wraps its input and every layer output in
        tf.debugging.check_numerics (PointNet.py:199-288).  The plan stores each ConvLayer's pre-BN output and BN coefficients, the
        pooled features and the head activations; they are scanned on the device (pn_count_nonfinite) in the reference's order and the
        first site holding a NaN / Inf raises with the reference's own message.  One host read per call: debug mode only, never inside
        a captured step."""
        B, N, _ = pc.shape
        def t(name, dtype=torch.float32):
            return self.workspace_tensor(name, B, N, training, dtype)
        act = self.activation_dtype
        conv = lambda wn: [t(wn + ".Z", act), t(wn + ".scale"), t(wn + ".shift")]          # noqa: E731
        sites = [("Input point cloud contains nan values", "pointnet_input", [pc])]
        if not self._vanilla:
            sites.append(("Input transform produced nan values.", self.input_transform.name, [t("iT.R"), t("Weff1")]))
        sites += [("mlp_1_1 produced nan values.", self.mlp_1_1.name, conv("m11")),
                  ("mlp_1_2 produced nan values.", self.mlp_1_2.name, conv("m12"))]
        if not self._vanilla:
            sites.append(("Feature transform produced nan values.", self.feature_transform.name, [t("fT.R"), t("X64", act)]))
        sites += [("mlp_2_1 produced nan values.", self.mlp_2_1.name, conv("m21")),
                  ("mlp_2_2 produced nan values.", self.mlp_2_2.name, conv("m22")),
                  ("mlp_2_3 produced nan values.", self.mlp_2_3.name, [t("mm23.zstar"), t("mm23.g")]),
                  ("mlp_cls_1 produced nan values.", self.mlp_cls_1.name, [t("c1.a")]),
                  ("mlp_cls_2 produced nan values.", self.mlp_cls_2.name, [t("c2.a")]),
                  ("mlp_cls_3 produced nan values.", self.mlp_cls_3.name, [cls]),
                  ("mlp_3_1 produced nan values.", self.mlp_seg_1.name, conv("s1")),
                  ("mlp_3_2 produced nan values.", self.mlp_seg_2.name, conv("s2")),
                  ("mlp_3_3 produced nan values.", self.mlp_seg_3.name, conv("s3")),
                  ("mlp_3_4 produced nan values.", self.mlp_seg_4.name, conv("s4")),
                  ("mlp_3_5 produced nan values.", self.mlp_seg_5.name, [seg])]
        counts = torch.zeros(len(sites), dtype=torch.int32, device=pc.device)
        for i, (_, _, tensors) in enumerate(sites):
            for x in tensors:
                check(lib().pn_count_nonfinite(ptr(x), x.numel(), int(x.dtype == torch.bfloat16), ptr(counts[i:i + 1]), current_stream()),
                      "pn_count_nonfinite")
        bad = counts.cpu().tolist()
        for (msg, layer, _), n in zip(sites, bad):
            if n:
                raise PointNetHipError(f"check_numerics: {msg} ({n} non-finite values at layer '{layer}')")

    def _run_backward(self, d_cls, d_seg, d_R, phase: int = 0):
        """phase 0: the whole backward pass; 1 / 2: its two halves (pn_model_io.bwd_phase) for gradient-bucket overlap"""
        lc = self._last_call
        io = lc["io"]
        io.bwd_phase = int(phase)
        def c(t):
            return None if t is None else t.contiguous()
        d_cls, d_seg, d_R = c(d_cls), c(d_seg), c(d_R)
        rc = lib().pn_model_backward(C.byref(self._desc), C.byref(io), ptr(d_cls), ptr(d_seg), ptr(d_R), current_stream())
        if rc != 0 and self._sync_error:
            raise PointNetHipError(f"pn_model_backward: collective failed inside sync_hook: {self._sync_error}")
        check(rc, "pn_model_backward")
        return self.grads_flat

    def forward(self, pc, training: Optional[bool] = None):
        training = self.training if training is None else bool(training)
        if training and torch.is_grad_enabled():
            outs = _PointNetFn.apply(self.params_flat, self, pc, True, None)
        else:
            outs = self._run_forward(pc, training, None)
        return [outs[0], outs[1], outs[2]]

    call = forward

    def predict(self, pc):
        """inference entry: ``(class index (B,), part index per point (B, N), R (B,3,3))`` -- the arg-max of the two softmax
        outputs (first maximum, np.argmax order), taken on the device (pn_argmax_rows).  Indices are int32."""
        from .. import ops
        cls, seg, R = self._run_forward(pc, False, None)
        return ops.argmax_rows(cls), ops.argmax_rows(seg), R

    def grad_extent(self):
        """(lo, hi): the smallest range of ``grads_flat`` (floats) that holds every gradient of a trainable block.  Outside it the
        gradient is identically zero on every rank and every step (frozen blocks, PointNet.py:294-349): the data-parallel all-reduce
        and the optimizer skip it -- with fresh optimizer state (the trainer builds one per profile stage, pointnet_train.py:310-319)
        Adam's update of a zero-gradient element is exactly zero, so nothing changes but the bytes moved."""
        flags = self._block_flags()
        lo, hi = None, 0
        for s in self._weights.slots.values():
            if s["kind"] not in (3, 4) and flags[s["block"]]:
                o, e = int(s["offset"]), int(s["offset"]) + int(s["rows"]) * int(s["cols"])
                lo = o if lo is None else min(lo, o)
                hi = max(hi, e)
        if lo is None:
            return 0, 0
        return lo, (hi + 63) // 64 * 64 if (hi + 63) // 64 * 64 <= self.grads_flat.numel() else hi

    def grad_bucket_boundary(self) -> int:
        """offset (floats) in ``grads_flat`` from which every slot is final after backward phase 1"""
        for n, s in self._weights.slots.items():
            if n.startswith("feature_transform." if not self._vanilla else "mlp_2_1."):
                return int(s["offset"])
        raise PointNetHipError("parameter layout has no feature_transform / mlp_2_1 slot")

    def fused_loss_step(self, pc, labels_cls, labels_seg, se3, loss_weights, keep=None, backward_phase: int = 0, dropout_rng=None):
        """forward + the three keras losses of pointnet_train.py:334-345 + backward, all native.  Leaves the
        gradients in ``grads_flat`` and the loss / metric sums in ``self.scalars`` (see pn_model_io).
        ``keep``: the two dropout keep masks (uint8 tensors) -- inputs, or, with ``dropout_rng=(seed, step_counter_tensor)``, buffers
        the forward pass fills from its counter-based generator (pn_model_io.dropout_step)."""
        fused = dict(labels_cls=labels_cls, labels_seg=labels_seg.reshape(-1), se3=se3, loss_weights=loss_weights, keep=keep,
                     dropout_rng=dropout_rng, zero_grads_in_forward=True)      # the backward pass follows: one clearing launch less
        outs = self._run_forward(pc, True, fused)
        self._run_backward(None, None, None, backward_phase)
        return outs
