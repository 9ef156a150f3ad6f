"""ONNX (opset 13) export of the PointNet inference graph, and the matching weight reader.

The reference exports every trained profile with ``tf2onnx.convert.from_keras(model, input_signature=[(None, input_width, 3)],
opset=13)`` (pointnet_train.py:238-248).  ``onnx`` / ``tf2onnx`` are not importable here, but an ONNX file is a protobuf message and
``onnx.proto`` is small: the messages are written field by field with the same wire helpers the tf.train.Example writer uses
(pointcloud/PointCloudSet.py).  Field numbers (onnx.proto3, IR version 7 = opset 13):

    ModelProto      ir_version 1, producer_name 2, producer_version 3, graph 7, opset_import 8 {domain 1, version 2},
                    metadata_props 14 {key 1, value 2}  (the model's constructor arguments, so a file restores as a checkpoint)
    GraphProto      node 1, name 2, initializer 5, input 11, output 12
    NodeProto       input 1, output 2, name 3, op_type 4, attribute 5
    AttributeProto  name 1, f 2, i 3, ints 8, type 20 (FLOAT 1, INT 2, INTS 7)
    TensorProto     dims 1, data_type 2 (FLOAT 1, INT64 7), name 8, raw_data 9
    ValueInfoProto  name 1, type 2 { tensor_type 1 { elem_type 1, shape 2 { dim 1 { dim_value 1 | dim_param 2 } } } }

Graph (same dataflow as PointNet.call, PointNet.py:197-292, inference mode; names of inputs / outputs as PointNet.py:113-114):
the 16 ConvLayers are ``Conv`` nodes with 1x1 kernels on (B, C, N, 1) tensors followed by ``BatchNormalization`` (epsilon 1e-3) and
``Relu``; the 7 DenseLayers are ``MatMul`` (+ BatchNormalization + Relu); the T-Net tails ``x @ w`` and the two transform applications
``tf.matmul(pc, R)`` / ``tf.matmul(X, R_64)`` are batched ``MatMul``; the three ``tf.reduce_max`` and the normalisation's max are
``ReduceMax``; tile + concat in front of seg_l1 are ``Expand`` + ``Concat`` as the reference graph has them.  That reproduces the op
histogram of the reference's own tf2onnx dump (16 Conv2D, 7 MatMul + 4 BatchMatMulV2, 4 Max, 21 Relu, 2 Softmax;
tests/golden/ref_tf2onnx_graph_f15.json), which tests/test_cpu_onnx.py checks.  Unlike tf2onnx the BatchNormalization nodes are kept
(not folded into the kernels), so every raw parameter survives in the file under its canonical name and ``read_onnx_weights``
restores a checkpoint from it.

NUMERIC PARITY UNPINNED: no onnxruntime here and the reference's own .onnx blobs are stripped; the tests evaluate the written file
with an independent NumPy interpreter against the CPU oracle.
"""
from __future__ import annotations

import json
from typing import Dict, List, Sequence

import numpy as np

from .pointcloud.PointCloudSet import _fields, _ld, _varint

BN_EPS = 1e-3

FLOAT, INT64 = 1, 7
A_FLOAT, A_INT, A_INTS = 1, 2, 7


def _vi(field: int, value: int) -> bytes:               # varint field
    return _varint((field << 3) | 0) + _varint(value)


def _str(field: int, s: str) -> bytes:
    return _ld(field, s.encode())


def tensor_proto(name: str, arr: np.ndarray) -> bytes:
    arr = np.ascontiguousarray(arr)
    if arr.dtype == np.float32:
        dt = FLOAT
    elif arr.dtype == np.int64:
        dt = INT64
    else:
        raise ValueError(f"unsupported initializer dtype {arr.dtype}")
    return b"".join(_vi(1, int(d)) for d in arr.shape) + _vi(2, dt) + _str(8, name) + _ld(9, arr.astype(arr.dtype.newbyteorder("<")).tobytes())


def _attr(name: str, value) -> bytes:
    if isinstance(value, float):
        body = _str(1, name) + _varint((2 << 3) | 5) + np.float32(value).tobytes() + _vi(20, A_FLOAT)
    elif isinstance(value, int):
        body = _str(1, name) + _vi(3, value) + _vi(20, A_INT)
    else:
        body = _str(1, name) + b"".join(_vi(8, int(v)) for v in value) + _vi(20, A_INTS)
    return _ld(5, body)


def node_proto(op: str, inputs: Sequence[str], outputs: Sequence[str], name: str, **attrs) -> bytes:
    body = b"".join(_str(1, i) for i in inputs) + b"".join(_str(2, o) for o in outputs) + _str(3, name) + _str(4, op)
    body += b"".join(_attr(k, v) for k, v in attrs.items())
    return body


def value_info(name: str, shape: Sequence) -> bytes:
    dims = b"".join(_ld(1, _str(2, d) if isinstance(d, str) else _vi(1, int(d))) for d in shape)
    return _str(1, name) + _ld(2, _ld(1, _vi(1, FLOAT) + _ld(2, dims)))


class _Graph:
    def __init__(self):
        self.nodes: List[bytes] = []
        self.inits: List[bytes] = []
        self.ops: List[str] = []
        self._n = 0

    def const(self, name: str, arr) -> str:
        self.inits.append(tensor_proto(name, np.asarray(arr)))
        return name

    def op(self, op: str, inputs: Sequence[str], name: str, n_out: int = 1, **attrs):
        outs = [f"{name}:{i}" for i in range(n_out)]
        self.nodes.append(node_proto(op, inputs, outs, name, **attrs))
        self.ops.append(op)
        return outs[0] if n_out == 1 else outs


def build_model_bytes(weights: Dict[str, np.ndarray], input_width: int, vanilla: bool = False, config: dict = None) -> bytes:
    """`weights`: canonical name -> array ('<block>[.<sub>].kernel|bn.gamma|bn.beta|bn.moving_mean|bn.moving_var|bias', '<tnet>.w', '.b')."""
    W = {k: np.asarray(v, dtype=np.float32) for k, v in weights.items()}
    g = _Graph()
    N = int(input_width)
    i64 = lambda *v: np.asarray(v, dtype=np.int64)                                  # noqa: E731

    def bn(x, prefix, name):
        return g.op("BatchNormalization", [x, g.const(f"{prefix}.bn.gamma", W[f"{prefix}.bn.gamma"]), g.const(f"{prefix}.bn.beta", W[f"{prefix}.bn.beta"]),
                                           g.const(f"{prefix}.bn.moving_mean", W[f"{prefix}.bn.moving_mean"]),
                                           g.const(f"{prefix}.bn.moving_var", W[f"{prefix}.bn.moving_var"])], name, epsilon=float(BN_EPS))

    def conv(x, prefix, rows=None, relu=True):
        """ConvLayer on a (B, Cin, N, 1) tensor: Conv 1x1 -> BatchNormalization -> Relu (or + bias when the layer has no BN)"""
        k = W[f"{prefix}.kernel"]                                                   # Keras layout (Cin, Cout)
        kw = g.const(f"{prefix}.kernel", np.ascontiguousarray(k.T).reshape(k.shape[1], k.shape[0], 1, 1))
        if f"{prefix}.bn.gamma" in W:
            y = g.op("Conv", [x, kw], f"{prefix}/Conv", kernel_shape=[1, 1])
            y = bn(y, prefix, f"{prefix}/BatchNormalization")
        else:
            y = g.op("Conv", [x, kw, g.const(f"{prefix}.bias", W[f"{prefix}.bias"])], f"{prefix}/Conv", kernel_shape=[1, 1])
        return g.op("Relu", [y], f"{prefix}/Relu") if relu else y

    def dense(x, prefix, relu=True):
        y = g.op("MatMul", [x, g.const(f"{prefix}.kernel", W[f"{prefix}.kernel"])], f"{prefix}/MatMul")
        if f"{prefix}.bn.gamma" in W:
            y = bn(y, prefix, f"{prefix}/BatchNormalization")
        else:
            y = g.op("Add", [y, g.const(f"{prefix}.bias", W[f"{prefix}.bias"])], f"{prefix}/BiasAdd")
        return g.op("Relu", [y], f"{prefix}/Relu") if relu else y

    def to_conv_layout(x, name):      # (B, N, C) -> (B, C, N, 1)
        t = g.op("Transpose", [x], f"{name}/to_nchw/Transpose", perm=[0, 2, 1])
        return g.op("Reshape", [t, g.const(f"{name}/to_nchw/shape", i64(0, 0, 0, 1))], f"{name}/to_nchw/Reshape")

    def from_conv_layout(x, name):    # (B, C, N, 1) -> (B, N, C)
        t = g.op("Reshape", [x, g.const(f"{name}/to_nwc/shape", i64(0, 0, -1))], f"{name}/to_nwc/Reshape")
        return g.op("Transpose", [t], f"{name}/to_nwc/Transpose", perm=[0, 2, 1])

    def tnet(x_conv, name, K):
        h = conv(x_conv, f"{name}.conv1")
        h = conv(h, f"{name}.conv2")
        h = conv(h, f"{name}.conv3")
        gmax = g.op("ReduceMax", [h], f"{name}/Max", axes=[2, 3], keepdims=0)                    # tf.reduce_max(X, axis=1)
        h = dense(gmax, f"{name}.dense1")
        h = dense(h, f"{name}.dense2")
        h = g.op("Reshape", [h, g.const(f"{name}/expand/shape", i64(0, 1, 256))], f"{name}/ExpandDims")
        t = g.op("MatMul", [h, g.const(f"{name}.w", W[f"{name}.w"])], f"{name}/MatMul")           # BatchMatMulV2 in the TF graph
        t = g.op("Reshape", [t, g.const(f"{name}/reshape/shape", i64(-1, K, K))], f"{name}/Reshape")
        return g.op("Add", [t, g.const(f"{name}.b", W[f"{name}.b"])], f"{name}/add")

    # ---- PointCloudNormalization (PointNet.py:691-706)
    x = "pointnet_input"
    cen = g.op("ReduceMean", [x], "input_normalization/Mean", axes=[1], keepdims=1)
    cx = g.op("Sub", [x, cen], "input_normalization/Sub")
    sq = g.op("Mul", [cx, cx], "input_normalization/Square")
    ss = g.op("ReduceSum", [sq, g.const("input_normalization/Sum/axes", i64(2))], "input_normalization/Sum", keepdims=0)
    dist = g.op("Sqrt", [ss], "input_normalization/Sqrt")
    md = g.op("ReduceMax", [dist], "input_normalization/Max", axes=[1], keepdims=1)
    md = g.op("Reshape", [md, g.const("input_normalization/scale/shape", i64(0, 1, 1))], "input_normalization/ExpandDims")
    sc = g.op("Max", [md, g.const("input_normalization/min_scale", np.asarray(1e-7, dtype=np.float32))], "input_normalization/Maximum")
    pcn = g.op("Div", [cx, sc], "input_normalization/RealDiv")

    if not vanilla:
        R = tnet(to_conv_layout(pcn, "input_transform/in"), "input_transform", 3)
        X = g.op("MatMul", [pcn, R], "MatMul")                                                    # tf.matmul(pc, R), PointNet.py:207
    else:
        X = pcn
        eye = g.const("se3/identity", np.eye(3, dtype=np.float32).reshape(1, 3, 3))
        zero = g.op("Mul", [g.op("ReduceMean", [x], "se3/zero/Mean", axes=[1, 2], keepdims=0), g.const("se3/zero/0", np.zeros((), np.float32))], "se3/zero/Mul")
        R = g.op("Add", [g.op("Reshape", [zero, g.const("se3/zero/shape", i64(-1, 1, 1))], "se3/zero/Reshape"), eye], "se3/eye")
    h = conv(to_conv_layout(X, "mlp_1/in"), "mlp_1_1")
    h = conv(h, "mlp_1_2")
    if not vanilla:
        R64 = tnet(h, "feature_transform", 64)
        X64 = g.op("MatMul", [from_conv_layout(h, "feature_transform/x"), R64], "MatMul_1")      # tf.matmul(X, R_64), PointNet.py:228
        x64c = to_conv_layout(X64, "mlp_2/in")
    else:
        x64c = h
    h = conv(x64c, "mlp_2_1")
    h = conv(h, "mlp_2_2")
    h = conv(h, "mlp_2_3")
    gf = g.op("ReduceMax", [h], "Max", axes=[2, 3], keepdims=0)                                    # global features (B, 1024)

    c = dense(gf, "mlp_cls_1")
    c = dense(c, "mlp_cls_2")
    c = dense(c, "mlp_cls_3", relu=False)
    cls = g.op("Softmax", [c], "output_dense_layer/Softmax", axis=-1)

    t = g.op("Reshape", [gf, g.const("Tile/in_shape", i64(0, 1024, 1, 1))], "ExpandDims_global")
    t = g.op("Expand", [t, g.const("Tile/multiples", i64(1, 1024, N, 1))], "Tile")
    s = g.op("Concat", [x64c, t], "concat", axis=1)                                                # (B, 1088, N, 1)
    s = conv(s, "mlp_seg_1")
    s = conv(s, "mlp_seg_2")
    s = conv(s, "mlp_seg_3")
    s = conv(s, "mlp_seg_4")
    s = conv(s, "mlp_seg_5", relu=False)
    s = from_conv_layout(s, "seg_out")
    seg = g.op("Softmax", [s], "seg_l5_output_convolution_layer/Softmax", axis=-1)

    outs = [g.op("Identity", [cls], "classification_output_id"), g.op("Identity", [seg], "segmentation_output_id"), g.op("Identity", [R], "se3_id")]
    # graph outputs carry the reference's output names (PointNet.py:114)
    names = ["classification_output", "segmentation_output", "se3"]
    final_nodes = g.nodes[:-3]
    for (src, nm, op_name) in zip([cls, seg, R], names, ["Identity", "Identity_2", "Identity_1"]):
        final_nodes.append(node_proto("Identity", [src], [nm], op_name))
    ccls = int(W["mlp_cls_3.kernel"].shape[1])
    cseg = int(W["mlp_seg_5.kernel"].shape[1])
    graph = b"".join(_ld(1, n) for n in final_nodes) + _str(2, "point_net") + b"".join(_ld(5, t_) for t_ in g.inits)
    graph += _ld(11, value_info("pointnet_input", ["unk__batch", N, 3]))
    graph += _ld(12, value_info(names[0], ["unk__batch", ccls])) + _ld(12, value_info(names[1], ["unk__batch", N, cseg]))
    graph += _ld(12, value_info(names[2], ["unk__batch", 3, 3]))
    model = _vi(1, 7) + _str(2, "pointcloudprocessing_amd") + _str(3, "1") + _ld(7, graph) + _ld(8, _str(1, "") + _vi(2, 13))
    # PointNet.get_config() as metadata_props: what the weights alone do not say (dropout rate, seed, regularisers, ...)
    for k, v in sorted((config or {}).items()):
        model += _ld(14, _str(1, "pointnet." + str(k)) + _str(2, json.dumps(v)))
    return model


def export_onnx(weights: Dict[str, np.ndarray], input_width: int, path: str, vanilla: bool = False, config: dict = None) -> None:
    """the counterpart of ``onnx.save(tf2onnx.convert.from_keras(...), path)`` (pointnet_train.py:238-248); ``config`` =
    PointNet.get_config(), kept in the file's metadata_props"""
    with open(path, "wb") as f:
        f.write(build_model_bytes(weights, input_width, vanilla, config))


# ----------------------------------------------------------------------------------------------------------------
# reader: the structure of a model file and its initializers (used to restore weights and by the tests)
# ----------------------------------------------------------------------------------------------------------------
def _parse_tensor(buf) -> tuple:
    dims, dt, name, raw = [], None, "", b""
    for f, wt, v in _fields(buf):
        if f == 1:
            dims.append(int(v))
        elif f == 2:
            dt = int(v)
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
    arr = np.frombuffer(raw, dtype="<f4" if dt == FLOAT else "<i8").reshape(dims)
    return name, arr


def _parse_attr(buf):
    name, val, ints, typ = "", None, [], None
    for f, wt, v in _fields(buf):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            val = float(np.frombuffer(bytes(v), dtype="<f4")[0])
        elif f == 3:
            val = int(v) - (1 << 64) if int(v) >= (1 << 63) else int(v)
        elif f == 8:
            ints.append(int(v) - (1 << 64) if int(v) >= (1 << 63) else int(v))
        elif f == 20:
            typ = int(v)
    return name, (ints if typ == A_INTS else val)


def parse_model(data: bytes) -> dict:
    """{'ir_version', 'opset', 'nodes': [{'op','name','inputs','outputs','attrs'}], 'initializers': {name: array}, 'inputs', 'outputs'}"""
    out = {"nodes": [], "initializers": {}, "inputs": [], "outputs": [], "metadata": {}}
    for f, wt, v in _fields(memoryview(data)):
        if f == 1:
            out["ir_version"] = int(v)
        elif f == 14:
            kv = {f2: bytes(v2).decode() for f2, _, v2 in _fields(v)}
            out["metadata"][kv.get(1, "")] = kv.get(2, "")
        elif f == 8:
            for f2, _, v2 in _fields(v):
                if f2 == 2:
                    out["opset"] = int(v2)
        elif f == 7:
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    node = {"inputs": [], "outputs": [], "attrs": {}}
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            node["inputs"].append(bytes(v3).decode())
                        elif f3 == 2:
                            node["outputs"].append(bytes(v3).decode())
                        elif f3 == 3:
                            node["name"] = bytes(v3).decode()
                        elif f3 == 4:
                            node["op"] = bytes(v3).decode()
                        elif f3 == 5:
                            k, a = _parse_attr(v3)
                            node["attrs"][k] = a
                    out["nodes"].append(node)
                elif f2 == 5:
                    n, a = _parse_tensor(v2)
                    out["initializers"][n] = a
                elif f2 in (11, 12):
                    nm, shape = "", []
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            nm = bytes(v3).decode()
                        elif f3 == 2:
                            for f4, _, v4 in _fields(v3):          # TypeProto.tensor_type
                                for f5, _, v5 in _fields(v4):
                                    if f5 == 2:                    # shape
                                        for f6, _, v6 in _fields(v5):
                                            for f7, _, v7 in _fields(v6):
                                                shape.append(int(v7) if f7 == 1 else bytes(v7).decode())
                    out["inputs" if f2 == 11 else "outputs"].append((nm, shape))
    return out


def read_onnx_weights(path: str) -> Dict[str, np.ndarray]:
    """canonical name -> array for every parameter of a file written by export_onnx (kernels back in Keras layout (Cin, Cout))"""
    with open(path, "rb") as f:
        m = parse_model(f.read())
    w = {}
    for name, arr in m["initializers"].items():
        if "/" in name:
            continue                                                  # shape / helper constants
        if name.endswith(".kernel") and arr.ndim == 4:
            arr = arr.reshape(arr.shape[0], arr.shape[1]).T
        w[name] = np.array(arr, dtype=np.float32)
    return w


def read_onnx_config(path: str) -> dict:
    """the PointNet constructor arguments export_onnx stored in the file's metadata_props ({} for a file without them)"""
    with open(path, "rb") as f:
        m = parse_model(f.read())
    return {k[len("pointnet."):]: json.loads(v) for k, v in m["metadata"].items() if k.startswith("pointnet.")}
