"""CPU tests (no GPU): the oracle against everything the reference itself pins, the host logic, and the C ABI
surface (symbols only -- no compute calls without a GPU)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import pointnet_oracle as O
from oracle import sampling_oracle as SO
from helpers import F15_CLASSES, F15_PARTS, GOLD, make_collect as _make_collect

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



# ---------------------------------------------------------------------------------------------- oracle pins
def test_parameter_census_matches_reference_log():
    """4,210,476 trainable + 14,208 non-trainable (computed from PointNet.py:116-141,406-416)."""
    assert O.census(O.init_params(23, 12)) == (4210476, 14208)


def test_trainability_names_match_reference_log():
    """the 16 names the reference run logged (models/f15_scale_lidar/log_20260126_16*0916.log:203-218)."""
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    m = PointNet(23, 12, 0.3, 42, device="cpu")
    assert list(m.get_layer_trainability().keys()) == O.TRAINABILITY_NAMES
    assert all(v for k, v in m.get_layer_trainability().items() if k != "input_normalization")
    assert m.get_layer_trainability()["input_normalization"] is False
    mv = PointNet(23, 12, 0.3, 42, vanilla=True, device="cpu")
    assert list(mv.get_layer_trainability().keys()) == [n for n in O.TRAINABILITY_NAMES if n not in ("input_transform", "feature_transform")]
    assert m.count_params() == (4210476, 14208)
    assert m.input_names == ['pointnet_input'] and m.output_names == ['classification_output', 'segmentation_output', 'se3']


def test_freeze_thaw_order_of_the_trainer():
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    m = PointNet(23, 12, 0.3, 42, device="cpu")
    # classification_pretrain profile of f15_lidar_config.json:57-62 in the trainer's call order (pointnet_train.py:322-332)
    m.thaw_shared_network(); m.thaw_input_transform(); m.thaw_classification_head(); m.freeze_segmentation_head()
    t = m.get_layer_trainability()
    assert [k for k, v in t.items() if not v] == ["input_normalization", "seg_l1_512_convolution_layer", "seg_l2_256_convolution_layer",
                                                  "seg_l3_128_convolution_layer", "seg_l4_128_convolution_layer",
                                                  "seg_l5_output_convolution_layer"]
    m.freeze_shared_network(); m.thaw_input_transform()
    t = m.get_layer_trainability()
    assert t["input_transform"] and not t["feature_transform"] and not t["s1_l1_64_convolution_layer"]
    mv = PointNet(23, 12, 0.3, 42, vanilla=True, device="cpu")
    mv.freeze_shared_network(); mv.thaw_shared_network()      # the reference raises AttributeError here (PointNet.py:302-318)
    cfg = mv.get_config()
    assert cfg["vanilla"] is True                              # the reference drops it (PointNet.py:354-362)
    assert PointNet.from_config(cfg)._vanilla is True


def _read_cloud(fn):
    pts, cls, parts = [], [], []
    for line in open(os.path.join(GOLD, fn)):
        m = re.match(r"\(([^)]*)\)\s+(\S+)\s+(\S+)", line.strip())
        pts.append([float(v) for v in m.group(1).split(",")])
        cls.append(m.group(2)); parts.append(m.group(3))
    return np.asarray(pts), cls, parts


def test_normalisation_on_the_reference_clouds():
    """centroids / extents of the two labelled clouds the reference ships."""
    pts, _, _ = _read_cloud("kc-46.txt")
    assert pts.shape == (490, 3)
    assert np.allclose(pts.mean(0), [-9.330, -0.005, -3.962], atol=1e-3)
    assert np.allclose(pts.min(0), [-26.059, -23.573, -7.353], atol=1e-3) and np.allclose(pts.max(0), [17.151, 23.501, 5.358], atol=1e-3)
    out, (cen, scale) = O.normalize(torch.from_numpy(pts).unsqueeze(0))
    assert np.allclose(cen[0, 0].numpy(), pts.mean(0))
    r = np.linalg.norm(pts - pts.mean(0), axis=1)
    assert abs(float(scale) - r.max()) < 1e-9
    assert abs(float(torch.linalg.norm(out[0], dim=-1).max()) - 1.0) < 1e-9
    p2, _, _ = _read_cloud("f-15_model.txt")
    assert p2.shape == (313, 3) and np.allclose(p2.mean(0), [-0.786, 0.016, -0.171], atol=1e-3)


def test_split_kernel_form_equals_tile_concat():
    """seg_l1 on [X_64 || tile(global)] (PointNet.py:268-275) == X_64 W[:64] + global W[64:]."""
    p = O.init_params(7, 5, seed=3, randomize_bn=True, dtype=torch.float64)
    pc = torch.randn(3, 50, 3, dtype=torch.float64)
    a = O.forward(p, pc)[1]
    b = O.forward_concat_form(p, pc)
    assert float((a - b).abs().max()) < 1e-12


def test_max_tie_rules_give_equal_parameter_gradients_on_duplicated_points():
    """R5: TF splits the reduce_max gradient among ties, the build sends it to the first; with ties that are duplicated
    points (the reference pads clouds with duplicates, PointCloudSet.py:459-463) parameter gradients are identical."""
    p = O.init_params(5, 4, seed=4, randomize_bn=True, dtype=torch.float64)
    pc = torch.randn(2, 40, 3, dtype=torch.float64)
    pc = torch.cat([pc, pc[:, :24]], dim=1)                   # 24 duplicated points per cloud
    tg = {"classification_output": torch.tensor([1, 3]), "segmentation_output": torch.randint(0, 4, (2, 64)),
          "se3": torch.eye(3, dtype=torch.float64).expand(2, 3, 3)}
    gs = []
    for split in (False, True):
        leaves = {k: v.clone().requires_grad_(O.is_trainable_name(k)) for k, v in p.items()}
        outs, ctx = O.forward(leaves, pc, training=True, tie_split=split, return_ctx=True)
        loss, _ = O.total_loss(outs, tg, dict(classification=1.0, segmentation=1.0, rotation=1.0))
        names = [k for k in leaves if leaves[k].requires_grad]
        gs.append(dict(zip(names, torch.autograd.grad(loss, [leaves[k] for k in names]))))
    for k in gs[0]:
        assert float((gs[0][k] - gs[1][k]).abs().max()) < 1e-10 * max(1.0, float(gs[0][k].abs().max())), k


def test_keras_adam_and_schedule_known_answers():
    assert abs(O.exponential_decay_lr(1e-4, 7000, 7000, 0.7) - 0.7e-4) < 1e-18
    assert abs(O.exponential_decay_lr(1e-4, 3500, 7000, 0.7) - 1e-4 * 0.7 ** 0.5) < 1e-18
    p, g = torch.tensor([1.0]), torch.tensor([0.5])
    m, v = torch.zeros(1), torch.zeros(1)
    O.keras_adam_step(p, g, m, v, 0, 1e-3)
    # first Adam step moves by ~lr * sign(g): alpha = lr*sqrt(1-b2)/(1-b1), m = 0.1 g, v = 0.001 g^2
    assert abs(float(p) - (1.0 - 1e-3 * 0.5 / (0.5 + 1e-7 / (0.001 ** 0.5)))) < 1e-7


def test_keras_sparse_cce_clipping():
    probs = torch.tensor([[1.0, 0.0, 0.0], [0.2, 0.3, 0.5]], dtype=torch.float64)
    l = O.keras_sparse_cce(probs, torch.tensor([1, 2]))
    # row 0: p clipped to 1e-7 / (1 - 1e-7): -log(1e-7 / (1 - 1e-7 + 2e-7)); row 1: -log(0.5)
    exp0 = -(np.log(1e-7) - np.log((1 - 1e-7) + 2e-7))
    assert abs(float(l) - (exp0 - np.log(0.5)) / 2) < 1e-9


# ---------------------------------------------------------------------------------------------- samplers (oracle only)
def test_fps_oracle_properties():
    rng = np.random.default_rng(0)
    xyz = rng.normal(size=(500, 3)).astype(np.float32)
    idx, md = SO.fps(xyz, 50, 3)
    assert idx[0] == 3 and len(set(idx.tolist())) == 50
    assert np.all(md[idx[:-1]] == 0)          # the last pick is reported before its own distance update
    # the k-th pick is the farthest point from the first k-1 picks
    d = ((xyz[:, None, :] - xyz[idx[:10]][None]) ** 2).sum(-1).min(1)
    assert idx[10] == int(np.argmax(d))


def test_voxel_oracle_properties():
    rng = np.random.default_rng(1)
    xyz = rng.uniform(0, 4, size=(2000, 3)).astype(np.float32)
    lab = rng.integers(0, 5, size=2000).astype(np.int32)
    c, n, m = SO.voxel_downsample(xyz, (1, 1, 1), (0, 0, 0), lab, 5)
    assert n.sum() == 2000 and c.shape[0] == 64
    k = np.floor(c).astype(int)
    key = k[:, 2] * 100 + k[:, 1] * 10 + k[:, 0]
    assert np.all(np.diff(key) > 0)                            # ascending (kz, ky, kx)
    sel = np.all(np.floor(xyz) == [1, 2, 3], axis=1)
    v = np.flatnonzero((k == [1, 2, 3]).all(1))[0]
    assert np.allclose(c[v], xyz[sel].astype(np.float64).mean(0), atol=1e-6) and n[v] == sel.sum()
    assert m[v] == np.bincount(lab[sel], minlength=5).argmax()


# ---------------------------------------------------------------------------------------------- dataset layer
def test_frame_parser_known_answers(tmp_path):
    """490 / 313 points and the part histograms of the reference's two clouds."""
    from pointcloudprocessing_amd.pointcloud.PointCloudSet import PointCloudSet
    pcs = PointCloudSet("kat", F15_CLASSES, F15_PARTS, 512, data_path=str(tmp_path) + "/", print_func=lambda s: None)
    obs, cl, pl, nonf = pcs._parse_frame(os.path.join(GOLD, "kc-46.txt"))
    assert obs.shape == (490, 3) and cl == F15_CLASSES.index("kc-46") and nonf == 0
    hist = {F15_PARTS[i]: int(c) for i, c in enumerate(np.bincount(pl, minlength=12)) if c}
    assert hist == {"engine": 60, "fuselage": 159, "wing": 137, "boom_hull": 12, "hstab": 83, "boom_wing": 4, "vstab": 35}
    ref_pts, _, _ = _read_cloud("kc-46.txt")
    assert np.array_equal(obs, ref_pts)                        # strtod == float()
    obs, cl, pl, _ = pcs._parse_frame(os.path.join(GOLD, "f-15_model.txt"))
    hist = {F15_PARTS[i]: int(c) for i, c in enumerate(np.bincount(pl, minlength=12)) if c}
    assert obs.shape == (313, 3) and hist == {"fuselage": 119, "engine": 64, "wing": 85, "hstab": 22, "vstab": 23}
    # error behaviour of PointCloudSet.py:179-185
    bad = tmp_path / "bad.txt"
    bad.write_text("(1.0, 2.0, 3.0) kc-46 not_a_part\n")
    with pytest.raises(Exception, match="Part label not_a_part not known"):
        pcs._parse_frame(str(bad))
    bad.write_text("(1.0, 2.0, 3.0) kc-46\n")
    with pytest.raises(Exception, match="both a class label and part label"):
        pcs._parse_frame(str(bad))
    bad.write_text("(1.0, nan, 3.0) kc-46 wing\n(1.0, 2.0, 3.0) kc-46 wing\n")
    obs, cl, pl, nonf = pcs._parse_frame(str(bad))
    assert obs.shape == (1, 3) and nonf == 1


def test_adjust_to_input_width_and_state_info(tmp_path):
    from pointcloudprocessing_amd.pointcloud.PointCloudSet import PointCloudSet
    pcs = PointCloudSet("adj", F15_CLASSES, F15_PARTS, 512, rand_seed=1, data_path=str(tmp_path) + "/", print_func=lambda s: None)
    obs, _, pl, _ = pcs._parse_frame(os.path.join(GOLD, "kc-46.txt"))
    o2, p2 = pcs._adjust_to_input_width(obs, pl)
    assert o2.shape == (512, 3) and np.array_equal(o2[:490], obs)
    for j in range(490, 512):                                  # padding rows are copies of existing points with their labels
        src = np.flatnonzero((obs == o2[j]).all(1))
        assert len(src) and p2[j] in pl[src]
    pcs._network_input_width = 100
    o3, p3 = pcs._adjust_to_input_width(obs, pl)
    assert np.array_equal(o3, obs[:100]) and np.array_equal(p3, pl[:100])
    d = _make_collect(str(tmp_path), "collect_kat", 3)
    st = pcs._parse_state_info(os.path.join(d, "_palindrome_state__collect_kat.log"))
    assert set(st.keys()) == {0, 1, 2}
    sp, tp, ts = st[1]['Sensor Pose'], st[1]['Tanker Pose'], st[1]['tanker_in_sensor_frame']
    assert np.allclose(ts[:3, :3], sp[:3, :3].T @ tp[:3, :3]) and np.allclose(ts[:3, 3], sp[:3, :3].T @ (tp[:3, 3] - sp[:3, 3]))
    assert np.allclose(ts[3], [0, 0, 0, 1]) and np.allclose(sp[:3, :3] @ sp[:3, :3].T, np.eye(3), atol=1e-6)


def test_tfrecord_bytes_and_round_trip(tmp_path):
    from pointcloudprocessing_amd.pointcloud import PointCloudSet as P
    # CRC-32C known answer (RFC 3720) and the TFRecord mask
    h = P._hostlib()
    assert h.pn_crc32c(b"123456789", 9) == 0xE3069283
    assert h.pn_masked_crc32c(b"123456789", 9) == (((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xa282ead8) & 0xFFFFFFFF
    # tf.train.Example bytes for a tiny example, written out by hand from the protobuf wire format
    ex = P.serialize_example({'a': P._int64_feature([3]), 'b': P._float_feature([1.0])})
    assert ex == bytes.fromhex("0a1b" "0a0a" "0a0161" "1205" "1a03" "0a0103" "0a0d" "0a0162" "1208" "1206" "0a04" "0000803f")
    back = P.parse_example(ex)
    assert back['a'].tolist() == [3] and back['b'].tolist() == [1.0]
    big = P._int64_feature([0, 127, 128, 300, -1])
    assert P.parse_example(P.serialize_example({'x': big}))['x'].tolist() == [0, 127, 128, 300, -1]
    fn = tmp_path / "t.tfrecord"
    with P.TFRecordWriter(str(fn)) as w:
        w.write(ex); w.write(b"")
    assert list(P.read_tfrecords(str(fn))) == [ex, b""]
    raw = bytearray(fn.read_bytes()); raw[14] ^= 1
    fn.write_bytes(bytes(raw))
    with pytest.raises(IOError):
        list(P.read_tfrecords(str(fn)))


def test_dataset_build_split_and_batches(tmp_path):
    from pointcloudprocessing_amd.pointcloud import PointCloudSet as P
    d = _make_collect(str(tmp_path), "collect_a", 21)          # 21 frames + 1 extra file -> one "missing frame" like the reference
    open(os.path.join(d, "Lidar", "_Lidar__.log"), "w").write("x")
    msgs = []
    pcs = P.PointCloudSet("ds", F15_CLASSES, F15_PARTS, 512, jitter_stdev_m=np.array([0.1, 0.1, 0.1]), batch_size=4, rand_seed=7,
                          data_path=str(tmp_path) + "/", print_func=msgs.append)
    assert pcs.add_from_aftr_output(d)
    assert any("Failed to add file" in m and "frame_21.txt" in m for m in msgs)
    # first ceil(0.10 n) test, next ceil(0.15 n) val, rest train (PointCloudSet.py:245-247)
    assert [pcs._data_size[s]['count'] for s in ('test', 'val', 'train')] == [3, 4, 14]
    assert sorted(os.listdir(tmp_path / "ds" / "collect_a")) == ["test_0.tfrecord", "train_0.tfrecord", "val_0.tfrecord"]
    assert sum(pcs._data_size['train']['part_count'].values()) == 14 * 512
    x, y = next(pcs.get_train_set())
    assert x.shape == (4, 512, 3) and y['classification_output'].shape == (4,) and y['segmentation_output'].shape == (4, 512)
    assert y['se3'].shape == (4, 3, 3) and x.dtype == torch.float32 and y['segmentation_output'].dtype == torch.int32
    assert torch.allclose(y['se3'] @ y['se3'].transpose(1, 2), torch.eye(3).expand(4, 3, 3), atol=1e-5)
    rel = P.load_from_file(str(tmp_path / "ds" / "pc_set.joblib"))
    assert rel._data_size == pcs._data_size and "Total count: 14" in rel.get_info()
    # shards for 2 ranks are disjoint streams
    a = next(pcs.get_val_set(rank=0, world_size=2))[1]['se3']
    b = next(pcs.get_val_set(rank=1, world_size=2))[1]['se3']
    assert a.shape == b.shape


# ---------------------------------------------------------------------------------------------- C ABI surface
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pointnet_hip.h")).read()
    declared = set(re.findall(r"\b(pn_[a-z0-9_]+)\s*\(", hdr)) - {"pn_operand"}
    from pointcloudprocessing_amd import _lib
    l = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(l, name), f"{name} declared in pointnet_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.lib().pn_abi_version() == _lib.ABI_VERSION == 6
    # host-only entry points can be exercised without a GPU
    d = _lib.pn_model_desc(ccls=23, cseg=12, vanilla=0, reg_in=0, reg_feat=0, prec=3, dropout_rate=0.3, bn_momentum=0.99, bn_eps=1e-3)
    assert _lib.lib().pn_model_num_slots(C.byref(d)) == len(O.init_params(23, 12))
    info = _lib.pn_slot_info()
    names = []
    for i in range(_lib.lib().pn_model_num_slots(C.byref(d))):
        assert _lib.lib().pn_model_slot_info(C.byref(d), i, C.byref(info)) == 0
        names.append(info.name.decode())
        assert info.offset % 64 == 0
    assert names == list(O.init_params(23, 12).keys())
    assert _lib.lib().pn_model_workspace_bytes(C.byref(d), 4, 1024, 1) > _lib.lib().pn_model_workspace_bytes(C.byref(d), 4, 1024, 0) > 0
    bad = _lib.pn_model_desc(ccls=23, cseg=99, prec=3, dropout_rate=0.3, bn_momentum=0.99, bn_eps=1e-3)
    assert _lib.lib().pn_model_num_slots(C.byref(bad)) == -1 and b"segmentation width" in _lib.lib().pn_last_error()


def test_oracle_reproduces_committed_golden_vectors():
    """tests/golden/oracle_b2_n64.npz (made by tests/golden/make_oracle_vectors.py) pins the oracle against drift: inference
    outputs, training-mode outputs, the three keras losses and a digest of every parameter gradient, all fp64."""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_oracle_vectors", os.path.join(here, "golden", "make_oracle_vectors.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    want = np.load(os.path.join(here, "golden", "oracle_b2_n64.npz"))
    got = mod.build()
    assert set(want.files) == set(got.keys())
    for k in want.files:
        assert np.allclose(np.asarray(got[k], dtype=np.float64), want[k].astype(np.float64), rtol=1e-9, atol=1e-12), k
    assert np.array_equal(got["inf_cls"].argmax(-1), want["inf_cls"].argmax(-1))


def test_gradient_bucket_boundary_matches_backward_phases():
    """engine.TrainStep all-reduces grads_flat[cut:] after backward phase 1 and grads_flat[:cut] after phase 2: the cut must sit
    exactly between the slots phase 2 writes (input transform, mlp_1) and everything else (pn_model_io.bwd_phase)."""
    from pointcloudprocessing_amd.pointnet.PointNet import PointNet
    for vanilla in (False, True):
        m = PointNet(23, 12, 0.3, 42, vanilla=vanilla, device="cpu")
        cut = m.grad_bucket_boundary()
        late = ("input_transform.", "mlp_1_1.", "mlp_1_2.")
        for n, s in m._weights.slots.items():
            assert (s["offset"] < cut) == n.startswith(late), (n, s["offset"], cut)
        assert 0 < cut < m.params_flat.numel()


def test_graft_entry_build_hook_runs():
    """the driver's build check: compiles every native piece in-tree and imports the package (no GPU needed)"""
    import __graft_entry__ as g
    g.build()


def test_library_load_pulls_in_pytorch_first():
    """one HIP runtime per process: _lib.lib() imports torch (whose wheel bundles libamdhip64) before it dlopens the library, so
    `build()` followed by `smoke()` in one interpreter works whatever was imported first"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from pointcloudprocessing_amd import _lib; assert 'torch' not in sys.modules; "
            "_lib.lib(); assert 'torch' in sys.modules; print('ok')" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-1500:]
