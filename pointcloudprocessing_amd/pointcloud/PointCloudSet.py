"""Drop-in for the reference's ``pointcloud/PointCloudSet.py`` without TensorFlow.

Same constructor, public methods, on-disk layout and TFRecord / ``tf.train.Example`` byte format as
``/root/reference/point_cloud_analysis/pointcloud/PointCloudSet.py`` (class :33-509, helpers :512-538), so
datasets written by either side can be read by the other:

  <data_path><name>/pc_set.joblib                                   (:111-114)
  <data_path><name>/<set_name>/{test,val,train}_<k>.tfrecord        (:251-288)
  Example features: observations float[N*3], class_label int64[1], part_labels int64[N], se3 float[9]  (:100-105,306-323)

``get_train_set()/get_val_set()/get_test_set()`` return infinite iterables of
``(x, {'classification_output', 'segmentation_output', 'se3'})`` batches (:347-391) -- torch tensors, placed on
``device`` when given so that shuffle / jitter / batch run on the GPU next to the model.

Reference defects tolerated, not replicated: ``rand_seed`` really seeds this object's
RNG (the reference discards ``np.random.default_rng(seed)``, :84-88); ``get_dir_contents`` reports errors
through ``_print`` without the ``file=`` keyword (which made ``logger.info`` raise, :531-537);
``load_from_file`` appends ``.pkl`` only when it is missing (:515-516).
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import struct
import sys
from collections.abc import Callable
from copy import deepcopy

import joblib
import numpy as np

from ..utils import global_constants as constants

_HERE = os.path.dirname(os.path.abspath(__file__))
_HOST_LIB = os.path.join(os.path.dirname(_HERE), "libpn_host.so")
_host = None


def _hostlib():
    """libpn_host.so: CRC-32C and the native Aftr frame parser (plain C, built by __graft_entry__.build())."""
    global _host
    if _host is None:
        if not os.path.exists(_HOST_LIB):
            raise RuntimeError(f"{_HOST_LIB} is missing: run __graft_entry__.build()")
        l = C.CDLL(_HOST_LIB)
        l.pn_masked_crc32c.restype = C.c_uint32
        l.pn_masked_crc32c.argtypes = [C.c_char_p, C.c_size_t]
        l.pn_crc32c.restype = C.c_uint32
        l.pn_crc32c.argtypes = [C.c_char_p, C.c_size_t]
        l.pn_parse_aftr_frame.restype = C.c_long
        l.pn_parse_aftr_frame.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_char_p), C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_long, C.POINTER(C.c_int32), C.POINTER(C.c_long),
                                          C.POINTER(C.c_long)]
        _host = l
    return _host


# ----------------------------------------------------------------------------------------------------------
# TFRecord framing + tf.train.Example wire format (protobuf), hand-encoded
# ----------------------------------------------------------------------------------------------------------
def _varint(n: int) -> bytes:
    if n < 0:
        n += 1 << 64
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:      # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _float_feature(values) -> bytes:
    raw = np.asarray(values, dtype="<f4").tobytes()
    return _ld(2, _ld(1, raw))                      # Feature.float_list(2) { value(1) packed }


def _int64_feature(values) -> bytes:
    v = np.atleast_1d(np.asarray(values, dtype=np.int64))
    if v.size and v.min() >= 0 and v.max() < 128:
        raw = v.astype(np.uint8).tobytes()
    else:
        raw = b"".join(_varint(int(x)) for x in v)
    return _ld(3, _ld(1, raw))                      # Feature.int64_list(3) { value(1) packed }


def serialize_example(features: dict) -> bytes:
    """tf.train.Example(features=Features(feature={...})).SerializeToString() (keys sorted)."""
    body = b"".join(_ld(1, _ld(1, k.encode()) + _ld(2, features[k])) for k in sorted(features))
    return _ld(1, body)


def _read_varint(buf, pos):
    r, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, pos
        shift += 7


def _fields(buf):
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 2:
            ln, pos = _read_varint(buf, pos)
            yield f, wt, buf[pos:pos + ln]
            pos += ln
        elif wt == 0:
            v, pos = _read_varint(buf, pos)
            yield f, wt, v
        elif wt == 5:
            yield f, wt, buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            yield f, wt, buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError(f"unsupported wire type {wt}")


def parse_example(buf: bytes) -> dict:
    """Inverse of serialize_example for float_list / int64_list features (packed or not)."""
    out = {}
    for f, _, feats in _fields(memoryview(buf)):
        if f != 1:
            continue
        for f2, _, entry in _fields(feats):
            if f2 != 1:
                continue
            key, feat = None, None
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    key = bytes(v).decode()
                elif f3 == 2:
                    feat = v
            for kind, _, lst in _fields(feat):
                if kind == 2:       # FloatList
                    vals = []
                    for f5, wt, v in _fields(lst):
                        if wt == 2:
                            vals.append(np.frombuffer(v, dtype="<f4"))
                        else:
                            vals.append(np.frombuffer(v, dtype="<f4", count=1))
                    out[key] = np.concatenate(vals) if vals else np.zeros(0, np.float32)
                elif kind == 3:     # Int64List
                    vals = []
                    for f5, wt, v in _fields(lst):
                        if wt == 2:
                            a = np.frombuffer(v, dtype=np.uint8)
                            if a.size and a.max() < 128:
                                vals.append(a.astype(np.int64))
                            else:
                                p, tmp = 0, []
                                while p < len(v):
                                    x, p = _read_varint(v, p)
                                    tmp.append(x - (1 << 64) if x >= (1 << 63) else x)
                                vals.append(np.asarray(tmp, dtype=np.int64))
                        else:
                            vals.append(np.asarray([v], dtype=np.int64))
                    out[key] = np.concatenate(vals) if vals else np.zeros(0, np.int64)
    return out


class TFRecordWriter:
    """tf.io.TFRecordWriter: u64 length, masked crc32c(length), payload, masked crc32c(payload)."""

    def __init__(self, path):
        self._f = open(path, "wb")

    def write(self, record: bytes):
        h = _hostlib()
        ln = struct.pack("<Q", len(record))
        self._f.write(ln)
        self._f.write(struct.pack("<I", h.pn_masked_crc32c(ln, 8)))
        self._f.write(record)
        self._f.write(struct.pack("<I", h.pn_masked_crc32c(record, len(record))))

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def read_tfrecords(path, verify: bool = True):
    h = _hostlib()
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if len(head) < 12:
                return
            (ln,) = struct.unpack("<Q", head[:8])
            if verify and struct.unpack("<I", head[8:])[0] != h.pn_masked_crc32c(head[:8], 8):
                raise IOError(f"{path}: corrupt record length")
            data = f.read(ln)
            (crc,) = struct.unpack("<I", f.read(4))
            if verify and crc != h.pn_masked_crc32c(data, ln):
                raise IOError(f"{path}: corrupt record payload")
            yield data


# ----------------------------------------------------------------------------------------------------------
class PointCloudSet:
    def __init__(self,
                 name: str,
                 class_labels: list,
                 part_labels: list,
                 network_input_width: int,
                 jitter_stdev_m: np.ndarray = np.array([0, 0, 0]),
                 val: float = 0.15,
                 test: float = 0.10,
                 batch_size: int = 32,
                 rand_seed=None,
                 description: str = '',
                 print_func: Callable[[str], None] = print,
                 data_path: str = ''):

        self._description: str = description
        self._batch_size: int = batch_size
        self._name: str = name
        self._class_labels: dict = {}
        self._class_str: dict = {}
        for i, label in enumerate(class_labels):
            self._class_labels[label] = i
            self._class_str[i] = label
        self._part_labels: dict = {}
        self._part_str: dict = {}
        for i, label in enumerate(part_labels):
            self._part_labels[label] = i
            self._part_str[i] = label
        self._network_input_width: int = network_input_width
        self._jitter_stdev_m: np.ndarray = np.asarray(jitter_stdev_m, dtype=np.float32)
        self._print = print_func
        self._data_path = data_path
        self._sets_added = 0
        self._data_size = {k: {'count': 0, 'class_count': {}, 'part_count': {}} for k in ('train', 'val', 'test')}

        if type(rand_seed) == int:
            self._random_seed = rand_seed
        else:
            self._random_seed = None
        self._rng = np.random.default_rng(self._random_seed)

        if val < 1.0 and test < 1.0 and 1.0 - (val + test) < 1.0:
            self._train_amt = 1.0 - (val + test)
            self._val_amt = val
            self._test_amt = test
        else:
            self._train_amt = 0.75
            self._val_amt = 0.15
            self._test_amt = 0.10
            self._print('PointCloudSet:  train_val_test_split incorrect format - set to default 75% / 15% / 10%')

        self._feature_description = {
            'observations': ([self._network_input_width * 3], 'float32'),
            'class_label': ([], 'int64'),
            'part_labels': ([self._network_input_width], 'int64'),
            'se3': ([9], 'float32'),
        }

        if not os.path.isdir(f"{self._data_path}{self._name}"):
            os.mkdir(f"{self._data_path}{self._name}")

        self.save()

    # logger functions and the RNG do not survive pickling on every platform: drop and restore
    def __getstate__(self):
        st = dict(self.__dict__)
        st['_print'] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        if self._print is None:
            self._print = print

    def save(self):
        with open(f"{self._data_path}{self._name}/pc_set.joblib", "wb") as jl:
            joblib.dump(self, jl)

    # ------------------------------------------------------------------------------------------------------
    def _parse_frame(self, path: str):
        """One Aftr frame -> (obs (n,3) float64, class id, part ids (n,), non-finite count).  Native parser with
        the reference's per-line rules (PointCloudSet.py:161-198); raises like the reference on bad labels."""
        h = _hostlib()
        with open(path, 'rb') as f:
            text = f.read()
        max_pts = text.count(b'\n') + 1
        xyz = np.empty((max_pts, 3), dtype=np.float64)
        part = np.empty(max_pts, dtype=np.int32)
        cn = (C.c_char_p * len(self._class_labels))(*[k.encode() for k in self._class_labels])
        pn = (C.c_char_p * len(self._part_labels))(*[k.encode() for k in self._part_labels])
        cls, nonf, errl = C.c_int32(), C.c_long(), C.c_long()
        n = h.pn_parse_aftr_frame(text, len(text), cn, len(self._class_labels), pn, len(self._part_labels), xyz.ctypes.data,
                                  part.ctypes.data, max_pts, C.byref(cls), C.byref(nonf), C.byref(errl))
        if n < 0:
            line = text.split(b'\n')[errl.value].decode(errors='replace').strip()
            labels = [l for l in line[line.find(')') + 1:].split(" ") if len(l) > 1]
            if n == -2:
                raise Exception(f"Class label {labels[0]} not known")
            if n == -3:
                raise Exception(f"Part label {labels[1]} not known")
            if n == -1 and '(' in line and ')' in line and len(labels) != 2:
                raise Exception(f"Dataset must contain both a class label and part label. {labels} is not correct.")
            raise ValueError(f"could not parse line {errl.value}: {line!r}")
        return xyz[:n].copy(), int(cls.value), part[:n].astype(np.int64), int(nonf.value)

    def add_from_aftr_output(self, dir_path: str, shuffle_points: bool = True) -> bool:
        '''
        Parses the standard SensorDatumLogger output (reference PointCloudSet.py:116-218):
            <dir_path>/Lidar/frame_<i>.txt      one "(x, y, z) <class> <part>" line per point
            <dir_path>/_palindrome_state_*.log  pose log

        @return True if parsing is successful / False otherwise
        '''
        observations, class_labels, part_labels, se3 = [], [], [], []
        frames_searched = 0
        non_num_found = 0

        collect_contents = get_dir_contents(dir_path, self._print)
        lidar_contents = get_dir_contents(f'{dir_path}/Lidar', self._print)

        pose_log = [i for i in collect_contents if '_palindrome_state' in i]
        if len(pose_log) == 1:
            state_info = self._parse_state_info(f'{dir_path}/{pose_log[0]}')
        else:
            raise Exception(f"No state info found in {dir_path}")

        self._print(f"Parsing frames in {dir_path}...")
        for i in range(len(lidar_contents)):
            try:
                obs, cl, pl, nonf = self._parse_frame(f'{dir_path}/Lidar/frame_{i}.txt')
                non_num_found += nonf
                if len(obs) != 0:
                    se = state_info[i]['tanker_in_sensor_frame'][:3, :3]
                    obs, pl = self._adjust_to_input_width(obs, pl)
                    if np.isfinite(obs).all():
                        observations.append(obs)
                        class_labels.append(cl)
                        part_labels.append(pl)
                        se3.append(se)
                    else:
                        self._print(f'Per-line check failed - frame_{i} discarded after detecting non-finite value.')
            except Exception as e:
                if frames_searched == 0:
                    frames_searched = i
                self._print(f"Failed to add file {dir_path}/Lidar/frame_{i}.txt:\n\t{type(e).__name__} : {e}")

        self.add_data(dir_path.split("/")[-1], np.array(observations), np.array(class_labels), np.array(part_labels), np.array(se3),
                      shuffle_points)
        return True

    def add_data(self, set_name: str, observations: np.ndarray, class_labels: np.ndarray, part_labels: np.ndarray, se3: np.ndarray,
                 shuffle_points: bool = True) -> None:
        '''
        Adds data and splits it test / val / train in that order (first ceil(n*test), next ceil(n*val), rest train;
        reference PointCloudSet.py:220-292).  One TFRecord file per split per call.
        '''
        if shuffle_points:
            indices = np.arange(observations.shape[0])
            self._rng.shuffle(indices)
            observations = observations[indices]
            class_labels = class_labels[indices]
            part_labels = part_labels[indices]
            se3 = se3[indices]

        n = observations.shape[0]
        n_test = int(np.ceil(n * self._test_amt))
        n_val = int(np.ceil(n * self._val_amt))
        splits = {'test': (0, n_test), 'val': (n_test, n_test + n_val), 'train': (n_test + n_val, n)}

        if not os.path.isdir(f"{self._data_path}{self._name}/{set_name}"):
            os.mkdir(f"{self._data_path}{self._name}/{set_name}")

        for split in ('test', 'val', 'train'):
            lo, hi = splits[split]
            with TFRecordWriter(f"{self._data_path}{self._name}/{set_name}/{split}_{self._sets_added}.tfrecord") as writer:
                for i in range(lo, min(hi, n)):
                    ds = self._data_size[split]
                    cname = self._class_str[int(class_labels[i])]
                    ds['class_count'][cname] = ds['class_count'].get(cname, 0) + 1
                    counts = np.bincount(np.asarray(part_labels[i], dtype=np.int64), minlength=len(self._part_labels))
                    for lbl, idx in self._part_labels.items():
                        ds['part_count'][lbl] = ds['part_count'].get(lbl, 0) + int(counts[idx])
                    writer.write(self._serialize_sample(observations[i], class_labels[i], part_labels[i], se3[i]))
                    ds['count'] += 1

        self._sets_added += 1
        self.save()

    def _serialize_sample(self, obs: np.ndarray, cls, seg: np.ndarray, se3: np.ndarray) -> bytes:
        """tf.train.Example with the four features of PointCloudSet.py:306-323."""
        return serialize_example({
            'observations': _float_feature(np.asarray(obs).reshape(-1)),
            'class_label': _int64_feature([int(cls)]),
            'part_labels': _int64_feature(np.asarray(seg).reshape(-1)),
            'se3': _float_feature(np.asarray(se3).reshape(-1)),
        })

    def _parse_function(self, example_proto: bytes):
        """Decode one record and apply the jitter of PointCloudSet.py:325-345 (host / numpy version)."""
        ex = parse_example(example_proto)
        x = ex['observations'].reshape(self._network_input_width, 3).astype(np.float32)
        y_cls = np.int32(ex['class_label'][0])
        y_seg = ex['part_labels'].astype(np.int32)
        y_se3 = ex['se3'].reshape(3, 3).astype(np.float32)
        noise = self._rng.standard_normal(x.shape).astype(np.float32)
        x = x + noise * self._jitter_stdev_m
        return x, {'classification_output': y_cls, 'segmentation_output': y_seg, 'se3': y_se3}

    # ------------------------------------------------------------------------------------------------------
    def _load_split(self, split: str):
        files = sorted(glob.glob(f"{self._data_path}{self._name}/*/{split}_*.tfrecord"))
        xs, yc, ys, yr = [], [], [], []
        for fn in files:
            for rec in read_tfrecords(fn):
                ex = parse_example(rec)
                xs.append(ex['observations'].reshape(self._network_input_width, 3))
                yc.append(ex['class_label'][0])
                ys.append(ex['part_labels'])
                yr.append(ex['se3'].reshape(3, 3))
        if not xs:
            w = self._network_input_width
            return (np.zeros((0, w, 3), np.float32), np.zeros((0,), np.int32), np.zeros((0, w), np.int32), np.zeros((0, 3, 3), np.float32))
        return (np.stack(xs).astype(np.float32), np.asarray(yc, dtype=np.int32), np.stack(ys).astype(np.int32),
                np.stack(yr).astype(np.float32))

    def _batches(self, split: str, device=None, shuffle_buffer: int = 2048, rank: int = 0, world_size: int = 1):
        """Infinite generator: shuffle(2048) -> repeat -> jitter -> batch(B) (reference :347-391).  With a device the
        whole split lives in HBM and index shuffle, gather, jitter and batching run there."""
        import torch
        x, yc, ys, yr = self._load_split(split)
        n = x.shape[0]
        if n == 0:
            raise RuntimeError(f"PointCloudSet {self._name}: no {split} records under {self._data_path}{self._name}")
        dev = torch.device(device) if device is not None else torch.device("cpu")
        X, YC, YS, YR = (torch.from_numpy(a).to(dev) for a in (x, yc, ys, yr))
        sig = torch.from_numpy(self._jitter_stdev_m.astype(np.float32)).to(dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed((self._random_seed if self._random_seed is not None else int(self._rng.integers(1 << 31))) + 7919 * rank)
        B = self._batch_size

        def index_stream():
            # tf.data shuffle(buffer) over a repeating sequential stream, sharded by rank
            buf, pos = [], rank
            rng = np.random.default_rng(None if self._random_seed is None else self._random_seed + 104729 * rank)
            while True:
                while len(buf) < min(shuffle_buffer, n):
                    buf.append(pos % n)
                    pos += world_size
                j = int(rng.integers(len(buf)))
                buf[j], buf[-1] = buf[-1], buf[j]
                yield buf.pop()

        stream = index_stream()
        while True:
            idx = torch.as_tensor([next(stream) for _ in range(B)], device=dev, dtype=torch.long)
            xb = X[idx]
            if float(self._jitter_stdev_m.max()) > 0:
                xb = xb + torch.randn(xb.shape, device=dev, generator=gen) * sig
            yield xb.contiguous(), {'classification_output': YC[idx].contiguous(), 'segmentation_output': YS[idx].contiguous(),
                                    'se3': YR[idx].contiguous()}

    def get_train_set(self, device=None, rank: int = 0, world_size: int = 1):
        return self._batches('train', device, rank=rank, world_size=world_size)

    def get_val_set(self, device=None, rank: int = 0, world_size: int = 1):
        return self._batches('val', device, rank=rank, world_size=world_size)

    def get_test_set(self, device=None, rank: int = 0, world_size: int = 1):
        return self._batches('test', device, rank=rank, world_size=world_size)

    def get_description(self):
        return self._description

    def get_info(self):
        out = f'{self._description}\n'
        out += f'Random seed: {self._random_seed}\n' if (type(self._random_seed) == int) else f'Is not seeded\n'
        out += f'Class labels: {self._class_labels.keys()}\n'
        out += f'Part labels: {self._part_labels.keys()}\n'
        total = sum(self._data_size[s]['count'] for s in ('train', 'val', 'test'))
        for split, title, amt in (('train', 'Train Set', self._train_amt), ('val', 'Validation Set', self._val_amt),
                                  ('test', 'Test Set', self._test_amt)):
            out += f'\n--- {title} ---\n'
            out += f'Specified proportion:  {amt}\n'
            out += f"Actual proportion: {self._data_size[split]['count'] / total if total else float('nan')}\n"
            out += f"Total count: {self._data_size[split]['count']}\n"
            out += f'Class count:\n'
            for label in list(self._class_labels.keys()):
                if label in self._data_size[split]['class_count']:
                    out += f"\t{label}: {self._data_size[split]['class_count'][label]}\n"
            out += f'Part count:\n'
            for label in list(self._part_labels.keys()):
                if label in self._data_size[split]['part_count']:
                    out += f"\t{label}: {self._data_size[split]['part_count'][label]}\n"
        return out

    def _adjust_to_input_width(self, observations: np.ndarray, part_labels: np.ndarray) -> tuple:
        '''
        Keep the first _network_input_width points, or append uniformly drawn duplicates with their labels
        (reference PointCloudSet.py:443-470).
        '''
        w = self._network_input_width
        if observations.shape[0] > w:
            return observations[:w], part_labels[:w]
        repeated_indices = self._rng.uniform(0, observations.shape[0], w - observations.shape[0]).astype(np.int_)
        observations = np.concatenate((observations, deepcopy(observations[repeated_indices])), axis=0)
        assert observations.shape[0] == w, f'Failed to adjust observations to the network input width - should be {w}, not {observations.shape[0]}'
        part_labels = np.concatenate((part_labels, deepcopy(part_labels[repeated_indices])), axis=0)
        assert part_labels.shape[0] == w, f'Failed to adjust part_labels to the network input width - should be {w}, not {part_labels.shape[0]}'
        return observations, part_labels

    def _parse_state_info(self, filepath: str) -> dict:
        '''
        Parses the _palindrome_state_ log (reference PointCloudSet.py:472-509): header keys separated by three spaces,
        then per line "<time> <frame>" followed by 16 column-major floats per SE3 key; adds
        tanker_in_sensor_frame = [Rs^T Rt | Rs^T (tt - ts); 0 0 0 1] when both poses are present.
        '''
        with open(filepath, 'r') as f:
            keys = f.readline().strip().split("   ")
            keys = [i for i in keys if len(i) > 1]
            data: dict = {}
            for line in f:
                data_line = line.strip().split(" ")
                fr = int(data_line[1])
                data[fr] = {}
                data[fr][keys[0]] = data_line[0]
                data[fr][keys[1]] = data_line[1]
                for i, key in enumerate(keys[2:]):
                    cols = []
                    for col in range(constants.SE3_COLS):
                        a = 2 + i * constants.SE3_SIZE + col * constants.SE3_ROWS
                        cols.append(data_line[a: a + constants.SE3_ROWS])
                    data[fr][key] = np.array(cols, dtype=np.float64).T
                if 'Sensor Pose' in keys and 'Tanker Pose' in keys:
                    sp, tp = data[fr]['Sensor Pose'], data[fr]['Tanker Pose']
                    so3 = sp[:3, :3].T @ tp[:3, :3]
                    t_t_s = sp[:3, :3].T @ (tp[:3, 3:] - sp[:3, 3:])
                    se3_partial = np.concatenate([so3, t_t_s], axis=1)
                    data[fr]['tanker_in_sensor_frame'] = np.concatenate([se3_partial, np.array([[0, 0, 0, 1]])], axis=0)
        return data


### FREE HELPER FUNCTIONS ###
def load_from_file(joblib_file: str) -> PointCloudSet:
    if joblib_file.split(".")[-1] not in ('pkl', 'joblib'):
        joblib_file += '.pkl'
    with open(joblib_file, 'rb') as pf:
        pc_set: PointCloudSet = joblib.load(pf)
    return pc_set


def get_dir_contents(dir_path: str, _print: Callable[[str], None] = print) -> list:
    try:
        contents = os.listdir(dir_path)
        return contents if contents else []
    except FileNotFoundError:
        _print(f"Error: The directory '{dir_path}' was not found.")
    except NotADirectoryError:
        _print(f"Error: The path '{dir_path}' is not a directory.")
    except PermissionError:
        _print(f"Error: Permission denied to read '{dir_path}'.")
    except Exception as e:
        _print(f"An error occurred: {e}")
    return []
