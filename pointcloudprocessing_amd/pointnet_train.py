"""Drop-in for the reference's ``pointnet_train.py`` entry point on PyTorch-ROCm + libpointnet_hip.so.

    python -m pointcloudprocessing_amd.pointnet_train  <name>_config.json                     (one GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
           -m pointcloudprocessing_amd.pointnet_train  <name>_config.json                     (one process per GPU)

Same JSON schema, stage chaining, artefact layout and history keys as
``/root/reference/point_cloud_analysis/pointnet_train.py`` (TrainProfile :63-362, train_pointnet :364-402):

    <model_path><name>/log_<timestamp>.log
    <model_path><name>/<profile>/<name>_<profile>.pt            best checkpoint (the reference writes .keras)
    <model_path><name>/<profile>/<name>_<profile>_history.json  keys: loss, classification_output_loss,
        classification_output_sparse_categorical_accuracy, segmentation_output_loss,
        segmentation_output_sparse_categorical_accuracy, se3_loss, se3_root_mean_squared_error and val_*
    <model_path><name>/<profile>/<config copy>

Data parallelism (new; the reference is single-device): every rank holds a replica, draws its own batches,
runs forward + fused loss + backward natively, then ONE RCCL all-reduce sums the flat gradient buffer
(16.8 MB fp32) over xGMI and each rank applies the same Adam update.  BatchNormalization statistics are per
rank (standard DDP practice); the moving statistics are averaged across ranks at every epoch end.

Reference defects tolerated, not replicated: missing ``vanilla`` / ``monitor`` keys default
to False / ``val_loss``; an invalid path raises ``ValueError`` instead of returning a half-built object; the log
file name has no ':'; no stdin prompt when no GPU is present -- without a HIP device training refuses to start,
because there is no CPU compute path.  The ONNX export of every profile (pointnet_train.py:238-248) is written by
onnx_export.py (hand-encoded protobuf, opset 13); `continue_training_model` accepts such a file as well as a .pt checkpoint.
The .keras container (zip of config.json + HDF5 weights) is not written: it needs an HDF5 writer (h5py), absent here.
"""
from __future__ import annotations

import datetime
import contextlib
import json
import logging
import math
import os
import shutil
import signal
import sys
from typing import Callable, Optional

import numpy as np

try:                                                  # package use
    from .pointcloud import PointCloudSet as PointCloudSet
except ImportError:                                   # script use from inside the package directory, like the reference
    import pointcloud.PointCloudSet as PointCloudSet  # type: ignore

HISTORY_KEYS = ["loss", "classification_output_loss", "classification_output_sparse_categorical_accuracy",
                "segmentation_output_loss", "segmentation_output_sparse_categorical_accuracy", "se3_loss",
                "se3_root_mean_squared_error"]


class CtrlC_InterruptHandler:
    """First Ctrl-C: stop after the current epoch; second: exit (reference pointnet_train.py:42-61)."""

    def __init__(self, print_func: Callable[[str], None] = print):
        self._stop_requested = False
        self._print = print_func
        self.stop_training = False

    def stop_signaled(self, sig, frame):
        if not self._stop_requested:
            self._stop_requested = True
            self._print(">>> TRAINING INTERRUPT INITIATED BY USER <<<\nTraining will stop after the current epoch.\nPress Ctrl+C again to force quit.")
        else:
            self._print(">>> FORCE QUIT INITIATED BY USER <<<")
            sys.exit(0)

    def on_epoch_end(self, epoch, logs=None):
        if self._stop_requested:
            self._print("User stop received by the trainer.")
            self.stop_training = True


def exponential_decay(lr0, step, decay_steps, decay_rate):
    """keras ExponentialDecay(staircase=False) (pointnet_train.py:310-315)."""
    return lr0 * decay_rate ** (step / decay_steps)


# ----------------------------------------------------------------------------------------------------------
# engines: the thing that runs one training / evaluation step and accumulates the keras metrics
# ----------------------------------------------------------------------------------------------------------
class HipEngine:
    """PointNet on libpointnet_hip.so.  One process per GPU; gradients all-reduced over RCCL when world_size > 1."""

    def __init__(self, cfg: dict, n_class: int, n_part: int, profile: dict, log, checkpoint: Optional[str] = None,
                 precision: str = "bf16x3"):
        import torch
        import torch.distributed as dist
        from .optim import KerasAdam
        from .pointnet import PointNet as PN
        if not torch.cuda.is_available():
            raise RuntimeError("No HIP device available: the PointNet hot path has no CPU compute path")
        self.torch, self.dist = torch, dist
        self._log = log
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.device = torch.device("cuda", torch.cuda.current_device())
        p = cfg['params']
        if checkpoint and checkpoint.endswith(".onnx"):
            # an exported inference graph (onnx_export.py) keeps every raw parameter under its canonical name
            from .onnx_export import read_onnx_config, read_onnx_weights
            w = read_onnx_weights(checkpoint)
            # the model's own constructor arguments travel in the file (metadata_props), as payload['config'] does in a .pt checkpoint;
            # a file without them (written by another exporter) falls back to the training config and the reference's 0.3
            mc = dict(classification_output_width=int(w["mlp_cls_3.kernel"].shape[1]),
                      segmentation_output_width=int(w["mlp_seg_5.kernel"].shape[1]), dropout_rate=0.3,
                      random_seed=p['random_seed'], debugging=p.get('debugging', False), vanilla="input_transform.w" not in w,
                      regularize_input_transform=p.get('regularize_input_transform', False),
                      regularize_feature_transform=p.get('regularize_feature_transform', False))
            mc.update({k: v for k, v in read_onnx_config(checkpoint).items() if k in mc})
            mc["precision"] = precision
            self.model = PN.PointNet(**mc)
            self.model.set_weights({k: torch.from_numpy(v) for k, v in w.items()})
        elif checkpoint:
            payload = torch.load(checkpoint, map_location="cpu", weights_only=True)
            mc = dict(payload["config"])
            mc["precision"] = precision
            self.model = PN.PointNet.from_config(mc)
            self.model.set_weights(payload["weights"])
        else:
            # params.sync_batchnorm (not a reference key; default false): with more than one rank, every training-mode BatchNormalization
            # takes its statistics over the clouds of all ranks -- the reference's single-device batch (PointNet.py:528,559,623,647)
            sync = dict(sync_bn_world=self.world, sync_bn_rank=self.rank) if (p.get('sync_batchnorm', False) and self.world > 1) else {}
            self.model = PN.PointNet(classification_output_width=n_class, segmentation_output_width=n_part, dropout_rate=0.3,
                                     random_seed=p['random_seed'], debugging=p.get('debugging', False),
                                     vanilla=p.get('vanilla', False),
                                     regularize_input_transform=p.get('regularize_input_transform', False),
                                     regularize_feature_transform=p.get('regularize_feature_transform', False),
                                     precision=precision, **sync)
            self.model.build(input_shape=(None, p['input_width'], 3))
        if self.world > 1:                                  # identical replicas: rank 0's weights everywhere
            dist.broadcast(self.model.params_flat.data, src=0)
        m, t = self.model, profile['trainable']
        # same call order as pointnet_train.py:322-332
        (m.thaw_shared_network if t['shared_network'] else m.freeze_shared_network)()
        (m.thaw_input_transform if t['input_transform'] else m.freeze_input_transform)()
        (m.thaw_classification_head if t['classification_head'] else m.freeze_classification_head)()
        (m.thaw_segmentation_head if t['segmentation_head'] else m.freeze_segmentation_head)()
        lw = profile['loss_weights']
        self.loss_weights = (float(lw['classification']), float(lw['segmentation']), float(lw['rotation']))
        lr = p['learning']
        self.opt = KerasAdam(m.params_flat.data, lr['rate'], lr['decay_steps'], lr['decay_rate'])
        self.acc = torch.zeros(16, dtype=torch.float32, device=self.device)
        self.n_steps = 0
        self.B = None
        self.N = None
        self._steps = {}          # (B, N) -> engine.TrainStep (hipGraph replay of the whole step)
        self.stream = torch.cuda.Stream(device=self.device)   # every launch of this engine; see engine.TrainStep

    def model_config(self) -> dict:
        """the model's constructor arguments (what an exported .onnx keeps in its metadata_props); the arithmetic mode is the loader's choice"""
        return {k: v for k, v in self.model.get_config().items() if k != "precision"}

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def get_layer_trainability(self):
        return self.model.get_layer_trainability()

    def _targets(self, y):
        t = self.torch
        return (y['classification_output'].to(self.device, t.int32).contiguous(),
                y['segmentation_output'].to(self.device, t.int32).contiguous(),
                y['se3'].to(self.device, t.float32).contiguous())

    def reset_metrics(self):
        self.acc.zero_()
        self.n_steps = 0

    def train_step(self, x, y):
        from .engine import TrainStep
        self.B, self.N = x.shape[0], x.shape[1]
        yc, ys, yr = self._targets(y)
        key = (self.B, self.N)
        if key not in self._steps:
            self._steps[key] = TrainStep(self.model, self.opt, self.B, self.N, self.loss_weights, stream=self.stream)
        # forward + fused losses + backward (graph), one RCCL all-reduce of the flat 16.8 MB gradient buffer, Adam (graph)
        ts = self._steps[key]
        ts(x.to(self.device), yc, ys, yr)
        if ts.capture_error and not getattr(ts, "_fallback_logged", False):
            # the step could not be captured into a hipGraph and is launched kernel by kernel: correct, but several times slower
            ts._fallback_logged = True
            self._log.warning(f"hipGraph capture of the training step failed ({ts.capture_error}); steps run eagerly (slower, same results)")
        self.acc += self.model.scalars
        self.n_steps += 1

    def eval_step(self, x, y):
        x = x.to(self.device).contiguous()
        self.B, self.N = x.shape[0], x.shape[1]
        yc, ys, yr = self._targets(y)
        fused = dict(labels_cls=yc, labels_seg=ys.reshape(-1), se3=yr, loss_weights=self.loss_weights, keep=None)
        with self.torch.no_grad():
            self.model._run_forward(x, False, fused)
        self.acc += self.model.scalars
        self.n_steps += 1

    def metrics(self) -> dict:
        """keras history entries for the steps since reset_metrics(); averaged over ranks."""
        acc = self.acc.clone()
        if self.world > 1:
            self.dist.all_reduce(acc)
        a = acc.double().cpu().numpy() / max(self.world, 1)
        n, B, N = max(self.n_steps, 1), self.B or 1, self.N or 1
        cls_loss = a[0] / (n * B)
        seg_loss = a[2] / (n * B * N)
        mse = a[4] / (n * B * 9)
        reg = (a[5] + a[6]) / n
        w = self.loss_weights
        return {"loss": float(w[0] * cls_loss + w[1] * seg_loss + w[2] * mse + reg),
                "classification_output_loss": float(cls_loss),
                "classification_output_sparse_categorical_accuracy": float(a[1] / (n * B)),
                "segmentation_output_loss": float(seg_loss),
                "segmentation_output_sparse_categorical_accuracy": float(a[3] / (n * B * N)),
                "se3_loss": float(mse),
                "se3_root_mean_squared_error": float(math.sqrt(max(mse, 0.0)))}

    def sync_moving_statistics(self):
        if self.world > 1:
            flat = self.model.params_flat.data
            for s in self.model._weights.slots.values():
                if s["kind"] in (3, 4):
                    v = flat[s["offset"]: s["offset"] + s["rows"] * s["cols"]]
                    self.dist.all_reduce(v)
                    v /= self.world

    def get_weights(self):
        return {k: v.detach().cpu().clone() for k, v in self.model.named_weights().items()}

    def set_weights(self, w):
        self.model.set_weights(w)

    def save(self, path):
        if self.rank == 0:
            self.torch.save({"config": self.model.get_config(), "weights": self.get_weights()}, path)


class TrainProfile:
    def __init__(self, config_file, engine_factory: Optional[Callable] = None, max_steps_per_epoch: Optional[int] = None,
                 precision: str = "bf16x3", data_device="auto"):
        '''
        Reads the config, prepares the per-profile datasets and directories (reference pointnet_train.py:63-172).
        `engine_factory(cfg, n_class, n_part, profile_dict, log, checkpoint)` lets tests plug a different step engine.
        '''
        with open(config_file, 'r') as cf:
            config = json.load(cf)
        self._config = config
        self._config_file = config_file
        self._engine_factory = engine_factory
        self._max_steps = max_steps_per_epoch
        self._precision = precision
        self._data_device = data_device

        self._name: str = config['info']['name']
        self._class_labels: list = list(config['info']['class_labels'].values())
        self._part_labels: list = list(config['info']['part_labels'].values())
        self._training_profiles: dict = config['info']['training_profiles']
        self._pretrained_model: str = config['info'].get('continue_training_model', "")

        p = config['params']
        self._input_width: int = p['input_width']
        self._epochs: int = p['epochs']
        self._patience: int = p['patience']
        self._batch_size: int = p['batch_size']
        self._learning_rate: float = p['learning']['rate']
        self._learning_decay_steps: int = p['learning']['decay_steps']
        self._learning_decay_rate: float = p['learning']['decay_rate']
        self._random_seed: int = p['random_seed']
        self._debugging: bool = p.get('debugging', False)
        self._vanilla: bool = p.get('vanilla', False)            # absent from f15_lidar_config.json (reference :99 KeyErrors)
        self._reg_input_transform: bool = p.get('regularize_input_transform', False)
        self._reg_feature_transform: bool = p.get('regularize_feature_transform', False)

        self._model_path: str = config['file_system']['model_path']
        self._input_path: str = config['file_system']['input_path']
        self._data_path: str = config['file_system']['data_path']

        self._rank = int(os.environ.get("RANK", "0"))
        self._world = int(os.environ.get("WORLD_SIZE", "1"))

        for pth, what in ((self._model_path, "model_path"), (self._input_path, "input_path"), (self._data_path, "data_path")):
            if not os.path.isdir(pth):
                raise ValueError(f"Error in TrainProfile:  {what} {pth} does not exist")
        for prof in self._training_profiles:
            for ds in self._training_profiles[prof]['datasets'].values():
                have = os.path.isdir(f"{self._data_path}{self._name}_{prof}/{ds}")
                if not have and not os.path.isdir(f"{self._input_path}{ds}"):
                    raise ValueError(f"Error in TrainProfile:  {self._input_path}{ds} does not exist")
        if self._pretrained_model != "" and not os.path.isfile(f"{self._model_path}{self._pretrained_model}"):
            raise ValueError(f"Error in TrainProfile:  {self._model_path}{self._pretrained_model} does not exist")

        self._specific_model_path = f"{self._name}/"
        os.makedirs(f"{self._model_path}{self._specific_model_path}", exist_ok=True)

        dt = datetime.datetime.now()
        self._log = logging.getLogger(f"pointnet_train.{self._name}.{self._rank}")
        self._log.setLevel(logging.DEBUG)
        self._log.handlers = []
        self._log.propagate = False
        if self._rank == 0:
            console_handler = logging.StreamHandler()
            file_handler = logging.FileHandler(f"{self._model_path}{self._specific_model_path}log_{dt.strftime('%Y%m%d_%H%M%S')}.log")
            console_handler.setFormatter(logging.Formatter('%(name)s - %(levelname)s - %(message)s'))
            file_handler.setFormatter(logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s'))
            self._log.addHandler(console_handler)
            self._log.addHandler(file_handler)
        else:
            self._log.addHandler(logging.NullHandler())

        for prof in self._training_profiles:
            pdir = f"{self._data_path}{self._name}_{prof}"
            if os.path.isdir(pdir) and os.path.isfile(f"{pdir}/pc_set.joblib"):
                self._log.info(f"Training profile {self._name}_{prof} already exists. Using existing profile...")
                self._training_profiles[prof]['pc'] = PointCloudSet.load_from_file(f"{pdir}/pc_set.joblib")
                self._training_profiles[prof]['pc']._print = self._log.info
                self._training_profiles[prof]['pc']._data_path = self._data_path
            elif self._rank == 0:
                noise = self._training_profiles[prof]['noise']
                self._training_profiles[prof]['pc'] = PointCloudSet.PointCloudSet(
                    name=f"{self._name}_{prof}", class_labels=self._class_labels, part_labels=self._part_labels,
                    network_input_width=self._input_width,
                    jitter_stdev_m=np.array([noise['x_stdev_m'], noise['y_stdev_m'], noise['z_stdev_m']]),
                    batch_size=self._batch_size, rand_seed=42, description=prof, print_func=self._log.info,
                    data_path=self._data_path)
            if self._rank == 0:
                self._profile_datasets(prof)
            self._training_profiles[prof]['path'] = f"{self._specific_model_path}{prof}/"
            os.makedirs(f"{self._model_path}{self._training_profiles[prof]['path']}", exist_ok=True)

    # ------------------------------------------------------------------------------------------------------
    def _make_engine(self, prof: str, checkpoint: Optional[str]):
        pd = self._training_profiles[prof]
        if self._engine_factory is not None:
            return self._engine_factory(self._config, len(self._class_labels), len(self._part_labels), pd, self._log, checkpoint)
        return HipEngine(self._config, len(self._class_labels), len(self._part_labels), pd, self._log, checkpoint, self._precision)

    def train(self):
        '''
        Runs every training profile in order; each profile continues from the best checkpoint of the previous one
        (reference pointnet_train.py:174-257).
        '''
        import torch.distributed as dist
        distributed = dist.is_available() and dist.is_initialized()
        for prof in list(self._training_profiles.keys()):
            pd = self._training_profiles[prof]
            ckpt = f"{self._model_path}{self._pretrained_model}" if self._pretrained_model != "" else None
            if ckpt:
                self._log.info(f"Continuing training on model {self._pretrained_model}")
            if distributed:
                dist.barrier()                      # rank 0 has finished writing the dataset / previous checkpoint
                if 'pc' not in pd:
                    pdir = f"{self._data_path}{self._name}_{prof}"
                    pd['pc'] = PointCloudSet.load_from_file(f"{pdir}/pc_set.joblib")
                    pd['pc']._data_path = self._data_path
            engine = self._make_engine(prof, ckpt)

            self._log.info(f"PointNet Build")
            self._log.info(f"\tTrainable Layers")
            for l, v in engine.get_layer_trainability().items():
                self._log.info(f"\t\t-> {l}: {v}")

            keyboard_interrupt = CtrlC_InterruptHandler(print_func=self._log.info)
            try:
                signal.signal(signal.SIGINT, keyboard_interrupt.stop_signaled)
            except ValueError:                      # not the main thread
                pass

            monitor = pd.get('monitor', 'val_loss')
            best, best_weights, wait = math.inf, None, 0
            history = {k: [] for k in HISTORY_KEYS + ["val_" + k for k in HISTORY_KEYS]}
            ckpt_path = f"{self._model_path}{pd['path']}{self._name}_{prof}.pt"

            pc = pd['pc']
            dev = getattr(engine, "device", None) if self._data_device == "auto" else self._data_device
            train_it = pc.get_train_set(device=dev, rank=self._rank, world_size=self._world)
            val_it = pc.get_val_set(device=dev, rank=self._rank, world_size=self._world)
            steps = int(pc._data_size['train']['count'] / self._batch_size) // self._world
            vsteps = int(pc._data_size['val']['count'] / self._batch_size) // self._world
            if self._max_steps:
                steps, vsteps = min(steps, self._max_steps), min(vsteps, self._max_steps)
            steps, vsteps = max(steps, 1), max(vsteps, 1)

            # the whole loop (data pipeline, steps, metric reads) runs on the engine's stream: no per-step cross-stream fences
            with getattr(engine, 'stream_context', contextlib.nullcontext)():
                for epoch in range(self._epochs):
                    engine.reset_metrics()
                    for _ in range(steps):
                        x, y = next(train_it)
                        engine.train_step(x, y)
                    logs = engine.metrics()
                    engine.sync_moving_statistics()
                    engine.reset_metrics()
                    for _ in range(vsteps):
                        x, y = next(val_it)
                        engine.eval_step(x, y)
                    logs.update({"val_" + k: v for k, v in engine.metrics().items()})
                    for k in history:
                        history[k].append(logs[k])
                    self._log.info(f"Epoch {epoch + 1}/{self._epochs} - " + " - ".join(f"{k}: {logs[k]:.4f}" for k in history))

                    cur = logs.get(monitor, logs["val_loss"])
                    if cur < best:                      # ModelCheckpoint(save_best_only) + EarlyStopping bookkeeping, mode='min'
                        self._log.info(f"Epoch {epoch + 1}: {monitor} improved from {best:.5f} to {cur:.5f}, saving model to {ckpt_path}")
                        best, wait = cur, 0
                        best_weights = engine.get_weights()
                        engine.save(ckpt_path)
                    else:
                        wait += 1
                        if wait >= self._patience:
                            self._log.info(f"Epoch {epoch + 1}: early stopping")
                            break
                    keyboard_interrupt.on_epoch_end(epoch, logs)
                    # a Ctrl-C reaches the ranks at different times: every rank must take the same branch here, or one leaves for the
                    # barrier below while the others enter the next epoch's all-reduce and the job hangs
                    if self._any_rank(keyboard_interrupt.stop_training, engine):
                        break
                if best_weights is not None:
                    engine.set_weights(best_weights)    # EarlyStopping(restore_best_weights=True)

            if self._rank == 0:
                with open(f"{self._model_path}{pd['path']}{self._name}_{prof}_history.json", 'w') as j:
                    json.dump(history, j)
                # reference :238-248: tf2onnx.convert.from_keras(model, input_signature=[(None, input_width, 3)], opset=13) of the restored
                # best weights -> <name>_<prof>.onnx.  Written by onnx_export.py (hand-encoded protobuf; numeric parity unpinned).
                if best_weights is not None:
                    try:
                        from .onnx_export import export_onnx
                        onnx_path = f"{self._model_path}{pd['path']}{self._name}_{prof}.onnx"
                        export_onnx({k: np.asarray(v, dtype=np.float32) for k, v in best_weights.items()}, self._input_width, onnx_path,
                                    vanilla=not any(k.startswith("input_transform.") for k in best_weights),
                                    config=engine.model_config() if hasattr(engine, "model_config") else None)
                        self._log.info(f"ONNX model (opset 13) written to {onnx_path}")
                    except Exception as e:            # the reference logs and carries on as well (:246-248)
                        self._log.info(f"ONNX export failed: {e}")
                shutil.copy(self._config_file, f"{self._model_path}{pd['path']}")
            self._pretrained_model = f"{pd['path']}{self._name}_{prof}.pt"
            if distributed:
                dist.barrier()
        return True

    @staticmethod
    def _any_rank(flag: bool, engine) -> bool:
        """logical OR of a per-rank flag over the process group (MAX all-reduce); the flag itself without one"""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return bool(flag)
        dev = getattr(engine, "device", None)
        t = torch.tensor([1.0 if flag else 0.0], device=dev if dev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0)

    def _profile_datasets(self, profile) -> None:
        pd = self._training_profiles[profile]
        datasets = PointCloudSet.get_dir_contents(f"{self._data_path}{self._name}_{profile}", self._log.info)
        wanted = list(pd['datasets'].values())
        if len(datasets) > 0:
            self._log.info(f"The following datasets were found in {self._data_path}{self._name}_{profile}:")
            for ds in datasets:
                self._log.info(f"\t-> {ds}\t{'' if ds in wanted or ds == 'pc_set.joblib' else '(not requested, but will be included in training profile)'}")
        for ds, set_name in enumerate(wanted):
            if set_name not in datasets:
                self._log.info(f"Adding data set {ds + 1} of {len(wanted)}")
                pd['pc'].add_from_aftr_output(dir_path=f"{self._input_path}{set_name}", shuffle_points=True)
        self._log.info('\nDatasets added successfully:\n')
        self._log.info(pd['pc'].get_info())


def init_distributed():
    """One process per GPU under torchrun: RCCL ('nccl' on ROCm) over xGMI; gloo when there is no GPU (tests)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return
    if torch.cuda.is_available():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend="gloo")


def train_pointnet(*args, **kwargs) -> bool:
    configs = [i for i in args[0] if i.split('.')[-1] == 'json']
    if len(configs) == 0 or "-h" in args[0] or "--help" in args[0]:
        print_help()
        return False
    import torch
    if torch.cuda.is_available():
        print('GPUs Available: ', torch.cuda.device_count())
    elif kwargs.get("engine_factory") is None:
        print("No HIP device available: the PointNet hot path has no CPU compute path.")
        return False
    init_distributed()
    for cf in configs:
        tp = TrainProfile(cf, **kwargs)
        tp.train()
    return True


def print_help():
    print('''PointNet Training Module (MI355X)

        Trains new or pretrained PointNet models.  The configuration file follows the reference's
        examples/train_config_template.json; the file name MUST end in {somename}_config.json:
        {
        \tinfo: { name, class_labels{}, part_labels{}, training_profiles{ <profile>: { datasets{}, noise{x,y,z}_stdev_m,
        \t        trainable{shared_network,input_transform,classification_head,segmentation_head},
        \t        loss_weights{classification,segmentation,rotation}, monitor } }, continue_training_model },
        \tparams: { input_width, epochs, patience, batch_size, learning{rate,decay_steps,decay_rate}, random_seed,
        \t          debugging, vanilla, regularize_input_transform, regularize_feature_transform },
        \tfile_system: { model_path, input_path, data_path }
        }''')


if __name__ == '__main__':
    if not any([i.split('_')[-1] == 'config.json' for i in sys.argv]):
        sys.argv = ['vizzer_config.json']
        print(f"No config file found. Defaulting to: {sys.argv}")
    if train_pointnet(sys.argv):
        print("Model training completed successfully.")
    else:
        print("Model training failed.")
