"""Training-step executor: forward + fused Keras losses + backward + (RCCL all-reduce) + Adam, replayed from hipGraphs.

A PointNet training step is ~200 short kernels; launched eagerly the host (ctypes + hipLaunchKernel, ~7 us each) is
the bottleneck.  The native plan never allocates or synchronises, so the whole step is captured once per input shape
and replayed.  One GPU: a single graph (dropout masks, forward, losses, backward, Adam).  Data parallel: graph 1 = dropout masks,
forward, losses and the backward pass down to the feature transform; then the RCCL all-reduce of that gradient bucket (13.5 of
16.8 MB) is issued asynchronously and graph 1b = the rest of the backward pass (mlp_1, input transform) runs under it; then the
small second bucket is reduced and graph 2 = Adam.

Everything a TrainStep launches goes to ITS OWN HIP stream (created here), fenced against the caller's stream with
events on entry and exit: the step stays off the legacy null stream's implicit synchronisation, and several
TrainSteps (models) in one process do not serialise on each other.

Graph hygiene (measured, ROCm 7.2): the native plan contains kernel nodes only.  A hipMemsetAsync of the 16 MiB
gradient buffer captured as a memset node replayed with a garbage fill pattern as soon as another model launched
work between two replays, so the library clears buffers with its own zero-fill kernel (pn_optim.hip:zero_fill);
tests/test_gpu_train.py::test_interleaved_models_graph_replay_is_exact keeps that pinned.
"""
from __future__ import annotations

from typing import Optional, Sequence

import os

import torch

from ._lib import check, current_stream, lib, ptr
from .optim import KerasAdam


def rank_salt() -> int:
    import torch.distributed as dist
    return dist.get_rank() + 1 if dist.is_available() and dist.is_initialized() else 1


class TrainStep:
    def __init__(self, model, optimizer: KerasAdam, batch: int, points: int, loss_weights: Sequence[float], use_graph: bool = True,
                 stream: Optional["torch.cuda.Stream"] = None, aux: bool = False,
                 split_optimizer: Optional[bool] = None):
        import torch.distributed as dist
        self.model, self.opt = model, optimizer
        self.B, self.N = batch, points
        self.lw = tuple(float(w) for w in loss_weights)
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # data-parallel layout of a step: [graph 1: forward+backward] -> all-reduce -> [graph 2: Adam]; a single process fuses
        # both graphs unless split_optimizer asks for the data-parallel layout anyway (rehearsal of the N > 1 path on one GPU)
        self.split = (self.world > 1) if split_optimizer is None else bool(split_optimizer)
        self.reduce = self.split and dist.is_available() and dist.is_initialized()
        # PN_DDP_OVERLAP=0: one all-reduce of the whole buffer after the backward pass instead of the two overlapped buckets
        self.overlap = os.environ.get("PN_DDP_OVERLAP", "1") != "0"
        dev = model.params_flat.device
        self.dev = dev
        # Synchronised BatchNormalization (PointNet(sync_bn_world=W)): the native plan exchanges statistics and the dense layers' rows
        # INSIDE the forward / backward pass (pn_model_io.sync_hook), so the step is launched eagerly, in one piece; the gradients are
        # summed (not averaged: every rank already seeds the global mean loss) after the slots every rank computed in full have been
        # zeroed on every rank but the first.
        self.sync_bn = getattr(model, "_sync_world", 1) > 1
        rows = batch * (model._sync_world if self.sync_bn else 1)
        if self.sync_bn:
            self.split, use_graph = False, False
            self._rep_mask = model.replicated_grad_mask() if self.dist.get_rank(model._sync_group) != 0 else None
        # static inputs: a graph replays fixed addresses
        self.pc = torch.zeros(batch, points, 3, device=dev)
        self.y_cls = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.y_seg = torch.zeros(batch, points, dtype=torch.int32, device=dev)
        self.se3 = torch.zeros(batch, 3, 3, device=dev)
        self.keep = (torch.ones(rows, 512, dtype=torch.uint8, device=dev), torch.ones(rows, 256, dtype=torch.uint8, device=dev))
        self._mask_seed = int(torch.randint(0, 2**62, (1,)).item()) ^ (rank_salt() << 20)   # per process / rank
        if self.sync_bn:                # one mask stream for the whole batch: every rank draws the rows of all ranks
            t = torch.tensor([self._mask_seed & (2**62 - 1)], dtype=torch.int64, device=dev if dev.type == "cuda" else "cpu")
            self.dist.broadcast(t, src=self.dist.get_global_rank(model._sync_group, 0) if model._sync_group is not None else 0, group=model._sync_group)
            self._mask_seed = int(t.item())
        self._mask_step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.mode = "eager"
        self.capture_error = None     # set when hipGraph capture failed and the step fell back to eager launches (the trainer logs it)
        self._g1 = self._g1b = self._g2 = None
        # A host-memory model has no streams or graphs: only the step SEQUENCE runs (forward/backward phases, the bucketed
        # all-reduce schedule, the optimizer).  The product never builds one -- PointNet refuses to compute without a HIP device --
        # but it lets tests/test_cpu_train.py drive this very schedule over gloo with world_size 2.
        self.on_gpu = dev.type == "cuda"
        self.stream = self._capture_stream = self.aux_stream = None
        if self.on_gpu:
            self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
            # Capture runs on a stream of its own that never carries anything else (a graph replays on any stream).  RCCL's watchdog
            # thread keeps calling hipEventQuery on the end events of the warm-up steps' collectives until it retires them; a
            # synchronous collective records its end event on the stream it was issued on (self.stream), and HIP refuses a query of an
            # event whose stream is capturing (hipErrorCapturedEvent -> the watchdog aborts the process).  No collective is ever issued
            # on the capture stream, so no such event exists on it.
            self._capture_stream = torch.cuda.Stream(device=dev)
            # optional second stream for the parameter-gradient kernels of the backward pass (pn_model_io.aux_stream).
            # Bit-identical results; measured at B=32, N=1024 it does not pay on ROCm 7.2 (graph cross-branch edges cost more
            # than the overlap wins: 1.52-1.68 ms vs 1.54 ms/step), so it is off by default.
            self.aux_stream = torch.cuda.Stream(device=dev) if aux else None
        use_graph = use_graph and self.on_gpu
        self._use_graph = use_graph
        self._calls = 0           # the first two steps run eagerly (they warm up allocator / lazy init), then the step is captured

    # -- pieces ---------------------------------------------------------------------------------------------
    def _fwd_bwd(self, phase: int = 0):
        rate = self.model._dropout_rate
        # the dropout masks are drawn by the forward pass's first launch (pn_model_io.dropout_step): the step counter lives on the
        # device, so graph replays draw fresh masks
        self.model._aux_stream = self.aux_stream
        try:
            self.model.fused_loss_step(self.pc, self.y_cls, self.y_seg, self.se3, self.lw, keep=self.keep if rate > 0 else None,
                                       backward_phase=phase,
                                       dropout_rng=(self._mask_seed, self._mask_step) if (rate > 0 and self.on_gpu) else None)
        finally:
            self.model._aux_stream = None

    def _bwd2(self):
        self.model._aux_stream = self.aux_stream
        try:
            self.model._run_backward(None, None, None, 2)
        finally:
            self.model._aux_stream = None

    def _reduce_async(self, lo, hi, last=False):
        """RCCL sum over xGMI of one gradient bucket; runs on RCCL's stream behind everything enqueued so far"""
        if not self.reduce:
            return None
        if not self.overlap:
            if not last:
                return None                                   # single collective: issued with the second (last) bucket, even an empty one
            # a synchronous collective: recent PyTorch runs it on the CURRENT stream (no cross-stream fence at all)
            a, _, b = self._buckets()
            self.dist.all_reduce(self.model.grads_flat[a:b] if b > a else self.model.grads_flat)
            return None
        if hi <= lo:
            return None
        return self.dist.all_reduce(self.model.grads_flat[lo:hi], async_op=True)

    def _wait_last(self, *handles):
        """the collectives of one process group run in order on one RCCL stream: waiting for the last issued one is enough, and
        every cross-stream wait costs ~50-100 us on this stack.  Other backends (gloo, in the CPU tests) complete their asynchronous
        collectives in any order: there every handle is waited for."""
        live = [h for h in handles if h is not None]
        if not live:
            return
        if self.dist.get_backend() == "nccl":
            live[-1].wait()
        else:
            for h in live:
                h.wait()

    def _buckets(self):
        """(lo, cut, hi): bucket 1 = [cut, hi) is final after backward phase 1, bucket 2 = [lo, cut) after phase 2; [lo, hi) = the
        extent of the trainable blocks' gradients (frozen blocks outside it carry zeros on every rank: not reduced, not stepped)"""
        n = self.model.grads_flat.numel()
        lo, hi = self.model.grad_extent() if hasattr(self.model, "grad_extent") else (0, n)
        cut = min(max(self.model.grad_bucket_boundary(), lo), hi)
        return lo, cut, hi

    def _opt_step(self, scale):
        lo, _, hi = self._buckets()
        if (lo, hi) == (0, self.model.grads_flat.numel()) or hi <= lo:
            self.opt.step(self.model.grads_flat, scale)
        else:
            self.opt.step(self.model.grads_flat, scale, lo, hi)

    def _eager(self):
        if self.sync_bn:
            self._fwd_bwd()
            g = self.model.grads_flat
            if self._rep_mask is not None:
                g.mul_(self._rep_mask)
            lo, _, hi = self._buckets()
            self.dist.all_reduce(g[lo:hi] if hi > lo else g, group=self.model._sync_group)
            self._opt_step(1.0)
            return
        if not self.split:
            self._fwd_bwd()
        else:
            # bucket 1 (feature transform .. heads, ~80 % of the bytes) is reduced while mlp_1 / the input transform run backward
            lo, cut, end = self._buckets()
            self._fwd_bwd(1)
            h1 = self._reduce_async(cut, end)
            self._bwd2()
            h2 = self._reduce_async(lo, cut, last=True)
            self._wait_last(h1, h2)
        self._opt_step(1.0 / self.world)

    def _capture(self):
        # capture_error_mode="thread_local": RCCL's watchdog thread polls its work events with hipEventQuery all the time; under the
        # default "global" mode that query is illegal while THIS thread captures and the watchdog aborts the process
        # (c10::DistBackendError "operation not permitted when stream is capturing" -- seen 2 times in 14 data-parallel rehearsals).
        try:
            torch.cuda.synchronize()
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, stream=self._capture_stream, capture_error_mode="thread_local"):
                self._fwd_bwd(1 if self.split else 0)
                if not self.split:
                    self._opt_step(1.0)
            g1b = g2 = None
            if self.split:
                g1b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1b, stream=self._capture_stream, capture_error_mode="thread_local"):
                    self._bwd2()
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, stream=self._capture_stream, capture_error_mode="thread_local"):
                    self._opt_step(1.0 / self.world)
            torch.cuda.synchronize()
            self._g1, self._g1b, self._g2, self.mode = g1, g1b, g2, "hipgraph"
        except Exception as e:                                  # capture unsupported: stay eager (a speed matter only)
            import sys
            self.capture_error = f"{type(e).__name__}: {e}"
            print(f"# hipGraph capture failed ({self.capture_error}); running the step eagerly", file=sys.stderr)
            self._g1 = self._g1b = self._g2 = None
            self.mode = "eager"

    # -- API ------------------------------------------------------------------------------------------------
    # A cross-stream event fence costs ~100 us per step on this stack (1.73 vs 1.54 ms/step at B=32, N=1024), so the
    # fences exist only for callers on another stream; the trainer and bench.py run their loops under
    # ``torch.cuda.stream(step.stream)`` and pay nothing.
    def _enter(self):
        if not self.on_gpu:
            return
        cur = torch.cuda.current_stream(self.dev)
        if cur.cuda_stream != self.stream.cuda_stream:
            self.stream.wait_stream(cur)                                 # whatever produced the batch / read the last results

    def _exit(self):
        if not self.on_gpu:
            return
        cur = torch.cuda.current_stream(self.dev)
        if cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_stream(self.stream)                                 # the caller may read scalars / weights on its stream

    def _on_stream(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.on_gpu else contextlib.nullcontext()

    def load(self, pc, y_cls, y_seg, se3):
        self._enter()
        with self._on_stream():
            self.pc.copy_(pc, non_blocking=True)
            self.y_cls.copy_(y_cls, non_blocking=True)
            self.y_seg.copy_(y_seg.reshape(self.B, self.N), non_blocking=True)
            self.se3.copy_(se3, non_blocking=True)
        for t in (pc, y_cls, y_seg, se3):
            if t.is_cuda and self.on_gpu:
                t.record_stream(self.stream)

    def run(self):
        """one training step on the currently loaded batch; results in model.scalars / model.grads_flat"""
        self._calls += 1
        if self._use_graph and self._g1 is None and self._calls == 3:
            self._capture()                                     # capture only records; the replay below executes step 3
            self._use_graph = self._g1 is not None
        self._enter()
        with self._on_stream():
            if self._g1 is None:
                self._eager()
            else:
                self._g1.replay()
                if self.split:
                    lo, cut, end = self._buckets()
                    h1 = self._reduce_async(cut, end)           # overlaps graph 1b
                    self._g1b.replay()
                    h2 = self._reduce_async(lo, cut, last=True)
                    self._wait_last(h1, h2)
                    self._g2.replay()
        self._exit()

    def run_eager(self):
        """the same step launched kernel by kernel (bench.py times the dominant kernel with HIP events this way)"""
        self._enter()
        with self._on_stream():
            self._eager()
        self._exit()

    def __call__(self, pc, y_cls, y_seg, se3):
        self.load(pc, y_cls, y_seg, se3)
        self.run()
