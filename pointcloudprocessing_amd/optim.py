"""keras.optimizers.Adam + ExponentialDecay (reference pointnet_train.py:310-319) on the model's flat buffers.

One native launch per step (schedule + update fused, pn_adam_step); the step counter and step size live on the
device, so a training step captured in a hipGraph replays with the right learning rate.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import check, current_stream, lib, ptr, require_gpu_tensor


class KerasAdam:
    def __init__(self, params_flat: torch.Tensor, learning_rate: float, decay_steps: float = 1.0, decay_rate: float = 1.0,
                 beta_1: float = 0.9, beta_2: float = 0.999, epsilon: float = 1e-7):
        require_gpu_tensor(params_flat, "params_flat", torch.float32)
        self.params = params_flat
        self.m = torch.zeros_like(params_flat)
        self.v = torch.zeros_like(params_flat)
        self.iterations = torch.zeros(1, dtype=torch.int32, device=params_flat.device)
        self._alpha = torch.zeros(4, dtype=torch.float32, device=params_flat.device)   # [step size, learning rate (current), ticket, -]
        self.lr0, self.decay_steps, self.decay_rate = float(learning_rate), float(decay_steps), float(decay_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.prepare()

    def prepare(self):
        """(re)evaluate the schedule for the current ``iterations``: at construction and after restoring a checkpoint"""
        check(lib().pn_adam_prepare(ptr(self.iterations), ptr(self._alpha), self.lr0, self.decay_rate, self.decay_steps, self.beta_1,
                                    self.beta_2, current_stream()), "pn_adam_prepare")

    def load_state_dict(self, state):
        self.m.copy_(state["m"]); self.v.copy_(state["v"]); self.iterations.copy_(state["iterations"])
        self.prepare()

    def step(self, grads_flat: torch.Tensor, grad_scale: float = 1.0, lo: int = 0, hi: int = None):
        """one Adam step over the flat range [lo, hi) (default: everything).  A caller restricts the range to the extent of the
        trainable blocks (PointNet.grad_extent): elements with zero gradient and zero moments do not move"""
        if lo or (hi is not None and hi != self.params.numel()):
            p, g, m, v = self.params[lo:hi], grads_flat[lo:hi], self.m[lo:hi], self.v[lo:hi]
            check(lib().pn_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(self.iterations), ptr(self._alpha), self.lr0,
                                     self.decay_rate, self.decay_steps, self.beta_1, self.beta_2, self.epsilon, float(grad_scale),
                                     current_stream()), "pn_adam_step")
            return
        check(lib().pn_adam_step(ptr(self.params), ptr(grads_flat), ptr(self.m), ptr(self.v), self.params.numel(),
                                 ptr(self.iterations), ptr(self._alpha), self.lr0, self.decay_rate, self.decay_steps, self.beta_1,
                                 self.beta_2, self.epsilon, float(grad_scale), current_stream()), "pn_adam_step")

    @property
    def learning_rate(self) -> float:
        return float(self._alpha[1].item())

    def state_dict(self):
        return {"m": self.m, "v": self.v, "iterations": self.iterations}
