"""CLIP text encoder with the reference's interface and weight ABI (sd/clip.py:7-261): 12 pre-norm layers,
12 heads, causal mask, quick-GELU ``x * sigmoid(1.702 x)``, final LayerNorm.
Native HIP only (csrc/clip.hip); the CPU restatement used as the checker lives in oracle/aux_ref.py."""
from __future__ import annotations

import torch

from . import arch
from .vae import _StateModule


class CLIP(_StateModule):
    """Hand-written HIP kernels through libsdmi (csrc/clip.hip), cuda device required, no fallback."""

    def __init__(self):
        super().__init__(arch.clip_manifest())
        self._handle = None

    def load_state_dict(self, state, strict: bool = True):
        super().load_state_dict(state, strict)
        self._drop()
        return self

    def to(self, device):
        before = self._device
        super().to(device)
        if self._device != before:
            self._drop()
        return self

    def _drop(self):
        if self._handle is not None:
            self._handle.close()
        self._handle = None

    def handle(self):
        from . import _native
        if self._handle is None:
            if self._device.type != "cuda":
                raise RuntimeError("CLIP needs a cuda (ROCm) device: the native HIP path has no CPU fallback")
            with torch.cuda.device(self._device):
                self._handle = _native.ClipHandle(self._state)
        return self._handle

    @torch.no_grad()
    def __call__(self, tokens: torch.Tensor) -> torch.Tensor:
        return self.handle().encode(tokens.to(self._device).type(torch.long))

    forward = __call__
