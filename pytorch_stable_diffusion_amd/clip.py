"""CLIP text encoder with the reference's interface and weight ABI (sd/clip.py:7-261): 12 pre-norm layers,
12 heads, causal mask, quick-GELU ``x * sigmoid(1.702 x)``, final LayerNorm.
Default backend: native HIP (csrc/clip.hip).  The torch-op restatement below is the explicit ``backend="torch"``
path used by CPU unit tests."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import arch
from .vae import _StateModule


class CLIP(_StateModule):
    """``backend="native"`` (default): hand-written HIP kernels through libsdmi (csrc/clip.hip), cuda device
    required, no fallback.  ``backend="torch"``: explicit opt-in to the PyTorch-op restatement (CPU unit tests)."""

    def __init__(self, backend: str = "native"):
        super().__init__(arch.clip_manifest())
        if backend not in ("native", "torch"):
            raise ValueError(f"unknown backend {backend}")
        self.backend = backend
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
                raise RuntimeError("CLIP(backend='native') needs a cuda (ROCm) device; there is no CPU fallback "
                                   "(use backend='torch' explicitly for a PyTorch-op restatement)")
            with torch.cuda.device(self._device):
                self._handle = _native.ClipHandle(self._state)
        return self._handle

    @torch.no_grad()
    def __call__(self, tokens: torch.Tensor) -> torch.Tensor:
        if self.backend == "native":
            return self.handle().encode(tokens.to(self._device).type(torch.long))
        sd = self._state
        tokens = tokens.to(self._device).type(torch.long)
        x = F.embedding(tokens, sd["embedding.token_embedding.weight"]) + sd["embedding.position_embedding"]
        b, s, c = x.shape
        hd = c // arch.CLIP_HEADS
        mask = torch.ones((s, s), dtype=torch.bool, device=x.device).triu(1)
        for i in range(arch.CLIP_LAYERS):
            p = f"layers.{i}"
            h = F.layer_norm(x, (c,), sd[f"{p}.layernorm_1.weight"], sd[f"{p}.layernorm_1.bias"])
            qkv = F.linear(h, sd[f"{p}.attention.in_proj.weight"], sd[f"{p}.attention.in_proj.bias"])
            q, k, v = (t.reshape(b, s, arch.CLIP_HEADS, hd).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
            w = (q @ k.transpose(-1, -2)).masked_fill(mask, float("-inf")) / math.sqrt(hd)   # mask before scale
            o = (torch.softmax(w, dim=-1) @ v).transpose(1, 2).reshape(b, s, c)
            x = x + F.linear(o, sd[f"{p}.attention.out_proj.weight"], sd[f"{p}.attention.out_proj.bias"])
            h = F.layer_norm(x, (c,), sd[f"{p}.layernorm_2.weight"], sd[f"{p}.layernorm_2.bias"])
            h = F.linear(h, sd[f"{p}.linear_1.weight"], sd[f"{p}.linear_1.bias"])
            h = h * torch.sigmoid(1.702 * h)
            x = x + F.linear(h, sd[f"{p}.linear_2.weight"], sd[f"{p}.linear_2.bias"])
        return F.layer_norm(x, (c,), sd["layernorm.weight"], sd["layernorm.bias"])

    forward = __call__
