"""CLIP text encoder with the reference's interface and weight ABI (sd/clip.py:7-261) -- INTERIM
PyTorch-ROCm implementation (13 GFLOP twice per image, <1 % of one UNet step).
12 pre-norm layers, 12 heads, causal mask, quick-GELU ``x * sigmoid(1.702 x)``, final LayerNorm."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import arch
from .vae import _StateModule


class CLIP(_StateModule):
    def __init__(self):
        super().__init__(arch.clip_manifest())

    @torch.no_grad()
    def __call__(self, tokens: torch.Tensor) -> torch.Tensor:
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
