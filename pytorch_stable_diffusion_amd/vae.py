"""VAE encoder / decoder with the reference's interface and weight ABI (sd/encoder.py:8-155,
sd/decoder.py:7-374); SURVEY 8f "next" rows.  Default backend: native HIP (csrc/vae.hip, the UNet's
implicit-GEMM / GroupNorm kernels plus a row-softmax).  The PyTorch-op restatement in this file is the explicit
``backend="torch"`` path used by CPU unit tests.  The reference's behaviour is reproduced, including its quirks:
  Q3  VAE_AttentionBlock never applies its GroupNorm          (sd/decoder.py:31,34-73)
  Q4  ``x.transpose(-1, 2)`` is a no-op on a 3-D tensor and the (n, h*w, c) attention output is
      REINTERPRETED as (n, c, h, w) by ``view``                (sd/decoder.py:62,67)
  in-place ``x /= 0.18215`` on the caller's latents            (sd/decoder.py:364)
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict

import torch
import torch.nn.functional as F

from . import arch


class _StateModule:
    """Minimal nn.Module-like holder: state_dict / load_state_dict(strict) / to / __call__."""

    def __init__(self, manifest):
        self._manifest = manifest
        self._state: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._device = torch.device("cpu")

    def state_dict(self):
        return OrderedDict(self._state)

    def load_state_dict(self, state: Dict[str, torch.Tensor], strict: bool = True):
        missing = [k for k in self._manifest if k not in state]
        unexpected = [k for k in state if k not in self._manifest]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: missing {missing[:5]} "
                               f"unexpected {unexpected[:5]}")
        for k, shape in self._manifest.items():
            if k in state:
                if tuple(state[k].shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(state[k].shape)} vs {tuple(shape)}")
                self._state[k] = state[k].detach().to(self._device)
        return self

    def to(self, device):
        device = torch.device(device)
        if device != self._device:
            for k in list(self._state.keys()):
                self._state[k] = self._state[k].to(device)
            self._device = device
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(self._state.values())


def _res(sd, p, x):
    # VAE_ResidualBlock (sd/decoder.py:148-190)
    h = F.group_norm(x, 32, sd[f"{p}.groupnorm_1.weight"], sd[f"{p}.groupnorm_1.bias"])
    h = F.conv2d(F.silu(h), sd[f"{p}.conv_1.weight"], sd[f"{p}.conv_1.bias"], padding=1)
    h = F.group_norm(h, 32, sd[f"{p}.groupnorm_2.weight"], sd[f"{p}.groupnorm_2.bias"])
    h = F.conv2d(F.silu(h), sd[f"{p}.conv_2.weight"], sd[f"{p}.conv_2.bias"], padding=1)
    w = sd.get(f"{p}.residual_layer.weight")
    return h + (x if w is None else F.conv2d(x, w, sd[f"{p}.residual_layer.bias"]))


def _attn(sd, p, x):
    # VAE_AttentionBlock (sd/decoder.py:34-73) with quirks Q3 (no groupnorm) and Q4 (reinterpreting view)
    n, c, h, w = x.shape
    t = x.reshape(n, c, h * w).transpose(1, 2)                       # (n, hw, c)
    qkv = F.linear(t, sd[f"{p}.attention.in_proj.weight"], sd[f"{p}.attention.in_proj.bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(c), dim=-1) @ v          # single head
    o = F.linear(a, sd[f"{p}.attention.out_proj.weight"], sd[f"{p}.attention.out_proj.bias"])
    return o.contiguous().view(n, c, h, w) + x                        # Q4: memory reinterpretation


def _run(sd, stages, x, pad_stride2: bool):
    for i, op in enumerate(stages):
        p = str(i)
        if op[0] == "conv":
            _, _cin, _cout, _ks, stride, pad = op
            if pad_stride2 and stride == 2:
                x = F.pad(x, (0, 1, 0, 1))                            # asymmetric pad, sd/encoder.py:120-122
            x = F.conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], stride=stride, padding=pad)
        elif op[0] == "res":
            x = _res(sd, p, x)
        elif op[0] == "attn":
            x = _attn(sd, p, x)
        elif op[0] == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        elif op[0] == "gn":
            x = F.group_norm(x, 32, sd[f"{p}.weight"], sd[f"{p}.bias"])
        elif op[0] == "silu":
            x = F.silu(x)
    return x


class VAE_Decoder(_StateModule):
    """``backend="native"`` (default): hand-written HIP kernels through libsdmi (csrc/vae.hip); needs a
    cuda device, no fallback.  ``backend="torch"``: explicit opt-in to the PyTorch-op restatement (used by
    CPU unit tests of the quirk semantics and as an A/B reference on the GPU)."""

    def __init__(self, backend: str = "native"):
        super().__init__(arch.vae_decoder_manifest()[0])
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
                raise RuntimeError("VAE_Decoder(backend='native') needs a cuda (ROCm) device; there is no CPU fallback "
                                   "(use backend='torch' explicitly for a PyTorch-op restatement)")
            with torch.cuda.device(self._device):
                self._handle = _native.VaeDecoderHandle(self._state)
        return self._handle

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """(B,4,h,w) -> (B,3,8h,8w); divides the caller's tensor by 0.18215 in place like the reference."""
        x /= 0.18215
        if self.backend == "torch":
            return _run(self._state, arch.VAE_DECODER, x.to(self._device), pad_stride2=False)
        return self.handle().decode(x.to(self._device, torch.float32))

    forward = __call__


class VAE_Encoder(_StateModule):
    """``backend="native"`` (default): HIP kernels through libsdmi (csrc/vae.hip), cuda only, no fallback;
    ``backend="torch"``: explicit PyTorch-op restatement (CPU unit tests)."""

    def __init__(self, backend: str = "native"):
        super().__init__(arch.vae_encoder_manifest()[0])
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
                raise RuntimeError("VAE_Encoder(backend='native') needs a cuda (ROCm) device; there is no CPU fallback")
            with torch.cuda.device(self._device):
                self._handle = _native.VaeDecoderHandle(self._state, encoder=True)
        return self._handle

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) in [-1,1], noise (B,4,H/8,W/8) -> latents (sd/encoder.py:95-155)."""
        if self.backend == "native":
            return self.handle().encode(x.to(self._device, torch.float32), noise)
        x = _run(self._state, arch.VAE_ENCODER, x.to(self._device), pad_stride2=True)
        mean, log_variance = torch.chunk(x, 2, dim=1)
        log_variance = torch.clamp(log_variance, -30, 20)
        stdev = log_variance.exp().sqrt()
        x = mean + stdev * noise.to(self._device)
        x *= 0.18215
        return x

    forward = __call__
