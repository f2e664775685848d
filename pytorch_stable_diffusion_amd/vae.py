"""VAE encoder / decoder with the reference's interface and weight ABI (sd/encoder.py:8-155,
sd/decoder.py:7-374); SURVEY 8f "next" rows.  Native HIP only (csrc/vae.hip: the UNet's implicit-GEMM / GroupNorm
kernels plus a row-softmax); there is no PyTorch-op path in the product (the CPU restatement used as the checker
lives in oracle/aux_ref.py).  The reference's behaviour is reproduced, including its quirks:
  Q3  VAE_AttentionBlock never applies its GroupNorm          (sd/decoder.py:31,34-73)
  Q4  ``x.transpose(-1, 2)`` is a no-op on a 3-D tensor and the (n, h*w, c) attention output is
      REINTERPRETED as (n, c, h, w) by ``view``                (sd/decoder.py:62,67)
  in-place ``x /= 0.18215`` on the caller's latents            (sd/decoder.py:364)
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import torch

from . import arch
from ._util import normalize_device


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
        device = normalize_device(device)
        if device != self._device:
            for k in list(self._state.keys()):
                self._state[k] = self._state[k].to(device)
            self._device = device
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(self._state.values())


class VAE_Decoder(_StateModule):
    """Hand-written HIP kernels through libsdmi (csrc/vae.hip); needs a cuda device, no fallback."""

    def __init__(self):
        super().__init__(arch.vae_decoder_manifest()[0])
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
                raise RuntimeError("VAE_Decoder needs a cuda (ROCm) device: the native HIP path has no CPU fallback")
            with torch.cuda.device(self._device):
                self._handle = _native.VaeDecoderHandle(self._state)
        return self._handle

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """(B,4,h,w) -> (B,3,8h,8w); divides the caller's tensor by 0.18215 in place like the reference."""
        x /= 0.18215
        return self.handle().decode(x.to(self._device, torch.float32))

    forward = __call__


class VAE_Encoder(_StateModule):
    """HIP kernels through libsdmi (csrc/vae.hip), cuda only, no fallback."""

    def __init__(self):
        super().__init__(arch.vae_encoder_manifest()[0])
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
                raise RuntimeError("VAE_Encoder needs a cuda (ROCm) device: the native HIP path has no CPU fallback")
            with torch.cuda.device(self._device):
                self._handle = _native.VaeDecoderHandle(self._state, encoder=True)
        return self._handle

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) in [-1,1], noise (B,4,H/8,W/8) -> latents (sd/encoder.py:95-155)."""
        return self.handle().encode(x.to(self._device, torch.float32), noise)

    forward = __call__
