"""``Diffusion``: drop-in for the reference's UNet noise predictor (sd/diffusion.py:797-837) backed by
the native HIP library.  It keeps the reference's weight ABI (state-dict keys of
sd/model_converter.py:13-650) and call convention ``model(latent, context, time) -> eps`` on NCHW fp32
tensors, plus a fused per-schedule interface used by ``pipeline.generate``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import arch
from ._util import normalize_device


class Diffusion:
    def __init__(self, stream_f32: bool = True, autotune: bool = True):
        self._manifest = arch.diffusion_manifest()
        self._state: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._device = torch.device("cpu")
        self._handle = None
        self._ctx_key = None
        self.stream_f32 = stream_f32
        self.autotune = autotune

    # ---- nn.Module-like surface used by the reference (model_loader.py:36-38, pipeline.py:199-200) --
    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict(self._state)

    def load_state_dict(self, state: Dict[str, torch.Tensor], strict: bool = True):
        missing = [k for k in self._manifest if k not in state]
        unexpected = [k for k in state if k not in self._manifest]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for Diffusion: missing {missing[:5]}"
                               f"{'...' if len(missing) > 5 else ''} unexpected {unexpected[:5]}")
        for k, shape in self._manifest.items():
            if k in state:
                if tuple(state[k].shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(state[k].shape)} vs {tuple(shape)}")
                self._state[k] = state[k].detach().to(self._device)      # like nn.Module: loaded INTO the model's device
        self._drop_handle()
        return self

    def to(self, device):
        device = normalize_device(device)
        if device != self._device:
            # fp16 copies on the GPU (the packer casts anyway); CPU copies stay as given
            for k in list(self._state.keys()):
                self._state[k] = self._state[k].to(device)
            self._device = device
            self._drop_handle()
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(self._state.values())

    def _drop_handle(self):
        if self._handle is not None:
            self._handle.close()
        self._handle = None
        self._ctx_key = None

    # ---- native handle -----------------------------------------------------------------------------
    def handle(self):
        from . import _native
        if self._handle is None:
            if self._device.type != "cuda":
                raise RuntimeError("Diffusion: native HIP path needs the model on a cuda (ROCm) device; "
                                   "call .to('cuda') first (there is no CPU fallback)")
            if len(self._state) != len(self._manifest):
                raise RuntimeError("Diffusion: weights not loaded")
            flags = (_native.FLAG_STREAM_F32 if self.stream_f32 else 0) | (0 if self.autotune else _native.FLAG_NO_TUNE)
            with torch.cuda.device(self._device):
                self._handle = _native.UNetHandle(self._state, flags)
        return self._handle

    def set_context(self, context: torch.Tensor):
        """Hoist the cross-attention K/V projections of ``context`` (B,77,768).  Always recomputed:
        a pointer/version cache would be unsafe (the caching allocator reuses addresses)."""
        self.handle().set_context(context.to(self._device, torch.float32))

    def set_schedule(self, time_embeddings: torch.Tensor):
        """time_embeddings: (n_steps, 320) rows of get_time_embedding(t)."""
        self.handle().set_schedule(time_embeddings.to(self._device, torch.float32))

    def step(self, latents: torch.Tensor, step_idx: int, do_cfg: bool, cfg_scale: float,
             noise: Optional[torch.Tensor], coef):
        """One fused denoising step in place on ``latents`` (1,4,h,w): UNet (batch 2 when do_cfg) +
        CFG combine + DDPM update."""
        self.handle().denoise_step(latents, step_idx, do_cfg, cfg_scale, noise, coef)

    @torch.no_grad()
    def denoise_native(self, latents: torch.Tensor, context: torch.Tensor, sampler, timesteps, do_cfg: bool,
                       cfg_scale: float) -> torch.Tensor:
        """The whole loop of sd/pipeline.py:205-237 as fused native steps; noise is drawn from the
        sampler's shared generator in the reference's order (one draw per step with t > 0)."""
        from .pipeline import get_time_embedding
        lat = latents.to(self._device, torch.float32).contiguous().clone()
        self.set_context(context)
        self.set_schedule(torch.cat([get_time_embedding(t) for t in timesteps]))
        for i, t in enumerate(timesteps):
            noise = sampler.draw_noise(lat.shape, self._device) if t > 0 else None
            self.step(lat, i, do_cfg, cfg_scale, noise, sampler.step_coefficients(t))
        return lat

    # ---- reference call convention -------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, latent: torch.Tensor, context: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        """latent (B,4,h,w), context (B,77,768), time (1,320) -> (B,4,h,w)   (sd/diffusion.py:797)."""
        self.set_context(context)
        lat = latent.to(self._device, torch.float32)
        temb = time.to(self._device, torch.float32).reshape(1, 320)
        return self.handle().forward(lat, lat.shape[0], temb=temb)

    forward = __call__
