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
    _GUARD_EARLY = 3     # steps after which a denoising loop first reads the LayerNorm-fold guard's counter

    def __init__(self, stream_f32: bool = True, autotune: bool = True, accurate: bool = False):
        self._manifest = arch.diffusion_manifest()
        self._state: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._device = torch.device("cpu")
        self._handle = None
        self._ctx_key = None
        self._parent = None
        self._lanes = []
        self.ln_guard_hits = 0           # rows beyond the LayerNorm-fold guard in the last denoise_native loop
        self.ln_guard_fallback = True    # repeat that loop unfused when there were any
        self.ln_guard_fallback_ran = False   # ... and whether the last loop was such a repeat
        self.ln_guard_total_hits = 0     # both summed over the life of the object (run_prompts / bench.py report them)
        self.ln_guard_fallbacks = 0
        self.stream_f32 = stream_f32
        self.autotune = autotune
        # accurate=True (include/sdmi.h SDMI_FLAG_ACCURATE): every GEMM / conv multiplies its activations as hi + lo fp16 pairs read
        # from fp32 tensors -- what remains of the fp16 path's error is the weights' own rounding.  Several times slower; for
        # validation and for weight laws on which fp16 activations miss the 1e-3 pixel tolerance.
        self.accurate = accurate

    # ---- nn.Module-like surface used by the reference (model_loader.py:36-38, pipeline.py:199-200) --
    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict(self._state)

    def load_state_dict(self, state: Dict[str, torch.Tensor], strict: bool = True):
        missing = [k for k in self._manifest if k not in state]
        unexpected = [k for k in state if k not in self._manifest]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for Diffusion: missing {missing[:5]}"
                               f"{'...' if len(missing) > 5 else ''} unexpected {unexpected[:5]}")
        self._refuse_on_lane("load_state_dict")
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
            self._refuse_on_lane("to(another device)")
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
        for ln in getattr(self, "_lanes", []):
            ln._drop_handle()
        self._lanes = []
        if self._handle is not None:
            self._handle.close()
        self._handle = None
        self._ctx_key = None

    def _refuse_on_lane(self, what: str):
        """A lane shares its parent's state dict BY REFERENCE and borrows the parent's packed weights: moving or replacing
        the tensors through a lane would change them under the parent (whose handle and device would not follow)."""
        if getattr(self, "_parent", None) is not None:
            raise RuntimeError(f"Diffusion lane: {what} must be called on the parent model (lanes borrow its weights)")

    def lane(self, index: Optional[int] = None) -> "Diffusion":
        """A second LANE of this model: same weights (the packed copies on the GPU are shared, nothing is re-packed), own
        activation arena / context / schedule.  Two lanes driven on two HIP streams run two independent generate() loops
        concurrently on one GPU and fill each other's per-launch latency (``replicas.run_prompts(streams_per_gpu=2)``).
        ``index``: reuse lane number ``index`` (0-based among the extra lanes) when it already exists instead of making
        another one -- a lane holds a 6 GiB arena, so callers that come back (a service loop) must not pile them up.
        Lanes are dropped with this model's handle or by ``release_lanes()``."""
        self._refuse_on_lane("lane()")
        if not hasattr(self, "_lanes"):
            self._lanes = []
        if index is not None and 0 <= index < len(self._lanes):
            return self._lanes[index]
        ln = Diffusion.__new__(Diffusion)
        ln._manifest, ln._state, ln._device = self._manifest, self._state, self._device
        ln.stream_f32, ln.autotune, ln.accurate = self.stream_f32, self.autotune, self.accurate
        ln._ctx_key = None
        ln._lanes = []
        ln.ln_guard_hits, ln.ln_guard_fallback = 0, self.ln_guard_fallback
        ln.ln_guard_fallback_ran, ln.ln_guard_total_hits, ln.ln_guard_fallbacks = False, 0, 0
        ln._parent = self
        ln._handle = self.handle().clone()
        self._lanes.append(ln)
        return ln

    def lanes(self, n: int):
        """[self, lane 0, ..., lane n-2]: ``n`` concurrent denoising loops over one copy of the packed weights; existing lanes
        are reused, missing ones created."""
        return [self] + [self.lane(i) for i in range(max(0, n - 1))]

    def release_lanes(self):
        """Free every lane's arena / slabs / context buffers (the parent keeps its own)."""
        for ln in getattr(self, "_lanes", []):
            ln._drop_handle()
        self._lanes = []

    # ---- native handle -----------------------------------------------------------------------------
    def handle(self):
        from . import _native
        if self._handle is None:
            if self._device.type != "cuda":
                raise RuntimeError("Diffusion: native HIP path needs the model on a cuda (ROCm) device; "
                                   "call .to('cuda') first (there is no CPU fallback)")
            if len(self._state) != len(self._manifest):
                raise RuntimeError("Diffusion: weights not loaded")
            flags = ((_native.FLAG_STREAM_F32 if self.stream_f32 else 0) | (0 if self.autotune else _native.FLAG_NO_TUNE)
                     | (_native.FLAG_ACCURATE if getattr(self, "accurate", False) else 0))
            with torch.cuda.device(self._device):
                self._handle = _native.UNetHandle(self._state, flags)
        return self._handle

    def set_context(self, context: torch.Tensor):
        """Hoist the cross-attention K/V projections of ``context`` (B,77,768) and build the folded cross-attention operands
        (32 GEMMs + 66 fold launches, ~0.6 ms).  Unconditional; ``__call__`` skips it when the context did not change."""
        self.handle().set_context(context.to(self._device, torch.float32))
        self._ctx_key = None

    def _context_signature(self, ctx32: torch.Tensor):
        """Content signature of a context tensor, computed ON the GPU (two fp64 reductions, one of them against a fixed
        pseudo-random weight vector; one 16-byte read-back): a pointer / version cache would be unsafe (the caching allocator
        reuses addresses), hashing on the host would copy 0.5 MB per call."""
        n = ctx32.numel()
        w = getattr(self, "_sig_w", None)
        if w is None or w.numel() != n or w.device != ctx32.device:
            g = torch.Generator(device="cpu").manual_seed(0x5D31)
            w = torch.rand(n, generator=g, dtype=torch.float64).to(ctx32.device)
            self._sig_w = w
        flat = ctx32.reshape(-1).double()
        sig = torch.stack([flat.sum(), (flat * w).sum()]).tolist()
        return (tuple(ctx32.shape), sig[0], sig[1])

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
        self.set_context(context)
        self.set_schedule(torch.cat([get_time_embedding(t) for t in timesteps]))
        h = self.handle()
        rng_state = sampler.generator.get_state()
        h.ln_guard(reset=True)

        def loop(early_check: bool):
            lat = latents.to(self._device, torch.float32).contiguous().clone()
            for i, t in enumerate(timesteps):
                noise = sampler.draw_noise(lat.shape, self._device) if t > 0 else None
                self.step(lat, i, do_cfg, cfg_scale, noise, sampler.step_coefficients(t))
                # a row beyond the guard shows up in the first steps if it shows up at all (the stream's statistics are set by
                # the weights, not by the step): one read of the counter after _GUARD_EARLY steps lets a doomed folded loop
                # stop there instead of paying for all of it before the unfused repeat
                if early_check and i + 1 == self._GUARD_EARLY and i + 1 < len(timesteps):
                    early[0] = h.ln_guard(reset=False)
                    if early[0]:
                        return None
            return lat

        early = [0]
        lat = loop(self.ln_guard_fallback)
        # Guard of the LayerNorm fold (include/sdmi.h sdmi_unet_ln_guard): the folded GEMMs multiply the raw stream's fp16
        # shadow, exact enough while a token row's |mean| stays within a few sigma.  The kernels count the rows beyond 8 sigma;
        # when any was met the loop is repeated -- same latents, same noise stream -- through the separate LayerNorm kernel.
        self.ln_guard_hits = max(h.ln_guard(reset=True), early[0])
        self.ln_guard_total_hits += self.ln_guard_hits
        self.ln_guard_fallback_ran = False
        if self.ln_guard_hits and self.ln_guard_fallback:
            sampler.generator.set_state(rng_state)
            h.ln_guard(reset=True, fold_on=False)
            try:
                lat = loop(False)
            finally:
                h.ln_guard(reset=True, fold_on=True)
            self.ln_guard_fallback_ran = True
            self.ln_guard_fallbacks += 1
        return lat

    @torch.no_grad()
    def denoise_native_batch(self, latents: torch.Tensor, context: torch.Tensor, samplers, timesteps, do_cfg: bool,
                             cfg_scale: float) -> torch.Tensor:
        """P independent prompts through ONE chain of launches (throughput mode): latents (P,4,h,w); context (2P,77,768) =
        cat([cond_0..cond_P-1, uncond_0..uncond_P-1]) ((P,77,768) without guidance); samplers: one DDPMSampler per prompt, each
        with its own generator -- prompt i's noise stream is the one its own generate() call would draw -- all on the same
        timesteps.  Every weight is read once per step for the P prompts."""
        from .pipeline import get_time_embedding
        lat = latents.to(self._device, torch.float32).contiguous().clone()
        P = lat.shape[0]
        if len(samplers) != P or context.shape[0] != (2 if do_cfg else 1) * P:
            raise ValueError(f"denoise_native_batch: {P} latents, {len(samplers)} samplers, context batch {context.shape[0]}")
        self.set_context(context)
        self.set_schedule(torch.cat([get_time_embedding(t) for t in timesteps]))
        one = (1,) + tuple(lat.shape[1:])
        h = self.handle()
        states = [s.generator.get_state() for s in samplers]
        h.ln_guard(reset=True)

        def loop(early_check: bool):
            x = lat.clone()
            for i, t in enumerate(timesteps):
                noise = torch.cat([s.draw_noise(one, self._device) for s in samplers]) if t > 0 else None
                self.step(x, i, do_cfg, cfg_scale, noise, samplers[0].step_coefficients(t))
                if early_check and i + 1 == self._GUARD_EARLY and i + 1 < len(timesteps):
                    early[0] = h.ln_guard(reset=False)            # (see denoise_native: one row of one prompt repeats all P)
                    if early[0]:
                        return None
            return x

        early = [0]
        out = loop(self.ln_guard_fallback)
        self.ln_guard_hits = max(h.ln_guard(reset=True), early[0])      # guard of the LayerNorm fold: see denoise_native
        self.ln_guard_total_hits += self.ln_guard_hits
        self.ln_guard_fallback_ran = False
        if self.ln_guard_hits and self.ln_guard_fallback:
            for s, st in zip(samplers, states):
                s.generator.set_state(st)
            h.ln_guard(reset=True, fold_on=False)
            try:
                out = loop(False)
            finally:
                h.ln_guard(reset=True, fold_on=True)
            self.ln_guard_fallback_ran = True
            self.ln_guard_fallbacks += 1
        return out

    # ---- reference call convention -------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, latent: torch.Tensor, context: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        """latent (B,4,h,w), context (B,77,768), time (1,320) -> (B,4,h,w)   (sd/diffusion.py:797).  The reference's own
        loop passes the same context on every step (sd/pipeline.py:225): the per-prompt hoist is redone only when its
        content changed (INTEGRATION.md mode 1: the object swapped in under the reference's loop)."""
        ctx32 = context.to(self._device, torch.float32)
        sig = self._context_signature(ctx32)
        if sig != self._ctx_key:
            self.handle().set_context(ctx32)
            self._ctx_key = sig
        lat = latent.to(self._device, torch.float32)
        temb = time.to(self._device, torch.float32).reshape(1, 320)
        return self.handle().forward(lat, lat.shape[0], temb=temb)

    forward = __call__
