"""DDPM sampler with the reference's interface (sd/ddpm.py:5-186) and a fused HIP step.

Host side (this file): schedule tables, timestep spacing, strength truncation and the per-step
coefficient chain, all evaluated on 0-d fp32 CPU tensors in the reference's operation order so the five
scalars handed to the kernel are bit-identical to the reference's (sd/ddpm.py:107-125).
Device side: ``sdmi_cfg_ddpm_step`` (csrc/misc.hip) applies x0-prediction, posterior mean and the
variance noise in one pass (optionally fused with the CFG combine).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch


class DDPMSampler:
    def __init__(self, generator: torch.Generator, num_training_steps: int = 1000,
                 beta_start: float = 0.000085, beta_end: float = 0.0120):
        # beta_start=0.000085 reproduces the reference (sd/ddpm.py:30), not canonical SD (0.00085)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_training_steps,
                                    dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.generator = generator
        self.num_train_timesteps = num_training_steps
        self.num_inference_steps = num_training_steps
        self.timesteps = torch.from_numpy(np.arange(0, num_training_steps)[::-1].copy())

    # -- schedule (sd/ddpm.py:56-99) -------------------------------------------------------------
    def set_inference_timesteps(self, num_inference_steps: int = 50):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def _get_previous_timestep(self, timestep: int) -> int:
        return timestep - self.num_train_timesteps // self.num_inference_steps

    def _get_variance(self, timestep: int) -> torch.Tensor:
        prev_t = self._get_previous_timestep(timestep)
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        cur_beta = 1 - a_t / a_p
        var = (1 - a_p) / (1 - a_t) * cur_beta
        return torch.clamp(var, min=1e-20)

    def set_strength(self, strength: float = 1):
        start_step = self.num_inference_steps - int(self.num_inference_steps * strength)
        self.timesteps = self.timesteps[start_step:]
        self.start_step = start_step

    # -- per-step scalars ------------------------------------------------------------------------
    def step_coefficients(self, timestep: int) -> Tuple[float, float, float, float, float]:
        """(sqrt(1-abar_t), sqrt(abar_t), pred_original_sample_coeff, current_sample_coeff,
        sqrt(variance) or 0 when t == 0), each an fp32 value computed as sd/ddpm.py:107-137 does."""
        t = int(timestep)
        prev_t = self._get_previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_p = 1 - a_p
        cur_a = a_t / a_p
        cur_b = 1 - cur_a
        sqrt_b = b_t ** 0.5
        sqrt_a = a_t ** 0.5
        c0 = a_p ** 0.5 * cur_b / b_t
        ct = cur_a ** 0.5 * b_p / b_t
        sigma = self._get_variance(t) ** 0.5 if t > 0 else torch.tensor(0.0)
        return (float(sqrt_b), float(sqrt_a), float(c0), float(ct), float(sigma))

    def draw_noise(self, shape, device, dtype=torch.float32) -> torch.Tensor:
        """One N(0,1) draw from the shared generator, on the generator's own device (a CPU generator
        reproduces the reference's CPU stream; the draw is then uploaded: 16 K floats per step)."""
        gdev = self.generator.device
        z = torch.randn(shape, generator=self.generator, device=gdev, dtype=dtype)
        return z.to(device)

    # -- step / add_noise (sd/ddpm.py:102-186) ---------------------------------------------------
    def step(self, timestep: int, latents: torch.Tensor, model_output: torch.Tensor) -> torch.Tensor:
        from . import _native
        t = int(timestep)
        coef = self.step_coefficients(t)
        noise = self.draw_noise(model_output.shape, model_output.device, model_output.dtype) if t > 0 else None
        if not latents.is_cuda:
            raise RuntimeError("DDPMSampler.step: the HIP step kernel needs CUDA/ROCm tensors (no CPU fallback)")
        out = latents.detach().clone().contiguous()
        _native.cfg_ddpm_step(model_output.contiguous(), False, 1.0, out, noise, coef)
        return out

    def add_noise(self, original_samples: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        acp = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sa = (acp[timesteps] ** 0.5).flatten()
        while sa.dim() < original_samples.dim():
            sa = sa.unsqueeze(-1)
        sb = ((1 - acp[timesteps]) ** 0.5).flatten()
        while sb.dim() < original_samples.dim():
            sb = sb.unsqueeze(-1)
        noise = self.draw_noise(original_samples.shape, original_samples.device, original_samples.dtype)
        return sa * original_samples + sb * noise
