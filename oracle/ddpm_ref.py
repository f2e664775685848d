"""Oracle: DDPM schedule/step and the CFG denoising loop (test infrastructure only).

Restates sd/ddpm.py:30-186 and the loop of sd/pipeline.py:205-237 including the
reference's quirk beta_start=0.000085 (sd/ddpm.py:30).  All scalar chains are evaluated on
0-d fp32 torch tensors exactly like the reference so coefficients are bit-identical.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch


class RefSchedule:
    def __init__(self, num_training_steps: int = 1000, beta_start: float = 0.000085,
                 beta_end: float = 0.0120):
        # sd/ddpm.py:43-53
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_training_steps,
                                    dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.one = torch.tensor(1.0)
        self.T = num_training_steps
        self.n = None
        self.timesteps = torch.from_numpy(np.arange(0, num_training_steps)[::-1].copy())

    def set_inference_timesteps(self, n: int = 50):
        # sd/ddpm.py:56-63
        self.n = n
        ratio = self.T // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def set_strength(self, strength: float = 1.0):
        # sd/ddpm.py:90-99
        start = self.n - int(self.n * strength)
        self.timesteps = self.timesteps[start:]
        self.start_step = start

    def prev_t(self, t: int) -> int:
        return t - self.T // self.n                     # sd/ddpm.py:66-69

    def variance(self, t: int) -> torch.Tensor:
        # sd/ddpm.py:72-87
        p = self.prev_t(t)
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[p] if p >= 0 else self.one
        cur_beta = 1 - a_t / a_p
        var = (1 - a_p) / (1 - a_t) * cur_beta
        return torch.clamp(var, min=1e-20)

    def step(self, t: int, latents: torch.Tensor, eps: torch.Tensor,
             noise: Optional[torch.Tensor]) -> torch.Tensor:
        """sd/ddpm.py:102-139 with the Gaussian draw supplied by the caller (``noise`` is
        ignored when t == 0, where the reference draws nothing)."""
        p = self.prev_t(t)
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[p] if p >= 0 else self.one
        b_t = 1 - a_t
        b_p = 1 - a_p
        cur_a = a_t / a_p
        cur_b = 1 - cur_a
        x0 = (latents - b_t ** 0.5 * eps) / a_t ** 0.5
        c0 = (a_p ** 0.5 * cur_b) / b_t
        ct = cur_a ** 0.5 * b_p / b_t
        prev = c0 * x0 + ct * latents
        if t > 0:
            prev = prev + (self.variance(t) ** 0.5) * noise
        return prev

    def add_noise(self, x0: torch.Tensor, t: int, noise: torch.Tensor) -> torch.Tensor:
        # sd/ddpm.py:143-186
        a = self.alphas_cumprod[t]
        return (a ** 0.5) * x0 + ((1 - a) ** 0.5) * noise


def time_embedding(t: int) -> torch.Tensor:
    """sd/pipeline.py:310-349: cat(cos, sin) of t * 10000^(-i/160), (1,320) fp32."""
    freqs = torch.pow(10000, -torch.arange(start=0, end=160, dtype=torch.float32) / 160)
    x = torch.tensor([t], dtype=torch.float32)[:, None] * freqs[None]
    return torch.cat([torch.cos(x), torch.sin(x)], dim=-1)


@torch.no_grad()
def denoise_loop(unet: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor],
                 latents: torch.Tensor, context: torch.Tensor, sched: RefSchedule,
                 generator: torch.Generator, cfg_scale: float = 7.5, do_cfg: bool = True,
                 on_step: Optional[Callable[[int, torch.Tensor], None]] = None) -> torch.Tensor:
    """sd/pipeline.py:205-237: per timestep time-embedding -> batch-2 UNet -> CFG combine
    (cond first, :122,:230-233) -> ancestral step drawing noise from ``generator`` when t>0."""
    for i, t in enumerate(sched.timesteps.tolist()):
        temb = time_embedding(t)
        x = latents.repeat(2, 1, 1, 1) if do_cfg else latents
        out = unet(x, context, temb)
        if do_cfg:
            cond, uncond = out.chunk(2)
            out = cfg_scale * (cond - uncond) + uncond
        noise = None
        if t > 0:
            noise = torch.randn(out.shape, generator=generator, dtype=out.dtype)
        latents = sched.step(t, latents, out, noise)
        if on_step is not None:
            on_step(i, latents)
    return latents
