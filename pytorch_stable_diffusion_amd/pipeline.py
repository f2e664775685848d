"""``generate()``: drop-in for the reference's sampling pipeline (sd/pipeline.py:13-262).

Same keyword surface, errors and RNG draw order as the reference; the denoising loop
(sd/pipeline.py:205-237) runs as fused native steps when ``models["diffusion"]`` is this package's
``Diffusion`` (time vectors precomputed per schedule, cross-attention K/V hoisted per prompt,
batch-2 CFG UNet + CFG combine + DDPM update per step on the GPU).

Extensions (all optional, defaults reproduce the reference): ``height``/``width`` (the reference
hard-codes 512x512, sd/pipeline.py:7-10), ``rng_device`` (the device of the shared noise generator;
"cpu" reproduces the reference's CPU noise stream bit for bit and uploads 16 K floats per step).
"""
from __future__ import annotations

import numpy as np
import torch

from .ddpm import DDPMSampler

WIDTH = 512
HEIGHT = 512
LATENTS_WIDTH = WIDTH // 8
LATENTS_HEIGHT = HEIGHT // 8


def rescale(x, old_range, new_range, clamp=False):
    """In-place affine range change (sd/pipeline.py:265-307)."""
    old_min, old_max = old_range
    new_min, new_max = new_range
    x -= old_min
    x *= (new_max - new_min) / (old_max - old_min)
    x += new_min
    if clamp:
        x = x.clamp(new_min, new_max)
    return x


def get_time_embedding(timestep):
    """(1,320) fp32 = cat(cos, sin)(t * 10000^(-i/160)), i in [0,160)  (sd/pipeline.py:310-349)."""
    freqs = torch.pow(10000, -torch.arange(start=0, end=160, dtype=torch.float32) / 160)
    x = torch.tensor([timestep], dtype=torch.float32)[:, None] * freqs[None]
    return torch.cat([torch.cos(x), torch.sin(x)], dim=-1)


def _encode_prompt(tokenizer, clip, text, device):
    tokens = tokenizer.batch_encode_plus([text], padding="max_length", max_length=77).input_ids
    tokens = torch.tensor(tokens, dtype=torch.long, device=device)
    return clip(tokens)


def generate(prompt, uncond_prompt=None, input_image=None, strength=0.8, do_cfg=True, cfg_scale=7.5,
             sampler_name="ddpm", n_inference_steps=50, models={}, seed=None, device=None, idle_device=None,
             tokenizer=None, height=HEIGHT, width=WIDTH, rng_device="cpu"):
    with torch.no_grad():
        if not 0 < strength <= 1:
            raise ValueError(f"Strength must be between 0 and 1, got {strength}")
        if height % 64 or width % 64:
            raise ValueError(f"height/width must be multiples of 64, got {height}x{width}")
        to_idle = (lambda x: x.to(idle_device)) if idle_device else (lambda x: x)
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        device = torch.device(device)
        generator = torch.Generator(device=rng_device if rng_device is not None else device)
        if seed is None:
            generator.seed()
        else:
            generator.manual_seed(seed)
        gdev = generator.device

        clip = models["clip"]
        clip.to(device)
        if do_cfg:
            cond_context = _encode_prompt(tokenizer, clip, prompt, device)
            uncond_context = _encode_prompt(tokenizer, clip, uncond_prompt, device)
            context = torch.cat([cond_context, uncond_context])      # cond first (sd/pipeline.py:122)
        else:
            context = _encode_prompt(tokenizer, clip, prompt, device)
        to_idle(clip)

        if sampler_name == "ddpm":
            sampler = DDPMSampler(generator)
            sampler.set_inference_timesteps(n_inference_steps)
        else:
            raise ValueError(f"Sampler {sampler_name} not found")

        lh, lw = height // 8, width // 8
        latents_shape = (1, 4, lh, lw)
        if input_image:
            encoder = models["encoder"]
            encoder.to(device)
            img = input_image.resize((width, height))
            img = torch.tensor(np.array(img), dtype=torch.float32, device=device)
            img = rescale(img, (0, 255), (-1, 1))
            img = img.unsqueeze(0).permute(0, 3, 1, 2)
            encoder_noise = torch.randn(latents_shape, generator=generator, device=gdev).to(device)
            latents = encoder(img, encoder_noise)
            sampler.set_strength(strength=strength)
            latents = sampler.add_noise(latents, sampler.timesteps[0])
            to_idle(encoder)
        else:
            latents = torch.randn(latents_shape, generator=generator, device=gdev).to(device)

        diffusion = models["diffusion"]
        diffusion.to(device)
        timesteps = sampler.timesteps.tolist()
        if hasattr(diffusion, "denoise_native"):
            latents = diffusion.denoise_native(latents, context, sampler, timesteps, do_cfg, cfg_scale)
        else:
            # any callable with the reference's model(latent, context, time) convention
            for t in timesteps:
                time_embedding = get_time_embedding(t).to(device)
                model_input = latents.repeat(2, 1, 1, 1) if do_cfg else latents
                model_output = diffusion(model_input, context, time_embedding)
                if do_cfg:
                    output_cond, output_uncond = model_output.chunk(2)
                    model_output = cfg_scale * (output_cond - output_uncond) + output_uncond
                latents = sampler.step(t, latents, model_output)
        to_idle(diffusion)

        decoder = models["decoder"]
        decoder.to(device)
        images = decoder(latents)
        to_idle(decoder)
        images = rescale(images, (-1, 1), (0, 255), clamp=True)
        images = images.permute(0, 2, 3, 1)
        images = images.to("cpu", torch.uint8).numpy()      # truncating cast, as the reference
        return images[0]


def generate_batch(prompts, uncond_prompt="", seeds=None, do_cfg=True, cfg_scale=7.5, sampler_name="ddpm", n_inference_steps=50,
                   models={}, device=None, idle_device=None, tokenizer=None, height=HEIGHT, width=WIDTH, rng_device="cpu"):
    """``len(prompts)`` txt2img calls of ``generate()`` as ONE batched denoising loop (throughput mode: more prompts than GPUs).
    Prompt i gets exactly what ``generate(prompt=prompts[i], seed=seeds[i], ...)`` would give it -- its own generator, the same
    draw order (initial latents, then one draw per step with t > 0), its own CLIP contexts, its own VAE decode -- but the UNet
    runs batch 2P through one chain of launches (``Diffusion.denoise_native_batch``), so the weights are streamed once per step
    for all P prompts and the launch-bound low-resolution levels do P times the work per launch.  Same numerics up to the tile
    plans of the larger GEMMs.  At most 8 prompts (UNet batch 16; the handle's activation arena grows by itself for the batch
    it is given: csrc/engine.h ensure_arena).  Returns a list of (H, W, 3) uint8 images."""
    with torch.no_grad():
        if height % 64 or width % 64:
            raise ValueError(f"height/width must be multiples of 64, got {height}x{width}")
        P = len(prompts)
        if not 1 <= P <= (8 if do_cfg else 16):
            raise ValueError(f"generate_batch: {P} prompts (1 .. {8 if do_cfg else 16})")
        if seeds is None:
            seeds = [None] * P
        if len(seeds) != P:
            raise ValueError("generate_batch: one seed per prompt")
        if sampler_name != "ddpm":
            raise ValueError(f"Sampler {sampler_name} not found")
        to_idle = (lambda x: x.to(idle_device)) if idle_device else (lambda x: x)
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        device = torch.device(device)
        gens = []
        for sd_ in seeds:
            g = torch.Generator(device=rng_device if rng_device is not None else device)
            if sd_ is None:
                g.seed()
            else:
                g.manual_seed(sd_)
            gens.append(g)
        clip = models["clip"]
        clip.to(device)
        cond = [_encode_prompt(tokenizer, clip, p, device) for p in prompts]
        if do_cfg:
            un = _encode_prompt(tokenizer, clip, uncond_prompt, device)
            context = torch.cat(cond + [un] * P)                  # all conditional contexts first (sd/pipeline.py:122 at batch P)
        else:
            context = torch.cat(cond)
        to_idle(clip)
        samplers = []
        for g in gens:
            sm = DDPMSampler(g)
            sm.set_inference_timesteps(n_inference_steps)
            samplers.append(sm)
        shape1 = (1, 4, height // 8, width // 8)
        latents = torch.cat([torch.randn(shape1, generator=g, device=g.device).to(device) for g in gens])
        diffusion = models["diffusion"]
        diffusion.to(device)
        if not hasattr(diffusion, "denoise_native_batch"):
            raise TypeError("generate_batch needs this package's Diffusion (denoise_native_batch)")
        latents = diffusion.denoise_native_batch(latents, context, samplers, samplers[0].timesteps.tolist(), do_cfg, cfg_scale)
        to_idle(diffusion)
        decoder = models["decoder"]
        decoder.to(device)
        out = []
        for i in range(P):
            images = decoder(latents[i:i + 1].clone())
            images = rescale(images, (-1, 1), (0, 255), clamp=True)
            out.append(images.permute(0, 2, 3, 1).to("cpu", torch.uint8).numpy()[0])
        to_idle(decoder)
        return out
