"""Build the ``models`` dict that ``pipeline.generate`` consumes (reference: sd/model_loader.py:9-50).

``preload_models_from_standard_weights`` mirrors the reference's loader (checkpoint -> converter -> models);
``preload_models_from_state_dicts`` takes the four converted state dicts
({'clip','encoder','decoder','diffusion'}, sd/model_converter.py:3-1056).  ``preload_models_synthetic`` builds the name-keyed synthetic set used by
tests and benchmarks (no checkpoint offline)."""
from __future__ import annotations

from typing import Dict

import torch

from . import arch, synth
from .clip import CLIP
from .diffusion import Diffusion
from .vae import VAE_Decoder, VAE_Encoder


def preload_models_from_state_dicts(state_dicts: Dict[str, Dict[str, torch.Tensor]], device, accurate: bool = False) -> Dict[str, object]:
    """accurate=True: the UNet runs the wide-operand kernels (``Diffusion(accurate=True)``, include/sdmi.h SDMI_FLAG_ACCURATE)."""
    encoder = VAE_Encoder().to(device)
    encoder.load_state_dict(state_dicts["encoder"], strict=True)
    decoder = VAE_Decoder().to(device)
    decoder.load_state_dict(state_dicts["decoder"], strict=True)
    diffusion = Diffusion(accurate=accurate).to(device)
    diffusion.load_state_dict(state_dicts["diffusion"], strict=True)
    clip = CLIP().to(device)
    clip.load_state_dict(state_dicts["clip"], strict=True)
    return {"clip": clip, "encoder": encoder, "decoder": decoder, "diffusion": diffusion}


def preload_models_from_standard_weights(ckpt_path: str, device) -> Dict[str, object]:
    """Same entry point as the reference (sd/model_loader.py:9-50): load a standard SD-v1.x checkpoint, convert
    its keys (``model_converter``), build the four models on ``device``."""
    from . import model_converter
    state_dicts = model_converter.load_from_standard_weights(ckpt_path, "cpu")
    return preload_models_from_state_dicts(state_dicts, device)


def synthetic_state_dicts(which=("clip", "encoder", "decoder", "diffusion")) -> Dict[str, Dict[str, torch.Tensor]]:
    out = {}
    if "clip" in which:
        out["clip"] = synth.synth_state_dict(arch.clip_manifest(), seed_prefix="clip.")
    if "encoder" in which:
        m, nk = arch.vae_encoder_manifest()
        out["encoder"] = synth.synth_state_dict(m, norm_keys=nk, seed_prefix="encoder.")
    if "decoder" in which:
        m, nk = arch.vae_decoder_manifest()
        out["decoder"] = synth.synth_state_dict(m, norm_keys=nk, seed_prefix="decoder.")
    if "diffusion" in which:
        out["diffusion"] = synth.synth_state_dict(arch.diffusion_manifest())
    return out


def preload_models_synthetic(device) -> Dict[str, object]:
    return preload_models_from_state_dicts(synthetic_state_dicts(), device)
