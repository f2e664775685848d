#!/usr/bin/env python3
"""Golden vectors for the "next rows" (CLIP, VAE) by importing the reference (build container only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/sd")

from pytorch_stable_diffusion_amd import model_loader  # noqa: E402


def seeded(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float32)


@torch.no_grad()
def main():
    import clip as ref_clip
    import decoder as ref_dec
    import encoder as ref_enc
    sds = model_loader.synthetic_state_dicts(("clip", "encoder", "decoder"))
    out = {}
    c = ref_clip.CLIP()
    c.load_state_dict(sds["clip"], strict=True)
    tokens = torch.tensor([[49406, 320, 1929, 49407] + [49407] * 73, [49406] + [49407] * 76], dtype=torch.long)
    out["clip_tokens"] = tokens
    out["clip_out"] = c(tokens)
    d = ref_dec.VAE_Decoder()
    d.load_state_dict(sds["decoder"], strict=True)
    lat = seeded((1, 4, 8, 8), 301) * 0.18215 * 3
    lat_in = lat.clone()
    out["dec_out"] = d(lat_in)
    out["dec_inplace_ratio"] = (lat_in / lat).mean()           # reference divides the caller's tensor in place
    e = ref_enc.VAE_Encoder()
    e.load_state_dict(sds["encoder"], strict=True)
    img = seeded((1, 3, 64, 64), 302).clamp(-1, 1)
    noise = seeded((1, 4, 8, 8), 303)
    out["enc_out"] = e(img.clone(), noise.clone())
    # the sizes generate() runs at (BASELINE configs 2/3): decoder on 64x64 latents -> 512x512 image (kept 4x
    # subsampled: 3x128x128), encoder on a 512x512 image -> 4x64x64 latents
    lat64 = seeded((1, 4, 64, 64), 305) * 0.18215 * 3
    out["dec64_out_sub4"] = d(lat64.clone())[:, :, ::4, ::4].contiguous()
    img512 = seeded((1, 3, 512, 512), 306).clamp(-1, 1)
    out["enc512_out"] = e(img512.clone(), seeded((1, 4, 64, 64), 307))
    np.savez(os.path.join(HERE, "aux.npz"), **{k: v.numpy() for k, v in out.items()})
    print({k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
