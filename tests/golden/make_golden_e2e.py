#!/usr/bin/env python3
"""End-to-end golden images: the reference's own pipeline.generate() with the reference's own CLIP / VAE /
UNet modules loaded with the name-keyed synthetic weights and a stub tokenizer (build container only).
Config-1 analogue: txt2img 'a dog', 512x512, 20 steps, CFG 7.5, seed 42.
Config-3 analogue: img2img from images/dog.jpg, strength 0.8, 10-step schedule (8 steps run), seed 7."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/sd")

from pytorch_stable_diffusion_amd import model_loader  # noqa: E402
from tests.stub_tokenizer import StubTokenizer  # noqa: E402


class Tap(torch.nn.Module):
    """Records the float tensor a module returns (the decoder's pre-quantisation image)."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner
        self.last = None
        self.last_in = None

    def forward(self, *a):
        self.last_in = a[0].clone()
        out = self.inner(*a)
        self.last = out.clone()
        return out


@torch.no_grad()
def main():
    import clip as ref_clip
    import decoder as ref_dec
    import diffusion as ref_diff
    import encoder as ref_enc
    import pipeline as ref_pipeline
    from PIL import Image
    sds = model_loader.synthetic_state_dicts()
    mods = {}
    for name, cls in (("clip", ref_clip.CLIP), ("encoder", ref_enc.VAE_Encoder), ("decoder", ref_dec.VAE_Decoder)):
        m = cls()
        m.load_state_dict(sds[name], strict=True)
        mods[name] = m
    with torch.device("meta"):
        d = ref_diff.Diffusion()
    d.load_state_dict(sds["diffusion"], strict=True, assign=True)
    mods["diffusion"] = d
    dec = Tap(mods["decoder"])
    mods["decoder"] = dec
    tok = StubTokenizer()
    out = {}
    t0 = time.time()
    img = ref_pipeline.generate(prompt="a dog", uncond_prompt="", input_image=None, strength=0.8, do_cfg=True,
                                cfg_scale=7.5, sampler_name="ddpm", n_inference_steps=20, models=mods, seed=42,
                                device="cpu", idle_device=None, tokenizer=tok)
    print(f"txt2img 20 steps: {time.time()-t0:.1f}s")
    out["txt2img_u8"] = img
    out["txt2img_latents"] = dec.last_in * 0.18215          # decoder divided its input in place before recording? no: Tap clones first
    out["txt2img_float"] = dec.last[0, :, ::4, ::4]          # subsampled float image in [-1,1] (3,128,128)
    t0 = time.time()
    dog = Image.open("/root/reference/images/dog.jpg")
    img2 = ref_pipeline.generate(prompt="a dog", uncond_prompt="", input_image=dog, strength=0.8, do_cfg=True,
                                 cfg_scale=7.5, sampler_name="ddpm", n_inference_steps=10, models=mods, seed=7,
                                 device="cpu", idle_device=None, tokenizer=tok)
    print(f"img2img 8 steps: {time.time()-t0:.1f}s")
    out["img2img_u8"] = img2
    out["img2img_latents"] = dec.last_in * 0.18215
    out["img2img_float"] = dec.last[0, :, ::4, ::4]
    out["dog_u8"] = np.array(dog.resize((512, 512)))         # the input pixels (data fixture for the GPU box)
    np.savez_compressed(os.path.join(HERE, "e2e.npz"), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print({k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
