#!/usr/bin/env python3
"""End-to-end goldens at the step counts BASELINE configs 2/3/5 are stated on: the reference's own
pipeline.generate() (imported from /root/reference/sd, build container only) with its own CLIP / VAE / UNet modules
loaded with the name-keyed synthetic weights and the stub tokenizer.

  config 2: txt2img 'a dog', 512x512, 50 DDPM steps, CFG 7.5, seed 42            -> txt50_*
  config 3: img2img images/dog.jpg, strength 0.8 (50-step schedule, 40 steps run from t=780,
            sd/ddpm.py:90-99), CFG 7.5, seed 7                                  -> img50_*
  config 5: txt2img 768x768 (4x96x96 latents), 3 steps, seed 1                   -> t768_*
            (the reference hard-codes 512: its module constants WIDTH/HEIGHT/LATENTS_* are set to the 768 values
             for this call only, sd/pipeline.py:8-11)

Per run: uint8 image, 4x-subsampled float image (decoder output, [-1,1]), final latents, the latents entering every
5th UNet call (drift localisation) and {mean, std} of the latents entering every step.

  config 5 at its stated step count: txt2img 768x768, 50 steps, seed 1                 -> t768x50_* (e2e768.npz; ~15 min of CPU)

usage: make_golden_e2e50.py [txt50] [img50] [t768] [t768x50]      (default: the first three; existing keys are kept)
"""
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
from tests import helpers as H  # noqa: E402
from tests.stub_tokenizer import StubTokenizer  # noqa: E402

OUT = os.path.join(HERE, "e2e50.npz")


class Tap(torch.nn.Module):
    """Records what goes into / comes out of a reference module."""

    def __init__(self, inner, keep_inputs=False):
        super().__init__()
        self.inner = inner
        self.keep_inputs = keep_inputs
        self.inputs = []
        self.last = None
        self.last_in = None

    def forward(self, *a):
        self.last_in = a[0].clone()
        if self.keep_inputs:
            self.inputs.append(a[0][:1].clone())      # CFG duplicates the latents: row 0 == row 1
        out = self.inner(*a)
        self.last = out.clone()
        return out


def record(out, tag, img, dec, unet):
    lat_in = torch.cat(unet.inputs)                     # (n_steps, 4, h, w): latents entering each UNet call
    out[f"{tag}_u8"] = img
    out[f"{tag}_float"] = dec.last[0, :, ::4, ::4].numpy()
    if tag == "txt50":
        out[f"{tag}_float_u16"] = H.float_image_u16(dec.last[0])      # FULL resolution, [-1,1] clamped, 16-bit fixed point
    out[f"{tag}_latents"] = dec.last_in.numpy()         # Tap clones before the decoder's in-place /0.18215
    out[f"{tag}_lat_every5"] = lat_in[::5].numpy()
    out[f"{tag}_lat_stats"] = torch.stack([lat_in.mean(dim=(1, 2, 3)), lat_in.std(dim=(1, 2, 3))], 1).numpy()
    unet.inputs.clear()


@torch.no_grad()
def main():
    import clip as ref_clip
    import decoder as ref_dec
    import diffusion as ref_diff
    import encoder as ref_enc
    import pipeline as ref_pipeline
    from PIL import Image
    which = set(sys.argv[1:]) or {"txt50", "img50", "t768"}
    sds = model_loader.synthetic_state_dicts()
    mods = {}
    for name, cls in (("clip", ref_clip.CLIP), ("encoder", ref_enc.VAE_Encoder), ("decoder", ref_dec.VAE_Decoder)):
        m = cls()
        m.load_state_dict(sds[name], strict=True)
        mods[name] = m
    with torch.device("meta"):
        d = ref_diff.Diffusion()
    d.load_state_dict(sds["diffusion"], strict=True, assign=True)
    unet = Tap(d, keep_inputs=True)
    dec = Tap(mods["decoder"])
    mods["diffusion"] = unet
    mods["decoder"] = dec
    tok = StubTokenizer()
    out = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    out["threads"] = np.array(torch.get_num_threads())

    def save():
        np.savez_compressed(OUT, **out)

    if "txt50" in which:
        t0 = time.time()
        img = ref_pipeline.generate(prompt="a dog", uncond_prompt="", input_image=None, strength=0.8, do_cfg=True,
                                    cfg_scale=7.5, sampler_name="ddpm", n_inference_steps=50, models=mods, seed=42,
                                    device="cpu", idle_device=None, tokenizer=tok)
        print(f"txt2img 50 steps: {time.time()-t0:.1f}s", flush=True)
        record(out, "txt50", img, dec, unet)
        save()
    if "img50" in which:
        t0 = time.time()
        dog = Image.open("/root/reference/images/dog.jpg")
        img = ref_pipeline.generate(prompt="a dog", uncond_prompt="", input_image=dog, strength=0.8, do_cfg=True,
                                    cfg_scale=7.5, sampler_name="ddpm", n_inference_steps=50, models=mods, seed=7,
                                    device="cpu", idle_device=None, tokenizer=tok)
        print(f"img2img 40 of 50 steps: {time.time()-t0:.1f}s", flush=True)
        record(out, "img50", img, dec, unet)
        save()
    if "t768" in which:
        saved = {k: getattr(ref_pipeline, k) for k in ("WIDTH", "HEIGHT", "LATENTS_WIDTH", "LATENTS_HEIGHT")}
        ref_pipeline.WIDTH = ref_pipeline.HEIGHT = 768
        ref_pipeline.LATENTS_WIDTH = ref_pipeline.LATENTS_HEIGHT = 96
        try:
            t0 = time.time()
            img = ref_pipeline.generate(prompt="a dog", uncond_prompt="", input_image=None, strength=0.8, do_cfg=True,
                                        cfg_scale=7.5, sampler_name="ddpm", n_inference_steps=3, models=mods, seed=1,
                                        device="cpu", idle_device=None, tokenizer=tok)
            print(f"txt2img 768x768, 3 steps: {time.time()-t0:.1f}s", flush=True)
        finally:
            for k, v in saved.items():
                setattr(ref_pipeline, k, v)
        record(out, "t768", img, dec, unet)
        del out["t768_lat_every5"]
        out["t768_u8"] = img[::2, ::2]                   # 384x384x3 subsample keeps the fixture small
        save()
    if "t768x50" in which:
        saved = {k: getattr(ref_pipeline, k) for k in ("WIDTH", "HEIGHT", "LATENTS_WIDTH", "LATENTS_HEIGHT")}
        ref_pipeline.WIDTH = ref_pipeline.HEIGHT = 768
        ref_pipeline.LATENTS_WIDTH = ref_pipeline.LATENTS_HEIGHT = 96
        try:
            t0 = time.time()
            img = ref_pipeline.generate(prompt="a dog", uncond_prompt="", input_image=None, strength=0.8, do_cfg=True,
                                        cfg_scale=7.5, sampler_name="ddpm", n_inference_steps=50, models=mods, seed=1,
                                        device="cpu", idle_device=None, tokenizer=tok)
            print(f"txt2img 768x768, 50 steps: {time.time()-t0:.1f}s", flush=True)
        finally:
            for k, v in saved.items():
                setattr(ref_pipeline, k, v)
        lat_in = torch.cat(unet.inputs)
        np.savez_compressed(os.path.join(HERE, "e2e768.npz"), t768x50_u8=img[::2, ::2],       # 384x384x3 subsample
                            t768x50_float=dec.last[0, :, ::8, ::8].numpy(), t768x50_latents=dec.last_in.numpy(),
                            t768x50_lat_stats=torch.stack([lat_in.mean(dim=(1, 2, 3)), lat_in.std(dim=(1, 2, 3))], 1).numpy(),
                            threads=np.array(torch.get_num_threads()))
        unet.inputs.clear()
    print({k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
