#!/usr/bin/env python3
"""Goldens under the STRESS weight law (synth.py law="stress"): per-channel scales over two decades, 1 % outlier
channels x30, norm gamma in [0.2, 5], beta in [-2, 2] -- the statistics a trained checkpoint has and the benign
U(+-1/sqrt(fan_in)) law never produces.  Build container only: imports /root/reference/sd (never copied, never
shipped), loads the synthetic weights into the reference's own nn.Modules and records seeds + outputs.

  stress.npz       block outputs (three attention blocks, an attention block on a 24x24 map, three ResBlocks), one full
                   UNet forward at 64x64 (t = 980) -- inputs are regenerated from the seeds by tests/helpers.py
  stress_e2e.npz   the reference's own generate(): txt2img 'a dog', 512x512, 20 DDPM steps, CFG 7.5, seed 42, with the
                   stress-law UNet and the benign CLIP / VAE decoder
  stress_meta.json per block: rel-L2 of the reference module itself run in torch-CPU fp16 against its fp32 output (the
                   yardstick: what fp16 storage alone costs under this law), thread count, torch version

usage: make_golden_stress.py [blocks] [full] [e2e] [e2e50]     (default: blocks full e2e)
  stress_e2e50.npz the same generate() at BASELINE configs[1]'s own 50 steps
"""
import json
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

from pytorch_stable_diffusion_amd import arch, model_loader, synth  # noqa: E402
from pytorch_stable_diffusion_amd.tokenizer import StubTokenizer  # noqa: E402
from tests import helpers as H  # noqa: E402

BLOCKS = {
    # name: (kind, key prefix, ctor args, input shape, input seed)
    "attn_8_40": ("attn", "unet.encoders.1.1", (8, 40), (2, 320, 8, 8), 211),
    "attn_8_80": ("attn", "unet.encoders.4.1", (8, 80), (2, 640, 8, 8), 212),
    "attn_8_160": ("attn", "unet.encoders.7.1", (8, 160), (2, 1280, 8, 8), 213),
    "attn_8_80_s16": ("attn", "unet.decoders.6.1", (8, 80), (2, 640, 16, 16), 214),
    "attn_8_40_s24": ("attn", "unet.decoders.11.1", (8, 40), (2, 320, 24, 24), 215),
    "res_320_640": ("res", "unet.encoders.4.0", (320, 640), (2, 320, 8, 8), 202),
    "res_2560_1280": ("res", "unet.decoders.0.0", (2560, 1280), (2, 2560, 8, 8), 203),
    "res_640_640_16": ("res", "unet.encoders.5.0", (640, 640), (2, 640, 16, 16), 204),
}
CTX_SEED, TIME_SEED = 7, 8


def sub_sd(full, prefix):
    n = len(prefix) + 1
    return {k[n:]: v for k, v in full.items() if k.startswith(prefix + ".")}


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@torch.no_grad()
def blocks(out, meta):
    import diffusion as ref_diff
    man = arch.diffusion_manifest()
    context = H.seeded((2, 77, 768), CTX_SEED)
    time_vec = H.seeded((1, 1280), TIME_SEED)
    for name, (kind, prefix, args, ishape, seed) in BLOCKS.items():
        sub_man = {k: v for k, v in man.items() if k.startswith(prefix + ".")}
        full = synth.synth_state_dict(sub_man, law="stress")
        x = H.stress_input(ishape, seed)
        m = (ref_diff.UNET_ResidualBlock if kind == "res" else ref_diff.UNET_AttentionBlock)(*args)
        m.load_state_dict(sub_sd(full, prefix), strict=True)
        second = time_vec if kind == "res" else context
        y = m(x.clone(), second.clone())
        out[name] = y.numpy()
        # yardstick: the same reference module with fp16 parameters and activations (torch CPU)
        try:
            y16 = m.half()(x.half(), second.half()).float()
            r16 = rel(y16, y)
            # ... and on the block's own contribution y - x (attention blocks and same-width ResBlocks add their input
            # back, so the plain rel-L2 is dominated by the untouched skip path)
            d16 = rel(y16 - x, y - x) if y.shape == x.shape else None
        except Exception as e:  # an op without a CPU half kernel
            r16 = d16 = None
            print(f"  ({name}: fp16 yardstick unavailable: {e})")
        row = x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])
        meta["blocks"][name] = dict(kind=kind, prefix=prefix, args=list(args), ishape=list(ishape), seed=seed,
                                    ref_fp16_rel_l2=r16, ref_fp16_delta_rel_l2=d16, out_std=float(y.std()), out_absmax=float(y.abs().max()),
                                    in_row_mean_over_std=float((row.mean(1).abs() / row.std(1)).max()))
        print(f"  block {name}: out std {y.std():.3f} max {y.abs().max():.2f}  torch-fp16 rel-L2 {r16} (delta {d16})", flush=True)


@torch.no_grad()
def full(out, meta):
    import diffusion as ref_diff
    import pipeline as ref_pipeline
    man = arch.diffusion_manifest()
    with torch.device("meta"):
        m = ref_diff.Diffusion()
    m.load_state_dict(synth.synth_state_dict(man, law="stress"), strict=True, assign=True)
    context = H.seeded((2, 77, 768), 1)
    lat = H.seeded((1, 4, 64, 64), 0).repeat(2, 1, 1, 1)
    t0 = time.time()
    y = m(lat.clone(), context.clone(), ref_pipeline.get_time_embedding(980))
    out["unet_64_t980"] = y.numpy()
    meta["unet_64_t980"] = dict(out_std=float(y.std()), out_absmax=float(y.abs().max()))
    print(f"  full UNet 64x64 t=980: {time.time()-t0:.1f}s  out std {y.std():.4f} max {y.abs().max():.3f}", flush=True)
    return m


class Tap(torch.nn.Module):
    def __init__(self, inner, keep=False):
        super().__init__()
        self.inner, self.keep, self.inputs, self.last, self.last_in = inner, keep, [], None, None

    def forward(self, *a):
        self.last_in = a[0].clone()
        if self.keep:
            self.inputs.append(a[0][:1].clone())
        out = self.inner(*a)
        self.last = out.clone()            # the caller rescales ``out`` in place (sd/pipeline.py:246)
        return out


@torch.no_grad()
def e2e(unet_mod, steps=20):
    import clip as ref_clip
    import decoder as ref_dec
    import diffusion as ref_diff
    import pipeline as ref_pipeline
    sds = model_loader.synthetic_state_dicts(("clip", "decoder"))
    clip = ref_clip.CLIP()
    clip.load_state_dict(sds["clip"], strict=True)
    dec = ref_dec.VAE_Decoder()
    dec.load_state_dict(sds["decoder"], strict=True)
    if unet_mod is None:
        with torch.device("meta"):
            unet_mod = ref_diff.Diffusion()
        unet_mod.load_state_dict(synth.synth_state_dict(arch.diffusion_manifest(), law="stress"), strict=True, assign=True)
    unet, dect = Tap(unet_mod, keep=True), Tap(dec)
    t0 = time.time()
    img = ref_pipeline.generate(prompt="a dog", uncond_prompt="", input_image=None, strength=0.8, do_cfg=True,
                                cfg_scale=7.5, sampler_name="ddpm", n_inference_steps=steps,
                                models={"clip": clip, "diffusion": unet, "decoder": dect}, seed=42, device="cpu",
                                idle_device=None, tokenizer=StubTokenizer())
    print(f"  stress txt2img {steps} steps: {time.time()-t0:.1f}s", flush=True)
    lat_in = torch.cat(unet.inputs)
    if steps != 20:
        # BASELINE configs[1]'s own step count under the stress law: the uint8 image, the decoder's float image at FULL
        # resolution ([-1,1] clamped, 16-bit fixed point: 1.5e-5 of the range per count) and the latents entering every 5th step
        np.savez_compressed(os.path.join(HERE, f"stress_e2e{steps}.npz"), u8=img, float_u16=H.float_image_u16(dect.last[0]),
                            latents=dect.last_in.numpy(), lat_every5=lat_in[::5].numpy(), threads=np.array(torch.get_num_threads()))
        return
    np.savez_compressed(os.path.join(HERE, "stress_e2e.npz"),
                        txt20_u8=img, txt20_float=dect.last[0, :, ::4, ::4].numpy(), txt20_latents=dect.last_in.numpy(),
                        txt20_lat_every5=lat_in[::5].numpy(),
                        txt20_lat_stats=torch.stack([lat_in.mean(dim=(1, 2, 3)), lat_in.std(dim=(1, 2, 3))], 1).numpy(),
                        threads=np.array(torch.get_num_threads()))
    print("  latent std per step:", [round(float(v), 3) for v in lat_in.std(dim=(1, 2, 3))])


@torch.no_grad()
def yardstick_e2e():
    """The yardstick for the end-to-end run: the reference's own generate() with its UNet cast to torch-CPU fp16 (parameters
    and activations; sampler, CLIP and VAE stay fp32) on the stress law, pixel MAE against the fp32 golden in stress_e2e.npz.
    Says what fp16 storage alone costs under this law.  Slow (tens of minutes); result goes into stress_meta.json."""
    import clip as ref_clip
    import decoder as ref_dec
    import diffusion as ref_diff
    import pipeline as ref_pipeline
    sds = model_loader.synthetic_state_dicts(("clip", "decoder"))
    clip = ref_clip.CLIP()
    clip.load_state_dict(sds["clip"], strict=True)
    dec = ref_dec.VAE_Decoder()
    dec.load_state_dict(sds["decoder"], strict=True)
    with torch.device("meta"):
        unet = ref_diff.Diffusion()
    unet.load_state_dict(synth.synth_state_dict(arch.diffusion_manifest(), law="stress"), strict=True, assign=True)
    unet = unet.half()

    class Half(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, lat, ctx, temb):
            return self.inner(lat.half(), ctx.half(), temb.half()).float()

    t0 = time.time()
    img = ref_pipeline.generate(prompt="a dog", uncond_prompt="", input_image=None, strength=0.8, do_cfg=True,
                                cfg_scale=7.5, sampler_name="ddpm", n_inference_steps=20,
                                models={"clip": clip, "diffusion": Half(unet), "decoder": dec}, seed=42, device="cpu",
                                idle_device=None, tokenizer=StubTokenizer())
    gold = np.load(os.path.join(HERE, "stress_e2e.npz"))["txt20_u8"]
    mae = float(np.abs(img.astype(np.float64) - gold.astype(np.float64)).mean() / 255.0)
    mx = int(np.abs(img.astype(np.int32) - gold.astype(np.int32)).max())
    print(f"  torch-CPU fp16 UNet, stress law, 20 steps: {time.time()-t0:.0f}s  pixel MAE {mae:.3e}  uint8 max diff {mx}", flush=True)
    mpath = os.path.join(HERE, "stress_meta.json")
    meta = json.load(open(mpath))
    meta["e2e20_ref_fp16"] = dict(pixel_mae=mae, u8_max_diff=mx)
    json.dump(meta, open(mpath, "w"), indent=1)


def main():
    if sys.argv[1:] == ["yardstick"]:
        return yardstick_e2e()
    which = set(sys.argv[1:]) or {"blocks", "full", "e2e"}
    path = os.path.join(HERE, "stress.npz")
    mpath = os.path.join(HERE, "stress_meta.json")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    meta = json.load(open(mpath)) if os.path.exists(mpath) else {"blocks": {}}
    meta.update(ctx_seed=CTX_SEED, time_seed=TIME_SEED, threads=torch.get_num_threads(), torch=torch.__version__)
    unet_mod = None
    if "blocks" in which:
        blocks(out, meta)
    if "full" in which:
        unet_mod = full(out, meta)
    if which & {"blocks", "full"}:
        np.savez_compressed(path, **out)
        json.dump(meta, open(mpath, "w"), indent=1)
    if "e2e" in which:
        e2e(unet_mod)
    if "e2e50" in which:
        e2e(unet_mod, steps=50)


if __name__ == "__main__":
    main()
