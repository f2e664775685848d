"""End-to-end drop-in check (-m gpu): this repo's pipeline.generate() on the GPU (native CLIP, UNet loop,
VAE encoder/decoder: every model on the HIP library) against the image the reference's own generate() produced on the CPU with the
same synthetic weights, stub tokenizer, seed and prompt (tests/golden/e2e.npz).
Stated tolerance (north_star): pixel MAE < 1e-3 on the [0,1] float image; uint8 images may differ by a few LSB."""
import numpy as np
import pytest
import torch

from tests import gpu_util as G
from tests import helpers as H
from tests.stub_tokenizer import StubTokenizer

pytestmark = pytest.mark.gpu
DEV = "cuda"
PIXEL_MAE = 1e-3


class Tap:
    def __init__(self, inner):
        self.inner = inner
        self.last = None

    def to(self, d):
        self.inner.to(d)
        return self

    def __call__(self, *a):
        out = self.inner(*a)
        self.last = out.clone()
        return out


@pytest.fixture(scope="module")
def models():
    from pytorch_stable_diffusion_amd import model_loader
    m = model_loader.preload_models_synthetic(DEV)
    m["decoder"] = Tap(m["decoder"])
    return m


def _check(name, img, float_img, g):
    ref_u8 = g[f"{name}_u8"]
    ref_f = torch.from_numpy(g[f"{name}_float"])
    assert img.shape == ref_u8.shape == (512, 512, 3) and img.dtype == np.uint8
    got_f = float_img[0, :, ::4, ::4].cpu()
    mae = ((got_f - ref_f).abs().mean() / 2.0).item()          # decoder output is in [-1,1] -> /2 = [0,1] scale
    du8 = np.abs(img.astype(np.int32) - ref_u8.astype(np.int32))
    G.log_metric(test="e2e", name=name, pixel_mae=mae, u8_max=int(du8.max()), u8_mean=float(du8.mean()))
    assert mae < PIXEL_MAE, f"{name}: pixel MAE {mae:.2e}"
    assert du8.max() <= 8, f"{name}: uint8 max diff {du8.max()}"


def test_txt2img_matches_reference_generate(models):
    from pytorch_stable_diffusion_amd import pipeline
    g = H.load_npz("e2e.npz")
    img = pipeline.generate(prompt="a dog", uncond_prompt="", input_image=None, strength=0.8, do_cfg=True,
                            cfg_scale=7.5, sampler_name="ddpm", n_inference_steps=20, models=models, seed=42,
                            device=DEV, idle_device=None, tokenizer=StubTokenizer())
    _check("txt2img", img, models["decoder"].last, g)


def test_img2img_matches_reference_generate(models):
    from PIL import Image
    from pytorch_stable_diffusion_amd import pipeline
    g = H.load_npz("e2e.npz")
    dog = Image.fromarray(g["dog_u8"])
    img = pipeline.generate(prompt="a dog", uncond_prompt="", input_image=dog, strength=0.8, do_cfg=True,
                            cfg_scale=7.5, sampler_name="ddpm", n_inference_steps=10, models=models, seed=7,
                            device=DEV, idle_device=None, tokenizer=StubTokenizer())
    _check("img2img", img, models["decoder"].last, g)


def test_generate_argument_errors(models):
    from pytorch_stable_diffusion_amd import pipeline
    with pytest.raises(ValueError):
        pipeline.generate("a", "", strength=0.0, models=models, device=DEV, tokenizer=StubTokenizer())
    with pytest.raises(ValueError):
        pipeline.generate("a", "", sampler_name="euler", models=models, device=DEV, tokenizer=StubTokenizer(), n_inference_steps=2)
    with pytest.raises(KeyError):
        pipeline.generate("a", "", models={}, device=DEV, tokenizer=StubTokenizer())


def test_768_generate_runs(models):
    """BASELINE config 5 shape (96x96 latents): the loop and VAE run and return a 768x768 image."""
    from pytorch_stable_diffusion_amd import pipeline
    img = pipeline.generate(prompt="a dog", uncond_prompt="", do_cfg=True, cfg_scale=7.5, n_inference_steps=2,
                            models=models, seed=1, device=DEV, tokenizer=StubTokenizer(), height=768, width=768)
    assert img.shape == (768, 768, 3)


def test_native_vae_decoder_vs_reference():
    """Native HIP VAE decoder (csrc/vae.hip) vs the reference decoder's golden output (8x8 latents) and vs
    the torch-op restatement at 64x64 latents (the size generate() uses), incl. quirks Q3/Q4."""
    from pytorch_stable_diffusion_amd import model_loader
    from pytorch_stable_diffusion_amd.vae import VAE_Decoder
    g = H.load_npz("aux.npz")
    sd = model_loader.synthetic_state_dicts(("decoder",))["decoder"]
    nat = VAE_Decoder(backend="native")
    nat.load_state_dict(sd, strict=True)
    nat.to(DEV)
    lat = (H.seeded((1, 4, 8, 8), 301) * 0.18215 * 3).to(DEV)
    img = nat(lat.clone()).cpu()
    ref = torch.from_numpy(g["dec_out"])
    rel = H.rel_l2(img, ref)
    G.log_metric(test="vae_native", size=8, rel_l2=rel, max_abs=(img - ref).abs().max().item())
    assert rel < 5e-3, f"8x8: rel L2 {rel:.2e}"
    tor = VAE_Decoder(backend="torch")
    tor.load_state_dict(sd, strict=True)
    tor.to(DEV)
    lat64 = (H.seeded((1, 4, 64, 64), 305) * 0.18215 * 3).to(DEV)
    a = nat(lat64.clone())
    b = tor(lat64.clone())
    rel = H.rel_l2(a.cpu(), b.cpu())
    mae = ((a - b).abs().mean() / 2).item()
    G.log_metric(test="vae_native", size=64, rel_l2=rel, pixel_mae=mae, launches=nat.handle().last_launch_count)
    assert rel < 5e-3 and mae < 1e-3, f"64x64: rel L2 {rel:.2e}, pixel MAE {mae:.2e}"


def test_native_clip_vs_reference():
    """Native HIP CLIP text encoder (csrc/clip.hip) vs the reference CLIP's golden output (incl. causal mask,
    quick-GELU, final LayerNorm) for two token rows."""
    from pytorch_stable_diffusion_amd import model_loader
    from pytorch_stable_diffusion_amd.clip import CLIP
    g = H.load_npz("aux.npz")
    sd = model_loader.synthetic_state_dicts(("clip",))["clip"]
    c = CLIP(backend="native")
    c.load_state_dict(sd, strict=True)
    c.to(DEV)
    tokens = torch.from_numpy(g["clip_tokens"]).to(DEV)
    out = c(tokens).cpu()
    ref = torch.from_numpy(g["clip_out"])
    rel = H.rel_l2(out, ref)
    G.log_metric(test="clip_native", rel_l2=rel, max_abs=(out - ref).abs().max().item(), launches=c.handle().last_launch_count)
    assert rel < 3e-3, f"rel L2 {rel:.2e}"
    one = c(tokens[:1]).cpu()                 # batch 1 == row 0 of the batch-2 call
    assert H.rel_l2(one, out[:1]) < 1e-3


def test_native_vae_encoder_vs_reference():
    """Native HIP VAE encoder (csrc/vae.hip) vs the reference encoder's golden latents (64x64 image) and vs the
    torch-op restatement at 512x512 (asymmetric stride-2 padding, clamp/exp/sqrt reparameterisation, 0.18215)."""
    from pytorch_stable_diffusion_amd import model_loader
    from pytorch_stable_diffusion_amd.vae import VAE_Encoder
    g = H.load_npz("aux.npz")
    sd = model_loader.synthetic_state_dicts(("encoder",))["encoder"]
    nat = VAE_Encoder(backend="native")
    nat.load_state_dict(sd, strict=True)
    nat.to(DEV)
    x = H.seeded((1, 3, 64, 64), 302).clamp(-1, 1).to(DEV)
    z = nat(x, H.seeded((1, 4, 8, 8), 303).to(DEV)).cpu()
    ref = torch.from_numpy(g["enc_out"])
    rel = H.rel_l2(z, ref)
    G.log_metric(test="vae_enc_native", size=64, rel_l2=rel, max_abs=(z - ref).abs().max().item())
    assert rel < 5e-3, f"64x64: rel L2 {rel:.2e}"
    tor = VAE_Encoder(backend="torch")
    tor.load_state_dict(sd, strict=True)
    tor.to(DEV)
    x2 = H.seeded((1, 3, 512, 512), 306).clamp(-1, 1).to(DEV)
    n2 = H.seeded((1, 4, 64, 64), 307).to(DEV)
    a, b = nat(x2, n2), tor(x2.clone(), n2)
    rel = H.rel_l2(a.cpu(), b.cpu())
    G.log_metric(test="vae_enc_native", size=512, rel_l2=rel)
    assert rel < 5e-3, f"512x512: rel L2 {rel:.2e}"
