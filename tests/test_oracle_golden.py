"""Pin the CPU oracle (oracle/) against golden vectors captured from the imported reference
(tests/golden/make_golden.py).  CPU only; this is what makes the oracle trustworthy as the
checker for the HIP path."""
import numpy as np
import pytest
import torch

from oracle import ddpm_ref, unet_ref
from tests import helpers as H

ATOL = 2e-5   # reference fp32 forward moves by up to 5.4e-6 between 1 and 8 threads


def test_ddpm_tables_bit_exact():
    g = H.load_npz("ddpm.npz")
    s = ddpm_ref.RefSchedule()
    assert np.array_equal(s.betas.numpy(), g["betas"])
    assert np.array_equal(s.alphas_cumprod.numpy(), g["alphas_cumprod"])
    # quirk Q1 (sd/ddpm.py:30): beta_start = 0.000085
    assert abs(float(s.betas[0]) - 8.5e-5) < 1e-9
    assert abs(float(s.alphas_cumprod[999]) - 0.0124941) < 1e-6
    for n in (20, 50):
        s.set_inference_timesteps(n)
        assert np.array_equal(s.timesteps.numpy(), g[f"timesteps_{n}"])
    assert s.timesteps[0] == 980 and s.timesteps[-1] == 0


@pytest.mark.parametrize("n,st", [(50, 0.8), (50, 0.9), (20, 0.5), (50, 1.0)])
def test_ddpm_strength(n, st):
    g = H.load_npz("ddpm.npz")
    s = ddpm_ref.RefSchedule()
    s.set_inference_timesteps(n)
    s.set_strength(st)
    assert np.array_equal(s.timesteps.numpy(), g[f"timesteps_{n}_s{int(st*100)}"])


@pytest.mark.parametrize("n", [20, 50])
def test_ddpm_step_bit_exact(n):
    g = H.load_npz("ddpm.npz")
    lat, eps = H.seeded((1, 4, 8, 8), 11), H.seeded((1, 4, 8, 8), 12)
    s = ddpm_ref.RefSchedule()
    s.set_inference_timesteps(n)
    ts = s.timesteps.tolist()
    for t in (ts[0], ts[len(ts) // 2], ts[-1]):
        gen = torch.Generator(device="cpu").manual_seed(77)
        noise = torch.randn(eps.shape, generator=gen, dtype=eps.dtype) if t > 0 else None
        out = s.step(t, lat.clone(), eps.clone(), noise)
        assert np.array_equal(out.numpy(), g[f"step_n{n}_t{t}"]), f"t={t}"
        if t > 0:
            assert np.array_equal(s.variance(t).numpy(), g[f"var_n{n}_t{t}"])


def test_ddpm_add_noise_bit_exact():
    g = H.load_npz("ddpm.npz")
    s = ddpm_ref.RefSchedule()
    s.set_inference_timesteps(50)
    s.set_strength(0.8)
    t = int(s.timesteps[0])
    assert t == int(g["add_noise_t"]) == 780
    gen = torch.Generator(device="cpu").manual_seed(78)
    lat = H.seeded((1, 4, 8, 8), 11)
    noise = torch.randn(lat.shape, generator=gen, dtype=lat.dtype)
    assert np.array_equal(s.add_noise(lat, t, noise).numpy(), g["add_noise_out"])


@pytest.mark.parametrize("n", [20, 50])
def test_time_embedding_bit_exact(n):
    g = H.load_npz("ddpm.npz")
    s = ddpm_ref.RefSchedule()
    s.set_inference_timesteps(n)
    te = torch.cat([ddpm_ref.time_embedding(t) for t in s.timesteps.tolist()], 0)
    assert np.array_equal(te.numpy(), g[f"temb_{n}"])
    assert te.shape == (n, 320)


def _run_block(name, meta):
    kind, prefix, args = meta["kind"], meta["prefix"], meta["args"]
    sd = H.block_weights(prefix)
    x = H.seeded(tuple(meta["ishape"]), meta["seed"])
    ctx = H.seeded((2, 77, 768), 7)
    tv = H.seeded((1, 1280), 8)
    with torch.no_grad():
        if kind == "res":
            return unet_ref.residual_block(sd, prefix, x, tv)
        if kind == "attn":
            return unet_ref.attention_block(sd, prefix, x, ctx, args[0])
        if kind == "up":
            return unet_ref.upsample(sd, prefix, x)
        if kind == "conv":
            return torch.nn.functional.conv2d(x, sd[prefix + ".weight"], sd[prefix + ".bias"],
                                              stride=args[2], padding=1)
        if kind == "final":
            return unet_ref.output_layer(sd, x)
    raise ValueError(kind)


_META = H.blocks_meta()


@pytest.mark.parametrize("name", sorted(_META["blocks"].keys()))
def test_block_matches_reference(name):
    g = H.load_npz("blocks.npz")
    y = _run_block(name, _META["blocks"][name])
    ref = torch.from_numpy(g[name])
    assert y.shape == ref.shape
    err = (y - ref).abs().max().item()
    assert err <= ATOL, f"{name}: max abs err {err}"


def test_time_mlp_matches_reference():
    g = H.load_npz("blocks.npz")
    sd = H.block_weights("time_embedding")
    y = unet_ref.time_mlp(sd, ddpm_ref.time_embedding(980))
    assert (y - torch.from_numpy(g["time_mlp_t980"])).abs().max().item() <= ATOL


def test_bare_attention_matches_reference():
    from pytorch_stable_diffusion_amd import synth
    g = H.load_npz("blocks.npz")
    sa_sd = synth.synth_state_dict({"sa.in_proj.weight": (192, 64), "sa.in_proj.bias": (192,),
                                    "sa.out_proj.weight": (64, 64), "sa.out_proj.bias": (64,)})
    xs = H.seeded((2, 20, 64), 131)
    y = unet_ref.self_attention(sa_sd, "sa", xs, 4)
    assert (y - torch.from_numpy(g["selfattn_plain"])).abs().max().item() <= 1e-6
    y = unet_ref.self_attention(sa_sd, "sa", xs, 4, causal=True)
    assert (y - torch.from_numpy(g["selfattn_causal"])).abs().max().item() <= 1e-6
    ca_sd = synth.synth_state_dict({"ca.q_proj.weight": (64, 64), "ca.q_proj.bias": (64,),
                                    "ca.k_proj.weight": (64, 48), "ca.k_proj.bias": (64,),
                                    "ca.v_proj.weight": (64, 48), "ca.v_proj.bias": (64,),
                                    "ca.out_proj.weight": (64, 64), "ca.out_proj.bias": (64,)})
    ys = H.seeded((2, 7, 48), 132)
    y = unet_ref.cross_attention(ca_sd, "ca", xs, ys, 4)
    assert (y - torch.from_numpy(g["crossattn"])).abs().max().item() <= 1e-6


def test_quirk_q2_gate_is_dead():
    """sd/diffusion.py:359-363: the GeGLU gate half never reaches the output."""
    g = H.load_npz("blocks.npz")
    assert float(g["q2_gate_zeroed_maxabs_diff"]) == 0.0
    prefix = "unet.encoders.1.1"
    sd = dict(H.block_weights(prefix))
    x = H.seeded((2, 320, 8, 8), 111)
    ctx = H.seeded((2, 77, 768), 7)
    y0 = unet_ref.attention_block(sd, prefix, x, ctx, 8)
    sd[prefix + ".linear_geglu_1.weight"] = sd[prefix + ".linear_geglu_1.weight"].clone()
    sd[prefix + ".linear_geglu_1.weight"][1280:] = 123.0
    y1 = unet_ref.attention_block(sd, prefix, x, ctx, 8)
    assert torch.equal(y0, y1)


def test_oracle_stage_tables_match_the_product():
    """The oracle restates the UNet graph itself (oracle/unet_ref.py ENCODERS / BOTTLENECK / DECODERS, from sd/diffusion.py:543-626)
    instead of importing the product's stage tables; the two copies must agree (and the full-UNet goldens captured from the
    imported reference pin both against the reference's own module tree)."""
    from oracle import unet_ref
    from pytorch_stable_diffusion_amd import arch
    norm = lambda t: [[tuple(op) for op in st] for st in t]
    assert norm(unet_ref.ENCODERS) == norm(arch.ENCODERS)
    assert [tuple(op) for op in unet_ref.BOTTLENECK] == [tuple(op) for op in arch.BOTTLENECK]
    assert norm(unet_ref.DECODERS) == norm(arch.DECODERS)


def test_aux_oracle_stage_tables_match_the_product():
    """The VAE / CLIP oracle (oracle/aux_ref.py) carries its own statement of the two nn.Sequential bodies (sd/encoder.py:56-92,
    sd/decoder.py:235-339) and of CLIP's sizes; it imports nothing from the product.  The two copies must agree."""
    import ast
    import os
    from oracle import aux_ref
    from pytorch_stable_diffusion_amd import arch
    assert [aux_ref.parse_stage(t) for t in aux_ref.ENCODER_STAGES] == [tuple(op) for op in arch.VAE_ENCODER]
    assert [aux_ref.parse_stage(t) for t in aux_ref.DECODER_STAGES] == [tuple(op) for op in arch.VAE_DECODER]
    assert (aux_ref.CLIP_LAYERS, aux_ref.CLIP_HEADS) == (arch.CLIP_LAYERS, arch.CLIP_HEADS)
    # no module under oracle/ imports the product package
    odir = os.path.dirname(aux_ref.__file__)
    for fn in sorted(os.listdir(odir)):
        if fn.endswith(".py"):
            tree = ast.parse(open(os.path.join(odir, fn)).read())
            for node in ast.walk(tree):
                names = [a.name for a in node.names] if isinstance(node, ast.Import) else [node.module or ""] if isinstance(node, ast.ImportFrom) else []
                assert not any(n.startswith("pytorch_stable_diffusion_amd") for n in names), f"oracle/{fn} imports the product"
