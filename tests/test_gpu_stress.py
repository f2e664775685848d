"""Parity under the STRESS weight / activation law (-m gpu).

The benign synthetic law (U(+-1/sqrt(fan_in)), gamma = 1 +- 0.1) makes the UNet nearly a contraction and never produces
the statistics under which the native path's load-time algebra loses precision first: token rows whose mean is many
sigma, outlier channels, gamma far from 1, weights whose rows span decades.  This file pins

  * the LayerNorm fold (csrc/gemm.hip epilogue / partial form, csrc/b2b.hip), the composed feed-forward and the folded
    cross-attention at kernel level against fp64, at row means of 10 and 30 sigma and with x50 outlier channels;
  * block, whole-UNet and 20-step end-to-end goldens captured from the imported reference under synth.py's law="stress"
    (tests/golden/make_golden_stress.py), next to the yardstick the fixture carries: the reference module itself run in
    torch-CPU fp16.
"""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import gpu_util as G
from tests import helpers as H
from pytorch_stable_diffusion_amd import _native as N_

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stress_rows(M, Cc, mean_sigma, outliers, g):
    """token rows with per-row mean = mean_sigma x sigma (sign alternating) and `outliers` channels 50x larger"""
    x = torch.randn((M, Cc), generator=g)
    if outliers:
        idx = torch.randperm(Cc, generator=g)[:outliers]
        x[:, idx] *= 50.0
    sig = x.std(1, keepdim=True)
    sign = torch.where(torch.arange(M) % 2 == 0, 1.0, -1.0).view(M, 1)
    return x + mean_sigma * sig * sign


def _stress_weight(Nn, Cc, g):
    """rows spread log-uniformly over two decades, 1 % of them another x30, unit RMS overall"""
    w = (torch.rand((Nn, Cc), generator=g) * 2 - 1) / math.sqrt(Cc)
    s = torch.exp((torch.rand(Nn, generator=g) * 2 - 1) * math.log(10.0))
    s[torch.randperm(Nn, generator=g)[:max(1, Nn // 100)]] *= 30.0
    return w * (s / s.square().mean().sqrt()).view(Nn, 1)


def _stress_norm(Cc, g):
    gamma = torch.exp((torch.rand(Cc, generator=g) * 2 - 1) * math.log(5.0))
    beta = (torch.rand(Cc, generator=g) * 2 - 1) * 2.0
    return gamma, beta


# Error relative to the RMS of the fp64 result; bounds are <= 3x what the path measures on MI355X (round 3: 1.9e-4 / 5.1e-4 /
# 1.5e-3 / 2.1e-3 / 3.8e-3), next to the same computation through the separate LayerNorm kernel (fp32 stream in, fp16
# normalised out: 1.8e-4 .. 2.6e-4 at every setting) + plain GEMM.  The folded form multiplies the RAW stream's fp16 shadow,
# so its error grows like |x| / sigma x 2^-12 per element: a row mean of 10 sigma costs 7x the unfused error, the stress LAW's
# rows (mean <= 3 sigma, outlier channels) 2.4x -- and end to end under that law the folded path is the more accurate one
# (fewer fp16 roundings: pixel MAE 1.59e-3 folded, 1.75e-3 with SDMI_NO_LNFOLD=1; test_stress_e2e_txt2img_20_steps).
LN_FOLD_REL = {(0, 0): 6e-4, (10, 0): 4.5e-3, (30, 0): 1.2e-2, (3, 4): 1.6e-3, (10, 4): 6.4e-3}


@pytest.mark.parametrize("mean_sigma,outliers", sorted(LN_FOLD_REL))
@pytest.mark.parametrize("M,Cc,Nn", [(256, 320, 960), (128, 1280, 1280)])
def test_ln_fold_stress(M, Cc, Nn, mean_sigma, outliers):
    """LayerNorm folded into the consumer GEMM on rows with a large mean / outlier channels (sd/diffusion.py:317-321)."""
    g = torch.Generator().manual_seed(M + Cc + 7 * mean_sigma + outliers)
    x = _stress_rows(M, Cc, mean_sigma, outliers, g)
    gamma, beta = _stress_norm(Cc, g)
    w = _stress_weight(Nn, Cc, g)
    bias = torch.randn((Nn,), generator=g)
    # the stream is produced by a GEMM epilogue (identity product + fp32 residual), as in the block
    a = torch.zeros((M, Cc)).half()
    wp = torch.zeros((Cc, Cc)).half()
    ref = F.layer_norm(x.double(), (Cc,), gamma.double(), beta.double(), 1e-5) @ w.double().t() + bias.double()
    scale = ref.square().mean().sqrt().item()
    wf, gf, hf = G.ln_fold_prep(w.to(DEV), gamma.to(DEV), beta.to(DEV), bias.to(DEV))
    bn = G.gemm_tile(1)[1]
    ntn = (Cc + bn - 1) // bn
    rowstat = torch.full((M, ntn, 2), float("nan"), device=DEV)
    x32, x16 = G.igemm(a.to(DEV).view(1, M, 1, Cc), wp.to(DEV), B=1, Hs=M, Ws=1, Ho=M, Wo=1, res=x.to(DEV), out_f32=True, cfg=1,
                       want16=True, rowstat=rowstat)
    out = G.igemm(x16.view(1, M, 1, Cc), wf, B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=hf, out_f32=True, ln_stat=rowstat, ln_g=gf, ln_c=Cc)
    rel = ((out.cpu().double() - ref).square().mean().sqrt() / scale).item()
    u = G.layernorm(x32, gamma.to(DEV), beta.to(DEV))
    out2 = G.igemm(u.view(1, M, 1, Cc), w.half().to(DEV), B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=bias.to(DEV), out_f32=True)
    rel2 = ((out2.cpu().double() - ref).square().mean().sqrt() / scale).item()
    G.log_metric(test="ln_fold_stress", M=M, C=Cc, N=Nn, mean_sigma=mean_sigma, outliers=outliers, folded_rel=rel, unfused_rel=rel2)
    assert rel < LN_FOLD_REL[(mean_sigma, outliers)], f"folded rel RMS {rel:.2e} (unfused {rel2:.2e})"


@pytest.mark.parametrize("mean_sigma,outliers", [(0, 0), (10, 0), (30, 0), (3, 4)])
@pytest.mark.parametrize("partial", [0, 1])
def test_b2b_stress(partial, mean_sigma, outliers):
    """csrc/b2b.hip (out_proj + residual, then LayerNorm -> Linear / composed feed-forward) with a stressed intermediate S"""
    M, Cc, bm = 128, 320, 32
    g = torch.Generator().manual_seed(11 + partial + mean_sigma + outliers)
    a1 = torch.randn((M, Cc), generator=g).half()
    w1 = (torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)).half()
    b1 = torch.randn((Cc,), generator=g)
    r1 = _stress_rows(M, Cc, mean_sigma, outliers, g) * 2.0          # the residual dominates S: rows of S are stressed
    r2 = torch.randn((M, Cc), generator=g)
    gamma, beta = _stress_norm(Cc, g)
    w2 = _stress_weight(Cc, Cc, g)
    b2 = torch.randn((Cc,), generator=g)
    wp = _stress_weight(Cc, Cc, g).half()
    s_ref = a1.double() @ w1.double().t() + b1.double() + r1.double()
    ln = F.layer_norm(s_ref, (Cc,), gamma.double(), beta.double(), 1e-5)
    ref = (ln @ w2.double().t() + s_ref @ wp.double().t() + b2.double() + r2.double()) if partial else 0.25 * (ln @ w2.double().t() + b2.double())
    wf, _, hf = G.ln_fold_prep(w2.to(DEV), gamma.to(DEV), beta.to(DEV), b2.to(DEV))
    if partial:
        wf = torch.cat([wf, wp.to(DEV)], dim=1).contiguous()
    a1d, w1d, b1d, r1d, r2d = a1.to(DEV), w1.to(DEV), b1.to(DEV), r1.to(DEV), r2.to(DEV)
    s32 = torch.full((M, Cc), float("nan"), device=DEV)
    s16 = torch.full((M, Cc), float("nan"), dtype=torch.float16, device=DEV)
    out = torch.full((M, Cc), float("nan"), device=DEV)
    out16 = torch.full((M, Cc), float("nan"), dtype=torch.float16, device=DEV)
    d = N_.B2bDesc()
    d.a1, d.lda1, d.w1, d.b1 = a1d.data_ptr(), Cc, w1d.data_ptr(), b1d.data_ptr()
    d.r1, d.r1_f32 = r1d.data_ptr(), 1
    d.s32, d.s16 = s32.data_ptr(), s16.data_ptr()
    d.w2, d.K2, d.h2, d.partial, d.cscale = wf.data_ptr(), (640 if partial else 320), hf.data_ptr(), partial, (0.0 if partial else 0.25)
    if partial:
        d.r2, d.r2_f32 = r2d.data_ptr(), 1
        d.out, d.out_f32, d.out16 = out.data_ptr(), 1, out16.data_ptr()
    else:
        d.out, d.out_f32 = out16.data_ptr(), 0
    d.M, d.eps, d.bm = M, 1e-5, bm
    N_.check(N_.load().sdmi_op_b2b(C.byref(d), 1, None, N_.cur_stream()), "b2b")
    torch.cuda.synchronize()
    got = (out if partial else out16.float()).cpu().double()
    # the LayerNorm branch alone (the plain s Wo^T branch and the residual are exact to fp16 rounding of s)
    rel = ((got - ref).square().mean().sqrt() / ref.square().mean().sqrt()).item()
    G.log_metric(test="b2b_stress", partial=partial, mean_sigma=mean_sigma, outliers=outliers, rel=rel)
    lim = {0: 1e-3, 10: 4.5e-3, 30: 8.5e-3, 3: 1.5e-3}[mean_sigma]      # measured 3.3e-4 / 1.5e-3 / 2.7e-3 / 4.7e-4 (partial = 0)
    assert rel < lim, f"rel RMS {rel:.2e}"


def test_folded_cross_attention_stress():
    """The two-GEMM folded cross-attention with wide-range weights, gamma far from 1 and a stressed stream: probabilities
    against the reference order of operations in fp64, next to the three-kernel form's error on the same data."""
    Bn, Hh, T, S, Cc = 2, 8, 77, 64, 640
    d = Cc // Hh
    M = Bn * S
    g = torch.Generator().manual_seed(99)
    x = _stress_rows(M, Cc, 3, 6, g)
    x16 = x.half()
    gamma, beta = _stress_norm(Cc, g)
    wq = _stress_weight(Cc, Cc, g) * 0.25            # keeps the logits' spread within a few units
    wo = _stress_weight(Cc, Cc, g).half()
    bo = torch.randn((Cc,), generator=g)
    k = (torch.randn((Bn, T, Cc), generator=g) * 1.5).half()
    v = (torch.randn((Bn, T, Cc), generator=g) * torch.exp((torch.rand(Cc, generator=g) * 2 - 1) * math.log(10.0))).half()
    xn = F.layer_norm(x16.double(), (Cc,), gamma.double(), beta.double(), 1e-5)
    q = (xn @ wq.double().t()).view(Bn, S, Hh, d).transpose(1, 2)
    kh = k.double().view(Bn, T, Hh, d).transpose(1, 2)
    vh = v.double().view(Bn, T, Hh, d).transpose(1, 2)
    logits = q @ kh.transpose(-1, -2) / math.sqrt(d)
    p_ref = torch.softmax(logits, dim=-1)
    o = (p_ref @ vh).transpose(1, 2).reshape(M, Cc)
    ref = o @ wo.double().t() + bo.double() + x.double()
    delta = ref - x.double()
    qs = math.log2(math.e) / math.sqrt(d)
    w1 = torch.zeros((Bn, Hh, 128, Cc), dtype=torch.float64)
    w2 = torch.zeros((Cc, Bn, Hh, 128), dtype=torch.float64)
    for h in range(Hh):
        sl = slice(h * d, (h + 1) * d)
        w1[:, h, :T] = qs * (k.double()[:, :, sl] @ wq.double()[sl])
        w2[:, :, h, :T] = torch.einsum("cd,btd->cbt", wo.double()[:, sl], v.double()[:, :, sl])
    w1f, g1, h1 = G.ln_fold_prep(w1.view(Bn * 1024, Cc).float().to(DEV), gamma.to(DEV), beta.to(DEV), None)
    w2h = w2.view(Cc, Bn * 1024).half().to(DEV)
    xf = x16.float()
    stat = torch.stack([xf.sum(1), (xf * xf).sum(1)], dim=1).view(M, 1, 2).to(DEV)
    pr = G.igemm(x16.to(DEV).view(1, M, 1, Cc), w1f, B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=h1, ln_stat=stat, ln_g=g1, ln_c=Cc,
                 act=2, sm_valid=T, img_rows=S, w_img_stride=1024 * Cc, vec_img_stride=1024, n_out=1024)
    got_p = pr.float().cpu().view(Bn, S, Hh, 128).permute(0, 2, 1, 3)
    perr = (got_p[..., :T].double() - p_ref).abs().max().item()
    out = G.igemm(pr.view(1, M, 1, 1024), w2h, B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=bo.to(DEV), res=x.to(DEV), out_f32=True,
                  img_rows=S, w_img_stride=1024, ldw=Bn * 1024, n_out=Cc)
    rel = (((out.cpu().double() - ref).norm()) / delta.norm()).item()
    G.log_metric(test="xattn_fold_stress", prob_max_abs=perr, delta_rel_l2=rel, logit_std=float(logits.std()), logit_absmax=float(logits.abs().max()))
    assert perr < 4e-4 and rel < 2.5e-4, f"probabilities max abs {perr:.2e}, delta rel-L2 {rel:.2e}"      # measured 1.3e-4 / 7.7e-5


# ---- goldens captured from the reference under the stress law ------------------------------------------------------
def _meta():
    with open(os.path.join(H.GOLDEN, "stress_meta.json")) as f:
        return json.load(f)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.fixture(scope="module")
def stress_handle():
    meta = _meta()["blocks"]
    state = {}
    for m in meta.values():
        for k, v in H.stress_block_weights(m["prefix"]).items():
            state[k] = v.to(DEV)
    h = N_.UNetHandle(state, N_.FLAG_PARTIAL | N_.FLAG_STREAM_F32)
    h.set_context(H.seeded((2, 77, 768), _meta()["ctx_seed"]).to(DEV))
    yield h
    h.close()


# rel-L2 of the whole output / of the block's own contribution y - x: measured 0.6e-4 .. 3.9e-4 / 2.3e-4 .. 6.1e-4.  Yardstick
# in stress_meta.json: the reference block itself in torch-CPU fp16 sits at 2.3e-4 .. 5.4e-4 / 0.8e-3 .. 2.3e-3.
STRESS_BLOCK_REL_L2, STRESS_BLOCK_DELTA_REL_L2 = 9e-4, 1.8e-3


@pytest.mark.parametrize("name", sorted(_meta()["blocks"].keys()) if os.path.exists(os.path.join(H.GOLDEN, "stress_meta.json")) else [])
def test_stress_block_vs_golden(stress_handle, name):
    h = stress_handle
    m = _meta()["blocks"][name]
    ref = torch.from_numpy(H.load_npz("stress.npz")[name])
    x = H.stress_input(tuple(m["ishape"]), m["seed"])
    kind = 0 if m["kind"] == "res" else 1
    time = H.seeded((1, 1280), _meta()["time_seed"]).to(DEV) if kind == 0 else None
    out = h.run_block(m["prefix"], kind, _nhwc(x).to(DEV), time=time, out_shape=(ref.shape[0], ref.shape[2], ref.shape[3], ref.shape[1]))
    got = out.permute(0, 3, 1, 2).cpu()
    rel = H.rel_l2(got, ref)
    drel = H.rel_l2(got - x, ref - x) if ref.shape == x.shape else None
    G.log_metric(test="stress_block", name=name, rel_l2=rel, delta_rel_l2=drel, ref_fp16=m["ref_fp16_rel_l2"],
                 ref_fp16_delta=m.get("ref_fp16_delta_rel_l2"), launches=h.last_launch_count)
    assert rel < STRESS_BLOCK_REL_L2, f"{name}: rel L2 {rel:.2e}"
    if drel is not None:
        assert drel < STRESS_BLOCK_DELTA_REL_L2, f"{name}: rel L2 of y - x {drel:.2e} (torch-CPU fp16 reference: {m.get('ref_fp16_delta_rel_l2')})"


@pytest.fixture(scope="module")
def stress_handle_accurate():
    meta = _meta()["blocks"]
    state = {}
    for m in meta.values():
        for k, v in H.stress_block_weights(m["prefix"]).items():
            state[k] = v.to(DEV)
    h = N_.UNetHandle(state, N_.FLAG_PARTIAL | N_.FLAG_STREAM_F32 | N_.FLAG_ACCURATE)
    h.set_context(H.seeded((2, 77, 768), _meta()["ctx_seed"]).to(DEV))
    yield h
    h.close()


@pytest.mark.parametrize("name", sorted(_meta()["blocks"].keys()) if os.path.exists(os.path.join(H.GOLDEN, "stress_meta.json")) else [])
def test_accurate_mode_stress_block_vs_golden(stress_handle_accurate, name):
    """the same blocks through the accurate mode's kernels (SDMI_FLAG_ACCURATE: wide activation operands, no folds, the plain
    attention kernel): what is left is the weights' fp16 rounding -- every block at most HALF the default mode's bound"""
    h = stress_handle_accurate
    m = _meta()["blocks"][name]
    ref = torch.from_numpy(H.load_npz("stress.npz")[name])
    x = H.stress_input(tuple(m["ishape"]), m["seed"])
    kind = 0 if m["kind"] == "res" else 1
    time = H.seeded((1, 1280), _meta()["time_seed"]).to(DEV) if kind == 0 else None
    out = h.run_block(m["prefix"], kind, _nhwc(x).to(DEV), time=time, out_shape=(ref.shape[0], ref.shape[2], ref.shape[3], ref.shape[1]))
    got = out.permute(0, 3, 1, 2).cpu()
    rel = H.rel_l2(got, ref)
    drel = H.rel_l2(got - x, ref - x) if ref.shape == x.shape else None
    G.log_metric(test="accurate_stress_block", name=name, rel_l2=rel, delta_rel_l2=drel, launches=h.last_launch_count)
    assert rel < STRESS_BLOCK_REL_L2 / 2, f"{name}: rel L2 {rel:.2e}"
    if drel is not None:
        assert drel < STRESS_BLOCK_DELTA_REL_L2 / 2, f"{name}: rel L2 of y - x {drel:.2e}"


@pytest.fixture(scope="module")
def stress_unet():
    from pytorch_stable_diffusion_amd import arch, synth
    from pytorch_stable_diffusion_amd.diffusion import Diffusion
    m = Diffusion(stream_f32=True)
    m.load_state_dict(synth.synth_state_dict(arch.diffusion_manifest(), law="stress"), strict=True)
    m.to(DEV)
    yield m
    m._drop_handle()


def test_stress_full_unet_vs_golden(stress_unet):
    """one whole UNet forward at 64x64, t = 980, stress-law weights, against the reference's own Diffusion.forward"""
    from oracle import ddpm_ref
    ref = torch.from_numpy(H.load_npz("stress.npz")["unet_64_t980"])
    lat = H.seeded((1, 4, 64, 64), 0).repeat(2, 1, 1, 1).to(DEV)
    got = stress_unet(lat, H.seeded((2, 77, 768), 1).to(DEV), ddpm_ref.time_embedding(980).to(DEV)).cpu()
    rel = H.rel_l2(got, ref)
    G.log_metric(test="stress_unet", rel_l2=rel, max_abs=(got - ref).abs().max().item())
    assert rel < 4e-3, f"rel L2 {rel:.2e}"


def _floor(steps):
    """tests/golden/stress_floor.py: pixel MAE of the ORACLE (fp32 arithmetic) through the same loop with one operand class
    stored in fp16 -- what fp16 storage alone costs under this law, whatever the kernels do"""
    with open(os.path.join(H.GOLDEN, "stress_floor.json" if steps == 20 else f"stress_floor{steps}.json")) as f:
        return {k: v["pixel_mae"] for k, v in json.load(f)["runs"].items()}


def _stress_models(accurate=False):
    from pytorch_stable_diffusion_amd import arch, model_loader, synth
    sds = model_loader.synthetic_state_dicts(("clip", "decoder", "encoder"))
    sds["diffusion"] = synth.synth_state_dict(arch.diffusion_manifest(), law="stress")
    return model_loader.preload_models_from_state_dicts(sds, DEV, accurate=accurate)


# Stated tolerance.  north_star's pixel MAE < 1e-3 is met on the benign law at BASELINE's own step counts (tests/test_gpu_e2e.py).
# Under the STRESS law it is not reachable by ANY path that hands fp16 operands to the matrix cores: the oracle itself (fp32
# arithmetic, pinned to the reference) with only the activation operand of every conv / linear rounded to fp16 lands at
# 1.59e-3 @20 steps, with weights + activations + attention operands in fp16 at 1.40e-3 (weights alone 4.7e-4, attention
# operands alone 3.5e-5; tests/golden/stress_floor.json, script beside it) -- and the per-op attribution
# (test_stress_per_block_attribution, profiles/r04_stress_attribution.json) shows no dominant layer: every one of the 45 stage
# ops adds 1.6e-4 .. 6.7e-4.  The HIP path measures 1.59e-3 .. 1.78e-3 / 4-6 LSB @20 (two builds whose single UNet forward is
# equally far from the reference, 1.40e-3 and 1.41e-3 rel-L2: the 20 steps amplify WHICH roundings are made, the two floor runs
# differ by as much): AT that floor.  Asserted: within 1.25x the larger fp16-activation floor (and the uint8 image within 6
# LSB); the torch-CPU fp16 reference sits at 2.86e-3 / 10 LSB.
STRESS_FLOOR_MARGIN = 1.25


def _floor_bound(fl):
    return STRESS_FLOOR_MARGIN * max(fl["weights + activations + attention fp16"], fl["activation operands fp16"])


def test_stress_e2e_txt2img_20_steps():
    """pipeline.generate() with the stress-law UNet (benign CLIP / VAE decoder), 512x512, 20 steps, CFG 7.5, seed 42,
    against the reference's own generate() on the CPU."""
    from pytorch_stable_diffusion_amd import pipeline
    from pytorch_stable_diffusion_amd.tokenizer import StubTokenizer
    gold = np.load(os.path.join(H.GOLDEN, "stress_e2e.npz"))
    img = pipeline.generate(prompt="a dog", uncond_prompt="", input_image=None, strength=0.8, do_cfg=True, cfg_scale=7.5,
                            sampler_name="ddpm", n_inference_steps=20, models=_stress_models(), seed=42, device=DEV, idle_device=None,
                            tokenizer=StubTokenizer())
    ref = gold["txt20_u8"]
    mae = float(np.abs(img.astype(np.float64) - ref.astype(np.float64)).mean() / 255.0)
    mx = int(np.abs(img.astype(np.int32) - ref.astype(np.int32)).max())
    fl = _floor(20)
    yard = _meta().get("e2e20_ref_fp16", {"pixel_mae": 2.858e-3, "u8_max_diff": 10})
    G.log_metric(test="stress_e2e20", pixel_mae=mae, u8_max_diff=mx, floor=fl, ref_fp16_mae=yard["pixel_mae"], ref_fp16_u8=yard["u8_max_diff"])
    bound = _floor_bound(fl)
    assert mae < bound and mx <= 6, f"pixel MAE {mae:.2e} (bound {bound:.2e} = {STRESS_FLOOR_MARGIN} x the fp16-storage floor), uint8 max diff {mx}"


def test_stress_e2e_txt2img_50_steps():
    """the same at BASELINE configs[1]'s own 50 steps (tests/golden/stress_e2e50.npz: uint8 image + the decoder's float image at
    full resolution); floor from tests/golden/stress_floor50.json"""
    from pytorch_stable_diffusion_amd import pipeline
    from pytorch_stable_diffusion_amd.tokenizer import StubTokenizer
    gold = np.load(os.path.join(H.GOLDEN, "stress_e2e50.npz"))
    models = _stress_models()

    class Tap:
        def __init__(self, inner):
            self.inner, self.last = inner, None

        def to(self, d):
            self.inner.to(d)
            return self

        def __call__(self, *a):
            out = self.inner(*a)
            self.last = out.clone()            # generate() rescales ``out`` in place
            return out

    models["decoder"] = Tap(models["decoder"])
    img = pipeline.generate(prompt="a dog", uncond_prompt="", input_image=None, strength=0.8, do_cfg=True, cfg_scale=7.5,
                            sampler_name="ddpm", n_inference_steps=50, models=models, seed=42, device=DEV, idle_device=None,
                            tokenizer=StubTokenizer())
    ref = gold["u8"]
    mae = float(np.abs(img.astype(np.float64) - ref.astype(np.float64)).mean() / 255.0)
    mx = int(np.abs(img.astype(np.int32) - ref.astype(np.int32)).max())
    fmae = H.float_image_mae(models["decoder"].last[0], gold["float_u16"])       # FULL resolution, before the uint8 cast, [0,1] scale
    fl = _floor(50)
    G.log_metric(test="stress_e2e50", pixel_mae=mae, float_mae_full_res=fmae, u8_max_diff=mx, floor=fl)
    bound = _floor_bound(fl)
    assert mae < bound and fmae < bound and mx <= 8, f"pixel MAE {mae:.2e} / float {fmae:.2e} (bound {bound:.2e}), uint8 max diff {mx}"


# ---- ACCURATE mode (include/sdmi.h SDMI_FLAG_ACCURATE, Diffusion(accurate=True)): north_star's tolerance under the stress law ------
# Every GEMM / conv multiplies its activation operand as a hi + lo fp16 pair read from fp32 tensors (csrc/gemm.hip igemm_kernel<..,
# ACC>); what is left of the fp16 path's error is the weights' own fp16 rounding, which the floor experiment prices at 4.7e-4 @20 /
# 4.8e-4 @50 (stress_floor*.json "weights fp16") plus 3.5e-5 for the attention kernel's fp16 q / k / v / p.  Asserted: pixel MAE
# < 1e-3 -- north_star's number, not a floor-derived bound -- on the uint8 image and on the full-resolution float image.
ACCURATE_PIXEL_MAE = 1e-3


@pytest.fixture(scope="module")
def accurate_models():
    m = _stress_models(accurate=True)
    yield m
    m["diffusion"]._drop_handle()


def test_accurate_mode_stress_full_unet_vs_golden(accurate_models):
    """one UNet forward, stress law, 64x64, t = 980 against the reference's own Diffusion.forward: the default mode measures
    1.40e-3 rel-L2 (weights, activations and attention operands in fp16); with wide activations what is left is the weights' rounding"""
    from oracle import ddpm_ref
    ref = torch.from_numpy(H.load_npz("stress.npz")["unet_64_t980"])
    lat = H.seeded((1, 4, 64, 64), 0).repeat(2, 1, 1, 1).to(DEV)
    unet = accurate_models["diffusion"]
    got = unet(lat, H.seeded((2, 77, 768), 1).to(DEV), ddpm_ref.time_embedding(980).to(DEV)).cpu()
    rel = H.rel_l2(got, ref)
    # the floor of ONE forward with fp16 weights (tests/golden/stress_floor_forward.py: the fp32 oracle with only the weights and the
    # attention operands rounded): 9.97e-4 -- in a single forward the weights' rounding alone costs what the activations' does
    # (9.8e-4; all three classes 1.40e-3, what the default mode measures); it is over the LOOP that they part (4.7e-4 against
    # 1.6e-3 at 20 steps, stress_floor.json).  Measured: 9.80e-4, on that floor.
    with open(os.path.join(H.GOLDEN, "stress_floor_forward.json")) as f:
        floor = json.load(f)["runs"]["weights + attention fp16"]["rel_l2"]
    G.log_metric(test="accurate_stress_unet", rel_l2=rel, floor=floor, max_abs=(got - ref).abs().max().item(), launches=unet.handle().last_launch_count)
    assert rel < 1.1 * floor, f"rel L2 {rel:.2e} (floor of fp16 weights {floor:.2e})"


def test_accurate_mode_stress_e2e_txt2img_20_steps(accurate_models):
    from pytorch_stable_diffusion_amd import pipeline
    from pytorch_stable_diffusion_amd.tokenizer import StubTokenizer
    gold = np.load(os.path.join(H.GOLDEN, "stress_e2e.npz"))
    img = pipeline.generate(prompt="a dog", uncond_prompt="", input_image=None, strength=0.8, do_cfg=True, cfg_scale=7.5,
                            sampler_name="ddpm", n_inference_steps=20, models=accurate_models, seed=42, device=DEV, idle_device=None,
                            tokenizer=StubTokenizer())
    ref = gold["txt20_u8"]
    mae = float(np.abs(img.astype(np.float64) - ref.astype(np.float64)).mean() / 255.0)
    mx = int(np.abs(img.astype(np.int32) - ref.astype(np.int32)).max())
    G.log_metric(test="accurate_stress_e2e20", pixel_mae=mae, u8_max_diff=mx, floor_weights_fp16=_floor(20).get("weights fp16"))
    assert mae < ACCURATE_PIXEL_MAE and mx <= 3, f"pixel MAE {mae:.2e}, uint8 max diff {mx}"


def test_accurate_mode_stress_e2e_txt2img_50_steps(accurate_models):
    """BASELINE configs[1]'s own 50 steps under the stress law: the default mode sits at 1.47e-3 (the fp16-activation floor),
    the accurate mode has to meet north_star's 1e-3 on the uint8 image AND the full-resolution float image"""
    from pytorch_stable_diffusion_amd import pipeline
    from pytorch_stable_diffusion_amd.tokenizer import StubTokenizer
    gold = np.load(os.path.join(H.GOLDEN, "stress_e2e50.npz"))
    models = dict(accurate_models)
    last = []

    class Tap:
        def __init__(self, inner):
            self.inner = inner

        def to(self, d):
            self.inner.to(d)
            return self

        def __call__(self, *a):
            out = self.inner(*a)
            last.append(out.clone())
            return out

    models["decoder"] = Tap(models["decoder"])
    img = pipeline.generate(prompt="a dog", uncond_prompt="", input_image=None, strength=0.8, do_cfg=True, cfg_scale=7.5,
                            sampler_name="ddpm", n_inference_steps=50, models=models, seed=42, device=DEV, idle_device=None,
                            tokenizer=StubTokenizer())
    ref = gold["u8"]
    mae = float(np.abs(img.astype(np.float64) - ref.astype(np.float64)).mean() / 255.0)
    mx = int(np.abs(img.astype(np.int32) - ref.astype(np.int32)).max())
    fmae = H.float_image_mae(last[-1][0], gold["float_u16"])
    G.log_metric(test="accurate_stress_e2e50", pixel_mae=mae, float_mae_full_res=fmae, u8_max_diff=mx, floor_weights_fp16=_floor(50).get("weights fp16"))
    assert mae < ACCURATE_PIXEL_MAE and fmae < ACCURATE_PIXEL_MAE and mx <= 3, f"pixel MAE {mae:.2e} / float {fmae:.2e}, uint8 max diff {mx}"


def test_stress_per_block_attribution(stress_unet):
    """Where the stress-law error comes from: every stage op of the UNet (ResBlock / AttentionBlock / down / up conv / output
    layer; sd/diffusion.py:543-626,714-748) is run ALONE on the HIP path (sdmi_unet_run_block) on the ORACLE's input for that
    op -- one oracle forward at 64x64, t = 980, traced -- and compared with the oracle's output of the same op.  The table
    (rel-L2 of the whole output, and of the op's own contribution y - x where it adds its input back) goes to
    gpurun_out/stress_attribution.json / profiles/; asserted: no op above 3x the worst the fixture blocks measure, and the
    per-op errors, root-sum-squared along the chain with the skip structure ignored, stay in the range of the whole-UNet error."""
    from oracle import ddpm_ref, unet_ref
    from pytorch_stable_diffusion_amd import arch, synth
    sd = synth.synth_state_dict(arch.diffusion_manifest(), law="stress")
    lat = H.seeded((1, 4, 64, 64), 0).repeat(2, 1, 1, 1)
    ctx = H.seeded((2, 77, 768), 1)
    temb = ddpm_ref.time_embedding(980)
    unet_ref.TRACE = []
    try:
        with torch.no_grad():
            time_vec = unet_ref.time_mlp(sd, temb)
            body = unet_ref.unet_body(sd, lat, ctx, time_vec)
            eps_ref = unet_ref.output_layer(sd, body)
        trace = unet_ref.TRACE
    finally:
        unet_ref.TRACE = None
    h = stress_unet.handle()
    stress_unet.set_context(ctx.to(DEV))
    tv = time_vec.to(DEV)
    rows = []
    prev_c = 4
    for p, op, x_in, x_out in trace:
        cin = x_in.shape[1]
        x1 = None
        if op[0] == "res" and cin != prev_c and p.startswith("unet.decoders") and p.endswith(".0"):
            x0, x1 = x_in[:, :prev_c], x_in[:, prev_c:]            # cat(x, skip): sd/diffusion.py:671
        else:
            x0 = x_in
        prev_c = x_out.shape[1]
        if op[0] == "conv" and op[1] == 4:
            continue                                               # the 4 -> 320 stem reads NCHW latents (covered by the full forward)
        kind, arg = {"res": (0, 1), "attn": (1, 1), "up": (2, 1), "conv": (3, op[3] if op[0] == "conv" else 1)}[op[0]]
        oshape = (x_out.shape[0], x_out.shape[2], x_out.shape[3], x_out.shape[1])
        got = h.run_block(p, kind, _nhwc(x0).to(DEV), None if x1 is None else _nhwc(x1).to(DEV), time=tv if kind == 0 else None,
                          arg=arg, out_shape=oshape).permute(0, 3, 1, 2).cpu()
        rel = H.rel_l2(got, x_out)
        drel = H.rel_l2(got - x_in, x_out - x_in) if (x_out.shape == x_in.shape and x1 is None) else None
        rows.append(dict(op=p, kind=op[0], shape=list(x_out.shape), rel_l2=rel, delta_rel_l2=drel, out_rms=float(x_out.square().mean().sqrt()),
                         launches=h.last_launch_count))
    got = h.run_block("final", 4, _nhwc(body).to(DEV), out_shape=tuple(eps_ref.shape)).cpu()
    rows.append(dict(op="final", kind="out", shape=list(eps_ref.shape), rel_l2=H.rel_l2(got, eps_ref), delta_rel_l2=None,
                     out_rms=float(eps_ref.square().mean().sqrt()), launches=h.last_launch_count))
    full = stress_unet(lat.to(DEV), ctx.to(DEV), temb.to(DEV)).cpu()
    full_rel = H.rel_l2(full, eps_ref)
    out = dict(weights="synth law=stress", latent="64x64, t=980", full_unet_rel_l2=full_rel, ops=rows)
    os.makedirs(G.OUT, exist_ok=True)
    with open(os.path.join(G.OUT, "stress_attribution.json"), "w") as f:
        json.dump(out, f, indent=1)
    worst = sorted(rows, key=lambda r: -r["rel_l2"])[:5]
    G.log_metric(test="stress_attribution", full_unet_rel_l2=full_rel, worst=[(r["op"], r["rel_l2"]) for r in worst])
    assert len(rows) >= 40
    for r in rows:
        assert r["rel_l2"] < 2.7e-3, f"{r['op']}: rel L2 {r['rel_l2']:.2e}"


@pytest.mark.parametrize("mean_sigma,expect", [(0, 0), (3, 0), (10, 1), (30, 1)])
def test_ln_fold_guard_counts_rows_beyond_the_threshold(mean_sigma, expect):
    """The guard of the LayerNorm fold (sdmi_gemm_desc::ln_guard, csrc/gemm.hip prologue): a folded GEMM counts the rows whose
    |mean| exceeds 8 sigma -- each row once, whatever the tiling -- so the caller can repeat its loop through the separate
    LayerNorm kernel.  Rows of the stress law (<= 3 sigma) never trip it."""
    M, Cc, Nn = 256, 640, 640
    g = torch.Generator().manual_seed(17 + mean_sigma)
    x = _stress_rows(M, Cc, mean_sigma, 0, g)
    gamma, beta = _stress_norm(Cc, g)
    w = _stress_weight(Nn, Cc, g)
    bias = torch.randn((Nn,), generator=g)
    wf, gf, hf = G.ln_fold_prep(w.to(DEV), gamma.to(DEV), beta.to(DEV), bias.to(DEV))
    a = torch.zeros((M, Cc)).half()
    wp = torch.zeros((Cc, Cc)).half()
    for cfg in (1, 9):                                       # 128x128 and 64x64 tiles: the count does not depend on the tiling
        bn = G.gemm_tile(cfg)[1]
        ntn = (Cc + bn - 1) // bn
        rowstat = torch.full((M, ntn, 2), float("nan"), device=DEV)
        x32, x16 = G.igemm(a.to(DEV).view(1, M, 1, Cc), wp.to(DEV), B=1, Hs=M, Ws=1, Ho=M, Wo=1, res=x.to(DEV), out_f32=True, cfg=cfg,
                           want16=True, rowstat=rowstat)
        cnt = torch.zeros((1,), dtype=torch.int32, device=DEV)
        G.igemm(x16.view(1, M, 1, Cc), wf, B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=hf, out_f32=True, ln_stat=rowstat, ln_g=gf, ln_c=Cc, cfg=cfg,
                ln_guard=cnt)
        assert int(cnt.item()) == (M if expect else 0), f"cfg {cfg}: {int(cnt.item())} rows counted at {mean_sigma} sigma"


def test_ln_fold_guard_fallback_path(stress_handle, monkeypatch):
    """Handle level: no row of the stress-law blocks trips the guard; with the fold switched off (what the fallback does) the
    block runs through the separate LayerNorm kernel -- more launches, the same result within the block bounds -- and
    Diffusion.denoise_native repeats its loop unfused, on the same noise stream, when the counter is not zero."""
    h = stress_handle
    m = _meta()["blocks"]["attn_8_80_s16"]
    ref = torch.from_numpy(H.load_npz("stress.npz")["attn_8_80_s16"])
    x = H.stress_input(tuple(m["ishape"]), m["seed"])
    oshape = (ref.shape[0], ref.shape[2], ref.shape[3], ref.shape[1])
    h.ln_guard(reset=True)
    y1 = h.run_block(m["prefix"], 1, _nhwc(x).to(DEV), out_shape=oshape).permute(0, 3, 1, 2).cpu()
    n1 = h.last_launch_count
    assert h.ln_guard(reset=True) == 0
    h.ln_guard(fold_on=False)
    try:
        y2 = h.run_block(m["prefix"], 1, _nhwc(x).to(DEV), out_shape=oshape).permute(0, 3, 1, 2).cpu()
        n2 = h.last_launch_count
    finally:
        h.ln_guard(fold_on=True)
    assert n2 > n1, (n1, n2)
    assert H.rel_l2(y2, ref) < STRESS_BLOCK_REL_L2 and H.rel_l2(y1, ref) < STRESS_BLOCK_REL_L2
    y3 = h.run_block(m["prefix"], 1, _nhwc(x).to(DEV), out_shape=oshape).permute(0, 3, 1, 2).cpu()
    assert torch.equal(y1, y3)                                   # switched back: the folded path again, bit for bit


def test_denoise_native_repeats_unfused_when_the_guard_fires(monkeypatch):
    from pytorch_stable_diffusion_amd import arch, synth
    from pytorch_stable_diffusion_amd.ddpm import DDPMSampler
    from pytorch_stable_diffusion_amd.diffusion import Diffusion
    model = Diffusion(stream_f32=True)
    model.load_state_dict(synth.synth_state_dict(arch.diffusion_manifest()), strict=True)
    model.to(DEV)
    try:
        ctx = H.seeded((2, 77, 768), 1).to(DEV)
        lat = H.seeded((1, 4, 16, 16), 3).to(DEV)

        def run():
            smp = DDPMSampler(torch.Generator().manual_seed(5))
            smp.set_inference_timesteps(4)
            return model.denoise_native(lat, ctx, smp, smp.timesteps.tolist(), True, 7.5).cpu()

        base = run()
        assert model.ln_guard_hits == 0
        h = model.handle()
        real, calls = h.ln_guard, []

        def fake(reset=True, fold_on=None):
            calls.append(fold_on)
            v = real(reset=reset, fold_on=fold_on)
            return 7 if len(calls) == 2 else v                  # the read after the first loop reports hits

        monkeypatch.setattr(h, "ln_guard", fake)
        again = run()
        assert model.ln_guard_hits == 7 and False in calls and calls[-1] is True      # unfused loop ran, the fold is back on
        # the separate-LayerNorm path: other roundings, the same latents (measured 2.0e-3 rel-L2 after four CFG 7.5 steps at 16x16)
        assert not torch.equal(again, base) and H.rel_l2(again, base) < 6e-3
        monkeypatch.setattr(h, "ln_guard", real)
        assert torch.equal(run(), base)
    finally:
        model._drop_handle()
