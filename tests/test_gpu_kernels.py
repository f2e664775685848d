"""Kernel-level parity (-m gpu): each HIP kernel through the C ABI vs a plain fp32/fp64 torch-CPU
restatement of the same op on the same fp16-rounded inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests import gpu_util as G
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"
from pytorch_stable_diffusion_amd import _native as N_  # noqa: E402


def _cfgs():
    from pytorch_stable_diffusion_amd import _native as N
    return list(range(N.load().sdmi_gemm_num_configs()))


def test_library_loaded_is_in_tree():
    from pytorch_stable_diffusion_amd import _native as N
    lib = N.load()
    assert lib.sdmi_version() >= 100
    assert "pytorch_stable_diffusion_amd/lib/libsdmi.so" in N.lib_path()


def _plain_cfgs():
    from pytorch_stable_diffusion_amd import _native as N
    lib = N.load()
    # "h..." (halo-reuse), "g..." (fused GroupNorm) and the 160-wide tiles are 3x3-conv configs with their own tests
    names = [lib.sdmi_gemm_config_name(i).decode() for i in range(lib.sdmi_gemm_num_configs())]
    return [i for i, nm in enumerate(names) if nm[0] not in "hg" and "x160" not in nm]


@pytest.mark.parametrize("cfg", _plain_cfgs())
def test_gemm_exact_integers(cfg):
    """MFMA fragment layouts: small-integer operands make every product/sum exact, so the result must
    equal the integer matmul bit for bit (asymmetric operands catch transposed/permuted layouts)."""
    torch.manual_seed(cfg)
    M, Nn, K = 200, 136, 128           # ragged M and N exercise the bounds paths
    a = torch.randint(-2, 3, (M, K)).to(torch.float16)
    w = torch.randint(-2, 3, (Nn, K)).to(torch.float16)
    w[:, 0] = torch.arange(Nn) % 3 - 1    # asymmetric
    ref = a.double() @ w.double().t()
    out = G.igemm(a.to(DEV).view(1, M, 1, K), w.to(DEV), B=1, Hs=M, Ws=1, Ho=M, Wo=1, cfg=cfg)
    assert torch.equal(out.cpu().double(), ref), f"cfg {cfg}: max diff {(out.cpu().double()-ref).abs().max()}"


@pytest.mark.parametrize("M,Nn,K,ksplit", [(8192, 320, 320, 1), (512, 1280, 2560, 4), (128, 1280, 1280, 8),
                                           (154, 640, 768, 1), (2048, 1920, 640, 1)])
def test_gemm_random_bias_residual(M, Nn, K, ksplit):
    g = torch.Generator().manual_seed(M + Nn + K)
    a = (torch.randn((M, K), generator=g)).to(torch.float16)
    w = (torch.randn((Nn, K), generator=g) / math.sqrt(K)).to(torch.float16)
    bias = torch.randn((Nn,), generator=g)
    res = torch.randn((M, Nn), generator=g)
    ref = a.double() @ w.double().t() + bias.double() + res.double()
    for cfg in [-1] + _plain_cfgs():
        out, out16 = G.igemm(a.to(DEV).view(1, M, 1, K), w.to(DEV), B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=bias.to(DEV),
                             res=res.to(DEV), out_f32=True, cfg=cfg, ksplit=ksplit, want16=True)
        err = (out.cpu().double() - ref).abs().max().item()
        G.log_metric(test="gemm_random", M=M, N=Nn, K=K, cfg=cfg, ksplit=ksplit, max_abs_err=err)
        assert err < 2e-3, f"cfg {cfg}: fp32-out max abs err {err}"
        assert (out16.cpu().double() - ref).abs().max().item() < 8e-3
        # fp16 residual / fp16 output path
        out_h = G.igemm(a.to(DEV).view(1, M, 1, K), w.to(DEV), B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=bias.to(DEV),
                        res=res.to(DEV).half(), out_f32=False, cfg=cfg, ksplit=ksplit)
        ref_h = a.double() @ w.double().t() + bias.double() + res.half().double()
        assert (out_h.cpu().double() - ref_h).abs().max().item() < 8e-3


@pytest.mark.parametrize("M,Nn,K", [(200, 136, 256), (2048, 640, 640), (512, 1280, 1280)])
def test_every_tile_config_gives_the_same_bits_without_split_k(M, Nn, K):
    """A plan may change the tile config of a shape (tools/retune.sh); without split-K that must not change a single bit: every
    output element is accumulated over K in k16-step order by ONE wave whatever the tile shape, ring depth, producer-wave
    count or K-steps per barrier.  (Split-K regroups the sum; the halo kernels walk K chunk-major: both have their own tests.)"""
    g = torch.Generator().manual_seed(M * 7 + K)
    a = torch.randn((M, K), generator=g).to(torch.float16).to(DEV)
    w = (torch.randn((Nn, K), generator=g) / math.sqrt(K)).to(torch.float16).to(DEV)
    bias = torch.randn((Nn,), generator=g).to(DEV)
    res = torch.randn((M, Nn), generator=g).to(DEV)
    ref, ref_cfg = None, None
    for cfg in _plain_cfgs():
        out = G.igemm(a.view(1, M, 1, K), w, B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=bias, res=res, out_f32=True, cfg=cfg, ksplit=1)
        if ref is None:
            ref, ref_cfg = out.clone(), cfg
        else:
            assert torch.equal(out, ref), f"config {cfg} differs from config {ref_cfg}: max {(out - ref).abs().max().item():.3e}"


def _conv_ref(x_nhwc, w_oihw, stride, ups):
    x = x_nhwc.float().permute(0, 3, 1, 2)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    ks = w_oihw.shape[-1]
    return F.conv2d(x.double(), w_oihw.double(), stride=stride, padding=1 if ks == 3 else 0).permute(0, 2, 3, 1)


@pytest.mark.parametrize("case", [
    dict(B=2, H=16, W=16, C0=128, C1=0, Co=128, ks=3, stride=1, ups=0),
    dict(B=2, H=16, W=16, C0=64, C1=0, Co=192, ks=3, stride=2, ups=0),
    dict(B=2, H=8, W=8, C0=128, C1=0, Co=64, ks=3, stride=1, ups=1),
    dict(B=2, H=12, W=12, C0=128, C1=64, Co=320, ks=3, stride=1, ups=0),     # concat, ragged tiles
    dict(B=1, H=10, W=6, C0=64, C1=128, Co=72, ks=1, stride=1, ups=0),       # 1x1 on concat
    dict(B=2, H=9, W=9, C0=64, C1=0, Co=64, ks=3, stride=2, ups=0),          # odd size, stride 2
    dict(B=1, H=64, W=64, C0=64, C1=64, Co=192, ks=3, stride=1, ups=0),      # halo-reuse kernels, concat, W=64
    dict(B=2, H=32, W=32, C0=128, C1=0, Co=64, ks=3, stride=1, ups=0),       # halo-reuse, W=32
    dict(B=2, H=8, W=8, C0=128, C1=0, Co=128, ks=3, stride=1, ups=0),        # halo-reuse, 8x8 map (one image per tile)
    dict(B=4, H=8, W=8, C0=192, C1=64, Co=160, ks=3, stride=1, ups=0),       # 8x8 maps: two whole images per 128-row tile, concat
    dict(B=2, H=16, W=16, C0=256, C1=0, Co=320, ks=3, stride=1, ups=0),      # 16x16, N = 320 (160-wide tiles)
    dict(B=1, H=16, W=16, C0=128, C1=0, Co=80, ks=3, stride=1, ups=1),       # x2-upsampled input through the halo kernels
])
def test_conv_implicit_gemm(case):
    c = case
    g = torch.Generator().manual_seed(c["C0"] * 7 + c["Co"])
    x0 = torch.randn((c["B"], c["H"], c["W"], c["C0"]), generator=g).half()
    x1 = torch.randn((c["B"], c["H"], c["W"], c["C1"]), generator=g).half() if c["C1"] else None
    cin = c["C0"] + c["C1"]
    w = (torch.randn((c["Co"], cin, c["ks"], c["ks"]), generator=g) / math.sqrt(cin * c["ks"] ** 2)).half().float()
    xin = x0 if x1 is None else torch.cat([x0, x1], -1)
    ref = _conv_ref(xin, w, c["stride"], c["ups"])
    Ho, Wo = ref.shape[1], ref.shape[2]
    wp = G.pack_conv(w.to(DEV))
    n_halo = 0
    for cfg in [-1] + _cfgs():
        for ksplit in (1, 3):
            try:
                out = G.igemm(x0.to(DEV), wp, B=c["B"], Hs=c["H"], Ws=c["W"], Ho=Ho, Wo=Wo, ks=c["ks"], stride=c["stride"],
                              ups=c["ups"], a1=None if x1 is None else x1.to(DEV), out_f32=True, cfg=cfg, ksplit=ksplit)
            except ValueError as exc:       # halo-reuse configs only accept 3x3 s1 convs tiled by whole image rows
                assert "not applicable" in str(exc) or "LDS" in str(exc), exc
                continue
            n_halo += cfg >= 0 and N_.load().sdmi_gemm_config_name(cfg).decode()[0] == "h"
            err = (out.cpu().double().view(ref.shape) - ref).abs().max().item()
            G.log_metric(test="conv", case=str(c), cfg=cfg, ksplit=ksplit, max_abs_err=err)
            assert err < 2e-3, f"{c} cfg {cfg} ksplit {ksplit}: max abs err {err}"
    if c["ks"] == 3 and c["stride"] == 1 and c["W"] << c["ups"] in (8, 16, 32, 64):
        assert n_halo > 0, "no halo-reuse config ran on an eligible conv"


@pytest.mark.parametrize("case", [
    dict(B=2, H=16, W=16, C0=128, C1=0, Co=128, ks=3, stride=1, ups=0, X0=0, X1=0),
    dict(B=2, H=12, W=12, C0=128, C1=64, Co=320, ks=3, stride=1, ups=0, X0=0, X1=0),     # concat, ragged tiles
    dict(B=1, H=10, W=6, C0=64, C1=128, Co=72, ks=1, stride=1, ups=0, X0=0, X1=0),       # 1x1 on concat
    dict(B=2, H=9, W=9, C0=64, C1=0, Co=64, ks=3, stride=2, ups=0, X0=0, X1=0),          # stride 2
    dict(B=2, H=8, W=8, C0=128, C1=0, Co=64, ks=3, stride=1, ups=1, X0=0, X1=0),         # x2-upsampled input
    dict(B=2, H=12, W=12, C0=128, C1=0, Co=192, ks=3, stride=1, ups=0, X0=128, X1=64),   # fused 1x1 skip segment over two sources
    dict(B=2, H=32, W=32, C0=320, C1=0, Co=320, ks=1, stride=1, ups=0, X0=0, X1=0),      # plain GEMM path
])
def test_accurate_gemm_wide_activation_operand(case):
    """The accurate mode's GEMM (include/sdmi.h sdmi_gemm_desc::accurate, csrc/gemm.hip igemm_kernel<.., ACC>): the activation operand
    read from fp32 tensors and multiplied as a hi + lo fp16 pair.  Inputs with a wide dynamic range (rows scaled over three decades,
    a large common offset: what fp16 storage loses most on); against the fp64 product of the SAME fp32 activations and fp16
    weights the error has to be that of fp32 accumulation (~1e-6 of the output scale), where the fp16-operand kernel on the rounded
    activations sits three orders above.  Every tile built with the variant, one-pass and split-K, and the mode's own plan."""
    c = case
    g = torch.Generator().manual_seed(c["C0"] * 3 + c["Co"] + c["X0"])
    scale = torch.exp(torch.rand((c["B"], c["H"], c["W"], 1), generator=g) * 6.9 - 3.45)          # 0.03 .. 30 per pixel
    mk = lambda ch: (torch.randn((c["B"], c["H"], c["W"], ch), generator=g) * scale + 3.0 * scale)
    x0 = mk(c["C0"])
    x1 = mk(c["C1"]) if c["C1"] else None
    cin = c["C0"] + c["C1"]
    w = (torch.randn((c["Co"], cin, c["ks"], c["ks"]), generator=g) / math.sqrt(cin * c["ks"] ** 2)).half().float()
    xin = x0 if x1 is None else torch.cat([x0, x1], -1)
    ref = _conv_ref_f64(xin, w, c["stride"], c["ups"])
    wp = G.pack_conv(w.to(DEV))
    kw = {}
    if c["X0"]:
        e0 = mk(c["X0"])
        e1 = mk(c["X1"]) if c["X1"] else None
        ws = (torch.randn((c["Co"], c["X0"] + c["X1"], 1, 1), generator=g) / math.sqrt(c["X0"] + c["X1"])).half().float()
        ref = ref + _conv_ref_f64(e0 if e1 is None else torch.cat([e0, e1], -1), ws, 1, 0)
        wp = torch.cat([wp, G.pack_conv(ws.to(DEV))], 1).contiguous()
        kw = dict(x0=e0.half().to(DEV), x0f=e0.to(DEV), x1=None if e1 is None else e1.half().to(DEV), x1f=None if e1 is None else e1.to(DEV))
    Ho, Wo = ref.shape[1], ref.shape[2]
    sc = ref.abs().mean().item()
    args = dict(B=c["B"], Hs=c["H"], Ws=c["W"], Ho=Ho, Wo=Wo, ks=c["ks"], stride=c["stride"], ups=c["ups"], out_f32=True,
                a1=None if x1 is None else x1.half().to(DEV), **kw)
    # the fp16-operand kernel on the rounded activations: the yardstick
    plain_kw = {k: v for k, v in args.items() if not k.endswith("f")}
    base = G.igemm(x0.half().to(DEV), wp, **plain_kw)
    base_err = (base.cpu().double().view(ref.shape) - ref).abs().max().item() / sc
    lib = N_.load()
    names = [lib.sdmi_gemm_config_name(i).decode() for i in range(lib.sdmi_gemm_num_configs())]
    ran = 0
    for cfg in [-1] + _plain_cfgs():
        for ksplit in (1, 3):
            try:
                out = G.igemm(x0.half().to(DEV), wp, accurate=True, a0f=x0.to(DEV), a1f=None if x1 is None else x1.to(DEV), cfg=cfg,
                              ksplit=ksplit, **args)
            except ValueError as exc:
                assert "wide A operand" in str(exc), exc
                continue
            ran += 1
            err = (out.cpu().double().view(ref.shape) - ref).abs().max().item() / sc
            G.log_metric(test="accurate_gemm", case=str(c), cfg=names[cfg] if cfg >= 0 else "auto", ksplit=ksplit, err=err, fp16_operand_err=base_err)
            assert err < 2e-5 and err < base_err / 20, f"{c} cfg {cfg} ksplit {ksplit}: err {err:.2e} of the output scale (fp16 operands: {base_err:.2e})"
    assert ran >= 6, ran


def _conv_ref_f64(x_nhwc, w_oihw, stride, ups):
    x = x_nhwc.double().permute(0, 3, 1, 2)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    ks = w_oihw.shape[-1]
    return F.conv2d(x, w_oihw.double(), stride=stride, padding=1 if ks == 3 else 0).permute(0, 2, 3, 1)


@pytest.mark.parametrize("X0,X1,H,W", [(64, 0, 16, 16), (128, 64, 12, 12), (192, 128, 8, 8)])
def test_conv_with_fused_skip_segment(X0, X1, H, W):
    """conv_merged 3x3 + residual_layer 1x1 over the block input (two concat sources) in one accumulator
    (sd/diffusion.py:143,194-209): K = 9*C + X0 + X1."""
    B, Cm, Co = 2, 128, 192
    g = torch.Generator().manual_seed(X0 + X1)
    t = torch.randn((B, H, W, Cm), generator=g).half()
    x0 = torch.randn((B, H, W, X0), generator=g).half()
    x1 = torch.randn((B, H, W, X1), generator=g).half() if X1 else None
    w3 = (torch.randn((Co, Cm, 3, 3), generator=g) / math.sqrt(9 * Cm)).half().float()
    ws = (torch.randn((Co, X0 + X1, 1, 1), generator=g) / math.sqrt(X0 + X1)).half().float()
    xs = x0 if x1 is None else torch.cat([x0, x1], -1)
    ref = _conv_ref(t, w3, 1, 0) + _conv_ref(xs, ws, 1, 0)
    wp = torch.cat([G.pack_conv(w3.to(DEV)), G.pack_conv(ws.to(DEV))], 1).contiguous()
    ran = 0
    for cfg in [-1] + _cfgs():
        for ksplit in (1, 2, 5):
            try:
                out = G.igemm(t.to(DEV), wp, B=B, Hs=H, Ws=W, Ho=H, Wo=W, ks=3, out_f32=True, cfg=cfg, ksplit=ksplit,
                              x0=x0.to(DEV), x1=None if x1 is None else x1.to(DEV))
            except ValueError as exc:       # halo-reuse kernels do not take the extra segment
                assert "not applicable" in str(exc) or "LDS" in str(exc), exc
                continue
            ran += 1
            err = (out.cpu().double().view(ref.shape) - ref).abs().max().item()
            assert err < 2e-3, f"cfg {cfg} ksplit {ksplit}: max abs err {err}"
    assert ran >= 3 * len(_plain_cfgs())


def test_gemm_transposed_tail():
    """in_proj epilogue: columns [0,2C) row-major, columns [2C,3C) written as V^T[b][c][s]."""
    B, S, Cc = 2, 192, 128
    g = torch.Generator().manual_seed(5)
    a = torch.randn((B * S, Cc), generator=g).half()
    w = (torch.randn((3 * Cc, Cc), generator=g) / math.sqrt(Cc)).half()
    ref = a.double() @ w.double().t()
    ldt = 256
    for cfg in _plain_cfgs():
        for ksplit in (1, 2):
            perm = (cfg + ksplit) & 1          # natural order (VAE) and the attention kernel's quad-permuted order
            vt = torch.zeros((B * Cc, ldt), dtype=torch.float16, device=DEV)
            out = G.igemm(a.to(DEV).view(1, B * S, 1, Cc), w.to(DEV), B=1, Hs=B * S, Ws=1, Ho=B * S, Wo=1, cfg=cfg,
                          ksplit=ksplit, out_t=vt, nt0=2 * Cc, S=S, ldt=ldt, out_t_perm=perm)
            assert (out.cpu().double() - ref[:, :2 * Cc]).abs().max().item() < 8e-3
            v_ref = ref[:, 2 * Cc:].view(B, S, Cc).permute(0, 2, 1)        # (B, C, S)
            got = (G.vt_natural_order(vt.cpu()) if perm else vt.cpu()).double().view(B, Cc, ldt)
            assert (got[:, :, :S] - v_ref).abs().max().item() < 8e-3, f"cfg {cfg} ksplit {ksplit}"
            assert got[:, :, S:].abs().max().item() == 0.0


def _attn_ref(q, k, v, B, Hh, d, Sq, Skv):
    qh = q.double().view(B, Sq, Hh, d).permute(0, 2, 1, 3)
    kh = k.double().view(B, Skv, Hh, d).permute(0, 2, 1, 3)
    vh = v.double().view(B, Skv, Hh, d).permute(0, 2, 1, 3)
    w = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), dim=-1)
    return (w @ vh).permute(0, 2, 1, 3).reshape(B * Sq, Hh * d)


@pytest.mark.parametrize("d,Sq,Skv", [(40, 256, 256), (40, 1024, 1024), (80, 256, 256), (160, 64, 64),
                                      (160, 256, 256), (40, 200, 77), (80, 64, 77), (160, 144, 144),
                                      (40, 4096, 4096)])
def test_flash_attention(d, Sq, Skv):
    B, Hh = 2, 8
    if Sq == 4096:
        Hh = 2
    Cc = Hh * d
    g = torch.Generator().manual_seed(d + Sq)
    q = torch.randn((B * Sq, Cc), generator=g).half()
    k = torch.randn((B * Skv, Cc), generator=g).half()
    v = torch.randn((B * Skv, Cc), generator=g).half()
    ref = _attn_ref(q, k, v, B, Hh, d, Sq, Skv)
    ldvt = ((Skv + 63) // 64) * 64
    vt = torch.zeros((B * Cc, ldvt), dtype=torch.float16)
    vt.view(B, Cc, ldvt)[:, :, :Skv] = v.view(B, Skv, Cc).permute(0, 2, 1)
    out = G.attention(q.to(DEV), k.to(DEV), vt.to(DEV), B, Hh, d, Sq, Skv)
    err = (out.cpu().double() - ref).abs().max().item()
    G.log_metric(test="attention", d=d, Sq=Sq, Skv=Skv, max_abs_err=err)
    assert err < 2e-3, f"d={d} Sq={Sq} Skv={Skv}: max abs err {err}"       # measured 1.4e-4 .. 8.1e-4


@pytest.mark.parametrize("S,d", [(512, 40), (2048, 40), (4096, 40), (1024, 80), (2048, 160)])
def test_flash_attention_large_logits(S, d):
    """Online-softmax rescale path of BOTH kernel forms -- the key-split form (S <= 1024) and the one-chain form that carries
    the S = 4096 / 9216 self-attention (csrc/attention.hip) -- on data that forces it: single keys dominate their query late in
    the sequence, so the running max of a 32-query block jumps between 64-key tiles, several times and in its last tile."""
    B, Hh = 1, 8
    g = torch.Generator().manual_seed(3 + S)
    q = torch.randn((B * S, Hh * d), generator=g).half()
    k = torch.randn((B * S, Hh * d), generator=g).half()
    v = (0.5 * torch.randn((B * S, Hh * d), generator=g)).half()      # |o| < 2.5: the fp16 output's own half-ulp stays below 1e-3
    amp = 4.0 * math.sqrt(40.0 / d)                 # keeps the spike's logit ~ 4 |q|^2 / sqrt(d) of the d = 40 case
    k[300] = (q[7] * amp).half()                    # query 7 . key 300 >> others: tile 4
    k[40] = (q[100] * 0.75 * amp).half()
    k[S - 3] = (q[S // 2 + 5] * amp).half()         # a jump in the LAST tile of the sequence
    k[S // 2 + 70] = (q[7] * 1.5 * amp).half()      # the same query's max jumps a second time, later
    ref = _attn_ref(q, k, v, B, Hh, d, S, S)
    vt = v.view(B, S, Hh * d).permute(0, 2, 1).contiguous().view(B * Hh * d, S)
    out = G.attention(q.to(DEV), k.to(DEV), vt.to(DEV), B, Hh, d, S, S)
    err = (out.cpu().double() - ref).abs().max().item()
    G.log_metric(test="attention_spike", S=S, d=d, max_abs_err=err)
    assert err < 2.5e-3, f"S={S} d={d}: max abs err {err}"


@pytest.mark.parametrize("C0,C1,P,in_f32,silu,eps", [(320, 0, 64, True, True, 1e-5), (640, 320, 256, False, True, 1e-5),
                                                     (1280, 640, 64, True, True, 1e-5), (320, 0, 4096, True, False, 1e-6),
                                                     (1280, 1280, 64, False, True, 1e-5), (1280, 640, 1024, True, True, 1e-5),
                                                     (640, 0, 1024, False, True, 1e-5), (640, 320, 4096, True, True, 1e-5)])
def test_groupnorm(C0, C1, P, in_f32, silu, eps):
    B = 2
    Hh = int(math.isqrt(P))
    g = torch.Generator().manual_seed(C0 + C1 + P)
    dt = torch.float32 if in_f32 else torch.float16
    x0 = (torch.randn((B, Hh, Hh, C0), generator=g) * 2 + 0.5).to(dt)
    x1 = (torch.randn((B, Hh, Hh, C1), generator=g) * 0.5 - 1).to(dt) if C1 else None
    Cc = C0 + C1
    gamma = 1 + 0.1 * torch.randn((Cc,), generator=g)
    beta = 0.1 * torch.randn((Cc,), generator=g)
    x = x0 if x1 is None else torch.cat([x0, x1], -1)
    ref = F.group_norm(x.double().permute(0, 3, 1, 2), 32, gamma.double(), beta.double(), eps)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    y = G.groupnorm(x0.to(DEV), None if x1 is None else x1.to(DEV), gamma.to(DEV), beta.to(DEV), eps, silu)
    err = (y.cpu().double() - ref).abs().max().item()
    G.log_metric(test="groupnorm", C0=C0, C1=C1, P=P, max_abs_err=err)
    assert err < 4e-3, f"max abs err {err}"        # measured 1.95e-3 = half an fp16 ulp of the largest outputs (|y| ~ 4)


@pytest.mark.parametrize("C,P,ksplit,res,silu", [(1280, 64, 12, None, 1), (1280, 256, 6, "f32", 1), (640, 1024, 3, "f16", 0),
                                                 (640, 256, 5, "f32", 1), (2560, 64, 16, None, 1)])
def test_groupnorm_from_split_k_slabs(C, P, ksplit, res, silu):
    """The split-K combine of a conv and the GroupNorm behind it in one launch (csrc/norm.hip gn_fused_kernel<.., SLAB>;
    sd/diffusion.py:179 -> 199, 205 -> 294): x = sum_z slab[z] + bias (+ residual) must equal splitk_finalize's sum bit for
    bit in its fp32 / fp16 copies, and the normalised output the fp64 GroupNorm of that sum."""
    B = 2
    g = torch.Generator().manual_seed(C + P + ksplit)
    slabs = torch.randn((ksplit, B * P, C), generator=g) * (0.5 + torch.rand((C,), generator=g)) + 0.3 * torch.randn((C,), generator=g)
    bias = torch.randn((C,), generator=g)
    r = None if res is None else torch.randn((B * P, C), generator=g)
    if res == "f16":
        r = r.half()
    gamma = 1 + 0.2 * torch.randn((C,), generator=g)
    beta = 0.2 * torch.randn((C,), generator=g)
    x = torch.zeros((B * P, C))
    for z in range(ksplit):                      # fp32 sums in slab order, as the kernels add them
        x = x + slabs[z]
    x = x + bias
    if r is not None:
        x = x + r.float()
    ref = F.group_norm(x.double().view(B, P, C).permute(0, 2, 1), 32, gamma.double(), beta.double(), 1e-5)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(B * P, C)
    lib = N_.load()
    sd, bd, gd, bed = slabs.to(DEV), bias.to(DEV), gamma.to(DEV), beta.to(DEV)
    rd = None if r is None else r.to(DEV)
    y = torch.full((B * P, C), float("nan"), dtype=torch.float16, device=DEV)
    o32 = torch.full((B * P, C), float("nan"), device=DEV)
    o16 = torch.full((B * P, C), float("nan"), dtype=torch.float16, device=DEV)
    N_.check(lib.sdmi_op_groupnorm_slab(N_.ptr(sd), ksplit, N_.ptr(bd), N_.ptr(rd), int(res == "f32"), C, B, P, N_.ptr(gd), N_.ptr(bed),
                                        1e-5, silu, N_.ptr(y), N_.ptr(o32), N_.ptr(o16), N_.cur_stream()), "groupnorm_slab")
    torch.cuda.synchronize()
    assert torch.equal(o32.cpu(), x), "combined tensor differs from the slab-order fp32 sum"
    assert torch.equal(o16.cpu(), x.half())
    err = (y.float().cpu().double() - ref).abs().max().item()
    G.log_metric(test="gn_slab", C=C, P=P, ksplit=ksplit, max_abs_err=err)
    assert err < 4e-3, f"max abs err {err}"


@pytest.mark.parametrize("P,offset", [(4096, 30.0), (1024, -30.0), (4096, 100.0), (64, 30.0)])
def test_groupnorm_large_mean(P, offset):
    """|mean| >= 30 sigma in every group: the two-launch path computes the variance as E[x^2] - E[x]^2 in fp32 from
    fixed-order partial sums (norm.hip gn_apply_kernel), whose relative error grows like eps_fp32 * (mean/sigma)^2
    = 5e-5 at 30 sigma, 6e-4 at 100 sigma -- both far below the fp16 rounding of the output.  (P = 64 takes the
    single-launch kernel, which is exact two-pass.)  Inputs are the fp32 stream, as in the UNet."""
    B, C0 = 2, 320
    Hh = int(math.isqrt(P))
    g = torch.Generator().manual_seed(P)
    x0 = torch.randn((B, Hh, Hh, C0), generator=g) + offset
    gamma = 1 + 0.1 * torch.randn((C0,), generator=g)
    beta = 0.1 * torch.randn((C0,), generator=g)
    ref = F.silu(F.group_norm(x0.double().permute(0, 3, 1, 2), 32, gamma.double(), beta.double(), 1e-5)).permute(0, 2, 3, 1)
    y = G.groupnorm(x0.to(DEV), None, gamma.to(DEV), beta.to(DEV), 1e-5, True)
    err = (y.cpu().double() - ref).abs().max().item()
    G.log_metric(test="groupnorm_large_mean", P=P, offset=offset, max_abs_err=err)
    assert err < (4e-3 if abs(offset) <= 30 else 8e-3), f"max abs err {err}"


@pytest.mark.parametrize("M,Cc,in_f32", [(128, 320, True), (300, 640, False), (64, 1280, True)])
def test_layernorm(M, Cc, in_f32):
    g = torch.Generator().manual_seed(M + Cc)
    x = (torch.randn((M, Cc), generator=g) * 3 + 1).to(torch.float32 if in_f32 else torch.float16)
    gamma = 1 + 0.1 * torch.randn((Cc,), generator=g)
    beta = 0.1 * torch.randn((Cc,), generator=g)
    ref = F.layer_norm(x.double(), (Cc,), gamma.double(), beta.double(), 1e-5)
    y = G.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV))
    err = (y.cpu().double() - ref).abs().max().item()
    assert err < 6e-3, f"max abs err {err}"


def test_cfg_ddpm_step_bit_exact_vs_oracle():
    """Fused CFG + DDPM step kernel vs the oracle's restatement of sd/pipeline.py:230-233 + sd/ddpm.py:102-139
    on identical eps: bit-exact for every step of the 20- and 50-step schedules probed."""
    from oracle import ddpm_ref
    from pytorch_stable_diffusion_amd import _native as N
    from pytorch_stable_diffusion_amd.ddpm import DDPMSampler
    lat = H.seeded((1, 4, 16, 16), 21)
    eps = H.seeded((2, 4, 16, 16), 22)
    for n in (20, 50):
        ref = ddpm_ref.RefSchedule()
        ref.set_inference_timesteps(n)
        smp = DDPMSampler(torch.Generator().manual_seed(0))
        smp.set_inference_timesteps(n)
        ts = ref.timesteps.tolist()
        for t in (ts[0], ts[1], ts[n // 2], ts[-2], ts[-1]):
            noise = H.seeded((1, 4, 16, 16), 23 + t) if t > 0 else None
            cond, uncond = eps.chunk(2)
            guided = 7.5 * (cond - uncond) + uncond
            want = ref.step(t, lat.clone(), guided, noise)
            got = lat.clone().to(DEV)
            N.cfg_ddpm_step(eps.to(DEV), True, 7.5, got, None if noise is None else noise.to(DEV),
                            smp.step_coefficients(t))
            torch.cuda.synchronize()
            assert torch.equal(got.cpu(), want), f"n={n} t={t}: max diff {(got.cpu()-want).abs().max()}"


@pytest.mark.parametrize("M,Cc,Nn", [(256, 320, 960), (200, 640, 640), (128, 1280, 5120)])
def test_layernorm_folded_into_gemm(M, Cc, Nn):
    """LayerNorm -> Linear as ONE GEMM on the raw tensor (sd/diffusion.py:317-321,334-339,351-356): the producer
    GEMM's epilogue emits per-row statistics, the consumer applies rstd*(x W'^T - mean*g) + h.  Checked against
    fp64 LayerNorm + Linear, for every plain tile config on both sides, incl. a stream with a large row mean."""
    g = torch.Generator().manual_seed(M + Cc + Nn)
    a = torch.randn((M, Cc), generator=g).half()                              # producer A
    wp = (torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)).half()          # producer weights
    res = (torch.randn((M, Cc), generator=g) * 2 + 3.0)                       # residual with a mean offset (fp32 stream)
    gamma = 1 + 0.1 * torch.randn((Cc,), generator=g)
    beta = 0.1 * torch.randn((Cc,), generator=g)
    w = torch.randn((Nn, Cc), generator=g) / math.sqrt(Cc)
    bias = torch.randn((Nn,), generator=g)
    x_ref = a.double() @ wp.double().t() + res.double()                       # the stream the LayerNorm sees
    ref = F.layer_norm(x_ref, (Cc,), gamma.double(), beta.double(), 1e-5) @ w.double().t() + bias.double()
    wf, gf, hf = G.ln_fold_prep(w.to(DEV), gamma.to(DEV), beta.to(DEV), bias.to(DEV))
    worst = 0.0
    cfgs = _plain_cfgs()
    for i, pc in enumerate(cfgs):
        bn = G.gemm_tile(pc)[1]
        ntn = (Cc + bn - 1) // bn
        rowstat = torch.full((M, ntn, 2), float("nan"), device=DEV)
        x32, x16 = G.igemm(a.to(DEV).view(1, M, 1, Cc), wp.to(DEV), B=1, Hs=M, Ws=1, Ho=M, Wo=1, res=res.to(DEV), out_f32=True,
                           cfg=pc, want16=True, rowstat=rowstat)
        # emitted statistics = sums of the fp16 shadow, per n-tile
        x16f = x16.float().cpu()
        got = rowstat.cpu()
        for t in range(ntn):
            blk = x16f[:, t * bn:(t + 1) * bn].double()
            assert (got[:, t, 0].double() - blk.sum(1)).abs().max().item() < 2e-2
            assert (got[:, t, 1].double() - (blk * blk).sum(1)).abs().max().item() < 2e-1
        cc = cfgs[(i * 7 + 3) % len(cfgs)]                                    # consumer tile independent of the producer's
        out = G.igemm(x16.view(1, M, 1, Cc), wf, B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=hf, out_f32=True, cfg=cc,
                      ln_stat=rowstat, ln_g=gf, ln_c=Cc)
        err = (out.cpu().double() - ref).abs().max().item()
        worst = max(worst, err)
        assert err < 1.5e-2, f"producer cfg {pc} consumer cfg {cc}: max abs err {err}"
    # same through the separate LayerNorm kernel + plain GEMM: the folded path must be about as accurate
    u = G.layernorm(x32, gamma.to(DEV), beta.to(DEV))
    out2 = G.igemm(u.view(1, M, 1, Cc), w.half().to(DEV), B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=bias.to(DEV), out_f32=True)
    err2 = (out2.cpu().double() - ref).abs().max().item()
    G.log_metric(test="ln_fold", M=M, C=Cc, N=Nn, folded_max_abs=worst, unfused_max_abs=err2)
    assert worst < 3 * err2 + 2e-3


def test_layernorm_fold_transposed_tail():
    """in_proj with the V^T tail, LayerNorm folded (statistics given per row, one n-tile)."""
    B, S, Cc = 2, 192, 128
    g = torch.Generator().manual_seed(11)
    x = (torch.randn((B * S, Cc), generator=g) * 1.5 + 0.7).half()
    gamma = 1 + 0.1 * torch.randn((Cc,), generator=g)
    beta = 0.1 * torch.randn((Cc,), generator=g)
    w = torch.randn((3 * Cc, Cc), generator=g) / math.sqrt(Cc)
    ref = F.layer_norm(x.double(), (Cc,), gamma.double(), beta.double(), 1e-5) @ w.double().t()
    wf, gf, hf = G.ln_fold_prep(w.to(DEV), gamma.to(DEV), beta.to(DEV), None)
    xs = x.float()
    stat = torch.stack([xs.sum(1), (xs * xs).sum(1)], 1).view(B * S, 1, 2).contiguous().to(DEV)
    ldt = 256
    for cfg in _plain_cfgs():
        vt = torch.zeros((B * Cc, ldt), dtype=torch.float16, device=DEV)
        out = G.igemm(x.to(DEV).view(1, B * S, 1, Cc), wf, B=1, Hs=B * S, Ws=1, Ho=B * S, Wo=1, cfg=cfg, bias=hf,
                      out_t=vt, nt0=2 * Cc, S=S, ldt=ldt, ln_stat=stat, ln_g=gf, ln_c=Cc, out_t_perm=1)
        assert (out.cpu().double() - ref[:, :2 * Cc]).abs().max().item() < 1.2e-2, f"cfg {cfg}"
        v_ref = ref[:, 2 * Cc:].view(B, S, Cc).permute(0, 2, 1)
        got = G.vt_natural_order(vt.cpu()).double().view(B, Cc, ldt)
        assert (got[:, :, :S] - v_ref).abs().max().item() < 1.2e-2, f"cfg {cfg} (V^T)"


@pytest.mark.parametrize("S,Cc", [(64, 640), (256, 1280)])
def test_folded_cross_attention_two_gemms(S, Cc):
    """Cross-attention with q_proj / out_proj multiplied into the per-prompt K / V (sd/attention.py:219-256
    regrouped): GEMM 1 = raw stream x (layernorm_2-folded K_h Wq_h) with the per-head softmax over 77 keys in the
    epilogue and per-image weights, GEMM 2 = probabilities x (Wo_h V_h) + bias + residual.  Checked against the
    reference order of operations in fp64 for every 128-wide tile that divides the map."""
    Bn, Hh, T = 2, 8, 77
    d = Cc // Hh
    M = Bn * S
    g = torch.Generator().manual_seed(S + Cc)
    x = torch.randn((M, Cc), generator=g) * 1.5 + 0.5
    x16 = x.half()
    gamma = 1 + 0.1 * torch.randn((Cc,), generator=g)
    beta = 0.1 * torch.randn((Cc,), generator=g)
    wq = torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)
    wo = (torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)).half()
    bo = torch.randn((Cc,), generator=g)
    k = (torch.randn((Bn, T, Cc), generator=g) * 1.5).half()
    v = torch.randn((Bn, T, Cc), generator=g).half()
    # reference: LayerNorm -> q_proj -> per-head softmax(q k^T / sqrt(d)) v -> out_proj + bias + residual
    xn = F.layer_norm(x16.double(), (Cc,), gamma.double(), beta.double(), 1e-5)
    q = (xn @ wq.double().t()).view(Bn, S, Hh, d).transpose(1, 2)
    kh = k.double().view(Bn, T, Hh, d).transpose(1, 2)
    vh = v.double().view(Bn, T, Hh, d).transpose(1, 2)
    p_ref = torch.softmax(q @ kh.transpose(-1, -2) / math.sqrt(d), dim=-1)          # (B, H, S, T)
    o = (p_ref @ vh).transpose(1, 2).reshape(M, Cc)
    ref = o @ wo.double().t() + bo.double() + x.double()
    # folded operands (what Engine::xattn_fold builds once per prompt)
    qs = math.log2(math.e) / math.sqrt(d)
    w1 = torch.zeros((Bn, Hh, 128, Cc), dtype=torch.float64)
    w2 = torch.zeros((Cc, Bn, Hh, 128), dtype=torch.float64)
    for h in range(Hh):
        sl = slice(h * d, (h + 1) * d)
        w1[:, h, :T] = qs * (k.double()[:, :, sl] @ wq.double()[sl])                 # (B, T, C)
        w2[:, :, h, :T] = torch.einsum("cd,btd->cbt", wo.double()[:, sl], v.double()[:, :, sl])
    w1f, g1, h1 = G.ln_fold_prep(w1.view(Bn * 1024, Cc).float().to(DEV), gamma.to(DEV), beta.to(DEV), None)
    w2h = w2.view(Cc, Bn * 1024).half().to(DEV)
    x16d = x16.to(DEV)
    xf = x16.float()
    stat = torch.stack([xf.sum(1), (xf * xf).sum(1)], dim=1).view(M, 1, 2).to(DEV)
    lib = N_.load()
    names = [lib.sdmi_gemm_config_name(i).decode() for i in range(lib.sdmi_gemm_num_configs())]
    n_run = 0
    for pc in _plain_cfgs():
        bm, bn = G.gemm_tile(pc)
        if bn != 128 or S % bm:
            continue
        pr = G.igemm(x16d.view(1, M, 1, Cc), w1f, B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=h1, cfg=pc, ln_stat=stat, ln_g=g1, ln_c=Cc,
                     act=2, sm_valid=T, img_rows=S, w_img_stride=1024 * Cc, vec_img_stride=1024, n_out=1024)
        got_p = pr.float().cpu().view(Bn, S, Hh, 128).permute(0, 2, 1, 3)
        assert got_p[..., T:].abs().max().item() == 0.0, names[pc]
        perr = (got_p[..., :T].double() - p_ref).abs().max().item()
        assert perr < 4e-3, f"{names[pc]}: probabilities max abs err {perr}"
        for pc2 in (pc, _plain_cfgs()[(n_run * 5 + 7) % len(_plain_cfgs())]):
            bm2 = G.gemm_tile(pc2)[0]
            if S % bm2:
                continue
            out = G.igemm(pr.view(1, M, 1, 1024), w2h, B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=bo.to(DEV), res=x.to(DEV), out_f32=True,
                          cfg=pc2, img_rows=S, w_img_stride=1024, ldw=Bn * 1024, n_out=Cc)
            err = (out.cpu().double() - ref).abs().max().item()
            rel = ((out.cpu().double() - ref).norm() / ref.norm()).item()
            assert err < 2e-2 and rel < 1e-3, f"{names[pc]} -> {names[pc2]}: max abs {err}, rel L2 {rel}"
        n_run += 1
    assert n_run >= 3
    G.log_metric(test="xattn_fold", S=S, C=Cc, tiles=n_run)


@pytest.mark.parametrize("Bn,Hs,Ws,Cin,Cout,ksplit", [(2, 16, 16, 128, 192, 1), (1, 8, 24, 64, 64, 1), (2, 32, 32, 64, 128, 2)])
def test_upsample_conv_as_four_phase_convs(Bn, Hs, Ws, Cin, Cout, ksplit):
    """Upsample (sd/diffusion.py:426-435: nearest x2, then 3x3 conv) as four 2x2 convs on the source grid with pre-summed taps:
    must match the 9-tap conv of the materialised upsampled tensor (fp64) and the library's own 9-tap upsample path."""
    g = torch.Generator().manual_seed(Hs * Ws + Cin)
    x = torch.randn((Bn, Hs, Ws, Cin), generator=g).half()
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn((Cout,), generator=g)
    up = F.interpolate(x.double().permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    ref = F.conv2d(up, w.double(), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    w4 = G.pack_ups_phase(w.to(DEV))
    nine = G.igemm(x.to(DEV), G.pack_conv(w.to(DEV)), B=Bn, Hs=Hs, Ws=Ws, Ho=2 * Hs, Wo=2 * Ws, ks=3, ups=1, bias=bias.to(DEV), out_f32=True)
    e9 = (nine.cpu().double() - ref).abs().max().item()
    rows = Bn * Hs * Ws
    n_run = 0
    for pc in _plain_cfgs():
        bm = G.gemm_tile(pc)[0]
        if rows % bm:
            continue
        got = G.igemm(x.to(DEV), w4, B=Bn, Hs=Hs, Ws=Ws, Ho=Hs, Wo=Ws, ks=2, bias=bias.to(DEV), out_f32=True, cfg=pc, ksplit=ksplit,
                      phase2=1, img_rows=rows, w_img_stride=Cout * 4 * Cin, n_out=Cout)
        err = (got.cpu().double() - ref).abs().max().item()
        assert err < max(2 * e9, 4e-3), f"cfg {pc}: max abs err {err} (9-tap path: {e9})"
        n_run += 1
    assert n_run >= 8
    G.log_metric(test="ups_phase", rows=rows, C=Cin, nine_tap_err=e9)


@pytest.mark.parametrize("M,partial,stream_f32,bm", [(128, 0, True, 64), (256, 1, True, 64), (192, 1, False, 64), (64, 0, False, 64),
                                                     (96, 0, True, 32), (160, 1, True, 32), (64, 1, False, 32), (32, 0, False, 32)])
def test_back_to_back_gemm(M, partial, stream_f32, bm):
    """csrc/b2b.hip: out_proj + residual, then LayerNorm -> Linear (q_proj, or the composed feed-forward over [LN(s) | s])
    in one launch, against the same chain in fp64 (sd/diffusion.py:325-363 for C = 320)."""
    import ctypes as C
    Cc = 320
    g = torch.Generator().manual_seed(M + partial)
    a1 = torch.randn((M, Cc), generator=g).half()
    w1 = (torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)).half()
    b1 = torch.randn((Cc,), generator=g)
    r1 = torch.randn((M, Cc), generator=g) * 2 + 1.0
    r2 = torch.randn((M, Cc), generator=g)
    if not stream_f32:
        r1, r2 = r1.half().float(), r2.half().float()
    gamma = 1 + 0.1 * torch.randn((Cc,), generator=g)
    beta = 0.1 * torch.randn((Cc,), generator=g)
    w2 = torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)
    b2 = torch.randn((Cc,), generator=g)
    wp = (torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)).half()
    s_ref = a1.double() @ w1.double().t() + b1.double() + r1.double()
    ln = F.layer_norm(s_ref, (Cc,), gamma.double(), beta.double(), 1e-5)
    if partial:
        ref = ln @ w2.double().t() + s_ref @ wp.double().t() + b2.double() + r2.double()
    else:
        ref = 0.25 * (ln @ w2.double().t() + b2.double())
    wf, _, hf = G.ln_fold_prep(w2.to(DEV), gamma.to(DEV), beta.to(DEV), b2.to(DEV))
    if partial:
        wf = torch.cat([wf, wp.to(DEV)], dim=1).contiguous()
    rdt = torch.float32 if stream_f32 else torch.float16
    a1d, w1d, b1d, r1d, r2d = a1.to(DEV), w1.to(DEV), b1.to(DEV), r1.to(DEV).to(rdt), r2.to(DEV).to(rdt)
    s32 = torch.full((M, Cc), float("nan"), device=DEV)
    s16 = torch.full((M, Cc), float("nan"), dtype=torch.float16, device=DEV)
    out = torch.full((M, Cc), float("nan"), device=DEV)
    out16 = torch.full((M, Cc), float("nan"), dtype=torch.float16, device=DEV)
    d = N_.B2bDesc()
    d.a1, d.lda1, d.w1, d.b1 = a1d.data_ptr(), Cc, w1d.data_ptr(), b1d.data_ptr()
    d.r1, d.r1_f32 = r1d.data_ptr(), int(stream_f32)
    d.s32, d.s16 = (s32.data_ptr() if stream_f32 else 0), s16.data_ptr()
    d.w2, d.K2, d.h2, d.partial, d.cscale = wf.data_ptr(), (640 if partial else 320), hf.data_ptr(), partial, (0.0 if partial else 0.25)
    if partial:
        d.r2, d.r2_f32 = r2d.data_ptr(), int(stream_f32)
    if partial and stream_f32:
        d.out, d.out_f32, d.out16 = out.data_ptr(), 1, out16.data_ptr()
    else:
        d.out, d.out_f32 = out16.data_ptr(), 0
    d.M, d.eps, d.bm = M, 1e-5, bm
    N_.check(N_.load().sdmi_op_b2b(C.byref(d), 1, None, N_.cur_stream()), "b2b")
    torch.cuda.synchronize()
    es = (s16.float().cpu().double() - s_ref).abs().max().item()
    assert es < 1.5e-2, f"S (fp16 copy): max abs err {es}"
    if stream_f32:
        assert (s32.cpu().double() - s_ref).abs().max().item() < 4e-3
    got = (out if (partial and stream_f32) else out16.float()).cpu().double()
    err = (got - ref).abs().max().item()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert err < 2.5e-2 and rel < 1.5e-3, f"max abs {err}, rel L2 {rel}"
    if partial and stream_f32:
        assert torch.equal(out16.cpu(), out.cpu().half())
    G.log_metric(test="b2b", M=M, partial=partial, stream_f32=stream_f32, rel_l2=rel, max_abs=err)


@pytest.mark.parametrize("Bn,S,bm,gn", [(2, 64, 64, 0), (2, 96, 32, 0), (1, 256, 32, 0), (3, 32, 32, 0),
                                        (2, 64, 64, 1), (2, 1024, 32, 1), (1, 96, 32, 2), (2, 256, 64, 2)])
def test_back_to_back_gemm_qkv(Bn, S, bm, gn):
    """b2b three-pass form: conv_input (1x1) then layernorm_1 + in_proj in one launch (sd/diffusion.py:312-321,
    sd/attention.py:42-52): q (pre-scaled) | k row-major, v transposed in the attention kernel's key order."""
    import ctypes as C
    Cc, M = 320, Bn * S
    g = torch.Generator().manual_seed(S + Bn)
    a1 = torch.randn((M, Cc), generator=g).half()
    w1 = (torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)).half()
    b1 = torch.randn((Cc,), generator=g)
    gamma = 1 + 0.1 * torch.randn((Cc,), generator=g)
    beta = 0.1 * torch.randn((Cc,), generator=g)
    w2 = torch.randn((3 * Cc, Cc), generator=g) / math.sqrt(Cc)
    gx = None
    if gn:
        # gn = 1 / 2: the first product's A operand is GroupNorm(32) of a raw fp32 / fp16 tensor with per-channel offsets
        # (sd/diffusion.py:294,312), normalised inside the kernel from sdmi_op_gn_stats' partials
        gx = torch.randn((M, Cc), generator=g) * (0.5 + torch.rand((Cc,), generator=g)) + 3.0 * torch.randn((Cc,), generator=g)
        if gn == 2:
            gx = gx.half().float()
        ggam = 1 + 0.2 * torch.randn((Cc,), generator=g)
        gbet = 0.2 * torch.randn((Cc,), generator=g)
        xn = F.group_norm(gx.double().view(Bn, S, Cc).permute(0, 2, 1), 32, ggam.double(), gbet.double(), 1e-6)
        a1 = xn.permute(0, 2, 1).reshape(M, Cc).half()       # the kernel rounds the normalised operand to fp16 as well
    s_ref = a1.double() @ w1.double().t() + b1.double()
    qkv = F.layer_norm(s_ref, (Cc,), gamma.double(), beta.double(), 1e-5) @ w2.double().t()
    wf, _, hf = G.ln_fold_prep(w2.to(DEV), gamma.to(DEV), beta.to(DEV), None)
    Spad = (S + 63) // 64 * 64
    a1d, w1d, b1d = a1.to(DEV), w1.to(DEV), b1.to(DEV)
    s32 = torch.full((M, Cc), float("nan"), device=DEV)
    s16 = torch.full((M, Cc), float("nan"), dtype=torch.float16, device=DEV)
    qk = torch.full((M, 2 * Cc), float("nan"), dtype=torch.float16, device=DEV)
    vt = torch.zeros((Bn * Cc, Spad), dtype=torch.float16, device=DEV)
    d = N_.B2bDesc()
    d.a1, d.lda1, d.w1, d.b1 = a1d.data_ptr(), Cc, w1d.data_ptr(), b1d.data_ptr()
    d.s32, d.s16 = s32.data_ptr(), s16.data_ptr()
    d.w2, d.K2, d.h2, d.partial, d.cscale = wf.data_ptr(), 320, hf.data_ptr(), 0, 0.5
    d.out, d.out_f32, d.ldo, d.npass2 = qk.data_ptr(), 0, 2 * Cc, 3
    d.vt, d.S, d.ldt = vt.data_ptr(), S, Spad
    d.M, d.eps, d.bm = M, 1e-5, bm
    if gn:
        lib = N_.load()
        gxd = gx.to(DEV) if gn == 1 else gx.half().to(DEV)
        nch = lib.sdmi_gn_num_chunks(S)
        part = torch.zeros((Bn, nch, 32, 2), device=DEV)
        N_.check(lib.sdmi_op_gn_stats(N_.ptr(gxd), None, int(gn == 1), Cc, 0, Bn, S, N_.ptr(part), N_.cur_stream()), "gn_stats")
        gg, gb = ggam.to(DEV), gbet.to(DEV)
        d.a1 = 0
        d.gx, d.gx_f32, d.gn_partial, d.gn_nchunk = gxd.data_ptr(), int(gn == 1), part.data_ptr(), nch
        d.gn_gamma, d.gn_beta, d.gn_eps = gg.data_ptr(), gb.data_ptr(), 1e-6
    N_.check(N_.load().sdmi_op_b2b(C.byref(d), 1, None, N_.cur_stream()), "b2b qkv")
    torch.cuda.synchronize()
    assert (s32.cpu().double() - s_ref).abs().max().item() < (4e-3 if not gn else 2.5e-2)    # gn: an fp16 ulp of the A operand flips
    got = qk.float().cpu().double()
    eq = (got[:, :Cc] - 0.5 * qkv[:, :Cc]).abs().max().item()
    ek = (got[:, Cc:] - qkv[:, Cc:2 * Cc]).abs().max().item()
    vnat = G.vt_natural_order(vt)[:, :S].float().cpu().double().view(Bn, Cc, S).permute(0, 2, 1).reshape(M, Cc)
    ev = (vnat - qkv[:, 2 * Cc:]).abs().max().item()
    lim = 1.0 if not gn else 2.0
    assert eq < 1.5e-2 * lim and ek < 2e-2 * lim and ev < 2e-2 * lim, (eq, ek, ev)
    if Spad != S:
        assert G.vt_natural_order(vt)[:, S:].abs().max().item() == 0.0      # key padding untouched
    G.log_metric(test="b2b_qkv", B=Bn, S=S, bm=bm, q_err=eq, k_err=ek, v_err=ev)


@pytest.mark.parametrize("M,Cc,Nn", [(128, 1280, 1280), (512, 640, 640), (64, 320, 320)])
def test_partial_layernorm_fold_with_split_k(M, Cc, Nn):
    """The composed feed-forward GEMM, y = LN(s) Wf^T + s Wo^T + b + x over K = 2C (sd/diffusion.py:351-381):
    one pass (accumulators rescaled in registers at the fold boundary) and split-K with the slices ending on that boundary
    (the rescale moves to splitk_finalize, row statistics handed over through ln_out) against fp64."""
    g = torch.Generator().manual_seed(M + Cc)
    s32 = torch.randn((M, Cc), generator=g) * 1.5 + 0.7
    s16 = s32.half()
    res = torch.randn((M, Nn), generator=g)
    gamma = 1 + 0.1 * torch.randn((Cc,), generator=g)
    beta = 0.1 * torch.randn((Cc,), generator=g)
    wf = torch.randn((Nn, Cc), generator=g) / math.sqrt(Cc)
    wo = (torch.randn((Nn, Cc), generator=g) / math.sqrt(Cc)).half()
    bias = torch.randn((Nn,), generator=g)
    ref = (F.layer_norm(s16.double(), (Cc,), gamma.double(), beta.double(), 1e-5) @ wf.double().t() + s16.double() @ wo.double().t()
           + bias.double() + res.double())
    wff, gf, hf = G.ln_fold_prep(wf.to(DEV), gamma.to(DEV), beta.to(DEV), bias.to(DEV))
    w = torch.cat([wff, wo.to(DEV)], dim=1).contiguous()
    xf = s16.float()
    stat = torch.stack([xf.sum(1), (xf * xf).sum(1)], dim=1).view(M, 1, 2).to(DEV)
    s16d = s16.to(DEV).view(1, M, 1, Cc)
    outs = {}
    splits = [k for k in (1, 2, 4, 10) if (Cc // 64) % max(k // 2, 1) == 0 and (2 * Cc // 64) % k == 0]
    for ksplit in splits:
        ln_out = torch.full((M, 2), float("nan"), device=DEV)
        out = G.igemm(s16d, w, B=1, Hs=M, Ws=1, Ho=M, Wo=1, a1=s16d, bias=hf, res=res.to(DEV), out_f32=True, ksplit=ksplit,
                      ln_stat=stat, ln_g=gf, ln_c=Cc, ln_ksteps=Cc // 64, ln_out=ln_out)
        err = (out.cpu().double() - ref).abs().max().item()
        assert err < 1.5e-2, f"ksplit {ksplit}: max abs err {err}"
        outs[ksplit] = out
        if ksplit > 1:
            mean = xf.mean(1)
            assert (ln_out[:, 0].cpu() - mean).abs().max().item() < 1e-4
    assert len(splits) >= 2
    for k in splits[1:]:
        assert (outs[k] - outs[1]).abs().max().item() < 2e-3, k


def _atom_moments(x, B, atom=10):
    """x [B*P][C] float64 -> (sum, sumsq) per (image, atom): what a producer's epilogue leaves behind"""
    P, Cc = x.shape[0] // B, x.shape[1]
    xa = x.view(B, P, Cc // atom, atom)
    return xa.sum(dim=(1, 3)), (xa * xa).sum(dim=(1, 3))


@pytest.mark.parametrize("B,P,Nn,K,ksplit,out_f32", [(2, 256, 320, 320, 1, True), (2, 1024, 640, 640, 1, False), (2, 256, 1280, 2560, 4, True),
                                                     (1, 512, 640, 1280, 2, False), (2, 4096, 320, 320, 1, True)])
def test_gemm_epilogue_groupnorm_statistics(B, P, Nn, K, ksplit, out_f32):
    """GroupNorm statistics from the producer (include/sdmi.h sdmi_gemm_desc::gacc; sd/diffusion.py:173,199,294): every tile config that
    can take them (one-pass epilogue) and the split-K combine leave, per image, row block and 10-channel atom, the moments of
    exactly the values they store; every record slot is written (none stays NaN), and two launches give the same bits."""
    from pytorch_stable_diffusion_amd import _native as N
    lib = N.load()
    M = B * P
    g = torch.Generator().manual_seed(P + Nn + K)
    a = torch.randn((M, K), generator=g).half()
    w = (torch.randn((Nn, K), generator=g) / math.sqrt(K)).half()
    bias = torch.randn((Nn,), generator=g) * 3          # atoms with a large mean
    res = torch.randn((M, Nn), generator=g)
    names = [lib.sdmi_gemm_config_name(i).decode() for i in range(lib.sdmi_gemm_num_configs())]
    tried = 0
    for cfg in _plain_cfgs():
        bm, bn = G.gemm_tile(cfg)
        if P % bm != 0 or bn % 64 != 0:
            continue
        kw = dict(B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=bias.to(DEV), res=res.to(DEV), out_f32=out_f32, cfg=cfg, ksplit=ksplit, gstat_rows_img=P)
        try:
            out = G.igemm(a.to(DEV).view(1, M, 1, K), w.to(DEV), **kw)
        except ValueError as e:                            # SDMI_EINVAL: this tile cannot keep a thread's columns fixed
            assert "cannot" in str(e) and "statistics" in str(e), e
            continue
        tried += 1
        rec, T, parts = G.LAST_STAT
        assert not torch.isnan(rec).any(), f"cfg {names[cfg]}: {int(torch.isnan(rec).sum())} record slots were never written"
        assert parts == (1 if ksplit > 1 else 2) and rec.shape[0] == B
        s1, s2 = G.stat_moments(rec)
        r1, r2 = _atom_moments(out.cpu().double(), B)
        e1 = ((s1 - r1).abs() / (r1.abs() + P * 1.0)).max().item()
        e2 = ((s2 - r2).abs() / r2).max().item()
        assert e1 < 2e-6 and e2 < 2e-6, f"cfg {names[cfg]} split {ksplit}: sum err {e1}, sumsq err {e2}"
        G.igemm(a.to(DEV).view(1, M, 1, K), w.to(DEV), **kw)
        assert torch.equal(rec, G.LAST_STAT[0]), f"cfg {names[cfg]}: records differ between two launches"
        if ksplit > 1 and tried >= 3:
            break
    assert tried >= (1 if ksplit > 1 else 8), tried


@pytest.mark.parametrize("B,Hh,Cin,Co,out_f32", [(2, 16, 128, 320, True), (2, 16, 64, 160, False), (1, 32, 64, 640, True)])
def test_conv_160_wide_tile_statistics(B, Hh, Cin, Co, out_f32):
    """The 160-wide conv tiles (N = 320 / 640 / 1280 divide by 160; the plans of the batched multi-prompt mode) leave the GroupNorm
    statistics of their output too (csrc/gemm.hip store_tile GACC160): column sums taken from the tile the epilogue rewrote, whole
    10-channel atoms per tile.  Records against the moments of exactly the stored values; two launches give the same bits."""
    lib = N_.load()
    names = [lib.sdmi_gemm_config_name(i).decode() for i in range(lib.sdmi_gemm_num_configs())]
    g = torch.Generator().manual_seed(B * 100 + Co)
    x = torch.randn((B, Hh, Hh, Cin), generator=g).half()
    w = (torch.randn((Co, Cin, 3, 3), generator=g) / math.sqrt(9 * Cin)).half().float()
    bias = torch.randn((Co,), generator=g) * 2
    wp = G.pack_conv(w.to(DEV))
    P = Hh * Hh
    ran = 0
    for cfg in [i for i, nm in enumerate(names) if "x160" in nm]:
        kw = dict(B=B, Hs=Hh, Ws=Hh, Ho=Hh, Wo=Hh, ks=3, bias=bias.to(DEV), out_f32=out_f32, cfg=cfg, gstat_rows_img=P)
        try:
            out = G.igemm(x.to(DEV), wp, **kw)
        except ValueError as exc:
            assert "not applicable" in str(exc) or "cannot" in str(exc) or "LDS" in str(exc), exc
            continue
        ran += 1
        rec, T, parts = G.LAST_STAT
        assert parts == 2 and not torch.isnan(rec).any(), f"{names[cfg]}: record slots never written"
        s1, s2 = G.stat_moments(rec)
        r1, r2 = _atom_moments(out.cpu().double().view(B * P, Co), B)
        e1 = ((s1 - r1).abs() / (r1.abs() + P * 1.0)).max().item()
        e2 = ((s2 - r2).abs() / r2).max().item()
        assert e1 < 2e-6 and e2 < 2e-6, f"{names[cfg]}: sum err {e1}, sumsq err {e2}"
        assert bool((rec[..., 1, :] == 0).all())                  # whole atoms per tile: the second part stays zero
        out2 = G.igemm(x.to(DEV), wp, **kw)
        assert torch.equal(out, out2) and torch.equal(rec, G.LAST_STAT[0])
    assert ran >= 2, ran


def test_statistics_follow_the_effective_split_k():
    """A split-K request the launcher clamps away (K = 64 is ONE K-step: ksplit 2 runs as 1) must not leave the one-pass
    epilogue writing parts = 2 records into a table laid out for the combine's parts = 1 (past its end from the second image on):
    sdmi_op_gemm_stat_layout reports -- and the launch writes -- the layout of the factor that really runs."""
    B, P, Nn, K = 3, 128, 320, 64
    M = B * P
    g = torch.Generator().manual_seed(77)
    a = torch.randn((M, K), generator=g).half()
    w = (torch.randn((Nn, K), generator=g) / math.sqrt(K)).half()
    bias = torch.randn((Nn,), generator=g)
    ran = 0
    for cfg in _plain_cfgs():
        bm, bn = G.gemm_tile(cfg)
        if P % bm != 0 or bn % 64 != 0:
            continue
        kw = dict(B=1, Hs=M, Ws=1, Ho=M, Wo=1, bias=bias.to(DEV), out_f32=True, cfg=cfg, gstat_rows_img=P)
        try:
            out1 = G.igemm(a.to(DEV).view(1, M, 1, K), w.to(DEV), ksplit=1, **kw)
        except ValueError:
            continue
        rec1, T1, parts1 = G.LAST_STAT
        guard = torch.full((4096,), 7.0, device=DEV)                      # memory right behind where a parts = 1 table would end
        out2 = G.igemm(a.to(DEV).view(1, M, 1, K), w.to(DEV), ksplit=2, **kw)
        rec2, T2, parts2 = G.LAST_STAT
        assert (T2, parts2) == (T1, parts1) and parts2 == 2, (T1, parts1, T2, parts2)
        assert torch.equal(out1, out2) and torch.equal(rec1, rec2) and not torch.isnan(rec2).any()
        assert bool((guard == 7.0).all())
        ran += 1
        if ran >= 4:
            break
    assert ran >= 2


@pytest.mark.parametrize("C0,C1,P,in_f32,silu,offset", [(320, 0, 4096, True, True, 0.0), (640, 320, 1024, True, True, 0.0),
                                                        (1280, 640, 256, False, False, 0.0), (320, 320, 4096, True, True, 30.0),
                                                        (1280, 1280, 256, True, True, -30.0), (320, 0, 4096, True, True, 100.0)])
def test_groupnorm_from_producer_statistics(C0, C1, P, in_f32, silu, offset):
    """One normalising pass over statistics the producers left behind (sdmi_op_groupnorm_acc): each concat source has its own
    records, groups of (C0 + C1) / 32 = 10 .. 80 channels are sums of 10-channel atoms.  The sources are written here by plain
    GEMMs with K = 64 (their epilogues take the statistics), the second one with another tile config (other T) than the first."""
    B = 2
    Hh = int(math.isqrt(P))
    g = torch.Generator().manual_seed(C0 + C1 + P)
    gamma = 1 + 0.1 * torch.randn((C0 + C1,), generator=g)
    beta = 0.1 * torch.randn((C0 + C1,), generator=g)
    xs, sts = [], []
    for Cs, cfg in ((C0, -1), (C1, 9)):
        if Cs == 0:
            xs.append(None); sts.append(None)
            continue
        a = torch.randn((B * P, 64), generator=g).half()
        w = (torch.randn((Cs, 64), generator=g) / 8).half()
        bias = torch.randn((Cs,), generator=g) * 0.5 + offset
        out = G.igemm(a.to(DEV).view(1, B * P, 1, 64), w.to(DEV), B=1, Hs=B * P, Ws=1, Ho=B * P, Wo=1, bias=bias.to(DEV),
                      out_f32=in_f32, cfg=cfg, gstat_rows_img=P)
        xs.append(out.view(B, Hh, Hh, Cs))
        sts.append(G.LAST_STAT)
    x = torch.cat([t.cpu().double() for t in xs if t is not None], dim=-1)
    ref = F.group_norm(x.permute(0, 3, 1, 2), 32, gamma.double(), beta.double(), 1e-5)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 3, 1)
    y = G.groupnorm_acc(xs[0], xs[1], sts[0], sts[1], gamma.to(DEV), beta.to(DEV), 1e-5, silu)
    err = (y.cpu().double() - ref).abs().max().item()
    y2 = G.groupnorm(xs[0], xs[1], gamma.to(DEV), beta.to(DEV), 1e-5, silu)          # the library's own statistics
    d2 = (y.float() - y2.float()).abs().max().item()
    G.log_metric(test="groupnorm_acc", C0=C0, C1=C1, P=P, offset=offset, max_abs_err=err, vs_own_stats=d2)
    assert err < (4e-3 if abs(offset) <= 30 else 8e-3), f"max abs err {err}"
    assert d2 < 4e-3, f"differs from the statistics-launch path by {d2}"


def _records_of(x_nhwc, atom, T, parts, gen):
    """statistics records of a tensor as a producer would have left them (common.h GnRec: [image][T][atoms][parts][2] fp32): the
    moments of each 10-channel atom over T row blocks; with parts = 2 every record is split arbitrarily between its two parts"""
    B, Hh, Ww, Cc = x_nhwc.shape
    P = Hh * Ww
    xa = x_nhwc.double().reshape(B, T, P // T, Cc // atom, atom)
    s1, s2 = xa.sum(dim=(2, 4)), (xa * xa).sum(dim=(2, 4))                   # [B][T][atoms]
    rec = torch.zeros((B, T, Cc // atom, parts, 2), dtype=torch.float32)
    if parts == 1:
        rec[..., 0, 0], rec[..., 0, 1] = s1.float(), s2.float()
    else:
        f = torch.rand(s1.shape, generator=gen).double()
        rec[..., 0, 0], rec[..., 0, 1] = (s1 * f).float(), (s2 * f).float()
        rec[..., 1, 0], rec[..., 1, 1] = (s1 * (1 - f)).float(), (s2 * (1 - f)).float()
    return rec


@pytest.mark.parametrize("case", [
    dict(B=2, H=32, W=32, C0=320, C1=0, Co=320, f32=True, silu=1, offset=0.0, T=8, parts=2, atom=10),       # a 32x32 ResBlock conv
    dict(B=2, H=32, W=32, C0=128, C1=128, Co=128, f32=True, silu=1, offset=0.0, T=4, parts=1, atom=4),      # concat of two sources
    dict(B=1, H=64, W=64, C0=128, C1=0, Co=128, f32=False, silu=1, offset=0.0, T=32, parts=2, atom=4),      # fp16 source (conv_merged's input), W = 64
    dict(B=2, H=16, W=16, C0=640, C1=640, Co=160, f32=True, silu=0, offset=0.0, T=1, parts=1, atom=10),     # 16x16, 1280 channels, 160-wide tile, no SiLU
    dict(B=2, H=32, W=32, C0=320, C1=0, Co=64, f32=True, silu=1, offset=30.0, T=8, parts=1, atom=10),       # |mean| = 30 sigma
    dict(B=2, H=32, W=32, C0=320, C1=0, Co=64, f32=True, silu=1, offset=-100.0, T=8, parts=1, atom=10),     # |mean| = 100 sigma
])
def test_conv_with_groupnorm_inside(case):
    """GroupNorm -> SiLU -> conv3x3 as ONE launch (include/sdmi.h sdmi_gemm_desc::hgn_*, csrc/gemm.hip conv3_halo_kernel<.., GN>;
    sd/diffusion.py:173-179,199-205): the halo conv's producer waves normalise the raw tensor(s) from their producers' statistics
    records.  Against (1) the fp64 GroupNorm(+SiLU) rounded to fp16 -- what the normalising launch writes -- through an fp64 conv,
    and (2) the library's own two-launch path (sdmi_op_groupnorm_acc + the same conv config).  Rows whose mean is 30 / 100 sigma
    from zero included (the offsets of test_groupnorm_large_mean): the raw tensor is read in fp32, so a large mean costs what it
    costs the separate kernel.  Every halo config built with the variant, one-pass and split-K."""
    c = case
    g = torch.Generator().manual_seed(c["C0"] + c["Co"] + int(abs(c["offset"])))
    mk = lambda ch: torch.randn((c["B"], c["H"], c["W"], ch), generator=g) * (1 + 0.5 * torch.rand((ch,), generator=g)) + c["offset"] + 0.3 * torch.randn((ch,), generator=g)
    x0, x1 = mk(c["C0"]), (mk(c["C1"]) if c["C1"] else None)
    if not c["f32"]:
        x0 = x0.half().float()
        x1 = None if x1 is None else x1.half().float()
    cin = c["C0"] + c["C1"]
    gamma = 1 + 0.2 * torch.randn((cin,), generator=g)
    beta = 0.2 * torch.randn((cin,), generator=g)
    w = (torch.randn((c["Co"], cin, 3, 3), generator=g) / math.sqrt(9 * cin)).half().float()
    xin = x0 if x1 is None else torch.cat([x0, x1], -1)
    yn = F.group_norm(xin.double().permute(0, 3, 1, 2), 32, gamma.double(), beta.double(), 1e-5)
    yn = (F.silu(yn) if c["silu"] else yn).permute(0, 2, 3, 1)
    ref = _conv_ref_f64(yn.half().float(), w, 1, 0)                       # the conv's operand is the fp16-rounded normalised tensor
    wp = G.pack_conv(w.to(DEV))
    dt = torch.float32 if c["f32"] else torch.float16
    x0d, x1d = x0.to(DEV, dt), (None if x1 is None else x1.to(DEV, dt))
    rec0 = _records_of(x0, c["atom"], c["T"], c["parts"], g).to(DEV)
    rec1 = None if x1 is None else _records_of(x1, c["atom"], max(1, c["T"] // 2), 1, g).to(DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    # the two-launch path: the normalising kernel from the same records, then the plain conv
    st = lambda r: None if r is None else (r, r.shape[1], r.shape[3])
    t0 = G.groupnorm_acc(x0d, x1d, st(rec0), st(rec1), gd, bd, 1e-5, c["silu"], atom=c["atom"])
    shape = torch.empty((c["B"], c["H"], c["W"], cin), dtype=torch.float16, device=DEV)        # a0 of the fused launch: shape only
    lib = N_.load()
    names = [lib.sdmi_gemm_config_name(i).decode() for i in range(lib.sdmi_gemm_num_configs())]
    ran = 0
    sc = ref.abs().mean().item()
    for cfg in [i for i, nm in enumerate(names) if nm[0] == "h"]:
        for ksplit in (1, 2, 3):
            kw = dict(B=c["B"], Hs=c["H"], Ws=c["W"], Ho=c["H"], Wo=c["W"], ks=3, out_f32=True, cfg=cfg, ksplit=ksplit)
            try:
                out = G.igemm(shape, wp, hgn=dict(x0=x0d, x1=x1d, gamma=gd, beta=bd, eps=1e-5, silu=c["silu"], rec0=rec0, rec1=rec1), **kw)
            except ValueError as exc:
                assert "cannot apply GroupNorm" in str(exc) or "not applicable" in str(exc) or "LDS" in str(exc), exc
                continue
            two = G.igemm(t0.view(c["B"], c["H"], c["W"], cin), wp, **kw)
            ran += 1
            err = (out.cpu().double().view(ref.shape) - ref).abs().max().item()
            d2 = (out - two).abs().max().item()
            G.log_metric(test="conv_gn_inside", case=str(c), cfg=names[cfg], ksplit=ksplit, max_abs_err=err, vs_two_launches=d2, out_scale=sc)
            tol = 3e-3 if abs(c["offset"]) <= 30 else 8e-3
            assert err < tol, f"{c} {names[cfg]} split {ksplit}: max abs err {err} (outputs ~{sc:.2f})"
            assert d2 < tol, f"{c} {names[cfg]} split {ksplit}: differs from GroupNorm launch + conv by {d2}"
    assert ran >= 2, f"no halo config ran the variant ({ran})"


@pytest.mark.parametrize("C,P,offset,parts", [(640, 1024, 0.0, 2), (1280, 256, 0.0, 1), (640, 1024, 10.0, 2), (1280, 64, 30.0, 1), (320, 4096, -30.0, 2)])
def test_gemm_groupnorm_on_a_fragments(C, P, offset, parts):
    """GroupNorm (no SiLU, eps 1e-6) -> 1x1 conv (the attention block's groupnorm -> conv_input, sd/diffusion.py:294-298) as ONE
    GEMM that normalises its own A fragments from the producer's statistics records (sdmi_gemm_desc::gna_rec): against the fp64
    result, and against the two-launch path (gn_apply + GEMM) whose error it has to stay close to -- also on rows whose mean sits
    10 / 30 sigma away from zero (the mean is carried as a two-term fp16 sum)."""
    B, Nn, atom = 2, 320, 10
    g = torch.Generator().manual_seed(C + P)
    x = (torch.randn((B, P, C), generator=g) * (1 + torch.rand((C,), generator=g)) + offset + torch.randn((C,), generator=g)).half()
    gamma = 1 + 0.2 * torch.randn((C,), generator=g)
    beta = 0.3 * torch.randn((C,), generator=g)
    w = (torch.randn((Nn, C), generator=g) / math.sqrt(C)).half()
    bias = torch.randn((Nn,), generator=g)
    # the producer's records: moments of the fp16 tensor per (row block of 64, atom); parts = 2 splits them over two slots
    T = P // 64
    xb = x.double().view(B, T, 64, C // atom, atom)
    rec = torch.zeros((B, T, C // atom, parts, 2), dtype=torch.float32)
    s1, s2 = xb.sum(dim=(2, 4)), (xb * xb).sum(dim=(2, 4))
    if parts == 2:
        rec[..., 0, 0], rec[..., 0, 1] = (0.25 * s1).float(), (0.25 * s2).float()
        rec[..., 1, 0], rec[..., 1, 1] = (0.75 * s1).float(), (0.75 * s2).float()
    else:
        rec[..., 0, 0], rec[..., 0, 1] = s1.float(), s2.float()
    yn = F.group_norm(x.double().permute(0, 2, 1).reshape(B, C, P, 1), 32, gamma.double(), beta.double(), 1e-6)
    yn = yn.reshape(B, C, P).permute(0, 2, 1)
    ref = yn.reshape(B * P, C) @ w.double().t() + bias.double()
    xd, wd, bd = x.to(DEV).view(1, B * P, 1, C), w.to(DEV), bias.to(DEV)
    y16 = G.groupnorm(x.to(DEV).view(B, P, 1, C), None, gamma.to(DEV), beta.to(DEV), 1e-6, False)          # two-launch path
    two = G.igemm(y16.view(1, B * P, 1, C), wd, B=1, Hs=B * P, Ws=1, Ho=B * P, Wo=1, bias=bd, out_f32=True)
    e_two = (two.cpu().double() - ref).abs().max().item()
    ran = 0
    for cfg in range(N_.load().sdmi_gemm_num_configs()):
        try:
            out = G.igemm(xd, wd, B=1, Hs=B * P, Ws=1, Ho=B * P, Wo=1, bias=bd, out_f32=True, cfg=cfg,
                          gna=(rec.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6, P))
        except ValueError as exc:
            assert "cannot apply GroupNorm" in str(exc) or "not applicable" in str(exc), exc
            continue
        ran += 1
        err = (out.cpu().double() - ref).abs().max().item()
        G.log_metric(test="gemm_gn_fragments", C=C, P=P, offset=offset, cfg=N_.load().sdmi_gemm_config_name(cfg).decode(), max_abs_err=err, two_launch_err=e_two)
        assert err < max(1.5 * e_two, 6e-3), f"cfg {cfg}: max abs err {err:.3e} (two launches: {e_two:.3e})"
    assert ran >= 3, "no config ran the variant"


@pytest.mark.parametrize("offset,bm", [(2.0, 32), (30.0, 32), (-100.0, 32), (2.0, 64), (30.0, 64)])
def test_back_to_back_gemm_groupnorm_statistics(offset, bm):
    """the feed-forward form of csrc/b2b.hip (the attention block's output at 64x64) leaves the statistics of its output: the
    moments of WHAT THE GROUPNORM WILL READ -- the fp32 stream when there is one (as store_tile, splitk_finalize and the stem take
    them), also where the group's mean is 30 / 100 sigma away from zero and the fp16 shadow is 2^-11 |mean| off"""
    import ctypes as C
    Cc, B, P = 320, 2, 128
    M = B * P
    g = torch.Generator().manual_seed(5)
    a1 = torch.randn((M, Cc), generator=g).half()
    w1 = (torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)).half()
    b1 = torch.randn((Cc,), generator=g)
    r1 = torch.randn((M, Cc), generator=g)
    r2 = torch.randn((M, Cc), generator=g) + offset
    gamma = 1 + 0.1 * torch.randn((Cc,), generator=g)
    beta = 0.1 * torch.randn((Cc,), generator=g)
    w2 = torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)
    b2 = torch.randn((Cc,), generator=g)
    wp = (torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)).half()
    wf, _, hf = G.ln_fold_prep(w2.to(DEV), gamma.to(DEV), beta.to(DEV), b2.to(DEV))
    wf = torch.cat([wf, wp.to(DEV)], dim=1).contiguous()
    a1d, w1d, b1d, r1d, r2d = a1.to(DEV), w1.to(DEV), b1.to(DEV), r1.to(DEV), r2.to(DEV)
    outs = []
    for with_acc in (False, True):
        out = torch.full((M, Cc), float("nan"), device=DEV)
        out16 = torch.full((M, Cc), float("nan"), dtype=torch.float16, device=DEV)
        rec = torch.full((B, P // bm, Cc // 10, 1, 2), float("nan"), device=DEV)       # one record row per bm-row tile (the 64-row form too: round 5)
        d = N_.B2bDesc()
        d.a1, d.lda1, d.w1, d.b1 = a1d.data_ptr(), Cc, w1d.data_ptr(), b1d.data_ptr()
        d.r1, d.r1_f32 = r1d.data_ptr(), 1
        d.w2, d.K2, d.h2, d.partial, d.cscale = wf.data_ptr(), 640, hf.data_ptr(), 1, 0.0
        d.r2, d.r2_f32 = r2d.data_ptr(), 1
        d.out, d.out_f32, d.out16 = out.data_ptr(), 1, out16.data_ptr()
        d.M, d.eps, d.bm = M, 1e-5, bm
        if with_acc:
            d.gacc, d.gacc_atom, d.gacc_rows_img = rec.data_ptr(), 10, P
        N_.check(N_.load().sdmi_op_b2b(C.byref(d), 1, None, N_.cur_stream()), "b2b")
        torch.cuda.synchronize()
        outs.append((out.clone(), out16.clone(), rec))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])     # the statistics form stores the same bits
    assert not torch.isnan(outs[1][2]).any()
    s1, s2 = G.stat_moments(outs[1][2])
    r1m, r2m = _atom_moments(outs[1][0].cpu().double(), B)                                  # moments of the fp32 stream values
    assert ((s1 - r1m).abs() / (r1m.abs() + P)).max().item() < 2e-6
    assert ((s2 - r2m).abs() / r2m).max().item() < 2e-6
    # the variance a GroupNorm derives from the records is the fp32 tensor's own (E[x^2] - E[x]^2 in fp64 from fp32 partial sums
    # loses |mean|^2 / var x 2^-24 per partial: a third of the variance at 100 sigma, which is why norm.hip states its 100 sigma bound)
    n = P * 10
    var_rec = (s2 / n - (s1 / n) ** 2)
    var_ref = (r2m / n - (r1m / n) ** 2)
    tol = 2e-5 if abs(offset) < 10 else (2e-3 if abs(offset) < 50 else 5e-2)
    assert ((var_rec - var_ref).abs() / var_ref).max().item() < tol
