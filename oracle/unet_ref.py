"""Oracle: UNet noise predictor, functional fp32 restatement (test infrastructure only).

State is a flat ``{key: tensor}`` dict with the reference's key names
(sd/model_converter.py:13-650).  Activations are NCHW fp32 like the reference.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# The UNet graph as the reference builds it (sd/diffusion.py:543-626), restated here so that the checker does not share its
# wiring with the product (pytorch_stable_diffusion_amd/arch.py has its own copy; tests/test_oracle_golden.py compares the two
# and the full-UNet goldens from the imported reference pin both).  ("conv", cin, cout, stride) | ("res", cin, cout) |
# ("attn", heads, head_dim) | ("up", channels)
ENCODERS = [
    [("conv", 4, 320, 1)],
    [("res", 320, 320), ("attn", 8, 40)],
    [("res", 320, 320), ("attn", 8, 40)],
    [("conv", 320, 320, 2)],
    [("res", 320, 640), ("attn", 8, 80)],
    [("res", 640, 640), ("attn", 8, 80)],
    [("conv", 640, 640, 2)],
    [("res", 640, 1280), ("attn", 8, 160)],
    [("res", 1280, 1280), ("attn", 8, 160)],
    [("conv", 1280, 1280, 2)],
    [("res", 1280, 1280)],
    [("res", 1280, 1280)],
]
BOTTLENECK = [("res", 1280, 1280), ("attn", 8, 160), ("res", 1280, 1280)]
DECODERS = [
    [("res", 2560, 1280)],
    [("res", 2560, 1280)],
    [("res", 2560, 1280), ("up", 1280)],
    [("res", 2560, 1280), ("attn", 8, 160)],
    [("res", 2560, 1280), ("attn", 8, 160)],
    [("res", 1920, 1280), ("attn", 8, 160), ("up", 1280)],
    [("res", 1920, 640), ("attn", 8, 80)],
    [("res", 1280, 640), ("attn", 8, 80)],
    [("res", 960, 640), ("attn", 8, 80), ("up", 640)],
    [("res", 960, 320), ("attn", 8, 40)],
    [("res", 640, 320), ("attn", 8, 40)],
    [("res", 640, 320), ("attn", 8, 40)],
]

# Rounding experiments (tests/golden/stress_floor.py): what fp16 STORAGE of one operand class alone costs, the arithmetic staying
# fp32 -- QUANT = set of {"w": weights of every conv / linear, "a": the activation operand of every conv / linear,
# "attn": q, k, v and the probabilities}.  Empty (the default): the plain fp32 oracle.
QUANT: set = set()
# Per-block trace (tests/test_gpu_stress.py attribution): a list that receives (key prefix, op, input, output) of every stage op
TRACE = None


def _q(t: torch.Tensor, what: str) -> torch.Tensor:
    return t.half().float() if (what in QUANT and t is not None) else t


def _linear(x, w, b=None):
    return F.linear(_q(x, "a"), _q(w, "w"), b)


def _conv2d(x, w, b=None, **kw):
    return F.conv2d(_q(x, "a"), _q(w, "w"), b, **kw)


def time_mlp(sd: SD, temb: torch.Tensor) -> torch.Tensor:
    """(1,320) -> (1,1280).  Reference: sd/diffusion.py:64-76 (Linear, SiLU, Linear)."""
    h = _linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    return _linear(F.silu(h), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])


def residual_block(sd: SD, p: str, x: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
    """Reference: sd/diffusion.py:145-209.  GN32->SiLU->conv3x3; + Linear(SiLU(time)) per
    (n,c); GN32->SiLU->conv3x3; + identity | conv1x1 skip."""
    h = F.group_norm(x, 32, sd[f"{p}.groupnorm_feature.weight"], sd[f"{p}.groupnorm_feature.bias"], eps=1e-5)
    h = _conv2d(F.silu(h), sd[f"{p}.conv_feature.weight"], sd[f"{p}.conv_feature.bias"], padding=1)
    t = _linear(F.silu(time), sd[f"{p}.linear_time.weight"], sd[f"{p}.linear_time.bias"])
    h = h + t[:, :, None, None]
    h = F.group_norm(h, 32, sd[f"{p}.groupnorm_merged.weight"], sd[f"{p}.groupnorm_merged.bias"], eps=1e-5)
    h = _conv2d(F.silu(h), sd[f"{p}.conv_merged.weight"], sd[f"{p}.conv_merged.bias"], padding=1)
    skip_w = sd.get(f"{p}.residual_layer.weight")
    skip = x if skip_w is None else _conv2d(x, skip_w, sd[f"{p}.residual_layer.bias"])
    return h + skip


def _heads(t: torch.Tensor, n_head: int) -> torch.Tensor:
    b, s, c = t.shape
    return t.reshape(b, s, n_head, c // n_head).permute(0, 2, 1, 3)


def _sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool = False) -> torch.Tensor:
    """softmax(q k^T / sqrt(d)) v over materialised scores.
    Reference: sd/attention.py:55-80 (mask before scale) and :231-244."""
    d = q.shape[-1]
    q, k, v = _q(q, "attn"), _q(k, "attn"), _q(v, "attn")
    w = q @ k.transpose(-1, -2)
    if causal:
        mask = torch.ones_like(w, dtype=torch.bool).triu(1)
        w = w.masked_fill(mask, float("-inf"))
    w = w / math.sqrt(d)
    w = _q(torch.softmax(w, dim=-1), "attn")
    o = w @ v                       # (b, h, s, d)
    b, h, s, _ = o.shape
    return o.permute(0, 2, 1, 3).reshape(b, s, h * d)


def self_attention(sd: SD, p: str, x: torch.Tensor, n_head: int, causal: bool = False) -> torch.Tensor:
    """Reference: sd/attention.py:27-93.  Fused q|k|v projection (bias optional)."""
    qkv = _linear(x, sd[f"{p}.in_proj.weight"], sd.get(f"{p}.in_proj.bias"))
    q, k, v = qkv.chunk(3, dim=-1)
    o = _sdpa(_heads(q, n_head), _heads(k, n_head), _heads(v, n_head), causal)
    return _linear(o, sd[f"{p}.out_proj.weight"], sd.get(f"{p}.out_proj.bias"))


def cross_attention(sd: SD, p: str, x: torch.Tensor, y: torch.Tensor, n_head: int) -> torch.Tensor:
    """Reference: sd/attention.py:161-253.  q from x, k/v from context y."""
    q = _linear(x, sd[f"{p}.q_proj.weight"], sd.get(f"{p}.q_proj.bias"))
    k = _linear(y, sd[f"{p}.k_proj.weight"], sd.get(f"{p}.k_proj.bias"))
    v = _linear(y, sd[f"{p}.v_proj.weight"], sd.get(f"{p}.v_proj.bias"))
    o = _sdpa(_heads(q, n_head), _heads(k, n_head), _heads(v, n_head))
    return _linear(o, sd[f"{p}.out_proj.weight"], sd.get(f"{p}.out_proj.bias"))


def attention_block(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, n_head: int) -> torch.Tensor:
    """Reference: sd/diffusion.py:271-381.  GN(eps 1e-6, no SiLU) -> conv1x1 -> tokens;
    LN->self-attn->+ ; LN->cross-attn->+ ; LN->Linear(C->8C)->FIRST HALF ONLY->Linear(4C->C)->+
    (the GeGLU gate is computed and discarded: sd/diffusion.py:359-363); conv1x1; + input."""
    n, c, hh, ww = x.shape
    h = F.group_norm(x, 32, sd[f"{p}.groupnorm.weight"], sd[f"{p}.groupnorm.bias"], eps=1e-6)
    h = _conv2d(h, sd[f"{p}.conv_input.weight"], sd[f"{p}.conv_input.bias"])
    t = h.reshape(n, c, hh * ww).transpose(1, 2)          # (n, hw, c)
    u = F.layer_norm(t, (c,), sd[f"{p}.layernorm_1.weight"], sd[f"{p}.layernorm_1.bias"])
    t = t + self_attention(sd, f"{p}.attention_1", u, n_head)
    u = F.layer_norm(t, (c,), sd[f"{p}.layernorm_2.weight"], sd[f"{p}.layernorm_2.bias"])
    t = t + cross_attention(sd, f"{p}.attention_2", u, context, n_head)
    u = F.layer_norm(t, (c,), sd[f"{p}.layernorm_3.weight"], sd[f"{p}.layernorm_3.bias"])
    g = _linear(u, sd[f"{p}.linear_geglu_1.weight"], sd[f"{p}.linear_geglu_1.bias"])
    live = g[..., : 4 * c]                                  # gate half g[..., 4c:] is dead
    t = t + _linear(live, sd[f"{p}.linear_geglu_2.weight"], sd[f"{p}.linear_geglu_2.bias"])
    h = t.transpose(1, 2).reshape(n, c, hh, ww)
    return _conv2d(h, sd[f"{p}.conv_output.weight"], sd[f"{p}.conv_output.bias"]) + x


def upsample(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Reference: sd/diffusion.py:412-435 (nearest x2, conv3x3)."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return _conv2d(x, sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"], padding=1)


def _run_stage(sd: SD, prefix: str, stage, x, context, time):
    """Reference: sd/diffusion.py:458-496 (type-switched sequential)."""
    for j, op in enumerate(stage):
        p = f"{prefix}.{j}"
        x_in = x
        if op[0] == "conv":
            x = _conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], stride=op[3], padding=1)
        elif op[0] == "res":
            x = residual_block(sd, p, x, time)
        elif op[0] == "attn":
            x = attention_block(sd, p, x, context, op[1])
        elif op[0] == "up":
            x = upsample(sd, p, x)
        if TRACE is not None:
            TRACE.append((p, op, x_in, x))
    return x


def output_layer(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """Reference: sd/diffusion.py:714-748 (GN32 -> SiLU -> conv3x3 320->4)."""
    h = F.group_norm(x, 32, sd["final.groupnorm.weight"], sd["final.groupnorm.bias"], eps=1e-5)
    return _conv2d(F.silu(h), sd["final.conv.weight"], sd["final.conv.bias"], padding=1)


def unet_body(sd: SD, x: torch.Tensor, context: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
    """Reference: sd/diffusion.py:628-676 (encoders with skip push, bottleneck, decoders with
    cat(x, skip.pop()))."""
    skips: List[torch.Tensor] = []
    for i, stage in enumerate(ENCODERS):
        x = _run_stage(sd, f"unet.encoders.{i}", stage, x, context, time)
        skips.append(x)
    x = _run_stage(sd, "unet.bottleneck", BOTTLENECK, x, context, time)
    for i, stage in enumerate(DECODERS):
        x = torch.cat((x, skips.pop()), dim=1)
        x = _run_stage(sd, f"unet.decoders.{i}", stage, x, context, time)
    return x


@torch.no_grad()
def diffusion_forward(sd: SD, latent: torch.Tensor, context: torch.Tensor, temb: torch.Tensor) -> torch.Tensor:
    """Reference: sd/diffusion.py:797-837.  latent (B,4,h,w), context (B,77,768), temb (1,320)
    (NOT duplicated over batch; broadcasts inside the residual blocks) -> (B,4,h,w)."""
    time = time_mlp(sd, temb)
    return output_layer(sd, unet_body(sd, latent, context, time))
