"""Static description of the SD-v1.5 UNet ("Diffusion") topology and its weight ABI.

The weight ABI of the drop-in is the set of state-dict key names / PyTorch-layout
shapes that the reference's converter produces (reference: sd/model_converter.py:13-650,
1009-1024) and that sd/diffusion.py's module tree consumes (sd/diffusion.py:529-626,
797-812).  This module re-derives that manifest from a compact stage table so that
synthetic weights, strict loading and the native packer agree on names and shapes
without the reference being present.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

N_TIME = 1280          # reference: sd/diffusion.py:111 (n_time=1280)
D_CONTEXT = 768        # reference: sd/diffusion.py:243 (d_context=768)
N_HEADS = 8
GN_GROUPS = 32

# Stage table.  Each stage is a list of ops:
#   ("conv", cin, cout, stride)      plain 3x3 conv            (sd/diffusion.py:545,553,561,569)
#   ("res", cin, cout)               UNET_ResidualBlock        (sd/diffusion.py:111-209)
#   ("attn", n_head, d_head)         UNET_AttentionBlock       (sd/diffusion.py:243-381)
#   ("up", c)                        Upsample                  (sd/diffusion.py:400-435)
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

Shape = Tuple[int, ...]


def _res_keys(p: str, cin: int, cout: int) -> List[Tuple[str, Shape]]:
    ks = [
        (f"{p}.groupnorm_feature.weight", (cin,)),
        (f"{p}.groupnorm_feature.bias", (cin,)),
        (f"{p}.conv_feature.weight", (cout, cin, 3, 3)),
        (f"{p}.conv_feature.bias", (cout,)),
        (f"{p}.linear_time.weight", (cout, N_TIME)),
        (f"{p}.linear_time.bias", (cout,)),
        (f"{p}.groupnorm_merged.weight", (cout,)),
        (f"{p}.groupnorm_merged.bias", (cout,)),
        (f"{p}.conv_merged.weight", (cout, cout, 3, 3)),
        (f"{p}.conv_merged.bias", (cout,)),
    ]
    if cin != cout:
        ks += [
            (f"{p}.residual_layer.weight", (cout, cin, 1, 1)),
            (f"{p}.residual_layer.bias", (cout,)),
        ]
    return ks


def _attn_keys(p: str, n_head: int, d_head: int) -> List[Tuple[str, Shape]]:
    c = n_head * d_head
    return [
        (f"{p}.groupnorm.weight", (c,)),
        (f"{p}.groupnorm.bias", (c,)),
        (f"{p}.conv_input.weight", (c, c, 1, 1)),
        (f"{p}.conv_input.bias", (c,)),
        (f"{p}.layernorm_1.weight", (c,)),
        (f"{p}.layernorm_1.bias", (c,)),
        (f"{p}.attention_1.in_proj.weight", (3 * c, c)),
        (f"{p}.attention_1.out_proj.weight", (c, c)),
        (f"{p}.attention_1.out_proj.bias", (c,)),
        (f"{p}.layernorm_2.weight", (c,)),
        (f"{p}.layernorm_2.bias", (c,)),
        (f"{p}.attention_2.q_proj.weight", (c, c)),
        (f"{p}.attention_2.k_proj.weight", (c, D_CONTEXT)),
        (f"{p}.attention_2.v_proj.weight", (c, D_CONTEXT)),
        (f"{p}.attention_2.out_proj.weight", (c, c)),
        (f"{p}.attention_2.out_proj.bias", (c,)),
        (f"{p}.layernorm_3.weight", (c,)),
        (f"{p}.layernorm_3.bias", (c,)),
        (f"{p}.linear_geglu_1.weight", (8 * c, c)),
        (f"{p}.linear_geglu_1.bias", (8 * c,)),
        (f"{p}.linear_geglu_2.weight", (c, 4 * c)),
        (f"{p}.linear_geglu_2.bias", (c,)),
        (f"{p}.conv_output.weight", (c, c, 1, 1)),
        (f"{p}.conv_output.bias", (c,)),
    ]


def _op_keys(p: str, op) -> List[Tuple[str, Shape]]:
    kind = op[0]
    if kind == "conv":
        _, cin, cout, _s = op
        return [(f"{p}.weight", (cout, cin, 3, 3)), (f"{p}.bias", (cout,))]
    if kind == "res":
        return _res_keys(p, op[1], op[2])
    if kind == "attn":
        return _attn_keys(p, op[1], op[2])
    if kind == "up":
        c = op[1]
        return [(f"{p}.conv.weight", (c, c, 3, 3)), (f"{p}.conv.bias", (c,))]
    raise ValueError(kind)


def diffusion_manifest() -> "OrderedDict[str, Shape]":
    """Ordered {state-dict key: shape} of the reference's ``Diffusion`` module
    (sd/diffusion.py:797-812); 654 tensors, 859 520 964 parameters."""
    m: "OrderedDict[str, Shape]" = OrderedDict()
    m["time_embedding.linear_1.weight"] = (N_TIME, 320)
    m["time_embedding.linear_1.bias"] = (N_TIME,)
    m["time_embedding.linear_2.weight"] = (N_TIME, N_TIME)
    m["time_embedding.linear_2.bias"] = (N_TIME,)
    for i, stage in enumerate(ENCODERS):
        for j, op in enumerate(stage):
            m.update(_op_keys(f"unet.encoders.{i}.{j}", op))
    for j, op in enumerate(BOTTLENECK):
        m.update(_op_keys(f"unet.bottleneck.{j}", op))
    for i, stage in enumerate(DECODERS):
        for j, op in enumerate(stage):
            m.update(_op_keys(f"unet.decoders.{i}.{j}", op))
    m["final.groupnorm.weight"] = (320,)
    m["final.groupnorm.bias"] = (320,)
    m["final.conv.weight"] = (4, 320, 3, 3)
    m["final.conv.bias"] = (4,)
    return m


def res_block_manifest(prefix: str, cin: int, cout: int) -> "OrderedDict[str, Shape]":
    return OrderedDict(_res_keys(prefix, cin, cout))


def attn_block_manifest(prefix: str, n_head: int, d_head: int) -> "OrderedDict[str, Shape]":
    return OrderedDict(_attn_keys(prefix, n_head, d_head))


def n_params(manifest: Dict[str, Shape]) -> int:
    total = 0
    for shp in manifest.values():
        n = 1
        for s in shp:
            n *= s
        total += n
    return total


# ---------------------------------------------------------------------------------------------
# "Next rows" (SURVEY 8f): CLIP text encoder and VAE.  Same idea: the weight ABI is the
# reference's state-dict key set (sd/model_converter.py:651-1008,1025-1054).
# VAE stages are nn.Sequential positions (sd/encoder.py:8-94, sd/decoder.py:196-340):
#   ("conv", cin, cout, ks, stride, pad) | ("res", cin, cout) | ("attn", c) | ("up",) | ("gn", c) | ("silu",)
VAE_ENCODER = [
    ("conv", 3, 128, 3, 1, 1), ("res", 128, 128), ("res", 128, 128),
    ("conv", 128, 128, 3, 2, 0), ("res", 128, 256), ("res", 256, 256),
    ("conv", 256, 256, 3, 2, 0), ("res", 256, 512), ("res", 512, 512),
    ("conv", 512, 512, 3, 2, 0), ("res", 512, 512), ("res", 512, 512), ("res", 512, 512),
    ("attn", 512), ("res", 512, 512), ("gn", 512), ("silu",),
    ("conv", 512, 8, 3, 1, 1), ("conv", 8, 8, 1, 1, 0),
]
VAE_DECODER = [
    ("conv", 4, 4, 1, 1, 0), ("conv", 4, 512, 3, 1, 1), ("res", 512, 512), ("attn", 512),
    ("res", 512, 512), ("res", 512, 512), ("res", 512, 512), ("res", 512, 512),
    ("up",), ("conv", 512, 512, 3, 1, 1), ("res", 512, 512), ("res", 512, 512), ("res", 512, 512),
    ("up",), ("conv", 512, 512, 3, 1, 1), ("res", 512, 256), ("res", 256, 256), ("res", 256, 256),
    ("up",), ("conv", 256, 256, 3, 1, 1), ("res", 256, 128), ("res", 128, 128), ("res", 128, 128),
    ("gn", 128), ("silu",), ("conv", 128, 3, 3, 1, 1),
]


def _vae_manifest(stages):
    m: "OrderedDict[str, Shape]" = OrderedDict()
    norm_keys = set()
    for i, op in enumerate(stages):
        p = str(i)
        if op[0] == "conv":
            _, cin, cout, ks, _s, _p = op
            m[f"{p}.weight"] = (cout, cin, ks, ks)
            m[f"{p}.bias"] = (cout,)
        elif op[0] == "res":
            _, cin, cout = op
            m[f"{p}.groupnorm_1.weight"] = (cin,)
            m[f"{p}.groupnorm_1.bias"] = (cin,)
            m[f"{p}.conv_1.weight"] = (cout, cin, 3, 3)
            m[f"{p}.conv_1.bias"] = (cout,)
            m[f"{p}.groupnorm_2.weight"] = (cout,)
            m[f"{p}.groupnorm_2.bias"] = (cout,)
            m[f"{p}.conv_2.weight"] = (cout, cout, 3, 3)
            m[f"{p}.conv_2.bias"] = (cout,)
            if cin != cout:
                m[f"{p}.residual_layer.weight"] = (cout, cin, 1, 1)
                m[f"{p}.residual_layer.bias"] = (cout,)
        elif op[0] == "attn":
            c = op[1]
            m[f"{p}.groupnorm.weight"] = (c,)
            m[f"{p}.groupnorm.bias"] = (c,)
            m[f"{p}.attention.in_proj.weight"] = (3 * c, c)
            m[f"{p}.attention.in_proj.bias"] = (3 * c,)
            m[f"{p}.attention.out_proj.weight"] = (c, c)
            m[f"{p}.attention.out_proj.bias"] = (c,)
        elif op[0] == "gn":
            m[f"{p}.weight"] = (op[1],)
            m[f"{p}.bias"] = (op[1],)
            norm_keys.update({f"{p}.weight", f"{p}.bias"})
    return m, norm_keys


def vae_encoder_manifest():
    """({key: shape}, norm-parameter keys with positional names) of VAE_Encoder: 104 tensors."""
    return _vae_manifest(VAE_ENCODER)


def vae_decoder_manifest():
    """Same for VAE_Decoder: 136 tensors, 49 490 199 parameters."""
    return _vae_manifest(VAE_DECODER)


CLIP_VOCAB, CLIP_DIM, CLIP_TOKENS, CLIP_LAYERS, CLIP_HEADS = 49408, 768, 77, 12, 12


def clip_manifest() -> "OrderedDict[str, Shape]":
    """CLIP text encoder (sd/clip.py:198-225): 148 tensors, 123 060 480 parameters."""
    m: "OrderedDict[str, Shape]" = OrderedDict()
    m["embedding.position_embedding"] = (CLIP_TOKENS, CLIP_DIM)
    m["embedding.token_embedding.weight"] = (CLIP_VOCAB, CLIP_DIM)
    for i in range(CLIP_LAYERS):
        p = f"layers.{i}"
        m[f"{p}.layernorm_1.weight"] = (CLIP_DIM,)
        m[f"{p}.layernorm_1.bias"] = (CLIP_DIM,)
        m[f"{p}.attention.in_proj.weight"] = (3 * CLIP_DIM, CLIP_DIM)
        m[f"{p}.attention.in_proj.bias"] = (3 * CLIP_DIM,)
        m[f"{p}.attention.out_proj.weight"] = (CLIP_DIM, CLIP_DIM)
        m[f"{p}.attention.out_proj.bias"] = (CLIP_DIM,)
        m[f"{p}.layernorm_2.weight"] = (CLIP_DIM,)
        m[f"{p}.layernorm_2.bias"] = (CLIP_DIM,)
        m[f"{p}.linear_1.weight"] = (4 * CLIP_DIM, CLIP_DIM)
        m[f"{p}.linear_1.bias"] = (4 * CLIP_DIM,)
        m[f"{p}.linear_2.weight"] = (CLIP_DIM, 4 * CLIP_DIM)
        m[f"{p}.linear_2.bias"] = (CLIP_DIM,)
    m["layernorm.weight"] = (CLIP_DIM,)
    m["layernorm.bias"] = (CLIP_DIM,)
    return m
