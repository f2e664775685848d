"""Checkpoint converter: standard SD-v1.x checkpoint (``v1-5-pruned-emaonly.ckpt`` layout) -> the four state
dicts of the reference's weight ABI (reference: sd/model_converter.py:3-1056, behaviour only).

Instead of a literal table, the mapping is DERIVED from the same stage tables that describe the models
(``arch.ENCODERS/BOTTLENECK/DECODERS``, ``arch.VAE_*``, CLIP depth): ``conversion_plan()`` returns, per model,
``{dest_key: {"op": "copy" | "cat" | "cat+reshape" | "copy+reshape", "src": [checkpoint keys], "shape": [...]}}``.
The plan is pinned key-for-key against the reference converter (tests/golden/converter_map.json, produced by
executing the reference on symbolic tensors)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import torch

from . import arch

_UNET = "model.diffusion_model"
_VAE = "first_stage_model"
_CLIP = "cond_stage_model.transformer.text_model"


def _copy(plan, dst, src):
    plan[dst] = {"op": "copy", "src": [src]}


def _wb(plan, dst, src):
    _copy(plan, dst + ".weight", src + ".weight")
    _copy(plan, dst + ".bias", src + ".bias")


def _unet_res(plan, dst, src, cin, cout):
    _wb(plan, f"{dst}.groupnorm_feature", f"{src}.in_layers.0")
    _wb(plan, f"{dst}.conv_feature", f"{src}.in_layers.2")
    _wb(plan, f"{dst}.linear_time", f"{src}.emb_layers.1")
    _wb(plan, f"{dst}.groupnorm_merged", f"{src}.out_layers.0")
    _wb(plan, f"{dst}.conv_merged", f"{src}.out_layers.3")
    if cin != cout:
        _wb(plan, f"{dst}.residual_layer", f"{src}.skip_connection")


def _unet_attn(plan, dst, src):
    t = f"{src}.transformer_blocks.0"
    _wb(plan, f"{dst}.groupnorm", f"{src}.norm")
    _wb(plan, f"{dst}.conv_input", f"{src}.proj_in")
    _wb(plan, f"{dst}.layernorm_1", f"{t}.norm1")
    plan[f"{dst}.attention_1.in_proj.weight"] = {"op": "cat", "src": [f"{t}.attn1.to_{x}.weight" for x in "qkv"]}
    _wb(plan, f"{dst}.attention_1.out_proj", f"{t}.attn1.to_out.0")
    _wb(plan, f"{dst}.layernorm_2", f"{t}.norm2")
    for x in "qkv":
        _copy(plan, f"{dst}.attention_2.{x}_proj.weight", f"{t}.attn2.to_{x}.weight")
    _wb(plan, f"{dst}.attention_2.out_proj", f"{t}.attn2.to_out.0")
    _wb(plan, f"{dst}.layernorm_3", f"{t}.norm3")
    _wb(plan, f"{dst}.linear_geglu_1", f"{t}.ff.net.0.proj")
    _wb(plan, f"{dst}.linear_geglu_2", f"{t}.ff.net.2")
    _wb(plan, f"{dst}.conv_output", f"{src}.proj_out")


def _unet_stage(plan, dst_group, src_group, stage):
    for j, op in enumerate(stage):
        dst, src = f"{dst_group}.{j}", f"{src_group}.{j}"
        if op[0] == "conv":
            _wb(plan, dst, src if op[3] == 1 else f"{src}.op")     # stride-2 convs live in a Downsample wrapper
        elif op[0] == "res":
            _unet_res(plan, dst, src, op[1], op[2])
        elif op[0] == "attn":
            _unet_attn(plan, dst, src)
        elif op[0] == "up":
            _wb(plan, f"{dst}.conv", f"{src}.conv")


def diffusion_plan() -> Dict[str, dict]:
    plan: Dict[str, dict] = OrderedDict()
    _wb(plan, "time_embedding.linear_1", f"{_UNET}.time_embed.0")
    _wb(plan, "time_embedding.linear_2", f"{_UNET}.time_embed.2")
    for i, stage in enumerate(arch.ENCODERS):
        _unet_stage(plan, f"unet.encoders.{i}", f"{_UNET}.input_blocks.{i}", stage)
    _unet_stage(plan, "unet.bottleneck", f"{_UNET}.middle_block", arch.BOTTLENECK)
    for i, stage in enumerate(arch.DECODERS):
        _unet_stage(plan, f"unet.decoders.{i}", f"{_UNET}.output_blocks.{i}", stage)
    _wb(plan, "final.groupnorm", f"{_UNET}.out.0")
    _wb(plan, "final.conv", f"{_UNET}.out.2")
    return plan


def _vae_res(plan, dst, src, cin, cout):
    _wb(plan, f"{dst}.groupnorm_1", f"{src}.norm1")
    _wb(plan, f"{dst}.conv_1", f"{src}.conv1")
    _wb(plan, f"{dst}.groupnorm_2", f"{src}.norm2")
    _wb(plan, f"{dst}.conv_2", f"{src}.conv2")
    if cin != cout:
        _wb(plan, f"{dst}.residual_layer", f"{src}.nin_shortcut")


def _vae_attn(plan, dst, src, c):
    _wb(plan, f"{dst}.groupnorm", f"{src}.norm")
    plan[f"{dst}.attention.in_proj.weight"] = {"op": "cat+reshape", "src": [f"{src}.{x}.weight" for x in "qkv"],
                                               "shape": [3 * c, c]}
    plan[f"{dst}.attention.in_proj.bias"] = {"op": "cat", "src": [f"{src}.{x}.bias" for x in "qkv"]}
    plan[f"{dst}.attention.out_proj.weight"] = {"op": "copy+reshape", "src": [f"{src}.proj_out.weight"], "shape": [c, c]}
    _copy(plan, f"{dst}.attention.out_proj.bias", f"{src}.proj_out.bias")


def _vae_plan(stages, side: str) -> Dict[str, dict]:
    """Walk the nn.Sequential positions and name the matching module of the LDM autoencoder:
    encoder: conv_in, down.L.block.B, down.L.downsample.conv, mid.{block_1,attn_1,block_2}, norm_out, conv_out,
             quant_conv;   decoder: post_quant_conv, conv_in, mid.*, up.L.block.B (L from 3 down to 0),
             up.L.upsample.conv, norm_out, conv_out."""
    plan: Dict[str, dict] = OrderedDict()
    root = f"{_VAE}.{side}"
    level = 0 if side == "encoder" else 3
    block = 0
    seen_mid_attn = False
    mid_blocks = 0
    n_conv = 0
    in_mid = False
    convs = [i for i, op in enumerate(stages) if op[0] == "conv"]
    for i, op in enumerate(stages):
        dst = str(i)
        if op[0] == "conv":
            first, last = i == convs[0], i == convs[-1]
            if side == "encoder":
                if first:
                    src = f"{root}.conv_in"
                elif last:
                    src = f"{_VAE}.quant_conv"
                elif i == convs[-2]:
                    src = f"{root}.conv_out"
                else:                                   # stride-2 downsample closes a level
                    src = f"{root}.down.{level}.downsample.conv"
                    level, block = level + 1, 0
            else:
                if first:
                    src = f"{_VAE}.post_quant_conv"
                elif i == convs[1]:
                    src = f"{root}.conv_in"
                    in_mid = True
                elif last:
                    src = f"{root}.conv_out"
                else:                                   # conv after an Upsample closes a level
                    src = f"{root}.up.{level}.upsample.conv"
                    level, block = level - 1, 0
            _wb(plan, dst, src)
            n_conv += 1
        elif op[0] == "res":
            nxt_attn = i + 1 < len(stages) and stages[i + 1][0] == "attn"
            if side == "encoder" and (nxt_attn or seen_mid_attn) and level == 3 and block >= 2:
                in_mid = True
            if in_mid and mid_blocks < 2 and (nxt_attn or seen_mid_attn or side == "decoder"):
                mid_blocks += 1
                src = f"{root}.mid.block_{mid_blocks}"
                if mid_blocks == 2:
                    in_mid = False
            else:
                group = "down" if side == "encoder" else "up"
                src = f"{root}.{group}.{level}.block.{block}"
                block += 1
            _vae_res(plan, dst, src, op[1], op[2])
        elif op[0] == "attn":
            seen_mid_attn = True
            _vae_attn(plan, dst, f"{root}.mid.attn_1", op[1])
        elif op[0] == "gn":
            _wb(plan, dst, f"{root}.norm_out")
    return plan


def encoder_plan() -> Dict[str, dict]:
    return _vae_plan(arch.VAE_ENCODER, "encoder")


def decoder_plan() -> Dict[str, dict]:
    return _vae_plan(arch.VAE_DECODER, "decoder")


def clip_plan() -> Dict[str, dict]:
    plan: Dict[str, dict] = OrderedDict()
    _copy(plan, "embedding.token_embedding.weight", f"{_CLIP}.embeddings.token_embedding.weight")
    _copy(plan, "embedding.position_embedding", f"{_CLIP}.embeddings.position_embedding.weight")
    for i in range(arch.CLIP_LAYERS):
        dst, src = f"layers.{i}", f"{_CLIP}.encoder.layers.{i}"
        _wb(plan, f"{dst}.layernorm_1", f"{src}.layer_norm1")
        for leaf in ("weight", "bias"):
            plan[f"{dst}.attention.in_proj.{leaf}"] = {"op": "cat", "src": [f"{src}.self_attn.{x}_proj.{leaf}" for x in "qkv"]}
        _wb(plan, f"{dst}.attention.out_proj", f"{src}.self_attn.out_proj")
        _wb(plan, f"{dst}.layernorm_2", f"{src}.layer_norm2")
        _wb(plan, f"{dst}.linear_1", f"{src}.mlp.fc1")
        _wb(plan, f"{dst}.linear_2", f"{src}.mlp.fc2")
    _wb(plan, "layernorm", f"{_CLIP}.final_layer_norm")
    return plan


def conversion_plan() -> Dict[str, Dict[str, dict]]:
    return {"diffusion": diffusion_plan(), "encoder": encoder_plan(), "decoder": decoder_plan(), "clip": clip_plan()}


def convert_state_dict(original: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """Apply the plan to a checkpoint's ``state_dict``.  Missing source keys raise KeyError."""
    out: Dict[str, Dict[str, torch.Tensor]] = {}
    for model, plan in conversion_plan().items():
        conv: Dict[str, torch.Tensor] = OrderedDict()
        for dst, rule in plan.items():
            srcs = [original[k] for k in rule["src"]]
            t = srcs[0] if rule["op"].startswith("copy") else torch.cat(srcs, 0)
            if "shape" in rule:
                t = t.reshape(rule["shape"])
            conv[dst] = t
        out[model] = conv
    return out


def load_from_standard_weights(input_file: str, device: str) -> Dict[str, Dict[str, torch.Tensor]]:
    """Same entry point as the reference (sd/model_converter.py:3-5): pickled Lightning checkpoint -> 4 dicts."""
    original = torch.load(input_file, map_location=device, weights_only=False)["state_dict"]
    return convert_state_dict(original)
