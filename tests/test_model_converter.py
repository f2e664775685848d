"""Checkpoint converter vs the reference converter's behaviour (tests/golden/converter_map.json: which checkpoint
keys feed each destination key, through which op), plus an application test on a fake checkpoint."""
import json
import os

import torch

from pytorch_stable_diffusion_amd import arch, model_converter
from tests import helpers as H


def _golden():
    with open(os.path.join(H.GOLDEN, "converter_map.json")) as f:
        return json.load(f)


def test_plan_matches_reference_converter_key_for_key():
    g = _golden()
    plan = model_converter.conversion_plan()
    assert set(plan) == set(g)
    for model in g:
        assert set(plan[model]) == set(g[model]), (model, sorted(set(plan[model]) ^ set(g[model]))[:6])
        for k, rule in g[model].items():
            mine = plan[model][k]
            assert mine["op"] == rule["op"] and mine["src"] == rule["src"], (model, k, mine, rule)
            assert mine.get("shape") == rule.get("shape"), (model, k)
    assert len(plan["diffusion"]) == 654 and len(plan["clip"]) == 148


def test_convert_fake_checkpoint_shapes_and_values():
    """Build a fake checkpoint whose tensors have the shapes the manifests imply, convert, and check every
    destination shape and the q|k|v concatenation order."""
    plan = model_converter.conversion_plan()
    manifests = {"diffusion": arch.diffusion_manifest(), "clip": arch.clip_manifest(),
                 "encoder": arch.vae_encoder_manifest()[0], "decoder": arch.vae_decoder_manifest()[0]}
    ckpt = {}
    tag = 0.0
    for model, p in plan.items():
        if model == "diffusion":
            p = {k: v for k, v in p.items() if k.startswith(("unet.encoders.1.", "time_embedding", "final"))}
        for dst, rule in p.items():
            shape = list(manifests[model][dst])
            n = len(rule["src"])
            for s in rule["src"]:
                shp = list(shape)
                if n == 3:
                    shp[0] //= 3
                if "shape" in rule:                       # VAE attention: 1x1 conv weights (C, C, 1, 1)
                    shp = [shp[0], shape[1], 1, 1]
                tag += 1.0
                ckpt[s] = torch.full(shp, tag)
    full = model_converter.conversion_plan()
    sub = {m: {k: v for k, v in full[m].items() if all(s in ckpt for s in v["src"])} for m in full}
    for model, p in sub.items():
        for dst, rule in p.items():
            srcs = [ckpt[k] for k in rule["src"]]
            t = srcs[0] if rule["op"].startswith("copy") else torch.cat(srcs, 0)
            if "shape" in rule:
                t = t.reshape(rule["shape"])
            assert tuple(t.shape) == tuple(manifests[model][dst]), (model, dst, t.shape)
    k = "unet.encoders.1.1.attention_1.in_proj.weight"
    r = full["diffusion"][k]
    t = torch.cat([ckpt[s] for s in r["src"]], 0)
    assert t[0, 0] < t[320, 0] < t[640, 0]          # rows ordered q | k | v (sd/model_converter.py:1009)


def test_load_from_standard_weights_executes_and_roundtrips(tmp_path):
    """SURVEY row f2, executed: a pickled ``{"state_dict": ...}`` checkpoint (every source key of the plan, the
    synthetic weights routed backwards through it) goes through the product's ``load_from_standard_weights``
    (torch.load + convert_state_dict: copy / cat / reshape, sd/model_converter.py:3-8,1009-1030) and must come back
    tensor-for-tensor, with the manifests' shapes, ready for the strict loads of sd/model_loader.py:26-42."""
    from pytorch_stable_diffusion_amd import model_loader
    plan = model_converter.conversion_plan()
    sds = model_loader.synthetic_state_dicts()
    ckpt = H.inverse_checkpoint(sds, plan)
    assert len(ckpt) == sum(len(r["src"]) for m in plan.values() for r in m.values())
    path = os.path.join(tmp_path, "fake-v1-5.ckpt")
    torch.save({"state_dict": ckpt, "global_step": 0}, path)
    del ckpt
    conv = model_converter.load_from_standard_weights(path, "cpu")
    manifests = {"diffusion": arch.diffusion_manifest(), "clip": arch.clip_manifest(),
                 "encoder": arch.vae_encoder_manifest()[0], "decoder": arch.vae_decoder_manifest()[0]}
    assert set(conv) == set(sds)
    for model, sd in sds.items():
        assert list(conv[model].keys()) == list(plan[model].keys())
        for k, v in sd.items():
            got = conv[model][k]
            assert tuple(got.shape) == tuple(manifests[model][k]), (model, k, got.shape)
            assert torch.equal(got, v), (model, k)
    # the strict loads the reference's loader performs (device-independent part)
    from pytorch_stable_diffusion_amd.clip import CLIP
    from pytorch_stable_diffusion_amd.diffusion import Diffusion
    from pytorch_stable_diffusion_amd.vae import VAE_Decoder, VAE_Encoder
    for cls, name in ((CLIP, "clip"), (VAE_Encoder, "encoder"), (VAE_Decoder, "decoder"), (Diffusion, "diffusion")):
        cls().load_state_dict(conv[name], strict=True)
    bad = dict(conv["clip"])
    bad.pop("layernorm.weight")
    import pytest
    with pytest.raises(RuntimeError):
        CLIP().load_state_dict(bad, strict=True)


def test_convert_state_dict_missing_source_key_raises():
    import pytest
    with pytest.raises(KeyError):
        model_converter.convert_state_dict({"model.diffusion_model.time_embed.0.weight": torch.zeros(1280, 320)})
