"""Shared test helpers: seeded inputs and synthetic weights identical to make_golden.py."""
import json
import os

import numpy as np
import torch

from pytorch_stable_diffusion_amd import arch, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def seeded(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32) * scale


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def blocks_meta():
    with open(os.path.join(GOLDEN, "blocks_meta.json")) as f:
        return json.load(f)


def block_weights(prefix):
    man = arch.diffusion_manifest()
    sub = {k: v for k, v in man.items() if k.startswith(prefix + ".")}
    return synth.synth_state_dict(sub)


_FULL = {}


def full_weights():
    if "sd" not in _FULL:
        _FULL["sd"] = synth.synth_state_dict(arch.diffusion_manifest())
    return _FULL["sd"]


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))
