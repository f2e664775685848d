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


def stress_input(shape, seed):
    """NCHW block input for the stress-law fixtures (tests/golden/make_golden_stress.py): N(0,1) times a per-channel scale
    spread over two decades with 1 % outlier channels (synth.stress_row_scale, x3), plus a per-channel offset of up to
    +-2 scales -- so token rows carry outlier channels and a mean of several sigma."""
    C = shape[1]
    cs = synth.stress_row_scale(f"input#{seed}", C) * 3.0
    g = torch.Generator(device="cpu").manual_seed(seed + 100000)
    co = (torch.rand(C, generator=g) * 4.0 - 2.0) * cs
    return seeded(shape, seed) * cs.view(1, C, 1, 1) + co.view(1, C, 1, 1)


def stress_block_weights(prefix):
    man = arch.diffusion_manifest()
    sub = {k: v for k, v in man.items() if k.startswith(prefix + ".")}
    return synth.synth_state_dict(sub, law="stress")


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


def inverse_checkpoint(state_dicts, plan):
    """Route converted state dicts ({'clip','encoder','decoder','diffusion'}) BACKWARDS through the converter's plan
    into the source layout of a standard SD-v1.x checkpoint: 'copy' -> the tensor itself (a '+reshape' rule's source
    is the 1x1-conv form (O, I, 1, 1), sd/model_converter.py:1026-1030), 'cat' -> equal row chunks in source order
    (q | k | v, sd/model_converter.py:1009).  Returns the checkpoint's ``state_dict``."""
    ckpt = {}
    for model, rules in plan.items():
        for dst, rule in rules.items():
            t = state_dicts[model][dst]
            parts = [t] if rule["op"].startswith("copy") else list(t.chunk(len(rule["src"]), 0))
            for k, p in zip(rule["src"], parts):
                if "shape" in rule:
                    p = p.reshape(p.shape[0], p.shape[1], 1, 1)
                assert k not in ckpt, k
                ckpt[k] = p.contiguous()
    return ckpt


def float_image_u16(x):
    """Decoder output (float, nominally [-1,1]) -> 16-bit fixed point of the clamped image: what the uint8 cast sees, at
    1.5e-5 of the range per count.  Full-resolution float goldens are stored this way (1.5 MB per 512x512 image)."""
    import numpy as np
    y = (x.detach().float().cpu().clamp(-1, 1) + 1) * 0.5 * 65535.0
    return y.round().numpy().astype(np.uint16)


def float_image_mae(x, ref_u16):
    """Mean |x - ref| on the [0,1] scale, x a float image in [-1,1] (clamped like the reference's rescale), ref from float_image_u16."""
    import numpy as np
    y = (x.detach().float().cpu().clamp(-1, 1) + 1) * 0.5
    return float((y.double() - torch.from_numpy(ref_u16.astype(np.float64) / 65535.0)).abs().mean())
