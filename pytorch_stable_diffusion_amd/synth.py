"""Name-keyed deterministic synthetic weights.

The real SD-v1.5 checkpoint is not available offline (reference: data/links.txt:1-6),
so parity fixtures, tests and the benchmark use weights that any process can
regenerate from the state-dict key alone: seed = CRC32(key); conv/linear weights and
biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (PyTorch's default-init scale);
norm gamma = 1 + 0.1*N(0,1), norm beta = 0.1*N(0,1).   
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch


def _seed(key: str) -> int:
    return zlib.crc32(key.encode("utf-8")) & 0x7FFFFFFF


def _is_norm(key: str) -> bool:
    leaf = key.rsplit(".", 2)[-2] if key.count(".") >= 1 else key
    return leaf.startswith("groupnorm") or leaf.startswith("layernorm")


def synth_tensor(key: str, shape: Tuple[int, ...], fan_in: int | None = None,
                 dtype=torch.float32, is_norm: bool | None = None, seed_key: str | None = None) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed(seed_key or key))
    if key.endswith("position_embedding"):
        return (torch.randn(shape, generator=g, dtype=torch.float32) * 0.02).to(dtype)
    if _is_norm(key) if is_norm is None else is_norm:
        t = torch.randn(shape, generator=g, dtype=torch.float32) * 0.1
        if key.endswith(".weight"):
            t += 1.0
        return t.to(dtype)
    if fan_in is None:
        if len(shape) < 2:
            raise ValueError(f"fan_in required for 1-D tensor {key}")
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
    bound = 1.0 / (fan_in ** 0.5)
    t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2.0 - 1.0) * bound
    return t.to(dtype)


def synth_state_dict(manifest: Dict[str, Tuple[int, ...]], dtype=torch.float32, norm_keys=(), seed_prefix: str = ""
                     ) -> "OrderedDict[str, torch.Tensor]":
    """Generate every tensor of ``manifest``.  A bias takes the fan_in of the ``.weight`` that shares
    its prefix.  ``norm_keys``: norm parameters whose names do not say so (nn.Sequential positions);
    ``seed_prefix`` disambiguates models whose keys collide (VAE encoder/decoder)."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape in manifest.items():
        fan_in = None
        norm = _is_norm(key) or key in norm_keys
        if len(shape) == 1 and not norm:
            wkey = key.rsplit(".", 1)[0] + ".weight"
            wshape = manifest[wkey]
            fan_in = 1
            for s in wshape[1:]:
                fan_in *= s
        out[key] = synth_tensor(key, tuple(shape), fan_in, dtype, is_norm=norm, seed_key=seed_prefix + key)
    return out
