"""Name-keyed deterministic synthetic weights.

The real SD-v1.5 checkpoint is not available offline (reference: data/links.txt:1-6),
so parity fixtures, tests and the benchmark use weights that any process can
regenerate from the state-dict key alone: seed = CRC32(key); conv/linear weights and
biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (PyTorch's default-init scale);
norm gamma = 1 + 0.1*N(0,1), norm beta = 0.1*N(0,1).   

A second law, ``law="stress"``, keeps the same seeds but gives the tensors the statistics a trained checkpoint has and
the benign law never produces: every conv/linear OUTPUT channel gets its own scale, log-uniform over two decades
(0.1 .. 10) with 1 % of the channels (at least one from 64 channels up) another 30x larger, the whole vector normalised
to unit RMS so a layer's gain stays O(1); biases follow their row's scale; norm gamma is log-uniform in [0.2, 5] and
norm beta uniform in [-2, 2].  The residual stream then carries outlier channels and rows whose mean is many sigma --
the regime in which the load-time algebra of the native path (LayerNorm folded into the consumer GEMM, the composed
feed-forward, the folded cross-attention) loses precision first.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch


def _seed(key: str) -> int:
    return zlib.crc32(key.encode("utf-8")) & 0x7FFFFFFF


def _is_norm(key: str) -> bool:
    leaf = key.rsplit(".", 2)[-2] if key.count(".") >= 1 else key
    return leaf.startswith("groupnorm") or leaf.startswith("layernorm")


LAWS = ("benign", "stress")


def stress_row_scale(prefix: str, n: int) -> torch.Tensor:
    """Per-output-channel scale of the layer ``prefix`` under the stress law (shared by its weight and bias)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed(prefix + "#rowscale"))
    s = torch.exp((torch.rand(n, generator=g, dtype=torch.float32) * 2.0 - 1.0) * math.log(10.0))
    n_out = max(1, round(0.01 * n)) if n >= 64 else 0
    if n_out:
        idx = torch.randperm(n, generator=g)[:n_out]
        s[idx] *= 30.0
    return s / s.square().mean().sqrt()


def synth_tensor(key: str, shape: Tuple[int, ...], fan_in: int | None = None,
                 dtype=torch.float32, is_norm: bool | None = None, seed_key: str | None = None,
                 law: str = "benign") -> torch.Tensor:
    if law not in LAWS:
        raise ValueError(f"unknown weight law {law!r}")
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed(seed_key or key))
    if key.endswith("position_embedding"):
        return (torch.randn(shape, generator=g, dtype=torch.float32) * 0.02).to(dtype)
    if (_is_norm(key) if is_norm is None else is_norm) and law == "stress":
        u = torch.rand(shape, generator=g, dtype=torch.float32) * 2.0 - 1.0
        if key.endswith(".weight"):
            return torch.exp(u * math.log(5.0)).to(dtype)       # gamma log-uniform in [0.2, 5]
        return (u * 2.0).to(dtype)                              # beta uniform in [-2, 2]
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
    if law == "stress" and not key.endswith("embedding"):
        rs = stress_row_scale((seed_key or key).rsplit(".", 1)[0], shape[0])
        t = t * rs.reshape((-1,) + (1,) * (len(shape) - 1))
    return t.to(dtype)


def synth_state_dict(manifest: Dict[str, Tuple[int, ...]], dtype=torch.float32, norm_keys=(), seed_prefix: str = "",
                     law: str = "benign") -> "OrderedDict[str, torch.Tensor]":
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
        out[key] = synth_tensor(key, tuple(shape), fan_in, dtype, is_norm=norm, seed_key=seed_prefix + key, law=law)
    return out
