"""Prompt-parallel replicas: the only multi-GPU structure the path has.

Every ``generate()`` call is independent (one prompt, one seed, latent batch 1 --
sd/pipeline.py:146), so N GPUs run N replicas with NO collective inside the sampling loop.  The one
exchange is the start-up weight hand-out: rank 0 reads/packs the checkpoint once and broadcasts a
flat fp16 buffer (1.72 GB for the UNet) over RCCL/xGMI; afterwards ranks only meet for timing
reductions or to gather result images.  ``torch.distributed`` (backend "nccl" = RCCL on ROCm, "gloo"
in CPU tests) is plumbing here; nothing in this file touches the kernels.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch


def numel(shape: Sequence[int]) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n


def flat_layout(manifest: Dict[str, Tuple[int, ...]]) -> "OrderedDict[str, Tuple[int, int]]":
    """key -> (offset, numel) in the flat weight buffer; offsets are 8-element (16 B) aligned."""
    out: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
    off = 0
    for k, shp in manifest.items():
        n = numel(shp)
        out[k] = (off, n)
        off += (n + 7) // 8 * 8
    return out


def flat_size(manifest) -> int:
    lay = flat_layout(manifest)
    last_off, last_n = next(reversed(lay.values()))
    return last_off + (last_n + 7) // 8 * 8


def pack_flat(state: Dict[str, torch.Tensor], manifest, flat: torch.Tensor) -> None:
    """Copy (and cast) every tensor of ``state`` into ``flat`` (rank 0 only)."""
    for k, (off, n) in flat_layout(manifest).items():
        flat[off:off + n].copy_(state[k].reshape(-1).to(flat.dtype))


def views_from_flat(flat: torch.Tensor, manifest) -> "OrderedDict[str, torch.Tensor]":
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, (off, n) in flat_layout(manifest).items():
        out[k] = flat[off:off + n].view(*manifest[k])
    return out


def broadcast_weights(flat: torch.Tensor, src: int = 0, group=None, chunk_elems: int = 1 << 28) -> None:
    """One-off weight broadcast rank ``src`` -> all.  Chunked (512 MiB of fp16 per call) so a single
    collective never needs more staging than a link can stream comfortably."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    n = flat.numel()
    for s in range(0, n, chunk_elems):
        dist.broadcast(flat[s:min(n, s + chunk_elems)], src=src, group=group)


def shard_prompts(items: Sequence, rank: int, world: int) -> List:
    """Prompt/seed i -> rank i mod world."""
    return [it for i, it in enumerate(items) if i % world == rank]


def max_over_ranks(value: float, device=None, group=None) -> float:
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def gather_images(img: torch.Tensor, dst: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """Collect the per-rank uint8 images (H,W,3) on ``dst`` (786 KB each at 512x512)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [img]
    world = dist.get_world_size(group)
    bufs = [torch.empty_like(img) for _ in range(world)] if dist.get_rank(group) == dst else None
    dist.gather(img, bufs, dst=dst, group=group)
    return bufs
