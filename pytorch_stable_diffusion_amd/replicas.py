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


def timed_broadcast(flat: torch.Tensor, src: int = 0, group=None, device=None) -> Optional[Dict[str, float]]:
    """``broadcast_weights`` with its wall time: {"bytes", "seconds" (max over ranks), "GB_per_s"}; None without a group."""
    import time
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    on_gpu = flat.is_cuda
    dist.barrier(group)
    if on_gpu:
        torch.cuda.synchronize(flat.device)
    t0 = time.perf_counter()
    broadcast_weights(flat, src=src, group=group)
    if on_gpu:
        torch.cuda.synchronize(flat.device)
    dt = max_over_ranks(time.perf_counter() - t0, device=device if device is not None else (flat.device if on_gpu else None), group=group)
    nbytes = flat.numel() * flat.element_size()
    return {"bytes": nbytes, "seconds": round(dt, 4), "GB_per_s": round(nbytes / dt / 1e9, 2) if dt > 0 else None}


def group_facts(my_rate: float, device=None, group=None) -> Dict[str, object]:
    """What the process group itself reports: backend name, ranks counted by an all-reduce of 1, and every rank's own rate
    (all-gather).  Without a group: backend None, one rank."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {"backend": None, "ranks_seen": 1, "per_rank": [round(float(my_rate), 3)]}
    one = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM, group=group)
    mine = torch.tensor([my_rate], dtype=torch.float64, device=device)
    rates = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(rates, mine, group=group)
    return {"backend": dist.get_backend(group), "ranks_seen": int(round(one.item())),
            "per_rank": [round(float(r.item()), 3) for r in rates]}


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


# =====================================================================================================================
# generate()-level launcher: BASELINE config 4 ("batch of 8 independent prompts, 512x512, 50 steps, sharded 1-per-GPU")
#
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
#       -m pytorch_stable_diffusion_amd.replicas --prompts-file prompts.txt --ckpt v1-5-pruned-emaonly.ckpt \
#       --vocab vocab.json --merges merges.txt --out-dir out/
#
# One process per GPU.  Rank 0 reads + converts the checkpoint ONCE and hands the four state dicts (CLIP, VAE
# encoder/decoder, UNet) to every rank in one flat buffer per model over RCCL (the "trivial broadcast" north_star
# names); each rank then runs the reference-compatible pipeline.generate() (sd/pipeline.py:13-262, latent batch 1 per
# call: :146) on prompts i = rank, rank + N, ...; rank 0 gathers the uint8 images.  No collective inside the loop.
# =====================================================================================================================

def model_manifests() -> "OrderedDict[str, Dict[str, Tuple[int, ...]]]":
    from . import arch
    return OrderedDict([("clip", arch.clip_manifest()), ("encoder", arch.vae_encoder_manifest()[0]),
                        ("decoder", arch.vae_decoder_manifest()[0]), ("diffusion", arch.diffusion_manifest())])


def broadcast_state_dicts(state_dicts: Optional[Dict[str, Dict[str, torch.Tensor]]], manifests, device, src: int = 0,
                          dtype=torch.float32, group=None) -> Dict[str, "OrderedDict[str, torch.Tensor]"]:
    """Rank ``src`` passes its state dicts ({model: {key: tensor}}, any device); every rank gets views into ONE flat
    ``dtype`` buffer per model on ``device``.  fp32 keeps every rank bit-identical to a single-GPU load (4.3 GB in
    all: well under a second over xGMI); fp16 halves it at the cost of rounding the norm/bias vectors too."""
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    out = {}
    for name, man in manifests.items():
        flat = torch.empty(flat_size(man), dtype=dtype, device=device)
        if rank == src:
            if state_dicts is None or name not in state_dicts:
                raise ValueError(f"rank {src} has no state dict for '{name}'")
            pack_flat(state_dicts[name], man, flat)
        broadcast_weights(flat, src=src, group=group)
        out[name] = views_from_flat(flat, man)
    return out


def gather_image_lists(mine: List[torch.Tensor], n_total: int, shape, device=None, dst: int = 0, group=None
                       ) -> Optional[List[torch.Tensor]]:
    """Ranks hold the images of prompts rank, rank + N, ... (possibly one fewer on the higher ranks); ``dst`` gets
    the full list in prompt order.  One gather per round of N prompts."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(mine)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rounds = (n_total + world - 1) // world
    out: List[torch.Tensor] = []
    for r in range(rounds):
        img = mine[r] if r < len(mine) else torch.zeros(shape, dtype=torch.uint8)
        if device is not None:
            img = img.to(device)
        got = gather_images(img.contiguous(), dst=dst, group=group)
        if rank == dst:
            out.extend(g.cpu() for i, g in enumerate(got) if r * world + i < n_total)
    return out if rank == dst else None


class _Serialised:
    """A model shared by several lanes of one GPU (CLIP, the VAE): one caller at a time, and the caller's stream has drained
    before the next one may reuse the handle's activation arena."""

    def __init__(self, inner, lock):
        self._inner, self._lock = inner, lock

    def to(self, device):
        with self._lock:
            self._inner.to(device)
        return self

    def __call__(self, *args, **kw):
        with self._lock:
            out = self._inner(*args, **kw)
            if torch.is_tensor(out) and out.is_cuda:
                torch.cuda.current_stream(out.device).synchronize()
            return out


def lane_models(models: Dict[str, object], n_lanes: int, device=None) -> List[Dict[str, object]]:
    """``n_lanes`` model dicts for concurrent generate() loops on ONE GPU: every lane shares the CLIP / VAE objects of
    ``models`` (serialised: one caller at a time); lane 0 drives ``models["diffusion"]`` itself, the others a lane of it
    (``Diffusion.lanes()``: shared packed weights, own scratch).  Lanes are REUSED across calls -- each holds a 6 GiB arena,
    so a service loop that called this once per batch used to leak one arena per extra lane per call -- and
    ``models["diffusion"].release_lanes()`` frees them.  ``device``: the model is moved there first (a lane cannot move)."""
    import threading
    if n_lanes <= 1:
        return [models]
    lock = threading.Lock()
    shared = {k: _Serialised(v, lock) for k, v in models.items() if k != "diffusion"}
    unet = models["diffusion"]
    if device is not None and hasattr(unet, "to"):
        unet.to(device)
    units = unet.lanes(n_lanes) if hasattr(unet, "lanes") else [unet] * n_lanes
    out = []
    for i in range(n_lanes):
        d = dict(shared)
        d["diffusion"] = units[i]
        out.append(d)
    return out


def run_prompts(prompts: Sequence[str], models: Dict[str, object], tokenizer, device, *, seed_base: int = 0,
                uncond_prompt: str = "", n_inference_steps: int = 50, cfg_scale: float = 7.5, height: int = 512,
                width: int = 512, input_images: Optional[Sequence] = None, strength: float = 0.8, group=None,
                gather_device=None, generate=None, streams_per_gpu: int = 1, batch_per_gpu: int = 1, generate_batch=None):
    """Shard ``prompts`` over the ranks of the initialised process group (prompt i -> rank i mod N, seed =
    seed_base + i), run generate() per prompt, gather on rank 0.  Returns (images or None, stats).

    ``streams_per_gpu`` > 1: throughput mode -- this rank's prompts are dealt to that many LANES, each a thread running
    generate() on its own HIP stream over the shared packed weights (``lane_models``).  Every image is the same as in the
    one-lane run (same seed, same kernels, same plans); only the wall time per BATCH of prompts changes.

    ``batch_per_gpu`` > 1 (txt2img only): this rank's prompts go through ``pipeline.generate_batch`` in groups of that many
    -- ONE chain of launches at UNet batch 2 x group, the weights streamed once per step for the whole group.  Prompt i keeps
    its seed ``seed_base + i`` and its own noise stream.

    Both together: ``streams_per_gpu`` lanes, each carrying groups of ``batch_per_gpu`` prompts -- the batched chains amortise
    the weight traffic and the launch-bound levels, the lanes fill what latency is left (measured on one MI355X, DESIGN.md 6:
    2 lanes x 6 prompts 520-535 UNet steps/s in aggregate beside 254-264 for one chain, one batched chain of 8 474-486; round 5).  The
    activation arena of each lane grows by itself for the batch it is given (csrc/engine.h ensure_arena): no environment knob
    is needed for groups beyond four prompts.

    stats also carries ``ln_guard_hits`` / ``ln_guard_fallbacks``: rows beyond the LayerNorm-fold guard and how many groups
    repeated their loop unfused because of them (each such group took twice the time)."""
    import threading
    import time

    import torch.distributed as dist
    if generate is None:
        from .pipeline import generate
    batch_per_gpu = max(1, int(batch_per_gpu))
    if batch_per_gpu > 1:
        if input_images is not None:
            raise ValueError("run_prompts: batch_per_gpu > 1 is txt2img only")
        if generate_batch is None:
            from .pipeline import generate_batch
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    todo = shard_prompts(list(enumerate(prompts)), rank, world)
    on_gpu = torch.cuda.is_available() and torch.device(device).type == "cuda"
    # this rank's prompts in groups of batch_per_gpu (one batched chain each), the groups dealt to the lanes
    groups = [todo[g0:g0 + batch_per_gpu] for g0 in range(0, len(todo), batch_per_gpu)]
    n_lanes = max(1, min(int(streams_per_gpu), len(groups))) if groups else 1
    lanes = lane_models(models, n_lanes, device=device)
    results: Dict[int, torch.Tensor] = {}
    per_image: Dict[int, float] = {}
    guard = {"hits": 0, "fallbacks": 0}          # LayerNorm-fold guard of this rank's loops (a fallback doubles a group's latency)
    errors: List[BaseException] = []

    def work(lane: int):
        stream = torch.cuda.Stream(device=device) if (on_gpu and n_lanes > 1) else None
        try:
            for grp in groups[lane::n_lanes]:
                t1 = time.perf_counter()
                ctx = torch.cuda.stream(stream) if stream is not None else _null_ctx()
                with ctx:
                    if batch_per_gpu > 1:
                        imgs = generate_batch(prompts=[p for _, p in grp], uncond_prompt=uncond_prompt, seeds=[seed_base + i for i, _ in grp],
                                              do_cfg=True, cfg_scale=cfg_scale, sampler_name="ddpm", n_inference_steps=n_inference_steps,
                                              models=lanes[lane], device=device, idle_device=None, tokenizer=tokenizer, height=height,
                                              width=width)
                    else:
                        i, prompt = grp[0]
                        imgs = [generate(prompt=prompt, uncond_prompt=uncond_prompt,
                                         input_image=None if input_images is None else input_images[i], strength=strength,
                                         do_cfg=True, cfg_scale=cfg_scale, sampler_name="ddpm", n_inference_steps=n_inference_steps,
                                         models=lanes[lane], seed=seed_base + i, device=device, idle_device=None,
                                         tokenizer=tokenizer, height=height, width=width)]
                dt = (time.perf_counter() - t1) / len(grp)
                unet = lanes[lane].get("diffusion")
                if getattr(unet, "ln_guard_fallback_ran", False):      # this group paid for its loop twice (Diffusion.denoise_native)
                    guard["fallbacks"] += 1
                guard["hits"] += int(getattr(unet, "ln_guard_hits", 0) or 0)
                for (i, _), im in zip(grp, imgs):
                    results[i] = torch.from_numpy(im)
                    per_image[i] = dt
        except BaseException as exc:      # re-raised on the caller's thread
            errors.append(exc)

    if dist.is_initialized():
        dist.barrier(group)
    t0 = time.perf_counter()
    if n_lanes == 1:
        work(0)
    else:
        threads = [threading.Thread(target=work, args=(k,), name=f"sdmi-lane-{k}") for k in range(n_lanes)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    if on_gpu:
        torch.cuda.synchronize(device)
    mine = [results[i] for i, _ in todo]
    elapsed = max_over_ranks(time.perf_counter() - t0, device=gather_device, group=group)
    images = gather_image_lists(mine, len(prompts), (height, width, 3), device=gather_device, group=group)
    stats = {"n_prompts": len(prompts), "world": world, "streams_per_gpu": n_lanes, "batch_per_gpu": batch_per_gpu, "elapsed_s": elapsed,
             "images_per_s": len(prompts) / elapsed if elapsed > 0 else 0.0,
             "rank0_s_per_image": [per_image[i] for i, _ in todo],
             "ln_guard_hits": guard["hits"], "ln_guard_fallbacks": guard["fallbacks"]}
    return images, stats


class _null_ctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _main(argv=None, hooks: Optional[Dict[str, object]] = None) -> int:
    """The launcher's entry point.  ``hooks`` (tests only: the world-8 gloo test drives THIS function on CPU with stub models)
    may replace what needs a GPU or a checkpoint: "device", "backend", "manifests", "state_dicts" (callable, rank 0 only),
    "make_models" (state dicts -> models), "generate", "tokenizer", "emit" (callable(record dict, images) on rank 0)."""
    import argparse
    import json
    import os
    hooks = hooks or {}

    ap = argparse.ArgumentParser(prog="python -m pytorch_stable_diffusion_amd.replicas",
                                 description="prompt-parallel generate() over the GPUs of one node (one process per GPU)")
    ap.add_argument("--prompts-file", required=True, help="one prompt per line")
    ap.add_argument("--ckpt", help="standard SD-v1.x checkpoint (.ckpt); read and converted on rank 0 only")
    ap.add_argument("--synthetic", action="store_true", help="name-keyed synthetic weights instead of a checkpoint")
    ap.add_argument("--vocab", help="CLIP vocab.json")
    ap.add_argument("--merges", help="CLIP merges.txt")
    ap.add_argument("--stub-tokenizer", action="store_true", help="hash tokenizer (no vocab files offline)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--cfg-scale", type=float, default=7.5)
    ap.add_argument("--seed-base", type=int, default=0)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--out-dir", default=None)
    ap.add_argument("--streams-per-gpu", type=int, default=1,
                    help="throughput mode: this many concurrent generate() lanes per GPU over one copy of the packed weights")
    ap.add_argument("--batch-per-gpu", type=int, default=1,
                    help="throughput mode: groups of this many prompts per GPU go through ONE batched denoising loop (txt2img); with "
                         "--streams-per-gpu C: C lanes of such groups (2 x 6 measured fastest on one MI355X)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) on GPUs")
    args = ap.parse_args(argv)
    if bool(args.ckpt) == bool(args.synthetic):
        ap.error("give exactly one of --ckpt / --synthetic")

    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = hooks.get("backend", args.backend)
    if "device" in hooks:
        dev = torch.device(hooks["device"])
    else:
        if not torch.cuda.is_available():
            raise SystemExit("replicas: needs GPUs (the HIP path has no CPU fallback)")
        # rendezvous first, on this rank's own device; nothing below re-execs the process
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    with open(args.prompts_file) as f:
        prompts = [ln.rstrip("\n") for ln in f if ln.strip()]
    if not prompts:
        raise SystemExit("replicas: empty prompts file")
    if "tokenizer" in hooks:
        tokenizer = hooks["tokenizer"]
    elif args.stub_tokenizer:
        from .tokenizer import StubTokenizer
        tokenizer = StubTokenizer()
    else:
        if not (args.vocab and args.merges):
            ap.error("--vocab and --merges are required unless --stub-tokenizer")
        from .tokenizer import CLIPTokenizer
        tokenizer = CLIPTokenizer(args.vocab, merges_file=args.merges)

    sds = None
    if rank == 0:                                   # the checkpoint is read and converted exactly once
        if "state_dicts" in hooks:
            sds = hooks["state_dicts"]()
        else:
            from . import model_converter, model_loader
            sds = (model_loader.synthetic_state_dicts() if args.synthetic
                   else model_converter.load_from_standard_weights(args.ckpt, "cpu"))
    manifests = hooks.get("manifests") or model_manifests()
    # the ONE collective of the path: timed once, between barriers, with what the group itself reports
    import time
    if world > 1:
        dist.barrier()
    t_b = time.perf_counter()
    state = broadcast_state_dicts(sds, manifests, dev)
    bcast_s = max_over_ranks(time.perf_counter() - t_b, device=dev if dev.type == "cuda" else None)
    bcast_bytes = sum(flat_size(m) for m in manifests.values()) * 4
    del sds
    if "make_models" in hooks:
        models = hooks["make_models"](state)
    else:
        from . import model_loader
        models = model_loader.preload_models_from_state_dicts(state, dev)

    gdev = dev if dev.type == "cuda" else None
    images, stats = run_prompts(prompts, models, tokenizer, dev, seed_base=args.seed_base, n_inference_steps=args.steps,
                                cfg_scale=args.cfg_scale, height=args.height, width=args.width, gather_device=gdev,
                                streams_per_gpu=args.streams_per_gpu, batch_per_gpu=args.batch_per_gpu,
                                generate=hooks.get("generate"))
    # the launch checks itself (as bench.py's line does): backend, ranks counted by an all-reduce, every rank's own rate
    n_mine = len(shard_prompts(prompts, rank, world))
    facts = group_facts(n_mine / stats["elapsed_s"] if stats["elapsed_s"] > 0 else 0.0, device=gdev)
    if rank == 0:
        if args.out_dir:
            from PIL import Image
            os.makedirs(args.out_dir, exist_ok=True)
            for i, im in enumerate(images):
                Image.fromarray(im.numpy()).save(os.path.join(args.out_dir, f"image_{i:04d}.png"))
        rec = {"metric": "images_per_s", "value": round(stats["images_per_s"], 4), "n_gpus": world,
               "n_prompts": len(prompts), "steps": args.steps, "elapsed_s": round(stats["elapsed_s"], 3),
               "dist_backend": facts["backend"], "ranks_seen": facts["ranks_seen"], "per_rank_images_per_s": facts["per_rank"],
               "weight_broadcast": None if world == 1 else {"bytes": bcast_bytes, "seconds": round(bcast_s, 4),
                                                             "GB_per_s": round(bcast_bytes / bcast_s / 1e9, 2) if bcast_s > 0 else None},
               "ln_guard_hits": stats.get("ln_guard_hits", 0), "ln_guard_fallbacks": stats.get("ln_guard_fallbacks", 0)}
        if "emit" in hooks:
            hooks["emit"](rec, images)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(_main())
