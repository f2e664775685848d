"""N>1 path on CPU: world_size-2 gloo processes exercise the replica plumbing (weight broadcast from
rank 0, prompt sharding, max-over-ranks timing, image gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pytorch_stable_diffusion_amd import arch, replicas, synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        man = {k: v for k, v in arch.diffusion_manifest().items() if k.startswith("unet.encoders.1.")}
        flat = torch.zeros(replicas.flat_size(man), dtype=torch.float16)
        if rank == 0:
            replicas.pack_flat(synth.synth_state_dict(man), man, flat)
        replicas.broadcast_weights(flat, src=0, chunk_elems=1 << 16)     # several chunks
        views = replicas.views_from_flat(flat, man)
        want = synth.synth_state_dict(man)
        ok = all(torch.equal(views[k], want[k].to(torch.float16)) for k in man)
        mine = replicas.shard_prompts(list(range(7)), rank, world)
        tmax = replicas.max_over_ranks(1.0 + rank)
        img = torch.full((4, 4, 3), rank, dtype=torch.uint8)
        gathered = replicas.gather_images(img, dst=0)
        g_ok = True
        if rank == 0:
            g_ok = all(int(g[0, 0, 0]) == i for i, g in enumerate(gathered))
        # the bench line's self-check fields (bench.py: dist_backend / ranks_seen / per_rank_steps_per_s / weight_broadcast)
        facts = replicas.group_facts(10.0 + rank)
        tb = replicas.timed_broadcast(flat, src=0)
        q.put((rank, ok, mine, tmax, g_ok, facts, tb))
    finally:
        dist.destroy_process_group()


def test_replica_plumbing_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], "broadcast weights differ from the rank-0 source"
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5]
    assert res[0][3] == res[1][3] == 2.0
    assert res[0][4]
    for r in res:
        assert r[5] == {"backend": "gloo", "ranks_seen": 2, "per_rank": [10.0, 11.0]}
        assert r[6]["bytes"] > 0 and r[6]["seconds"] > 0 and r[6]["GB_per_s"] is not None
    assert res[0][6] == res[1][6]                       # max-over-ranks time: every rank reports the same line
    assert replicas.group_facts(3.0) == {"backend": None, "ranks_seen": 1, "per_rank": [3.0]}     # no group: one rank
    assert replicas.timed_broadcast(torch.zeros(8)) is None


def test_flat_layout_alignment():
    man = arch.diffusion_manifest()
    lay = replicas.flat_layout(man)
    assert all(off % 8 == 0 for off, _ in lay.values())
    assert replicas.flat_size(man) >= arch.n_params(man)
    assert replicas.shard_prompts(["a", "b", "c"], 0, 1) == ["a", "b", "c"]


# ---- generate()-level launcher (BASELINE config 4) on CPU: world-2 gloo, stub models -------------------------------
class _Stub:
    def __init__(self, sd):
        self.sd = sd

    def to(self, device):
        return self


class _StubClip(_Stub):
    def __call__(self, tokens):                       # (1,77) -> (1,77,768)
        return self.sd["emb"][tokens % self.sd["emb"].shape[0]]


class _StubUNet(_Stub):
    def __call__(self, latent, context, time):        # reference convention model(latent, context, time)
        return self.sd["mix"][0] * latent + self.sd["mix"][1] * context.mean(dim=(1, 2)).view(-1, 1, 1, 1) + 0.01 * time.mean()


class _StubDecoder(_Stub):
    def __call__(self, latents):                      # (1,4,h,w) -> (1,3,8h,8w)
        up = torch.nn.functional.interpolate(latents[:, :3], scale_factor=8, mode="nearest")
        return torch.tanh(up * self.sd["gain"][0])


_STUB_MANIFESTS = {"clip": {"emb": (101, 768)}, "diffusion": {"mix": (2,)}, "decoder": {"gain": (1,)}}
_PROMPTS = ["a dog", "a cat on a mat", "two birds", "a red car", "the sea at night"]


def _stub_weights():
    g = torch.Generator().manual_seed(9)
    return {"clip": {"emb": torch.randn((101, 768), generator=g)}, "diffusion": {"mix": torch.tensor([0.05, 0.3])},
            "decoder": {"gain": torch.tensor([0.7])}}


def _stub_models(sds):
    return {"clip": _StubClip(sds["clip"]), "diffusion": _StubUNet(sds["diffusion"]), "decoder": _StubDecoder(sds["decoder"])}


def _stub_generate(prompt, uncond_prompt=None, models=None, seed=None, tokenizer=None, n_inference_steps=3, height=64,
                   width=64, cfg_scale=7.5, **_):
    """Same call surface as pipeline.generate (the product's sampler step is a HIP kernel: no CPU path), reduced to
    what the launcher depends on: the image is a function of (prompt, seed, every model's weights)."""
    gen = torch.Generator().manual_seed(seed)
    ids = lambda t: torch.tensor(tokenizer.batch_encode_plus([t], padding="max_length", max_length=77).input_ids)
    ctx = torch.cat([models["clip"](ids(prompt)), models["clip"](ids(uncond_prompt))])
    lat = torch.randn((1, 4, height // 8, width // 8), generator=gen)
    for t in range(n_inference_steps):
        c, u = models["diffusion"](lat.repeat(2, 1, 1, 1), ctx, torch.full((1, 320), float(t))).chunk(2)
        lat = lat - 0.1 * (cfg_scale * (c - u) + u)
    img = models["decoder"](lat)
    return ((img.clamp(-1, 1) + 1) * 127.5).permute(0, 2, 3, 1).to(torch.uint8).numpy()[0]


def _gen_worker(rank, world, port, q):
    from tests.stub_tokenizer import StubTokenizer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sds = replicas.broadcast_state_dicts(_stub_weights() if rank == 0 else None, _STUB_MANIFESTS, "cpu")
        imgs, stats = replicas.run_prompts(_PROMPTS, _stub_models(sds), StubTokenizer(), "cpu", seed_base=100,
                                           n_inference_steps=3, height=64, width=64, generate=_stub_generate)
        q.put((rank, None if imgs is None else [i.numpy() for i in imgs], stats["n_prompts"], stats["elapsed_s"]))
    finally:
        dist.destroy_process_group()


def test_generate_level_launcher_world2_matches_single_process():
    """run_prompts over two gloo ranks (weights handed out by broadcast_state_dicts, prompts i -> rank i mod 2,
    seeds seed_base + i, images gathered in prompt order on rank 0) == the same prompts in one process."""
    from tests.stub_tokenizer import StubTokenizer
    models = _stub_models(_stub_weights())
    want = [_stub_generate(prompt=p, uncond_prompt="", models=models, seed=100 + i, tokenizer=StubTokenizer())
            for i, p in enumerate(_PROMPTS)]
    assert len({w.tobytes() for w in want}) == len(want)          # prompts/seeds give distinct images
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gen_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1][1] is None and res[0][2] == 5
    assert res[0][3] == res[1][3]                                  # max-over-ranks elapsed agrees
    got = res[0][1]
    assert len(got) == 5
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.shape == (64, 64, 3) and (g == w).all(), f"prompt {i}"


def test_streams_per_gpu_lanes_keep_prompt_order_and_images():
    """run_prompts(streams_per_gpu=2): the rank's prompts are dealt to two lanes (threads); the images come back in prompt
    order and equal the one-lane run (CPU stub models: the lane plumbing, no GPU)."""
    from tests.stub_tokenizer import StubTokenizer
    models = _stub_models(_stub_weights())
    one, s1 = replicas.run_prompts(_PROMPTS, models, StubTokenizer(), "cpu", seed_base=100, n_inference_steps=3, height=64,
                                   width=64, generate=_stub_generate)
    two, s2 = replicas.run_prompts(_PROMPTS, models, StubTokenizer(), "cpu", seed_base=100, n_inference_steps=3, height=64,
                                   width=64, generate=_stub_generate, streams_per_gpu=2)
    assert s1["streams_per_gpu"] == 1 and s2["streams_per_gpu"] == 2
    assert len(one) == len(two) == len(_PROMPTS)
    for a, b in zip(one, two):
        assert torch.equal(a, b)
    lanes = replicas.lane_models(models, 3)
    assert len(lanes) == 3 and lanes[0]["diffusion"] is models["diffusion"]
    assert all(isinstance(l["clip"], replicas._Serialised) and isinstance(l["decoder"], replicas._Serialised) for l in lanes)
    assert replicas.lane_models(models, 1) == [models]


def test_lane_error_is_raised_on_the_caller():
    from tests.stub_tokenizer import StubTokenizer

    def boom(**kw):
        raise ValueError("lane failure")

    with pytest.raises(ValueError, match="lane failure"):
        replicas.run_prompts(_PROMPTS, _stub_models(_stub_weights()), StubTokenizer(), "cpu", generate=boom, streams_per_gpu=2)


def test_batch_per_gpu_groups_keep_prompt_order_and_seeds():
    """run_prompts(batch_per_gpu=2): the rank's prompts go to generate_batch in groups of two (the last group may be smaller)
    with their own seeds seed_base + i; the images come back in prompt order and equal the one-by-one run (CPU stub: the grouping
    and ordering plumbing; the batched HIP loop has its own GPU test)."""
    from tests.stub_tokenizer import StubTokenizer
    models = _stub_models(_stub_weights())
    calls = []

    def stub_batch(prompts, uncond_prompt="", seeds=None, models=None, tokenizer=None, n_inference_steps=3, height=64, width=64,
                   cfg_scale=7.5, **_):
        calls.append((list(prompts), list(seeds)))
        return [_stub_generate(prompt=p, uncond_prompt=uncond_prompt, models=models, seed=s, tokenizer=tokenizer,
                               n_inference_steps=n_inference_steps, height=height, width=width, cfg_scale=cfg_scale)
                for p, s in zip(prompts, seeds)]

    one, _ = replicas.run_prompts(_PROMPTS, models, StubTokenizer(), "cpu", seed_base=100, n_inference_steps=3, height=64, width=64,
                                  generate=_stub_generate)
    two, st = replicas.run_prompts(_PROMPTS, models, StubTokenizer(), "cpu", seed_base=100, n_inference_steps=3, height=64, width=64,
                                   generate=_stub_generate, generate_batch=stub_batch, batch_per_gpu=2)
    assert st["batch_per_gpu"] == 2 and st["streams_per_gpu"] == 1
    assert calls == [(_PROMPTS[0:2], [100, 101]), (_PROMPTS[2:4], [102, 103]), (_PROMPTS[4:5], [104])]
    assert len(one) == len(two) == len(_PROMPTS)
    for a, b in zip(one, two):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        replicas.run_prompts(_PROMPTS, models, StubTokenizer(), "cpu", generate_batch=stub_batch, batch_per_gpu=2, input_images=[None] * 5)


def test_lanes_of_batched_groups_keep_prompt_order_and_seeds():
    """run_prompts(streams_per_gpu=2, batch_per_gpu=2): the rank's prompts form groups of two (one batched chain each) and the
    groups are dealt to two lanes; every prompt is generated once, with its own seed, by the lane its group belongs to, and the
    images come back in prompt order."""
    import threading
    from tests.stub_tokenizer import StubTokenizer
    models = _stub_models(_stub_weights())
    calls, lock = [], threading.Lock()

    def stub_batch(prompts, uncond_prompt="", seeds=None, models=None, tokenizer=None, n_inference_steps=3, height=64, width=64,
                   cfg_scale=7.5, **_):
        with lock:
            calls.append((list(prompts), list(seeds), threading.current_thread().name))
        return [_stub_generate(prompt=p, uncond_prompt=uncond_prompt, models=models, seed=s, tokenizer=tokenizer,
                               n_inference_steps=n_inference_steps, height=height, width=width, cfg_scale=cfg_scale)
                for p, s in zip(prompts, seeds)]

    one, _ = replicas.run_prompts(_PROMPTS, models, StubTokenizer(), "cpu", seed_base=100, n_inference_steps=3, height=64, width=64,
                                  generate=_stub_generate)
    two, st = replicas.run_prompts(_PROMPTS, models, StubTokenizer(), "cpu", seed_base=100, n_inference_steps=3, height=64, width=64,
                                   generate=_stub_generate, generate_batch=stub_batch, batch_per_gpu=2, streams_per_gpu=2)
    assert st["batch_per_gpu"] == 2 and st["streams_per_gpu"] == 2
    by_group = {tuple(s): (p, t) for p, s, t in calls}
    assert sorted(by_group) == [(100, 101), (102, 103), (104,)]
    assert by_group[(100, 101)][0] == _PROMPTS[0:2] and by_group[(104,)][0] == _PROMPTS[4:5]
    assert by_group[(100, 101)][1] == by_group[(104,)][1] == "sdmi-lane-0" and by_group[(102, 103)][1] == "sdmi-lane-1"
    for a, b in zip(one, two):
        assert torch.equal(a, b)


# ---- BASELINE configs[3]'s real shape on CPU: EIGHT ranks through the launcher's own entry point ---------------------------
def _main_worker(rank, world, port, prompts_file, q):
    """One rank of `python -m pytorch_stable_diffusion_amd.replicas` as torch.distributed.run would start it (RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_* in the environment), with the hooks that stand in for the GPU: gloo, CPU, stub models."""
    from tests.stub_tokenizer import StubTokenizer
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank)})
    seen = []

    def gen(**kw):                                  # records which (prompt, seed) THIS rank generated
        seen.append((kw["prompt"], kw["seed"]))
        kw.pop("n_inference_steps", None), kw.pop("height", None), kw.pop("width", None)
        return _stub_generate(n_inference_steps=3, height=64, width=64, **kw)

    out = {}
    hooks = {"device": "cpu", "backend": "gloo", "manifests": _STUB_MANIFESTS, "state_dicts": _stub_weights,
             "make_models": _stub_models, "generate": gen, "tokenizer": StubTokenizer(),
             "emit": lambda rec, images: out.update(rec=rec, images=[i.numpy() for i in images])}
    rc = replicas._main(["--prompts-file", prompts_file, "--synthetic", "--stub-tokenizer", "--steps", "3", "--seed-base", "100",
                         "--height", "64", "--width", "64"], hooks=hooks)
    q.put((rank, rc, seen, out.get("rec"), out.get("images")))


@pytest.mark.parametrize("n_prompts", [8, 11])
def test_launcher_world8_one_prompt_per_rank(tmp_path, n_prompts):
    """BASELINE configs[3]: 8 independent prompts over 8 ranks, one each (and 11 prompts: three ranks take a second one).
    Through replicas._main itself: rank 0 alone makes the weights, ONE broadcast hands them out, prompt i runs on rank i mod 8
    with seed seed_base + i, rank 0 gathers the images in prompt order, and the record it prints checks the group
    (ranks_seen == 8, eight per-rank rates, the broadcast timed once)."""
    from tests.stub_tokenizer import StubTokenizer
    world = 8
    prompts = [f"prompt number {i} of the batch" for i in range(n_prompts)]
    pf = tmp_path / "prompts.txt"
    pf.write_text("\n".join(prompts) + "\n")
    models = _stub_models(_stub_weights())
    want = [_stub_generate(prompt=p, uncond_prompt="", models=models, seed=100 + i, tokenizer=StubTokenizer())
            for i, p in enumerate(prompts)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_main_worker, args=(r, world, port, str(pf), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rc, seen, rec, images in res:
        assert rc == 0
        assert seen == [(prompts[i], 100 + i) for i in range(rank, n_prompts, world)], f"rank {rank} generated {seen}"
        assert (rec is None) == (rank != 0)                      # only rank 0 emits the record and holds the images
    rec, images = res[0][3], res[0][4]
    assert rec["n_gpus"] == 8 and rec["n_prompts"] == n_prompts and rec["dist_backend"] == "gloo"
    assert rec["ranks_seen"] == 8 and len(rec["per_rank_images_per_s"]) == 8 and all(r > 0 for r in rec["per_rank_images_per_s"])
    wb = rec["weight_broadcast"]
    assert wb["bytes"] == 4 * sum(replicas.flat_size(m) for m in _STUB_MANIFESTS.values()) and wb["seconds"] > 0
    assert len(images) == n_prompts
    for i, (g, w) in enumerate(zip(images, want)):
        assert g.shape == (64, 64, 3) and (g == w).all(), f"prompt {i} came back out of order or from the wrong seed"
