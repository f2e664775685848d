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
        q.put((rank, ok, mine, tmax, g_ok))
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


def test_flat_layout_alignment():
    man = arch.diffusion_manifest()
    lay = replicas.flat_layout(man)
    assert all(off % 8 == 0 for off, _ in lay.values())
    assert replicas.flat_size(man) >= arch.n_params(man)
    assert replicas.shard_prompts(["a", "b", "c"], 0, 1) == ["a", "b", "c"]
