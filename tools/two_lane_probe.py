"""Probe: does running the two CFG halves as two concurrent batch-1 chains (two HIP streams, two host threads)
beat one batch-2 chain?  Kernel boundaries cost ~5 us each on MI355X; two independent chains overlap them."""
import sys, threading, time
import torch
sys.path.insert(0, ".")
from pytorch_stable_diffusion_amd import arch, synth, _native as N

dev = "cuda"
sd = synth.synth_state_dict(arch.diffusion_manifest(), torch.float16)
sd = {k: v.to(dev) for k, v in sd.items()}
hs = [N.UNetHandle(sd, N.FLAG_STREAM_F32) for _ in range(2)]
g = torch.Generator().manual_seed(0)
ctx = torch.randn((2, 77, 768), generator=g).to(dev)
temb = torch.randn((4, 320), generator=g).to(dev)
lat = torch.randn((1, 4, 64, 64), generator=g).to(dev)
hs[0].set_context(ctx); hs[0].set_schedule(temb)
for _ in range(3):
    hs[0].forward(lat, 2, step_idx=0)
torch.cuda.synchronize()
K = 40
t0 = time.perf_counter()
for _ in range(K):
    hs[0].forward(lat, 2, step_idx=0)
torch.cuda.synchronize()
t_b2 = (time.perf_counter() - t0) / K * 1e3
print(f"batch-2 single chain: {t_b2:.3f} ms/forward", flush=True)

streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for i in range(2):
    with torch.cuda.stream(streams[i]):
        hs[i].set_context(ctx[i:i + 1].contiguous()); hs[i].set_schedule(temb)
        for _ in range(3):
            hs[i].forward(lat, 1, step_idx=0)
torch.cuda.synchronize()
# one chain alone at batch 1
with torch.cuda.stream(streams[0]):
    t0 = time.perf_counter()
    for _ in range(K):
        hs[0].forward(lat, 1, step_idx=0)
    torch.cuda.synchronize()
print(f"batch-1 single chain: {(time.perf_counter() - t0) / K * 1e3:.3f} ms/forward", flush=True)

def work(i):
    with torch.cuda.stream(streams[i]):
        for _ in range(K):
            hs[i].forward(lat, 1, step_idx=0)
ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
t_host = (time.perf_counter() - t0) / K * 1e3
torch.cuda.synchronize()
t_2l = (time.perf_counter() - t0) / K * 1e3
print(f"two batch-1 chains concurrently: {t_2l:.3f} ms per pair (host enqueue {t_host:.3f})  -> x{t_b2 / t_2l:.2f}", flush=True)

# ---- same comparison with captured graphs (no host enqueue cost) ----
def capture(h, batch, stream):
    g = torch.cuda.CUDAGraph()
    out = torch.empty((batch, 4, 64, 64), device=dev)
    with torch.cuda.graph(g, stream=stream):
        h.forward(lat, batch, step_idx=0, out=out)
    return g, out
hs[0].set_context(ctx)
for _ in range(2):
    hs[0].forward(lat, 2, step_idx=0)
torch.cuda.synchronize()
g2, _ = capture(hs[0], 2, streams[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(streams[0]):
    for _ in range(K):
        g2.replay()
torch.cuda.synchronize()
t_g2 = (time.perf_counter() - t0) / K * 1e3
print(f"graph, batch-2 single chain: {t_g2:.3f} ms", flush=True)
for i in range(2):
    with torch.cuda.stream(streams[i]):
        hs[i].set_context(ctx[i:i + 1].contiguous())
        hs[i].forward(lat, 1, step_idx=0)
torch.cuda.synchronize()
gs = [capture(hs[i], 1, streams[i])[0] for i in range(2)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            gs[i].replay()
torch.cuda.synchronize()
t_gl = (time.perf_counter() - t0) / K * 1e3
print(f"graphs, two batch-1 chains on two streams: {t_gl:.3f} ms per pair -> x{t_g2 / t_gl:.2f}", flush=True)
