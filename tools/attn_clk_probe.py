#!/usr/bin/env python3
"""In-kernel clock stamps of the S = 4096 self-attention (diagnostic build: tools/build_variant.sh probe attention -DSDMI_ATTN_PROBE;
SDMI_LIB=pytorch_stable_diffusion_amd/lib/variants/libsdmi_probe.so): shader-clock cycles per 64-key tile a wave spends in each part
of its tile loop, and the shader clock over the wave's life (profiles/r05_attn_clk_probe.txt)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytorch_stable_diffusion_amd import _native as N

lib = N.load()
dev = "cuda"
d, Sq, Skv, B, H = 40, 4096, 4096, 2, 8
Cc = H * d
q = torch.randn((B * Sq, Cc), device=dev).half()
k = torch.randn((B * Skv, Cc), device=dev).half()
vt = torch.randn((B * Cc, Skv), device=dev).half()
o = torch.empty((B * Sq, Cc), device=dev, dtype=torch.float16)


def run():
    N.check(lib.sdmi_op_attention(N.ptr(q), Cc, N.ptr(k), Cc, Skv, N.ptr(vt), Skv, N.ptr(o), Cc, B, H, d, Sq, Skv, N.cur_stream()), "attn")


for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
buf = (C.c_ulonglong * (8 * 4096))()
assert lib.sdmi_dbg_read_attn(buf, 4096) == 0
a = np.array(list(buf), dtype=np.float64).reshape(4096, 8)
a = a[a[:, 5] > 0]
per = a[:, :5] / a[:, 5:6]
names = ("stage issue", "tile body", "-", "DMA wait", "barrier")
print(f"{us:.1f} us per launch (with the stamps); {int(a[0, 5])} tiles per wave stamped; cycles per tile, median over {len(a)} waves "
      f"(100 MHz s_memtime ticks x shader clock / 100 MHz is NOT applied: s_memtime counts shader clocks on gfx950)")
for i, nm in enumerate(names):
    print(f"   {nm:24s} {np.median(per[:, i]):8.0f}   (p10 {np.percentile(per[:, i], 10):6.0f}  p90 {np.percentile(per[:, i], 90):6.0f})")
print(f"   {'sum':24s} {np.median(per.sum(axis=1)):8.0f}   mean of sums {per.sum(axis=1).mean():8.0f}")
life_c, life_r = a[:, 6], a[:, 7]
print(f"   wave life (before the final merge / store): {np.median(life_c):.0f} shader clocks, {np.median(life_r) / 100.0:.1f} us  ->  shader clock {np.median(life_c / life_r) * 100.0 / 1000.0:.2f} GHz")
