#!/usr/bin/env python3
"""Time the flash-attention kernel on the UNet's shapes (events around back-to-back launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_stable_diffusion_amd import _native as N
lib = N.load()
dev = "cuda"
for (d, Sq, Skv) in ((40, 4096, 4096), (40, 4096, 77), (80, 1024, 1024), (80, 1024, 77), (160, 256, 256), (160, 256, 77)):
    B, H = 2, 8
    C = H * d
    q = torch.randn((B * Sq, C), device=dev).half()
    kb = max(Skv, 80) if Skv == 77 else Skv
    k = torch.randn((B * kb, C), device=dev).half()
    ld = ((Skv + 63) // 64) * 64
    vt = torch.randn((B * C, ld), device=dev).half()
    o = torch.empty((B * Sq, C), device=dev, dtype=torch.float16)
    def run():
        N.check(lib.sdmi_op_attention(N.ptr(q), C, N.ptr(k), C, kb, N.ptr(vt), ld, N.ptr(o), C, B, H, d, Sq, Skv, N.cur_stream()), "attn")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"d={d} Sq={Sq} Skv={Skv}: {us:.1f} us  ({4.0*B*H*Sq*Skv*d/us*1e-6:.0f} TF/s)", flush=True)
