#!/usr/bin/env python3
"""GPU micro-benchmark: time the implicit-GEMM kernel per (shape, tile config, split-K)."""
import ctypes as C
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_stable_diffusion_amd import _native as N

lib = N.load()
dev = "cuda"
ncfg = lib.sdmi_gemm_num_configs()
names = [lib.sdmi_gemm_config_name(i).decode() for i in range(ncfg)]


def bench(M, Nn, K, ks=1, B=2, H=None, cfgs=None, splits=(1,), iters=50, res=True):
    if ks == 3:
        Cin = K // 9
        a = torch.randn((B, H, H, Cin), device=dev).half()
        Hs = Ws = Ho = Wo = H
    else:
        a = torch.randn((1, M, 1, K), device=dev).half()
        Hs, Ws, Ho, Wo, B, Cin = M, 1, M, 1, 1, K
    w = (torch.randn((Nn, K), device=dev) / K ** 0.5).half()
    bias = torch.randn((Nn,), device=dev)
    r = torch.randn((M, Nn), device=dev) if res else None
    out = torch.empty((M, Nn), device=dev, dtype=torch.float32)
    out16 = torch.empty((M, Nn), device=dev, dtype=torch.float16)
    rows = []
    for cfg in (cfgs if cfgs is not None else range(ncfg)):
        for sp in splits:
            d = N.GemmDesc()
            d.a0 = a.data_ptr(); d.a1 = 0; d.c0 = Cin; d.c1 = 0
            d.hs, d.ws, d.ho, d.wo = Hs, Ws, Ho, Wo
            d.ups, d.stride, d.pad, d.ks = 0, 1, (1 if ks == 3 else 0), ks
            d.M, d.N, d.K = M, Nn, K
            d.w = w.data_ptr(); d.bias = bias.data_ptr()
            d.res = r.data_ptr() if res else 0; d.res_f32 = 1; d.ldr = Nn
            d.out = out.data_ptr(); d.out_f32 = 1; d.ldc = Nn; d.out16 = out16.data_ptr()
            d.out_t = 0; d.cfg = cfg; d.ksplit = sp
            us = C.c_float()
            rc = lib.sdmi_bench_gemm(C.byref(d), iters, C.byref(us), N.cur_stream())
            if rc != 0:
                continue
            rows.append((us.value, names[cfg], sp))
    rows.sort()
    tf = 2.0 * M * Nn * K / rows[0][0] * 1e-6
    print(f"M={M} N={Nn} K={K} ks={ks}: best {rows[0][0]:.1f} us {rows[0][1]} split {rows[0][2]} ({tf:.0f} TF/s) | " +
          ", ".join(f"{n}/{s}:{u:.1f}" for u, n, s in rows[1:6]), flush=True)
    return rows


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    if which == "small":
        for K in (64, 320, 1280):
            bench(8192, 320, K)
        for K in (64, 640, 2560):
            bench(2048, 640, K)
        bench(512, 1280, 1280, splits=(1, 2, 4))
        bench(128, 1280, 1280, splits=(1, 2, 4, 8))
        bench(160, 320, 768)
        bench(8192, 1280, 320)
    elif which == "conv":
        bench(8192, 320, 2880, ks=3, H=64, splits=(1, 2))
        bench(8192, 320, 5760, ks=3, H=64, splits=(1, 2))
        bench(2048, 640, 5760, ks=3, H=32, splits=(1, 2, 4, 6))
        bench(512, 1280, 11520, ks=3, H=16, splits=(4, 6, 8, 12))
        bench(128, 1280, 11520, ks=3, H=8, splits=(6, 8, 12, 16))
    elif which == "stream":      # weight-streaming shapes of the 8x8 level (M = 128)
        for K in (11520, 23040):
            bench(128, 1280, K, ks=3, H=8, splits=(4, 6, 8, 12, 16))
        bench(128, 1280, 5120, splits=(2, 4, 6, 8))
        bench(512, 1280, 11520, ks=3, H=16, splits=(3, 4, 6, 8))
    elif which == "stream1x1":   # 1x1 GEMMs whose A streams from beyond L2 (attention-block linears at 64x64 / 32x32)
        for (m_, n_, k_) in ((8192, 320, 1600), (8192, 1280, 320), (8192, 960, 320), (8192, 320, 320), (2048, 640, 3200),
                          (2048, 640, 640), (2048, 2560, 640), (512, 1280, 1280), (512, 5120, 1280)):
            bench(m_, n_, k_)
