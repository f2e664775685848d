#!/usr/bin/env python3
"""A/B of one GEMM launch with and without the producer-side GroupNorm statistics (sdmi_gemm_desc::gacc): what the statistics
cost in the epilogue / the split-K combine, back to back (warm) and behind an L2 flush (cold)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_stable_diffusion_amd import _native as N

lib = N.load()
dev = "cuda"
names = [lib.sdmi_gemm_config_name(i).decode() for i in range(lib.sdmi_gemm_num_configs())]


def run(M, Nn, K, cfgname, split, ks=1, H=None, B=2, P=None):
    if ks == 3:
        Cin = K // 9
        a = torch.randn((B, H, H, Cin), device=dev).half()
        Hs = Ws = Ho = Wo = H
        P = H * H
    else:
        a = torch.randn((1, M, 1, K), device=dev).half()
        Hs, Ws, Ho, Wo, Cin = M, 1, M, 1, K
    w = (torch.randn((Nn, K), device=dev) / K ** 0.5).half()
    bias = torch.randn((Nn,), device=dev)
    r = torch.randn((M, Nn), device=dev)
    out = torch.empty((M, Nn), device=dev, dtype=torch.float32)
    out16 = torch.empty((M, Nn), device=dev, dtype=torch.float16)
    rec = torch.empty((8192 * 8,), device=dev)
    res = []
    for stat in (0, 1, 0, 1):
        d = N.GemmDesc()
        d.a0 = a.data_ptr(); d.c0 = Cin
        d.hs, d.ws, d.ho, d.wo = Hs, Ws, Ho, Wo
        d.stride, d.pad, d.ks = 1, (1 if ks == 3 else 0), ks
        d.M, d.N, d.K = M, Nn, K
        d.w = w.data_ptr(); d.bias = bias.data_ptr()
        d.res = r.data_ptr(); d.res_f32 = 1; d.ldr = Nn
        d.out = out.data_ptr(); d.out_f32 = 1; d.ldc = Nn; d.out16 = out16.data_ptr()
        d.cfg = names.index(cfgname); d.ksplit = split
        if stat:
            d.gacc, d.gacc_atom, d.gacc_rows_img = rec.data_ptr(), 10, P
        t = []
        for iters in (50, -20):
            us = C.c_float()
            N.check(lib.sdmi_bench_gemm(C.byref(d), iters, C.byref(us), N.cur_stream()), "bench")
            t.append(us.value)
        res.append(t)
    print(f"M={M} N={Nn} K={K} ks={ks} {cfgname} split {split}: warm off/on {res[0][0]:.2f}/{res[1][0]:.2f} {res[2][0]:.2f}/{res[3][0]:.2f}  "
          f"cold off/on {res[0][1]:.2f}/{res[1][1]:.2f} {res[2][1]:.2f}/{res[3][1]:.2f}", flush=True)


run(512, 1280, 2560, "t64x64s4q2", 1, P=256)
run(2048, 640, 1280, "t128x64s6pc8", 1, P=1024)
run(2048, 640, 1280, "t64x64s4p", 1, P=1024)
run(8192, 320, 2880, "h128x128s3", 1, ks=3, H=64)
run(8192, 320, 2880, "t128x128s3q2", 1, ks=3, H=64)



