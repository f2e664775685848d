#!/usr/bin/env python3
"""GPU micro-benchmark of the fused GroupNorm+SiLU 3x3 conv (conv3_gn_kernel, 'g' configs) against the unfused kernels on
the UNet's ResBlock conv shapes.  usage: conv_gn_probe.py [shape ...]   (SDMI_LIB selects a diagnostic build)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_stable_diffusion_amd import _native as N

lib = N.load()
dev = "cuda"
names = [lib.sdmi_gemm_config_name(i).decode() for i in range(lib.sdmi_gemm_num_configs())]


def run(B, H, Cin, Cout, cfg_names, gn, silu=1, splits=(1,), iters=30, X=0):
    a = torch.randn((B, H, H, Cin), device=dev).half()
    x0 = torch.randn((B, H, H, X), device=dev).half() if X else None
    K = 9 * Cin + X
    w = (torch.randn((Cout, K), device=dev) / K ** 0.5).half()
    bias = torch.randn((Cout,), device=dev)
    M = B * H * H
    out = torch.empty((M, Cout), device=dev, dtype=torch.float16)
    gamma = torch.ones((Cin,), device=dev); beta = torch.zeros((Cin,), device=dev)
    nch = lib.sdmi_gn_num_chunks(H * H)
    part = torch.zeros((B, nch, 32, 2), device=dev)
    N.check(lib.sdmi_op_gn_stats(a.data_ptr(), 0, 0, Cin, 0, B, H * H, part.data_ptr(), N.cur_stream()), "stats")
    res = []
    for nm in cfg_names:
        for sp in splits:
            d = N.GemmDesc()
            d.a0 = a.data_ptr(); d.c0 = Cin; d.hs = d.ws = d.ho = d.wo = H
            d.stride, d.pad, d.ks = 1, 1, 3
            d.M, d.N, d.K = M, Cout, K
            d.w = w.data_ptr(); d.bias = bias.data_ptr(); d.out = out.data_ptr(); d.ldc = Cout
            d.cfg = names.index(nm); d.ksplit = sp
            if X:
                d.x0 = x0.data_ptr(); d.cx0 = X
            if gn and nm.startswith("g"):
                d.gn_partial = part.data_ptr(); d.gn_nchunk = nch; d.gn_gamma = gamma.data_ptr(); d.gn_beta = beta.data_ptr()
                d.gn_eps = 1e-5; d.gn_silu = silu
            us = C.c_float()
            rc = lib.sdmi_bench_gemm(C.byref(d), iters, C.byref(us), N.cur_stream())
            if rc != 0:
                res.append(f"{nm}/{sp}: n/a")
                continue
            res.append(f"{nm}/{sp}: {us.value:6.1f} us ({2.0 * M * Cout * K / us.value * 1e-6:5.0f} TF/s)")
    print(f"B={B} H={H} {Cin}->{Cout} X={X} gn={gn} silu={silu} | " + " | ".join(res), flush=True)


if __name__ == "__main__":
    tag = os.environ.get("SDMI_LIB", "default")
    print("lib:", tag)
    run(2, 64, 320, 320, ["t128x128s3p", "h128x128s3", "g128x128d2", "t128x160s3p", "t128x160s4p", "h128x160s3"], gn=False, splits=(1, 2))
    run(2, 64, 640, 320, ["h256x128s3", "g128x128d2", "t128x160s3p", "t128x160s4p", "h128x160s3"], gn=False, splits=(1, 2, 4))
    run(2, 64, 960, 320, ["h256x128s3", "t128x160s4p", "h128x160s3"], gn=False, splits=(2, 4, 6))
    run(2, 32, 640, 640, ["t128x128s3p", "h128x128s4", "t128x160s4p", "h128x160s3", "t64x160s4p"], gn=False, splits=(2, 3, 4))
    run(2, 16, 1280, 1280, ["t128x128s3p", "t128x160s4p", "h128x160s3", "t64x160s4p"], gn=False, splits=(4, 6, 8))
    run(2, 8, 1280, 1280, ["t64x64s3p2", "t64x128s4p", "t64x160s4p"], gn=False, splits=(6, 8, 12, 16))
