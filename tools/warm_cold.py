#!/usr/bin/env python3
"""How much of a small GEMM's time is the cold fetch of its operands?  Per shape and tuned config: one launch timed with an event pair
(a) after a 64 MiB fill (the tuner's cold measure), (b) right after an untimed run of the same launch (operands in the L2s that will
read them), (c) after a run of the same launch followed by a fill-free pause.  usage: python tools/warm_cold.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_stable_diffusion_amd import _native as N  # noqa: E402

lib = N.load()
dev = "cuda"
names = [lib.sdmi_gemm_config_name(i).decode() for i in range(lib.sdmi_gemm_num_configs())]


def desc(M, Nn, K, cfg, sp, a, w, bias, r, out, out16):
    d = N.GemmDesc()
    d.a0 = a.data_ptr(); d.a1 = 0; d.c0 = K; d.c1 = 0
    d.hs, d.ws, d.ho, d.wo = M, 1, M, 1
    d.ups, d.stride, d.pad, d.ks = 0, 1, 0, 1
    d.M, d.N, d.K = M, Nn, K
    d.w = w.data_ptr(); d.bias = bias.data_ptr()
    d.res = r.data_ptr(); d.res_f32 = 1; d.ldr = Nn
    d.out = out.data_ptr(); d.out_f32 = 1; d.ldc = Nn; d.out16 = out16.data_ptr()
    d.out_t = 0; d.cfg = cfg; d.ksplit = sp
    return d


def one(M, Nn, K, cfg_name, sp=1, reps=15):
    cfg = names.index(cfg_name)
    a = torch.randn((M, K), device=dev).half()
    w = (torch.randn((Nn, K), device=dev) / K ** 0.5).half()
    bias = torch.randn((Nn,), device=dev)
    r = torch.randn((M, Nn), device=dev)
    out = torch.empty((M, Nn), device=dev, dtype=torch.float32)
    out16 = torch.empty((M, Nn), device=dev, dtype=torch.float16)
    thrash = torch.empty((64 << 20,), device=dev, dtype=torch.uint8)
    d = desc(M, Nn, K, cfg, sp, a, w, bias, r, out, out16)
    st = N.cur_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(prep):
        best = 1e9
        for i in range(reps):
            prep(i)
            e0.record()
            assert lib.sdmi_op_gemm(C.byref(d), st) == 0
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        return best

    def cold(i):
        thrash.fill_(i & 255)

    def warm(i):
        assert lib.sdmi_op_gemm(C.byref(d), st) == 0

    def warm_w_only(i):            # operands cold, then ONLY the weights touched by a run on a copy of the activations
        thrash.fill_(i & 255)
        d2 = desc(M, Nn, K, cfg, sp, a.clone(), w, bias, r.clone(), torch.empty_like(out), torch.empty_like(out16))
        assert lib.sdmi_op_gemm(C.byref(d2), st) == 0

    c, wm, ww = timed(cold), timed(warm), timed(warm_w_only)
    print(f"M={M:5d} N={Nn:5d} K={K:5d} {cfg_name:14s} split {sp}: cold {c:6.2f} us   weights warm (own-XCD L2s) {ww:6.2f} us   all warm {wm:6.2f} us", flush=True)


if __name__ == "__main__":
    one(512, 1280, 1280, "t64x64s4q2")
    one(2048, 640, 640, "t64x64s4p")
    one(512, 1280, 2560, "t64x64s4q2")
    one(512, 3840, 1280, "t64x128s4q2")
    one(2048, 640, 1280, "t64x128s4q2")
    one(2048, 1920, 640, "t128x128s3pc8")
    one(128, 1280, 1280, "t64x64s4q2")
    one(8192, 320, 320, "t128x64s2")
