#!/usr/bin/env python3
"""Where the GroupNorm-inside-the-conv variant of the halo kernel (GemmArgs::hgn) spends its extra time: the same conv at several
depths of K (channel chunks), plain (on the normalised fp16 tensor) against fused (raw fp32 tensor + records), cold-L2 timing
(sdmi_bench_gemm iters < 0).  A per-interval cost grows with K, a fixed cost (statistics prologue, first chunk through registers)
does not.  usage: python tools/hgn_probe.py [cfg name ...]"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_stable_diffusion_amd import _native as N
from tests import gpu_util as G

lib = N.load()
dev = "cuda"
names = [lib.sdmi_gemm_config_name(i).decode() for i in range(lib.sdmi_gemm_num_configs())]
want = sys.argv[1:] or ["h128x128s3", "h128x160s3", "h128x128s4"]


def desc(x16, wp, B, H, W, cfg, ksplit, hgn=None):
    d = N.GemmDesc()
    cin = x16.shape[-1]
    M, Nn, K = B * H * W, wp.shape[0], wp.shape[1]
    out = torch.zeros((M, Nn), dtype=torch.float32, device=dev)
    d.a0 = x16.data_ptr(); d.c0 = cin; d.hs, d.ws, d.ho, d.wo = H, W, H, W
    d.stride, d.pad, d.ks = 1, 1, 3
    d.M, d.N, d.K = M, Nn, K
    d.w = wp.data_ptr(); d.out = out.data_ptr(); d.out_f32 = 1; d.ldc = Nn
    d.cfg, d.ksplit = cfg, ksplit
    keep = [out]
    if hgn is not None:
        x32, gamma, beta, rec = hgn
        d.hgn_x0 = x32.data_ptr(); d.hgn_in_f32 = int(x32.dtype == torch.float32); d.hgn_c0 = cin
        d.hgn_gamma, d.hgn_beta, d.hgn_eps, d.hgn_silu = gamma.data_ptr(), beta.data_ptr(), 1e-5, 1
        d.hgn_rec0, d.hgn_t0, d.hgn_p0, d.hgn_atom = rec.data_ptr(), rec.shape[1], rec.shape[3], 10
        keep += [x32, gamma, beta, rec]
    return d, keep


for B, H, Nn in ((2, 64, 320), (2, 32, 640)):
    for cin in (320, 640, 1280):
        g = torch.Generator().manual_seed(cin)
        x = torch.randn((B, H, H, cin), generator=g)
        w = (torch.randn((Nn, cin, 3, 3), generator=g) / math.sqrt(9 * cin)).half()
        wp = G.pack_conv(w.to(dev))
        x32 = x.to(dev)
        x16 = x.half().to(dev)
        gamma, beta = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
        T = (H * H) // 128
        rec = torch.zeros((B, T, cin // 10, 1, 2), device=dev)
        xa = x.double().reshape(B, T, (H * H) // T, cin // 10, 10)
        rec[..., 0, 0] = xa.sum(dim=(2, 4)).float().to(dev)
        rec[..., 0, 1] = (xa * xa).sum(dim=(2, 4)).float().to(dev)
        for nm in want:
            cfg = names.index(nm)
            for ks in (1, 2):
                res = []
                for hg in (None, (x32, gamma, beta, rec), (x16, gamma, beta, rec)):
                    d, keep = desc(x16, wp, B, H, H, cfg, ks, hg)
                    us = C.c_float(0)
                    rc = lib.sdmi_bench_gemm(C.byref(d), -12, C.byref(us), N.cur_stream())
                    res.append(us.value if rc == 0 else float("nan"))
                nk = 9 * cin // 64 // ks
                print(f"M={B*H*H} N={Nn} Cin={cin:4d} {nm} split {ks} ({nk:3d} intervals per workgroup): plain {res[0]:6.1f} us   fused fp32 {res[1]:6.1f}"
                      f" (+{res[1]-res[0]:5.1f})   fused fp16 {res[2]:6.1f} (+{res[2]-res[0]:5.1f})", flush=True)
