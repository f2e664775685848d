#!/usr/bin/env python3
"""GPU sweep of the 3x3 / phase convs of one denoising step (512x512 CFG): every applicable tile config x split-K, timed with
cold L2s (sdmi_bench_gemm, iters < 0), best per kernel family.  usage: conv_sweep.py [64|32|16|8|all] [--fam k,h,t]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pytorch_stable_diffusion_amd import _native as N  # noqa: E402

lib = N.load()
dev = "cuda"
ncfg = lib.sdmi_gemm_num_configs()
names = [lib.sdmi_gemm_config_name(i).decode() for i in range(ncfg)]

# (H, Cin0, Cin1, Cout, X, phase) of the step's convs, with their per-step count
SHAPES = {
    64: [(64, 320, 0, 320, 0, 0, 4), (64, 320, 0, 320, 640, 0, 2), (64, 320, 0, 320, 960, 0, 1), (64, 640, 0, 320, 0, 0, 2),
         (64, 960, 0, 320, 0, 0, 1), (32, 640, 0, 640, 0, 1, 1)],
    32: [(32, 320, 0, 640, 0, 0, 1), (32, 640, 0, 640, 0, 0, 2), (32, 640, 0, 640, 320, 0, 1), (32, 640, 0, 640, 960, 0, 1),
         (32, 640, 0, 640, 1280, 0, 1), (32, 640, 0, 640, 1920, 0, 1), (32, 960, 0, 640, 0, 0, 1), (32, 1280, 0, 640, 0, 0, 1),
         (32, 1920, 0, 640, 0, 0, 1), (16, 1280, 0, 1280, 0, 1, 1)],
    16: [(16, 640, 0, 1280, 0, 0, 1), (16, 1280, 0, 1280, 0, 0, 2), (16, 1280, 0, 1280, 640, 0, 1), (16, 1280, 0, 1280, 1920, 0, 1),
         (16, 1280, 0, 1280, 2560, 0, 2), (16, 1920, 0, 1280, 0, 0, 1), (16, 2560, 0, 1280, 0, 0, 2)],
    8: [(8, 1280, 0, 1280, 0, 0, 8), (8, 1280, 0, 1280, 2560, 0, 3), (8, 2560, 0, 1280, 0, 0, 3)],
}


def bench(H, C0, C1, Co, X, phase, fams, splits, reps=6, B=2):
    Cin = C0 + C1
    a = torch.randn((B, H, H, Cin), device=dev).half()
    x = torch.randn((B, H, H, max(X, 64)), device=dev).half()
    T = 4 if phase else 9
    K = T * Cin + X
    M = B * H * H * (4 if phase else 1)
    w = (torch.randn(((4 if phase else 1) * Co, K), device=dev) / K ** 0.5).half()
    bias = torch.randn((Co,), device=dev)
    Mo = M
    out = torch.empty((Mo, Co), device=dev, dtype=torch.float32)
    out16 = torch.empty((Mo, Co), device=dev, dtype=torch.float16)
    res = torch.randn((Mo, Co), device=dev)
    rows = []
    for cfg in range(ncfg):
        if names[cfg][0] not in fams or names[cfg][0] == "g":
            continue
        for sp in splits:
            if sp == 1 and names[cfg].endswith("r") and names[cfg][0] == "k":
                continue
            d = N.GemmDesc()
            d.a0 = a.data_ptr(); d.a1 = 0; d.c0 = Cin; d.c1 = 0
            d.hs = d.ws = d.ho = d.wo = H
            d.ups, d.stride, d.pad, d.ks = 0, 1, (0 if phase else 1), (2 if phase else 3)
            d.M, d.N, d.K = M, Co, K
            d.w = w.data_ptr(); d.bias = bias.data_ptr()
            if X:
                d.x0 = x.data_ptr(); d.cx0 = X
            else:
                d.res = res.data_ptr(); d.res_f32 = 1; d.ldr = Co
            d.out = out.data_ptr(); d.out_f32 = 1; d.ldc = Co; d.out16 = out16.data_ptr()
            d.cfg = cfg; d.ksplit = sp
            if phase:
                d.phase2 = 1; d.img_rows = M // 4; d.w_img_stride = Co * K
            us = C.c_float()
            rc = lib.sdmi_bench_gemm(C.byref(d), -reps, C.byref(us), N.cur_stream())
            if rc != 0:
                continue
            rows.append((us.value, names[cfg], sp))
    rows.sort()
    flops = 2.0 * M * Co * K
    best = {}
    for u, n, s in rows:
        fam = "r" if (n[0] == "k" and n.endswith("r")) else n[0]      # "r": k configs with the in-launch split-K combine
        best.setdefault(fam, (u, n, s))
    line = f"H={H} M={M} N={Co} K={K}{' phase' if phase else ''}{' skip' if X else ''}: " + " | ".join(
        f"{f}: {u:.1f} us {n}/{s} ({flops / u * 1e-6:.0f} TF/s)" for f, (u, n, s) in sorted(best.items()))
    print(line, flush=True)
    print("    top: " + ", ".join(f"{n}/{s}:{u:.1f}" for u, n, s in rows[:8]), flush=True)
    return best, flops


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    fams = "ht"
    for i, arg in enumerate(sys.argv):
        if arg == "--fam":
            fams = sys.argv[i + 1].replace(",", "")
    levels = [64, 32, 16, 8] if which == "all" else [int(which)]
    tot = {}
    for lv in levels:
        for (H, C0, C1, Co, X, ph, cnt) in SHAPES[lv]:
            splits = {64: (1, 2, 3), 32: (1, 2, 3, 4, 6, 8), 16: (2, 3, 4, 6, 8, 10, 12), 8: (4, 6, 8, 10, 12, 16)}[lv]
            best, fl = bench(H, C0, C1, Co, X, ph, fams, splits)
            for f, (u, n, s) in best.items():
                tot[f] = tot.get(f, 0.0) + u * cnt
            tot["best"] = tot.get("best", 0.0) + min(u for u, _, _ in best.values()) * cnt
    print("per-step totals (us, finalize launches of the slab path included):", {k: round(v, 1) for k, v in tot.items()})
