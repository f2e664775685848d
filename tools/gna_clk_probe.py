#!/usr/bin/env python3
"""In-kernel clock stamps of gn_apply_kernel with producer records (diagnostic build: tools/build_variant.sh probe norm -DSDMI_GNA_PROBE;
SDMI_LIB=pytorch_stable_diffusion_amd/lib/variants/libsdmi_probe.so): shader clocks from a workgroup's start to each phase, median over
the workgroups, for the GroupNorm shapes of the 64x64 / 32x32 levels (one source and a concat of two)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytorch_stable_diffusion_amd import _native as N
from tests import gpu_util as G

lib = N.load()
dev = "cuda"


def records(x, T, parts=1):
    B, H, W, Cc = x.shape
    xa = x.double().reshape(B, T, (H * W) // T, Cc // 10, 10)
    rec = torch.zeros((B, T, Cc // 10, parts, 2), device=dev)
    rec[..., 0, 0] = xa.sum(dim=(2, 4)).float().to(dev)
    rec[..., 0, 1] = (xa * xa).sum(dim=(2, 4)).float().to(dev)
    return rec


for (c0, c1, hw, T) in ((320, 0, 64, 32), (320, 320, 64, 32), (640, 0, 32, 16), (320, 0, 32, 8), (960, 960, 32, 16)):
    g = torch.Generator().manual_seed(c0 + c1 + hw)
    x0 = torch.randn((2, hw, hw, c0), generator=g)
    x1 = torch.randn((2, hw, hw, c1), generator=g) if c1 else None
    gamma, beta = torch.ones(c0 + c1, device=dev), torch.zeros(c0 + c1, device=dev)
    st0 = (records(x0, T), T, 1)
    st1 = (records(x1, T), T, 1) if c1 else None
    x0d, x1d = x0.to(dev), (x1.to(dev) if c1 else None)
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for _ in range(3):
        scratch.fill_(1)                      # the inputs leave the L2s, as behind a producer kernel
        G.groupnorm_acc(x0d, x1d, st0, st1, gamma, beta, 1e-5, True)
    buf = (C.c_ulonglong * (6 * 2048))()
    assert lib.sdmi_dbg_read_gna(buf, 2048) == 0
    a = np.array(list(buf), dtype=np.float64).reshape(2048, 6)
    a = a[a[:, 5] > 0]
    names = ("loads requested", "records summed", "mean / rstd ready", "table ready", "stores issued")
    print(f"C = {c0}+{c1}, {hw}x{hw}, B = 2, T = {T}: {len(a)} workgroups; shader clocks since the workgroup's start (median, p10 - p90); "
          f"first -> last workgroup start {a[:, 0].max() - a[:, 0].min():.0f} clocks")
    for i, nm in enumerate(names):
        col = a[:, 1 + i]
        print(f"   {nm:20s} {np.median(col):8.0f}   ({np.percentile(col, 10):6.0f} - {np.percentile(col, 90):6.0f})")
