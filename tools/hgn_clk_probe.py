#!/usr/bin/env python3
"""In-kernel clock stamps of the halo conv with GroupNorm inside (diagnostic build: tools/build_variant.sh probe gemm -DSDMI_CLK_PROBE
-DSDMI_CLK_PROBE_FINE; SDMI_LIB=pytorch_stable_diffusion_amd/lib/variants/libsdmi_probe.so): per workgroup, summed over its intervals,
cycles the weight producers / MFMA waves / normaliser waves spend in each part of an interval -- plain kernel beside the fused one."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytorch_stable_diffusion_amd import _native as N
from tests import gpu_util as G
from tools.hgn_probe import desc, names, lib, dev

B, H, cin, Nn = 2, 64, int(os.environ.get("PROBE_CIN", "320")), 320
g = torch.Generator().manual_seed(1)
x = torch.randn((B, H, H, cin), generator=g)
w = (torch.randn((Nn, cin, 3, 3), generator=g) / math.sqrt(9 * cin)).half()
wp = G.pack_conv(w.to(dev))
x32, x16 = x.to(dev), x.half().to(dev)
gamma, beta = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
T = (H * H) // 128
rec = torch.zeros((B, T, cin // 10, 1, 2), device=dev)
xa = x.double().reshape(B, T, (H * H) // T, cin // 10, 10)
rec[..., 0, 0] = xa.sum(dim=(2, 4)).float().to(dev)
rec[..., 0, 1] = (xa * xa).sum(dim=(2, 4)).float().to(dev)
for nm in sys.argv[1:] or ["h128x128s3"]:
    cfg = names.index(nm)
    for label, hg in (("plain", None), ("fused fp32", (x32, gamma, beta, rec)), ("fused fp16", (x16, gamma, beta, rec))):
        d, keep = desc(x16, wp, B, H, H, cfg, 1, hg)
        us = C.c_float(0)
        for _ in range(3):
            N.check(lib.sdmi_bench_gemm(C.byref(d), -8, C.byref(us), N.cur_stream()), "bench")
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (2 * 2048))()
        lib.sdmi_dbg_read_clk(buf, 2048)
        full = np.array(list(buf), dtype=np.float64).reshape(-1, 2)
        pre = (C.c_ulonglong * (10 * 2048))()
        lib.sdmi_dbg_read_pre(pre, 2048)
        pr = np.array(list(pre), dtype=np.float64).reshape(-1, 10)
        ph = (C.c_ulonglong * (6 * 2048))()
        lib.sdmi_dbg_read_phase(ph, 2048)
        phs = np.array(list(ph), dtype=np.float64).reshape(-1, 6)
        nb = 192
        nk = 9 * cin // 64
        med = lambda a: float(np.median(a[:nb]))
        print(f"{nm} {label}: {us.value:.1f} us, {nk} intervals; per interval (cycles, median over workgroups):")
        print(f"   weight producers: DMA issue {med(full[512:, 0]) / nk:6.0f}  vmcnt wait {med(full[512:, 1]) / nk:6.0f}  barrier wait {med(full[1024:, 0]) / nk:6.0f}")
        print(f"   MFMA waves      : ds_read + MFMA {med(full[1536:, 0]) / nk:6.0f}  barrier wait {med(full[1536:, 1]) / nk:6.0f}")
        if hg is not None:
            print(f"   normalisers     : request {med(pr[:, 0]) / nk:6.0f}  vmcnt wait {med(pr[:, 1]) / nk:6.0f}  normalise {med(pr[:, 2]) / nk:6.0f}  barrier wait {med(pr[:, 3]) / nk:6.0f}")
        print(f"   workgroup life (cycles): set-up {med(phs[:, 2]):.0f}  K loop done {med(phs[:, 3]):.0f}  tile in LDS {med(phs[:, 4]):.0f}  end {med(phs[:, 5]):.0f}", flush=True)
