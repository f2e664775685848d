#!/usr/bin/env python3
"""Diagnostic: in-kernel clock of the GEMM (build with SDMI_HIPCC_FLAGS=-DSDMI_CLK_PROBE, SDMI_LIB=...)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools.gemm_sweep import bench, names
from pytorch_stable_diffusion_amd import _native as N
lib = C.CDLL(os.environ["SDMI_LIB"])
KK = int(os.environ.get("PROBE_K", "5760"))
KS = int(os.environ.get("PROBE_KS", "3"))
MM = int(os.environ.get("PROBE_M", "8192")); NN = int(os.environ.get("PROBE_N", "320")); HH = int(os.environ.get("PROBE_H", "64"))
SP = int(os.environ.get("PROBE_SPLIT", "1"))
for cfgname in sys.argv[1:]:
    cfg = names.index(cfgname)
    for _ in range(3):
        bench(MM, NN, KK, ks=KS, H=HH, cfgs=[cfg], splits=(SP,), iters=200)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (2 * 2048))()
    lib.sdmi_dbg_read_clk(buf, 2048)
    full = np.array(list(buf), dtype=np.float64).reshape(-1, 2)
    nb = int(os.environ.get("PROBE_BLOCKS", "192"))
    a = full[:nb]
    pi, pv, pb = full[512:512 + nb, 0], full[512:512 + nb, 1], full[1024:1024 + nb, 0]
    cc, cb = full[1536:1536 + nb, 0], full[1536:1536 + nb, 1]
    print(f"   producer per WG (cycles): issue {np.median(pi):.0f}  vmcnt-wait {np.median(pv):.0f}  barrier-wait {np.median(pb):.0f}")
    print(f"   consumer per WG (cycles): compute(ds_read+mfma issue) {np.median(cc):.0f}  barrier-wait {np.median(cb):.0f}")
    ghz = a[:, 0] / (a[:, 1] * 10.0) / 1e0   # cycles / (ticks * 10 ns) -> cycles/ns = GHz
    print(f"{cfgname}: shader cycles median {np.median(a[:,0]):.0f}, realtime {np.median(a[:,1])*10/1e3:.1f} us, clock median {np.median(ghz):.3f} GHz (min {ghz.min():.3f} max {ghz.max():.3f})")
