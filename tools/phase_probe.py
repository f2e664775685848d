#!/usr/bin/env python3
"""Diagnostic: life of one workgroup of a small GEMM (build with SDMI_HIPCC_FLAGS=-DSDMI_CLK_PROBE, SDMI_LIB=...).
usage: [PROBE_KS=3 PROBE_H=16 PROBE_SPLIT=6] phase_probe.py M N K cfgname"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools.gemm_sweep import bench, names
lib = C.CDLL(os.environ["SDMI_LIB"])
M, Nn, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
KS = int(os.environ.get("PROBE_KS", "1")); HH = int(os.environ.get("PROBE_H", "1")); SP = int(os.environ.get("PROBE_SPLIT", "1"))
for cfgname in sys.argv[4:]:
    cfg = names.index(cfgname)
    for _ in range(2):
        r = bench(M, Nn, K, ks=KS, H=HH, cfgs=[cfg], splits=(SP,), iters=100)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (6 * 2048))()
    lib.sdmi_dbg_read_phase(buf, 2048)
    a = np.array(list(buf), dtype=np.float64).reshape(-1, 6)
    a = a[a[:, 1] > 0]
    t0 = a[:, 0].min()
    st, en = (a[:, 0] - t0) * 10, (a[:, 1] - t0) * 10     # ns
    print(f"{cfgname} M={M} N={Nn} K={K}: {len(a)} WGs, bench {r}")
    print(f"   WG start  (ns after first): median {np.median(st):.0f}  p90 {np.percentile(st, 90):.0f}  max {st.max():.0f}")
    print(f"   WG end    (ns after first): median {np.median(en):.0f}  p90 {np.percentile(en, 90):.0f}  max {en.max():.0f}")
    print(f"   WG life (ns): median {np.median(en - st):.0f}")
    if hasattr(lib, "sdmi_dbg_read_pre"):
        pb = (C.c_ulonglong * (10 * 2048))()
        lib.sdmi_dbg_read_pre(pb, 2048)
        pre = np.array(list(pb), dtype=np.float64).reshape(-1, 10)[:len(a)]
        print("   producer: setup done %.0f, ring primed (issued) %.0f, first stage landed %.0f | consumer past first barrier %.0f (cycles, medians)"
              % tuple(np.median(pre[:, i]) for i in range(4)))
        print("   kernel arguments in the scalar cache after %.0f cycles" % np.median(pre[:, 4]))
        print("   producer wave: tile map done %.0f, LayerNorm block %.0f, A row state %.0f, B pointers %.0f" % tuple(np.median(pre[:, i]) for i in (5, 6, 7, 8)))
    for i, nm in ((2, "setup"), (3, "K loop done"), (4, "tile in LDS"), (5, "stores retired")):
        print(f"   cycles to {nm:15s}: median {np.median(a[:, i]):.0f}")
