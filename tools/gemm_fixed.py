#!/usr/bin/env python3
"""Fixed-cost experiment: time vs K for given config under SDMI_GEMM_DBG."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import bench, names
cfgs = [names.index(c) for c in sys.argv[1].split(",")]
for res in (True, False):
    for K in (64, 640, 2560):
        print("res" if res else "nores", end=" ")
        bench(8192, 320, K, cfgs=cfgs, res=res, iters=30)
