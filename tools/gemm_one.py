#!/usr/bin/env python3
"""Run ONE implicit-GEMM shape/config repeatedly (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import bench, names
M, N, K, ks, H, cfgname, split = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], int(sys.argv[7])
cfg = names.index(cfgname)
bench(M, N, K, ks=ks, H=H, cfgs=[cfg], splits=(split,), iters=20)
