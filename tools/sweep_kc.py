#!/usr/bin/env python3
"""Cold-L2 ranking of every tile config on the K = C GEMM shapes of a step (the tuner's measure, csrc/engine.h tune_gemm)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import bench
for (m, n, k, sp) in ((512, 1280, 1280, (1, 2)), (2048, 640, 640, (1,)), (512, 1280, 2560, (1, 2)), (512, 3840, 1280, (1,)),
                      (2048, 640, 1280, (1,)), (2048, 1920, 640, (1,)), (2048, 640, 1024, (1,)), (2048, 1024, 640, (1,)), (512, 1280, 1024, (1,)),
                      (128, 1280, 1280, (1, 2, 4))):
    bench(m, n, k, splits=sp, iters=-15)
