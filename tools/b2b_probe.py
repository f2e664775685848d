"""Times the back-to-back GEMM (csrc/b2b.hip) in isolation against the two igemm launches it replaces.
   python tools/b2b_probe.py [M]      (SDMI_LIB selects a variant build)"""
import ctypes as C
import math
import sys

import torch

sys.path.insert(0, ".")
from pytorch_stable_diffusion_amd import _native as N  # noqa: E402
from tests import gpu_util as G  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    dev = "cuda"
    lib = N.load()
    g = torch.Generator().manual_seed(0)
    Cc = 320
    a1 = torch.randn((M, Cc), generator=g).half().to(dev)
    w1 = (torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)).half().to(dev)
    b1 = torch.randn((Cc,), generator=g).to(dev)
    r1 = torch.randn((M, Cc), generator=g).to(dev)
    r2 = torch.randn((M, Cc), generator=g).to(dev)
    gamma = (1 + 0.1 * torch.randn((Cc,), generator=g)).to(dev)
    beta = (0.1 * torch.randn((Cc,), generator=g)).to(dev)
    for K2, partial, bm in ((320, 0, 64), (320, 0, 32), (640, 1, 64), (640, 1, 32)):
        w2 = (torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)).to(dev)
        wf, gf, hf = G.ln_fold_prep(w2, gamma, beta, b1)
        if partial:
            wp = (torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)).half().to(dev)
            wf = torch.cat([wf, wp], dim=1).contiguous()
        s32 = torch.empty((M, Cc), device=dev)
        s16 = torch.empty((M, Cc), dtype=torch.float16, device=dev)
        out = torch.empty((M, Cc), device=dev)
        out16 = torch.empty((M, Cc), dtype=torch.float16, device=dev)
        d = N.B2bDesc()
        d.a1, d.lda1, d.w1, d.b1 = a1.data_ptr(), Cc, w1.data_ptr(), b1.data_ptr()
        d.r1, d.r1_f32 = r1.data_ptr(), 1
        if not partial:
            d.s32, d.s16 = s32.data_ptr(), s16.data_ptr()
        d.w2, d.K2, d.h2, d.partial, d.cscale = wf.data_ptr(), K2, hf.data_ptr(), partial, (0.0 if partial else 0.25)
        if partial:
            d.r2, d.r2_f32 = r2.data_ptr(), 1
            d.out, d.out_f32, d.out16 = out.data_ptr(), 1, out16.data_ptr()
        else:
            d.out, d.out_f32 = out16.data_ptr(), 0
        d.M, d.eps, d.bm = M, 1e-5, bm
        us = C.c_float(0)
        N.check(lib.sdmi_op_b2b(C.byref(d), 50, C.byref(us), N.cur_stream()), "b2b")
        torch.cuda.synchronize()
        print(f"b2b M={M} K2={K2} partial={partial} BM={bm}: {us.value:.2f} us / launch (warm, back to back)")
        if hasattr(lib, "sdmi_dbg_read_b2b"):            # -DSDMI_B2B_PROBE build: shader-clock stamps of workgroup 0
            buf = (C.c_ulonglong * 16)()
            lib.sdmi_dbg_read_b2b.argtypes = [C.POINTER(C.c_ulonglong)]
            lib.sdmi_dbg_read_b2b(buf)
            mf, dm = list(buf[0:8]), list(buf[8:16])
            t0 = min(mf[0], dm[0])
            print("   MFMA wave: start %d | first tile ready %d | product 1 done %d | stats in regs %d | first W2 tile %d | product 2 done %d | tile in LDS %d"
                  % tuple(x - t0 for x in mf[:7]))
            print("   DMA wave : start %d | residual fetch issued %d | W1 streamed %d | Cs ready %d | epilogue 1 done %d | ring free %d | W2 streamed %d | end %d"
                  % tuple(x - t0 for x in dm[:8]))


def qkv(M=8192, S=4096):
    """conv_input + layernorm_1 + in_proj (three-pass form)"""
    dev, Cc = "cuda", 320
    lib = N.load()
    g = torch.Generator().manual_seed(1)
    a1 = torch.randn((M, Cc), generator=g).half().to(dev)
    w1 = (torch.randn((Cc, Cc), generator=g) / math.sqrt(Cc)).half().to(dev)
    b1 = torch.randn((Cc,), generator=g).to(dev)
    gamma = (1 + 0.1 * torch.randn((Cc,), generator=g)).to(dev)
    beta = (0.1 * torch.randn((Cc,), generator=g)).to(dev)
    w2 = (torch.randn((3 * Cc, Cc), generator=g) / math.sqrt(Cc)).to(dev)
    wf, _, hf = G.ln_fold_prep(w2, gamma, beta, None)
    s32 = torch.empty((M, Cc), device=dev)
    s16 = torch.empty((M, Cc), dtype=torch.float16, device=dev)
    qk = torch.empty((M, 2 * Cc), dtype=torch.float16, device=dev)
    vt = torch.zeros((M // S * Cc, S), dtype=torch.float16, device=dev)
    for bm in (64, 32):
        d = N.B2bDesc()
        d.a1, d.lda1, d.w1, d.b1 = a1.data_ptr(), Cc, w1.data_ptr(), b1.data_ptr()
        d.s32, d.s16 = s32.data_ptr(), s16.data_ptr()
        d.w2, d.K2, d.h2, d.cscale = wf.data_ptr(), 320, hf.data_ptr(), 0.25
        d.out, d.ldo, d.npass2, d.vt, d.S, d.ldt = qk.data_ptr(), 2 * Cc, 3, vt.data_ptr(), S, S
        d.M, d.eps, d.bm = M, 1e-5, bm
        us = C.c_float(0)
        N.check(lib.sdmi_op_b2b(C.byref(d), 50, C.byref(us), N.cur_stream()), "b2b qkv")
        torch.cuda.synchronize()
        print(f"b2b qkv M={M} BM={bm}: {us.value:.2f} us / launch (warm, back to back)")


if __name__ == "__main__":
    main()
    qkv(int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 4096 if len(sys.argv) < 2 or int(sys.argv[1]) == 8192 else int(sys.argv[1]) // 2)
