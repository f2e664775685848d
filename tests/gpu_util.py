"""Helpers for -m gpu tests: thin wrappers over the C ABI (kernel-level entry points)."""
import ctypes as C
import json
import os
import time

import torch

from pytorch_stable_diffusion_amd import _native as N

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def log_metric(**kw):
    try:
        os.makedirs(OUT, exist_ok=True)
        kw["ts"] = time.time()
        with open(os.path.join(OUT, "test_metrics.jsonl"), "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def pack_conv(w: torch.Tensor, o_keep=None) -> torch.Tensor:
    """OIHW (or [O][I]) cuda tensor -> packed [O][kh][kw][I] fp16 via the library's packer."""
    lib = N.load()
    if w.dim() == 2:
        O, I, ks = w.shape[0], w.shape[1], 1
    else:
        O, I, ks = w.shape[0], w.shape[1], w.shape[2]
    o_keep = O if o_keep is None else o_keep
    out = torch.empty((o_keep, ks * ks * I), dtype=torch.float16, device=w.device)
    code = N.SDMI_F32 if w.dtype == torch.float32 else N.SDMI_F16
    N.check(lib.sdmi_op_pack_conv(N.ptr(w.contiguous()), code, N.ptr(out), O, I, ks, o_keep, N.cur_stream()), "pack")
    return out


def pack_ups_phase(w: torch.Tensor) -> torch.Tensor:
    """OIHW 3x3 cuda tensor -> [4*O][4*I] fp16: the four phase matrices of the x2-upsample conv, stacked"""
    lib = N.load()
    O, I = w.shape[0], w.shape[1]
    out = torch.empty((4 * O, 4 * I), dtype=torch.float16, device=w.device)
    code = N.SDMI_F32 if w.dtype == torch.float32 else N.SDMI_F16
    N.check(lib.sdmi_op_pack_ups_phase(N.ptr(w.contiguous()), code, N.ptr(out), O, I, N.cur_stream()), "pack_ups_phase")
    return out


def igemm(a0, w_packed, *, B, Hs, Ws, Ho, Wo, ks=1, stride=1, ups=0, a1=None, bias=None, res=None,
          out_f32=False, cfg=-1, ksplit=1, out_t=None, nt0=0, S=0, ldt=0, want16=False, x0=None, x1=None,
          rowstat=None, ln_stat=None, ln_g=None, ln_c=0, ln_eps=1e-5, out_t_perm=0, act=0, sm_valid=0, img_rows=0,
          w_img_stride=0, vec_img_stride=0, ldw=0, n_out=None, phase2=0, ln_ksteps=0, ln_out=None, gstat_rows_img=0, gstat_atom=10, ln_guard=None,
          ln_guard_sigma=8.0, gna=None, accurate=False, a0f=None, a1f=None, x0f=None, x1f=None, hgn=None):
    """a0/a1: NHWC fp16 (B,Hs,Ws,C).  Returns out [M][N'] (N' = nt0 if out_t given).
    gstat_rows_img > 0: the launch also leaves the GroupNorm statistics of its output (sdmi_gemm_desc::gacc); they are returned
    in LAST_STAT = (records [images][T][atoms][parts][2] fp32, T, parts).  ValueError if the config cannot take them."""
    global LAST_STAT
    lib = N.load()
    d = N.GemmDesc()
    c0 = a0.shape[-1]
    c1 = 0 if a1 is None else a1.shape[-1]
    Nn = w_packed.shape[0] if n_out is None else n_out      # n_out: per-image weights stacked in w_packed
    K = w_packed.shape[1] if not ldw else ks * ks * (c0 + c1)
    M = B * Ho * Wo * (4 if phase2 else 1)
    ncols = Nn if out_t is None else max(nt0, 8)
    out = torch.zeros((M, ncols), dtype=torch.float32 if out_f32 else torch.float16, device=a0.device)
    out16 = torch.zeros((M, ncols), dtype=torch.float16, device=a0.device) if (want16 and out_f32) else None
    d.a0 = a0.data_ptr(); d.a1 = 0 if a1 is None else a1.data_ptr()
    d.c0, d.c1, d.hs, d.ws, d.ho, d.wo = c0, c1, Hs, Ws, Ho, Wo
    d.ups, d.stride, d.pad, d.ks = ups, stride, (1 if ks == 3 else 0), ks
    d.phase2 = phase2
    d.ln_ksteps = ln_ksteps
    d.ln_out = 0 if ln_out is None else ln_out.data_ptr()
    d.M, d.N, d.K = M, Nn, K
    d.w = w_packed.data_ptr()
    d.bias = 0 if bias is None else bias.data_ptr()
    d.res = 0 if res is None else res.data_ptr()
    d.res_f32 = int(res is not None and res.dtype == torch.float32)
    d.ldr = 0 if res is None else res.shape[-1]
    d.out = out.data_ptr(); d.out_f32 = int(out_f32); d.ldc = ncols
    d.out16 = 0 if out16 is None else out16.data_ptr()
    d.out_t = 0 if out_t is None else out_t.data_ptr()
    d.nt0, d.S, d.ldt = nt0, S, ldt
    d.out_t_perm = out_t_perm
    d.cfg, d.ksplit = cfg, ksplit
    d.x0 = 0 if x0 is None else x0.data_ptr(); d.cx0 = 0 if x0 is None else x0.shape[-1]
    d.x1 = 0 if x1 is None else x1.data_ptr(); d.cx1 = 0 if x1 is None else x1.shape[-1]
    d.rowstat = 0 if rowstat is None else rowstat.data_ptr()
    if ln_stat is not None:
        d.ln_stat, d.ln_ntn, d.ln_g, d.ln_c, d.ln_eps = ln_stat.data_ptr(), ln_stat.shape[1], ln_g.data_ptr(), ln_c, ln_eps
    d.act, d.sm_valid, d.img_rows, d.w_img_stride, d.vec_img_stride, d.ldw = act, sm_valid, img_rows, w_img_stride, vec_img_stride, ldw
    if ln_guard is not None:
        d.ln_guard, d.ln_guard_sigma = ln_guard.data_ptr(), ln_guard_sigma
    if gna is not None:        # (records [images][T][atoms][parts][2], gamma, beta, eps, rows per image): GroupNorm on the A fragments
        rec, gamma, beta, eps, rows = gna
        d.gna_rec, d.gna_gamma, d.gna_beta, d.gna_eps = rec.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps
        d.gna_t, d.gna_parts, d.gna_atom, d.gna_rows = rec.shape[1], rec.shape[3], c0 // rec.shape[2], rows
    if hgn is not None:        # GroupNorm(+SiLU) inside the halo conv: dict(x0, x1, gamma, beta, eps, silu, rec0, rec1) -- a0 only gives the shape
        x0r, x1r = hgn["x0"], hgn.get("x1")
        d.hgn_x0 = x0r.data_ptr(); d.hgn_x1 = 0 if x1r is None else x1r.data_ptr()
        d.hgn_in_f32 = int(x0r.dtype == torch.float32)
        d.hgn_c0, d.hgn_c1 = x0r.shape[-1], 0 if x1r is None else x1r.shape[-1]
        d.hgn_gamma, d.hgn_beta, d.hgn_eps, d.hgn_silu = hgn["gamma"].data_ptr(), hgn["beta"].data_ptr(), hgn["eps"], int(hgn["silu"])
        r0, r1 = hgn["rec0"], hgn.get("rec1")          # records [images][T][atoms][parts][2] fp32
        d.hgn_rec0, d.hgn_t0, d.hgn_p0 = r0.data_ptr(), r0.shape[1], r0.shape[3]
        d.hgn_atom = d.hgn_c0 // r0.shape[2]
        if r1 is not None:
            d.hgn_rec1, d.hgn_t1, d.hgn_p1 = r1.data_ptr(), r1.shape[1], r1.shape[3]
    if accurate:               # the wide-operand kernels: fp32 copies of every A source (a0 / a1 / x0 / x1 only give the shapes)
        d.accurate = 1
        d.a0f = a0f.data_ptr(); d.a1f = 0 if a1f is None else a1f.data_ptr()
        d.x0f = 0 if x0f is None else x0f.data_ptr(); d.x1f = 0 if x1f is None else x1f.data_ptr()
        assert a0f.dtype == torch.float32 and tuple(a0f.shape) == tuple(a0.shape)
    if gstat_rows_img:
        d.gacc, d.gacc_atom, d.gacc_rows_img = 16, gstat_atom, gstat_rows_img          # (any non-null pointer for the layout query)
        T, parts = C.c_int(0), C.c_int(0)
        N.check(lib.sdmi_op_gemm_stat_layout(C.byref(d), C.byref(T), C.byref(parts)), "stat_layout")
        imgs = (M // (4 if phase2 else 1)) // gstat_rows_img
        rec = torch.full((imgs, T.value, Nn // gstat_atom, parts.value, 2), float("nan"), dtype=torch.float32, device=a0.device)
        d.gacc = rec.data_ptr()
        LAST_STAT = (rec, T.value, parts.value)
    N.check(lib.sdmi_op_gemm(C.byref(d), N.cur_stream()), "sdmi_op_gemm")
    torch.cuda.synchronize()
    return (out, out16) if want16 else out


def _vt_index(n):
    """stored position -> key index along a V^T row: inside each group of 16 keys the 4-key quads are stored
    in the order (q0, q2, q1, q3)  (csrc/gemm.hip vt_pos; the PV MFMA reads quads h and 2+h as one 16-byte load)."""
    idx = torch.arange(n)
    u, g = idx // 4, (idx // 4) % 4
    gp = torch.where(g == 1, torch.full_like(g, 2), torch.where(g == 2, torch.full_like(g, 1), g))
    pos = ((u - g + gp) * 4) + idx % 4          # pos[key] = stored position
    inv = torch.empty_like(pos)
    inv[pos] = idx
    return pos, inv


def vt_store_order(vt_natural):
    """(rows, keys) with keys in natural order -> the library's stored order (keys padded to a multiple of 16)"""
    _, inv = _vt_index(vt_natural.shape[-1])
    return vt_natural[..., inv.to(vt_natural.device)].contiguous()


def vt_natural_order(vt_stored):
    pos, _ = _vt_index(vt_stored.shape[-1])
    return vt_stored[..., pos.to(vt_stored.device)].contiguous()


def attention(q, k, vt, B, H, d, Sq, Skv, k_batch_stride=None):
    """vt: V^T with keys in NATURAL order (rows padded to a multiple of 64 keys); permuted here for the library."""
    lib = N.load()
    vt = vt_store_order(vt)
    o = torch.zeros((B * Sq, H * d), dtype=torch.float16, device=q.device)
    kbs = Skv if k_batch_stride is None else k_batch_stride
    N.check(lib.sdmi_op_attention(N.ptr(q), q.shape[-1], N.ptr(k), k.shape[-1], kbs, N.ptr(vt), vt.shape[-1],
                                  N.ptr(o), o.shape[-1], B, H, d, Sq, Skv, N.cur_stream()), "sdmi_op_attention")
    torch.cuda.synchronize()
    return o


def groupnorm(x0, x1, gamma, beta, eps, silu):
    lib = N.load()
    B, H, W, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[-1]
    y = torch.empty((B, H, W, c0 + c1), dtype=torch.float16, device=x0.device)
    N.check(lib.sdmi_op_groupnorm(N.ptr(x0), N.ptr(x1), int(x0.dtype == torch.float32), c0, c1, B, H * W,
                                  N.ptr(gamma), N.ptr(beta), eps, int(silu), N.ptr(y), N.cur_stream()), "gn")
    torch.cuda.synchronize()
    return y


LAST_STAT = None


def stat_moments(rec):
    """records [images][T][atoms][parts][2] -> (sum, sumsq) per (image, atom) as float64"""
    r = rec.cpu().double().sum(dim=(1, 3))
    return r[..., 0], r[..., 1]


def groupnorm_acc(x0, x1, st0, st1, gamma, beta, eps, silu, atom=10):
    """st0 / st1: (records, T, parts) as igemm's LAST_STAT"""
    lib = N.load()
    B, H, W, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[-1]
    y = torch.empty((B, H, W, c0 + c1), dtype=torch.float16, device=x0.device)
    r1, T1, p1 = st1 if st1 is not None else (None, 0, 0)
    N.check(lib.sdmi_op_groupnorm_acc(N.ptr(x0), N.ptr(x1), int(x0.dtype == torch.float32), c0, c1, B, H * W, N.ptr(st0[0]), st0[1], st0[2],
                                      N.ptr(r1), T1, p1, atom, N.ptr(gamma), N.ptr(beta), eps, int(silu), N.ptr(y), N.cur_stream()), "gn_acc")
    torch.cuda.synchronize()
    return y


def layernorm(x, gamma, beta, eps=1e-5):
    lib = N.load()
    M, Cc = x.shape
    y = torch.empty((M, Cc), dtype=torch.float16, device=x.device)
    N.check(lib.sdmi_op_layernorm(N.ptr(x), int(x.dtype == torch.float32), M, Cc, N.ptr(gamma), N.ptr(beta), eps,
                                  N.ptr(y), N.cur_stream()), "ln")
    torch.cuda.synchronize()
    return y


def gemm_tile(cfg):
    lib = N.load()
    bm, bn = C.c_int(0), C.c_int(0)
    lib.sdmi_gemm_config_dims(cfg, C.byref(bm), C.byref(bn))
    return bm.value, bn.value


def ln_fold_prep(w, gamma, beta, bias, n_rows=None):
    """w: [O][C] cuda fp32/fp16 -> (w_folded fp16 [N][C], g [N], h [N])"""
    lib = N.load()
    O, Cc = w.shape
    n_rows = O if n_rows is None else n_rows
    wo = torch.empty((n_rows, Cc), dtype=torch.float16, device=w.device)
    g = torch.empty((n_rows,), dtype=torch.float32, device=w.device)
    h = torch.empty((n_rows,), dtype=torch.float32, device=w.device)
    code = N.SDMI_F32 if w.dtype == torch.float32 else N.SDMI_F16
    N.check(lib.sdmi_op_ln_fold_prep(N.ptr(w.contiguous()), code, N.ptr(gamma), N.ptr(beta), N.ptr(bias), N.ptr(wo), N.ptr(g),
                                     N.ptr(h), n_rows, Cc, N.cur_stream()), "ln_fold_prep")
    torch.cuda.synchronize()
    return wo, g, h
