"""ctypes binding of libsdmi.so (include/sdmi.h).  PyTorch supplies device memory and the current
stream only; every kernel on the hot path lives in the library.  There is NO fallback: if the
library cannot be loaded the import error propagates."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import build as _build

SDMI_F32, SDMI_F16 = 0, 1
FLAG_STREAM_F32, FLAG_PARTIAL, FLAG_NO_TUNE, FLAG_ACCURATE = 1, 2, 4, 8


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data_dev", C.c_void_p), ("dtype", C.c_int), ("ndim", C.c_int),
                ("shape", C.c_int64 * 4)]


class GemmDesc(C.Structure):
    _fields_ = [("a0", C.c_void_p), ("a1", C.c_void_p),
                ("c0", C.c_int), ("c1", C.c_int), ("hs", C.c_int), ("ws", C.c_int), ("ho", C.c_int),
                ("wo", C.c_int), ("ups", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("ks", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("w", C.c_void_p), ("bias", C.c_void_p),
                ("res", C.c_void_p), ("res_f32", C.c_int), ("ldr", C.c_int),
                ("out", C.c_void_p), ("out_f32", C.c_int), ("ldc", C.c_int),
                ("out16", C.c_void_p),
                ("out_t", C.c_void_p), ("nt0", C.c_int), ("S", C.c_int), ("ldt", C.c_int),
                ("cfg", C.c_int), ("ksplit", C.c_int),
                ("x0", C.c_void_p), ("x1", C.c_void_p), ("cx0", C.c_int), ("cx1", C.c_int),
                ("rowstat", C.c_void_p), ("ln_stat", C.c_void_p), ("ln_ntn", C.c_int), ("ln_g", C.c_void_p),
                ("ln_c", C.c_int), ("ln_eps", C.c_float), ("out_t_perm", C.c_int),
                ("act", C.c_int), ("sm_valid", C.c_int), ("img_rows", C.c_int), ("w_img_stride", C.c_int),
                ("vec_img_stride", C.c_int), ("ldw", C.c_int), ("phase2", C.c_int),
                ("ln_ksteps", C.c_int), ("ln_out", C.c_void_p),
                ("gacc", C.c_void_p), ("gacc_atom", C.c_int), ("gacc_rows_img", C.c_int),
                ("ln_guard", C.c_void_p), ("ln_guard_sigma", C.c_float),
                ("gna_rec", C.c_void_p), ("gna_gamma", C.c_void_p), ("gna_beta", C.c_void_p), ("gna_eps", C.c_float),
                ("gna_t", C.c_int), ("gna_parts", C.c_int), ("gna_atom", C.c_int), ("gna_rows", C.c_int),
                ("a0f", C.c_void_p), ("a1f", C.c_void_p), ("x0f", C.c_void_p), ("x1f", C.c_void_p), ("accurate", C.c_int),
                ("hgn_x0", C.c_void_p), ("hgn_x1", C.c_void_p), ("hgn_in_f32", C.c_int), ("hgn_c0", C.c_int), ("hgn_c1", C.c_int),
                ("hgn_gamma", C.c_void_p), ("hgn_beta", C.c_void_p), ("hgn_eps", C.c_float), ("hgn_silu", C.c_int),
                ("hgn_rec0", C.c_void_p), ("hgn_rec1", C.c_void_p), ("hgn_t0", C.c_int), ("hgn_t1", C.c_int), ("hgn_p0", C.c_int),
                ("hgn_p1", C.c_int), ("hgn_atom", C.c_int)]


class B2bDesc(C.Structure):
    _fields_ = [("a1", C.c_void_p), ("lda1", C.c_int), ("w1", C.c_void_p), ("b1", C.c_void_p),
                ("r1", C.c_void_p), ("r1_f32", C.c_int), ("s32", C.c_void_p), ("s16", C.c_void_p),
                ("w2", C.c_void_p), ("K2", C.c_int), ("h2", C.c_void_p), ("partial", C.c_int), ("cscale", C.c_float),
                ("r2", C.c_void_p), ("r2_f32", C.c_int), ("out", C.c_void_p), ("out_f32", C.c_int), ("out16", C.c_void_p),
                ("M", C.c_int), ("eps", C.c_float), ("bm", C.c_int),
                ("npass2", C.c_int), ("ldo", C.c_int), ("vt", C.c_void_p), ("S", C.c_int), ("ldt", C.c_int),
                ("gx", C.c_void_p), ("gx_f32", C.c_int), ("gn_partial", C.c_void_p), ("gn_nchunk", C.c_int),
                ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p), ("gn_eps", C.c_float),
                ("gacc", C.c_void_p), ("gacc_atom", C.c_int), ("gacc_rows_img", C.c_int)]


_LIB: Optional[C.CDLL] = None

_SIGNATURES = {
    "sdmi_last_error": (C.c_char_p, []),
    "sdmi_version": (C.c_int, []),
    "sdmi_unet_create": (C.c_int, [C.POINTER(TensorDesc), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sdmi_unet_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "sdmi_unet_destroy": (None, [C.c_void_p]),
    "sdmi_unet_set_context": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sdmi_unet_set_schedule": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "sdmi_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p]),
    "sdmi_cfg_ddpm_step": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_float),
                                     C.c_int64, C.c_void_p, C.c_void_p]),
    "sdmi_unet_denoise_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                         C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p]),
    "sdmi_unet_denoise_step_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                               C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p]),
    "sdmi_unet_ln_guard": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p]),
    "sdmi_unet_run_block": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdmi_unet_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "sdmi_unet_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "sdmi_unet_last_launch_count": (C.c_int, [C.c_void_p]),
    "sdmi_unet_weight_bytes": (C.c_int64, [C.c_void_p]),
    "sdmi_unet_tuned_shapes": (C.c_int, [C.c_void_p]),
    "sdmi_unet_device": (C.c_int, [C.c_void_p]),
    "sdmi_library_hash": (C.c_uint64, []),
    "sdmi_unet_arena": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sdmi_vae_decoder_create": (C.c_int, [C.POINTER(TensorDesc), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sdmi_vae_destroy": (None, [C.c_void_p]),
    "sdmi_vae_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sdmi_vae_last_launch_count": (C.c_int, [C.c_void_p]),
    "sdmi_vae_encoder_create": (C.c_int, [C.POINTER(TensorDesc), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sdmi_vae_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sdmi_clip_create": (C.c_int, [C.POINTER(TensorDesc), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sdmi_clip_destroy": (None, [C.c_void_p]),
    "sdmi_clip_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "sdmi_clip_last_launch_count": (C.c_int, [C.c_void_p]),
    "sdmi_op_gemm": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "sdmi_bench_gemm": (C.c_int, [C.POINTER(GemmDesc), C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "sdmi_op_b2b": (C.c_int, [C.POINTER(B2bDesc), C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "sdmi_gemm_num_configs": (C.c_int, []),
    "sdmi_gemm_config_name": (C.c_char_p, [C.c_int]),
    "sdmi_op_pack_conv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sdmi_op_pack_ups_phase": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sdmi_op_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sdmi_op_groupnorm_slab": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdmi_op_groupnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "sdmi_op_groupnorm_acc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "sdmi_op_gemm_stat_layout": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sdmi_op_layernorm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                    C.c_void_p, C.c_void_p]),
    "sdmi_gemm_config_dims": (None, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sdmi_gn_num_chunks": (C.c_int, [C.c_int]),
    "sdmi_op_gn_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sdmi_op_ln_fold_prep": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())


def lib_path() -> str:
    """In-tree library; ``SDMI_LIB`` may point at another build of the same ABI (A/B benchmarking)."""
    return os.environ.get("SDMI_LIB") or _build.LIB


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load libsdmi.so (building it with hipcc first when it is absent/stale and hipcc exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # Kernel arguments in DEVICE memory: a step is 244 dependent launches whose first instructions read their 100-360 byte
    # argument blocks; fetched from host-coherent memory every one of them pays a PCIe round trip (measured on one MI355X,
    # same box and binary: 236.8 steps/s with HIP_FORCE_DEV_KERNARG=0 against 260.2 with 1, profiles/r05_kernarg_placement.json).
    # The HIP runtime reads the variable when it initialises, i.e. at the first HIP call of the process; an explicit setting wins.
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    path = lib_path()
    if build_if_missing and not os.environ.get("SDMI_LIB") and _build.is_stale():
        try:
            _build.build_native(verbose=False)
        except Exception as exc:  # no toolchain on this box and no prebuilt library
            if not os.path.exists(path):
                raise ImportError(f"libsdmi.so is not built and could not be built: {exc}") from exc
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def library_hash() -> str:
    """16 hex digits: FNV-1a 64 of the loaded libsdmi.so (the plan cache's key; profiles/*.json are stamped with it)."""
    return f"{load().sdmi_library_hash():016x}"


class SdmiError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc == 0:
        return
    msg = load().sdmi_last_error().decode("utf-8", "replace")
    if rc == -22:
        raise ValueError(f"{what}: {msg}")
    if rc == -2:
        raise KeyError(f"{what}: {msg}")
    raise SdmiError(f"{what}: rc={rc}: {msg}")


def cur_stream(device=None) -> C.c_void_p:
    """The current torch stream of ``device`` (default: the current device) as a hipStream_t."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _state_device(state: Dict[str, torch.Tensor]) -> torch.device:
    devs = {v.device for v in state.values()}
    if len(devs) != 1 or next(iter(devs)).type != "cuda":
        raise ValueError(f"weights must all live on ONE cuda device, got {sorted(map(str, devs))}")
    return next(iter(devs))


class _DeviceBound:
    """Base of the native handles.  A handle's packed weights, arena and plans live on ONE device (the device of
    its weights); every native call runs with that device current and on that device's current torch stream, and
    every tensor argument must live there -- otherwise the kernels would be launched on another GPU's stream with
    pointers it cannot reach (a memory fault, not an error code).  The library re-checks the current device
    itself (Engine::enter) and returns -22 on a mismatch."""

    _dev: torch.device

    def _guard(self):
        return torch.cuda.device(self._dev)

    def _stream(self) -> C.c_void_p:
        return cur_stream(self._dev)

    def _on_dev(self, *tensors):
        for t in tensors:
            if t is not None and t.device != self._dev:
                raise ValueError(f"tensor on {t.device} passed to a native handle that lives on {self._dev}")


def ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return SDMI_F32
    if t.dtype == torch.float16:
        return SDMI_F16
    raise TypeError(f"unsupported weight dtype {t.dtype} (need float32 or float16)")


def _tensor_descs(state: Dict[str, torch.Tensor]):
    if not state:
        raise ValueError("empty state dict")
    descs = (TensorDesc * len(state))()
    keep = []
    for i, (k, v) in enumerate(state.items()):
        if not v.is_cuda:
            raise ValueError(f"weight '{k}' is not on the GPU")
        v = v.contiguous()
        kb = k.encode()
        keep += [v, kb]
        descs[i].name = kb
        descs[i].data_dev = v.data_ptr()
        descs[i].dtype = _dtype_code(v)
        descs[i].ndim = v.dim()
        for j, s in enumerate(v.shape):
            descs[i].shape[j] = s
    return descs, keep


class VaeDecoderHandle(_DeviceBound):
    """Owns one native sdmi_vae (VAE decoder, or encoder with ``encoder=True``)."""

    def __init__(self, state: Dict[str, torch.Tensor], flags: int = FLAG_STREAM_F32, encoder: bool = False):
        lib = load()
        descs, keep = _tensor_descs(state)
        self._dev = _state_device(state)
        h = C.c_void_p()
        with self._guard():
            torch.cuda.synchronize(self._dev)
            if encoder:
                check(lib.sdmi_vae_encoder_create(descs, len(state), flags, C.byref(h)), "sdmi_vae_encoder_create")
            else:
                check(lib.sdmi_vae_decoder_create(descs, len(state), flags, C.byref(h)), "sdmi_vae_decoder_create")
        del keep
        self._h, self._lib = h, lib

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdmi_vae_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, image: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        assert image.is_cuda and image.dtype == torch.float32 and image.dim() == 4 and image.shape[1] == 3
        self._on_dev(image)
        image, noise = image.contiguous(), noise.to(image.device, torch.float32).contiguous()
        b, _, hh, ww = image.shape
        assert tuple(noise.shape) == (b, 4, hh // 8, ww // 8)
        with self._guard():
            out = torch.empty((b, 4, hh // 8, ww // 8), dtype=torch.float32, device=self._dev)
            check(self._lib.sdmi_vae_encode(self._h, ptr(image), ptr(noise), ptr(out), b, hh, ww, self._stream()), "sdmi_vae_encode")
        return out

    def decode(self, latents: torch.Tensor) -> torch.Tensor:
        assert latents.is_cuda and latents.dtype == torch.float32 and latents.dim() == 4 and latents.shape[1] == 4
        self._on_dev(latents)
        latents = latents.contiguous()
        b, _, h, w = latents.shape
        with self._guard():
            out = torch.empty((b, 3, 8 * h, 8 * w), dtype=torch.float32, device=self._dev)
            check(self._lib.sdmi_vae_decode(self._h, ptr(latents), ptr(out), b, h, w, self._stream()), "sdmi_vae_decode")
        return out

    @property
    def last_launch_count(self) -> int:
        return self._lib.sdmi_vae_last_launch_count(self._h)


class ClipHandle(_DeviceBound):
    """Owns one native sdmi_clip (CLIP text encoder)."""

    def __init__(self, state: Dict[str, torch.Tensor], flags: int = 0):
        lib = load()
        descs, keep = _tensor_descs(state)
        self._dev = _state_device(state)
        h = C.c_void_p()
        with self._guard():
            torch.cuda.synchronize(self._dev)
            check(lib.sdmi_clip_create(descs, len(state), flags, C.byref(h)), "sdmi_clip_create")
        del keep
        self._h, self._lib = h, lib

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdmi_clip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, tokens: torch.Tensor) -> torch.Tensor:
        assert tokens.is_cuda and tokens.dtype == torch.int64 and tokens.dim() == 2 and tokens.shape[1] == 77
        self._on_dev(tokens)
        tokens = tokens.contiguous()
        with self._guard():
            out = torch.empty((tokens.shape[0], 77, 768), dtype=torch.float32, device=self._dev)
            check(self._lib.sdmi_clip_encode(self._h, ptr(tokens), ptr(out), tokens.shape[0], self._stream()), "sdmi_clip_encode")
        return out

    @property
    def last_launch_count(self) -> int:
        return self._lib.sdmi_clip_last_launch_count(self._h)


class UNetHandle(_DeviceBound):
    """Owns one native sdmi_unet.  ``state`` maps the reference's state-dict keys to CUDA tensors."""

    def __init__(self, state: Dict[str, torch.Tensor], flags: int = 0):
        lib = load()
        descs, keep = _tensor_descs(state)
        self._dev = _state_device(state)
        h = C.c_void_p()
        with self._guard():
            torch.cuda.synchronize(self._dev)
            check(lib.sdmi_unet_create(descs, len(state), flags, C.byref(h)), "sdmi_unet_create")
        del keep
        self._h = h
        self._lib = lib
        self.flags = flags

    def clone(self) -> "UNetHandle":
        """A second lane over the same packed weights (own arena / slabs / context): drive it on another stream to run a
        second, independent denoising loop concurrently.  The clone keeps this handle alive."""
        other = UNetHandle.__new__(UNetHandle)
        other._dev, other._lib, other.flags = self._dev, self._lib, self.flags
        other._parent = self
        self._children = getattr(self, "_children", 0) + 1
        h = C.c_void_p()
        with self._guard():
            check(self._lib.sdmi_unet_clone(self._h, C.byref(h)), "sdmi_unet_clone")
        other._h = h
        return other

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_children", 0) > 0:
                raise RuntimeError("UNetHandle.close(): lanes cloned from this handle are still open (they borrow its weights)")
            self._lib.sdmi_unet_destroy(self._h)
            self._h = None
            parent = getattr(self, "_parent", None)
            if parent is not None:
                parent._children -= 1
                self._parent = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_context(self, ctx: torch.Tensor):
        assert ctx.is_cuda and ctx.dtype == torch.float32 and ctx.dim() == 3 and ctx.shape[2] == 768
        self._on_dev(ctx)
        ctx = ctx.contiguous()
        with self._guard():
            check(self._lib.sdmi_unet_set_context(self._h, ptr(ctx), ctx.shape[0], ctx.shape[1], self._stream()),
                  "sdmi_unet_set_context")

    def set_schedule(self, temb: torch.Tensor):
        assert temb.is_cuda and temb.dtype == torch.float32 and temb.dim() == 2 and temb.shape[1] == 320
        self._on_dev(temb)
        temb = temb.contiguous()
        with self._guard():
            check(self._lib.sdmi_unet_set_schedule(self._h, ptr(temb), temb.shape[0], self._stream()),
                  "sdmi_unet_set_schedule")

    def forward(self, latents: torch.Tensor, batch: int, temb: Optional[torch.Tensor] = None,
                step_idx: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert latents.is_cuda and latents.dtype == torch.float32 and latents.dim() == 4 and latents.shape[1] == 4
        self._on_dev(latents, temb, out)
        latents = latents.contiguous()
        lb, _, h, w = latents.shape
        if temb is not None:
            assert temb.is_cuda and temb.dtype == torch.float32 and temb.numel() == 320
            temb = temb.contiguous()
        with self._guard():
            if out is None:
                out = torch.empty((batch, 4, h, w), dtype=torch.float32, device=self._dev)
            check(self._lib.sdmi_unet_forward(self._h, ptr(latents), lb, ptr(temb), step_idx, ptr(out), batch, h, w,
                                              self._stream()), "sdmi_unet_forward")
        return out

    def denoise_step(self, latents: torch.Tensor, step_idx: int, do_cfg: bool, cfg_scale: float,
                     noise: Optional[torch.Tensor], coef):
        """latents (P,4,h,w), updated in place; P > 1: P prompts through one chain (the context set before has batch 2P -- P without
        guidance -- in the order cat([cond_0..cond_P-1, uncond_0..uncond_P-1]); noise (P,4,h,w) or None)."""
        self._on_dev(latents, noise)
        n_prompts, _, h, w = latents.shape
        assert latents.is_contiguous() and (noise is None or (noise.is_contiguous() and noise.shape == latents.shape))
        c = (C.c_float * 5)(*[float(x) for x in coef])
        with self._guard():
            check(self._lib.sdmi_unet_denoise_step_batch(self._h, ptr(latents), n_prompts, step_idx, int(do_cfg), float(cfg_scale),
                                                         ptr(noise), c, h, w, self._stream()), "sdmi_unet_denoise_step_batch")

    def ln_guard(self, reset: bool = True, fold_on: Optional[bool] = None) -> int:
        """Rows beyond the LayerNorm-fold guard's threshold since the last reset (synchronises the stream); fold_on switches
        the handle between the folded GEMMs and the separate LayerNorm kernel."""
        hits = C.c_int(0)
        with self._guard():
            check(self._lib.sdmi_unet_ln_guard(self._h, C.byref(hits), int(reset), -1 if fold_on is None else int(fold_on),
                                               self._stream()), "sdmi_unet_ln_guard")
        return hits.value

    def run_block(self, prefix: str, kind: int, x0: torch.Tensor, x1: Optional[torch.Tensor] = None,
                  time: Optional[torch.Tensor] = None, arg: int = 1, out_shape=None) -> torch.Tensor:
        """x0/x1: NHWC fp32 CUDA tensors (B,H,W,C).  Returns NHWC fp32 (kind 4: NCHW (B,4,H,W))."""
        self._on_dev(x0, x1, time)
        B, H, W, c0 = x0.shape
        c1 = 0 if x1 is None else x1.shape[3]
        with self._guard():
            out = torch.empty(out_shape, dtype=torch.float32, device=self._dev)
            check(self._lib.sdmi_unet_run_block(self._h, prefix.encode(), kind, arg, ptr(x0.contiguous()), c0,
                                                ptr(None if x1 is None else x1.contiguous()), c1, B, H, W,
                                                ptr(time), ptr(out), self._stream()), "sdmi_unet_run_block")
        return out

    def profile(self, enable: bool):
        check(self._lib.sdmi_unet_profile(self._h, int(enable)), "sdmi_unet_profile")

    def profile_read(self):
        ms = (C.c_double * 4)()
        fl = (C.c_double * 4)()
        nl = (C.c_int * 4)()
        check(self._lib.sdmi_unet_profile_read(self._h, ms, fl, nl), "sdmi_unet_profile_read")
        names = ("mfma", "attention", "norm", "finalize")
        return {n: dict(ms=ms[i], flops=fl[i], launches=nl[i]) for i, n in enumerate(names)}

    @property
    def last_launch_count(self) -> int:
        return self._lib.sdmi_unet_last_launch_count(self._h)

    @property
    def weight_bytes(self) -> int:
        return self._lib.sdmi_unet_weight_bytes(self._h)

    @property
    def tuned_shapes(self) -> int:
        """GEMM shapes this handle had to time itself (0 = every plan came from the shipped table / the cache)."""
        return self._lib.sdmi_unet_tuned_shapes(self._h)

    @property
    def device_index(self) -> int:
        return self._lib.sdmi_unet_device(self._h)

    def arena(self):
        """(capacity, high-water mark) of the handle's activation arena in bytes."""
        cap, peak = C.c_int64(0), C.c_int64(0)
        check(self._lib.sdmi_unet_arena(self._h, C.byref(cap), C.byref(peak)), "sdmi_unet_arena")
        return cap.value, peak.value


def cfg_ddpm_step(eps, do_cfg, cfg_scale, latents, noise, coef, eps_out=None):
    lib = load()
    c = (C.c_float * 5)(*[float(x) for x in coef])
    n = latents.numel()
    for t in (eps, noise, eps_out):
        if t is not None and t.device != latents.device:
            raise ValueError(f"cfg_ddpm_step: tensor on {t.device}, latents on {latents.device}")
    with torch.cuda.device(latents.device):
        check(lib.sdmi_cfg_ddpm_step(ptr(eps), int(do_cfg), float(cfg_scale), ptr(latents), ptr(noise), c, n,
                                     ptr(eps_out), cur_stream(latents.device)), "sdmi_cfg_ddpm_step")
