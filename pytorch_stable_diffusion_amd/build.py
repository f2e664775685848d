"""Build libsdmi.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libsdmi.so")
SOURCES = ["gemm.hip", "b2b.hip", "attention.hip", "norm.hip", "misc.hip", "unet.hip", "vae.hip", "clip.hip"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


FLAGS_STAMP = os.path.join(LIB_DIR, "obj", "flags.txt")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    # a library built with other SDMI_HIPCC_FLAGS (e.g. the diagnostic -DSDMI_CLK_PROBE build) is stale too
    extra = os.environ.get("SDMI_HIPCC_FLAGS", "")
    if not os.path.exists(FLAGS_STAMP) or open(FLAGS_STAMP).read() != extra:
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "sdmi.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_native(force: bool = False, verbose: bool = True) -> str:
    """Compile every .hip for gfx950 and link libsdmi.so.  hipcc cross-compiles without a GPU."""
    if not force and not is_stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    # -amdgpu-mfma-vgpr-form: keep MFMA accumulators in arch VGPRs (gfx950's register file is unified);
    # removes ~100 v_accvgpr_read/write per attention tile (measured -6% on the S=4096 self-attention)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             "-mllvm", "-amdgpu-mfma-vgpr-form",
             "-Rpass-analysis=kernel-resource-usage"]     # per-kernel VGPR / scratch report -> lib/obj/*.resources.txt
    flags += os.environ.get("SDMI_HIPCC_FLAGS", "").split()
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        remarks = [l for l in out.splitlines() if "[-Rpass-analysis=kernel-resource-usage]" in l]
        with open(os.path.join(obj_dir, src.replace(".hip", ".resources.txt")), "w") as f:
            f.write("\n".join(remarks) + "\n")
        rest = "\n".join(l for l in out.splitlines() if "[-Rpass-analysis=kernel-resource-usage]" not in l)
        if verbose and rest.strip():
            print(rest, file=sys.stderr)
    with open(FLAGS_STAMP, "w") as f:
        f.write(os.environ.get("SDMI_HIPCC_FLAGS", ""))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv))
