"""C ABI surface (CPU, no GPU calls): the in-tree library loads and exports every symbol include/sdmi.h declares."""
import ctypes
import os
import re

from pytorch_stable_diffusion_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "sdmi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sdmi_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"libsdmi.so does not export {s}"
    assert lib.sdmi_version() >= 100
    assert isinstance(lib.sdmi_last_error(), bytes)


def test_binding_covers_header():
    missing = [s for s in _header_symbols() if s not in _native.EXPORTED_SYMBOLS]
    assert not missing, f"_native.py has no signature for {missing}"


def test_no_oracle_import_in_product():
    """The product package must never import the oracle (test infrastructure only)."""
    pkg = os.path.join(ROOT, "pytorch_stable_diffusion_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_no_kernel_spills_to_scratch():
    """Every gfx950 kernel must fit its registers: a spill (private-memory scratch) in a GEMM/attention epilogue or
    main loop is a silent slowdown.  The build stores hipcc's per-kernel resource report next to the objects."""
    import glob
    import pytest
    _native.load()
    files = glob.glob(os.path.join(ROOT, "pytorch_stable_diffusion_amd", "lib", "obj", "*.resources.txt"))
    if not files:
        pytest.skip("prebuilt library without resource reports")
    n_kernels, bad = 0, []
    for fn in files:
        name = None
        for line in open(fn):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                n_kernels += 1
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and int(m.group(1)) > 0:
                bad.append((os.path.basename(fn), name, int(m.group(1))))
    assert n_kernels >= 60, n_kernels
    assert not bad, f"kernels spilling to scratch: {bad}"
