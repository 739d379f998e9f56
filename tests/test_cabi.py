"""The C-ABI shared library loads and exports every symbol include/pbr355.h declares.  No compute
call is made here; without a GPU pbr_create must fail loudly, not fall back."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pbr355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pbr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(pkg):
    lib = pkg._lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/pbr355.h but not exported"
    assert set(names) == set(pkg._lib.EXPORTS), set(names) ^ set(pkg._lib.EXPORTS)


def test_struct_layouts_match_header(pkg):
    assert C.sizeof(pkg._lib.FrameUniforms) == 5 * 64 + 16 + 64 + 64 + 16 + 16
    assert C.sizeof(pkg._lib.FrameStats) == 40


def test_no_silent_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the failure path needs a machine without one")
    lib = pkg._lib.load()
    ctx = C.c_void_p()
    rc = lib.pbr_create(0, C.byref(ctx))
    assert rc == pkg._lib.PBR_E_DEVICE and not ctx
    assert b"HIP device" in lib.pbr_last_error(None)
    r = pkg.Renderer(width=64, height=64)
    with pytest.raises(pkg._lib.PbrError):
        r.render()


def test_product_does_not_reference_the_oracle():
    """Nothing under the package may include, import, link or load the CPU oracle."""
    pkg_dir = os.path.join(ROOT, "wgpu-renderer_amd")
    patterns = [r"oracle/", r"liboracle", r"oracle_binding", r"oracle\.h", r"import\s+oracle", r"orc_[a-z_]+\s*\("]
    for base, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".js", ".cc", ".ts", ".gyp")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="replace").read()
                for pat in patterns:
                    assert not re.search(pat, text), f"{os.path.join(base, f)} refers to the oracle ({pat})"
