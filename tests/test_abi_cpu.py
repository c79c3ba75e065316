"""The C-ABI library loads on a CPU-only box and exports every symbol include/splat.h declares
(no compute calls here).  Creating a context without a GPU must fail loudly — no CPU fallback."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "splat.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(splat_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_something():
    names = declared_functions()
    assert "splat_project" in names and "splat_composite" in names and len(names) >= 35


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    from splat_renderer_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    lib = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in splat.h but not exported: {missing}"


def test_python_binding_covers_the_header():
    from splat_renderer_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    lib = _lib.load()
    assert lib.splat_abi_version() == 2


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import splat_renderer_amd as sr
    with pytest.raises(sr.SplatError) as ei:
        sr.Device(0)
    assert ei.value.code in (-6, -2)  # NO_DEVICE (or a HIP error from the missing driver)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "splat_renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".js")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)
                assert "liboracle" not in text, os.path.join(dirpath, f)
