"""The C-ABI library loads on a CPU-only box and exports every symbol include/splat.h declares
(no compute calls here).  Creating a context without a GPU must fail loudly — no CPU fallback."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(hooks=False):
    """The entry points include/splat.h declares: the product's (hooks=False), or the ones inside its #ifdef SPLAT_TEST_HOOKS
    block (hooks=True: compiled only into the test build, libsplat_hip_hooks.so)."""
    text = open(os.path.join(ROOT, "include", "splat.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    blocks = re.findall(r"#ifdef SPLAT_TEST_HOOKS(.*?)#endif", text, flags=re.S)
    text = "".join(blocks) if hooks else re.sub(r"#ifdef SPLAT_TEST_HOOKS.*?#endif", "", text, flags=re.S)
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


def test_hooks_live_in_the_test_build_only():
    """splat_debug_* (and the kernel parameters behind them) are compiled under -DSPLAT_TEST_HOOKS only: the
    shipped library exports none of them, the test build (libsplat_hip_hooks.so, built beside it) all of them plus the whole
    product ABI; the Python binding keeps the two sets apart."""
    import __graft_entry__ as g
    from splat_renderer_amd import _lib
    hooks = declared_functions(hooks=True)
    assert hooks and all(n.startswith("splat_debug_") for n in hooks)
    assert sorted(_lib.HOOK_SIGNATURES) == hooks and not set(hooks) & set(_lib.SIGNATURES)
    if not os.path.exists(_lib.HOOKS_LIB_PATH) or not os.path.exists(os.path.join(ROOT, "splat_renderer_amd", "libsplat_hip.so")):
        g.build()
    shipped = C.CDLL(os.path.join(ROOT, "splat_renderer_amd", "libsplat_hip.so"))
    assert [n for n in hooks if hasattr(shipped, n)] == [], "the shipped library exports test hooks"
    test_build = C.CDLL(_lib.HOOKS_LIB_PATH)
    assert [n for n in hooks + declared_functions() if not hasattr(test_build, n)] == []


def test_python_binding_covers_the_header():
    from splat_renderer_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    lib = _lib.load()
    assert lib.splat_abi_version() == 3


def test_napi_addon_has_one_method_per_abi_entry_point():
    """the N-API shim carries one method per export (same name without the prefix) — read from its source, and
    from the built addon itself when Node is here (loading it needs no GPU)."""
    import shutil
    import subprocess
    src = open(os.path.join(ROOT, "splat_renderer_amd", "napi", "splat_napi.c")).read()
    exported = set(re.findall(r"EXPORT\(([a-z0-9_]+)\)", src.split("napi_property_descriptor d[]")[1]))
    wanted = {n[len("splat_"):] for n in declared_functions()}
    assert wanted - exported == set(), f"ABI entry points without an addon method: {sorted(wanted - exported)}"
    assert exported - wanted == set(), f"addon methods that are not ABI entry points: {sorted(exported - wanted)}"
    addon = os.path.join(ROOT, "splat_renderer_amd", "napi", "splat_napi.node")
    if shutil.which("node") and os.path.exists(addon):
        out = subprocess.run(["node", "-e", f"console.log(Object.keys(require({addon!r})).sort().join(' '))"], capture_output=True, text=True,
                             timeout=60)
        assert out.returncode == 0, out.stderr
        assert set(out.stdout.split()) == wanted


def test_no_null_stream_work_in_the_library():
    """A context's stream is non-blocking: nothing orders it against the null stream, and plain hipMemset / hipMemcpy /
    hipMemcpy*Symbol return before a device-side fill or copy has run.  One such call cleared a new sorter's workspace until
    round 4 and raced with the sort that followed (profiles/r04_s_gpu_test_matrix_and_null_stream_memset.txt): every fill and
    copy in csrc/ names the context's stream."""
    import glob
    bad = []
    for path in sorted(glob.glob(os.path.join(ROOT, "splat_renderer_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "splat_renderer_amd", "csrc", "*.h"))):
        text = re.sub(r"//[^\n]*", "", open(path).read())
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\bhip(Memset|Memcpy|MemcpyDtoD|MemcpyHtoD|MemcpyDtoH|MemcpyToSymbol|MemcpyFromSymbol|Memset2D|MemsetD32|MemsetD8)\s*\(", text):
            bad.append((os.path.basename(path), m.group(0)))
    assert not bad, f"null-stream calls: {bad}"


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


def test_missing_rccl_is_an_error_code_not_a_crash():
    """include/splat.h: "when it cannot be loaded these return SPLAT_ERR_COMM (nothing falls back)".  A library that does
    not exist (SPLAT_RCCL_LIB names the one to use) must come back as SPLAT_ERR_COMM with the loader's message — the
    message is built from ONE dlerror() call (a second call returns NULL, and std::string + NULL crashed here)."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "from splat_renderer_amd import _lib\n"
        "lib = _lib.load()\n"
        "buf = C.create_string_buffer(_lib.COMM_ID_BYTES)\n"
        "rc = lib.splat_comm_unique_id(buf)\n"
        "msg = lib.splat_last_error(None)\n"
        "print(rc, (msg or b'').decode())\n" % ROOT)
    env = dict(os.environ, SPLAT_RCCL_LIB="/nonexistent/librccl_not_here.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    rc, msg = out.stdout.strip().split(" ", 1)
    assert int(rc) == -7 and "librccl could not be loaded" in msg and "librccl_not_here" in msg, out.stdout
