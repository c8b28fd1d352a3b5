"""CPU tests of the drop-in boundary: the C-ABI library loads, exports exactly what
include/vrod.h declares, and fails loudly (no CPU fallback) without a gfx950 device."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "vrod.h")


def declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vrod_[a-z_0-9]+)\s*\(", src)))


def test_header_and_loader_agree():
    import vrod_amd
    assert declared_symbols() == sorted(vrod_amd.SYMBOLS)


def test_library_exports_every_declared_symbol():
    import vrod_amd
    lib = vrod_amd.load()
    for s in declared_symbols():
        assert hasattr(lib, s), s
    out = subprocess.run(["nm", "-D", "--defined-only", vrod_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (vrod_[a-z_0-9]+)", out))
    assert exported == set(declared_symbols())
    assert b"gfx950" in lib.vrod_version()


def test_library_is_a_gfx950_code_object():
    import vrod_amd
    blob = open(vrod_amd.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob


def test_no_device_means_error_not_fallback():
    import torch
    import vrod_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(vrod_amd.VrodError) as e:
        vrod_amd.Index(16, "f32", "cosine")
    assert e.value.code == 3 and "no CPU fallback" in str(e.value)


def test_argument_validation_without_device():
    import vrod_amd
    L = vrod_amd.load()
    h = C.c_void_p()
    assert L.vrod_index_create(C.byref(h), 0, 0, 0, None, 0) == 1        # dim 0
    assert L.vrod_index_create(C.byref(h), 8, 7, 0, None, 0) == 1        # bad dtype
    assert L.vrod_index_create(C.byref(h), 8, 0, 9, None, 0) == 1        # bad metric
    assert L.vrod_index_create(None, 8, 0, 0, None, 0) == 1              # null out
    assert L.vrod_index_count(None, None) == 1
    assert L.vrod_search(None, None, 1, 1, None, None) == 1
    assert L.vrod_index_destroy(None) == 0
    assert L.vrod_last_error()  # message text is set


def test_missing_library_raises(monkeypatch, tmp_path):
    import vrod_amd._lib as lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "libvrod_hip.so"))
    with pytest.raises(ImportError):
        lib.load()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under vrod_amd/ may reference it."""
    pkg = os.path.join(ROOT, "vrod_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert '#include "vrod_oracle.h"' not in txt, os.path.join(dp, f)
                # the one run-time binding in the product is librccl (multi-device handles): the dlopen call
                # takes its names from a list of rccl names and nothing in the product names the oracle
                if "dlopen" in txt:
                    assert f == "vrod_index.hip", os.path.join(dp, f)
                    assert txt.count("dlopen(") == 1 and "oracle" not in txt.replace("the CPU oracle", "").replace("oracle order", "").replace("oracle's", ""), os.path.join(dp, f)
                    names = re.search(r"const char\* names\[\] = \{([^}]*)\}", txt).group(1)
                    assert "rccl" in names.lower() and "oracle" not in names.lower()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)
                assert "libvrod_oracle" not in txt, os.path.join(dp, f)


def test_rust_binding_crate_declares_exactly_the_abi():
    """bindings/rust (source only: no rustc in the image) must stay in sync with include/vrod.h."""
    src = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    ext = src[src.index('extern "C" {'):]
    ext = ext[:ext.index("\n}\n")]
    rust = sorted(set(re.findall(r"pub fn (vrod_[a-z_0-9]+)\s*\(", ext)))
    assert rust == declared_symbols()
