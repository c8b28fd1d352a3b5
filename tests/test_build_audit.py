"""Build-time audit of the 4-wave MFMA scan kernel (no GPU needed: hipcc cross-compiles).

scan_mfma_w4_kernel names the accumulator registers a[0:255] literally in inline asm.  That is
only sound while hipcc itself never uses an AGPR where the accumulators are live, and never
spills: scripts/audit_w4.py checks the emitted gfx950 assembly for exactly that.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_w4_kernel_accumulator_file_is_untouched_by_the_compiler(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    asm = tmp_path / "kernels_mfma_w4.s"
    # same flags as vrod_amd/csrc/Makefile
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S",
                    "-o", str(asm), os.path.join(ROOT, "vrod_amd", "csrc", "kernels_mfma_w4.hip")], check=True, timeout=900)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import audit_w4
    finally:
        sys.path.pop(0)
    text = asm.read_text()
    assert "scan_mfma_w4_kernel" in text, "the 4-wave kernel is not in the build"
    assert audit_w4.audit(str(asm)) == 0


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_a_build_that_lets_the_compiler_into_the_accumulator_file_leaves_no_library(tmp_path):
    """The audit is a step of the product Makefile: a variant that makes hipcc write an AGPR while the accumulators
    are live (-DVROD_W4_AUDIT_SELFTEST: a value handed to an "a"-constrained asm operand in the tile epilogue, which is
    what round 2's faulting variant did by itself) must fail `make` and leave no .so behind."""
    out = tmp_path / "libvrod_broken.so"
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "vrod_amd", "csrc"), "-j8", f"OUT={out}", f"OBJDIR={tmp_path / 'obj'}",
                        "W4FLAGS=-DVROD_W4_AUDIT_SELFTEST"], capture_output=True, text=True, timeout=1500)
    assert r.returncode != 0, "the broken variant built"
    assert "AUDIT FAILED" in r.stdout + r.stderr
    assert not out.exists()
