"""The built library contains no packed-fp32 VALU instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) in any kernel.
On MI355X such an instruction sporadically gives the upper lanes of a wave a wrong result while another stream's MFMA kernel shares the SIMD (found with
tools/debug_det3.py: the max-pool backward dropped a window's contribution in ~10 of 4 M elements, only with the weight-gradient stream busy), so
csrc/Makefile builds with -fno-slp-vectorize -fno-vectorize; this test disassembles every embedded gfx950 code object to make sure the flags held."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_packed_fp32_instructions_in_any_kernel():
    spec = importlib.util.spec_from_file_location('scan_packed', os.path.join(ROOT, 'tools', 'scan_packed.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip('llvm-objdump not available')
    if not os.path.exists(mod.SO):
        pytest.skip('library not built')
    found, kernels = mod.scan()
    assert kernels > 100                      # the scan saw the code objects
    assert not found, found
