"""The C-ABI library loads on a CPU-only box and exports every entry point that include/p3d_hip.h declares
(no compute call is made here)."""
import ctypes
import os
import re

import pytest


def declared_functions(header_path):
    text = open(header_path).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(p3d_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound(pkg):
    lib_mod = pkg._lib
    names = declared_functions(lib_mod.HEADER_PATH)
    assert len(names) >= 25
    assert os.path.exists(lib_mod.LIB_PATH), 'run __graft_entry__.build() first'
    handle = ctypes.CDLL(lib_mod.LIB_PATH)
    for name in names:
        assert hasattr(handle, name), '%s declared in p3d_hip.h but not exported' % name
    assert sorted(lib_mod.SIGNATURES) == names, 'ctypes SIGNATURES and the header disagree'


def test_version_and_error_string(pkg):
    lib = pkg._lib.lib()
    assert lib.p3d_version() == 100
    # a call with a null descriptor must fail cleanly with a message, without touching the GPU
    code = lib.p3d_conv2d_fwd(None, None, None, None, None, None, None, None, 0, None)
    assert code == -1
    assert b'null descriptor' in lib.p3d_last_error()


def test_workspace_queries_are_host_only(pkg):
    lib = pkg._lib.lib()
    d = pkg.ops._desc((64, 256, 16, 16), (256, 256, 3, 3), 1, 1, 1)
    assert lib.p3d_conv2d_wgrad_workspace_bytes(ctypes.byref(d)) >= 256 * 256 * 9 * 4
    slabs, image = 3 * 64 * 256 * 16 * 16 * 4, 256 * 256 * 9 * 6             # stride 1: optional split-K slabs + the pre-split weight image (three bf16 pieces per weight)
    assert lib.p3d_conv2d_dgrad_workspace_bytes(ctypes.byref(d)) in (0, slabs, image, slabs + image)
    d1 = pkg.ops._desc((64, 64, 64, 64), (256, 64, 1, 1), 1, 0, 1)
    assert lib.p3d_conv2d_dgrad_workspace_bytes(ctypes.byref(d1)) in (0, 128 * 256 * 6)              # thousands of blocks: never split; image rows are padded to the 128-row tile
    d2 = pkg.ops._desc((4, 128, 64, 64), (128, 128, 3, 3), 2, 1, 1)
    assert lib.p3d_conv2d_dgrad_workspace_bytes(ctypes.byref(d2)) in (4 * 4 * 128 * 32 * 32 * 4, 128 * 128 * 9 * 6)          # fp32-MFMA class staging | x3 weight image
    assert lib.p3d_fx_act_image_bytes(64, 256, 256) == 3 * 64 * 256 * 256 * 2
    assert lib.p3d_bn_workspace_bytes(64, 256, 256) > 0
    bad = pkg.ops._desc((4, 128, 64, 64), (128, 128, 3, 3), 2, 1, 1)
    bad.Ho = 7
    assert lib.p3d_conv2d_wgrad_workspace_bytes(ctypes.byref(bad)) == 0


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg._lib, '_lib', None)
    monkeypatch.setattr(pkg._lib, 'LIB_PATH', '/nonexistent/libp3d_hip.so')
    with pytest.raises(pkg._lib.P3DError):
        pkg._lib.lib()
