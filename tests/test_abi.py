"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/sdt_gpu.h declares.
No compute calls here (no GPU in this container)."""
import os
import re
import subprocess

import pytest


def header_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "include", "sdt_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sdt_gpu.h but not exported"
    assert sorted(pkg.ABI_SYMBOLS) == syms, "python binding table and header disagree"
    assert lib.sdt_gpu_abi_version() == 8


def test_is_gfx950_code_object(pkg):
    """the fat binary embeds exactly one device target: amdgcn-amd-amdhsa--gfx950"""
    blob = open(pkg.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_no_device_fails_loudly(pkg):
    """no CPU fallback: without a GPU, init must fail with SDT_ENODEV, not silently compute"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.SdtError) as e:
        pkg.PregraphGPU(31)
    assert e.value.code == pkg.SDT_ENODEV


def test_owner_hash_is_host_callable(pkg):
    import numpy as np
    lib = pkg.load_library()
    k = np.array([0x123456789ABCDEF], dtype=np.uint64)
    h1 = lib.sdt_owner_hash(k.ctypes.data, 1)
    k2 = np.array([0x123456789ABCDEE], dtype=np.uint64)
    assert h1 != lib.sdt_owner_hash(k2.ctypes.data, 1)
    assert h1 == lib.sdt_owner_hash(k.ctypes.data, 1)


def test_clamp_K(pkg):
    # pregraph.c:38-59
    assert pkg.clamp_K(24, 31) == 25
    assert pkg.clamp_K(11, 31) == 13
    assert pkg.clamp_K(12, 31) == 13
    assert pkg.clamp_K(63, 31) == 31
    assert pkg.clamp_K(128, 127) == 127
    assert pkg.clamp_K(23, 31) == 23
