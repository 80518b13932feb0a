"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol that
include/pevae_hip.h declares, and the ctypes table matches the header one-to-one (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "pevae_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = re.findall(r"\b(?:int|size_t|const char\*)\s+(pv_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S)
    return {name: [a.strip() for a in args.split(",") if a.strip() and a.strip() != "void"] for name, args in decls}


def test_library_exports_every_declared_symbol():
    from pevae_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `python __graft_entry__.py` first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = header_functions()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), name


def test_ctypes_table_matches_header():
    from pevae_amd import _lib
    decl = header_functions()
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, args in decl.items():
        assert len(args) == len(_lib.SIGNATURES[name][1]), name
    lib = _lib.load()
    assert lib.pv_abi_version() == 2


def test_argument_validation_needs_no_gpu():
    """Bad arguments are rejected on the host before anything is launched."""
    from pevae_amd import _lib
    lib = _lib.load()
    rc = lib.pv_gemm_bf16_nt(None, 64, None, 64, 4, 4, 64, None, 0, None, 0, None, 0, 0, None, None, None)
    assert rc == -1 and b"null operand" in lib.pv_last_error()
    hk, k3, bp, ng = (ctypes.c_int() for _ in range(4))
    assert lib.pv_gru_geometry(256, 512, ctypes.byref(hk), ctypes.byref(k3), ctypes.byref(bp), ctypes.byref(ng)) == 0
    assert (hk.value, k3.value, bp.value, ng.value) == (512, 1536, 256, 8)
    assert lib.pv_gru_geometry(4, 10, None, None, None, None) == -1      # hidden size must be a multiple of 16


def test_product_has_no_cpu_fallback():
    import torch
    from pevae_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.randn(2, 3), torch.randn(4, 3), None)
    for root, _, files in os.walk(os.path.join(ROOT, "pevae_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
