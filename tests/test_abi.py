"""CPU checks of the drop-in boundary: libmm_hip.so loads without a GPU and exports exactly the symbols
include/mm_hip.h declares; the ctypes table matches the header's argument counts."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:int|size_t|const char\*)\s+(mm_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


def test_header_symbols_are_exported_and_bound():
    from multimodal_amd import _lib
    funcs = header_functions()
    assert len(funcs) >= 20
    lib = _lib.get_lib()  # loads without a GPU; raises if the library was not built
    for name, nargs in funcs.items():
        assert hasattr(lib, name), f"{name} declared in mm_hip.h but not exported by libmm_hip.so"
        assert name in _lib.PROTOTYPES, f"{name} has no ctypes prototype"
        assert len(_lib.PROTOTYPES[name][1]) == nargs, f"{name}: header has {nargs} args, ctypes table {len(_lib.PROTOTYPES[name][1])}"
    assert set(_lib.PROTOTYPES) == set(funcs), "ctypes table lists symbols the header does not declare"
    assert lib.mm_target_arch() == b"gfx950"
    # developer scaffolding stays out of the product: no mm_debug_* export, no environment lookup in the library
    for name in _lib.DEV_PROTOTYPES:
        assert not hasattr(lib, name), f"{name} belongs to libmm_hip_dev.so only"
    assert b"MM_GEMMPP" not in open(_lib.LIB_PATH, "rb").read(), "the product library must not read kernel-forcing switches"


def test_developer_library_is_a_superset():
    from multimodal_amd import _lib
    if not os.path.exists(_lib.DEV_LIB_PATH):
        pytest.skip("libmm_hip_dev.so not built (make -C multimodal_amd/csrc DEV=1)")
    dev = _lib.use_dev_lib()._dev
    for name in list(_lib.PROTOTYPES) + list(_lib.DEV_PROTOTYPES):
        assert hasattr(dev, name), name


def test_host_only_entry_points_and_argument_errors():
    from multimodal_amd import _lib
    lib = _lib.get_lib()
    # reference formula with the wave size a ROCm build sees (scaled_masked_softmax.h:338-352, C10_WARP_SIZE = 64)
    assert lib.mm_softmax_batch_per_block(2048, 2048, 4, 16) == 2
    assert lib.mm_softmax_batch_per_block(64, 64, 4, 16) == 4
    assert lib.mm_softmax_batch_per_block(32, 32, 4, 16) == 8
    assert lib.mm_softmax_batch_per_block(128, 128, 4, 16) == 4
    # argument validation happens before any device work: safe to exercise on CPU
    rc = lib.mm_softmax_causal_fwd(None, None, 1.0, 1, 8, _lib.MM_BF16, None)
    assert rc == -1 and b"null" in lib.mm_last_error()
    rc = lib.mm_gemm_nt(ctypes.c_void_p(16), None, ctypes.c_void_p(16), ctypes.c_void_p(16), None, None, 8, 8, 64, 32, 64, 8, 0, 0, None)
    assert rc == -1 and b"leading dimensions" in lib.mm_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc, "gemm")


def test_product_path_has_no_oracle_or_cpu_fallback():
    """The package must not import the oracle, and ops must refuse CPU tensors."""
    import torch
    pkg = os.path.join(ROOT, "multimodal_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), f"{f} mentions the oracle"
    from multimodal_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.softmax_causal_fwd(torch.zeros(1, 8, 8, dtype=torch.bfloat16), 1.0)
