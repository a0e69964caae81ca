"""CPU tests of the drop-in boundary: the library builds for gfx950, loads, exports every symbol
include/allwave_hip.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "allwave_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(awv_[a-z_]+)\s*\(", txt)) - {"awv_sink"})


def test_header_symbols_exported(hip_lib):
    from allwave_amd import ffi
    syms = declared_symbols()
    assert set(syms) == set(ffi.EXPORTS)
    for s in syms:
        assert getattr(hip_lib, s) is not None, s
    assert hip_lib.awv_abi_version() == 3


def test_struct_layouts_match_header():
    from allwave_amd import ffi
    assert C.sizeof(ffi.EngineConfig) == 40
    assert C.sizeof(ffi.Penalties) == 28
    assert ffi.PAIR_DTYPE.itemsize == 12
    assert ffi.RESULT_DTYPE.itemsize == 48
    assert C.sizeof(ffi.Stats) == 8 * 36


def test_flag_and_status_constants_match_header():
    """Every AWV_F_* / AWV_ST_* / AWV_ERR_* value the ctypes binding uses is the header's."""
    import re
    from allwave_amd import ffi
    hdr = open(os.path.join(ROOT, "include", "allwave_hip.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(AWV_[A-Z0-9_]+)\s+\(?(-?\d+)\)?", hdr)}
    flags = [k for k in defs if k.startswith("AWV_F_")]
    assert len(flags) >= 6 and "AWV_F_NO_ARENA_PROBE" in flags
    for k in defs:
        if hasattr(ffi, k):
            assert getattr(ffi, k) == defs[k], k
    for k in flags:
        assert hasattr(ffi, k), k


def test_code_object_is_gfx950(hip_lib):
    from allwave_amd import ffi
    blob = open(ffi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"biwfa_align_kernel" in blob


def test_no_gpu_fails_loudly(hip_lib):
    """On a box without a GPU the product path must refuse, not fall back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from allwave_amd import ffi
    with pytest.raises(ffi.EngineError) as ei:
        ffi.Engine()
    assert ei.value.code == ffi.AWV_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "allwave_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "liboracle" not in txt and "awo_" not in txt, f
