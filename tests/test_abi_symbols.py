"""CPU: the C-ABI library loads and exports every symbol include/radiorust_amd.h
declares, and the ctypes table covers exactly that set (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "radiorust_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(rr_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


@pytest.fixture(scope="module")
def built_lib():
    from radiorust_amd import build

    return build.build_library()


def test_header_declares_the_path():
    names = declared_functions()
    for block in ("freqshifter", "filter", "downsampler", "fourier", "chain"):
        assert f"rr_{block}_create" in names
        assert f"rr_{block}_process" in names
        assert f"rr_{block}_process_dev" in names
        assert f"rr_{block}_destroy" in names


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_ctypes_table_matches_header(built_lib):
    from radiorust_amd import _lib

    assert sorted(_lib.SIGNATURES) == declared_functions()
    _lib.lib()  # binds every symbol; AttributeError if one is absent


def test_every_entry_point_cites_the_reference():
    text = open(HEADER).read()
    # each block section names the reference file it replaces
    for ref in ("transform.rs", "filters.rs", "resampling.rs", "analysis.rs", "bufferpool.rs", "math.rs", "windowing.rs"):
        assert ref in text


def test_no_gpu_means_loud_failure(built_lib):
    """Without a device a create must fail with RR_ERR_HIP, never fall back."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import radiorust_amd as rr
    from radiorust_amd._lib import BackendError, RR_ERR_HIP

    with pytest.raises(BackendError) as e:
        rr.FreqShifter.with_shift(700.0)
    assert e.value.status == RR_ERR_HIP


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure; nothing under radiorust_amd/ may
    reference it."""
    pkg = os.path.join(ROOT, "radiorust_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "rr_oracle" not in src and "oracle_np" not in src, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
