"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(nchmm_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_every_declared_symbol_is_exported_and_bound():
    from nanocall_amd._lib import lib, SIGNATURES, LIB_PATH
    L = ctypes.CDLL(LIB_PATH)
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ but not exported"
    # the python binding table covers the same set (no entry point is unreachable from the tests)
    assert set(names) == set(SIGNATURES), set(names) ^ set(SIGNATURES)
    lib()


def test_device_entry_points_fail_loudly_without_a_gpu():
    """No CPU fallback: on a box without a usable device nchmm_create returns NCHMM_E_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import nanocall_amd as na
    with pytest.raises(na.api.NchmmError) as e:
        na.Context(0)
    assert e.value.code == -2


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under nanocall_amd/ or include/ may reference it."""
    bad = []
    for pat in ("nanocall_amd/**/*.py", "nanocall_amd/csrc/*", "include/**/*"):
        for f in glob.glob(os.path.join(ROOT, pat), recursive=True):
            if os.path.isfile(f) and not f.endswith((".so", ".o")):
                t = open(f, errors="ignore").read()
                if re.search(r"#\s*include[^\n]*oracle|import\s+nc_oracle|from\s+nc_oracle|libnc_oracle|libnc_ref|dlopen[^\n]*oracle", t):
                    bad.append(f)
    assert not bad, bad
