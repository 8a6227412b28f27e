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
    for pat in ("nanocall_amd/**/*.py", "nanocall_amd/csrc/*", "nanocall_amd/cli/*", "include/**/*"):
        for f in glob.glob(os.path.join(ROOT, pat), recursive=True):
            if os.path.isfile(f) and not f.endswith((".so", ".o")):
                t = open(f, errors="ignore").read()
                if re.search(r"#\s*include[^\n]*oracle|import\s+nc_oracle|from\s+nc_oracle|libnc_oracle|libnc_ref|dlopen[^\n]*oracle", t):
                    bad.append(f)
    assert not bad, bad


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: include/nanocall_hip.h must compile as C99 (-pedantic) and a C program must link against
    the library and call a device-free entry point."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for h in ("nanocall_hip.h", "nanocall_fast5.h"):
        hdr = os.path.join(root, "include", h)
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", "-I",
                            os.path.join(root, "include"), hdr], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    src = tmp_path / "use_abi.c"
    src.write_text('#include <stdio.h>\n#include "nanocall_hip.h"\n'
                   'int main(void) { unsigned short km[4096]; unsigned n = 0; int rc = nchmm_st_train_kmers(km, &n);\n'
                   '  printf("%d %d %u %s\\n", nchmm_abi_version(), rc, n, nchmm_strerror(NCHMM_E_NO_DEVICE)); return rc; }\n')
    exe = tmp_path / "use_abi"
    libdir = os.path.join(root, "nanocall_amd")
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-L", libdir, "-lnanocall_hip",
                        "-Wl,-rpath," + libdir, "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    ver, rc, n, *msg = r.stdout.split()
    import nanocall_amd as na
    assert int(ver) == 2 and int(rc) == 0 and int(n) == len(na.st_train_kmers())
    assert "no CPU fallback" in r.stdout


def test_reference_call_sites_compile_against_the_mirror_headers(tmp_path):
    """The reference's own call sites of the HMM core (2D round loop nanocall.cpp:360-426, basecall_strand :645-690),
    kept as the reference writes them in tests/boundary/reference_call_sites.cpp, compile (C++11, the reference's
    standard, src/CMakeLists.txt:144) and link against include/nanocall_amd/nanocall_amd.hpp + the library; so does
    the Fast5_Summary mirror.  (Their behaviour is checked on the GPU in tests/test_cpp_layer_gpu.py.)"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    libdir = os.path.join(ROOT, "nanocall_amd")
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-O0", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "boundary", "reference_call_sites.cpp"), "-L", libdir, "-lnanocall_hip",
                        "-Wl,-rpath," + libdir, "-pthread", "-o", str(tmp_path / "rcs")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c++",
                        os.path.join(ROOT, "include", "nanocall_amd", "fast5_summary.hpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(tmp_path / "rcs")], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
