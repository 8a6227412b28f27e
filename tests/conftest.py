import os
import subprocess
import sys

import pytest

# torch first, as bench.py and any torch-based host would have it: its HIP runtime is then the one in the process before the product
# library asks for one.  (The other order -- the product initialises HIP, torch arrives later -- is what
# test_product_before_torch_shares_one_hip_runtime exercises on purpose, in a fresh interpreter of its own.)
try:
    import torch  # noqa: F401
except Exception:      # no torch: the C ABI tests do not need it
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is the checker: build it if its .so is absent (gcc only, ~2 s)
    if not os.path.exists(os.path.join(ROOT, "oracle", "libnc_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True, capture_output=True)
    # the reference's own Kmer.hpp / Builtin_Model.cpp compiled as they lie (only where /root/reference exists)
    if os.path.isdir("/root/reference/src/nanocall") and not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libnc_ref.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=False, capture_output=True)
    # the product library must already be built (python -c 'import __graft_entry__ as g; g.build()');
    # build it here only when hipcc is available and the .so is missing
    so = os.path.join(ROOT, "nanocall_amd", "libnanocall_hip.so")
    if not os.path.exists(so) and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "nanocall_amd", "csrc"), "-j4"], check=True,
                       capture_output=True)


@pytest.fixture(scope="session")
def r73t():
    import nanocall_amd as na
    return na.builtin_model("r73.t")


@pytest.fixture(scope="session")
def r9t():
    import nanocall_amd as na
    return na.builtin_model("r9.t")


@pytest.fixture(scope="session")
def gpu_ctx():
    import nanocall_amd as na
    ctx = na.Context(0)
    yield ctx
    ctx.close()
