import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the HIP library and the oracle's C checker once,
    exactly as ``__graft_entry__.build()`` does (hipcc cross-compiles gfx950 without a GPU).  On the GPU box the prebuilt
    files travel with the snapshot and nothing happens here."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "idvs", "morec_amd", "libmorec_hip.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "idvs", "morec_amd", "csrc"), "-j", "8"])
    ref = os.path.join(ROOT, "oracle", "libmorec_oracle_ref.so")
    if not os.path.exists(ref) and shutil.which("gcc") and os.path.exists(os.path.join(ROOT, "oracle", "Makefile")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
