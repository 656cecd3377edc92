import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no binaries (they are git-ignored): build them once, as __graft_entry__.build() does
    lib = os.path.join(ROOT, "icet_amd", "lib", "libicet_hip.so")
    ora = os.path.join(ROOT, "oracle", "_build", "libicet_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import subprocess
        if not os.path.exists(lib):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "icet_amd", "csrc"), "-j3"])
        if not os.path.exists(ora):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


def load_pair(name):
    d = np.load(os.path.join(GOLDEN, "scans_%s.npz" % name))
    return d["scan1"], d["scan2"]


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, "golden_%s.npz" % name)))


@pytest.fixture(scope="session")
def frames():
    return load_pair("frame_804_805")


@pytest.fixture(scope="session")
def frames_golden():
    return load_golden("frame_804_805")


@pytest.fixture(scope="session")
def sample_pc():
    return load_pair("sample_pc_1_2")


@pytest.fixture(scope="session")
def sample_pc_golden():
    return load_golden("sample_pc_1_2")


@pytest.fixture(scope="session")
def gpu_ctx():
    import icet_amd
    ctx = icet_amd.Context(0)
    yield ctx
    ctx.close()
