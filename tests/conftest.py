import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session", autouse=True)
def native_build():
    """CPU runs build the core on demand; the GPU box uses the in-tree build it was shipped."""
    lib = os.path.join(ROOT, "build", "libgpumon.so")
    if not _has_gpu() or not os.path.exists(lib):
        subprocess.run(["make", "-j8", "all"], cwd=ROOT, check=True, stdout=subprocess.DEVNULL)
    yield


@pytest.fixture
def engine():
    """A fresh embedded engine on an 8-GPU fake node."""
    from k8s_gpu_monitor_amd import engine as E
    E.init(backend="fake", backend_spec="gpus=8,seed=1")
    try:
        yield E
    finally:
        E.shutdown()


@pytest.fixture
def engine2():
    from k8s_gpu_monitor_amd import engine as E
    E.init(backend="fake", backend_spec="gpus=2,seed=7")
    try:
        yield E
    finally:
        E.shutdown()


def bin_path(name):
    return os.path.join(ROOT, "build", "bin", name)
