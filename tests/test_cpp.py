"""Runs the C++ test binary (tests/cpp/test_core.cc): concurrency stress of the
sampler / readers / control plane, RPC with several clients, parser fuzzing.

The plain build always runs.  ThreadSanitizer and AddressSanitizer+UBSan builds
(`make tsan`, `make asan`) take a few minutes to compile, so they run only when
GPUMON_SANITIZERS=1; both are clean on this tree.  The reference has no race
detection of any kind (no `-race` in its CI: .circleci/config.yml:24-98)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(target, timeout):
    r = subprocess.run(["make", "-j8", target], cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "ALL OK" in r.stdout, tail
    return r.stdout + r.stderr


def test_cpp_core_suite():
    out = _make("test", 900)
    assert "torn=0" in out and "parsers ok" in out and "rpc:" in out


@pytest.mark.skipif(os.environ.get("GPUMON_SANITIZERS") != "1", reason="set GPUMON_SANITIZERS=1 (slow build)")
@pytest.mark.parametrize("target", ["tsan", "asan"])
def test_cpp_core_suite_under_sanitizers(target):
    out = _make(target, 3000)
    assert "ThreadSanitizer" not in out and "AddressSanitizer" not in out and "runtime error" not in out
