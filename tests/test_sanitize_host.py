"""The host logic of the row tiling and of the group (pwn_tiled.cpp, pwn_group.cpp, pwn_api.cpp) on the CPU: compiled with g++
against a stand-in HIP runtime and stand-in kernels (tools/sanitize/), run under ThreadSanitizer and AddressSanitizer + UBSan.
Every frame of 2..5 members -- blocking calls, frames in flight delivered and resident, moving cuts, a deep band that leaves the
halo (repeat with whole strips), depth that carries over -- must equal the frame of one context; a member that is late past the
deadline comes back as PWN_ETIMEDOUT and the handle recovers; processes over the shared-memory transport.  No GPU needed: the
N > 1 host paths run in the CPU suite."""
import os
import subprocess

import pytest

from conftest import ROOT

OUT = "/tmp/pwn_sanitize_pytest"


def _build(target):
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "tools", "sanitize"), target, "OUT=" + OUT], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]


@pytest.mark.parametrize("target,modes", [("tsan", ("group", "late", "shm")), ("asan", ("all",))])
def test_host_logic_under_sanitizers(target, modes):
    _build(target)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    for m in modes:
        p = subprocess.run([os.path.join(OUT, "group_" + target), m], capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
        out = p.stdout + p.stderr
        assert p.returncode == 0 and out.strip().endswith("ok"), out[-4000:]
        assert "Sanitizer" not in out, out[-4000:]
        if m in ("late", "all"):
            assert "one call failed at the deadline" in out
