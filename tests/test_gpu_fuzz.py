"""A short run of the randomised parity campaign (tools/fuzz_parity.py: random
levels, cameras with and without w components, sphere sets, frame sizes, blur
on/off, counters on) as part of the GPU suite.  The campaign is what found the
two toolchain problems recorded in pwnfps_amd/csrc/Makefile and dev_math.h; the
second run sends w-free cameras through the general 4-lane kernel variant, whose
output must not depend on which variant renders a frame."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,seed,force_w,sched", [(120, 3, False, "units"), (80, 4, True, "units"),
                                                   (120, 5, False, "refill"), (80, 6, True, "refill")])
def test_fuzz_campaign(n, seed, force_w, sched):
    """both trace schedulers (PWN_OPT_SCHEDULER: 16x4 units in step / ballot + prefix refill)"""
    env = dict(os.environ)
    env["PWN_SCHEDULER"] = sched
    if sched == "refill":
        env["PWN_REFILL_LIMIT"] = str(1 + (seed * 37) % 400)
    if force_w:
        env["PWN_DBG_FORCE_HASW"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), str(n), str(seed)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "%d scenes, 0 mismatches" % n in p.stdout


def test_group_campaign():
    """tools/fuzz_group.py, a short run: pwn_init_multi with 2..7 members on device 0 -- blocking calls with depth carried over 20 calls
    while the cuts move, frames in flight delivered and resident -- against the same calls on one context, colour and depth"""
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_group.py"), "18", "5150"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and ", 0 bad" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])
