"""Failure containment of the row tiling (the round-3 review's first item): a rank that never arrives, a rank that leaves in
the middle of a run, a communicator that does not come up -- each must come back as PWN_ETIMEDOUT within the deadline of
pwn_tiled_set_timeouts, with a message that names the rank and how far it got, never as a hang.  The reference's error model
is print-and-return (level.h:35-37,110-115); its frame loop has no peers to lose (main.c:93-109).

Ranks are child processes (tools/tiled_rank.py) on the test box's one GPU; RCCL itself is exercised with one rank (both ways of
driving the communicator), with a second rank that never calls ncclCommInitRank, and -- where the box has two devices -- with
two ranks on two devices."""
import json
import os
import re
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import GOLD, ROOT

pytestmark = pytest.mark.gpu
RANK = os.path.join(ROOT, "tools", "tiled_rank.py")


def start(rank, world, idfile, transport, w=640, h=360, frames=6, halo=-1, device=0, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    return subprocess.Popen([sys.executable, RANK, str(rank), str(world), idfile, transport, str(w), str(h), "pwnfps_level", str(frames), str(halo), str(device)],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)


def finish(p, limit):
    try:
        o, e = p.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        p.kill()
        raise AssertionError("a rank was still running after %d s: the deadline did not hold" % limit)
    return p.returncode, o, e


def golden_hash(w, h):
    with open(os.path.join(GOLD, "frames.json")) as f:
        cases = json.load(f)["cases"]
    return [c["post"] for c in cases if c["name"] == "level_spawn_%dx%d" % (w, h)][0]


def test_preflight_says_what_a_first_run_needs():
    import pwnfps_amd
    r = pwnfps_amd.Renderer(320, 240)
    r.tiled_set_timeouts(7, 3)
    p = r.tiled_preflight()
    assert p["devices_visible"] >= 1 and len(p["can_access_peer"]) == p["devices_visible"] and p["can_access_peer"][p["device"]] == 1
    assert p["librccl"] and os.path.exists(p["librccl"]) and p["rccl_version"] > 20000
    assert p["rccl_has_nonblocking_api"] is True and p["rccl_has_abort"] is True
    assert p["rccl_mode"] == "blocking" and (p["init_timeout_ms"], p["wait_timeout_ms"]) == (7000, 3000)
    r.tiled_set_timeouts(-1, -1)
    assert r.tiled_preflight()["init_timeout_ms"] == 120000
    r.close()


@pytest.mark.parametrize("mode", ["blocking", "nonblocking"])
def test_rccl_with_one_rank_both_ways_of_driving_it(mode, tmp_path):
    """The real library: communicator, the first exchange (a word to itself), six frames through both grouped launches."""
    p = start(0, 1, str(tmp_path / "id"), "rccl", w=1280, h=720, frames=6, PWN_TILED_RCCL_MODE=mode, TILED_SAME_SCENE=1, TILED_TIMEOUTS="60,30")
    rc, o, e = finish(p, 300)
    assert rc == 0, (o[-2000:], e[-3000:])
    hashes = re.findall(r"frame (\d+) fnv64 ([0-9a-f]{16})", o)
    assert [h for _, h in hashes] == [golden_hash(1280, 720)] * 6
    info = json.loads(re.search(r"info (\{.*\})", o).group(1))
    assert info["rccl_nonblocking"] == (1 if mode == "nonblocking" else 0) and info["dead"] == 0
    assert (info["init_timeout_ms"], info["wait_timeout_ms"]) == (60000, 30000)


@pytest.mark.parametrize("mode", ["blocking", "nonblocking"])
def test_rccl_bring_up_gives_up_on_a_rank_that_never_calls_init(mode, tmp_path):
    """World 2, and only rank 0 exists: ncclCommInitRank waits for a peer that will not come.  pwn_tiled_init returns
    PWN_ETIMEDOUT after the bring-up deadline and says where it was."""
    t0 = time.time()
    p = start(0, 2, str(tmp_path / "id"), "rccl", PWN_TILED_RCCL_MODE=mode, TILED_TIMEOUTS="4,2")
    rc, o, e = finish(p, 120)
    assert rc == 42, (rc, o[-2000:], e[-3000:])
    m = re.search(r"error (-?\d+) after ([\d.]+) s: (.*)", o)
    assert m and int(m.group(1)) == -10, o
    assert 3.5 <= float(m.group(2)) <= 12.0, m.group(2)          # the limit, plus at most the 5 s the helper is given to leave
    assert "rank 0" in m.group(3) and "ncclCommInitRank" in m.group(3), m.group(3)
    assert time.time() - t0 < 100


def test_shm_bring_up_gives_up_on_a_rank_that_never_arrives(tmp_path):
    p = start(0, 2, str(tmp_path / "id"), "shm", TILED_TIMEOUTS="4,2")
    rc, o, e = finish(p, 60)
    assert rc == 42, (rc, o[-2000:], e[-3000:])
    m = re.search(r"error (-?\d+) after ([\d.]+) s: (.*)", o)
    assert m and int(m.group(1)) == -10 and 1.5 <= float(m.group(2)) <= 6.0, o
    assert "rank 0 of 2" in m.group(3) and "did not answer" in m.group(3), m.group(3)


@pytest.mark.parametrize("hostsink", [0, 1])
def test_a_rank_that_leaves_mid_run_costs_its_peers_the_deadline_not_the_run(hostsink, tmp_path):
    """Three ranks; rank 1 leaves before frame 3.  The others get PWN_ETIMEDOUT from the next call that needs it, within
    the wait deadline, naming the frame; pwn_tiled_info says dead; no rank is left running."""
    idfile = str(tmp_path / "id")
    env = dict(TILED_TIMEOUTS="20,2", TILED_DIE_AT="3:1")
    if hostsink:
        env["TILED_HOSTSINK"] = 1
    procs = [start(r, 3, idfile, "shm", frames=8, **env) for r in range(3)]
    res = [finish(p, 90) for p in procs]
    assert res[1][0] == 17
    for r in (0, 2):
        rc, o, e = res[r]
        assert rc == 42, (r, rc, o[-2000:], e[-3000:])
        m = re.search(r"error (-?\d+) after ([\d.]+) s: (.*)", o)
        assert m and int(m.group(1)) == -10 and float(m.group(2)) <= 8.0, o
        assert "rank %d of 3" % r in m.group(3), m.group(3)
        info = json.loads(re.search(r"info (\{.*\})", o).group(1))
        assert info["dead"] == 1
        assert len(re.findall(r"rows \d+", o)) >= 1          # the frames before the loss were delivered
    if hostsink:
        for f in os.listdir("/dev/shm"):
            if f.startswith("pwn_frames_id"):
                os.unlink(os.path.join("/dev/shm", f))


def test_counted_frames_of_a_tiling_on_two_streams_keep_their_counters_apart(tmp_path):
    """with PWN_OPT_COUNTERS / PWN_OPT_WAVE_LOG on, frames of the tiling must not run side by side on the two
    compute streams (one set of counters, cleared by every launch).  The last launch's counts are its own: three rays per
    pixel of the strip, and one wave-log entry per wave of its grid."""
    for env in (dict(TILED_COUNTERS=1), dict(TILED_WAVELOG=1)):
        procs = [start(r, 2, str(tmp_path / ("id_%s" % list(env)[0])), "shm", w=1920, h=1080, frames=7, TILED_SAME_SCENE=1, TILED_BALANCE=0, **env) for r in range(2)]
        res = [finish(p, 300) for p in procs]
        for rc, o, e in res:
            assert rc == 0, (o[-2000:], e[-3000:])
            info = json.loads(re.search(r"info (\{.*\})", o).group(1))
            assert info["two_streams"] == 1
            rays, waves, y0, y1 = (int(v) for v in re.search(r"laststats (\d+) (\d+) (\d+) (\d+)", o).groups())
            if "TILED_COUNTERS" in env:
                assert rays == 3 * 1920 * (y1 - y0), (rays, y0, y1)
            else:
                assert 0 < waves <= 256 * 5 * 4, waves
        assert re.findall(r"frame \d+ fnv64 ([0-9a-f]{16})", res[0][1]) == [golden_hash(1920, 1080)] * 7


def test_a_timed_frame_reports_how_long_its_halo_group_took(tmp_path):
    """pwn_tiled_frame.halo_ms was never filled (the event it was read from carries no time stamp)."""
    procs = [start(r, 2, str(tmp_path / "id"), "shm", w=1280, h=720, frames=6, TILED_TIMING=1) for r in range(2)]
    res = [finish(p, 300) for p in procs]
    for rc, o, e in res:
        assert rc == 0, (o[-2000:], e[-3000:])
        t = [[float(v) for v in m] for m in re.findall(r"times \d+ ([\d.]+) ([\d.]+) ([\d.]+) ([\d.]+)", o)]
        assert len(t) == 6
        assert all(x[0] > 0 and x[1] > 0 and x[2] > 0 for x in t), t
        assert any(x[3] > 0 for x in t), t          # (the gather of the last frames is launched by pwn_tiled_wait itself: 0)


def test_rccl_between_two_devices(tmp_path):
    """Two ranks on two devices over RCCL: skipped on a box with one GPU (every box of this pool)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (hipGetDeviceCount() = %d)" % torch.cuda.device_count())
    idfile = str(tmp_path / "id")
    procs = [start(r, 2, idfile, "rccl", w=1920, h=1080, frames=8, device=r, TILED_SAME_SCENE=1, TILED_TIMEOUTS="120,30", TILED_PREFLIGHT=1) for r in range(2)]
    res = [finish(p, 400) for p in procs]
    for rc, o, e in res:
        assert rc == 0, (o[-2000:], e[-3000:])
    assert re.findall(r"frame \d+ fnv64 ([0-9a-f]{16})", res[0][1]) == [golden_hash(1920, 1080)] * 8
