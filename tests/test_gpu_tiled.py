"""Row tiling behind the C ABI (pwn_tiled_*, pwnfps_amd/csrc/pwn_tiled.cpp) with HIP kernels and
several ranks: fresh child processes, one per rank, all on the one GPU of the test box, over the
shared-memory transport (RCCL cannot run two ranks on one device; the choreography, buffers,
offsets and the miss protocol are the same code for both transports).  Every delivered frame of
a sequence with changing camera, clock and spheres is the oracle's frame."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT, level_path

pytestmark = pytest.mark.gpu
RANK = os.path.join(ROOT, "tools", "tiled_rank.py")


_runs = [0]


def run_ranks(world, w, h, level, frames, halo, tmp_path, transport="shm", blur=None):
    _runs[0] += 1
    idfile = str(tmp_path / ("id_%d" % _runs[0]))          # a fresh file per run: the ranks wait for it to appear
    env = dict(os.environ)
    if blur is not None:
        env["TILED_BLUR"] = str(blur)
    procs = [subprocess.Popen([sys.executable, RANK, str(r), str(world), idfile, transport, str(w), str(h), level, str(frames), str(halo)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-3000:]
        outs.append(o)
    hashes = re.findall(r"frame (\d+) fnv64 ([0-9a-f]{16}) redone (\d)", outs[0])
    infos = [json.loads(re.search(r"info (\{.*\})", o).group(1)) for o in outs]
    return hashes, infos


def oracle_hashes(w, h, level, frames, oracle_lib, blur=1):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import tiled_rank
    from oracle import Oracle
    O = Oracle()
    O.load_level(level_path(level))
    base = np.load(os.path.join(GOLD, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy")))
    _, _, spawn = O.get_level()
    want = []
    for k in range(frames):
        cam, sec, sph = tiled_rank.scene(k, base, spawn)
        O.set_spheres(sph)
        img, _ = O.render(w, h, cam, sec=sec, blur=blur)
        want.append(oracle_lib.fnv64(img))
    return want


@pytest.mark.parametrize("world", [1, 2, 3])
def test_tiled_frames_are_the_oracles_frames(world, tmp_path, oracle_lib):
    w, h, frames = 640, 360, 6
    want = oracle_hashes(w, h, "pwnfps_level", frames, oracle_lib)
    # default halo (depth 24: 19 rows of 360), whole strips (0), and a 1-row halo that the blur's taps leave
    for halo in (-1, 0, 1):
        hashes, infos = run_ranks(world, w, h, "pwnfps_level", frames, halo, tmp_path)
        assert [x[1] for x in hashes] == want, (world, halo)
        assert [i["frames"] for i in infos] == [frames] * world
        redone = [int(x[2]) for x in hashes]
        if world == 1 or halo <= 0:
            assert sum(redone) == 0 and all(i["frames_redone"] == 0 for i in infos)
        if world > 1 and halo == 1:
            # the first frame whose taps leave the halo is repeated with whole strips on EVERY rank,
            # later frames use whole strips: at most three frames were in flight with the small halo
            assert 1 <= sum(redone) <= 3 and len({i["frames_redone"] for i in infos}) == 1
            assert all(i["halo_rows"] == 0 for i in infos)
        if world > 1 and halo == -1:
            assert all(i["halo_rows"] == int(0.002 * h * 24) + 2 for i in infos)
            # one grouped exchange per frame plus two that drain the last two frames
            assert all(i["groups"] == frames + 2 for i in infos)


def test_tiled_without_blur_and_uneven_strips(tmp_path, oracle_lib):
    # 3 ranks, 100 rows: strips of 40, 40 and 20 rows; POSTPROC_BLUR off: a single gather
    w, h, frames = 256, 100, 4
    want = oracle_hashes(w, h, "synth64", frames, oracle_lib, blur=0)
    hashes, infos = run_ranks(3, w, h, "synth64", frames, -1, tmp_path, blur=0)
    assert [x[1] for x in hashes] == want
    assert [(i["y0"], i["y1"]) for i in infos] == [(0, 40), (40, 80), (80, 100)]
    want = oracle_hashes(w, h, "synth64", frames, oracle_lib, blur=1)
    hashes, infos = run_ranks(3, w, h, "synth64", frames, -1, tmp_path)      # halo 6 rows fits the 20-row strip
    assert [x[1] for x in hashes] == want


def test_tiled_eight_strip_geometry_at_4k(tmp_path, oracle_lib, cases):
    """3 ranks at the BASELINE frame size against the compiled reference's golden hash (the frame
    sequence's first frame is the spawn pose with the t=0 spheres)."""
    want = [c for c in cases if c["name"] == "level_spawn_3840x2160"][0]["post"]
    hashes, infos = run_ranks(3, 3840, 2160, "pwnfps_level", 1, -1, tmp_path)
    assert hashes[0][1] == want
    assert infos[0]["halo_rows"] == 105


def test_tiled_over_rccl_with_one_rank(tmp_path, oracle_lib):
    """What a one-GPU box can run of the RCCL transport: librccl is found and loaded, the communicator
    is created, a word makes the round trip through ncclSend / ncclRecv (pwn_tiled_init's own check:
    with one rank, to itself), frames are delivered."""
    w, h, frames = 640, 360, 3
    want = oracle_hashes(w, h, "pwnfps_level", frames, oracle_lib)
    hashes, infos = run_ranks(1, w, h, "pwnfps_level", frames, -1, tmp_path, transport="rccl")
    assert [x[1] for x in hashes] == want
    assert infos[0]["transport"] == 0 and infos[0]["world"] == 1
