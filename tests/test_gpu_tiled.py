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


def run_ranks(world, w, h, level, frames, halo, tmp_path, transport="shm", blur=None, hostsink=False, seen=None, balance=None, cuts_at=None,
              rows=None, extra_env=None, by_rank=None):
    _runs[0] += 1
    idfile = str(tmp_path / ("id_%d" % _runs[0]))          # a fresh file per run: the ranks wait for it to appear
    env = dict(os.environ)
    if blur is not None:
        env["TILED_BLUR"] = str(blur)
    if hostsink:
        env["TILED_HOSTSINK"] = "1"
    if balance is not None:
        env["TILED_BALANCE"] = str(balance)
    if cuts_at:
        env["TILED_CUTS_AT"] = ";".join("%d:%s" % (k, ",".join(str(v) for v in c)) for k, c in cuts_at.items())
    env.update(extra_env or {})
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
    if by_rank is not None:                               # rotating gather root: which rank printed which frame
        by_rank.extend(re.findall(r"frame (\d+) fnv64 ([0-9a-f]{16}) redone (\d)", o) for o in outs)
    if seen is not None:                                  # host sink: what the other ranks saw in the shared frame
        seen.extend([h for _, h in re.findall(r"seen (\d+) fnv64 ([0-9a-f]{16})", o)] for o in outs[1:])
    infos = [json.loads(re.search(r"info (\{.*\})", o).group(1)) for o in outs]
    if rows is not None:                                  # per rank: [(frame, y0, y1, cost)], and the cuts at the end
        rows.extend([tuple(int(v) for v in m) for m in re.findall(r"rows (\d+) (\d+) (\d+) (\d+)", o)] for o in outs)
        rows.append([json.loads(re.search(r"cuts (\[.*\])", o).group(1)) for o in outs])
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


@pytest.mark.parametrize("world,hostsink,rotate", [(1, False, False), (3, False, False), (4, False, True), (3, True, False)])
def test_split_choreography_gives_the_same_frames(world, hostsink, rotate, tmp_path, oracle_lib):
    """PWN_OPT_TILED_CHOREO = split (bench.py's sweep leg `choreo_split`; the default of rounds 2-3): the exchanges on a
    third stream, blur f enqueued by submit f+1 and its gather by submit f+2.  The same frames, groups and bytes as the
    in-stream default, bounded halo, missed halo and whole strips."""
    w, h, frames = 640, 360, 6
    want = oracle_hashes(w, h, "pwnfps_level", frames, oracle_lib)
    env = {"PWN_TILED_CHOREO": "split"}
    if rotate:
        env["TILED_ROTATE"] = "1"
    for halo in (-1, 1):
        by_rank, seen = [], []
        hashes, infos = run_ranks(world, w, h, "pwnfps_level", frames, halo, tmp_path, hostsink=hostsink, seen=seen, extra_env=env, by_rank=by_rank)
        if rotate:
            got = {int(k): hh for per in by_rank for k, hh, _ in per}
            assert [got[k] for k in range(frames)] == want, (world, halo)
        else:
            assert [x[1] for x in hashes] == want, (world, halo)
        assert not hostsink or all(s_ == want for s_ in seen)
        assert [i["frames"] for i in infos] == [frames] * world
        if world > 1 and halo == 1:
            assert len({i["frames_redone"] for i in infos}) == 1 and infos[0]["frames_redone"] >= 1 and all(i["halo_rows"] == 0 for i in infos)
        if world > 1 and halo == -1:
            assert all(i["frames_redone"] == 0 and i["groups"] == 2 * frames for i in infos)


@pytest.mark.parametrize("world", [1, 2, 3, 4])
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
            # two grouped launches per frame: its halo rows in front of its blur, its gather behind
            assert all(i["groups"] == 2 * frames for i in infos)


@pytest.mark.parametrize("world,halo,balance,blur", [(3, -1, 0, 1), (4, 1, 2, 1), (2, 0, None, 1), (3, -1, None, 0)])
def test_tiled_with_rotating_gather_root(world, halo, balance, blur, tmp_path, oracle_lib):
    """pwn_tiled_gather_root(PWN_TILED_ROOT_ROTATE): frame k is assembled on rank k mod world (the strips of every frame go to
    another GPU, so that no rank's links carry every frame).  Every frame is the oracle's, and is handed out by its root and
    by nobody else -- with the default halo, with a 1-row halo that the taps leave (the frame repeated with whole strips, on
    its own root) while the cuts move, with whole strips, and without blur (the trace writes a root's own strip straight
    into the assembled frame)."""
    w, h, frames = 640, 360, 9
    want = oracle_hashes(w, h, "pwnfps_level", frames, oracle_lib, blur=blur)
    by_rank = []
    _, infos = run_ranks(world, w, h, "pwnfps_level", frames, halo, tmp_path, balance=balance, blur=blur, extra_env={"TILED_ROTATE": "1"}, by_rank=by_rank)
    assert all(i["gather_root"] == 1 and i["frames"] == frames for i in infos)
    got = {}
    for r, lines in enumerate(by_rank):
        for k, hsh, _ in lines:
            assert int(k) % world == r, (k, r)                 # only a frame's root hands it out
            assert int(k) not in got
            got[int(k)] = hsh
    assert [got[k] for k in range(frames)] == want
    if halo == 1:
        assert all(i["frames_redone"] >= 1 and i["halo_rows"] == 0 for i in infos)
    if halo == -1 and blur:
        # what a rank's links carried: its own strip out for the frames it is not the root of, the others' strips in for
        # those it is (equal strips of 120 / 90 rows here), and the halo rows both ways
        for i in infos:
            mine = i["y1"] - i["y0"]
            roots = len(range(i["rank"], frames, world))
            H = i["halo_rows"]
            nb = (1 if i["rank"] > 0 else 0) + (1 if i["rank"] < world - 1 else 0)
            assert i["bytes_sent"] == (frames - roots) * mine * w * 4 + frames * nb * H * w * 4, i
            assert i["bytes_received"] == roots * (h - mine) * w * 4 + frames * nb * H * w * 4, i


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


@pytest.mark.parametrize("comms,streams,depth", [("one", 2, 3), ("perstream", 2, 3), ("perstream", 3, 4), ("one", 3, 5)])
def test_one_rank_exchanging_with_itself_over_rccl(comms, streams, depth, tmp_path, oracle_lib):
    """PWN_TILED_SELF=1 (measurement mode): the rank sends itself what a rank of a real tiling sends -- border rows
    behind every trace, its finished strip and its words behind every blur -- in the library's two grouped launches per frame,
    on the frames' own compute streams.  With one communicator for all streams, and with one per stream (PWN_OPT_TILED_COMMS:
    every further communicator is brought up like the first, its id travelling over the first); two and three compute streams;
    three to five frames in flight.  The frames are the oracle's."""
    w, h, frames = 640, 360, 12
    want = oracle_hashes(w, h, "pwnfps_level", frames, oracle_lib)
    env = {"PWN_TILED_SELF": "1", "PWN_TILED_COMMS": comms, "PWN_TILED_STREAMS": str(streams), "TILED_DEPTH": str(depth)}
    hashes, infos = run_ranks(1, w, h, "pwnfps_level", frames, -1, tmp_path, transport="rccl", extra_env=env)
    assert [x[1] for x in hashes] == want
    i = infos[0]
    assert i["transport"] == 0 and i["world"] == 1 and i["compute_streams"] == streams and i["choreography"] == 0
    assert i["communicators"] == (streams if comms == "perstream" else 1)
    assert i["groups"] == 2 * frames and i["bytes_sent"] == i["bytes_received"] > frames * h * w * 4


@pytest.mark.parametrize("name", ["level_spawn_3840x2160", "synth256_cam0_7680x4320"])
def test_eight_strip_geometry_single_process(name, oracle_lib, cases):
    """The per-rank kernels of an 8-GPU run at the BASELINE sizes, one process: every strip of the 8-way
    tiling traced with pwn_trace_rows_device, then blurred with pwn_blur_rows_device_bounded from a
    plane that holds ONLY the rows that rank would have (its strip and the default halo: 105 rows at 4K,
    209 at 8K) and poison everywhere else.  The assembled frame is the compiled reference's golden frame,
    or, where the taps leave the halo (the mirror halls of synth256), the strips say so."""
    import torch
    import pwnfps_amd
    from pwnfps_amd.dist import default_halo, strip_range
    c = [x for x in cases if x["name"] == name][0]
    w, h = c["w"], c["h"]
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(level_path(c["level"]))
    key = "t0" if c["level"] == "pwnfps_level" else c["level"]
    from conftest import load_spheres
    r.set_objects(load_spheres(key))
    cam = np.array(c["cam"], np.float32)
    dev = torch.device("cuda:0")
    pre = torch.zeros((h, w), dtype=torch.int32, device=dev)
    z = torch.zeros((h, w), dtype=torch.float32, device=dev)
    out = torch.zeros((h, w), dtype=torch.int32, device=dev)
    miss = torch.zeros(8, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for rank in range(8):
        y0, y1 = strip_range(h, 8, rank)
        r.trace_rows_device(cam, c["sec"], y0, y1, pre.data_ptr(), z.data_ptr(), s)
    torch.cuda.synchronize()
    assert oracle_lib.fnv64(pre.cpu().numpy()) == c["pre"]
    H = default_halo(h)
    assert H == {2160: 105, 4320: 209}[h]
    for rank in range(8):
        y0, y1 = strip_range(h, 8, rank)
        a0, a1 = (y0 - H if rank > 0 else 0), (y1 + H if rank < 7 else h)
        have = torch.full_like(pre, 0x5EADBEEF)
        have[a0:a1] = pre[a0:a1]
        r.blur_rows_device_bounded(y0, y1, have.data_ptr(), z.data_ptr(), out.data_ptr(), a0, a1, miss[rank:rank + 1].data_ptr(), s)
    torch.cuda.synchronize()
    misses = miss.cpu().numpy()
    if (misses == 0).all():
        assert oracle_lib.fnv64(out.cpu().numpy()) == c["post"]
    else:
        # taps left the halo somewhere: those strips are repeated with whole strips (pwn_tiled_wait);
        # the strips that reported nothing are already the golden's
        assert c["level"] == "synth256"
        for rank in range(8):
            y0, y1 = strip_range(h, 8, rank)
            if misses[rank]:
                r.blur_rows_device(y0, y1, pre.data_ptr(), z.data_ptr(), out.data_ptr(), s)
        torch.cuda.synchronize()
        assert oracle_lib.fnv64(out.cpu().numpy()) == c["post"]
    if c["level"] == "pwnfps_level":
        assert (misses == 0).all()
    r.close()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_tiled_host_sink(world, tmp_path, oracle_lib):
    """pwn_tiled_host_sink: every rank copies its strip into ONE frame in POSIX shared memory, no gather; when a
    rank's wait returns, the whole frame is there -- on every rank.  Default halo, whole strips, and a 1-row halo
    that is missed (the frame is repeated and copied again)."""
    w, h, frames = 640, 360, 6
    want = oracle_hashes(w, h, "pwnfps_level", frames, oracle_lib)
    for halo in (-1, 0, 1):
        seen = []
        hashes, infos = run_ranks(world, w, h, "pwnfps_level", frames, halo, tmp_path, hostsink=True, seen=seen)
        assert [x[1] for x in hashes] == want, (world, halo)
        assert all(sv == want for sv in seen), (world, halo)
        assert all(i["host_sink"] == 1 for i in infos)
        redone = infos[0]["frames_redone"]
        assert (redone > 0) == (world > 1 and halo == 1)
        for i in infos:
            assert i["bytes_to_host"] == (frames + i["frames_redone"]) * (i["y1"] - i["y0"]) * w * 4
            # nothing but halo rows, whole pre-blur strips after a miss, and one word per pair travels between ranks
            if world > 1 and halo == -1:
                assert i["bytes_sent"] <= frames * 2 * (int(0.002 * h * 24) + 2) * w * 4
    # no blur: the traced strips go straight to the host
    want0 = oracle_hashes(w, h, "pwnfps_level", 3, oracle_lib, blur=0)
    hashes, infos = run_ranks(world, w, h, "pwnfps_level", 3, -1, tmp_path, blur=0, hostsink=True)
    assert [x[1] for x in hashes] == want0


def test_host_sink_argument_checks_and_one_rank(oracle_lib, cases):
    """pwn_tiled_host_sink in one process (world 1, no transport traffic): a buffer that is too small, a second
    sink, a sink after the first frame are refused; frames arrive in the caller's memory."""
    import pwnfps_amd
    from pwnfps_amd import _lib
    from pwnfps_amd.render import PwnError
    from conftest import load_spheres
    c = [x for x in cases if x["name"] == "level_pose1_1280x720"][0]
    w, h = c["w"], c["h"]
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(level_path(c["level"]))
    r.set_objects(load_spheres(c["spheres"]))
    cam = np.array(c["cam"], np.float32)
    r.tiled_init(0, 1, pwnfps_amd.Renderer.tiled_unique_id("shm"), "shm", -1)
    small = np.zeros(w * h, np.uint32)
    with pytest.raises(PwnError) as e:
        r.tiled_host_sink(small)
    assert e.value.code == -1                      # PWN_EINVAL: a sink holds PWN_TILED_SLOTS frames
    frames = np.zeros((_lib.PWN_TILED_SLOTS, h, w), np.uint32)
    r.tiled_host_sink(frames)
    with pytest.raises(PwnError) as e:
        r.tiled_host_sink(frames)
    assert e.value.code == -8                      # PWN_EBUSY
    for k in range(5):
        r.tiled_submit(cam, c["sec"])
        fr = r.tiled_wait()
        assert fr["d_sbuf"] in (None, 0) and oracle_lib.fnv64(fr["sbuf"]) == c["post"]
        # the frame IS the caller's memory: slot k mod 4
        assert oracle_lib.fnv64(frames[k % _lib.PWN_TILED_SLOTS]) == c["post"]
    assert r.tiled_info()["bytes_to_host"] == 5 * w * h * 4
    r.tiled_shutdown()
    # after a frame without a sink it is too late
    r.tiled_init(0, 1, pwnfps_amd.Renderer.tiled_unique_id("shm"), "shm", -1)
    r.tiled_submit(cam, c["sec"])
    with pytest.raises(PwnError) as e:
        r.tiled_host_sink(frames)
    assert e.value.code == -8
    # pwn_tiled_gather_root: not with a frame in flight, not with a mode the header does not name; with one rank every
    # frame's root is rank 0 either way
    with pytest.raises(PwnError) as e:
        r.tiled_gather_root(True)
    assert e.value.code == -8
    r.tiled_wait()
    assert pwnfps_amd.render.lib.pwn_tiled_gather_root(r._ctx, 7) == -1
    r.tiled_gather_root(True)
    for k in range(3):
        r.tiled_submit(cam, c["sec"])
        fr = r.tiled_wait(host=True)
        assert fr["root"] == 0 and oracle_lib.fnv64(fr["sbuf"]) == c["post"]
    assert r.tiled_info()["gather_root"] == 1
    r.tiled_gather_root(False)
    assert r.tiled_info()["gather_root"] == 0
    r.tiled_shutdown()
    r.close()



def _check_rows(rows, world, h, frames):
    """every frame's strips tile the frame, in rank order, the same on every rank's account"""
    per_rank, final_cuts = rows[:world], rows[world]
    assert all(c == final_cuts[0] for c in final_cuts)
    cuts_of = []
    for k in range(frames):
        edges = [0]
        for r in range(world):
            fk = [x for x in per_rank[r] if x[0] == k]
            assert len(fk) == 1 and fk[0][1] == edges[-1] and fk[0][2] > fk[0][1]
            edges.append(fk[0][2])
        assert edges[-1] == h and all(e % 8 == 0 for e in edges[1:-1])
        cuts_of.append(edges)
    return cuts_of


@pytest.mark.parametrize("world,halo,hostsink", [(3, -1, False), (2, -1, False), (3, 1, False), (3, -1, True), (4, 0, False), (5, -1, False)])
def test_tiled_frames_with_moving_cuts(world, halo, hostsink, tmp_path, oracle_lib):
    """Moving cuts (pwn_tiled_balance / pwn_tiled_set_cuts): the strips are re-cut every second delivered frame from
    what the trace launches measured, and twice by hand to cuts far from equal; every frame is the oracle's frame
    whatever cuts it was traced with -- halo exchange, blur, gather offsets, the repeat after a missed halo (halo 1)
    and the host-sink copies all follow the frame's own cuts."""
    w, h, frames = 640, 360, 14
    want = oracle_hashes(w, h, "pwnfps_level", frames, oracle_lib)
    eq = [min(r * (-(-(-(-h // world)) // 8) * 8), h) for r in range(world)] + [h]
    # (five rank processes and this one: the box allows six processes on its GPU)
    by_hand = {3: 24 if world <= 4 else 8, 9: -16}
    cuts_at = {k: [0] + [c + d for c in eq[1:-1]] + [h] for k, d in by_hand.items()}
    rows, seen = [], []
    hashes, infos = run_ranks(world, w, h, "pwnfps_level", frames, halo, tmp_path, hostsink=hostsink, seen=seen, balance=2, cuts_at=cuts_at, rows=rows)
    assert [x[1] for x in hashes] == want
    if hostsink:
        assert all(sn == want for sn in seen)
    cuts_of = _check_rows(rows, world, h, frames)
    assert cuts_of[0] == eq and cuts_of[3] == cuts_at[3] and cuts_of[9] == cuts_at[9]
    # the launches measured something, and it moved the cuts away from what was set by hand
    assert all(x[3] > 0 for x in rows[0])
    assert len({tuple(c) for c in cuts_of}) >= 4
    assert infos[0]["recuts"] >= 2 and len({i["recuts"] for i in infos}) == 1
    assert all(i["balance_every"] == 2 and i["max_rows"] >= -(-h // world) for i in infos)


def test_moving_cuts_even_out_the_strips_of_a_4k_frame(tmp_path, oracle_lib, cases):
    """3 ranks at the BASELINE frame size, the same frame sixteen times, a re-cut every second frame: the frames stay the
    compiled reference's golden frame and the strips' costs come closer together than with the equal split."""
    want = [c for c in cases if c["name"] == "level_spawn_3840x2160"][0]["post"]
    rows = []
    frames = 16
    hashes, infos = run_ranks(3, 3840, 2160, "pwnfps_level", frames, -1, tmp_path, balance=2, rows=rows, extra_env={"TILED_SAME_SCENE": "1"})
    assert [x[1] for x in hashes] == [want] * frames
    cuts_of = _check_rows(rows, 3, 2160, frames)
    # (three processes take turns on ONE GPU here: now and then a launch is held up and reports a cost too high, never
    # one too low -- the library's re-cut works from the smallest cost per rank for the same reason)
    least = lambda ks: [min([x for x in rows[r] if x[0] == k][0][3] for k in ks) for r in range(3)]      # noqa: E731
    spread = lambda c: max(c) / (sum(c) / len(c))                                                     # noqa: E731
    first, last = least(range(0, 4)), least(range(frames - 6, frames))        # frames 0..3 were traced with the equal split
    assert cuts_of[3] == cuts_of[0] and cuts_of[-1] != cuts_of[0], cuts_of[-1]
    assert spread(last) < 1.03 and spread(last) <= spread(first) + 0.005, (first, last, cuts_of[-1])
