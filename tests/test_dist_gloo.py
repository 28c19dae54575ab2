"""The row-tiling choreography (pwnfps_amd/csrc/pwn_tiled.cpp, restated statement for statement in
pwnfps_amd/dist.py over torch.distributed point-to-point operations) on CPU with the gloo
backend, world_size 2, 3 and 8: what goes into which grouped exchange, which buffers a frame
owns with two frames in flight, the repeat of a frame whose blur taps left the halo, and that
every rank decides the same.  The strip work is done by a CHECKER backend built on the oracle;
the C implementation with the HIP kernels runs the same frame sequence in tests/test_gpu_tiled.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLD, HERE, ROOT, level_path, load_spheres


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleStripBackend:
    """trace_rows/blur_rows on CPU torch tensors via the oracle (checker only)."""

    def __init__(self, w, h, level, spheres):
        import oracle
        self.o = oracle.Oracle()
        self.o.load_level(level)
        self.o.set_spheres(spheres)
        self.w, self.h = w, h

    def _np(self, t, dtype):
        return t.numpy().view(dtype)

    def trace_rows(self, cam, sec, y0, y1, pre, z):
        hp = pre.shape[0]
        self.o.L.pwno_trace_rows(self.o.lv, self.w, self.h, y0, y1, np.ascontiguousarray(cam, np.float32).ctypes.data,
                                 float(sec), 1, pre.data_ptr(), z.data_ptr(), None)
        assert hp >= self.h
        # what the rows "cost" (the C code measures its trace launch): a horizon band three times as dear as the rest
        y = np.arange(y0, y1, dtype=np.float64)
        return int(np.sum(100.0 + 300.0 * np.exp(-((y - 0.5 * self.h) / (0.12 * self.h)) ** 2)))

    def blur_rows(self, y0, y1, pre, z, out):
        self.o.L.pwno_blur_rows(self.w, self.h, y0, y1, 1, pre.data_ptr(), z.data_ptr(), out.data_ptr())

    def blur_rows_bounded(self, y0, y1, pre, z, out, avail_y0, avail_y1, miss):
        """The contract of pwn_blur_rows_device_bounded with the oracle: rows outside
        [avail_y0, avail_y1) are NOT this frame's - overwrite them with garbage so that a
        wrong halo shows in the pixels - and report (conservatively, from the depth) whether
        a tap can have left the available rows."""
        p = pre.numpy()
        p[:avail_y0] = 0x5EADBEEF
        p[avail_y1:] = 0x5EADBEEF
        zz = z.numpy()[y0:y1, :self.w]
        reach = int(np.floor(np.float32(0.002 * self.h) * np.abs(zz - 1.0).max())) + 1 if y1 > y0 else 0
        lo, hi = max(y0 - reach, 0), min(y1 - 1 + reach, self.h - 1)
        if lo < avail_y0 or hi >= avail_y1:
            miss += 1
        self.o.L.pwno_blur_rows(self.w, self.h, y0, y1, 1, pre.data_ptr(), z.data_ptr(), out.data_ptr())


def _worker(rank, world, port, w, h, level, frames, blur, halo, q, sink_path=None, balance=None, cuts_at=None, rotate=False):
    import sys
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        import tiled_rank                      # the frame sequence of the GPU test of the C implementation
        from pwnfps_amd.dist import NSLOT, TiledFrames, strip_range
        key = "t0" if level == "pwnfps_level" else level
        base = load_spheres(key)
        be = OracleStripBackend(w, h, level_path(level), base)
        _, _, spawn = be.o.get_level()
        sink = None
        if sink_path is not None:
            # pwn_tiled_host_sink: one file mapped by every rank plays the shared host memory
            sink = np.memmap(sink_path, dtype=np.uint32, mode="r+", shape=(NSLOT, h, w))
        fr = TiledFrames(w, h, be, torch.device("cpu"), rank=rank, world=world, blur_passes=blur, halo_rows=halo, host_sink=sink,
                         balance_every=balance, rotate_root=rotate)
        assert (fr.y0, fr.y1) == strip_range(h, world, rank)
        halo0 = fr.halo
        got = []
        cuts_seen = []
        cuts_at = cuts_at or {}

        def deliver(k):
            frame, redone = fr.wait()
            if sink is not None:
                got.append((oracle.fnv64(np.array(frame)), redone))        # every rank sees the whole frame
                return
            root = k % world if rotate else 0                          # pwn_tiled_gather_root
            assert (frame is not None) == (rank == root)
            got.append((oracle.fnv64(fr.to_host(frame)) if rank == root else None, redone))
        for k in range(frames):
            cam, sec, sph = tiled_rank.scene(k, base, spawn)
            be.o.set_spheres(sph)
            if k in cuts_at:
                fr.set_cuts(cuts_at[k])
            cuts_seen.append(list(fr.cuts))
            fr.submit(cam, sec)
            if k >= 2:
                deliver(k - 2)
        cam, sec, sph = tiled_rank.scene(frames, base, spawn)
        be.o.set_spheres(sph)
        fr.submit(cam, sec)
        # at most NSLOT - 1 frames in flight (the guard alone: this loop keeps three, like the hosts of rounds 2-3)
        held = fr.delivered
        fr.delivered = fr.submitted - (NSLOT - 1)
        with pytest.raises(RuntimeError, match="in flight"):
            fr.submit(cam, sec)
        fr.delivered = held
        deliver(frames - 2)
        deliver(frames - 1)
        deliver(frames)
        with pytest.raises(RuntimeError, match="nothing in flight"):
            fr.wait()
        info = dict(fr.info)
        info.update(recuts=fr.recuts, cuts_seen=cuts_seen, last_cost=list(fr.last_cost))
        q.put((rank, got, halo0, fr.halo, info))
    finally:
        dist.destroy_process_group()


def _run(world, w, h, level, frames, blur, halo, sink_path=None, balance=None, cuts_at=None, rotate=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, level, frames, blur, halo, q, sink_path, balance, cuts_at, rotate)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=600)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _want(w, h, level, frames, blur):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import oracle
    import tiled_rank
    O = oracle.Oracle()
    O.load_level(level_path(level))
    base = load_spheres("t0" if level == "pwnfps_level" else level)
    _, _, spawn = O.get_level()
    out = []
    for k in range(frames):
        cam, sec, sph = tiled_rank.scene(k, base, spawn)
        O.set_spheres(sph)
        out.append(oracle.fnv64(O.render(w, h, cam, sec=sec, blur=blur)[0]))
    return out


# halo: -1 = the default (depth 24: 13 rows at h = 240, never left by this scene), 0 = whole strips to
# everybody, 1 = one row, which the blur's taps leave: the frame is repeated with whole strips
@pytest.mark.parametrize("world,blur,halo", [(2, 1, -1), (3, 1, -1), (8, 1, -1), (2, 1, 0), (3, 1, 1), (8, 1, 1), (2, 0, -1), (3, 0, -1)])
def test_tiled_frames_are_the_oracles_frames(world, blur, halo):
    w, h, frames = 320, 240, 5
    want = _want(w, h, "pwnfps_level", frames + 1, blur)
    res = _run(world, w, h, "pwnfps_level", frames, blur, halo)
    got, halo0, halo1, info0 = res[0]
    assert [g[0] for g in got] == want
    redone = [g[1] for g in got]
    for r in range(world):
        g, h0, h1, info = res[r]
        # every rank made the same decisions
        assert [x[1] for x in g] == redone and (h0, h1) == (halo0, halo1)
        assert info["frames"] == frames + 1 and info["frames_redone"] == sum(redone)
    if blur == 0 or halo == 0:
        assert halo0 == 0 and sum(redone) == 0
    elif halo == -1:
        assert halo0 == 13 and halo1 == 13 and sum(redone) == 0
        # two grouped exchanges per frame: its halo rows, and its gather (with a later submit, or when the run drains)
        assert info0["groups"] == 2 * (frames + 1)
    else:
        # the first frame whose taps leave the one-row halo is repeated -- and so are the two frames that were
        # already in flight with that halo if their taps leave it too; whole strips from then on
        assert halo0 == 1 and halo1 == 0 and 1 <= sum(redone) <= 3 and redone[0]
    if world > 1 and blur and halo == -1:
        # a strip's neighbours send it 13 rows each, rank 0 takes in every other strip
        per = res[0][3]["bytes_received"] / (frames + 1)
        assert per >= (h - 32) * w * 4 if world == 8 else per > 0


@pytest.mark.parametrize("world,blur,halo,balance", [(3, 1, -1, 0), (8, 1, 1, 2), (2, 0, -1, None), (4, 1, 0, None)])
def test_rotating_gather_root(world, blur, halo, balance):
    """pwn_tiled_gather_root(ROTATE) restated: frame k is assembled on rank k mod world and handed out there and nowhere
    else; every frame is the oracle's whether the halo holds, is left (the repeat lands on the frame's own root), or
    whole strips travel, with and without blur, while the cuts move.  With the default halo and fixed cuts a rank
    receives exactly the other ranks' strips of the frames it is the root of (plus its neighbours' halo rows)."""
    w, h, frames = 320, 240, 8
    want = _want(w, h, "pwnfps_level", frames + 1, blur)
    res = _run(world, w, h, "pwnfps_level", frames, blur, halo, balance=balance, rotate=True)
    for k in range(frames + 1):
        for r in range(world):
            assert res[r][0][k][0] == (want[k] if r == k % world else None), (k, r)
    assert len({tuple(x[1] for x in res[r][0]) for r in range(world)}) == 1        # every rank made the same decisions
    if halo == -1 and blur:
        for r in range(world):
            info = res[r][3]
            mine = info["cuts_seen"][0][r + 1] - info["cuts_seen"][0][r]
            roots = len(range(r, frames + 1, world))
            nb = (1 if r > 0 else 0) + (1 if r < world - 1 else 0)
            words = (frames + 1) * (world - 1) * 2
            assert info["bytes_received"] == (roots * (h - mine) * w + (frames + 1) * nb * 13 * w + words) * 4, (r, info)
            assert info["bytes_sent"] == ((frames + 1 - roots) * mine * w + (frames + 1) * nb * 13 * w + words) * 4, (r, info)


def test_uneven_strips_and_a_strip_shorter_than_the_halo():
    # 100 rows over 3 ranks: 40 + 40 + 20; the default halo of 100 rows is 6
    w, h, frames = 128, 100, 4
    want = _want(w, h, "synth64", frames + 1, 1)
    res = _run(3, w, h, "synth64", frames, 1, -1)
    assert [g[0] for g in res[0][0]] == want and res[0][1] == 6
    # a halo taller than the shortest strip cannot be exchanged with neighbours only: whole strips
    res = _run(3, w, h, "synth64", frames, 1, 30)
    assert [g[0] for g in res[0][0]] == want and res[0][1] == 0


def test_strip_geometry():
    from pwnfps_amd.dist import default_halo, strip_range, strip_rows
    assert strip_rows(2160, 8) == 272 and strip_rows(4320, 8) == 544 and strip_rows(240, 3) == 80
    assert [strip_range(2160, 8, r) for r in (0, 6, 7)] == [(0, 272), (1632, 1904), (1904, 2160)]
    assert strip_range(100, 3, 2) == (80, 100) and strip_range(16, 4, 3) == (16, 16)
    assert default_halo(2160) == 105 and default_halo(4320) == 209 and default_halo(240) == 13


@pytest.mark.parametrize("world,blur,halo", [(2, 1, -1), (3, 1, 1), (8, 1, -1), (3, 0, -1)])
def test_host_sink_delivers_whole_frames_to_every_rank(world, blur, halo, tmp_path):
    """pwn_tiled_host_sink restated: no gather; every rank copies its strip into one frame that all ranks have
    mapped, a word per pair of ranks follows the copy, and a delivered frame is whole on EVERY rank."""
    w, h, frames = 320, 240, 5
    path = str(tmp_path / "frames.bin")
    np.zeros((6, h, w), np.uint32).tofile(path)          # pwnfps_amd.dist.NSLOT frames
    want = _want(w, h, "pwnfps_level", frames + 1, blur)
    res = _run(world, w, h, "pwnfps_level", frames, blur, halo, sink_path=path)
    redone = [g[1] for g in res[0][0]]
    for r in range(world):
        g, h0, h1, info = res[r]
        assert [x[0] for x in g] == want, r
        assert [x[1] for x in g] == redone
        assert info["bytes_to_host"] > 0
        # between ranks: halo rows (or whole pre-blur strips after a miss) and two words per pair -- never finished strips
        if halo == -1 and blur:
            assert info["bytes_sent"] <= (frames + 1) * (2 * 13 * w * 4 + 8 * (world - 1))
        if blur == 0:
            assert info["bytes_sent"] == (frames + 1) * 8 * (world - 1)
    assert (sum(redone) > 0) == (halo == 1)



# ---- moving cuts (pwn_tiled_balance / pwn_tiled_set_cuts) -------------------------------------------------------

@pytest.mark.parametrize("world,halo,sink", [(8, -1, False), (3, -1, False), (3, 1, False), (4, 0, False), (3, -1, True)])
def test_moving_cuts_keep_every_frame_exact(world, halo, sink, tmp_path):
    """The strips are re-cut every second delivered frame from the ranks' cost words and once by hand; every rank
    computes the same cuts, a frame in flight keeps the cuts it was traced with (exchange, blur, gather, host-sink
    copies, the repeat after a missed halo), and every delivered frame is the oracle's."""
    from pwnfps_amd.dist import equal_cuts
    w, h, frames = 320, 240, 11
    path = None
    if sink:
        path = str(tmp_path / "frames.bin")
        np.zeros((6, h, w), np.uint32).tofile(path)          # pwnfps_amd.dist.NSLOT frames
    eq = equal_cuts(h, world)
    by_hand = {5: [0] + [c + 8 for c in eq[1:-2]] + [eq[-2], h]}            # (the last strip of the equal split is the short one)
    want = _want(w, h, "pwnfps_level", frames + 1, 1)
    res = _run(world, w, h, "pwnfps_level", frames, 1, halo, path, balance=2, cuts_at=by_hand)
    assert [g[0] for g in res[0][0]] == want
    seen0 = res[0][3]["cuts_seen"]
    for r in range(world):
        info = res[r][3]
        assert info["cuts_seen"] == seen0 and info["recuts"] == res[0][3]["recuts"] and info["last_cost"] == res[0][3]["last_cost"]
        if sink:
            assert [g[0] for g in res[r][0]] == want
    assert seen0[0] == eq and seen0[5] == by_hand[5] and res[0][3]["recuts"] >= 2
    # the cuts went where the cost is: the strips over the dear middle band got shorter than the equal split's
    rows = lambda c: [c[i + 1] - c[i] for i in range(world)]        # noqa: E731
    assert rows(seen0[4])[world // 2] < rows(eq)[world // 2] or world <= 3
    for c in seen0:
        assert c[0] == 0 and c[-1] == h and all(v % 8 == 0 for v in c[1:-1]) and all(b > a for a, b in zip(c, c[1:]))


def test_the_recut_rule_is_the_librarys():
    """pwnfps_amd.dist.recut restates pwn_tiled.cpp's rule; the library exports it (pwn_tiled_recut, no device needed)."""
    import ctypes as C
    from pwnfps_amd import _lib
    from pwnfps_amd.dist import equal_cuts, max_strip_rows, recut
    rng = np.random.default_rng(7)
    moved = 0
    for trial in range(400):
        world = int(rng.integers(2, 9))
        h = int(rng.choice([240, 360, 720, 1080, 2160, 4320]))
        cuts = equal_cuts(h, world)
        mx = max_strip_rows(h, world)
        mn = int(rng.choice([8, int(0.002 * h * 24.0) + 2]))
        for step in range(4):
            cost = (rng.integers(1, 1000, world) * rng.choice([1, 1, 1, 50], world)).astype(np.uint32)
            if trial % 17 == 0:
                cost[int(rng.integers(0, world))] = 0
            a = np.array(cuts, np.int32)
            out = np.zeros(world + 1, np.int32)
            rc = _lib.lib.pwn_tiled_recut(a.ctypes.data, cost.ctypes.data, world, h, mn, mx, out.ctypes.data)
            mine = recut(cuts, [int(v) for v in cost], world, h, mn, mx)
            assert rc in (0, 1) and (rc == 1) == (mine is not None), (trial, cuts, cost)
            if mine is not None:
                assert mine == out.tolist(), (trial, cuts, cost.tolist(), mine, out.tolist())
                rows = np.diff(mine)
                assert rows.min() >= (mn + 7) // 8 * 8 and rows.max() <= mx and mine[0] == 0 and mine[-1] == h
                cuts = mine
                moved += 1
            else:
                assert out.tolist() == list(cuts)
    assert moved > 300


def test_recut_converges_on_a_fixed_cost_profile():
    """rows of fixed cost (a horizon band twice as dear as floor and ceiling -- level.txt's busiest strip of eight costs 1.3x
    the mean): a few re-cuts bring the strips within 3 % of each other.  (A strip may grow to 1.5 equal strips; a
    profile so peaked that the cheap strips want more than that stops there.)"""
    from pwnfps_amd.dist import equal_cuts, max_strip_rows, recut
    h, world = 2160, 8
    y = np.arange(h, dtype=np.float64)
    per_row = 100.0 + 100.0 * np.exp(-((y - 0.5 * h) / (0.12 * h)) ** 2)
    cuts = equal_cuts(h, world)
    cost_of = lambda c: [int(per_row[c[r]:c[r + 1]].sum()) for r in range(world)]      # noqa: E731
    first = cost_of(cuts)
    for _ in range(8):
        nc = recut(cuts, cost_of(cuts), world, h, 105, max_strip_rows(h, world))
        if nc is None:
            break
        cuts = nc
    last = cost_of(cuts)
    assert max(first) / np.mean(first) > 1.3 and max(last) / np.mean(last) < 1.03, (cuts, last)
