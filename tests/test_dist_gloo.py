"""The row-tiling choreography of pwnfps_amd/dist.py (strip ranges, in-place
all-gather of pre-blur colour, per-strip blur, gather to rank 0) on CPU with
the gloo backend, world_size 2 and 3.  The strip work is done by a CHECKER
backend built on the oracle, so what is under test here is the host logic;
the GPU strip kernels are tested by test_gpu_parity.py."""
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


def _worker(rank, world, port, w, h, case, blur, q, exchange="halo", halo_depth=24.0):
    import sys
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pwnfps_amd.dist import RowTiledFrame, strip_range
        be = OracleStripBackend(w, h, level_path(case["level"]), load_spheres(case["spheres"]))
        fr = RowTiledFrame(w, h, be, torch.device("cpu"), rank=rank, world=world, blur_passes=blur,
                           exchange=exchange, halo_depth=halo_depth)
        assert (fr.y0, fr.y1) == strip_range(h, world, rank)
        out = fr.render(np.array(case["cam"], np.float32), case["sec"], gather_depth=True)
        # second frame with the same inputs must be identical (buffers are reused)
        out = fr.render(np.array(case["cam"], np.float32), case["sec"], gather_depth=True)
        if rank == 0:
            import oracle
            res = [oracle.fnv64(fr.to_host(out)), oracle.fnv64(fr.final_z[:h].numpy())]
        else:
            assert out is None
        # frames in flight: five different frames through the two slots; the last one
        # and (by flushing in the middle) the third one are checked on rank 0
        cams = [np.array(case["cam"], np.float32).reshape(4, 4).copy() for _ in range(5)]
        for i, c in enumerate(cams):
            c[3, 0] += 0.05 * (i % 3)            # frames 0 and 3 are the golden pose
        hashes = []
        for i, c in enumerate(cams):
            fr.submit(c, case["sec"])
            if i in (3, 4):
                o = fr.flush()
                if rank == 0:
                    hashes.append(oracle.fnv64(fr.to_host(o)))
        if rank == 0:
            # reference for frame 4 (pose index 1): a plain render of the same camera
            want4 = oracle.fnv64(fr.to_host(fr.render(cams[4], case["sec"])))
            q.put(tuple(res) + (hashes[0], hashes[1] == want4, fr.halo, fr.halo_misses))
        else:
            fr.render(cams[4], case["sec"])
    finally:
        dist.destroy_process_group()


# exchange / halo depth: the default halo (depth 24: 13 rows at h = 240, never missed by this
# scene), a halo of 1 row that IS missed (every frame falls back to the all-gather), and the
# plain all-gather
@pytest.mark.parametrize("world,blur,exchange,depth", [(2, 1, "halo", 24.0), (2, 0, "halo", 24.0), (3, 1, "halo", 24.0),
                                                       (2, 1, "halo", 0.0), (3, 1, "halo", 0.0), (2, 1, "allgather", 24.0)])
def test_row_tiled_frame_matches_golden(cases, world, blur, exchange, depth):
    case = next(c for c in cases if c["name"] == "level_pose1_320x240")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case["w"], case["h"], case, blur, q, exchange, depth))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    col, z, piped3, piped4_ok, halo, misses = q.get(timeout=5)
    if exchange == "halo" and blur == 1:
        assert halo == int(np.ceil(0.002 * case["h"] * depth)) + 1
        assert (misses > 0) == (depth < 1.0), (halo, misses)     # the 1-row halo must have been caught
    else:
        assert halo == 0 and misses == 0
    assert col == (case["post"] if blur else case["pre"])
    assert z == case["z"]
    assert piped3 == (case["post"] if blur else case["pre"])     # frame 3 of the pipeline = the golden pose
    assert piped4_ok


def test_eight_ranks_uneven_last_strip(cases):
    """World size 8 at 1280x720: strips of 96 rows, the last one 48; the 36-row halo fits the
    shortest strip, so the bounded exchange runs with the geometry of the 8-GPU bench."""
    case = next(c for c in cases if c["name"] == "level_pose1_1280x720")
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case["w"], case["h"], case, 1, q, "halo", 24.0))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    col, z, piped3, piped4_ok, halo, misses = q.get(timeout=5)
    assert halo == 36 and misses == 0
    assert col == case["post"] and z == case["z"] and piped3 == case["post"] and piped4_ok


def test_strip_ranges():
    from pwnfps_amd.dist import strip_range, strip_rows
    for h in (200, 240, 720, 1080, 2160, 4320, 7, 8, 9):
        for world in (1, 2, 3, 4, 8):
            per = strip_rows(h, world)
            assert per % 8 == 0 and per * world >= h
            rows = []
            for r in range(world):
                y0, y1 = strip_range(h, world, r)
                assert 0 <= y0 <= y1 <= h and (y0 % 8 == 0 or y0 == h)
                rows += list(range(y0, y1))
            assert rows == list(range(h))
