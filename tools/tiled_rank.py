#!/usr/bin/env python3
"""One rank of a row-tiled run through the C ABI (pwn_tiled_*): the program the multi-process
tests and experiments start once per rank.

    python3 tools/tiled_rank.py RANK WORLD IDFILE TRANSPORT W H LEVEL FRAMES [HALO [DEVICE]]

Rank 0 creates the group id and writes it to IDFILE (.tmp + rename); the others wait for the
file.  Every frame has its own camera, sec_current and sphere set (tools/tiled_rank.scene), so a
frame that is delivered late or from the wrong buffers shows.  Rank 0 prints one line per frame:
    frame K fnv64 HASH redone R
and every rank a line `info {...}` (pwn_tiled_info) and per frame `rows K Y0 Y1 COST` (the rows it traced of that frame and
what they cost).  TILED_DEPTH=n: n frames in flight (default 3, at most PWN_TILED_SLOTS - 1).  TILED_ROTATE=1: pwn_tiled_gather_root(ROTATE), the `frame` lines then come from the frame's root, rank K mod
WORLD.  TILED_BALANCE=k: pwn_tiled_balance(k) (moving cuts); TILED_CUTS_AT="K:c0,c1,..;K2:.." calls
pwn_tiled_set_cuts in front of frame K.  TILED_HOSTSINK=1: frames are delivered into POSIX shared
memory by every rank (pwn_tiled_host_sink); the other ranks then print `seen K fnv64 HASH` too.
Failure containment (tests/test_gpu_deadlines.py): TILED_TIMEOUTS="INIT_S,WAIT_S" -> pwn_tiled_set_timeouts;
TILED_DIE_AT="K:RANK": that rank leaves the process (exit code 17) right before it would submit frame K.  A rank whose
tiling call fails prints `error CODE after S s: MESSAGE`, then `info {...}` if there still is a tiling, and exits with 42.
TILED_COUNTERS=1 / TILED_WAVELOG=1: the counting options on; every rank prints `stats K RAYS WAVES` per delivered frame.
TILED_TIMING=1: every frame timed (PWN_OPT_FRAME_TIMING 1); `times K TRACE BLUR HALO GATHER` per delivered frame.
Every rank ends with a line `host frames N in S s = MS ms per frame; pwn_tiled_submit US us per frame` (+ the means of the
timed kernels and exchanges); TILED_QUIET=1: only the last frame is hashed and printed, so that MS is the loop's own time.
PWN_TILED_SELF=1 (read by the library, one rank over RCCL): the rank sends itself what a rank of a real tiling sends."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def scene(k, base, spawn):
    """frame k of the test sequence: camera, sec_current, spheres"""
    import pwnfps_amd
    sph = base.copy()
    sph["x"] += np.float32(0.09 * k)
    if k:                                   # frame 0 is the reference's own scene at t = 0 (golden hashes exist)
        sph["cb"] = np.float32(0.3 + 0.1 * (k % 4))
    if k % 3 == 1:
        sph = sph[:max(1, len(sph) - 2)]
    return pwnfps_amd.spawn_camera(spawn, ang_y=0.21 * k, ang_x=0.03 * k), 0.1 * k, sph


def shm_path(idfile):
    return "/dev/shm/pwn_frames_" + os.path.basename(idfile) + "_%d" % os.getppid()


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    idfile, transport = sys.argv[3], sys.argv[4]
    w, h, level, frames = int(sys.argv[5]), int(sys.argv[6]), sys.argv[7], int(sys.argv[8])
    halo = int(sys.argv[9]) if len(sys.argv) > 9 else -1
    device = int(sys.argv[10]) if len(sys.argv) > 10 else 0
    import pwnfps_amd
    import oracle  # checker: the frame hash only
    gold = os.path.join(ROOT, "tests", "golden")
    r = pwnfps_amd.Renderer(w, h, device=device)
    r.level_load(os.path.join(gold, "levels", level + ".txt"))
    base = np.load(os.path.join(gold, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy")))
    _, _, spawn = r.get_level()
    if os.environ.get("TILED_BLUR") is not None:
        r.set_blur_passes(int(os.environ["TILED_BLUR"]))
    if rank == 0:
        uid = pwnfps_amd.Renderer.tiled_unique_id(transport)
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120:
                sys.exit("rank %d: no id file" % rank)
            time.sleep(0.01)
        uid = open(idfile, "rb").read()
    hostsink = os.environ.get("TILED_HOSTSINK") == "1"
    if hostsink and rank == 0:
        # the frames' host memory: POSIX shared memory that every rank maps (created before the id file appears)
        with open(shm_path(idfile), "wb") as f:
            f.truncate(pwnfps_amd._lib.PWN_TILED_SLOTS * w * h * 4)
    t_call = [time.time()]

    def failed(e):
        print("error %d after %.2f s: %s" % (e.code, time.time() - t_call[0], e), flush=True)
        try:
            print("info " + json.dumps(r.tiled_info()), flush=True)
        except pwnfps_amd.PwnError:
            pass
        sys.stdout.flush()
        os._exit(42)                        # (a helper thread may still sit inside RCCL: no interpreter shutdown)
    if os.environ.get("TILED_TIMEOUTS"):
        a, b = (float(v) for v in os.environ["TILED_TIMEOUTS"].split(","))
        r.tiled_set_timeouts(a, b)
    if os.environ.get("TILED_PREFLIGHT"):
        print("preflight " + json.dumps(r.tiled_preflight()), flush=True)
    try:
        r.tiled_init(rank, world, uid, transport, halo)
    except pwnfps_amd.PwnError as e:
        failed(e)
    if os.environ.get("TILED_COUNTERS"):
        r.set_counters(True)
    if os.environ.get("TILED_WAVELOG"):
        r.set_wave_log(True)
    if os.environ.get("TILED_TIMING"):
        r.set_frame_timing(1)
    die_at = None
    if os.environ.get("TILED_DIE_AT"):
        k, who = os.environ["TILED_DIE_AT"].split(":")
        die_at = int(k) if int(who) == rank else None
    mm = None
    if hostsink:
        import mmap
        fd = os.open(shm_path(idfile), os.O_RDWR)
        mm = mmap.mmap(fd, pwnfps_amd._lib.PWN_TILED_SLOTS * w * h * 4)
        os.close(fd)
        r.tiled_host_sink(mm)

    if os.environ.get("TILED_ROTATE") == "1":
        r.tiled_gather_root(True)
    if os.environ.get("TILED_BALANCE") is not None:
        r.tiled_balance(int(os.environ["TILED_BALANCE"]))
    cuts_at = {}
    for part in filter(None, os.environ.get("TILED_CUTS_AT", "").split(";")):
        k, c = part.split(":")
        cuts_at[int(k)] = [int(v) for v in c.split(",")]

    fr_rows = [0, 0]
    enq = []
    tms = {"trace_ms": [], "blur_ms": [], "halo_ms": [], "gather_ms": [], "frame_ms": []}
    quiet = bool(os.environ.get("TILED_QUIET"))          # only the last frame is hashed and printed: the loop's own time per frame means something then
    t_run = time.time()

    def deliver(k):
        t_call[0] = time.time()
        try:
            fr = r.tiled_wait(host=(not quiet) or k + 1 >= frames)
        except pwnfps_amd.PwnError as e:
            failed(e)
        assert fr["seq"] == k + 1
        fr_rows[0], fr_rows[1] = fr["y0"], fr["y1"]
        if not quiet:
            print("rows %d %d %d %d" % (k, fr["y0"], fr["y1"], fr["cost"]), flush=True)
        if os.environ.get("TILED_COUNTERS") or os.environ.get("TILED_WAVELOG"):
            st = r.stats()
            print("stats %d %d %d" % (k, st["rays"], st["waves"]), flush=True)
        if os.environ.get("TILED_TIMING"):
            if not quiet:
                print("times %d %.6f %.6f %.6f %.6f" % (k, fr["trace_ms"], fr["blur_ms"], fr["halo_ms"], fr["gather_ms"]), flush=True)
            for key in tms:
                if fr[key] > 0:
                    tms[key].append(fr[key])
        enq.append(fr["enqueue_us"])
        if quiet and k + 1 < frames:
            return
        if (rank == 0 and hostsink) or (not hostsink and fr.get("sbuf") is not None):
            # (without a host sink: the frame's root has it -- rank 0, or rank k mod world with TILED_ROTATE=1)
            assert hostsink or fr["root"] == rank
            print("frame %d fnv64 %s redone %d" % (k, oracle.fnv64(fr["sbuf"]), int(fr["redone"])), flush=True)
        elif hostsink:
            # with a host sink every rank holds the whole frame when its wait returns
            print("seen %d fnv64 %s" % (k, oracle.fnv64(fr["sbuf"])), flush=True)
    depth = min(max(int(os.environ.get("TILED_DEPTH", "3")), 1), pwnfps_amd._lib.PWN_TILED_SLOTS - 1)      # frames in flight
    for k in range(frames):
        cam, sec, sph = scene(0 if os.environ.get("TILED_SAME_SCENE") else k, base, spawn)
        r.set_objects(sph)
        if k in cuts_at:
            r.tiled_set_cuts(cuts_at[k])
        if die_at == k:
            sys.stdout.flush()
            os._exit(17)
        t_call[0] = time.time()
        try:
            r.tiled_submit(cam, sec)
        except pwnfps_amd.PwnError as e:
            failed(e)
        if k >= depth - 1:
            deliver(k - (depth - 1))
    for k in range(max(0, frames - (depth - 1)), frames):
        deliver(k)
    if os.environ.get("TILED_COUNTERS") or os.environ.get("TILED_WAVELOG"):
        st = r.stats()                       # every frame delivered: the last launch's own counts
        print("laststats %d %d %d %d" % (st["rays"], st["waves"], fr_rows[0], fr_rows[1]), flush=True)
    if enq:
        print("host frames %d in %.4f s = %.4f ms per frame; pwn_tiled_submit %.1f us per frame (median %.1f)%s" % (
            frames, time.time() - t_run, (time.time() - t_run) / frames * 1e3, float(np.mean(enq)), float(np.median(enq)),
            "".join("; %s %.4f" % (k, float(np.mean(v))) for k, v in tms.items() if v)), flush=True)
    print("info " + json.dumps(r.tiled_info()), flush=True)
    print("cuts " + json.dumps([int(v) for v in r.tiled_get_cuts()[0]]), flush=True)
    r.tiled_shutdown()
    r.close()
    if mm is not None:                       # (the mapping itself goes with the process: views of it are still alive)
        if rank == 0:
            time.sleep(0.5)
            try:
                os.unlink(shm_path(idfile))
            except OSError:
                pass


if __name__ == "__main__":
    main()
