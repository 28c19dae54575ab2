#!/usr/bin/env python3
"""How far behind its submissions should the host of a tiled rank collect frames?  One rank (world 1: shm transport, or RCCL
exchanging with itself under PWN_TILED_SELF=1), a lean loop -- set_objects / pwn_tiled_submit / pwn_tiled_wait, nothing else per
frame -- with DEPTH frames in flight, no timing events; prints ms per frame and the seconds inside each call.

    python3 tools/tiled_depth.py W H FRAMES DEPTH[,DEPTH...] [shm|rccl]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pwnfps_amd
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    depths = [int(x) for x in sys.argv[4].split(",")]
    transport = sys.argv[5] if len(sys.argv) > 5 else "shm"
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(os.path.join(gold, "levels", "pwnfps_level.txt"))
    sph = np.load(os.path.join(gold, "spheres_t0.npy"))
    r.set_objects(sph)
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn)
    r.tiled_init(0, 1, pwnfps_amd.Renderer.tiled_unique_id(transport), transport, -1)
    r.set_frame_timing(0)
    tag = "%dx%d %s%s" % (w, h, transport, " self-exchange" if os.environ.get("PWN_TILED_SELF") else "")

    def loop(n, depth, split):
        ts = [0.0, 0.0, 0.0]
        clock = time.perf_counter
        t_all = clock()
        for i in range(n):
            if split:
                t0 = clock(); r.set_objects(sph); t1 = clock(); r.tiled_submit(cam, 0.0); t2 = clock()
                if i >= depth - 1: r.tiled_wait()
                t3 = clock()
                ts[0] += t1 - t0; ts[1] += t2 - t1; ts[2] += t3 - t2
            else:
                r.set_objects(sph); r.tiled_submit(cam, 0.0)
                if i >= depth - 1: r.tiled_wait()
        for _ in range(min(depth - 1, n)): r.tiled_wait()
        return (clock() - t_all) / n * 1e3, [x / n * 1e6 for x in ts]

    for d in depths:
        loop(200, d, False)
        a = [loop(n, d, False)[0] for _ in range(3)]
        b, ts = loop(n, d, True)
        print("%s, %d in flight: %.4f / %.4f / %.4f ms per frame; inside the calls: set_objects %.1f us, tiled_submit %.1f, tiled_wait %.1f (loop %.4f ms)" % (
            tag, d, a[0], a[1], a[2], ts[0], ts[1], ts[2], b), flush=True)
    r.tiled_shutdown()


if __name__ == "__main__":
    main()
