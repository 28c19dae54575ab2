#!/usr/bin/env python3
"""The blocking call's strip plan with ONE copy stream (boxes whose second copy queue is not a DMA engine): first strip 128 rows (h / 17)
against 160 / 192 at 4K, alternating in one process.   PWN_CALL_COPY_STREAMS=1 python tools/r5/one_copy_stream_plan.py [W H [FIRST ...]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402
GOLD = os.path.join(ROOT, "tests", "golden")
w = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
h = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
firsts = [int(a) for a in sys.argv[3:]] or [128, 160, 192]
r = pwnfps_amd.Renderer(w, h)
r.level_load(os.path.join(GOLD, "levels", "pwnfps_level.txt"))
r.set_objects(np.load(os.path.join(GOLD, "spheres_t0.npy")))
_, _, spawn = r.get_level()
cam = pwnfps_amd.spawn_camera(spawn)
sb = np.zeros((h, w), np.uint32)
r.host_register(sb)
for _ in range(20):
    r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb)
print("copy streams", r.call_strips_state()["copy_streams"])
for rep in range(4):
    for first in firsts:
        os.environ["PWN_DBG_STRIP_FIRST"] = str(first)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb); ts.append(time.perf_counter() - t0)
        ts.sort()
        print("first %3d: best %.4f ms  median %.4f ms  strips %d" % (first, ts[0] * 1e3, ts[len(ts) // 2] * 1e3, r.call_strips_state()["strips_last"]))
