#!/usr/bin/env python3
"""PWN_DBG_TILED_PROF / PWN_DBG_GROUP_PROF on a group of N members on device 0, small frames: where the host's time per frame goes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pwnfps_amd
GOLD = os.path.join(ROOT, "tests", "golden")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 128)
sbuf = len(sys.argv) > 4 and sys.argv[4] == "sink"
sph = np.load(os.path.join(GOLD, "spheres_t0.npy"))
r = pwnfps_amd.Renderer(w, h, devices=[0] * n)
r.level_load(os.path.join(GOLD, "levels", "pwnfps_level.txt"))
r.set_objects(sph)
r.set_frame_timing(0)
_, _, spawn = r.get_level()
cam = pwnfps_amd.spawn_camera(spawn)
r.frames_config(3, sbuf=sbuf)
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(2000):
        s = i % 3
        r.set_objects(sph)
        if i >= 3:
            r.wait_frame(s)
        r.submit_frame(cam, 0.0, s)
    for i in range(1997, 2000):
        r.wait_frame(i % 3)
    dt = time.perf_counter() - t0
print("members %d %dx%d %s: %.1f us per frame" % (n, w, h, "delivered" if sbuf else "resident", dt / 2000 * 1e6))
r.frames_config(0)
r.close()
