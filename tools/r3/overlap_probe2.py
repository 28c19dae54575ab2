#!/usr/bin/env python3
"""Do the two compute streams of the frames API really run side by side?  ms per frame at 3840x272 and 1280x720."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402

gold = os.path.join(ROOT, "tests", "golden")
sph = np.load(os.path.join(gold, "spheres_t0.npy"))


def run(w, h, overlap, slots=3, frames=300):
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(os.path.join(gold, "levels", "pwnfps_level.txt"))
    r.set_objects(sph)
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn)
    r.set_frame_overlap(overlap)
    r.set_frame_timing(0)
    r.frames_config(slots, sbuf=False)
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        for i in range(frames):
            k = i % slots
            if i >= slots:
                r.wait_frame(k)
            r.submit_frame(cam, 0.0, k)
        for i in range(frames - slots, frames):
            r.wait_frame(i % slots)
        best = min(best, (time.perf_counter() - t0) / frames)
    r.close()
    return best * 1e3


for (w, h) in ((3840, 272), (1280, 720)):
    print("%dx%d: one stream %.4f, two streams %.4f ms/frame  (GPU_MAX_HW_QUEUES=%s)" % (w, h, run(w, h, 0), run(w, h, 1), os.environ.get("GPU_MAX_HW_QUEUES")), flush=True)
