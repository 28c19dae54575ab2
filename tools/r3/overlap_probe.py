#!/usr/bin/env python3
"""Frames API with one or two compute streams (PWN_OPT_FRAME_OVERLAP) at several frame sizes, with and without the
per-frame table upload: ms per frame of 300 frames back to back, 3 slots.   python3 tools/r3/overlap_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402

gold = os.path.join(ROOT, "tests", "golden")
sph = np.load(os.path.join(gold, "spheres_t0.npy"))


def run(w, h, overlap, upload, timing, slots=3, frames=300):
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(os.path.join(gold, "levels", "pwnfps_level.txt"))
    r.set_objects(sph)
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn)
    r.set_frame_overlap(overlap)
    r.set_frame_timing(timing)
    r.frames_config(slots, sbuf=False)
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        for i in range(frames):
            k = i % slots
            if upload:
                r.set_objects(sph)
            if i >= slots:
                r.wait_frame(k)
            r.submit_frame(cam, 0.0, k)
        for i in range(frames - slots, frames):
            r.wait_frame(i % slots)
        best = min(best, (time.perf_counter() - t0) / frames)
    r.close()
    return best * 1e3


for (w, h) in ((320, 240), (1280, 720), (3840, 272), (1920, 1080), (3840, 1080), (3840, 2160)):
    for upload in (0, 1):
        for timing in (0, 8):
            a = run(w, h, 0, upload, timing)
            b = run(w, h, 1, upload, timing)
            print("%4dx%-4d upload %d timing %d: one stream %.4f ms/frame, two streams %.4f (%+.1f %%)" % (w, h, upload, timing, a, b, (b / a - 1) * 100), flush=True)
    for slots in (2, 4):
        a = run(w, h, 0, 1, 0, slots)
        b = run(w, h, 1, 1, 0, slots)
        print("%4dx%-4d upload 1 timing 0 slots %d: one stream %.4f, two streams %.4f (%+.1f %%)" % (w, h, slots, a, b, (b / a - 1) * 100), flush=True)
