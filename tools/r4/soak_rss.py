#!/usr/bin/env python3
"""Does the host's memory grow with the number of frames?  Current RSS (/proc/self/statm) at checkpoints of a frames-API loop, in four forms.
    python3 tools/r4/soak_rss.py [SECONDS_PER_FORM]"""
import gc
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402
from pwnfps_amd import _lib  # noqa: E402
import ctypes as C  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
gold = os.path.join(ROOT, "tests", "golden")
w, h = 1280, 720


def rss():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2 ** 20


def form(name, objects, sbuf, raw):
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(os.path.join(gold, "levels", "pwnfps_level.txt"))
    sph = np.load(os.path.join(gold, "spheres_t0.npy"))
    r.set_objects(sph)
    _, _, spawn = r.get_level()
    cam = np.ascontiguousarray(pwnfps_amd.spawn_camera(spawn), np.float32)
    r.frames_config(3, sbuf=sbuf)
    lib = _lib.lib
    fr = _lib.Frame()
    marks = []
    n = 0
    t0 = time.time()
    nxt = 0.0
    while time.time() - t0 < secs:
        for k in range(300):
            if n + k >= 3:
                if raw:
                    lib.pwn_wait_frame(r._ctx, k % 3, C.byref(fr))
                else:
                    r.wait_frame(k % 3)
            if objects:
                r.set_objects(sph)
            if raw:
                lib.pwn_submit_frame(r._ctx, cam.ctypes.data, C.c_float(0.0), k % 3)
            else:
                r.submit_frame(cam, 0.0, k % 3)
        n += 300
        if time.time() - t0 >= nxt:
            gc.collect()
            marks.append((n, rss()))
            nxt += secs / 5.0
    for k in range(3):
        r.wait_frame(k)
    gc.collect()
    marks.append((n, rss()))
    per = (marks[-1][1] - marks[1][1]) * 1024.0 * 1024.0 / max(marks[-1][0] - marks[1][0], 1)
    print("%-44s %s -> %.0f bytes per frame" % (name, " ".join("%d:%.0fMiB" % m for m in marks), per), flush=True)
    r.frames_config(0)
    r.close()


form("objects + sbuf (python wrappers)", True, True, False)
form("no objects, sbuf", False, True, False)
form("objects, frames stay on the device", True, False, False)
form("no objects, resident, raw ctypes calls", False, False, True)
