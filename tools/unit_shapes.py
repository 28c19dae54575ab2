#!/usr/bin/env python3
"""What would another unit shape do to the walk loop's lane occupancy?  (CPU only: the oracle's step map.)

A wave traces the 64 pixels of a unit in lock step, segment by segment (primary ray, then each reflection), so a
unit costs sum over segments of max over its pixels of the walk-loop iterations; the useful share of that is the
"walk lane fraction" the kernel's counters report.  This tool takes the per-pixel, per-segment iteration counts
from the oracle (pwno_step_map) and evaluates several shapes of 64 pixels.
    python3 tools/unit_shapes.py [level W H [pose]] ..."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (analysis of the checker's own counts; nothing of the product runs here)

GOLD = os.path.join(ROOT, "tests", "golden")
SHAPES = [(64, 1), (32, 2), (16, 4), (8, 8), (4, 16)]


def spawn_camera(spawn):
    cam = np.eye(4, dtype=np.float32).reshape(16)
    cam[12], cam[13], cam[14] = spawn[0] + 0.5, 0.5, spawn[1] + 0.5
    return cam


def one(level, w, h, pose=0):
    o = oracle.Oracle()
    o.load_level(os.path.join(GOLD, "levels", level + ".txt"))
    sph = np.load(os.path.join(GOLD, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy")))
    o.set_spheres(sph)
    if level == "pwnfps_level":
        cam = spawn_camera(o.get_level()[2])
    else:
        cam = np.load(os.path.join(GOLD, "levels", level + "_cams.npy"))[pose]
    L = oracle.lib()
    L.pwno_step_map.argtypes = [C.c_void_p]
    L.pwno_step_map.restype = None
    m = np.zeros((h, w, 3), np.uint16)
    L.pwno_step_map(m.ctypes.data)
    try:
        o.trace_rows(w, h, 0, h, cam)
    finally:
        L.pwno_step_map(None)
    total = int(m.sum(dtype=np.int64))
    print("%s %dx%d pose %d: %.3f steps per pixel" % (level, w, h, pose, total / (w * h)))
    for uw, uh in SHAPES:
        ph, pw = (-h) % uh, (-w) % uw
        mm = np.pad(m, ((0, ph), (0, pw), (0, 0)))
        H, W = mm.shape[:2]
        u = mm.reshape(H // uh, uh, W // uw, uw, 3).max(axis=(1, 3))
        it = int(u.sum(dtype=np.int64))
        # ... and without the lock step between segments (a lane starts its next segment when it is ready):
        # max over the unit of the pixel's total
        free = int(mm.sum(axis=2, dtype=np.int64).reshape(H // uh, uh, W // uw, uw).max(axis=(1, 3)).sum())
        print("   %2dx%-2d  wave iterations %10d   lane fraction %.3f   (segments not in lock step: %10d, %.3f)" % (
            uw, uh, it, total / (64.0 * it), free, total / (64.0 * free)))


if __name__ == "__main__":
    a = sys.argv[1:]
    if not a:
        a = ["pwnfps_level", "3840", "2160", "0"]
    while a:
        lvl, w, h = a[0], int(a[1]), int(a[2])
        pose = int(a[3]) if len(a) > 3 and a[3].isdigit() and len(a[3]) < 3 else 0
        a = a[4:] if len(a) > 3 and a[3].isdigit() and len(a[3]) < 3 else a[3:]
        one(lvl, w, h, pose)
