#!/usr/bin/env python3
"""Render a few frames through the C ABI; the program to put after `--` in
rocprofv3 runs (kernel-trace or --pmc passes).
    python3 tools/prof_frame.py [W H [frames [level [blur]]]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402

w = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
h = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 5
level = sys.argv[4] if len(sys.argv) > 4 else "pwnfps_level"
blur = int(sys.argv[5]) if len(sys.argv) > 5 else 1
gold = os.path.join(ROOT, "tests", "golden")
r = pwnfps_amd.Renderer(w, h)
r.level_load(os.path.join(gold, "levels", level + ".txt"))
sph = np.load(os.path.join(gold, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy")))
if os.environ.get("PWN_NOSPH"):
    sph = sph[:0]
r.set_objects(sph)
r.set_blur_passes(blur)
if not os.environ.get("PWN_PROF_STRIPS"):
    r.set_call_strips(0)             # one launch per pass: the counters are per whole-frame dispatch
_, _, spawn = r.get_level()
cam = pwnfps_amd.spawn_camera(spawn)
if level != "pwnfps_level":
    cam = np.load(os.path.join(gold, "levels", level + "_cams.npy"))[0]
sb = np.empty((h, w), np.uint32)
tr, bl = [], []
for _ in range(frames):
    r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb)
    st = r.stats()
    tr.append(st["trace_ms"])
    bl.append(st["blur_ms"])
hsh = "-"
if os.environ.get("PWN_HASH"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle  # checker only
    hsh = oracle.fnv64(sb)
if os.environ.get("PWN_COUNT"):
    r.set_counters(True)
    r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb)
    st = r.stats()
    print("rays %d steps %d wave_steps %d -> steps/ray %.3f, wave iterations/ray-wave %.2f, walk lane fraction %.3f" % (
        st["rays"], st["steps"], st["wave_steps"], st["steps"] / st["rays"], st["wave_steps"] / (st["rays"] / 64.0),
        st["steps"] / (64.0 * st["wave_steps"])))
print("%s %s %dx%d trace_ms min %.4f med %.4f blur_ms min %.4f fnv %s" % (
    os.path.basename(os.environ.get("PWNHIP_LIB", "libpwnhip.so")), level, w, h,
    min(tr), sorted(tr)[len(tr) // 2], min(bl), hsh))
