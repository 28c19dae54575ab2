#!/usr/bin/env python3
"""Divergence profile of the walk loop (pwn_stats.wave_paths): per wave64 iteration, how
often each code path is entered by at least one lane.
    python3 tools/wave_paths.py [W H [level]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402

w = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
h = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
level = sys.argv[3] if len(sys.argv) > 3 else "pwnfps_level"
gold = os.path.join(ROOT, "tests", "golden")
r = pwnfps_amd.Renderer(w, h)
r.level_load(os.path.join(gold, "levels", level + ".txt"))
r.set_objects(np.load(os.path.join(gold, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy"))))
_, _, spawn = r.get_level()
cam = pwnfps_amd.spawn_camera(spawn) if level == "pwnfps_level" else np.load(os.path.join(gold, "levels", level + "_cams.npy"))[0]
r.set_blur_passes(0)
r.set_call_strips(0)   # one launch per pass
r.set_counters(True)
r.trace_screen_centred(cam, 0.0, want_z=False)
st = r.stats()
ws = st["wave_steps"]
names = ["sphere list", "room body", "fog", "two-level transition", "ramp", "portal", "solid", "sphere hit maths"]
print("%s %dx%d: %d rays, %.3f steps/ray, %d wave iterations, active lanes %.3f" % (
    level, w, h, st["rays"], st["steps"] / st["rays"], ws, st["steps"] / (64.0 * ws)))
print("  lane-level: portal crossings / step %.3f, sphere tests / step %.3f" % (st["portals"] / st["steps"], st["sphere_tests"] / st["steps"]))
for n, v in zip(names, st["wave_paths"]):
    print("  %-22s entered in %5.1f %% of wave iterations (%d)" % (n, 100.0 * v / ws, v))
