#!/usr/bin/env python3
"""In how many wave64 iterations of the walk is NO lane in a cell with spheres or in anything but a room (and none
in a 2-high or '"' cell either)?  Needs a counting build with -DPWN_PLAIN_STATS (trace_walk.inc):
    make -C pwnfps_amd/csrc VARIANT=plain EXTRA=-DPWN_PLAIN_STATS;  PWNHIP_LIB=pwnfps_amd/libpwnhip_plain.so python3 tools/plain_stats.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402

gold = os.path.join(ROOT, "tests", "golden")
for level, w, h in [("pwnfps_level", 3840, 2160), ("synth64", 1920, 1080), ("synth256", 7680, 4320)]:
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(os.path.join(gold, "levels", level + ".txt"))
    sph = np.load(os.path.join(gold, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy")))
    r.set_objects(sph)
    r.set_blur_passes(0)
    r.set_call_strips(0)   # one launch per pass
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn)
    if level != "pwnfps_level":
        cam = np.load(os.path.join(gold, "levels", level + "_cams.npy"))[0]
    r.set_counters(True)
    r.trace_screen_centred(cam, 0.0, want_z=False)
    st = r.stats()
    print(level, "wave iterations", st["wave_steps"], "no lane in (sphere|non-room):", round(st["phase_passes"] / st["wave_steps"], 3),
          " and none in 2-high/dq either:", round(st["phase_lanes"] / st["wave_steps"], 3), "paths", [round(x / st["wave_steps"], 3) for x in st["wave_paths"]])
