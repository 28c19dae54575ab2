#!/usr/bin/env python3
"""Per-unit costs of one trace launch (PWN_OPT_WAVE_LOG + PWN_DBG_UNIT_COST) for tools/unit_order_sim.py.
    python3 tools/r4/unit_costs.py OUTDIR       -> OUTDIR/<scene>.u16 and the simulator's report per scene"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
gold = os.path.join(ROOT, "tests", "golden")
for level, w, h in (("pwnfps_level", 3840, 2160), ("pwnfps_level", 1280, 720), ("synth64", 1920, 1080), ("synth256", 3840, 2160), ("synth256", 7680, 4320)):
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(os.path.join(gold, "levels", level + ".txt"))
    sph = np.load(os.path.join(gold, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy")))
    r.set_objects(sph)
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn) if level == "pwnfps_level" else np.load(os.path.join(gold, "levels", level + "_cams.npy"))[0]
    sb = np.empty((h, w), np.uint32)
    for _ in range(3):
        r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb)
    r.set_wave_log(True)
    path = os.path.join(out, "%s_%dx%d.u16" % (level, w, h))
    os.environ["PWN_DBG_UNIT_COST"] = path          # (read by the launch -- the variant that writes the costs -- and by pwn_get_stats, which dumps them)
    r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb)
    st = r.stats()
    del os.environ["PWN_DBG_UNIT_COST"]
    r.set_wave_log(False)
    print("== %s %dx%d: launch span %.1f us (wave stamps), %d waves, mean wave residency %.3f" % (
        level, w, h, st["kernel_span"] / 100.0, st["waves"], st["wave_time"] / max(st["waves"] * st["kernel_span"], 1)), flush=True)
    print(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "unit_order_sim.py"), path, str(w), str(h), "--waves", str(st["waves"])],
                         capture_output=True, text=True).stdout, flush=True)
    r.close()
