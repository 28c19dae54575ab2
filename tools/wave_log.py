#!/usr/bin/env python3
"""When is each wave64 of the trace kernel resident?  Renders one counted frame with
PWN_DBG_WAVE_LOG set and prints the distribution of wave start and end times.
    python3 tools/wave_log.py [W H [level]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = "/tmp/pwn_wave_log.bin"
os.environ["PWN_DBG_WAVE_LOG"] = path
import pwnfps_amd  # noqa: E402

w = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
h = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
level = sys.argv[3] if len(sys.argv) > 3 else "pwnfps_level"
gold = os.path.join(ROOT, "tests", "golden")
r = pwnfps_amd.Renderer(w, h)
r.level_load(os.path.join(gold, "levels", level + ".txt"))
r.set_objects(np.load(os.path.join(gold, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy"))))
r.set_blur_passes(0)
r.set_call_strips(0)   # one launch per pass
_, _, spawn = r.get_level()
cam = pwnfps_amd.spawn_camera(spawn)
if level != "pwnfps_level":
    cam = np.load(os.path.join(gold, "levels", level + "_cams.npy"))[0]
for _ in range(3):
    r.trace_screen_centred(cam, 0.0, want_z=False)
if os.environ.get("WAVE_LOG_COUNTED"):
    r.set_counters(True)
r.set_wave_log(True)
for _ in range(4):
    r.trace_screen_centred(cam, 0.0, want_z=False)
    st = r.stats()
log = np.fromfile(path, np.uint64).reshape(-1, 2)[1:]
log = log[log[:, 1] != 0]
t0 = log[:, 0].min()
b = (log[:, 0] - t0) / 100.0      # us
e = (log[:, 1] - t0) / 100.0
span = e.max()
print("%s %dx%d: %d waves logged, span %.1f us, counted-frame kernel %.1f us" % (level, w, h, len(log), span, st["trace_ms"] * 1e3))
print("mean wave residency %.3f" % ((e - b).sum() / (len(log) * span)))
q = [0, 1, 10, 25, 50, 75, 90, 99, 100]
print("start us  percentiles", dict(zip(q, np.round(np.percentile(b, q), 1))))
print("end us    percentiles", dict(zip(q, np.round(np.percentile(e, q), 1))))
print("life us   percentiles", dict(zip(q, np.round(np.percentile(e - b, q), 1))))
late = b > 0.1 * span
print("waves that start after 10 %% of the span: %d (mean life %.1f us)" % (late.sum(), (e - b)[late].mean() if late.any() else 0))
