#!/usr/bin/env python3
"""Trace-kernel time of the two schedulers (PWN_OPT_SCHEDULER) and of the refill limits, on the
three BASELINE scenes: HIP-event time of the trace kernel over N frames (median, min), active
lanes in the walk loop, lanes with a ray to shade per phase-A pass.
    python3 tools/sched_sweep.py [frames [limits...]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
limits = [int(v) for v in sys.argv[2:]] or [0, 4, 8, 16, 24, 32, 48]
gold = os.path.join(ROOT, "tests", "golden")
scenes = [("pwnfps_level", 3840, 2160), ("synth64", 1920, 1080), ("synth256", 7680, 4320), ("pwnfps_level", 1280, 720)]
if os.environ.get("SWEEP_SCENES"):
    scenes = [scenes[int(i)] for i in os.environ["SWEEP_SCENES"].split(",")]
for level, w, h in scenes:
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(os.path.join(gold, "levels", level + ".txt"))
    sph = np.load(os.path.join(gold, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy")))
    r.set_objects(sph)
    r.set_blur_passes(0)
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn)
    if level != "pwnfps_level":
        cam = np.load(os.path.join(gold, "levels", level + "_cams.npy"))[0]
    r.frames_config(2, sbuf=True)

    def run(tag):
        r.submit_frame(cam, 0.0, 0); r.wait_frame(0)
        t = []
        for f in range(frames):
            r.submit_frame(cam, 0.0, f & 1)
            if f:
                t.append(r.wait_frame((f - 1) & 1)["trace_ms"])
        t.append(r.wait_frame((frames - 1) & 1)["trace_ms"])
        r.set_counters(True)
        r.submit_frame(cam, 0.0, 0); r.wait_frame(0)
        st = r.stats()
        r.set_counters(False)
        lane = st["steps"] / max(64.0 * st["wave_steps"], 1)
        pa = st["phase_lanes"] / max(64.0 * st["phase_passes"], 1) if st["phase_passes"] else float("nan")
        print("%-13s %4dx%-4d %-12s trace ms med %.4f min %.4f | walk lanes %.3f  wave-iterations %9d | phase-A passes %8d lanes %.3f" % (
            level, w, h, tag, float(np.median(t)), min(t), lane, st["wave_steps"], st["phase_passes"], pa), flush=True)
    r.set_scheduler("units")
    run("units")
    r.set_scheduler("refill")
    for lim in limits:
        r.set_refill_limit(lim)
        run("refill L=%d" % lim)
    r.frames_config(0)
    r.close()
