#!/usr/bin/env python3
"""The dynamic part of the issue model (tools/issue_model.py), taken on the GPU: for a few scenes one COUNTED frame
(how often a wave64 enters each region of the trace kernel) and, from uncounted frames of the timed build, the trace
launch's duration between HIP events and its mean wave residency.
    python3 tools/region_counts.py [out.json]        (default profiles/r4_region_counts.json)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402

gold = os.path.join(ROOT, "tests", "golden")
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r4_region_counts.json")
RG = ["segs", "setup_slow", "exhausted_w", "wall", "sphere", "floor", "sphrefl", "jitter", "comp1", "comp1_fog", "comp2", "comp2_fog",
      "help", "units", "sphtest", "sphupd", "else", "unit_half", "hc_r2", "hc_out", "portal_wall", "portal_go", "portal_odd", "portal_rot2", "waves"]


def pmc_4k():
    """instruction counts of the 4K level.txt launch from the committed PMC summary, if it is there"""
    path = os.path.join(ROOT, "profiles", "pmc_latest.csv")
    try:
        rows = [l.rstrip("\n").rsplit(",", 3) for l in open(path)]
        v = {r[1]: float(r[3]) for r in rows if len(r) == 4 and "pwn_trace_kernel" in r[0] and "<true" not in r[0]}
        return {k: v[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS")}
    except (OSError, KeyError, ValueError):
        return None


scenes = []
for level, w, h in (("pwnfps_level", 3840, 2160), ("pwnfps_level", 1280, 720), ("synth64", 1920, 1080), ("synth256", 7680, 4320), ("synth256", 3840, 2160)):
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(os.path.join(gold, "levels", level + ".txt"))
    r.set_objects(np.load(os.path.join(gold, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy"))))
    r.set_blur_passes(0)
    r.set_call_strips(0)               # one launch per pass
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn) if level == "pwnfps_level" else np.load(os.path.join(gold, "levels", level + "_cams.npy"))[0]
    sb = np.empty((h, w), np.uint32)
    ms = []
    for _ in range(12):
        r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb)
        ms.append(r.stats()["trace_ms"])
    r.set_wave_log(True)
    r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb)
    sw = r.stats()
    r.set_wave_log(False)
    r.set_counters(True)
    r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb)
    st = r.stats()
    cnt = {"wave_steps": st["wave_steps"]}
    cnt.update({"wp%d" % i: v for i, v in enumerate(st["wave_paths"])})
    cnt.update({k: st["regions"][i] for i, k in enumerate(RG)})
    sc = {"level": level, "w": w, "h": h, "trace_ms": round(float(np.median(ms[2:])), 4),
          "residency": round(sw["wave_time"] / max(sw["waves"] * sw["kernel_span"], 1), 4), "span_ms": round(sw["kernel_span"] / 1e5, 4),
          "rays": st["rays"], "steps": st["steps"], "counts": cnt, "simds": 1024}
    if (level, w, h) == ("pwnfps_level", 3840, 2160):
        sc["pmc"] = pmc_4k()
    scenes.append(sc)
    print(level, w, h, sc["trace_ms"], sc["residency"], cnt, flush=True)
    r.close()
with open(out_path, "w") as f:
    json.dump({"scenes": scenes}, f, indent=1)
