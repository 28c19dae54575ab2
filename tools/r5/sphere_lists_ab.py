#!/usr/bin/env python3
"""A/B of the per-cell sphere lists' form (PWN_SPHERE_LISTS=indexed|inline, tables.h) on one box: the trace launch by itself (frames on ONE
compute stream, HIP events around every launch) and the frame rate on two streams; A/B/A/B.  -> profiles/r5/sphere_lists_ab.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pwnfps_amd
GOLD = os.path.join(ROOT, "tests", "golden")


def measure(level, w, h, lists, has_w=False):
    os.environ["PWN_SPHERE_LISTS"] = lists
    if has_w:
        os.environ["PWN_DBG_FORCE_HASW"] = "1"
    sph = np.load(os.path.join(GOLD, "spheres_t0.npy")) if level == "pwnfps_level" else np.load(os.path.join(GOLD, "levels", level + "_spheres.npy"))
    r = pwnfps_amd.Renderer(w, h)
    os.environ.pop("PWN_DBG_FORCE_HASW", None)
    r.level_load(os.path.join(GOLD, "levels", level + ".txt"))
    r.set_objects(sph)
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn) if level == "pwnfps_level" else np.load(os.path.join(GOLD, "levels", level + "_cams.npy"))[0]
    out = {}
    for two in (False, True):
        r.set_frame_overlap(two)
        r.set_frame_timing(1 if not two else 0)
        r.frames_config(3, sbuf=False)
        ms = []
        for rep in range(3):
            t0 = time.perf_counter()
            n = 300
            for i in range(n):
                s = i % 3
                r.set_objects(sph)
                if i >= 3:
                    f = r.wait_frame(s)
                    if f["timed"] and rep:
                        ms.append(f["trace_ms"])
                r.submit_frame(cam, 0.0, s)
            for i in range(n - 3, n):
                r.wait_frame(i % 3)
            dt = (time.perf_counter() - t0) / n * 1e3
        out["two" if two else "one"] = (dt, float(np.median(ms)) if ms else 0.0)
        r.frames_config(0)
    r.close()
    return out


for level, w, h in (("pwnfps_level", 3840, 2160), ("synth64", 1920, 1080), ("pwnfps_level", 1280, 720), ("synth256", 1920, 1080)):
    for has_w in (False, True):
        if has_w and level != "pwnfps_level":
            continue
        for rep in range(2):
            for lists in ("indexed", "inline"):
                o = measure(level, w, h, lists, has_w)
                print("%-13s %4dx%-4d %s %-8s: trace launch alone %.4f ms (frame on one stream %.4f ms); frame on two streams %.4f ms" % (
                    level, w, h, "HAS_W" if has_w else "     ", lists, o["one"][1], o["one"][0], o["two"][0]), flush=True)
