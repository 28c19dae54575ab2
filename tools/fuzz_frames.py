#!/usr/bin/env python3
"""Randomised parity of the FRAMES IN FLIGHT (pwn_submit_frame / pwn_wait_frame) with two compute streams: every frame
of a long sequence has its own camera, clock and sphere set, the host never waits between preparing tables and
submitting, 2..4 slots, and every delivered frame (colour and depth) is the oracle's frame for the inputs of ITS
submit.  What this exercises is not the kernel's arithmetic (tools/fuzz_parity.py does that) but what two frames on
the chip at once share: pre-blur planes, work-queue counter sets, copies of the tables, slot planes.
    python3 tools/fuzz_frames.py [FRAMES [SEED [WxH]]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402  (checker)
import pwnfps_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w, h = (int(v) for v in sys.argv[3].lower().split("x")) if len(sys.argv) > 3 else (1280, 720)
rng = np.random.default_rng(seed)
gold = os.path.join(ROOT, "tests", "golden")
level = os.path.join(gold, "levels", "pwnfps_level.txt")
base = np.load(os.path.join(gold, "spheres_t0.npy"))
O = oracle.Oracle()
O.load_level(level)
bad = 0
done = 0
for slots in (2, 3, 4):
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(level)
    r.frames_config(slots, sbuf=True, zbuf=True)
    scenes = {}
    per = n // 3

    def scene(f):
        sph = base.copy()
        sph["x"] += np.float32(rng.uniform(-0.5, 0.5))
        sph["z"] += np.float32(rng.uniform(-0.5, 0.5))
        sph["cr"] = np.float32(rng.uniform(0.1, 1.0))
        if f % 4 == 3:
            sph = sph[:int(rng.integers(1, len(sph)))]
        cam = pwnfps_amd.spawn_camera((9, 4), ang_y=float(rng.uniform(-3.1, 3.1)), ang_x=float(rng.uniform(-0.4, 0.4)))
        return cam, float(rng.uniform(0.0, 3.0)), sph

    def check(k):
        global bad, done
        fr = r.wait_frame(k % slots)
        cam, sec, sph = scenes.pop(k)
        O.set_spheres(sph)
        ob, oz = O.render(w, h, cam, sec=sec, blur=1)
        ok = bool((fr["sbuf"] == ob).all() and (fr["zbuf"].view(np.uint32) == oz.view(np.uint32)).all())
        done += 1
        if not ok:
            bad += 1
            print("MISMATCH slots %d frame %d: %d colour pixels, %d depth pixels" % (slots, k, int((fr["sbuf"] != ob).sum()),
                                                                                int((fr["zbuf"].view(np.uint32) != oz.view(np.uint32)).sum())), flush=True)
    for f in range(per):
        scenes[f] = scene(f)
        r.set_objects(scenes[f][2])            # level_prepare_render of frame f, before the wait for its slot
        if f >= slots:
            check(f - slots)
        r.submit_frame(scenes[f][0], scenes[f][1], f % slots)
    for k in range(max(0, per - slots), per):
        check(k)
    r.close()
print("fuzz_frames: %d frames, %d mismatches (seed %d, %dx%d, 2 / 3 / 4 slots, two compute streams)" % (done, bad, seed, w, h))
sys.exit(1 if bad else 0)
