#!/usr/bin/env python3
"""Soak: the frames API (two streams) and the row tiling over RCCL with one rank, a minute each, device and host memory before / after.
    python3 tools/r4/soak.py [SECONDS]"""
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import pwnfps_amd  # noqa: E402
import oracle  # noqa: E402  (the frame hash only)

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
gold = os.path.join(ROOT, "tests", "golden")
w, h = 1920, 1080
want = "5828c65f66814845"          # level.txt 1920x1080 with the t = 0 spheres


def mem():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2 ** 20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


r = pwnfps_amd.Renderer(w, h)
r.level_load(os.path.join(gold, "levels", "pwnfps_level.txt"))
sph = np.load(os.path.join(gold, "spheres_t0.npy"))
r.set_objects(sph)
_, _, spawn = r.get_level()
cam = pwnfps_amd.spawn_camera(spawn)
r.frames_config(3, sbuf=True)
for k in range(6):
    if k >= 3:
        r.wait_frame(k % 3)
    r.set_objects(sph)
    r.submit_frame(cam, 0.0, k % 3)
for k in range(3):
    r.wait_frame(k)
m0 = mem()
t0 = time.time()
n = 0
bad = 0
while time.time() - t0 < secs:
    for k in range(300):
        f = r.wait_frame(k % 3) if (n + k) >= 3 else None
        r.set_objects(sph)
        r.submit_frame(cam, 0.0, k % 3)
    n += 300
    f = r.wait_frame(2)
    if oracle.fnv64(f["sbuf"]) != want:
        bad += 1
    r.submit_frame(cam, 0.0, 2)
for k in range(3):
    r.wait_frame(k)
m1 = mem()
print("frames API: %d frames in %.1f s (%.3f ms per frame), %d hash mismatches; device MiB in use %.1f -> %.1f, host peak RSS MiB %.1f -> %.1f" % (
    n, time.time() - t0, (time.time() - t0) / n * 1e3, bad, m0[0], m1[0], m0[1], m1[1]), flush=True)
r.frames_config(0)

uid = pwnfps_amd.Renderer.tiled_unique_id("rccl")
r.tiled_set_timeouts(30, 10)
r.tiled_init(0, 1, uid, "rccl", -1)
for i in range(6):
    r.set_objects(sph)
    r.tiled_submit(cam, 0.0)
    if i >= 2:
        r.tiled_wait()
r.tiled_wait(); r.tiled_wait()
m0 = mem()
t0 = time.time()
n = 0
bad = 0
while time.time() - t0 < secs:
    for i in range(300):
        r.set_objects(sph)
        r.tiled_submit(cam, 0.0)
        if i >= 2:
            fr = r.tiled_wait()
    fr = r.tiled_wait(); fr = r.tiled_wait(host=True)
    n += 300
    if oracle.fnv64(fr["sbuf"]) != want:
        bad += 1
m1 = mem()
print("tiling over RCCL, one rank: %d frames in %.1f s (%.3f ms per frame), %d hash mismatches; device MiB in use %.1f -> %.1f, host peak RSS MiB %.1f -> %.1f; info %s" % (
    n, time.time() - t0, (time.time() - t0) / n * 1e3, bad, m0[0], m1[0], m0[1], m1[1], {k: v for k, v in r.tiled_info().items() if k in ("frames", "groups", "dead", "frames_redone")}), flush=True)
r.tiled_shutdown()
r.close()
