import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pwnfps_amd
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
w, h = 64, 32
r = pwnfps_amd.Renderer(w, h)
r.level_load(os.path.join(gold, "levels", "pwnfps_level.txt"))
sph = np.load(os.path.join(gold, "spheres_t0.npy"))
r.set_objects(sph)
_, _, spawn = r.get_level()
cam = pwnfps_amd.spawn_camera(spawn)
# frames API, resident
r.frames_config(3, sbuf=False)
def loop_frames(n, upload):
    t0 = time.perf_counter()
    for i in range(n):
        k = i % 3
        if i >= 3: r.wait_frame(k)
        if upload: r.set_objects(sph)
        r.submit_frame(cam, 0.0, k)
    for i in range(max(0, n - 3), n): r.wait_frame(i % 3)
    return (time.perf_counter() - t0) / n * 1e6
loop_frames(200, True)
print("frames API 64x32: %.1f us/frame with set_objects, %.1f without" % (loop_frames(3000, True), loop_frames(3000, False)))
r.frames_config(0)
r.tiled_init(0, 1, pwnfps_amd.Renderer.tiled_unique_id("shm"), "shm", -1)
def loop_tiled(n, upload):
    t0 = time.perf_counter()
    for i in range(n):
        if upload: r.set_objects(sph)
        r.tiled_submit(cam, 0.0)
        if i >= 2: r.tiled_wait()
    r.tiled_wait(); r.tiled_wait()
    return (time.perf_counter() - t0) / n * 1e6
loop_tiled(200, True)
print("tiled API (world 1) 64x32: %.1f us/frame with set_objects, %.1f without" % (loop_tiled(3000, True), loop_tiled(3000, False)))
# where the host's time goes: seconds inside each call, per frame (3000 frames, 64x32: the GPU is never the limit)
def split(n):
    ts = {"set_objects": 0.0, "tiled_submit": 0.0, "tiled_wait": 0.0}
    t_all = time.perf_counter()
    for i in range(n):
        t0 = time.perf_counter(); r.set_objects(sph); t1 = time.perf_counter(); r.tiled_submit(cam, 0.0); t2 = time.perf_counter()
        ts["set_objects"] += t1 - t0; ts["tiled_submit"] += t2 - t1
        if i >= 2:
            r.tiled_wait(); ts["tiled_wait"] += time.perf_counter() - t2
    r.tiled_wait(); r.tiled_wait()
    t_all = time.perf_counter() - t_all
    return {k: round(v / n * 1e6, 1) for k, v in ts.items()}, round(t_all / n * 1e6, 1)
print("tiled, per frame us inside the calls:", split(3000))
