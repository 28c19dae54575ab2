import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pwnfps_amd
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
w, h = 3840, 2160
r = pwnfps_amd.Renderer(w, h)
r.level_load(os.path.join(GOLD, "levels", "pwnfps_level.txt"))
r.set_objects(np.load(os.path.join(GOLD, "spheres_t0.npy")))
_, _, spawn = r.get_level()
cam = pwnfps_amd.spawn_camera(spawn)
sb = np.zeros((h, w), np.uint32)
r.host_register(sb)
for _ in range(20):
    r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb)
print("copy streams", r.call_strips_state()["copy_streams"])
for rep in range(3):
    for first, grow, room in ((160, 1.2, 256), (160, 1.3, 256), (160, 1.1, 256), (160, 1.2, 384), (160, 1.2, 512)):
        os.environ["PWN_DBG_STRIP_FIRST"] = str(first); os.environ["PWN_DBG_STRIP_GROW"] = str(grow); os.environ["PWN_DBG_STRIP_ROOM"] = str(room)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb); ts.append(time.perf_counter() - t0)
        ts.sort()
        print("first %3d grow %.1f room %3d: best %.4f ms  median %.4f ms  strips %d" % (first, grow, room, ts[0] * 1e3, ts[len(ts) // 2] * 1e3, r.call_strips_state()["strips_last"]))
