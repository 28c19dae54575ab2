import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import pwnfps_amd
GOLD = '/root/repo/tests/golden'
w, h = 3840, 2160
sph = np.load(GOLD + "/spheres_t0.npy")
for streams in ("2", "1", "2"):
    os.environ["PWN_CALL_COPY_STREAMS"] = streams
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(GOLD + "/levels/pwnfps_level.txt"); r.set_objects(sph)
    _, _, spawn = r.get_level(); cam = pwnfps_amd.spawn_camera(spawn)
    sb = np.zeros((h, w), np.uint32); r.host_register(sb)
    ts = []
    for i in range(120):
        r.set_objects(sph)
        t0 = time.perf_counter(); r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb); ts.append((time.perf_counter() - t0) * 1e3)
    print("copy streams %s:" % streams, " ".join("%.3f" % t for t in ts[:40]), "... mean of calls 80-120: %.3f" % np.mean(ts[80:]), flush=True)
    r.host_unregister(sb); r.close()
