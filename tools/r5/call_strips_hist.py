#!/usr/bin/env python3
"""Distribution of the blocking call's wall time over many calls (4K, level.txt, registered sbuf): one against two copy streams.
    python tools/r5/call_strips_hist.py [CALLS]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pwnfps_amd
GOLD = os.path.join(ROOT, "tests", "golden")
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 600
w, h = 3840, 2160
sph = np.load(os.path.join(GOLD, "spheres_t0.npy"))
for streams in ("1", "2", "auto", "1", "2", "auto"):
    os.environ.pop("PWN_CALL_COPY_STREAMS", None)
    if streams != "auto":
        os.environ["PWN_CALL_COPY_STREAMS"] = streams
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(os.path.join(GOLD, "levels", "pwnfps_level.txt"))
    r.set_objects(sph)
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn)
    sb = np.zeros((h, w), np.uint32)
    r.host_register(sb)
    ts = []
    for i in range(calls + 20):
        r.set_objects(sph)
        t0 = time.perf_counter()
        r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[20:]) * 1e3
    print("copy streams %s: min %.3f p10 %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f ms; calls over 0.9 ms: %d of %d" % (
        streams, ts.min(), *np.percentile(ts, [10, 50, 90, 99]), ts.max(), int((ts > 0.9).sum()), len(ts)), flush=True)
    r.host_unregister(sb)
    r.close()
