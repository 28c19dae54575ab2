#!/usr/bin/env python3
"""Wall time of one pwn_trace_screen_centred (main.c:107) at a frame size: one launch per pass against row strips
(PWN_OPT_CALL_STRIPS), pageable and registered host buffers, with and without zbuf.  -> profiles/r5/call_strips.txt

    python tools/r5/call_strips_sweep.py [W H [LEVEL]]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwnfps_amd  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
    level = sys.argv[3] if len(sys.argv) > 3 else "pwnfps_level"
    sph = np.load(os.path.join(GOLD, "spheres_t0.npy")) if level == "pwnfps_level" else np.load(os.path.join(GOLD, "levels", level + "_spheres.npy"))
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(os.path.join(GOLD, "levels", level + ".txt"))
    r.set_objects(sph)
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn)
    sb = np.zeros((h, w), np.uint32)
    zb = np.zeros((h, w), np.float32)
    ref = None

    def leg(name, strips, reg, want_z, env=None):
        nonlocal ref
        for k, v in (env or {}).items():
            os.environ[k] = v
        r.set_call_strips(strips)
        if reg:
            r.host_register(sb)
            r.host_register(zb)
        best, ts = 1e9, []
        for i in range(12):
            r.set_objects(sph)
            t0 = time.perf_counter()
            r.trace_screen_centred(cam, 0.0, want_z=want_z, sbuf=sb, zbuf=zb if want_z else None)
            dt = time.perf_counter() - t0
            if i >= 2:
                ts.append(dt)
        st = r.call_strips_state()
        s = r.stats()
        if ref is None:
            ref = sb.copy()
        same = bool((sb == ref).all())
        if reg:
            r.host_unregister(sb)
            r.host_unregister(zb)
        for k in (env or {}):
            del os.environ[k]
        print("%-44s strips %2d  best %.3f ms  median %.3f ms  %8.1f Mpixels/s   device span %.3f ms (trace..%.3f, blur tail %.3f)  redone %d  same %s" % (
            name, st["strips_last"], min(ts) * 1e3, float(np.median(ts)) * 1e3, w * h / min(ts) / 1e6, s["total_ms"], s["trace_ms"], s["blur_ms"], st["redone"], same), flush=True)

    print("%dx%d %s" % (w, h, level))
    for want_z in (False, True):
        print("-- zbuf %s" % ("wanted" if want_z else "NULL"))
        leg("one launch per pass, pageable", 0, False, want_z)
        leg("one launch per pass, registered", 0, True, want_z)
        leg("default strips, pageable", -1, False, want_z)
        leg("default strips, registered", -1, True, want_z)
        leg("default strips, registered, one copy stream", -1, True, want_z, {"PWN_CALL_COPY_STREAMS": "1"})
        leg("default strips, registered, two copy streams", -1, True, want_z, {"PWN_CALL_COPY_STREAMS": "2"})
        if not want_z:
            for k in (4, 6, 8):
                leg("%d equal strips, registered" % k, k, True, want_z)
            for room in (0, 128, 512, 768):
                leg("default strips, registered, room %d" % room, -1, True, want_z, {"PWN_DBG_STRIP_ROOM": str(room)})
            for first in (96, 128, 160, 192):
                for grow in (1.2, 1.4, 1.6):
                    leg("first %d grow %.2f, registered" % (first, grow), -1, True, want_z, {"PWN_DBG_STRIP_FIRST": str(first), "PWN_DBG_STRIP_GROW": str(grow)})
            for reach in (40, 64):
                leg("default strips, registered, reach %d rows" % reach, -1, True, want_z, {"PWN_DBG_STRIP_REACH": str(reach)})
    r.close()


if __name__ == "__main__":
    main()
