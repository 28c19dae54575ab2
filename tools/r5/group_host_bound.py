#!/usr/bin/env python3
"""What a group's frame costs the HOST (pwn_init_multi: the caller's thread + one library thread per further member): frames small
enough that the GPU's share vanishes, N members on device 0, three frames in flight, frames left on the devices and frames
delivered to the host; against the one-device frames API at the same size.  -> profiles/r5/group_host_bound.txt

    python tools/r5/group_host_bound.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def rate(r, cam, sph, sbuf, frames=3000, nsl=3):
    r.frames_config(nsl, sbuf=sbuf)
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(frames):
            s = i % nsl
            r.set_objects(sph)
            if i >= nsl:
                r.wait_frame(s)
            r.submit_frame(cam, 0.0, s)
        for i in range(frames - nsl, frames):
            r.wait_frame(i % nsl)
        dt = time.perf_counter() - t0
    r.frames_config(0)
    return dt / frames * 1e6


def blocking(r, cam, sph, w, h, calls=1500):
    sb = np.zeros((h, w), np.uint32)
    r.host_register(sb)
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(calls):
            r.set_objects(sph)
            r.trace_screen_centred(cam, 0.0, want_z=False, sbuf=sb)
        dt = time.perf_counter() - t0
    r.host_unregister(sb)
    return dt / calls * 1e6


def main():
    sph = np.load(os.path.join(GOLD, "spheres_t0.npy"))
    for w, h in ((256, 128), (3840, 272)):
        print("%dx%d frames (us per frame, host loop set_objects / wait / submit)" % (w, h))
        for n in (1, 2, 4, 8):
            if h // n < 16:
                continue
            r = pwnfps_amd.Renderer(w, h, devices=[0] * n) if n > 1 else pwnfps_amd.Renderer(w, h)
            r.level_load(os.path.join(GOLD, "levels", "pwnfps_level.txt"))
            r.set_objects(sph)
            r.set_frame_timing(0)
            _, _, spawn = r.get_level()
            cam = pwnfps_amd.spawn_camera(spawn)
            a = rate(r, cam, sph, False)
            b = rate(r, cam, sph, True)
            c = blocking(r, cam, sph, w, h)
            print("  members %d: resident %.1f   delivered %.1f   blocking call %.1f" % (n, a, b, c), flush=True)
            r.close()


if __name__ == "__main__":
    main()
