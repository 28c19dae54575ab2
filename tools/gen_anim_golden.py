#!/usr/bin/env python3
"""Generate tests/golden/anim.npz: two runs of the game script (pwnfps_amd/script.py,
game.lua restated) rendered frame by frame by the COMPILED REFERENCE (oracle/_ref).
Runs only in the build container.  What it writes is data: per frame the clock,
the camera, the sphere table the script produced and the reference's frame hashes.

  static  16 frames, 0.05 s per frame, mainloop's camera (identity at the spawn, main.c:61-64)
  chase   24 frames, 0.25 s per frame, camera one unit behind the cluster looking along its heading

Frame loop order is main.c:93-140: level_prepare_render, trace at sec_current,
sec_current += tdiff (float), on_tick(sec_current, tdiff).

Every frame is rendered by two builds of the reference: its shipped flags
("hashes") and the same plus -fno-finite-math-only ("hashes_nf").  They agree
unless a pixel's arithmetic left the finite range: the chase camera is axis
aligned, so some rays reach a ramp with ray.y == 0.5*ray.x exactly and the tilt
(trace.h:447-461) divides by zero.  "nonfinite" counts the pixels whose depth is
not finite; frames where it is non-zero are compared against hashes_nf.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from refharness import RefHarness  # noqa: E402
import oracle as orc  # noqa: E402  (FNV routine only)
from pwnfps_amd.script import GameScript, ObjectTable, frame_times  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
W, H = 320, 200


def chase_cam(g):
    vx, vz = float(g.obvx), float(g.obvz)
    cam = np.zeros((4, 4), np.float32)
    cam[0] = (vz, 0, -vx, 0)
    cam[1] = (0, 1, 0, 0)
    cam[2] = (vx, 0, vz, 0)
    cam[3] = (g.obx - vx, 0.5, g.obz - vz, 1)
    return cam


def run(R, N, data, spawn, n, dt, chase):
    T = ObjectTable(data)
    g = GameScript(T)
    secs, ticks = frame_times(n, dt)
    out = dict(sec=[], cam=[], spheres=[], hashes=[], hashes_nf=[], nonfinite=[], centre=[])
    for f in range(n):
        if chase:
            cam = chase_cam(g)
        else:
            cam = np.eye(4, dtype=np.float32)
            cam[3, :3] = (spawn[0] + 0.5, 0.5, spawn[1] + 0.5)
        sph = T.live()
        R.set_spheres(sph)
        pre, z = R.render(W, H, cam, sec=secs[f], blur=0)
        post, _ = R.render(W, H, cam, sec=secs[f], blur=1)
        out["sec"].append(secs[f]); out["cam"].append(cam); out["spheres"].append(sph)
        out["hashes"].append([orc.fnv64(pre), orc.fnv64(post), orc.fnv64(z)])
        N.set_spheres(sph)
        pre_n, z_n = N.render(W, H, cam, sec=secs[f], blur=0)
        post_n, _ = N.render(W, H, cam, sec=secs[f], blur=1)
        out["hashes_nf"].append([orc.fnv64(pre_n), orc.fnv64(post_n), orc.fnv64(z_n)])
        bad = int((~np.isfinite(z)).sum())
        out["nonfinite"].append(bad)
        assert (z.view(np.uint32) == z_n.view(np.uint32)).all()
        assert int((pre != pre_n).sum()) <= bad, "the two builds differ at a finite pixel"
        out["centre"].append((g.obx, g.obz, g.obvx, g.obvz))
        g.on_tick(*ticks[f])
    return out


def main():
    R = RefHarness("tab")
    R.load_level(os.path.join(G, "levels", "pwnfps_level.txt"))
    data, _, spawn = R.get_level()
    N = RefHarness("nf")
    N.load_level(os.path.join(G, "levels", "pwnfps_level.txt"))
    save = {}
    for name, n, dt, chase in (("static", 16, 0.05, False), ("chase", 24, 0.25, True)):
        o = run(R, N, data, spawn, n, dt, chase)
        save[name + "_dt"] = np.float32(dt)
        save[name + "_sec"] = np.array(o["sec"], np.float32)
        save[name + "_cam"] = np.stack(o["cam"])
        save[name + "_spheres"] = np.stack(o["spheres"])
        save[name + "_hashes"] = np.array(o["hashes"])
        save[name + "_hashes_nf"] = np.array(o["hashes_nf"])
        save[name + "_nonfinite"] = np.array(o["nonfinite"], np.int32)
        save[name + "_centre"] = np.array(o["centre"], np.float64)
        print(name, n, "frames; distinct post hashes:", len({h[1] for h in o["hashes"]}),
              "; frames with non-finite pixels:", [i for i, b in enumerate(o["nonfinite"]) if b])
    np.savez_compressed(os.path.join(G, "anim.npz"), **save)


if __name__ == "__main__":
    main()
