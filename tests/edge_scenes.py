"""Scenes at the edges of the input domain, shared by the oracle-vs-reference
test (CPU, where /root/reference is mounted) and the HIP-vs-oracle test (GPU).

ref_safe = False marks inputs on which the reference itself has undefined
behaviour (level_part_add_bbox writes outside parts[][] for a sphere whose
bounding box leaves the grid, level.h:5-17); there the oracle and the HIP
path agree on "cells outside the grid are skipped" and only they are compared.
"""
from collections import namedtuple

import numpy as np

Scene = namedtuple("Scene", "name text w h sec cam spheres ref_safe")


def _cam(x, y, z, ang_y=0.0, ang_x=0.0):
    cy, sy, cx, sx = np.cos(ang_y), np.sin(ang_y), np.cos(ang_x), np.sin(ang_x)
    cam = np.eye(4, dtype=np.float32)
    cam[:3, :3] = (np.array([[1, 0, 0], [0, cx, sx], [0, -sx, cx]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])).astype(np.float32)
    cam[3, :3] = (x, y, z)
    return cam


def scenes(sphere_dtype):
    rng = np.random.default_rng(99)
    open_level = "\n".join("".join(rng.choice(list(';;;;$#&"<>,^.'), 64)) for _ in range(64)) + "\n"
    letters = "\n".join(["." * 64] * 3 + ["..;;;;A;;;;a;;;;B;;;b;;;m;;;;N;;;y;;;Z;;." + "." * 23]
                        + ["..;;;;;;;;;;;;;;;;;;;;;;;;;;;;;;;;;;;;;;." + "." * 23] * 3) + "\n"

    def spheres(n, border=False):
        s = np.zeros(n, sphere_dtype)
        for i in range(n):
            s[i] = (rng.uniform(0.02, 0.2), rng.choice([0.0, 0.5]), 9.5 + rng.uniform(-0.4, 0.4), rng.uniform(0.1, 0.9),
                    5.5 + rng.uniform(-0.4, 0.4), *rng.uniform(0, 1, 3))
        if border:
            s["x"][:5] = (0.05, 63.95, 31.5, 0.2, 63.7)
            s["z"][:5] = (31.5, 31.5, 0.05, 0.2, 63.9)
        return s

    spawn = _cam(9.5, 0.5, 4.5, 0.3, -0.2)
    out = []
    # tiny and ragged frames, 300 spheres binned into one cell
    for w, h in ((4, 1), (1, 1), (33, 7), (320, 200), (68, 40)):
        out.append(Scene("level_%dx%d" % (w, h), None, w, h, 0.0, spawn, spheres(300), True))
    out.append(Scene("empty_level", "\n", 64, 32, 0.0, spawn, spheres(5), True))
    # no walls at the border: rays leave the grid, get_cell clamps per axis (util.h:151-158)
    out.append(Scene("open_outside", open_level, 128, 64, 3.25, _cam(-3.5, 0.5, 70.25, 0.9, 0.1), spheres(40), True))
    out.append(Scene("open_inside", open_level, 128, 64, 3.25, _cam(31.5, 0.4, 30.5, 2.2, -0.3), spheres(40), True))
    out.append(Scene("open_border_spheres", open_level, 128, 64, 1.0, _cam(1.5, 0.4, 31.5, 4.5, 0.0), spheres(40, True), False))
    # unpaired / lower-case / foreign portal letters (level.h:144-178, trace.h:514-559)
    out.append(Scene("letters", letters, 128, 64, 0.5, _cam(3.5, 0.5, 4.5, 1.2, 0.0), spheres(5), True))
    # large sec_current: the floor ripple angle leaves sinf/cosf's fast range (trace.h:42-46)
    out.append(Scene("sec_5000", None, 160, 96, 5000.0, spawn, spheres(14), True))
    out.append(Scene("sec_3e7", None, 64, 32, 3.0e7, spawn, spheres(14), True))
    # sphere parameters nobody would script on purpose: zero / tiny / negative radius (c/r^2
    # divides by zero, trace.h:277), a sphere around the camera, reflectivity outside [0,1],
    # colours that are negative, huge or NaN, coincident spheres (strict '<' keeps the first,
    # trace.h:281), spheres above and below the rooms
    weird = {
        "r0": [(0.0, 0.5, 9.5, 0.3, 5.5, 1, 1, 1)],
        "r_tiny": [(1e-20, 0.5, 9.5, 0.3, 5.5, 1, 1, 1)],
        "r_big": [(3.0, 0.5, 9.5, 0.3, 8.5, 1, 0.5, 0.2)],
        "cam_inside": [(1.5, 0.6, 9.5, 0.5, 4.5, 1, 0.5, 0.2)],
        "refl_gt1": [(0.3, 1.5, 9.5, 0.3, 5.5, 1, 1, 1)],
        "refl_neg": [(0.3, -0.5, 9.5, 0.3, 5.5, 1, 1, 1)],
        "col_neg_big": [(0.3, 0.5, 9.5, 0.3, 5.5, -1, 50, 1e30)],
        "r_neg": [(-0.3, 0.5, 9.5, 0.3, 5.5, 1, 1, 1)],
        "coincident": [(0.3, 0.5, 9.5, 0.3, 5.5, 1, 0, 0), (0.3, 0.5, 9.5, 0.3, 5.5, 0, 1, 0)],
        "y_outside": [(0.3, 0.5, 9.5, 5.3, 5.5, 1, 1, 1), (0.3, 0.5, 9.5, -3.0, 5.5, 1, 1, 1)],
        "nan_col": [(0.3, 0.5, 9.5, 0.3, 5.5, np.nan, 1, 1)],
    }
    for name, rows in weird.items():
        for tag, c in (("", spawn), ("_axis", _cam(9.5, 0.5, 4.5))):
            out.append(Scene("sphere_" + name + tag, None, 160, 100, 0.5, c, np.array(rows, sphere_dtype), True))
    # (A degenerate camera that makes every ray NaN is NOT a test input: the reference is
    # built with -ffast-math, i.e. -ffinite-math-only, so its NaN behaviour is whatever
    # the compiler happened to emit and is outside the contract.)
    return out
