#!/usr/bin/env python3
"""Randomised parity campaign: HIP path (through the C ABI) vs the oracle on
random levels, cameras, sphere sets, frame sizes and sec_current values.
Not part of the test suite (minutes of oracle time); run on the GPU box:
    python3 tools/fuzz_parity.py [N_SCENES [SEED]]
Prints one line per mismatch and a summary; exit code 1 on any mismatch.

--lattice draws degenerate poses instead of generic ones: axis-aligned or 45-degree
headings, camera and sphere coordinates on the integer / half / quarter lattice.  Rays
then have exactly rational slopes: zero components (the EPSILON clamp, trace.h:220-222),
cell boundaries hit at distance exactly 0, ties between axes (trace.h:156-184), and ramps
whose tilt cancels ray.y so that trace.h:461 divides by zero.  In --ref mode the oracle is
then held against the reference built with -fno-finite-math-only (oracle/_ref/libpwnref_nf.so)
at every pixel, and against the shipped-flags build wherever depth is finite;
--keep-nonfinite DIR saves the scenes that held non-finite pixels together with both
reference renderings (tests/golden/nonfinite/ was made this way: seed 77, 3000 scenes)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402  (checker)

# second mode, for the container where /root/reference is mounted and there is no GPU:
#     python3 tools/fuzz_parity.py N SEED --ref I J K ...
# re-generates the same scenes and compares the ORACLE with the compiled reference on
# scenes I, J, K (all scenes when no index is given)
KEEP = None
if "--keep-nonfinite" in sys.argv:
    i = sys.argv.index("--keep-nonfinite")
    KEEP = sys.argv[i + 1]
    del sys.argv[i:i + 2]
FORCE_SIZE = None                 # --size WxH: every scene at this frame size (e.g. 1600x900: many units per wave)
if "--size" in sys.argv:
    i = sys.argv.index("--size")
    FORCE_SIZE = tuple(int(v) for v in sys.argv[i + 1].lower().split("x"))
    del sys.argv[i:i + 2]
LATTICE = "--lattice" in sys.argv
if LATTICE:
    sys.argv.remove("--lattice")
REF_MODE = "--ref" in sys.argv
ref_idx = [int(a) for a in sys.argv[sys.argv.index("--ref") + 1:]] if REF_MODE else []
argv = sys.argv[:sys.argv.index("--ref")] if REF_MODE else sys.argv
if REF_MODE:
    import tempfile
    import refharness
    R = refharness.RefHarness("tab")
    RN = refharness.RefHarness("nf") if LATTICE else None
else:
    import pwnfps_amd  # noqa: E402

n = int(argv[1]) if len(argv) > 1 else 300
seed = int(argv[2]) if len(argv) > 2 else 1
rng = np.random.default_rng(seed)
gold = os.path.join(ROOT, "tests", "golden", "levels")
fixed = [open(os.path.join(gold, f + ".txt"), "rb").read().decode("latin-1") for f in ("pwnfps_level", "synth64", "synth256")]


def random_level():
    kind = rng.integers(0, 4)
    if kind == 0:
        return fixed[rng.integers(0, 3)]
    # random grid: mostly rooms, some walls, ramps, and letters in the interior
    w, h = int(rng.integers(8, 65)), int(rng.integers(8, 65))
    p_wall = rng.uniform(0.05, 0.35)
    cells = rng.choice(list(';;;;$$##&&"'), (h, w))
    cells[rng.random((h, w)) < p_wall] = '.'
    cells[rng.random((h, w)) < 0.03] = rng.choice(list('<>,^'))
    if kind >= 2:      # closed border
        cells[0, :] = '.'; cells[-1, :] = '.'; cells[:, 0] = '.'; cells[:, -1] = '.'
    letters = [chr(c) for c in range(ord('A'), ord('Z') + 1)]
    rng.shuffle(letters)
    for L in letters[:int(rng.integers(0, 14))]:
        for _ in range(int(rng.integers(1, 4))):       # 1 = unpaired, 2 = pair, 3 = extra sighting
            z, x = int(rng.integers(1, h - 1)), int(rng.integers(1, w - 1))
            cells[z, x] = L if rng.random() < 0.85 else L.lower()
    return "\n".join("".join(r) for r in cells) + "\n"


bad = 0
sizes = [(64, 32), (36, 20), (128, 72), (33, 9), (256, 64), (8, 8), (100, 52), (160, 96)]
# every 16th scene: a frame larger than one blur tile, so that deep pixels send taps outside
# the staged halo (post_kernels.hip) and several tiles / XCD bands take part
big_sizes = [(640, 400), (1024, 136), (388, 260)]
for it in range(n):
    text = random_level()
    O = oracle.Oracle()
    O.load_level_text(text)
    data, _, _ = O.get_level()
    free = [(x, z) for z in range(64) for x in range(64) if chr(data[z, x]) in ';$"#&><,^']
    if not free:
        continue
    x, z = free[rng.integers(len(free))]
    ay, ax = rng.uniform(0, 6.28), rng.uniform(-1.4, 1.4)
    cy, sy, cx, sx = np.cos(ay), np.sin(ay), np.cos(ax), np.sin(ax)
    cam = np.eye(4, dtype=np.float32)
    cam[:3, :3] = (np.array([[1, 0, 0], [0, cx, sx], [0, -sx, cx]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])).astype(np.float32)
    cam[3, :3] = (x + rng.uniform(0.02, 0.98), rng.uniform(0.02, 0.98), z + rng.uniform(0.02, 0.98))
    if LATTICE:
        def lat(lo):
            return lo + float(rng.choice([0.0, 0.5, 0.25, 0.75, 1.0, rng.uniform(0.02, 0.98)]))
        q = np.float32(np.sqrt(0.5))
        hx, hz = [(0, 1), (1, 0), (0, -1), (-1, 0), (q, q), (q, -q), (-q, q), (-q, -q)][int(rng.integers(0, 8))]
        cam = np.zeros((4, 4), np.float32)
        cam[0], cam[1], cam[2] = (hz, 0, -hx, 0), (0, 1, 0, 0), (hx, 0, hz, 0)
        if rng.random() < 0.25:                          # looking straight up / down
            sgn = float(rng.choice([-1.0, 1.0]))
            cam[1], cam[2] = (hx * -sgn, 0, hz * -sgn, 0), (0, sgn, 0, 0)
        cam[3] = (lat(x), float(rng.choice([0.5, 0.25, 0.0, 1.0, 0.75, rng.uniform(0.02, 0.98)])), lat(z), 1)
    if it % 7 == 3 and not LATTICE:
        cam[:, 3] = (rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), rng.uniform(0.5, 1.5))
    if it % 11 == 5:
        cam[3, 1] = rng.uniform(1.02, 1.9)               # upper half of a two-level room
    ns = int(rng.integers(0, 40))
    sph = np.zeros(ns, oracle.SPHERE_DTYPE)
    for i in range(ns):
        sph[i] = (rng.uniform(0.02, 0.6), rng.choice([0.0, 0.25, 0.6, 1.0]), np.clip(x + rng.uniform(-2, 3), 0.7, 62.3),
                  rng.uniform(0.0, 1.8), np.clip(z + rng.uniform(-2, 3), 0.7, 62.3), *rng.uniform(0, 1.5, 3))
    if LATTICE:
        for i in range(ns):
            if rng.random() < 0.7:
                sph[i]["r"] = rng.choice([0.25, 0.5, 0.125])
                sph[i]["x"] = np.clip(np.round(sph[i]["x"] * 4) / 4, 0.75, 62.25)
                sph[i]["y"] = np.round(sph[i]["y"] * 4) / 4
                sph[i]["z"] = np.clip(np.round(sph[i]["z"] * 4) / 4, 0.75, 62.25)
    sec = float(np.float32(rng.choice([0.0, rng.uniform(0, 60), rng.uniform(0, 4000)])))
    if LATTICE and rng.random() < 0.5:
        sec = float(rng.choice([0.0, 0.25, 0.5, 1.0]))
    w, h = sizes[it % len(sizes)]
    if it % 16 == 9:
        w, h = big_sizes[(it // 16) % len(big_sizes)]
    if FORCE_SIZE:
        w, h = FORCE_SIZE
    blur = int(rng.integers(0, 2)) if w % 4 == 0 else 0
    O.set_spheres(sph)
    if REF_MODE:
        if ref_idx and it not in ref_idx:
            continue
        with tempfile.NamedTemporaryFile("wb", suffix=".txt", delete=False) as f:
            f.write(text.encode("latin-1"))
        R.load_level(f.name)
        os.unlink(f.name)
        R.set_spheres(sph)
        a, za = R.render(w, h, cam, sec=sec, blur=blur)
        b, zb = O.render(w, h, cam, sec=sec, blur=blur)
        dz = int((za.view(np.uint32) != zb.view(np.uint32)).sum())
        if LATTICE:
            with tempfile.NamedTemporaryFile("wb", suffix=".txt", delete=False) as f:
                f.write(text.encode("latin-1"))
            RN.load_level(f.name)
            os.unlink(f.name)
            RN.set_spheres(sph)
            c, zc = RN.render(w, h, cam, sec=sec, blur=blur)
            # blur mixes neighbours, so "finite pixel" is only meaningful before it
            fin = np.isfinite(zb) if blur == 0 else np.ones_like(zb, bool) * bool(np.isfinite(zb).all())
            dp = int((c != b).sum()) + int(((a != b) & fin).sum())
            dz += int((zc.view(np.uint32) != zb.view(np.uint32)).sum())
            nonfin = int((~np.isfinite(zb)).sum())
            if nonfin and KEEP:
                # data for tests/golden: the scene and what the two reference builds rendered
                os.makedirs(KEEP, exist_ok=True)
                np.savez_compressed(os.path.join(KEEP, "lattice_%d_%d.npz" % (seed, it)), text=np.array(text), cam=cam, sph=sph,
                                    sec=np.float32(sec), w=w, h=h, blur=blur, ref_nf=c, ref_nf_z=zc, ref_shipped=a)
            if nonfin:
                print("scene %d: %d non-finite depths, shipped-flags build differs at %d pixels" % (it, nonfin, int((a != b).sum())))
        else:
            dp = int((a != b).sum())
        if dp or dz:
            bad += 1
        print("scene %d: %dx%d blur %d sec %r: reference vs oracle: %d pixels, %d depths differ" % (it, w, h, blur, sec, dp, dz))
        continue
    r = pwnfps_amd.Renderer(w, h)
    r.level_load_text(text)
    r.set_objects(sph)
    r.set_blur_passes(blur)
    r.set_counters(True)
    a, za = r.trace_screen_centred(cam, sec)
    st = r.stats()
    b, zb, ost = O.render(w, h, cam, sec=sec, blur=blur, stats=True)
    ok = (a == b).all() and (za.view(np.uint32) == zb.view(np.uint32)).all()
    cnt_ok = (st["rays"], st["steps"], st["portals"], st["sphere_tests"], st["exhausted"]) == (ost.rays, ost.steps, ost.portals, ost.sphere_tests, ost.exhausted)
    if not (ok and cnt_ok):
        bad += 1
        print("MISMATCH scene %d seed %d: %dx%d blur %d sec %r pixels %d depth %d counters %s" % (
            it, seed, w, h, blur, sec, int((a != b).sum()), int((za.view(np.uint32) != zb.view(np.uint32)).sum()), cnt_ok))
        # keep the scene and the pre-blur difference for analysis
        r.set_blur_passes(0)
        a0, z0 = r.trace_screen_centred(cam, sec)
        b0, zb0 = O.render(w, h, cam, sec=sec, blur=0)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_%d_%d.npz" % (seed, it)), text=np.array(text), cam=cam, sph=sph, sec=sec,
                 w=w, h=h, gpu=a0, gpu_z=z0, ora=b0, ora_z=zb0,
                 gpu_cnt=np.array([st["rays"], st["steps"], st["portals"], st["sphere_tests"], st["exhausted"]]),
                 ora_cnt=np.array([ost.rays, ost.steps, ost.portals, ost.sphere_tests, ost.exhausted]))
    r.close()
print("fuzz_parity: %d scenes, %d mismatches (seed %d)" % (n, bad, seed))
sys.exit(1 if bad else 0)
