#!/usr/bin/env python3
"""Generate tests/golden/* from the COMPILED REFERENCE (oracle/_ref, built by
oracle/Makefile from /root/reference).  Runs only in the build container;
what it writes is data (inputs + expected outputs), committed to the repo:

  levels/pwnfps_level.txt   the reference's level data file (input of configs 1,2,4)
  spheres_t0.npy            the 14 spheres game.lua creates at load (t = 0)
  levels/*_tables.npz       parsed level tables + per-cell sphere bins (reference loader)
  frames.json               per-case FNV-64 hashes of pre-blur / post-blur / depth
                            frames and the work counters, every BASELINE config
  raw_320x240.npz           full raw frames (pre, post, z) for the spawn pose
  strips.npz                raw 32-row strips of large frames
  campaign.npz              random scenes (cams, spheres, sec) + hashes
  libm_kat.npz              glibc sinf/cosf/expf known answers
  helpers_kat.npz           col_ftoint / v_normalise / v_dot / randfs / upscale KATs

Usage: python tools/gen_goldens.py [--skip-8k]
"""
import json
import os
import re
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from refharness import RefHarness, SPHERE_DTYPE  # noqa: E402
import oracle as orc  # noqa: E402  (only for its fast FNV routine)

REF = "/root/reference"
G = os.path.join(ROOT, "tests", "golden")
LV = os.path.join(G, "levels")


HOST_CPU = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")


def fnv(a):
    return orc.fnv64(a)


def game_lua_spheres():
    """Parse the numeric table of game.lua:2-23 and apply game.lua:25-30:
    obj_set(r, refl, obx+dx, oby+dy, obz+dz, c1, c2, c3); sums in double (Lua
    numbers), narrowed to float by script.h:22-32."""
    src = open(os.path.join(REF, "game.lua")).read()
    body = src[src.index("opos = {"):src.index("obx, oby, obz")]
    rows = re.findall(r"\{([^{}]+)\}", body)
    m = re.search(r"obx, oby, obz = ([\d.]+), ([\d.]+), ([\d.]+)", src)
    obx, oby, obz = (float(v) for v in m.groups())
    s = np.zeros(len(rows), SPHERE_DTYPE)
    for i, r in enumerate(rows):
        dx, dy, dz, rad, c1, c2, c3, refl = (float(v) for v in r.split(","))
        s[i] = (rad, refl, obx + dx, oby + dy, obz + dz, c1, c2, c3)
    return s


def cam_pose(x, y, z, ay, ax):
    cy, sy = np.float32(np.cos(ay)), np.float32(np.sin(ay))
    cx, sx = np.float32(np.cos(ax)), np.float32(np.sin(ax))
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], np.float32)
    rx = np.array([[1, 0, 0], [0, cx, sx], [0, -sx, cx]], np.float32)
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = (rx @ ry).astype(np.float32)
    m[3, :3] = (x, y, z)
    return m


def main():
    skip_8k = "--skip-8k" in sys.argv
    os.makedirs(LV, exist_ok=True)
    shutil.copyfile(os.path.join(REF, "level.txt"), os.path.join(LV, "pwnfps_level.txt"))
    sph_t0 = game_lua_spheres()
    np.save(os.path.join(G, "spheres_t0.npy"), sph_t0)

    R = RefHarness("tab")
    RC = RefHarness("cnt")
    RH = RefHarness("hw")

    levels = {
        "pwnfps_level": (os.path.join(LV, "pwnfps_level.txt"), sph_t0),
        "synth64": (os.path.join(LV, "synth64.txt"), np.load(os.path.join(LV, "synth64_spheres.npy"))),
        "synth256": (os.path.join(LV, "synth256.txt"), np.load(os.path.join(LV, "synth256_spheres.npy"))),
    }
    spawn = {}
    for name, (path, sph) in levels.items():
        R.load_level(path)
        data, pmap, sp = R.get_level()
        R.set_spheres(sph)
        counts, idx = R.get_bins()
        np.savez_compressed(os.path.join(LV, name + "_tables.npz"), data=data, pmap=pmap, spawn=sp,
                            bin_counts=counts, bin_idx=idx)
        spawn[name] = sp
        print(name, "spawn", sp, "bins", int(counts.sum()))

    cases = []

    def add_case(name, level, sph_key, sph, cam, sec, w, h, counters=True, keep=None, check_hw=True):
        path, _ = levels[level]
        t0 = time.time()
        for H in (R, RC):
            H.load_level(path)
            H.set_spheres(sph)
        pre, z = R.render(w, h, cam, sec=sec, blur=0)
        post, z2 = R.render(w, h, cam, sec=sec, blur=1)
        assert (z.view(np.uint32) == z2.view(np.uint32)).all()
        c = {"name": name, "level": level, "spheres": sph_key, "nspheres": int(len(sph)),
             "cam": [float(v) for v in np.asarray(cam, np.float32).reshape(16)],
             "sec": float(np.float32(sec)), "w": w, "h": h,
             "pre": fnv(pre), "post": fnv(post), "z": fnv(z)}
        if counters:
            RC.reset_counters()
            pre_c, _ = RC.render(w, h, cam, sec=sec, blur=0)
            assert (pre_c == pre).all()
            k = RC.counters()
            c.update(rays=int(k[0]), steps=int(k[1]), portals=int(k[2]), sphere_tests=int(k[3]), exhausted=int(k[4]))
        if check_hw:
            RH.load_level(path)
            RH.set_spheres(sph)
            hw_post, hw_z = RH.render(w, h, cam, sec=sec, blur=1)
            # (every case: the round-3 review's item 2; tools/check_hw_goldens.py does the same on an existing frames.json)
            c["hw_equal"] = bool((hw_post == post).all() and (hw_z.view(np.uint32) == z.view(np.uint32)).all())
            c["hw_host"] = HOST_CPU
        cases.append(c)
        print("%-28s %5dx%-5d pre %s post %s z %s  %.1fs" % (name, w, h, c["pre"], c["post"], c["z"], time.time() - t0),
              flush=True)
        if keep is not None:
            keep(pre, post, z)
        return pre, post, z

    # --- the level.txt scene at every BASELINE resolution ---
    sx, sz = spawn["pwnfps_level"]
    cam0 = cam_pose(sx + 0.5, 0.5, sz + 0.5, 0.0, 0.0)
    raw = {}
    strips = {}
    none = sph_t0[:0]
    sizes = [(320, 200), (320, 240), (1280, 720), (1920, 1080), (3840, 2160)]
    if not skip_8k:
        sizes.append((7680, 4320))
    for (w, h) in sizes:
        keep = None
        if (w, h) == (320, 240):
            keep = lambda p, q, z: raw.update(pre=p, post=q, z=z)
        if (w, h) == (3840, 2160):
            keep = lambda p, q, z: strips.update(c4_rows=np.array([1024, 1056]), c4_pre=p[1024:1056], c4_post=q[1024:1056], c4_z=z[1024:1056])
        add_case("level_spawn_%dx%d" % (w, h), "pwnfps_level", "t0", sph_t0, cam0, 0.0, w, h,
                 counters=(w <= 3840), keep=keep)
        if w <= 1920:
            add_case("level_spawn_nosph_%dx%d" % (w, h), "pwnfps_level", "none", none, cam0, 0.0, w, h)
    # rotated poses + moving time at 320x240 / 720p
    poses = [(0.7, 0.1, 0.0), (2.3, -0.25, 1.5), (4.0, 0.3, 12.25), (5.5, 0.0, 1000.5)]
    for i, (ay, ax, sec) in enumerate(poses):
        cam = cam_pose(sx + 0.5, 0.5, sz + 0.5, ay, ax)
        add_case("level_pose%d_320x240" % i, "pwnfps_level", "t0", sph_t0, cam, sec, 320, 240)
    add_case("level_pose1_1280x720", "pwnfps_level", "t0", sph_t0, cam_pose(sx + 0.5, 0.5, sz + 0.5, 2.3, -0.25), 1.5, 1280, 720)

    # --- synthetic levels ---------------------------------------------------------
    for lvl, big in (("synth64", (1920, 1080)), ("synth256", (7680, 4320))):
        sph = levels[lvl][1]
        cams = np.load(os.path.join(LV, lvl + "_cams.npy"))
        for i, cam in enumerate(cams):
            add_case("%s_cam%d_480x272" % (lvl, i), lvl, lvl, sph, cam, 0.25 * i, 480, 272)
        for i, cam in enumerate(cams):
            if lvl == "synth256":
                if i > 0 or skip_8k:
                    continue
            keep = None
            if lvl == "synth64" and i == 1:
                keep = lambda p, q, z: strips.update(c3_rows=np.array([512, 544]), c3_pre=p[512:544], c3_post=q[512:544], c3_z=z[512:544])
            add_case("%s_cam%d_%dx%d" % (lvl, i, big[0], big[1]), lvl, lvl, sph, cam, 0.25 * i, big[0], big[1],
                     counters=(big[0] <= 3840), keep=keep)
        if lvl == "synth256":
            add_case("synth256_cam0_1920x1080", lvl, lvl, sph, cams[0], 0.0, 1920, 1080)

    np.savez_compressed(os.path.join(G, "raw_320x240.npz"), **raw)
    np.savez_compressed(os.path.join(G, "strips.npz"), **strips)

    # --- random campaign ------------------------------------------------------------
    rng = np.random.default_rng(20141108)
    camp_cams, camp_sec, camp_sph, camp_lvl, camp_hash, camp_size = [], [], [], [], [], []
    sizes = [(256, 128), (200, 152), (320, 96)]
    for lvl, n in (("pwnfps_level", 24), ("synth64", 10), ("synth256", 10)):
        path, _ = levels[lvl]
        R.load_level(path)
        data, _, _ = R.get_level()
        free = [(x, z) for z in range(64) for x in range(64) if chr(data[z, x]) in ';$"#&><,^']
        for it in range(n):
            x, z = free[rng.integers(len(free))]
            hi = chr(data[z, x]) in "#&"
            cam = cam_pose(x + rng.uniform(0.05, 0.95), rng.uniform(0.05, 1.9 if hi else 0.95), z + rng.uniform(0.05, 0.95),
                           rng.uniform(0, 2 * np.pi), rng.uniform(-1.2, 1.2))
            ns = int(rng.integers(0, 32))
            sph = np.zeros(ns, SPHERE_DTYPE)
            for i in range(ns):
                px, pz = free[rng.integers(len(free))] if rng.random() < 0.5 else (x, z)
                sph[i] = (rng.uniform(0.03, 0.45), rng.choice([0.0, 0.2, 0.4, 0.6, 0.9]),
                          min(px + rng.uniform(0.2, 1.2), 62.5), rng.uniform(0.1, 1.5), min(pz + rng.uniform(0.2, 1.2), 62.5),
                          rng.uniform(0, 1.3), rng.uniform(0, 1.3), rng.uniform(0, 1.3))
            sec = float(np.float32(rng.choice([0.0, rng.uniform(0, 10), rng.uniform(0, 3000)])))
            w, h = sizes[it % 3]
            R.set_spheres(sph)
            pre, zb = R.render(w, h, cam, sec=sec, blur=0)
            post, _ = R.render(w, h, cam, sec=sec, blur=1)
            camp_cams.append(cam); camp_sec.append(sec); camp_lvl.append(lvl); camp_size.append((w, h))
            pad = np.zeros(32, SPHERE_DTYPE); pad[:ns] = sph
            camp_sph.append(pad)
            camp_hash.append([fnv(pre), fnv(post), fnv(zb), str(ns)])
    np.savez_compressed(os.path.join(G, "campaign.npz"), cams=np.stack(camp_cams), sec=np.array(camp_sec, np.float32),
                        spheres=np.stack(camp_sph), level=np.array(camp_lvl), size=np.array(camp_size, np.int32),
                        hashes=np.array(camp_hash))
    print("campaign", len(camp_cams), "scenes")

    # --- libm known answers (the glibc the compiled reference links) -----------------
    L = R.lib
    def libm(fn, x):
        f = getattr(L, "pwnref_" + fn)
        return np.array([f(float(v)) for v in x], np.float32)
    n = 24576
    xs_in = rng.uniform(0, 101, n).astype(np.float32)            # pi/2 * pos, pos in [0,64]
    xs_out = rng.uniform(-4 * np.pi, 4 * np.pi, n).astype(np.float32)
    xs_big = (rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(2, 6, n)).astype(np.float32)  # 2*pi*sec_current
    xs_exp = np.concatenate([-rng.uniform(0, 60, n), -rng.uniform(60, 110, n // 4)]).astype(np.float32)
    xs = np.concatenate([xs_in, xs_out, xs_big])
    np.savez_compressed(os.path.join(G, "libm_kat.npz"), x_sincos=xs, sinf=libm("sinf", xs), cosf=libm("cosf", xs),
                        x_exp=xs_exp, expf=libm("expf", xs_exp))

    # --- helper KATs -------------------------------------------------------------------
    v = (rng.standard_normal((4096, 4)) * 10.0 ** rng.uniform(-3, 3, (4096, 1))).astype(np.float32)
    v[:64] = rng.uniform(-1, 2, (64, 4)).astype(np.float32)
    special = np.array([[0, 0, 0, 0], [1, 1, 1, 1], [30, 30, 0, 0], [np.nan, 1, -1, 0.5], [np.inf, -np.inf, 1e10, -1e10],
                        [0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255], [1e-40, -0.0, 255.0, 256.0]], np.float32)
    cv = np.concatenate([special, rng.uniform(-0.2, 1.3, (4096, 4)).astype(np.float32), v[:512]])
    col = np.array([L.pwnref_col_ftoint(np.ascontiguousarray(r).ctypes.data) for r in cv], np.uint32)
    norm = np.zeros_like(v)
    for i in range(len(v)):
        L.pwnref_normalise(np.ascontiguousarray(v[i]).ctypes.data, norm[i].ctypes.data)
    dot = np.array([L.pwnref_dot(np.ascontiguousarray(v[i]).ctypes.data, np.ascontiguousarray(v[(i * 7 + 1) % len(v)]).ctypes.data)
                    for i in range(len(v))], np.float32)
    seeds = rng.integers(0, 2 ** 32, 4096, dtype=np.uint32)
    rfs = np.zeros(4096, np.float32); rseed = np.zeros(4096, np.uint32)
    for i, s in enumerate(seeds):
        a = np.array([s], np.uint32)
        rfs[i] = L.pwnref_randfs(a.ctypes.data); rseed[i] = a[0]
    up_src = rng.integers(0, 2 ** 32, (7, 12), dtype=np.uint32)
    up3 = R.upscale(up_src, 3, pitch_bytes=12 * 3 * 4 + 16)
    up1 = R.upscale(up_src, 1)
    np.savez_compressed(os.path.join(G, "helpers_kat.npz"), col_in=cv, col_out=col, vec=v, norm=norm, dot=dot,
                        seeds=seeds, randfs=rfs, seed_after=rseed, up_src=up_src, up3=up3, up1=up1)

    with open(os.path.join(G, "frames.json"), "w") as f:
        json.dump({"generator": "tools/gen_goldens.py", "reference_flags": "gcc -g -O3 -fopenmp -ffast-math -funroll-loops",
                   "note": "hashes are FNV-1a-64 over uint32 words (SURVEY.md App. B6); depth zero-filled before each frame",
                   "cases": cases}, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
