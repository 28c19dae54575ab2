#!/usr/bin/env python3
"""Randomised campaign for pwn_init_multi: random levels, cameras, sphere sets, frame sizes and member counts -- every frame of a
group of 2..7 members on device 0 (blocking calls with depth carried over 20 calls while the cuts move; frames in flight delivered
resident, and with the upscaled SDL surface) against the SAME calls on one context, which tools/fuzz_parity.py holds against the oracle.  Run on the GPU box:
    python3 tools/fuzz_group.py [N_SCENES [SEED]]
One line per mismatch, a summary, exit code 1 on any mismatch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwnfps_amd  # noqa: E402
from pwnfps_amd.render import SPHERE_DTYPE  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
gold = os.path.join(ROOT, "tests", "golden", "levels")
fixed = [open(os.path.join(gold, f + ".txt"), "rb").read().decode("latin-1") for f in ("pwnfps_level", "synth64", "synth256")]


def random_level():
    if rng.integers(0, 3) == 0:
        return fixed[rng.integers(0, 3)]
    w, h = int(rng.integers(8, 65)), int(rng.integers(8, 65))
    cells = rng.choice(list(';;;;$$##&&"'), (h, w))
    cells[rng.random((h, w)) < rng.uniform(0.05, 0.3)] = '.'
    cells[rng.random((h, w)) < 0.03] = rng.choice(list('<>,^'))
    letters = [chr(c) for c in range(ord('A'), ord('Z') + 1)]
    rng.shuffle(letters)
    for L in letters[:int(rng.integers(0, 10))]:
        for _ in range(2):
            z, x = int(rng.integers(1, h - 1)), int(rng.integers(1, w - 1))
            cells[z, x] = L
    return "\n".join("".join(r) for r in cells) + "\n"


sizes = [(64, 64), (128, 72), (256, 128), (100, 200), (640, 400), (388, 260), (1280, 720), (36, 96)]
bad = 0
frames_checked = 0
for it in range(n):
    text = random_level()
    w, h = sizes[it % len(sizes)]
    members = int(rng.integers(2, 8))
    one = pwnfps_amd.Renderer(w, h)
    one.level_load_text(text)
    data, _, spawn = one.get_level()
    free = [(x, z) for z in range(64) for x in range(64) if chr(data[z, x]) in ';$"#&']
    if not free:
        one.close()
        continue
    grp = pwnfps_amd.Renderer(w, h, devices=[0] * members)
    grp.level_load_text(text)
    x, z = free[rng.integers(len(free))]
    blur = int(rng.integers(0, 4) != 0)
    one.set_blur_passes(blur)
    grp.set_blur_passes(blur)
    mode = it % 4                 # 0 blocking, 1 delivered in flight, 2 resident in flight, 3 delivered with the upscaled surface
    rscale = int(rng.integers(1, 4))
    pitch = 4 * (w * rscale + int(rng.integers(0, 9)))
    nframes = 20 if mode == 0 else 9
    cams, secs, sphs = [], [], []
    ay = rng.uniform(0, 6.28)
    for f in range(nframes):
        ay += rng.uniform(-0.2, 0.2)
        ax = rng.uniform(-0.6, 0.6)
        cy, sy, cx, sx = np.cos(ay), np.sin(ay), np.cos(ax), np.sin(ax)
        cam = np.eye(4, dtype=np.float32)
        cam[:3, :3] = (np.array([[1, 0, 0], [0, cx, sx], [0, -sx, cx]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])).astype(np.float32)
        cam[3, :3] = (x + rng.uniform(0.1, 0.9), rng.uniform(0.1, 0.9), z + rng.uniform(0.1, 0.9))
        ns = int(rng.integers(0, 24))
        sph = np.zeros(ns, SPHERE_DTYPE)
        for i in range(ns):
            sph[i] = (rng.uniform(0.02, 0.5), rng.choice([0.0, 0.25, 0.6, 1.0]), np.clip(x + rng.uniform(-2, 3), 0.7, 62.3),
                      rng.uniform(0.0, 1.8), np.clip(z + rng.uniform(-2, 3), 0.7, 62.3), *rng.uniform(0, 1.5, 3))
        cams.append(cam); secs.append(float(np.float32(rng.uniform(0, 60)))); sphs.append(sph)

    def frames_of(r):
        out = []
        if mode == 0:
            for f in range(nframes):
                r.set_objects(sphs[f])
                sb, zb = r.trace_screen_centred(cams[f], secs[f])
                out.append((sb.copy(), zb.copy()))
            return out
        if mode == 3:
            r.frames_config(3, sbuf=True, zbuf=False, surface_scale=rscale, pitch_bytes=pitch)
        else:
            r.frames_config(3, sbuf=(mode == 1), zbuf=(mode == 1))
        for f in range(nframes + 3):
            if f >= 3:
                fr = r.wait_frame(f % 3)
                if mode == 3:
                    # (the surface's pixels: the pad columns behind w * rscale belong to the host)
                    out.append((np.concatenate([fr["sbuf"].ravel(), fr["surface"][:, :w * rscale].ravel()]), None))
                else:
                    out.append((fr["sbuf"].copy(), fr["zbuf"].copy()) if mode == 1 else (r.read_plane(fr["d_sbuf"]), None))
            if f < nframes:
                r.set_objects(sphs[f])
                r.submit_frame(cams[f], secs[f], f % 3)
        r.frames_config(0)
        return out
    try:
        a, b = frames_of(one), frames_of(grp)
        gi = grp.group_info()
        for f, ((sa, za), (sb, zb)) in enumerate(zip(a, b)):
            frames_checked += 1
            # colour of every frame; depth of the blocking calls and of the delivered frames (a pixel whose ray ran out of steps keeps the depth
            # of the slot's previous frame / of the previous call, on one context and in a group alike)
            same = (sa == sb).all() and (za is None or (za.view(np.uint32) == zb.view(np.uint32)).all())
            if not same:
                bad += 1
                print("MISMATCH scene %d (seed %d) frame %d: %dx%d, %d members, mode %d, blur %d, %d px differ, cuts %s" % (
                    it, seed, f, w, h, members, mode, blur, int((sa != sb).sum()), gi["cuts"]), flush=True)
                break
    except Exception as e:                                           # noqa: BLE001
        bad += 1
        print("ERROR scene %d (seed %d): %dx%d, %d members, mode %d: %s" % (it, seed, w, h, members, mode, e), flush=True)
    one.close()
    grp.close()
print("fuzz_group: %d scenes, %d frames compared, %d bad (seed %d)" % (n, frames_checked, bad, seed))
sys.exit(1 if bad else 0)
