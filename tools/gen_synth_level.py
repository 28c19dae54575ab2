#!/usr/bin/env python3
"""Synthetic levels for BASELINE.json configs 3 and 5, emitted in the
reference's level.txt format (level.h:107-228) so the compiled reference can
render them too.  Deterministic (fixed seeds); SURVEY.md section 8(d).

  synth64   8x8 rooms of 6x6 cells on a pitch-7 lattice (57x57 cells), room
            type from {; $ # &}, doors between 4-neighbours with p = 0.5,
            26 portal pairs (all 26 letters) in wall cells with one (preferred)
            or two open neighbours, 64 spheres.  seed 0xC0FFEE
  synth256  16x16 rooms of 2x2 cells on a pitch-3 lattice (49x49 cells, offset 2 so
            that no portal touches the grid border: util.h:140-149 is unguarded);
            13 letters link distant rooms; the other 13 are halls of mirrors
            (".A;;A." strips in the spare columns whose endpoints face each
            other with rot12 = 0: rays along a strip loop until maxsteps = 1000
            runs out, trace.h:247,677); 128 small spheres.  seed 0xBADC0DE

Writes tests/golden/levels/<name>.txt, <name>_spheres.npy, <name>_cams.npy.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "levels")
SPHERE_DTYPE = np.dtype([("r", "<f4"), ("refl", "<f4"), ("x", "<f4"), ("y", "<f4"),
                         ("z", "<f4"), ("cb", "<f4"), ("cg", "<f4"), ("cr", "<f4")])
OPEN = set(';$"#&><,^')


def open_neighbours(g, x, z):
    n = []
    for dx, dz in ((1, 0), (0, 1), (-1, 0), (0, -1)):
        if 0 <= x + dx < 64 and 0 <= z + dz < 64 and g[z + dz][x + dx] in OPEN:
            n.append((dx, dz))
    return n


def emit(g, spawn):
    rows = []
    used_z = max(z for z in range(64) if any(c != '.' for c in g[z])) + 2
    used_x = max(x for z in range(64) for x in range(64) if g[z][x] != '.') + 2
    for z in range(min(used_z, 64)):
        row = list(g[z][:min(used_x, 64)])
        if z == spawn[1]:
            row[spawn[0]] = '*'
        rows.append("".join(row))
    return "\n".join(rows) + "\n"


def cam(x, y, z, ay, ax):
    cy, sy = np.float32(np.cos(ay)), np.float32(np.sin(ay))
    cx, sx = np.float32(np.cos(ax)), np.float32(np.sin(ax))
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], np.float32)
    rx = np.array([[1, 0, 0], [0, cx, sx], [0, -sx, cx]], np.float32)
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = (rx @ ry).astype(np.float32)
    m[3, :3] = (x, y, z)
    return m


def synth64():
    rng = np.random.default_rng(0xC0FFEE)
    g = [['.'] * 64 for _ in range(64)]
    types = [[";$#&"[rng.integers(4)] for _ in range(8)] for _ in range(8)]
    high = lambda t: t in "#&"
    for rz in range(8):
        for rx in range(8):
            for z in range(6):
                for x in range(6):
                    g[1 + rz * 7 + z][1 + rx * 7 + x] = types[rz][rx]
    # doors through the 1-cell walls between rooms
    for rz in range(8):
        for rx in range(8):
            for (nx, nz) in ((rx + 1, rz), (rx, rz + 1)):
                if nx >= 8 or nz >= 8 or rng.random() >= 0.5:
                    continue
                a, b = types[rz][rx], types[nz][nx]
                door = '"' if high(a) != high(b) else (a if a in ";#" else (';' if not high(a) else '#'))
                k = int(rng.integers(1, 5))
                if nx > rx:
                    g[1 + rz * 7 + k][1 + rx * 7 + 6] = door
                else:
                    g[1 + rz * 7 + 6][1 + rx * 7 + k] = door
    # a few ramps inside 1-high rooms
    for _ in range(10):
        rz, rx = int(rng.integers(8)), int(rng.integers(8))
        if not high(types[rz][rx]):
            g[1 + rz * 7 + int(rng.integers(1, 5))][1 + rx * 7 + int(rng.integers(1, 5))] = "><,^"[rng.integers(4)]
    # portals: wall cells with exactly one open neighbour, off the border
    # (cells with exactly one open neighbour first, then walls between two rooms:
    #  the loader takes the first open direction in +x,+z,-x,-z order, util.h:140-149)
    one = [(x, z) for z in range(1, 63) for x in range(1, 63)
           if g[z][x] == '.' and len(open_neighbours(g, x, z)) == 1]
    two = [(x, z) for z in range(1, 63) for x in range(1, 63)
           if g[z][x] == '.' and len(open_neighbours(g, x, z)) == 2]
    cand = [one[i] for i in rng.permutation(len(one))] + [two[i] for i in rng.permutation(len(two))]
    order = range(len(cand))
    letters = [chr(ord('A') + i) for i in range(26)]
    placed, li, taken = [], 0, set()
    for i in order:
        if li >= 26:
            break
        x, z = cand[i]
        # keep endpoints apart so that no portal opens onto another portal
        if any(abs(x - px) + abs(z - pz) < 3 for px, pz in taken):
            continue
        placed.append((x, z))
        taken.add((x, z))
        if len(placed) == 2:
            for (px, pz) in placed:
                g[pz][px] = letters[li]
            li += 1
            placed = []
    assert li == 26, li
    spawn = (1 + 3 * 7 + 2, 1 + 3 * 7 + 2)
    sph = np.zeros(64, SPHERE_DTYPE)
    for i in range(64):
        rz, rx = int(rng.integers(8)), int(rng.integers(8))
        r = rng.uniform(0.05, 0.3)
        sph[i] = (r, rng.choice([0.0, 0.2, 0.4, 0.6]), 1 + rx * 7 + rng.uniform(0.6, 5.4),
                  rng.uniform(0.15, 1.7 if high(types[rz][rx]) else 0.8), 1 + rz * 7 + rng.uniform(0.6, 5.4),
                  rng.uniform(0.2, 1.2), rng.uniform(0.2, 1.2), rng.uniform(0.2, 1.2))
    cams = np.stack([
        cam(spawn[0] + 0.5, 0.5, spawn[1] + 0.5, 0.0, 0.0),
        cam(spawn[0] + 0.3, 0.6, spawn[1] + 0.7, 1.1, 0.15),
        cam(1 + 5 * 7 + 2.4, 0.4, 1 + 2 * 7 + 3.2, 2.6, -0.2),
        cam(1 + 1 * 7 + 3.5, 0.7, 1 + 6 * 7 + 1.5, 4.4, 0.3),
    ])
    return emit(g, spawn), sph, cams


def synth256():
    rng = np.random.default_rng(0xBADC0DE)
    g = [['.'] * 64 for _ in range(64)]
    types = [[";;$#"[rng.integers(4)] for _ in range(16)] for _ in range(16)]
    for rz in range(16):
        for rx in range(16):
            for z in range(2):
                for x in range(2):
                    g[2 + rz * 3 + z][2 + rx * 3 + x] = types[rz][rx]
    rooms = [(rx, rz) for rz in range(16) for rx in range(16)]
    perm = rng.permutation(len(rooms))
    link_rooms = [rooms[i] for i in perm[:26]]
    letters = [chr(ord('A') + i) for i in range(26)]
    # 13 halls of mirrors in the spare columns: ".L;;L." strips walled on every
    # other side, so each endpoint's only open neighbour is the strip itself and
    # rot12 = 0: a ray along the strip re-enters it until maxsteps runs out
    hall_type = [";$"[i % 2] for i in range(13)]
    for k in range(13):
        r = 2 + 3 * k
        g[r][53] = letters[k]
        g[r][54] = g[r][55] = hall_type[k]
        g[r][56] = letters[k]
    # links: north (else south) wall of one room to that of another
    for k in range(13):
        for (rx, rz) in (link_rooms[2 * k], link_rooms[2 * k + 1]):
            col = 2 + rx * 3 + int(rng.integers(2))
            if g[2 + rz * 3 - 1][col] == '.':
                g[2 + rz * 3 - 1][col] = letters[13 + k]
            else:
                g[2 + rz * 3 + 2][col] = letters[13 + k]
    # doors (only where the wall cell is still plain)
    for rz in range(16):
        for rx in range(16):
            if rx + 1 < 16 and rng.random() < 0.5:
                x, z = 2 + rx * 3 + 2, 2 + rz * 3 + int(rng.integers(2))
                if g[z][x] == '.':
                    a, b = types[rz][rx], types[rz][rx + 1]
                    g[z][x] = '"' if (a == '#') != (b == '#') else ('#' if a == '#' else ';')
            if rz + 1 < 16 and rng.random() < 0.5:
                x, z = 2 + rx * 3 + int(rng.integers(2)), 2 + rz * 3 + 2
                if g[z][x] == '.':
                    a, b = types[rz][rx], types[rz + 1][rx]
                    g[z][x] = '"' if (a == '#') != (b == '#') else ('#' if a == '#' else ';')
    spawn = (54, 2)
    sph = np.zeros(128, SPHERE_DTYPE)
    for i in range(128):
        if i < 8:   # a few inside the halls
            px, pz = 54 + rng.uniform(0.3, 1.7), 2 + 3 * int(rng.integers(13)) + rng.uniform(0.3, 0.7)
        else:
            rx, rz = rooms[int(rng.integers(len(rooms)))]
            px, pz = 2 + rx * 3 + rng.uniform(0.3, 1.7), 2 + rz * 3 + rng.uniform(0.3, 1.7)
        sph[i] = (rng.uniform(0.03, 0.15), rng.choice([0.0, 0.2, 0.4, 0.6]), px, rng.uniform(0.15, 0.8), pz,
                  rng.uniform(0.2, 1.2), rng.uniform(0.2, 1.2), rng.uniform(0.2, 1.2))
    lr = link_rooms[0]
    cams = np.stack([
        cam(54.6, 0.5, 2.5, -np.pi / 2, 0.0),              # hall 0, straight down the strip (+x)
        cam(2 + lr[0] * 3 + 1.2, 0.45, 2 + lr[1] * 3 + 1.3, 0.4, 0.1),
        cam(55.3, 0.55, 2 + 3 * 3 + 0.4, np.pi / 2 + 0.02, -0.01),   # hall 3 (fog), looking -x
        cam(2 + link_rooms[5][0] * 3 + 1.0, 0.5, 2 + link_rooms[5][1] * 3 + 1.2, 3.3, 0.2),
    ])
    return emit(g, spawn), sph, cams


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, fn in (("synth64", synth64), ("synth256", synth256)):
        text, sph, cams = fn()
        with open(os.path.join(OUT, name + ".txt"), "w", newline="") as f:
            f.write(text)
        np.save(os.path.join(OUT, name + "_spheres.npy"), sph)
        np.save(os.path.join(OUT, name + "_cams.npy"), cams)
        print(name, "rows", text.count("\n"), "spheres", len(sph))
        if "-v" in sys.argv:
            print(text)


if __name__ == "__main__":
    main()
