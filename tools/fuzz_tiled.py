#!/usr/bin/env python3
"""Randomised parity of the ROW TILING with moving cuts: rank processes on one GPU over the shared-memory transport
(tools/tiled_rank.py), a frame sequence with changing camera, clock and spheres, cuts re-cut from the launches' own cost words
every second frame AND thrown somewhere else by hand every few frames, random world size / halo mode / host sink; every delivered
frame is the oracle's.      python3 tools/fuzz_tiled.py [RUNS [SEED]]"""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle  # noqa: E402  (checker)
import tiled_rank  # noqa: E402
from pwnfps_amd.dist import default_halo, equal_cuts, max_strip_rows  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
gold = os.path.join(ROOT, "tests", "golden")
RANK = os.path.join(ROOT, "tools", "tiled_rank.py")
bad = 0
frames_done = 0
for run in range(runs):
    world = int(rng.integers(2, 6))                      # (five rank processes and this one: the box allows six on its GPU)
    w, h = [(640, 360), (512, 400), (960, 544), (320, 240)][int(rng.integers(0, 4))]
    level = ["pwnfps_level", "synth64"][int(rng.integers(0, 2))]
    halo = [-1, -1, 0, 1][int(rng.integers(0, 4))]
    hostsink = bool(rng.integers(0, 2))
    frames = int(rng.integers(20, 40))
    eq = equal_cuts(h, world)
    H = default_halo(h) if halo < 0 else halo
    shortest = min(eq[i + 1] - eq[i] for i in range(world))
    lo = H if (0 < H <= shortest) else 8
    hi = max_strip_rows(h, world)
    cuts_at = {}
    for k in range(3, frames, int(rng.integers(3, 8))):
        # random cuts within the constraints (multiples of 8, every strip lo..hi rows); a few tries, else none
        for _ in range(50):
            c = sorted(int(v) * 8 for v in rng.integers(1, h // 8, world - 1))
            c = [0] + c + [h]
            rows = [c[i + 1] - c[i] for i in range(world)]
            if min(rows) >= max(lo, 8) and max(rows) <= hi:
                cuts_at[k] = c
                break
    O = oracle.Oracle()
    O.load_level(os.path.join(gold, "levels", level + ".txt"))
    base = np.load(os.path.join(gold, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy")))
    _, _, spawn = O.get_level()
    want = []
    for k in range(frames):
        cam, sec, sph = tiled_rank.scene(k, base, spawn)
        O.set_spheres(sph)
        want.append(oracle.fnv64(O.render(w, h, cam, sec=sec, blur=1)[0]))
    env = dict(os.environ)
    env["TILED_BALANCE"] = "2"
    if cuts_at:
        env["TILED_CUTS_AT"] = ";".join("%d:%s" % (k, ",".join(str(v) for v in c)) for k, c in cuts_at.items())
    rotate = (not hostsink) and bool(rng.integers(0, 2))      # pwn_tiled_gather_root(ROTATE): frame k comes from rank k mod world
    if hostsink:
        env["TILED_HOSTSINK"] = "1"
    if rotate:
        env["TILED_ROTATE"] = "1"
    split = bool(rng.integers(0, 3) == 0)                     # PWN_OPT_TILED_CHOREO: one run in three with the exchanges on a third stream
    env["PWN_TILED_CHOREO"] = "split" if split else "instream"
    depth = int(rng.integers(1, 6))                           # frames in flight, 1 .. PWN_TILED_SLOTS - 1
    env["TILED_DEPTH"] = str(depth)
    streams = int(rng.integers(2, 4))                         # PWN_OPT_TILED_STREAMS (in-stream only: the split form keeps two)
    env["PWN_TILED_STREAMS"] = str(streams)
    with tempfile.TemporaryDirectory() as td:
        idfile = os.path.join(td, "id_%d" % run)
        procs = [subprocess.Popen([sys.executable, RANK, str(r), str(world), idfile, "shm", str(w), str(h), level, str(frames), str(halo)],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
        outs = []
        ok = True
        for p in procs:
            try:
                o, e = p.communicate(timeout=900)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                o, e, ok = "", "timeout", False
            if p.returncode != 0:
                ok = False
                print("run %d: a rank failed: %s" % (run, e[-500:]))
            outs.append(o)
    if rotate:
        by = {}
        for r, o in enumerate(outs):
            for k, hh in re.findall(r"frame (\d+) fnv64 ([0-9a-f]{16})", o):
                by[int(k)] = hh if int(k) % world == r else "from the wrong rank"
        got = [by.get(k, "") for k in range(frames)]
    else:
        got = [hh for _, hh in re.findall(r"frame (\d+) fnv64 ([0-9a-f]{16})", outs[0])]
    mism = [k for k in range(frames) if k >= len(got) or got[k] != want[k]]
    seen_bad = 0
    if hostsink:
        for o in outs[1:]:
            sn = [hh for _, hh in re.findall(r"seen (\d+) fnv64 ([0-9a-f]{16})", o)]
            seen_bad += sum(1 for k in range(frames) if k >= len(sn) or sn[k] != want[k])
    info = json.loads(re.search(r"info (\{.*\})", outs[0]).group(1)) if ok and "info" in outs[0] else {}
    frames_done += frames
    if not ok or mism or seen_bad:
        bad += 1
        print("MISMATCH run %d: world %d %dx%d %s halo %d sink %d %s: frames %s, other ranks' views %d" % (run, world, w, h, level, halo, hostsink, "split" if split else "in-stream", mism[:8], seen_bad))
    else:
        print("run %d ok: world %d %dx%d %s halo %d sink %d rotating root %d %s, %d frames, %d cuts by hand, the cuts moved %s times, %s frames repeated" % (
            run, world, w, h, level, halo, hostsink, int(rotate), ("split" if split else "in-stream on %d streams" % streams) + ", %d in flight" % depth, frames, len(cuts_at), info.get("recuts"), info.get("frames_redone")), flush=True)
print("fuzz_tiled: %d runs, %d frames, %d bad runs (seed %d)" % (runs, frames_done, bad, seed))
sys.exit(1 if bad else 0)
