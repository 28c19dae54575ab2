#!/usr/bin/env python3
"""Randomised failure containment of the row tiling: rank processes on one GPU over the shared-memory transport (tools/tiled_rank.py),
one of them leaves the process in front of a random frame -- random world size, halo mode, host sink, rotating root.  Every other
rank must come back with PWN_ETIMEDOUT (exit status 42) within the wait deadline, say so in pwn_tiled_info (dead = 1), and have
delivered only correct frames until then (the frames it printed are checked against the oracle).  No rank may be left running.
    python3 tools/fuzz_deadlines.py [RUNS [SEED]]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle  # noqa: E402  (checker)
import tiled_rank  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
gold = os.path.join(ROOT, "tests", "golden")
RANK = os.path.join(ROOT, "tools", "tiled_rank.py")
bad = 0
for run in range(runs):
    world = int(rng.integers(2, 6))
    w, h = [(640, 360), (512, 400), (320, 240)][int(rng.integers(0, 3))]
    halo = [-1, -1, 0][int(rng.integers(0, 3))]
    hostsink = bool(rng.integers(0, 2))
    rotate = (not hostsink) and bool(rng.integers(0, 2))
    frames = int(rng.integers(8, 20))
    die_at = int(rng.integers(0, frames))
    who = int(rng.integers(0, world))
    wait_s = 1.5
    env = dict(os.environ, TILED_TIMEOUTS="20,%g" % wait_s, TILED_DIE_AT="%d:%d" % (die_at, who))
    if hostsink:
        env["TILED_HOSTSINK"] = "1"
    if rotate:
        env["TILED_ROTATE"] = "1"
    with tempfile.TemporaryDirectory() as tmp:
        idfile = os.path.join(tmp, "id")
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, RANK, str(r), str(world), idfile, "shm", str(w), str(h), "pwnfps_level", str(frames), str(halo)],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
        outs = []
        hung = False
        for p in procs:
            try:
                o, e = p.communicate(timeout=120)
            except subprocess.TimeoutExpired:
                hung = True
                for q in procs:
                    q.kill()
                o, e = p.communicate()
            outs.append((p.returncode, o, e))
        took = time.time() - t0
    what = "run %d: world %d %dx%d halo %d %s%sframes %d, rank %d leaves at frame %d" % (run, world, w, h, halo, "hostsink " if hostsink else "", "rotate " if rotate else "", frames, who, die_at)
    ok = not hung and outs[who][0] == 17
    # the frames the survivors delivered before the loss are the oracle's
    O = oracle.Oracle()
    O.load_level(os.path.join(gold, "levels", "pwnfps_level.txt"))
    base = np.load(os.path.join(gold, "spheres_t0.npy"))
    _, _, spawn = O.get_level()
    want = {}
    for r, (rc, o, e) in enumerate(outs):
        if r == who:
            continue
        if rc != 42:
            ok = False
            what += " | rank %d exit %s: %s" % (r, rc, e[-300:].replace("\n", " "))
            continue
        m = re.search(r"error (-?\d+) after ([\d.]+) s", o)
        inf = re.search(r"info (\{.*\})", o)
        if not m or int(m.group(1)) != -10 or float(m.group(2)) > wait_s + 6.0 or not inf or json.loads(inf.group(1))["dead"] != 1:
            ok = False
            what += " | rank %d: %s" % (r, (m.group(0) if m else o[-200:]))
        for k, hsh in re.findall(r"(?:frame|seen) (\d+) fnv64 ([0-9a-f]{16})", o):
            k = int(k)
            if k not in want:
                cam, sec, sph = tiled_rank.scene(k, base, spawn)
                O.set_spheres(sph)
                img, _ = O.render(w, h, cam, sec=sec, blur=1)
                want[k] = oracle.fnv64(img)
            if hsh != want[k]:
                ok = False
                what += " | rank %d frame %d differs" % (r, k)
    if hostsink:
        for f in os.listdir("/dev/shm"):
            if f.startswith("pwn_frames_id"):
                try:
                    os.unlink(os.path.join("/dev/shm", f))
                except OSError:
                    pass
    print(("ok   " if ok else "BAD  ") + what + " (%.1f s, %d frames checked)" % (took, len(want)), flush=True)
    bad += 0 if ok else 1
print("fuzz_deadlines: %d runs, %d bad (seed %d)" % (runs, bad, seed))
sys.exit(1 if bad else 0)
