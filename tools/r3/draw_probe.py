#!/usr/bin/env python3
"""How long does a wave of the trace kernel wait for a ticket it draws when it is done with a unit?  (PWNHIP_LIB = the
-DPWN_DRAW_PROBE build: every draw after the unit, timed with the 100 MHz clock.)   python3 tools/r3/draw_probe.py [N RANK | W H]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
path = "/tmp/pwn_wave_log.bin"
os.environ["PWN_DBG_WAVE_LOG"] = path
import pwnfps_amd  # noqa: E402
from pwnfps_amd.dist import strip_range  # noqa: E402

w, h = 3840, 2160
gold = os.path.join(ROOT, "tests", "golden")
r = pwnfps_amd.Renderer(w, h)
r.level_load(os.path.join(gold, "levels", "pwnfps_level.txt"))
r.set_objects(np.load(os.path.join(gold, "spheres_t0.npy")))
_, _, spawn = r.get_level()
cam = pwnfps_amd.spawn_camera(spawn)
dev = torch.device("cuda:0")
pre = torch.zeros((h, w), dtype=torch.int32, device=dev)
z = torch.zeros((h, w), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for (y0, y1) in (strip_range(h, 8, 4), (0, h)):
    r.set_wave_log(True)
    for _ in range(4):
        r.trace_rows_device(cam, 0.0, y0, y1, pre.data_ptr(), z.data_ptr(), s)
        torch.cuda.synchronize()
    r.stats()
    log = np.fromfile(path, np.uint64).reshape(-1, 2)[1:]
    log = log[log[:, 1] != 0]
    ticks = (log[:, 0] & np.uint64((1 << 40) - 1)).astype(np.float64) / 100.0
    n = (log[:, 0] >> np.uint64(40)).astype(np.float64)
    life = log[:, 1].astype(np.float64) / 100.0
    q = [10, 50, 90, 99]
    print("rows [%d,%d): %d waves, draws per wave %.2f, wait per draw us %s (mean %.2f), waiting share of a wave's life %.3f, life us %s" % (
        y0, y1, len(log), n.mean(), np.round(np.percentile(ticks / np.maximum(n, 1), q), 2).tolist(), (ticks.sum() / n.sum()),
        ticks.sum() / life.sum(), np.round(np.percentile(life, q), 1).tolist()))
