#!/usr/bin/env python3
"""Wave lifetimes of the trace kernel on ONE strip of an N-way row tiling (what a rank of an
N-GPU run launches per frame).   python3 tools/strip_wave_log.py [N [RANK [W H]]]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = "/tmp/pwn_wave_log.bin"
os.environ["PWN_DBG_WAVE_LOG"] = path
import pwnfps_amd  # noqa: E402
from pwnfps_amd.dist import strip_range  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else n // 2
w = int(sys.argv[3]) if len(sys.argv) > 3 else 3840
h = int(sys.argv[4]) if len(sys.argv) > 4 else 2160
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
y0, y1 = strip_range(h, n, rank)
r.set_wave_log(True)
for _ in range(5):
    r.trace_rows_device(cam, 0.0, y0, y1, pre.data_ptr(), z.data_ptr(), s)
    torch.cuda.synchronize()
st = r.stats()
log = np.fromfile(path, np.uint64).reshape(-1, 2)[1:]
log = log[log[:, 1] != 0]
t0 = log[:, 0].min()
b, e = (log[:, 0] - t0) / 100.0, (log[:, 1] - t0) / 100.0
life = e - b
units = ((w + 15) // 16) * ((y1 - y0 + 3) // 4)
print("strip %d of %d, rows [%d,%d): %d units, %d waves logged, span %.1f us, mean residency %.3f" % (
    rank, n, y0, y1, units, len(log), e.max(), life.sum() / (len(log) * e.max())))
q = [0, 10, 25, 50, 75, 90, 99, 100]
print("wave start us", dict(zip(q, np.round(np.percentile(b, q), 1))))
print("wave end us  ", dict(zip(q, np.round(np.percentile(e, q), 1))))
print("wave life us ", dict(zip(q, np.round(np.percentile(life, q), 1))))
late = np.sort(e)[::-1][:12]
print("last wave ends us", np.round(late, 1).tolist(), "| waves ending after p99 + 3 us:", int((e > np.percentile(e, 99) + 3).sum()))

