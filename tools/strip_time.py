#!/usr/bin/env python3
"""Time the strip forms on one GPU: trace + bounded blur of a few strips of an N-way
row tiling, i.e. the per-rank kernels of an N-GPU run.
    python3 tools/strip_time.py [N [W H]]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402
from pwnfps_amd.dist import strip_range  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
h = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
gold = os.path.join(ROOT, "tests", "golden")
r = pwnfps_amd.Renderer(w, h)
r.level_load(os.path.join(gold, "levels", "pwnfps_level.txt"))
r.set_objects(np.load(os.path.join(gold, "spheres_t0.npy")))
_, _, spawn = r.get_level()
cam = pwnfps_amd.spawn_camera(spawn)
dev = torch.device("cuda:0")
pre = torch.zeros((h, w), dtype=torch.int32, device=dev)
z = torch.zeros((h, w), dtype=torch.float32, device=dev)
out = torch.zeros((h, w), dtype=torch.int32, device=dev)
miss = torch.zeros(1, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for rank in (range(n) if n <= 16 else sorted({0, n // 2, n - 1})):
    y0, y1 = strip_range(h, n, rank)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tt, tb = [], []
    for it in range(30):
        ev[0].record()
        r.trace_rows_device(cam, 0.0, y0, y1, pre.data_ptr(), z.data_ptr(), s)
        ev[1].record()
        r.blur_rows_device_bounded(y0, y1, pre.data_ptr(), z.data_ptr(), out.data_ptr(), max(y0 - 105, 0), min(y1 + 105, h), miss.data_ptr(), s)
        ev[2].record()
        torch.cuda.synchronize()
        tt.append(ev[0].elapsed_time(ev[1]))
        tb.append(ev[1].elapsed_time(ev[2]))
    print("N=%d rank %d rows [%d,%d): trace min %.4f ms, blur min %.4f ms" % (n, rank, y0, y1, min(tt[5:]), min(tb[5:])))
