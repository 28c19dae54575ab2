#!/usr/bin/env python3
"""Time the strip forms on one GPU: trace + bounded blur of the strips of an N-way row tiling, i.e. the
per-rank kernels of an N-GPU run.  Three figures per strip:
  isolated   one launch at a time between HIP events (min of 25): what a frame's LATENCY is made of
  pipelined  200 frames (trace, blur, trace, blur ...) back to back on ONE stream, per frame
  2 streams  the same with the frames alternating between two streams, planes by parity (what
             pwn_tiled_submit does): the per-frame kernel time that bounds a rank's THROUGHPUT
    python3 tools/strip_time.py [N [W H [cuts]]]        cuts: comma-separated row boundaries (N-1 of them)
                                                        instead of equal strips"""
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
cuts = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else None
level = os.environ.get("STRIP_LEVEL", "pwnfps_level")
gold = os.path.join(ROOT, "tests", "golden")
r = pwnfps_amd.Renderer(w, h)
r.level_load(os.path.join(gold, "levels", level + ".txt"))
r.set_objects(np.load(os.path.join(gold, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy"))))
_, _, spawn = r.get_level()
if os.environ.get("STRIP_ROOM") is not None:          # PWN_OPT_TRACE_ROOM: what the tiling's trace launches leave free on two streams
    r.set_trace_room(int(os.environ["STRIP_ROOM"]))
cam = pwnfps_amd.spawn_camera(spawn)
if level != "pwnfps_level":
    cam = np.load(os.path.join(gold, "levels", level + "_cams.npy"))[0]
dev = torch.device("cuda:0")
pre = [torch.zeros((h, w), dtype=torch.int32, device=dev) for _ in range(2)]
z = [torch.zeros((h, w), dtype=torch.float32, device=dev) for _ in range(2)]
out = [torch.zeros((h, w), dtype=torch.int32, device=dev) for _ in range(2)]
miss = torch.zeros(64, dtype=torch.int32, device=dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
H = int(0.002 * h * 24.0) + 2

if cuts is None:
    ranges = [strip_range(h, n, k) for k in range(n)]
else:
    edges = [0] + cuts + [h]
    ranges = [(edges[k], edges[k + 1]) for k in range(len(edges) - 1)]


def frame(y0, y1, p, s):
    r.trace_rows_device(cam, 0.0, y0, y1, pre[p].data_ptr(), z[p].data_ptr(), s.cuda_stream)
    r.blur_rows_device_bounded(y0, y1, pre[p].data_ptr(), z[p].data_ptr(), out[p].data_ptr(), max(y0 - H, 0), min(y1 + H, h),
                               miss.data_ptr() + 128 * p, s.cuda_stream)


def pipelined(y0, y1, two, frames=200):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[0])
        streams[1].wait_event(e0)
        for f in range(frames):
            p = f & 1 if two else 0
            frame(y0, y1, p, streams[p])
        if two:
            j = torch.cuda.Event()
            j.record(streams[1])
            streams[0].wait_event(j)
        e1.record(streams[0])
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / frames)
    return best


def measure(ranges, quiet=False):
    rows = []
    for rank, (y0, y1) in enumerate(ranges):
        if n > 16 and rank not in (0, n // 2, n - 1):
            continue
        s = streams[0]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tt, tb = [], []
        for it in range(30):
            ev[0].record(s)
            r.trace_rows_device(cam, 0.0, y0, y1, pre[0].data_ptr(), z[0].data_ptr(), s.cuda_stream)
            ev[1].record(s)
            r.blur_rows_device_bounded(y0, y1, pre[0].data_ptr(), z[0].data_ptr(), out[0].data_ptr(), max(y0 - H, 0), min(y1 + H, h), miss.data_ptr(), s.cuda_stream)
            ev[2].record(s)
            torch.cuda.synchronize()
            tt.append(ev[0].elapsed_time(ev[1]))
            tb.append(ev[1].elapsed_time(ev[2]))
        p1, p2 = pipelined(y0, y1, False), pipelined(y0, y1, True)
        rows.append((min(tt[5:]), min(tb[5:]), p1, p2))
        if not quiet:
            print("N=%d rank %d rows [%d,%d): isolated trace %.4f + blur %.4f = %.4f ms | pipelined %.4f ms/frame | 2 streams %.4f ms/frame" % (
                len(ranges), rank, y0, y1, rows[-1][0], rows[-1][1], rows[-1][0] + rows[-1][1], p1, p2), flush=True)
    a = np.array(rows)
    iso = a[:, 0] + a[:, 1]
    print("N=%d %dx%d %s cuts %s: slowest strip isolated %.4f ms (max/mean %.3f) | pipelined %.4f (%.3f) | 2 streams %.4f (%.3f) | sum over strips of the 2-stream figure %.4f ms" % (
        len(ranges), w, h, level, [y0 for y0, _ in ranges[1:]], iso.max(), iso.max() / iso.mean(), a[:, 2].max(), a[:, 2].max() / a[:, 2].mean(),
        a[:, 3].max(), a[:, 3].max() / a[:, 3].mean(), a[:, 3].sum()), flush=True)
    return a


a = measure(ranges)
# STRIP_BALANCE=k: k rounds of the library's own re-cut rule (pwn_tiled_recut) with each strip's 2-stream time as its cost --
# what the moving cuts of pwn_tiled_balance converge to (there the cost is the sum of the trace waves' lifetimes)
rounds = int(os.environ.get("STRIP_BALANCE", "0"))
if rounds and len(ranges) > 1 and len(ranges) <= 16:
    from pwnfps_amd import _lib
    from pwnfps_amd.dist import max_strip_rows
    cuts_now = [y0 for y0, _ in ranges] + [h]
    for k in range(rounds):
        cost = np.array([max(1, int(v * 1e6)) for v in a[:, 3]], np.uint32)
        cin, cout = np.array(cuts_now, np.int32), np.zeros(len(cuts_now), np.int32)
        moved = _lib.lib.pwn_tiled_recut(cin.ctypes.data, cost.ctypes.data, len(ranges), h, H, max_strip_rows(h, len(ranges)), cout.ctypes.data)
        if moved != 1:
            print("re-cut %d: the cuts stay (within 2 %% of each other, or a constraint)" % (k + 1))
            break
        cuts_now = [int(v) for v in cout]
        a = measure([(cuts_now[i], cuts_now[i + 1]) for i in range(len(ranges))], quiet=(k + 1 < rounds))
