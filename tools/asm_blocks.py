#!/usr/bin/env python3
"""Per-basic-block instruction counts of one kernel in hipcc -S output.
    python3 tools/asm_blocks.py file.s mangled_kernel_name"""
import re
import sys

src, name = sys.argv[1], sys.argv[2]
on = False
blocks = []
cur = None
for line in open(src):
    if line.startswith(name + ":"):
        on = True
        cur = ["entry", 0, 0, 0, 0, ""]
        blocks.append(cur)
        continue
    if not on:
        continue
    if line.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):", line) or re.match(r"^; %bb\.(\d+):", line)
    if m:
        cur = [m.group(1), 0, 0, 0, 0, ""]
        mm = re.search(r"Depth=(\d)", line)
        if mm:
            cur[5] = "d" + mm.group(1)
        blocks.append(cur)
        continue
    t = line.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cur[1] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_swappc") or op.startswith("s_setpc"):
        cur[3] += 1
        cur[5] += " " + t.split()[-1] if "branch" in op else " call"
    elif op.startswith("s_"):
        cur[2] += 1
    else:
        cur[4] += 1
print("%-12s %5s %5s %5s %5s" % ("block", "valu", "salu", "br", "mem"))
tv = ts = 0
for b in blocks:
    print("%-12s %5d %5d %5d %5d  %s" % tuple(b))
    tv += b[1]
    ts += b[2]
print("total valu %d salu %d" % (tv, ts))
