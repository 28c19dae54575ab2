#!/usr/bin/env python3
"""Which opcodes the trace kernel's VALU instructions are, dynamically: the ISA as tools/issue_model.py parses it (every instruction
attributed to a source region) weighted by how often a wave enters each region (the counting variant's counters,
profiles/r5_region_counts.json, 4K level.txt scene).  The table that showed 14 % of all VALU instructions to be register copies
(v_mov_b32) the compiler makes at joins -- profiles/r5/sphere_lists_ab.txt.  CPU only.
    python3 tools/op_histogram.py [--moves]          PWN_ISA_EXTRA="-DFOO ..." adds compiler flags (experiment builds)"""
import collections
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import issue_model as im  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    marks = im.region_maps()
    asm = os.path.join(tempfile.mkdtemp(), "k.s")
    im.build_asm(asm)
    ins = im.parse(asm, marks)
    sc = [s for s in json.load(open(os.path.join(ROOT, "profiles", "r5_region_counts.json")))["scenes"]
          if (s["level"], s["w"], s["h"]) == ("pwnfps_level", 3840, 2160)][0]
    cnt = sc["counts"]

    def entries(region):
        key = im.COUNT_OF.get(region)
        if key == "waves":
            return 5120
        if key == "units":
            return 129600
        return cnt.get(key, 0) if key else 0
    w = collections.Counter()
    by = collections.defaultdict(collections.Counter)
    movs, valu = collections.Counter(), collections.Counter()
    for _, reg, cl, op in ins:
        if reg.endswith("~slow") or cl not in ("full", "half", "quarter"):
            continue
        base = reg.split("~")[0]
        n = entries(base)
        w[op] += n
        by[op][base] += n
        valu[base] += 1
        if op.startswith("v_mov_b32"):
            movs[base] += 1
    tot = sum(w.values())
    print("VALU wave-instructions per 4K frame by the model: %.1f M" % (tot / 1e6))
    if "--moves" in sys.argv:
        for r, k in sorted(movs.items(), key=lambda x: -x[1] * entries(x[0]))[:16]:
            print("%-14s %2d v_mov_b32 of %3d VALU per entry x %8d entries = %5.2f M" % (r, k, valu[r], entries(r), k * entries(r) / 1e6))
        return
    for op, n in w.most_common(30):
        top = ", ".join("%s %.1f" % (r, v / 1e6) for r, v in by[op].most_common(4))
        print("%-22s %6.1f M  %4.1f %%   %s" % (op, n / 1e6, 100.0 * n / tot, top))


if __name__ == "__main__":
    main()
