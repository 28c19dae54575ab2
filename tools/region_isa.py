#!/usr/bin/env python3
"""The instructions of one region of pwn_trace_kernel<false,false> as the issue model attributes them (tools/issue_model.py):
    python3 tools/region_isa.py REGION [REGION ...]      e.g. w_room k_unit"""
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import issue_model as im  # noqa: E402


def main():
    want = set(sys.argv[1:])
    marks = im.region_maps()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "k.s")
        im.build_asm(path)
        ins = im.parse(path, marks)
        # the text of every instruction, in order (parse() keeps order; re-read the lines for the operands)
        texts = []
        on = False
        for line in open(path):
            if line.startswith(im.KERNEL + ":"):
                on = True
                continue
            if not on:
                continue
            if line.startswith(".Lfunc_end"):
                break
            t = line.strip()
            if not t or t.startswith(";") or t.startswith(".") or re.match(r"^(\.LBB\d+_\d+):", line):
                continue
            texts.append(t.split(";")[0].strip())
    assert len(texts) == len(ins), (len(texts), len(ins))
    last = None
    for (label, region, cls, op), t in zip(ins, texts):
        if region in want:
            if label != last:
                print("%s:" % label)
                last = label
            print("   %-8s %-7s %s" % (region, cls, t))


if __name__ == "__main__":
    main()
