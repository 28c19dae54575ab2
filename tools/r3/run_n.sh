#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_n; mkdir -p $O
VALU_RATE_CALIBRATE=1 ./tools/ubench/valu_rate > $O/valu_rate_raw.txt 2>&1
python3 - <<'PY'
import re
out=[]
for line in open("gpurun_out/r3_n/valu_rate_raw.txt"):
    m=re.match(r"^(\S+)\s", line)
    rates=re.findall(r"w(\d): +[\d.]+ \[tick \d+ MHz, launch [\d.]+ ms = ([\d.]+) instr/ns/SIMD\]", line)
    if m and rates:
        out.append("%-14s " % m.group(1) + "  ".join("w%s %s" % (w, r) for w, r in rates))
open("gpurun_out/r3_n/valu_rate.txt","w").write("\n".join(out)+"\n")
print("\n".join(out))
PY
bash tools/variants.sh base 3840 2160 40 | grep -v amdgpu
python bench.py --no-cpu-baseline --min-time 1 --no-d2h | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'])"
