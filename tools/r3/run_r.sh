#!/bin/bash
# 320x240: the resident loop's two modes (0.024 / 0.037 ms per frame run to run): which leg, which stream setting
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_r; mkdir -p $O
for rep in 1 2 3 4 5 6; do
  for opt in "" "--one-stream"; do
    python bench.py --no-cpu-baseline --min-time 1 --no-d2h --width 320 --height 240 $opt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['timing']; print('320x240 $opt', d['value'], d['ms_per_step'], t['block_ms_p10_p50_p90'], t['roofline_leg']['ms_per_step'], d['kernel_ms'])"
  done
done > $O/small.txt 2>&1
cat $O/small.txt
