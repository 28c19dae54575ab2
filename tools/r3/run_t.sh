#!/bin/bash
# what the 4K blur launch is made of: the shipped kernel, without its staging loads (exp1: taps read whatever is in LDS), without its taps (exp2: stage, then copy)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_t; mkdir -p $O
for rep in 1 2; do
for v in "" _exp1 _exp2; do
  export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip$v.so
  python bench.py --no-cpu-baseline --min-time 1 --no-d2h 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib \"$v\"', d['value'], d['ms_per_step'], d['timing']['roofline_leg']['ms_per_step'], 'blur launch ms', d['blur_roofline']['avg_launch_ms'], d['parity_vs_reference_golden'])"
done; done > $O/blur_parts.txt 2>&1
cat $O/blur_parts.txt
