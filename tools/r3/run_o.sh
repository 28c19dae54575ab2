#!/bin/bash
# static first tickets (-DPWN_STATIC_FIRST) against the shipped build: strips of an 8-way 4K tiling, 4K, 720p, 320x240
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_o; mkdir -p $O
for rep in 1 2; do
for v in "" _static; do
  export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip$v.so
  echo "== lib '$v' rep $rep"
  python3 tools/strip_time.py 8 2>&1 | grep -v amdgpu | tail -14
  for wh in "3840 2160" "1280 720" "320 240"; do set -- $wh
    python bench.py --no-cpu-baseline --min-time 1 --no-d2h --width $1 --height $2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1x$2', d['value'], d['ms_per_step'], d['kernel_ms'])"
  done
done; done > $O/static_first.txt 2>&1
cat $O/static_first.txt
