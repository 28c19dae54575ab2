#!/bin/bash
# blur with the jitter table in LDS against the build before it: parity, then kernel times (one-stream leg of bench.py) and frame rates
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_s; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_tiled.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
for v in _prev ""; do
  export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip$v.so
  echo "== lib '$v' rep $rep"
  for wh in "3840 2160 pwnfps_level" "1280 720 pwnfps_level" "7680 4320 synth256"; do set -- $wh
    python bench.py --no-cpu-baseline --min-time 1.5 --no-d2h --width $1 --height $2 --level $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1x$2 $3', d['value'], d['ms_per_step'], d['kernel_ms'], d['blur_roofline']['avg_launch_ms'])"
  done
  python3 tools/strip_time.py 8 2>&1 | grep -v amdgpu | tail -1
done; done > $O/ab.txt 2>&1
cat $O/ab.txt
