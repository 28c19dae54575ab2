#!/bin/bash
# A/B of two builds: libpwnhip_prev.so against libpwnhip.so -- parity first, then strips of an 8-way 4K tiling, 4K, 720p, 1080p synth64
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_p; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_probes.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for v in _prev ""; do
  export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip$v.so
  echo "== lib '$v' rep $rep"
  python3 tools/strip_time.py 8 2>&1 | grep -v amdgpu | tail -1
  for wh in "3840 2160 pwnfps_level" "1280 720 pwnfps_level" "320 240 pwnfps_level" "1920 1080 synth64"; do set -- $wh
    python bench.py --no-cpu-baseline --min-time 1 --no-d2h --width $1 --height $2 --level $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1x$2 $3', d['value'], d['ms_per_step'], d['kernel_ms'])"
  done
done; done > $O/ab.txt 2>&1
cat $O/ab.txt
