#!/bin/bash
# blur tile 128 x {32,16} halo 16 (shipped) against 64-wide tiles (halo 16 / 8) on the other workloads, frame rate on two streams
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_v; mkdir -p $O
for rep in 1 2; do
for v in "" _w64 _w64h8; do
  export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip$v.so
  for wh in "3840 2160 pwnfps_level" "1280 720 pwnfps_level" "1920 1080 synth64" "7680 4320 synth256" "3840 2160 synth256"; do set -- $wh
    python bench.py --no-cpu-baseline --min-time 1 --no-d2h --width $1 --height $2 --level $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib \"$v\" $1x$2 $3', d['value'], d['ms_per_step'], 'blur launch ms', d['blur_roofline']['avg_launch_ms'], d['parity_vs_reference_golden'])"
  done
  python3 tools/strip_time.py 8 2>&1 | grep -v amdgpu | tail -1
done; done > $O/blur_w64.txt 2>&1
cat $O/blur_w64.txt
