#!/bin/bash
# blur tile shapes around 32x32 (and halo 8), judged by the frame rate on two streams
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_x; mkdir -p $O
for rep in 1 2; do
for cfg in "sweep 32 32" "sweep 32 16" "sweep 16 32" "sweep 32 64" "sweep 64 64" "sweep 16 64" "sweeph8 32 32" "sweeph8 64 32" "sweeph8 32 64" "sweeph8 128 16"; do set -- $cfg
  export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip_$1.so PWN_DBG_BLUR_TW=$2 PWN_DBG_BLUR_TH=$3 PWN_DBG_BLUR_BATCH=1
  line="$1 tile $2x$3:"
  for wh in "3840 2160 pwnfps_level" "1280 720 pwnfps_level" "1920 1080 synth64" "7680 4320 synth256"; do set -- $wh
    r=$(python bench.py --no-cpu-baseline --min-time 1 --no-d2h --width $1 --height $2 --level $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f (blur %.4f)' % (d['value'], d['blur_roofline']['avg_launch_ms']))")
    line="$line  $1x$2 $r"
  done
  s=$(python3 tools/strip_time.py 8 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*sum over strips of the 2-stream figure //')
  echo "$line  strips8 $s"
done; done > $O/blur_sweep2.txt 2>&1
cat $O/blur_sweep2.txt
