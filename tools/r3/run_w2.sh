#!/bin/bash
# blur tile shapes again, now with PWN_OPT_TRACE_ROOM 256 (room beside the trace grid): frame rate on two streams
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_w2; mkdir -p $O
export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip_sweep.so PWN_TRACE_ROOM=256 STRIP_ROOM=256
for rep in 1 2; do
for cfg in "32 32 1" "64 32 1" "128 16 1" "128 32 1" "128 32 0" "64 16 1" "32 64 1" "64 64 1"; do set -- $cfg
  export PWN_DBG_BLUR_TW=$1 PWN_DBG_BLUR_TH=$2 PWN_DBG_BLUR_BATCH=$3
  line="tile $1x$2 batch $3:"
  for wh in "3840 2160 pwnfps_level" "1280 720 pwnfps_level" "1920 1080 synth64" "7680 4320 pwnfps_level"; do set -- $wh
    r=$(python bench.py --no-cpu-baseline --min-time 1 --no-d2h --width $1 --height $2 --level $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f' % (d['value']))")
    line="$line  $1x$2 $r"
  done
  s=$(python3 tools/strip_time.py 8 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*sum over strips of the 2-stream figure //')
  echo "$line  strips8 $s"
done; done > $O/blur_room.txt 2>&1
cat $O/blur_room.txt
