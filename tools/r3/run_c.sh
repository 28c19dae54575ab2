#!/bin/bash
# room left beside the persistent trace grid (PWN_DBG_GRID_RESERVE workgroups of ~1280) for the OTHER stream's kernels: frame rate on two streams
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_c; mkdir -p $O
for rep in 1 2; do
for rsv in 0 128 192 256 320 384 512 640; do
  export PWN_DBG_GRID_RESERVE=$rsv
  line="reserve $rsv:"
  for wh in "3840 2160 pwnfps_level" "1280 720 pwnfps_level" "1920 1080 synth64" "7680 4320 synth256"; do set -- $wh
    r=$(python bench.py --no-cpu-baseline --min-time 1 --no-d2h --width $1 --height $2 --level $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f %.4f' % (d['value'], d['ms_per_step']))")
    line="$line  $1x$2 $r"
  done
  s=$(python3 tools/strip_time.py 8 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*2 streams //')
  echo "$line  strips8 $s"
done; done > $O/reserve.txt 2>&1
cat $O/reserve.txt
