#!/bin/bash
# can a blur workgroup that needs no more than the 32 VGPRs a full trace grid leaves on a SIMD (5 x 96 of 512) run BESIDE that grid?
# shipped build against a build whose 32x32 loop-staging blur is held to 32 VGPRs (-DBLUR_HALVES, amdgpu_num_vgpr(32)); frame rate on two streams
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_y; mkdir -p $O
for rep in 1 2 3; do
for cfg in ": :" "_v32:0:" "_v32:1:"; do
  IFS=: read v b _ <<< "$cfg"
  export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip$v.so
  if [ -n "$b" ]; then export PWN_DBG_BLUR_BATCH=$b; else unset PWN_DBG_BLUR_BATCH; fi
  line="lib '$v' batch '$b':"
  for wh in "3840 2160 pwnfps_level" "7680 4320 synth256"; do set -- $wh
    r=$(python bench.py --no-cpu-baseline --min-time 1.5 --no-d2h --width $1 --height $2 --level $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f %.4f (blur %.4f) %s' % (d['value'], d['ms_per_step'], d['blur_roofline']['avg_launch_ms'], d['parity_vs_reference_golden']))")
    line="$line  $1x$2 $r"
  done
  s=$(python3 tools/strip_time.py 8 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*sum over strips of the 2-stream figure //')
  echo "$line  strips8 $s"
done; done > $O/blur_v32.txt 2>&1
cat $O/blur_v32.txt
