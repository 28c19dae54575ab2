#!/bin/bash
# blur and sink of every frame on a stream of their own (PWN_DBG_POST_STREAM: 1 = high priority, 2 = default priority), pre-blur planes per slot:
# the next trace on a compute stream no longer waits for the frame's blur.  Frame parity (fuzz_frames) and frame rates.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_e; mkdir -p $O
PWN_DBG_POST_STREAM=1 python tools/fuzz_frames.py 60 9961 1280x720 2>&1 | tail -1
for rep in 1 2; do
for m in "" 1 2; do
  export PWN_DBG_POST_STREAM=$m
  line="post '$m':"
  for wh in "3840 2160 pwnfps_level" "1280 720 pwnfps_level" "1920 1080 synth64" "7680 4320 synth256"; do set -- $wh
    r=$(python bench.py --no-cpu-baseline --min-time 1.5 --width $1 --height $2 --level $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f %.4f d2h %.0f %s' % (d['value'], d['ms_per_step'], d['d2h_inclusive']['value'], d['parity_vs_reference_golden']))")
    line="$line  $1x$2 $r"
  done
  echo "$line"
done; done > $O/post.txt 2>&1
cat $O/post.txt
