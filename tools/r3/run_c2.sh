#!/bin/bash
# PWN_OPT_TRACE_ROOM: fixed 0, fixed one-per-CU (256), and the library's own choice (-1), frame rate on two streams; what it chose
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_c2; mkdir -p $O
python tools/fuzz_frames.py 60 9971 1280x720 2>&1 | tail -1
for rep in 1 2; do
for room in 0 256 -1; do
  line="room $room:"
  for wh in "3840 2160 pwnfps_level" "1280 720 pwnfps_level" "1920 1080 synth64" "7680 4320 synth256" "3840 2160 synth256" "7680 4320 pwnfps_level" "320 240 pwnfps_level"; do set -- $wh
    r=$(python bench.py --no-cpu-baseline --min-time 2 --no-d2h --trace-room $room --width $1 --height $2 --level $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['config']['trace_room']; print('%.0f %.4f [now %d, %d looks, %d changes]' % (d['value'], d['ms_per_step'], t['room_now'], t['comparisons'], t['changes']))")
    line="$line  $1x$2 $3 $r |"
  done
  echo "$line"
done; done > $O/room.txt 2>&1
cat $O/room.txt
