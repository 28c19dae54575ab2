#!/bin/bash
# units numbered by pairs (a tile's two halves on consecutive tickets, the right half's add chain starting from the left half's end) against the build before
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_g2; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_probes.py tests/test_gpu_refill.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
for v in _prev ""; do
  export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip$v.so
  line="lib '$v':"
  for wh in "3840 2160 pwnfps_level" "1280 720 pwnfps_level" "7680 4320 pwnfps_level" "3840 2160 synth256"; do set -- $wh
    r=$(python bench.py --no-cpu-baseline --min-time 1.5 --no-d2h --trace-room 256 --width $1 --height $2 --level $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f %.4f (solo %.4f)' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))")
    line="$line  $1x$2 $r"
  done
  echo "$line"
done; done > $O/ab.txt 2>&1
cat $O/ab.txt
