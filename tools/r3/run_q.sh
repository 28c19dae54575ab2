#!/bin/bash
# A/B/C on one box: _prev (before the unit-loop diet), _dpp (magic division + 32-bit index, DPP chain), "" (+ the chain under hand-set execution masks)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_q; mkdir -p $O
for rep in 1 2 3; do
for v in _prev _dpp ""; do
  export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip$v.so
  echo "== lib '$v' rep $rep"
  for wh in "3840 2160 pwnfps_level" "1280 720 pwnfps_level" "1920 1080 synth64"; do set -- $wh
    python bench.py --no-cpu-baseline --min-time 2 --no-d2h --width $1 --height $2 --level $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1x$2 $3', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done; done > $O/abc.txt 2>&1
cat $O/abc.txt
