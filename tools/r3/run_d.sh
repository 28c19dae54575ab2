#!/bin/bash
# blur staging straight into LDS (global_load_lds_dwordx4, batch 2) against staging through registers (batch 1): parity, then frame rates on two streams
# (needs the sweep build: make -C pwnfps_amd/csrc VARIANT=sweep EXTRA=-DPWN_BLUR_SWEEP)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_d; mkdir -p $O
PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip_sweep.so PWN_DBG_BLUR_BATCH=2 python -m pytest tests/test_gpu_parity.py -x -q -k "blur or golden or random or ragged or extreme" 2>&1 | tail -3
for rep in 1 2; do
for b in 1 2; do
  export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip_sweep.so PWN_DBG_BLUR_BATCH=$b
  line="batch $b:"
  for wh in "3840 2160 pwnfps_level" "1280 720 pwnfps_level" "1920 1080 synth64" "7680 4320 synth256"; do set -- $wh
    r=$(python bench.py --no-cpu-baseline --min-time 1 --no-d2h --width $1 --height $2 --level $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f (blur %.4f) %s' % (d['value'], d['blur_roofline']['avg_launch_ms'], d['parity_vs_reference_golden']))")
    line="$line  $1x$2 $r"
  done
  s=$(python3 tools/strip_time.py 8 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*sum over strips of the 2-stream figure //')
  echo "$line  strips8 $s"
done; done > $O/direct.txt 2>&1
cat $O/direct.txt
