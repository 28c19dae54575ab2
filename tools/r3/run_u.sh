#!/bin/bash
# blur tile shapes judged by the FRAME rate on two streams (4K), not by the isolated launch: halo 8 / 12 / 16, tile width 64 / 128 / 256, height 16 / 32
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_u; mkdir -p $O
for rep in 1 2; do
for v in "" _w64 _w64h8 _w64h4 _w32 _w32h8 _w32h4; do
  for th in 32 16; do
    if [ "$v" = "_w256" ] && [ $th = 32 ]; then continue; fi
    export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip$v.so PWN_DBG_BLUR_TH=$th
    python bench.py --no-cpu-baseline --min-time 1 --no-d2h 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib \"$v\" th $th', d['value'], d['ms_per_step'], d['timing']['roofline_leg']['ms_per_step'], 'blur launch ms', d['blur_roofline']['avg_launch_ms'], d['parity_vs_reference_golden'])"
  done
done; done > $O/blur_shapes.txt 2>&1
cat $O/blur_shapes.txt
