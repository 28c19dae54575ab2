#!/bin/bash
# round 3, GPU session A: sanity tests, then the short-launch sweeps (late ticket draws, blur tile height, two streams)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_a; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_frames.py tests/test_gpu_tiled.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/tests.txt
tail -3 $O/tests.txt
{
echo "== old behaviour: always draw ahead, 32-row blur tiles"; PWN_DBG_LATE_ROUNDS=0 PWN_DBG_BLUR_TH=32 python tools/strip_time.py 8
echo "== launcher's choice";  python tools/strip_time.py 8
echo "== never draw ahead"; PWN_DBG_LATE_ROUNDS=100000 python tools/strip_time.py 8
echo "== blur 16"; PWN_DBG_BLUR_TH=16 python tools/strip_time.py 8
echo "== blur 8"; PWN_DBG_BLUR_TH=8 python tools/strip_time.py 8
echo "== blur 32"; PWN_DBG_BLUR_TH=32 python tools/strip_time.py 8
echo "== 4 strips"; python tools/strip_time.py 4
echo "== 2 strips"; python tools/strip_time.py 2
echo "== 1 strip"; python tools/strip_time.py 1
echo "== 1 strip old"; PWN_DBG_LATE_ROUNDS=0 python tools/strip_time.py 1
} > $O/strips.txt 2>&1
{
for lr in 0 -1 100000; do
  echo "== wave log strip 4 of 8, late_rounds $lr"; PWN_DBG_LATE_ROUNDS=$lr python tools/strip_wave_log.py 8 4
  echo "== wave log strip 3 of 8, late_rounds $lr"; PWN_DBG_LATE_ROUNDS=$lr python tools/strip_wave_log.py 8 3
done
for lr in 0 -1 40 160; do
echo "== wave log 4K late_rounds $lr"; PWN_DBG_LATE_ROUNDS=$lr python tools/wave_log.py 3840 2160
done
for lr in 0 -1 100000; do
echo "== wave log 720p late_rounds $lr"; PWN_DBG_LATE_ROUNDS=$lr python tools/wave_log.py 1280 720
echo "== wave log 320x240 late_rounds $lr"; PWN_DBG_LATE_ROUNDS=$lr python tools/wave_log.py 320 240
done
} > $O/wavelogs.txt 2>&1
for ov in 0 1; do
  for lr in 0 -1; do
    PWN_FRAME_OVERLAP=$ov PWN_DBG_LATE_ROUNDS=$lr python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-d2h > $O/bench_ov${ov}_lr${lr}.json 2> $O/bench_ov${ov}_lr${lr}.err
  done
done
PWN_FRAME_OVERLAP=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err
PWN_FRAME_OVERLAP=0 PWN_DBG_LATE_ROUNDS=0 PWN_DBG_BLUR_TH=32 bash tools/configs_table.sh > $O/configs_old.txt 2>&1
bash tools/configs_table.sh > $O/configs_new.txt 2>&1
grep -h '"value"' $O/bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('kernel_ms'))"
cat $O/configs_old.txt $O/configs_new.txt
