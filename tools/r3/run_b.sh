#!/bin/bash
# round 3, GPU session B: the tiling with moving cuts + two streams under test, then the strip / small-frame figures of the new defaults
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_b; mkdir -p $O
python -m pytest tests/test_gpu_tiled.py tests/test_gpu_frames.py tests/test_gpu_parity.py tests/test_gpu_bench_ranks.py tests/test_c_host.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/tests.txt
tail -15 $O/tests.txt
{
echo "== defaults"; python tools/strip_time.py 8
echo "== 4"; python tools/strip_time.py 4
echo "== 2"; python tools/strip_time.py 2
echo "== 8K 8 strips"; python tools/strip_time.py 8 7680 4320
echo "== synth256 8K 8 strips"; STRIP_LEVEL=synth256 python tools/strip_time.py 8 7680 4320
} > $O/strips.txt 2>&1
PWN_FRAME_OVERLAP=0 bash tools/configs_table.sh > $O/configs_ov0.txt 2>&1
bash tools/configs_table.sh > $O/configs_ov1.txt 2>&1
for te in 8 32 1000000; do
PWN_FRAME_OVERLAP=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-d2h --time-every $te > $O/bench_ov1_te$te.json 2> $O/bench_ov1_te$te.err
PWN_FRAME_OVERLAP=0 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-d2h --time-every $te > $O/bench_ov0_te$te.json 2> $O/bench_ov0_te$te.err
done
grep -h '"value"' $O/bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('kernel_ms'))"
cat $O/configs_ov0.txt $O/configs_ov1.txt
grep -v amdgpu $O/strips.txt | grep "slowest"
