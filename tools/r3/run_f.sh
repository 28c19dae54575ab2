#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_f; mkdir -p $O
python tools/r3/overlap_probe.py 2>&1 | grep -v amdgpu.ids > $O/overlap_probe.txt
PWN_FRAME_OVERLAP=0 bash tools/configs_table.sh > $O/configs_ov0.txt 2>&1
PWN_FRAME_OVERLAP=1 bash tools/configs_table.sh > $O/configs_ov1.txt 2>&1
for te in 8 32; do
PWN_FRAME_OVERLAP=1 python bench.py --steps 100 --warmup 20 --min-time 1 --no-cpu-baseline --no-d2h --time-every $te > $O/bench_ov1_te$te.json 2> $O/bench_ov1_te$te.err
PWN_FRAME_OVERLAP=0 python bench.py --steps 100 --warmup 20 --min-time 1 --no-cpu-baseline --no-d2h --time-every $te > $O/bench_ov0_te$te.json 2> $O/bench_ov0_te$te.err
done
PWN_FRAME_OVERLAP=1 python bench.py --steps 100 --warmup 20 --min-time 1 --no-cpu-baseline > $O/bench_ov1_d2h.json 2> $O/bench_ov1_d2h.err
PWN_FRAME_OVERLAP=0 python bench.py --steps 100 --warmup 20 --min-time 1 --no-cpu-baseline > $O/bench_ov0_d2h.json 2> $O/bench_ov0_d2h.err
python -m pytest tests/test_gpu_frames.py tests/test_gpu_tiled.py -x -q -m gpu > $O/tests.txt 2>&1; echo "rc $?" >> $O/tests.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("kernel_ms"), (d.get("d2h_inclusive") or {}).get("value"))
PY
done
cat $O/configs_ov0.txt $O/configs_ov1.txt; tail -3 $O/tests.txt
