#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_g; mkdir -p $O
python -m pytest tests/test_gpu_frames.py tests/test_gpu_tiled.py tests/test_gpu_bench_ranks.py tests/test_c_host.py tests/test_gpu_parity.py tests/test_gpu_script.py -x -q -m gpu > $O/tests.txt 2>&1; echo "rc $?" >> $O/tests.txt
tail -25 $O/tests.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
PWN_FRAME_OVERLAP=0 python bench.py --no-cpu-baseline > $O/bench_ov0.json 2> $O/bench_ov0.err
cat $O/bench_default.json
bash tools/configs_table.sh > $O/configs.txt 2>&1; cat $O/configs.txt
