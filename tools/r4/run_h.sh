# round 4, session H: bench.py with the host-delivered leg in front of the sweep (2 / 3 ranks over shm), pwnhost with a rank that never comes,
# and the frames-in-flight / tiled fuzz legs with PWN_UNIT_ORDER=1 (the option's own campaign)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_h; mkdir -p $O
python -m pytest tests/test_gpu_bench_ranks.py tests/test_c_host.py tests/test_gpu_deadlines.py -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
export PWN_UNIT_ORDER=1
python tools/fuzz_frames.py 200 10209 1280x720 > $O/order_frames.txt 2>&1; tail -1 $O/order_frames.txt
python tools/fuzz_frames.py 60 10210 3840x2160 > $O/order_frames4k.txt 2>&1; tail -1 $O/order_frames4k.txt
python tools/fuzz_tiled.py 10 10211 > $O/order_tiled.txt 2>&1; tail -1 $O/order_tiled.txt
python tools/fuzz_parity.py 1500 10212 > $O/order_generic.txt 2>&1; tail -1 $O/order_generic.txt
