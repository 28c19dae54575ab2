# round 4, session C: per-unit costs + the order simulator; what driving the RCCL communicator non-blocking costs the host (one rank);
# the one-GPU bench line with the new keys
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_c; mkdir -p $O
python -m pytest tests/test_gpu_bench_ranks.py -q -k "headline or one_gpu" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
python3 tools/r4/unit_costs.py $O/unit_costs > $O/unit_order.txt 2>&1
cat $O/unit_order.txt
for mode in blocking nonblocking blocking nonblocking; do
	for size in "3840 2160" "1280 720" "3840 272"; do
		PWN_TILED_RCCL_MODE=$mode TILED_SAME_SCENE=1 python3 tools/tiled_rank.py 0 1 $O/id_$mode rccl $size pwnfps_level 600 -1 2>&1 | grep "^host" | sed "s/^/$mode $size: /"
		rm -f $O/id_$mode
	done
done > $O/rccl_mode_host_cost.txt 2>&1
cat $O/rccl_mode_host_cost.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench.json
