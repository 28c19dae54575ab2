# round 4, session Y: frames of an in-stream tiling on two or three compute streams (PWN_TILED_STREAMS), one rank, lean loop; then the tiled tests on three
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_y; mkdir -p $O
{
for rep in 1 2; do
	for ns in 2 3; do
		export PWN_TILED_STREAMS=$ns
		for size in "64 32" "3840 272" "3840 2160"; do
			python3 tools/tiled_depth.py $size 3000 3,4 shm 2>&1 | grep "in flight" | sed "s/^/$ns streams: /"
			PWN_TILED_SELF=1 python3 tools/tiled_depth.py $size 3000 3,4 rccl 2>&1 | grep "in flight" | sed "s/^/$ns streams: /"
		done
	done
done
} > $O/streams.txt 2>&1
cat $O/streams.txt
export PWN_TILED_STREAMS=3
timeout 1500 python3 -m pytest tests -q -m gpu -x -k "tiled or deadlines or bench_ranks or c_host or frames or parity" 2>&1 | tail -5 | tee $O/pytest_three_streams.txt
timeout 600 python3 tools/fuzz_tiled.py 12 10601 2>&1 | tail -3 | tee $O/fuzz_tiled_three_streams.txt
