# round 4, session U: the in-stream choreography against the split one (PWN_TILED_CHOREO=split), one rank, lean loop
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_u; mkdir -p $O
{
for rep in 1 2; do
	for choreo in instream split; do
		export PWN_TILED_CHOREO=$choreo
		for size in "64 32" "3840 272" "3840 2160"; do
			python3 tools/tiled_depth.py $size 3000 3,5 shm 2>&1 | grep "in flight" | sed "s/^/$choreo: /"
			PWN_TILED_SELF=1 python3 tools/tiled_depth.py $size 3000 3,5 rccl 2>&1 | grep "in flight" | sed "s/^/$choreo: /"
		done
	done
done
} > $O/choreo.txt 2>&1
cat $O/choreo.txt
unset PWN_TILED_CHOREO
timeout 1500 python3 -m pytest tests -q -m gpu -x -k "tiled or deadlines or bench_ranks or c_host" 2>&1 | tail -5 | tee $O/pytest_tiled.txt
