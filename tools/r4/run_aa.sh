# round 4, session AA: one RCCL communicator against one per compute stream (PWN_TILED_COMMS), a rank's exchange with itself; then the RCCL tests with it
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_aa; mkdir -p $O
{
for rep in 1 2; do
	for comms in one perstream; do
		for ns in 2 3; do
			export PWN_TILED_COMMS=$comms PWN_TILED_STREAMS=$ns
			for size in "64 32" "3840 272" "3840 2160"; do
				PWN_TILED_SELF=1 python3 tools/tiled_depth.py $size 3000 3,4 rccl 2>&1 | grep "in flight" | sed "s/^/comms $comms, $ns streams: /"
			done
		done
	done
done
} > $O/comms.txt 2>&1
cat $O/comms.txt
export PWN_TILED_COMMS=perstream PWN_TILED_STREAMS=2
timeout 900 python3 -m pytest tests/test_gpu_deadlines.py tests/test_gpu_tiled.py -q -x -k "rccl or deadline or timeout or one_rank" 2>&1 | tail -4 | tee $O/pytest_perstream.txt
