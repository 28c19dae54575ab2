# round 4, session AG: the headline leg with the trace grid's room fixed at other numbers than 0 and one workgroup per CU (256)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_ag; mkdir -p $O
for rep in 1 2; do
	for room in -1 0 64 128 192 256 320 384 512; do
		python3 bench.py --trace-room $room --no-cpu-baseline --no-d2h --min-time 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('room $room:', d['value'], d['ms_per_step'], d['config']['trace_room']['room_now'])"
	done
done | tee $O/room.txt
