# round 4, session P: what bounds a rank of an 8-way tiling per frame -- its kernels (0.051 ms, tools/strip_time.py) or its host?  One rank, no exchange,
# a strip-sized frame (3840x272), no timing events: ms per frame of the Python loop (tools/tiled_rank.py) and of the C host (host/pwnhost -W 1)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_p; mkdir -p $O
for rep in 1 2; do
	for size in "3840 272" "3840 2160"; do
		TILED_SAME_SCENE=1 TILED_QUIET=1 python3 tools/tiled_rank.py 0 1 $O/id_$$ shm $size pwnfps_level 4000 -1 2>&1 | grep -E "^host" | sed "s/^/python loop, world 1 $size: /"; rm -f $O/id_$$
		PWN_TILED_SELF=1 TILED_SAME_SCENE=1 TILED_QUIET=1 python3 tools/tiled_rank.py 0 1 $O/id_$$ rccl $size pwnfps_level 4000 -1 2>&1 | grep -E "^host" | sed "s/^/python loop, world 1, self exchange over RCCL $size: /"; rm -f $O/id_$$
	done
	python3 tools/cpu_overhead.py 2>&1 | tail -4
done > $O/host_bound.txt 2>&1
cat $O/host_bound.txt
