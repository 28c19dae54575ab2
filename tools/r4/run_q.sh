# round 4, session Q: the host-bound tiled path (session P) with the words brought by a kernel and the gather leaving one submit earlier;
# the round-3 forms behind PWN_TILED_WORDS_COPY=1 / PWN_TILED_GATHER_LATE=1 for the A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_q; mkdir -p $O
run() {   # $1 = label, rest = env assignments
	label=$1; shift
	for size in "3840 272" "3840 2160"; do
		env "$@" TILED_SAME_SCENE=1 TILED_QUIET=1 python3 tools/tiled_rank.py 0 1 $O/id_$$ shm $size pwnfps_level 4000 -1 2>&1 | grep -E "^host" | sed "s/^/$label | world 1, no exchange, $size: /"; rm -f $O/id_$$
		env "$@" PWN_TILED_SELF=1 TILED_SAME_SCENE=1 TILED_QUIET=1 python3 tools/tiled_rank.py 0 1 $O/id_$$ rccl $size pwnfps_level 4000 -1 2>&1 | grep -E "^host" | sed "s/^/$label | world 1, self exchange over RCCL, $size: /"; rm -f $O/id_$$
	done
	env "$@" python3 tools/cpu_overhead.py 2>&1 | tail -3 | sed "s/^/$label | /"
}
{
for rep in 1 2; do
	run "new (words by kernel, gather early)" PWN_X=0
	run "words by copy, gather early" PWN_TILED_WORDS_COPY=1
	run "words by kernel, gather late" PWN_TILED_GATHER_LATE=1
	run "round-3 form (copy, late)" PWN_TILED_WORDS_COPY=1 PWN_TILED_GATHER_LATE=1
done
} > $O/host_bound_ab.txt 2>&1
cat $O/host_bound_ab.txt
timeout 1500 python3 -m pytest tests -q -m gpu -x -k "tiled or deadlines or bench_ranks or c_host" 2>&1 | tail -5 | tee $O/pytest_tiled.txt
