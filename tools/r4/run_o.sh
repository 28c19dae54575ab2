# round 4, session O: RCCL with ONE rank sending itself what a rank of a real tiling sends (PWN_TILED_SELF=1): do RCCL's kernels find room beside the
# persistent trace grids, what do two grouped launches per frame cost the host, blocking against non-blocking communicator.  A 3840x272 frame stands for
# one strip of an 8-way 4K tiling (same kernels, same message sizes: 4.2 MB strip, 2 x 1.7 MB border rows), 3840x2160 for a whole frame
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_o; mkdir -p $O
run() { # label, then env assignments
	local label=$1; shift
	for size in "3840 272" "3840 2160"; do
		env "$@" TILED_SAME_SCENE=1 TILED_QUIET=1 TILED_TIMING=1 TILED_TIMEOUTS=60,30 python3 tools/tiled_rank.py 0 1 $O/id_$$ rccl $size pwnfps_level 2000 -1 2>&1 | grep -E "^host|^frame|^error" | sed "s/^/$label $size: /"
		rm -f $O/id_$$
	done
}
for rep in 1 2; do
run "no exchange          " PWN_TILED_SELF=0
run "self, reserve 0      " PWN_TILED_SELF=1 PWN_TILED_RESERVE=0
run "self, reserve 16     " PWN_TILED_SELF=1 PWN_TILED_RESERVE=16
run "self, reserve 64     " PWN_TILED_SELF=1 PWN_TILED_RESERVE=64
run "self, reserve 16, nb " PWN_TILED_SELF=1 PWN_TILED_RESERVE=16 PWN_TILED_RCCL_MODE=nonblocking
run "self, r16, one stream" PWN_TILED_SELF=1 PWN_TILED_RESERVE=16 PWN_FRAME_OVERLAP=0
done > $O/self_exchange.txt 2>&1
cat $O/self_exchange.txt
