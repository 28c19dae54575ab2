# round 4, session AE: room for RCCL's kernels beside the persistent trace grid (PWN_TILED_RESERVE workgroups), in-stream tiling, a rank's exchange with itself
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_ae; mkdir -p $O
{
for rep in 1 2; do
	for rsv in 0 16 64 256; do
		for size in "3840 272" "3840 2160"; do
			PWN_TILED_RESERVE=$rsv PWN_TILED_SELF=1 python3 tools/tiled_depth.py $size 3000 3 rccl 2>&1 | grep "in flight" | sed "s/^/reserve $rsv: /"
		done
	done
done
} | tee $O/reserve.txt
