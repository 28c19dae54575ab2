# round 4, session S: frames in flight of a tiled rank, 3 (rounds 2-3) .. 5; one rank, lean loop (tools/tiled_depth.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_s; mkdir -p $O
{
for rep in 1 2; do
	for size in "64 32" "3840 272" "3840 2160"; do
		python3 tools/tiled_depth.py $size 3000 3,4,5 shm 2>&1 | grep -v amdgpu.ids
		PWN_TILED_SELF=1 python3 tools/tiled_depth.py $size 3000 3,4,5 rccl 2>&1 | grep -v amdgpu.ids
	done
	PWN_TILED_GATHER_LATE=1 python3 tools/tiled_depth.py 3840 272 3000 3,4,5 shm 2>&1 | grep -v amdgpu.ids | sed "s/^/gather late: /"
	PWN_TILED_GATHER_LATE=1 PWN_TILED_SELF=1 python3 tools/tiled_depth.py 3840 272 3000 3,4,5 rccl 2>&1 | grep -v amdgpu.ids | sed "s/^/gather late: /"
done
} > $O/depth.txt 2>&1
cat $O/depth.txt
