# round 4, session AC: soak of the in-stream tiling over RCCL with one rank, plain / exchanging with itself / a communicator per stream on three streams
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_ac; mkdir -p $O
{
python3 tools/r4/soak.py 30 2>&1 | grep -v amdgpu.ids | sed "s/^/default: /"
PWN_TILED_SELF=1 python3 tools/r4/soak.py 30 2>&1 | grep -v amdgpu.ids | grep tiling | sed "s/^/self exchange: /"
PWN_TILED_SELF=1 PWN_TILED_COMMS=perstream PWN_TILED_STREAMS=3 python3 tools/r4/soak.py 30 2>&1 | grep -v amdgpu.ids | grep tiling | sed "s/^/self exchange, three streams, a communicator each: /"
} | tee $O/soak.txt
