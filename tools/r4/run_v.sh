# round 4, session V: the whole GPU suite on the in-stream tiling, the tiled fuzz (choreography and frames in flight at random), a rank bench on shm
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_v; mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 1200 python3 tools/fuzz_tiled.py 24 10501 2>&1 | tail -30 | tee $O/fuzz_tiled_10501.txt
timeout 600 python3 tools/fuzz_deadlines.py 8 10502 2>&1 | tail -12 | tee $O/fuzz_deadlines_10502.txt
