# round 4, session X: time line of the in-stream tiling on a strip-sized frame with RCCL carrying a rank's bytes to itself
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_x; mkdir -p $O
PWN_TILED_SELF=1 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/self -o t -- python3 tools/tiled_depth.py 3840 272 300 3 rccl > $O/self.log 2>&1
grep "in flight" $O/self.log
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/shm -o t -- python3 tools/tiled_depth.py 3840 272 300 3 shm > $O/shm.log 2>&1
grep "in flight" $O/shm.log
for f in $(find $O -name "*_kernel_trace.csv" -o -name "*hip_api_trace.csv"); do gzip -f $f; done
