# round 4, session T: time line of the lean tiled loop (tools/tiled_depth.py) on a 64x32 frame, no timing events: where do 170 us per frame go?
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_t; mkdir -p $O
for form in late early; do
	if [ $form = late ]; then export PWN_TILED_GATHER_LATE=1; else unset PWN_TILED_GATHER_LATE; fi
	rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/$form -o t -- python3 tools/tiled_depth.py 64 32 300 3 shm > $O/$form.log 2>&1
	grep "in flight" $O/$form.log
done
for f in $(find $O -name "*_kernel_trace.csv" -o -name "*hip_api_trace.csv"); do gzip -f $f; done
