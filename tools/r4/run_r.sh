# round 4, session R: a time line of the tiled loop at world 1 on a strip-sized frame (kernels and HIP calls, rocprofv3 traces, no counters)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_r; mkdir -p $O
export TILED_SAME_SCENE=1 TILED_QUIET=1
for form in early late; do
	if [ $form = late ]; then export PWN_TILED_GATHER_LATE=1; else unset PWN_TILED_GATHER_LATE; fi
	rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/$form -o t -- python3 tools/tiled_rank.py 0 1 $O/id_$form shm 3840 272 pwnfps_level 400 -1 > $O/$form.log 2>&1
	tail -2 $O/$form.log
done
find $O -name "*.csv" | xargs ls -la
for f in $(find $O -name "*_kernel_trace.csv" -o -name "*hip_api_trace.csv"); do gzip -f $f; done
find $O -type f | xargs ls -la | tail
