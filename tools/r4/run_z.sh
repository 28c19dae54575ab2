# round 4, session Z: rocprofv3 kernel stats of one rank's tiled loop on a strip-sized frame (RCCL self exchange), in-stream and split
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_z; mkdir -p $O
for choreo in instream split; do
	export PWN_TILED_CHOREO=$choreo
	PWN_TILED_SELF=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$choreo -o kt -- python3 tools/tiled_depth.py 3840 272 2000 3 rccl > $O/$choreo.log 2>&1
	grep "in flight" $O/$choreo.log | sed "s/^/$choreo (under rocprofv3): /" | tee -a $O/loop.txt
	find $O/$choreo -name "*kernel_stats.csv" -exec cp {} $O/tiled_${choreo}_kernel_stats.csv \;
	rm -rf $O/$choreo
done
head -8 $O/tiled_instream_kernel_stats.csv $O/tiled_split_kernel_stats.csv
