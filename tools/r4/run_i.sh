# round 4, session I: the depth store as scalar base + 32-bit pixel index (base) against the per-lane 64-bit pointer handed into trace_pixel (zptr64)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_i; mkdir -p $O
for rep in 1 2 3; do
for t in base zptr64; do
	if [ "$t" = base ]; then unset PWNHIP_LIB; else export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip_$t.so; fi
	for cfg in "pwnfps_level 3840 2160" "pwnfps_level 1280 720" "synth64 1920 1080"; do
		set -- $cfg
		python3 bench.py --level $1 --width $2 --height $3 --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$t  $1 $2x$3: two streams %.1f Mpix/s %.4f ms/frame | one stream %.4f ms/frame | launch by itself %.4f ms | blur %.4f | hash %s' % (
 d['value'], d['ms_per_step'], d['timing']['roofline_leg']['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms']['blur'], d['frame_fnv64']))"
	done
done
done > $O/zoff_ab.txt 2>&1
unset PWNHIP_LIB
cat $O/zoff_ab.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h --one-stream > $O/bench_one_stream.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_one_stream.csv \;
head -4 $O/kernel_stats_one_stream.csv
