# round 4, session L: the colour store behind the read of the next ticket (slate) against in front of it (cur = the code as committed); both = slate + the late QBASE sum
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_l; mkdir -p $O
for rep in 1 2 3; do
for t in cur slate both; do
	export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip_$t.so
	for cfg in "pwnfps_level 3840 2160" "pwnfps_level 1280 720" "synth64 1920 1080"; do
		set -- $cfg
		python3 bench.py --level $1 --width $2 --height $3 --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$t  $1 $2x$3: two streams %.1f Mpix/s %.4f ms/frame | one stream %.4f ms/frame | launch by itself %.4f ms | span %.4f residency %.3f | hash %s' % (
 d['value'], d['ms_per_step'], d['timing']['roofline_leg']['ms_per_step'], d['roofline']['avg_launch_ms'], d['work']['trace_kernel_span_ms'], d['work']['mean_wave_residency'], d['frame_fnv64']))"
	done
done
done > $O/store_ab.txt 2>&1
cat $O/store_ab.txt
