# round 4, session N: the unit loop's two waits chosen per launch (adapt: long launches as committed, short ones without either wait) against the committed code
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_n; mkdir -p $O
for rep in 1 2 3; do
for t in base adapt; do
	if [ "$t" = base ]; then unset PWNHIP_LIB; else export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip_$t.so; fi
	for cfg in "pwnfps_level 3840 2160" "pwnfps_level 1280 720" "synth64 1920 1080" "synth256 3840 2160" "pwnfps_level 320 240"; do
		set -- $cfg
		python3 bench.py --level $1 --width $2 --height $3 --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$t  $1 $2x$3: two streams %.1f Mpix/s %.4f ms/frame | one stream %.4f ms/frame | launch by itself %.4f ms | span %.4f residency %.3f | hash %s' % (
 d['value'], d['ms_per_step'], d['timing']['roofline_leg']['ms_per_step'], d['roofline']['avg_launch_ms'], d['work']['trace_kernel_span_ms'], d['work']['mean_wave_residency'], d['frame_fnv64']))"
	done
done
done > $O/adapt_ab.txt 2>&1
cat $O/adapt_ab.txt
for t in base adapt base adapt; do
	if [ "$t" = base ]; then unset PWNHIP_LIB; else export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip_$t.so; fi
	echo "== $t: strips of an 8-way 4K tiling"
	STRIP_ROOM=256 python3 tools/strip_time.py 8 3840 2160 296,584,840,1088,1352,1624,1896 2>&1 | grep -v amdgpu | tail -1
done > $O/adapt_strips.txt 2>&1
unset PWNHIP_LIB
cat $O/adapt_strips.txt
