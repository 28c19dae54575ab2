# round 5: the sphere lists' loop as the scalar unit's where a whole wave walks ONE list (trace_walk.inc) -- A/B/A/B of builds on one box
#   gpurun -- bash tools/r5/uniform_list_ab.sh "r5a base uni2"      (tags: libpwnhip_<tag>.so, base = libpwnhip.so)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/uniform_ab; mkdir -p $O
TAGS=${1:-"r5a base"}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_call_strips.py tests/test_gpu_probes.py tests/test_gpu_frames.py -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for rep in 1 2 3; do
for t in $TAGS; do
	if [ "$t" = base ]; then unset PWNHIP_LIB; else export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip_$t.so; fi
	for cfg in "pwnfps_level 3840 2160" "pwnfps_level 1280 720" "synth64 1920 1080"; do
		set -- $cfg
		python3 bench.py --level $1 --width $2 --height $3 --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-5s $1 $2x$3: two streams %.1f Mpix/s %.4f ms/frame | one stream %.4f ms/frame | launch by itself %.4f ms | blur %.4f ms | hash %s' % (
 '$t', d['value'], d['ms_per_step'], d['timing']['roofline_leg']['ms_per_step'], d['roofline']['avg_launch_ms'], d['blur_roofline']['avg_launch_ms'], d['frame_fnv64']))"
	done
done
done 2>&1 | tee $O/ab.txt
