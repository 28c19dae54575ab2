# round 4, session D: PWN_OPT_UNIT_ORDER (units handed out by last launch's cost) -- parity (new test + frames / tiled suites), then A/B
# against the arithmetic order: isolated launches, one stream, two streams; five scenes; the strips of an 8-way 4K tiling
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_d; mkdir -p $O
python -m pytest tests/test_gpu_order.py tests/test_gpu_frames.py tests/test_gpu_parity.py tests/test_gpu_tiled.py -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2; do
for ord in 0 1; do
	export PWN_UNIT_ORDER=$ord
	for cfg in "pwnfps_level 3840 2160" "pwnfps_level 1280 720" "synth64 1920 1080" "synth256 3840 2160" "pwnfps_level 7680 4320"; do
		set -- $cfg
		python3 bench.py --level $1 --width $2 --height $3 --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('order $ord  $1 $2x$3: two streams %.1f Mpix/s %.4f ms/frame | one stream %.4f ms/frame | launch by itself %.4f ms | blocking frame trace %.4f blur %.4f | span %.4f residency %.3f | hash %s' % (
 d['value'], d['ms_per_step'], d['timing']['roofline_leg']['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms']['trace'], d['kernel_ms']['blur'], d['work']['trace_kernel_span_ms'], d['work']['mean_wave_residency'], d['frame_fnv64']))"
	done
done
done > $O/order_ab.txt 2>&1
cat $O/order_ab.txt
for ord in 0 1 0 1; do
	echo "== PWN_UNIT_ORDER=$ord: strips of an 8-way 4K tiling (tools/strip_time.py, cuts 296,584,840,1088,1352,1624,1896)"
	PWN_UNIT_ORDER=$ord STRIP_ROOM=256 python3 tools/strip_time.py 8 3840 2160 296,584,840,1088,1352,1624,1896 2>&1 | tail -14
done > $O/strips_ab.txt 2>&1
cat $O/strips_ab.txt
