# round 5: A/B of blur kernel builds (tags: libpwnhip_<tag>.so, base = libpwnhip.so): parity of the blur tests on every tag, then the bench's numbers
#   gpurun -- bash tools/r5/blur_ab.sh "base bedge"
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/blur_ab; mkdir -p $O
TAGS=${1:-"base"}
for t in $TAGS; do
	if [ "$t" = base ]; then unset PWNHIP_LIB; else export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip_$t.so; fi
	python -m pytest tests/test_gpu_parity.py tests/test_gpu_call_strips.py tests/test_gpu_frames.py tests/test_gpu_tiled.py tests/test_gpu_group.py -q -x > $O/pytest_$t.log 2>&1; echo "$t pytest rc $? $(tail -1 $O/pytest_$t.log)"
	python tools/fuzz_parity.py 1200 12301 > $O/fuzz_$t.log 2>&1; echo "$t $(tail -1 $O/fuzz_$t.log)"
	PWN_CALL_STRIPS=5 python tools/fuzz_parity.py 800 12302 --lattice > $O/fuzz2_$t.log 2>&1; echo "$t $(tail -1 $O/fuzz2_$t.log)"
done
unset PWNHIP_LIB
bash tools/r5/uniform_list_ab.sh "$TAGS" 2>&1 | grep -v pytest | grep -v passed
