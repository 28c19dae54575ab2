# round 4, session B: full GPU suite with the bounded tiling + new bench line; A/B of walk-loop variants (two steps per trip, xlt selects as bit selects)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_b; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
export PWN_HASH=1
for rep in 1 2 3; do
for t in base unroll2 xmask xmask2 ux2; do
	if [ "$t" = base ]; then unset PWNHIP_LIB; else export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip_$t.so; fi
	python3 tools/prof_frame.py 3840 2160 30 pwnfps_level 1 2>&1 | tail -1
	python3 tools/prof_frame.py 1920 1080 30 synth64 1 2>&1 | tail -1
	python3 tools/prof_frame.py 7680 4320 8 synth256 1 2>&1 | tail -1
	python3 tools/prof_frame.py 1280 720 30 pwnfps_level 1 2>&1 | tail -1
done
done > $O/variants.txt 2>&1
unset PWNHIP_LIB
cat $O/variants.txt
