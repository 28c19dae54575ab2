# round 4, session AD: full-size scenes of the parity campaign on the tree with the work-queue counters in 2R sets (every trace launch goes through them)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_ad; mkdir -p $O
python tools/fuzz_parity.py 300 10901 --size 3840x2160 > $O/full4k.txt 2>&1; tail -1 $O/full4k.txt
python tools/fuzz_parity.py 200 10902 --lattice --size 3840x2160 > $O/full4k_lattice.txt 2>&1; tail -1 $O/full4k_lattice.txt
PWN_DBG_FORCE_HASW=1 python tools/fuzz_parity.py 150 10903 --size 3840x2160 > $O/full4k_hasw.txt 2>&1; tail -1 $O/full4k_hasw.txt
python tools/fuzz_parity.py 40 10904 --size 7680x4320 > $O/full8k.txt 2>&1; tail -1 $O/full8k.txt
python tools/fuzz_frames.py 120 10905 3840x2160 2>&1 | tail -1
