# round 4, session J: full-size scenes of the parity campaign on the final build (the scheduler draws tickets in pairs only on long launches),
# the GPU suite once more on the tree as committed, smoke()
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_j; mkdir -p $O
python tools/fuzz_parity.py 300 10301 --size 3840x2160 > $O/full4k.txt 2>&1; tail -1 $O/full4k.txt
python tools/fuzz_parity.py 200 10302 --lattice --size 3840x2160 > $O/full4k_lattice.txt 2>&1; tail -1 $O/full4k_lattice.txt
PWN_DBG_FORCE_HASW=1 python tools/fuzz_parity.py 150 10303 --size 3840x2160 > $O/full4k_hasw.txt 2>&1; tail -1 $O/full4k_hasw.txt
python tools/fuzz_parity.py 40 10304 --size 7680x4320 > $O/full8k.txt 2>&1; tail -1 $O/full8k.txt
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
