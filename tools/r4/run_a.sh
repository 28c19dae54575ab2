# round 4, session A: the 4-lane (HAS_W) variant without its scratch slots: parity (GPU suite) and launch time against the 3-lane variant
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
for hw in 0 1; do
	if [ $hw = 1 ]; then export PWN_DBG_FORCE_HASW=1; else unset PWN_DBG_FORCE_HASW; fi
	rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$hw -o kt -- python3 bench.py --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h --one-stream > $O/bench_hasw$hw.json 2> $O/kt$hw.err
	find $O/kt$hw -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_hasw$hw.csv \;
	head -3 $O/kernel_stats_hasw$hw.csv
done
