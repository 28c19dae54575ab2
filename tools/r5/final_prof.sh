# round 5 measurement set: GPU suite, bench line, rocprofv3 kernel stats of the same command (one stream / two streams / 4-lane variant), PMC passes
#   gpurun -- bash tools/r5/final_prof.sh   -> gpurun_out/r5_final/ (copied into profiles/ as r5_final_*)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_final; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -o kt -- python3 bench.py --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h --one-stream > $O/bench_under_rocprof_one_stream.json 2> $O/kt1.err
find $O/kt1 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_one_stream.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -o kt -- python3 bench.py --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h > $O/bench_under_rocprof_two_streams.json 2> $O/kt2.err
find $O/kt2 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_two_streams.csv \;
export PWN_DBG_FORCE_HASW=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -o kt -- python3 bench.py --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h --one-stream > $O/bench_under_rocprof_hasw.json 2> $O/kt3.err
unset PWN_DBG_FORCE_HASW
find $O/kt3 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_hasw.csv \;
bash tools/prof_pmc.sh $O/pmc 3840 2160 6 pwnfps_level 1 > $O/pmc.log 2>&1
python3 tools/pmc_summary.py $O/pmc "level.txt scene 3840x2160, blur on, round-5 build ${PWN_PROF_TAG:-}" > $O/pmc_summary.csv 2> $O/pmc_summary.err
python3 tools/region_counts.py $O/region_counts.json > $O/region_counts.log 2>&1
rm -rf $O/kt1 $O/kt2 $O/kt3 $O/pmc
tail -3 $O/pmc.log
head -4 $O/kernel_stats_one_stream.csv; head -4 $O/kernel_stats_two_streams.csv; head -3 $O/kernel_stats_hasw.csv
cat $O/bench.json
