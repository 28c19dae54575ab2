# round-end measurement set: bench line, rocprofv3 kernel stats of the same command, PMC passes
#   gpurun -- bash tools/final_prof.sh   -> gpurun_out/final/ (copy into profiles/ as r3_final_*)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
# kernel durations by themselves: the resident loop on ONE compute stream (what roofline.avg_launch_ms is defined on)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -o kt -- python3 bench.py --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h --one-stream > $O/bench_under_rocprof_one_stream.json 2> $O/kt1.err
find $O/kt1 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_one_stream.csv \;
# ... and as the headline leg runs them: frames alternating between two compute streams (durations of kernels that share the chip)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -o kt -- python3 bench.py --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h > $O/bench_under_rocprof.json 2> $O/kt2.err
find $O/kt2 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_two_streams.csv \;
# the general 4-lane variant (cameras with w components), forced through it
export PWN_DBG_FORCE_HASW=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -o kt -- python3 bench.py --steps 50 --warmup 10 --min-time 1 --no-cpu-baseline --no-d2h --one-stream > $O/bench_under_rocprof_hasw.json 2> $O/kt3.err
unset PWN_DBG_FORCE_HASW
find $O/kt3 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_hasw.csv \;
bash tools/prof_pmc.sh $O/pmc 3840 2160 6 pwnfps_level 1 > $O/pmc.log 2>&1
python3 tools/pmc_summary.py $O/pmc "level.txt scene 3840x2160, blur on, round-3 build ${PWN_PROF_TAG:-}" > $O/pmc_summary.csv 2> $O/pmc_summary.err
python3 tools/region_counts.py $O/region_counts.json > $O/region_counts.log 2>&1
tail -3 $O/pmc.log
head -4 $O/kernel_stats_one_stream.csv; head -4 $O/kernel_stats_two_streams.csv; head -3 $O/kernel_stats_hasw.csv
cat $O/bench.json
