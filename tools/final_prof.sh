# round-end measurement set: bench line, rocprofv3 kernel stats of the same command, PMC passes
#   gpurun -- bash tools/final_prof.sh   -> gpurun_out/final/ (copy into profiles/)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
python bench.py --steps 200 --warmup 20 > gpurun_out/final/bench_nopmc.json 2> gpurun_out/final/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt -o kt -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-d2h > gpurun_out/final/bench_prof.json 2> gpurun_out/final/kt.err
find gpurun_out/final/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/kernel_stats.csv \;
bash tools/prof_pmc.sh gpurun_out/final/pmc 3840 2160 6 pwnfps_level 1 > gpurun_out/final/pmc.log 2>&1
python3 tools/pmc_summary.py gpurun_out/final/pmc "level.txt scene 3840x2160, blur on, round-2 build ${PWN_PROF_TAG:-}" > gpurun_out/final/pmc_summary.csv 2> gpurun_out/final/pmc_summary.err
tail -3 gpurun_out/final/pmc.log
head -3 gpurun_out/final/kernel_stats.csv
