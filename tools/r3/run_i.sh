#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_i; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "rc $?" >> $O/tests.txt
tail -5 $O/tests.txt
bash tools/prof_pmc.sh $O/pmc 3840 2160 6 pwnfps_level 1 > $O/pmc.log 2>&1
python3 tools/pmc_summary.py $O/pmc "level.txt scene 3840x2160, blur on, round-3 build (lstep)" > $O/pmc_summary.csv 2> $O/pmc_summary.err
cp $O/pmc_summary.csv profiles/pmc_latest.csv
python tools/region_counts.py $O/region_counts.json > $O/region_counts.log 2>&1; tail -2 $O/region_counts.log | cut -c1-300
grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_BRANCH|SQ_INSTS_LDS|WRITE_SIZE|FETCH_SIZE" $O/pmc_summary.csv | grep -v "<true" | head -20
