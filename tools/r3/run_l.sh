#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_l; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "rc $?" >> $O/tests.txt; tail -3 $O/tests.txt
bash tools/fuzz_campaign.sh 9701 > $O/fuzz_campaign.txt 2>&1; cat $O/fuzz_campaign.txt
