#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_m; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "rc $?" >> $O/tests.txt; tail -3 $O/tests.txt
python tools/fuzz_parity.py 1500 9801 > $O/fuzz_g.txt 2>&1; tail -1 $O/fuzz_g.txt
python tools/fuzz_parity.py 1500 9802 --lattice > $O/fuzz_l.txt 2>&1; tail -1 $O/fuzz_l.txt
PWN_DBG_FORCE_HASW=1 python tools/fuzz_parity.py 1000 9803 --lattice > $O/fuzz_wl.txt 2>&1; tail -1 $O/fuzz_wl.txt
PWN_SCHEDULER=refill python tools/fuzz_parity.py 1000 9804 > $O/fuzz_r.txt 2>&1; tail -1 $O/fuzz_r.txt
python tools/fuzz_parity.py 60 9805 --size 3840x2160 > $O/fuzz_4k.txt 2>&1; tail -1 $O/fuzz_4k.txt
bash tools/final_prof.sh > $O/final_prof.log 2>&1; tail -12 $O/final_prof.log | cut -c1-400
bash tools/configs_table.sh > $O/configs.txt 2>&1; cat $O/configs.txt
STRIP_BALANCE=5 python tools/strip_time.py 8 2>&1 | grep -v amdgpu > $O/strips_8_balanced.txt; grep "slowest\|re-cut" $O/strips_8_balanced.txt
