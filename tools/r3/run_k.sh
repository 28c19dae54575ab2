#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_k; mkdir -p $O
{
for lr in 0 20 40 80 100000; do
  echo "== late_rounds $lr"
  PWN_DBG_LATE_ROUNDS=$lr python tools/strip_wave_log.py 8 4 2>&1 | grep -E "strip|wave end"
  PWN_DBG_LATE_ROUNDS=$lr python tools/strip_time.py 8 2>&1 | grep slowest
  PWN_DBG_LATE_ROUNDS=$lr python tools/wave_log.py 1280 720 2>&1 | grep -E "span|residency"
  PWN_DBG_LATE_ROUNDS=$lr bash tools/variants.sh base 3840 2160 30 | grep -v amdgpu
done
} > $O/late_draws_before_store.txt 2>&1
cat $O/late_draws_before_store.txt
