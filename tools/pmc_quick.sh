#!/bin/bash
# one PMC pass (instruction counts + lane activity) for the library in PWNHIP_LIB
#   tools/pmc_quick.sh OUTDIR [prof_frame args]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/q" -o q -- python3 tools/prof_frame.py "$@" > "$OUT/q.log" 2>&1
python3 tools/pmc_summary.py "$OUT" | grep "trace_kernel<false, false" | cut -d, -f3- 
