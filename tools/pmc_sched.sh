#!/bin/bash
# one PMC pass (instruction counts + lane activity) per trace scheduler on one scene
#   tools/pmc_sched.sh OUTDIR [prof_frame args]      (PWN_REFILL_LIMIT from the environment)
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in units refill; do
	mkdir -p "$OUT/$s"
	export PWN_SCHEDULER=$s
	rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/$s/q" -o q -- python3 tools/prof_frame.py "$@" > "$OUT/$s/q.log" 2>&1
	echo "== $s"; python3 tools/pmc_summary.py "$OUT/$s" | grep "pwn_trace" | cut -d, -f2-
done
