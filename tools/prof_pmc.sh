#!/bin/bash
# PMC passes over a few frames of the hot path (tools/prof_frame.py), one
# rocprofv3 run per counter group (SQ: 8 slots; TCC: FETCH_SIZE and WRITE_SIZE
# do not fit together; see /opt/skills/guides/MI355X_MICROARCH.md).
# Counters only: no --kernel-trace/--stats in these runs.
#   tools/prof_pmc.sh OUTDIR [W H [FRAMES [LEVEL [BLUR]]]]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
pass() { # name counters...
	local name=$1; shift
	rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 tools/prof_frame.py $ARGS > "$OUT/$name.log" 2>&1
	echo "pass $name rc=$?"
}
pass valu    SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass wait    SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
pass misc    SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_CBRANCH SQ_INSTS_CBRANCH_TAKEN GRBM_GUI_ACTIVE
pass fetch   FETCH_SIZE
pass write   WRITE_SIZE
pass tcc     TCC_HIT_sum TCC_MISS_sum
find "$OUT" -name "*counter_collection.csv" | head -20
