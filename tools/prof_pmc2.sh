#!/bin/bash
# extra PMC groups (instruction cache, instruction mix); same conventions as prof_pmc.sh
set -u
OUT=${1:-gpurun_out/pmc2}; shift || true
ARGS="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
pass() { local name=$1; shift
	rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 tools/prof_frame.py $ARGS > "$OUT/$name.log" 2>&1
	echo "pass $name rc=$?"; }
pass icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_VSKIPPED SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU
pass mix    SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32
pass mix2   SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_IOPS SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VALU2
