#!/bin/bash
# PMC passes (counters only) for one of the one-scene-per-wave shapes. usage: tools/prof_pmc_shapes.sh <outdir> cfg5|yaml
set -u
OUT=$1; WHICH=${2:-cfg5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/prof_target_shapes.py $WHICH > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM
run sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
run sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ SQ_INSTS_BRANCH SQC_ICACHE_BUSY_CYCLES
SMPC_PMC_WORKLOAD="tools/prof_target_shapes.py $WHICH" python3 tools/pmc_summary.py "$OUT" > /dev/null
grep -A45 "solve_kernel" "$OUT/pmc_summary.txt" | grep -E "==|ICACHE|IFETCH|INSTS_VALU |INSTS_SALU|INSTS_LDS|INSTS_SMEM|WAIT|ACTIVE_INST|WAVE_CYCLES|SQ_WAVES"
