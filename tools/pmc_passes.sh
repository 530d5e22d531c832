#!/bin/bash
# PMC passes over the dominant kernel of a bench workload (one rocprofv3 run per counter group, counters only).
# Usage (on the GPU box, from the repo root): bash tools/pmc_passes.sh <out_dir> [workload=C3] [quick] [bench args...]
# "quick": only the HBM-traffic and issue groups (4 passes instead of 9).
set -e
OUT=$(realpath -m "$1"); shift
WL=${1:-C3}; [ $# -gt 0 ] && shift
QUICK=0; if [ "$1" = "quick" ]; then QUICK=1; shift; fi
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
GROUPS_ALL=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
  "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
  "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS"
  "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU"
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD"
  "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
)
i=0
for group in "${GROUPS_ALL[@]}"; do
  i=$((i+1))
  if [ $QUICK = 1 ] && [ $i -gt 4 ]; then break; fi
  rocprofv3 --kernel-trace --pmc $group -d "$OUT/pass$i" -o p --output-format csv -- python3 "$REPO/bench.py" --workload "$WL" --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; }
  echo "pass $i done"
done
cd "$REPO"
python3 tools/pmc_collect.py "$OUT" "$WL" "$@"
