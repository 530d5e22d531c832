#!/bin/bash
# Two PMC passes (wave/issue cycles, VALU/SALU instruction counts) over the render kernel of any bench workload.
# Usage (on the GPU box, from the repo root): bash tools/pmc_issue.sh <out_dir> [bench args...]
set -e
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for group in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
  "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group -d "$OUT/pass$i" -o p --output-format csv -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline "$@" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; }
done
cd "$REPO"
python3 tools/pmc_collect.py "$OUT"
