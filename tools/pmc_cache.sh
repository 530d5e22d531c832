#!/bin/bash
# L2 / vector-L1 hit counters over the render kernel of any bench workload.
# Usage (on the GPU box, from the repo root): bash tools/pmc_cache.sh <out_dir> [workload=C5] [bench args...]
set -e
OUT=$(realpath -m "$1"); shift
WL=${1:-C5}; [ $# -gt 0 ] && shift
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for group in \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group -d "$OUT/pass$i" -o p --output-format csv -- python3 "$REPO/bench.py" --workload "$WL" --steps 1 --warmup 1 --no-cpu-baseline "$@" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; }
done
cd "$REPO"
python3 tools/pmc_collect.py "$OUT" "$WL" "$@"
