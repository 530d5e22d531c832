#!/bin/bash
# A/B of library builds on the bench workloads (GPU box): tools/ab_libs.sh "<lib> ..." "<workload> ..." [bench args]
# Prints value / ms_per_step / kernel_ms per (library, workload).
libs=$1; wls=$2; shift 2
for l in $libs; do
  for w in $wls; do
    RPT_LIB=$l timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    try: d=json.loads(line)
    except Exception: continue
    print('$l', '$w', 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])
"
  done
done
