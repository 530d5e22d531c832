#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root):  bash tools/final_profiles.sh <tag>
# Writes bench JSON lines, rocprofv3 --stats summaries and the PMC passes under gpurun_out/<tag>/.
# (The traced per-workload runs use one stream: with two, a traced kernel's duration includes the time it shares the CUs
# with its neighbour.  The PMC passes come FIRST: the bench lines use a profile only if it was taken from the sources in the
# tree, and the copies under profiles/rNN/ are what they look up -- tools/final_profiles.sh leaves them in <out>/pmc_*/.)
set -e
TAG=${1:-final}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
bash tools/pmc_passes.sh "$OUT/pmc_c3" C3 > "$OUT/pmc_c3.log" 2>&1
echo "pmc c3 done"
bash tools/pmc_passes.sh "$OUT/pmc_C3eps" C3eps > "$OUT/pmc_C3eps.log" 2>&1
echo "pmc C3eps done"
for wl in C5 C4 C4eps C2 C2eps C5G; do
  bash tools/pmc_passes.sh "$OUT/pmc_$wl" $wl quick > "$OUT/pmc_$wl.log" 2>&1
  echo "pmc $wl done"
done
cd /tmp && export TMPDIR=/tmp
# the driver's own command, traced
rocprofv3 --kernel-trace --stats -d "$OUT/stats_default" -o s --output-format csv -- python3 "$REPO/bench.py" --no-cpu-baseline > "$OUT/stats_default.log" 2>&1
echo "stats default done"
rocprofv3 --kernel-trace --stats -d "$OUT/stats_c3" -o s --output-format csv -- python3 "$REPO/bench.py" --streams 1 --no-cpu-baseline --no-secondary > "$OUT/stats_c3.log" 2>&1
for wl in C3eps C2 C2eps C4 C4eps C5 C5G; do
  rocprofv3 --kernel-trace --stats -d "$OUT/stats_$wl" -o s --output-format csv -- python3 "$REPO/bench.py" --workload $wl --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/stats_$wl.log" 2>&1
  echo "stats $wl done"
done
cd "$REPO"
find "$OUT" -name "*kernel_stats.csv" | head
