#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root):  bash tools/final_profiles.sh <tag>
# Writes bench JSON lines, rocprofv3 --stats summaries and the PMC passes under gpurun_out/<tag>/.
# (The traced runs use one stream: with two, a traced kernel's duration includes the time it shares the CUs with its neighbour.)
set -e
TAG=${1:-final}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
python3 bench.py --workload C2 --no-cpu-baseline > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
python3 bench.py --workload C5 --steps 2 --warmup 1 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
python3 bench.py --workload C4 --steps 2 --warmup 1 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
python3 tools/fractal_bench.py 100 > "$OUT/fractal.json" 2> "$OUT/fractal.err"
echo "benches done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats_c3" -o s --output-format csv -- python3 "$REPO/bench.py" --streams 1 --no-cpu-baseline > "$OUT/stats_c3.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/stats_c4" -o s --output-format csv -- python3 "$REPO/bench.py" --workload C4 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/stats_c4.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/stats_c5" -o s --output-format csv -- python3 "$REPO/bench.py" --workload C5 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/stats_c5.log" 2>&1
echo "stats done"
cd "$REPO"
bash tools/pmc_passes.sh "$OUT/pmc_c3" C3 > "$OUT/pmc_c3.log" 2>&1
bash tools/pmc_passes.sh "$OUT/pmc_c5" C5 quick > "$OUT/pmc_c5.log" 2>&1
bash tools/pmc_passes.sh "$OUT/pmc_c4" C4 quick > "$OUT/pmc_c4.log" 2>&1
echo "pmc done"
find "$OUT" -name "*kernel_stats.csv" | head
