#!/bin/bash
# Issue / wait counters of the per-mesh-tree render kernel's flavours (parked, detached, streamed) on C5, one rocprofv3 --pmc run
# per counter group (counters only, no tracing beside --kernel-trace).
# Usage (GPU box, repo root): bash tools/pmc_flavours.sh <out_dir> [width=2048] [spp=64]
set -e
OUT=$(realpath -m "$1"); W=${2:-2048}; SPP=${3:-64}
mkdir -p "$OUT"
REPO=$(pwd)
export CHUNK_SPP=16
cd /tmp && export TMPDIR=/tmp
GROUPS_ALL=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
  "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_LDS"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
i=0
for group in "${GROUPS_ALL[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group -d "$OUT/pass$i" -o p --output-format csv -- python3 "$REPO/tools/detach_sweep.py" C5 $W $SPP 44:28:16 s:48:16:4 > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; }
  echo "pass $i done"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, json, re, sys
from collections import defaultdict
out = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(set))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        m = re.search(r"render_kernel<true, 1, false, false, (\d)>", name) or re.search(r"render_kernelILb1ELi1ELb0ELb0ELi(\d)E", name)
        if not m:
            continue
        k = {"0": "parked", "1": "detached", "2": "streamed"}[m.group(1)]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]].add(row.get("Dispatch_Id"))
res = {k: {c: v / max(1, len(n[k][c])) for c, v in sorted(d.items())} for k, d in tot.items()}
for k, d in res.items():   # derived
    if "SQ_INSTS_VALU" in d and "GRBM_GUI_ACTIVE" in d:
        d["_valu_issue_frac"] = d["SQ_INSTS_VALU"] / (1024 * (d["GRBM_GUI_ACTIVE"] / 8) / 2)
    if d.get("SQ_THREAD_CYCLES_VALU") and d.get("SQ_ACTIVE_INST_VALU"):
        d["_active_lanes"] = d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_ACTIVE_INST_VALU"])
    if d.get("SQ_WAVE_CYCLES"):
        d["_wait_share"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]; d["_wait_inst_share"] = d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"]
    d["_launches"] = {c: len(v) for c, v in n[k].items()}
json.dump(res, open(out + "/pmc_flavours.json", "w"), indent=1)
for k, d in res.items():
    print(k, {c: (round(v, 4) if isinstance(v, float) and v < 10 else (int(v) if isinstance(v, float) else v)) for c, v in d.items() if not c.startswith("_l")})
PY
