"""Sum the counter_collection CSVs of tools/pmc_passes.sh per counter for the dominant kernel (per launch) and write
<out>/pmc_<workload>.json -- the file bench.py looks up under profiles/rNN/ for its roofline / issue figures.

    python tools/pmc_collect.py <out_dir> [workload] [--width W --height H --spp S --photons P]
"""
import argparse
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("workload", nargs="?", default="C3")
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--height", type=int, default=0)
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--photons", type=int, default=0)
args, _ = ap.parse_known_args()

from rpt_amd import scenes  # noqa: E402  (pure Python: scene constants only)
eps = args.workload.endswith("eps")
cfg = scenes.CONFIGS[args.workload[:-3] if eps else args.workload]()[2]
# (C4eps: a camera pass is two kernels -- the fp32 one that hands the selections over, then the fp64 one: their per-launch means are added)
kernels = ((["photon_query_kernel", "photon_surface_f64_kernel"] if eps else ["photon_query_kernel"]) if "photons" in cfg
           else ["render_f64_kernel"] if eps else ["render_kernel"])
kernel = " + ".join(kernels)
res, launches_per_counter, durations = defaultdict(float), {}, []
for kern in kernels:
    tot, launches = defaultdict(float), defaultdict(set)
    for f in glob.glob(args.out + "/pass*/**/*counter_collection.csv", recursive=True):
        rows = [row for row in csv.DictReader(open(f)) if kern in row.get("Kernel_Name", "")]
        if kern.startswith("photon_") and rows:
            # the camera pass has no separate counters instantiation: the LAST launch of a bench.py run is its untimed counters pass
            # (diagnostic timers and atomics: ten times slower; one sample per pixel) and does not belong in the means
            last = max(int(row["Dispatch_Id"]) for row in rows)
            rows = [row for row in rows if int(row["Dispatch_Id"]) != last]
        for row in rows:
            name = row.get("Kernel_Name", "")
            if kern == "render_f64_kernel" and re.search(r"render_f64_kernel<\w+, true", name):
                continue  # (the counters instantiation)
            if kern == "render_kernel" and (re.search(r"render_kernel<\w+, \d, true[,>]", name) or re.search(r"render_kernelILb[01]ELi\dELb1E", name)):
                continue  # the timed launches only (the COUNT = true instantiation is bench.py's untimed counter pass)
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            launches[row["Counter_Name"]].add(row.get("Dispatch_Id"))
    for k in tot:
        res[k] += tot[k] / max(1, len(launches[k]))
        launches_per_counter[f"{kern}:{k}"] = len(launches[k])
res = {k: res[k] for k in sorted(res)}
res["_launches_per_counter"] = launches_per_counter
res["_kernel"] = kernel
res["_config"] = {"workload": args.workload, "width": args.width or cfg["width"], "height": args.height or cfg["height"],
                  "spp": args.spp or cfg["spp"], "n_gpus": 1}
if "photons" in cfg:
    res["_config"]["photons"] = args.photons or cfg["photons"]
from __graft_entry__ import library_source_digest  # noqa: E402
res["_source_digest"] = library_source_digest()   # the kernel sources this profile describes (bench.py uses no other profile)
res["_how"] = ("rocprofv3 --kernel-trace --pmc <group>, one run per counter group (tools/pmc_passes.sh), bench.py --steps 1 --warmup 1; "
               "values are per-launch means over the timed launches; FETCH_SIZE / WRITE_SIZE in KB")
path = os.path.join(args.out, f"pmc_{args.workload}.json")
json.dump(res, open(path, "w"), indent=1)
print(json.dumps(res, indent=1))
