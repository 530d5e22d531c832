"""Sum the counter_collection CSVs of tools/pmc_passes.sh per counter for the render kernel (per launch)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

out = sys.argv[1]
tot, launches = defaultdict(float), defaultdict(set)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "render_kernel" not in name or re.search(r"render_kernel<\w+, \d, true[,>]", name) or re.search(r"render_kernelILb[01]ELi\dELb1E", name):
            continue  # the timed launches only (the COUNT = true instantiation is bench.py's untimed counter pass)
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
        launches[row["Counter_Name"]].add(row.get("Dispatch_Id"))
res = {k: tot[k] / max(1, len(launches[k])) for k in sorted(tot)}
res["_launches_per_counter"] = {k: len(v) for k, v in launches.items()}
json.dump(res, open(out + "/pmc_render_kernel.json", "w"), indent=1)
print(json.dumps(res, indent=1))
