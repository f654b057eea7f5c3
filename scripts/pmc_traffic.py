#!/usr/bin/env python3
"""Turn the per-dispatch FETCH_SIZE / WRITE_SIZE CSVs of scripts/pmc_bench.sh into
profiles/r03_pmc_traffic.json: HBM-side bytes per anneal-kernel launch, averaged over the launches of one
bench step (a 1000-sweep schedule is served by ceil(1000/32) launches).  FETCH_SIZE is in KiB and on gfx950
counts half of a wide coalesced stream -- MI355X_MICROARCH.md section HBM -- so it is doubled.
usage: pmc_traffic.py <dir with FETCH_SIZE.csv, WRITE_SIZE.csv> <replicas> <sweeps> [kernel name] [steps profiled = 4]"""
import csv, glob, json, os, sys
d = sys.argv[1]
KERNEL = sys.argv[4] if len(sys.argv) > 4 else "k_anneal_csr_rank1_pair<16>"
PATTERN = KERNEL.split("<")[0].replace("k_", "", 1) + "<"
vals = {}
for f in glob.glob(os.path.join(d, "*_SIZE.csv")):
    for r in csv.DictReader(open(f)):
        if PATTERN in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
STEPS = int(sys.argv[5]) if len(sys.argv) > 5 else 4            # bench.py --steps 3 --warmup 1
total = len(vals["FETCH_SIZE"])
launches = max(1, total // STEPS)                               # launches that serve ONE step
fetch = sum(vals["FETCH_SIZE"]) * 1024.0 * 2.0 / total
write = sum(vals["WRITE_SIZE"]) * 1024.0 / len(vals["WRITE_SIZE"])
out = {"replicas": int(sys.argv[2]), "sweeps": int(sys.argv[3]), "launches": launches,
       "kernel": KERNEL,
       "fetch_bytes_x2": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
       "hbm_bytes_per_step": (fetch + write) * launches,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 1; "
                 "per launch = mean over the anneal launches of the step; FETCH_SIZE KiB x 1024 x 2 (gfx950 "
                 "half-count) + WRITE_SIZE KiB x 1024"}
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                                 os.environ.get("MI_PROFILE_TAG", "r03") + "_pmc_traffic.json"), "w"), indent=1)
print(out)
