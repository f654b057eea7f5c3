#!/usr/bin/env python3
"""Per-kernel durations of a rocprofv3 --kernel-trace run under gpurun_out/<tag>, split by grid size (the two parts of a
K1g full pass share a kernel name).  usage: trace_by_grid.py <tag> ..."""
import csv,glob,sys,collections,re
for tag in sys.argv[1:]:
    files=glob.glob(f"/root/repo/gpurun_out/{tag}/**/*kernel_trace.csv", recursive=True)
    agg=collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            m=re.search(r"(k_\w+(<[^>]*>)?)", row["Kernel_Name"]); name=m.group(1) if m else row["Kernel_Name"][:40]
            key=(name, row.get("Grid_Size_X", row.get("Grid_Size","")), row.get("Grid_Size_Y",""))
            agg[key].append(int(row["End_Timestamp"])-int(row["Start_Timestamp"]))
    print(tag)
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:12]:
        print("   %-40s grid %s,%s  n=%5d  avg %8.1f us  total %8.2f ms" % (k[0][:40],k[1],k[2],len(v),sum(v)/len(v)/1e3,sum(v)/1e6))
