#!/usr/bin/env python3
"""<dir>/p*.csv (scripts/pmc_k2.sh: separate rocprofv3 --pmc passes over the headline anneal kernel at the bench
shape) -> profiles/r03_k2_binding.json: what the kernel uses of each resource that could bind it.
  L2      TCC_REQ x 128 B per request / kernel time, against 34.5 TB/s (MI355X_MICROARCH.md, L2 aggregate)
  LDS     SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles); share of it that is bank-conflict cycles
  VALU    SQ_INSTS_VALU x c / (1024 SIMDs x kernel cycles) for c = 2 (SIMD-32 pass count of a wave64 op) and
          c = 3.1 (measured issue cost of the VOP3 forms that dominate this kernel, scripts/ubench_valu.hip)
  waves   SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY as shares of SQ_WAVE_CYCLES
kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs.   usage: k2_binding.py <dir> <replicas> <sweeps> [out.json]"""
import csv, glob, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, replicas, sweeps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
outp = sys.argv[4] if len(sys.argv) > 4 else os.path.join(root, "profiles", "r03_k2_binding.json")
vals, kernel, dur = {}, None, []
for f in sorted(glob.glob(os.path.join(d, "p*.csv"))):
    rows = list(csv.DictReader(open(f)))
    last = max(int(r["Dispatch_Id"]) for r in rows)            # several anneals per pass: the last one (warm)
    for r in rows:
        if int(r["Dispatch_Id"]) != last:
            continue
        vals[r["Counter_Name"]] = float(r["Counter_Value"])
        kernel = r["Kernel_Name"]
        dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
name = re.search(r"(k_\w+<[^>]*>)", kernel).group(1)
name = name.replace(", true>", ", tw>").replace(", false>", ">")      # (the name the library reports: mi_sa_last_kernel_name)
cycles = vals["GRBM_GUI_ACTIVE"] / 8.0
ms = sorted(dur)[len(dur) // 2]
l2_bytes = vals["TCC_REQ_sum"] * 128.0
out = {
    "kernel": name, "replicas": replicas, "sweeps": sweeps,
    "kernel_ms_profiled": ms, "kernel_cycles": cycles, "clock_GHz": cycles / (ms * 1e6),
    "l2": {"request_bytes": l2_bytes, "GBps": l2_bytes / (ms * 1e-3) / 1e9, "peak_GBps": 34500.0,
           "frac": l2_bytes / (ms * 1e-3) / 1e9 / 34500.0, "hit_rate": vals["TCC_HIT_sum"] / vals["TCC_REQ_sum"]},
    "lds": {"busy_frac": vals["SQ_LDS_IDX_ACTIVE"] / (256.0 * cycles),
            "bank_conflict_share": vals["SQ_LDS_BANK_CONFLICT"] / max(vals["SQ_LDS_IDX_ACTIVE"], 1.0)},
    "valu": {"wave_instructions": vals["SQ_INSTS_VALU"],
             "issue_frac_at_2_cycles": vals["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cycles),
             "issue_frac_at_3p1_cycles": vals["SQ_INSTS_VALU"] * 3.1 / (1024.0 * cycles)},
    "salu_wave_instructions": vals["SQ_INSTS_SALU"], "lds_instructions": vals["SQ_INSTS_LDS"],
    "vmem_read_instructions": vals["SQ_INSTS_VMEM_RD"],
    "waves": {"active_issuing": vals["SQ_ACTIVE_INST_ANY"] / vals["SQ_WAVE_CYCLES"],
              "stalled_on_issue": vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"],
              "parked_on_waitcnt": vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"]},
    "source": "%s/p*.csv (scripts/pmc_k2.sh / pmc_k3.sh)" % os.path.relpath(d, root),
}
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out, indent=1))
