#!/usr/bin/env python3
"""<dir>/p*.csv (scripts/pmc_k1m.sh: separate rocprofv3 --pmc passes over K1m at a constant hot temperature on the
bench model) -> profiles/r02_k1m_binding.json: how busy the matrix pipe is, and what the waves do meanwhile.
  mfma    SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); cycles per MFMA instruction
  flops   2 * 16 * n_pad^2 per pass and workgroup (all rows, accepted or not) against 157.3 TFLOP/s
  waves   SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY (stalled at issue), SQ_WAIT_ANY (parked on a counter) / SQ_WAVE_CYCLES
kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs.   usage: k1m_binding.py <dir> <replicas> <passes> <n> [out.json]"""
import csv, glob, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, replicas, passes, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
outp = sys.argv[5] if len(sys.argv) > 5 else os.path.join(root, "profiles", "r02_k1m_binding.json")
vals, kernel, dur = {}, None, []
for f in sorted(glob.glob(os.path.join(d, "p*.csv"))):
    rows = list(csv.DictReader(open(f)))
    if not rows:
        continue
    last = max(int(r["Dispatch_Id"]) for r in rows)            # several anneals per pass: the last one (warm)
    for r in rows:
        if int(r["Dispatch_Id"]) != last:
            continue
        vals[r["Counter_Name"]] = float(r["Counter_Value"])
        kernel = r["Kernel_Name"]
        dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
name = re.search(r"(k_\w+<[^>]*>)", kernel).group(1)
cycles = vals["GRBM_GUI_ACTIVE"] / 8.0
ms = sorted(dur)[len(dur) // 2]
wgs = (replicas + 15) // 16
npad = ((n + 255) // 256) * 256
units = passes * 4 * ((n + 15) // 16) * wgs                    # 4-row units, all workgroups
flop = 2.0 * 16 * npad * (4 * ((n + 15) // 16) * 4) * passes * wgs
out = {
    "kernel": name, "replicas": replicas, "passes": passes, "n": n,
    "kernel_ms_profiled": ms, "kernel_cycles": cycles, "clock_GHz": cycles / (ms * 1e6),
    "mfma": {"instructions": vals["SQ_INSTS_MFMA"], "busy_cycles": vals["SQ_VALU_MFMA_BUSY_CYCLES"],
             "cycles_per_instruction": vals["SQ_VALU_MFMA_BUSY_CYCLES"] / vals["SQ_INSTS_MFMA"],
             "pipe_busy_frac": vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles),
             "tflops": flop / (ms * 1e-3) / 1e12, "peak_tflops": 157.3, "frac_of_peak": flop / (ms * 1e-3) / 157.3e12},
    "per_unit_per_cu": {k: vals[c] / units for k, c in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"),
                                                        ("lds", "SQ_INSTS_LDS"), ("branch", "SQ_INSTS_BRANCH"),
                                                        ("vmem", "SQ_INSTS_VMEM_RD"), ("mfma", "SQ_INSTS_MFMA"),
                                                        ("ifetch", "SQ_IFETCH"))},
    "cycles_per_unit": cycles * 256.0 / units * (wgs / 256.0) if wgs >= 256 else cycles / (units / wgs),
    "lds": {"busy_frac": vals["SQ_LDS_IDX_ACTIVE"] / (256.0 * cycles),
            "bank_conflict_share": vals["SQ_LDS_BANK_CONFLICT"] / max(vals["SQ_LDS_IDX_ACTIVE"], 1.0)},
    "l2": {"request_GBps": vals["TCC_REQ_sum"] * 128.0 / (ms * 1e-3) / 1e9, "hit_rate": vals["TCC_HIT_sum"] / vals["TCC_REQ_sum"]},
    "waves": {"active_issuing": vals["SQ_ACTIVE_INST_ANY"] / vals["SQ_WAVE_CYCLES"],
              "stalled_on_issue": vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"],
              "parked_on_waitcnt": vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"]},
    "source": "%s/p*.csv (scripts/pmc_k1m.sh)" % os.path.relpath(d, root),
}
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out, indent=1))
