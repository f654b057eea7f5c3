#!/usr/bin/env python3
"""profiles/r01_i_pmc_k2/p*.csv (scripts/pmc_k2.sh) -> profiles/r01_k2_binding.json: the resource that binds the
headline kernel, from rocprofv3 SQ counters.  VALU issue utilisation = SQ_INSTS_VALU x 4 cycles (a wave64
instruction occupies its SIMD's VALU for 4 cycles) / (1024 SIMDs x kernel cycles); kernel cycles =
GRBM_GUI_ACTIVE / 8 XCDs."""
import csv, glob, json, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = {}
for f in glob.glob(os.path.join(root, "profiles", "r01_i_pmc_k2", "p*.csv")):
    for r in csv.DictReader(open(f)):
        vals[r["Counter_Name"]] = float(r["Counter_Value"])
cycles = vals["GRBM_GUI_ACTIVE"] / 8.0
simd_cycles = 1024.0 * cycles
out = {
    "kernel": "k_anneal_csr_rank1<16, *>", "workload": "4096 replicas x 200 sweeps, n = 2638 (scripts/perf_k2.py --order slots)",
    "resource": "VALU instruction issue", "utilisation": vals["SQ_INSTS_VALU"] * 4.0 / simd_cycles,
    "salu_issue_utilisation": vals["SQ_INSTS_SALU"] * 4.0 / simd_cycles,
    "valu_wave_instructions": vals["SQ_INSTS_VALU"], "salu_wave_instructions": vals["SQ_INSTS_SALU"],
    "lds_instructions": vals["SQ_INSTS_LDS"], "vmem_read_instructions": vals["SQ_INSTS_VMEM_RD"],
    "l2_hit_rate": vals["TCC_HIT_sum"] / vals["TCC_REQ_sum"], "kernel_cycles": cycles,
    "source": "profiles/r01_i_pmc_k2/p1..p5.csv",
}
json.dump(out, open(os.path.join(root, "profiles", "r01_k2_binding.json"), "w"), indent=1)
print(out)
