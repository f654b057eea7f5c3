#!/bin/bash
# SQ / MFMA counters of the full pass of K1g (k_xg_panel<false>) on the 50 000-variable dense model, 1024 replicas:
# two rocprofv3 --pmc passes over scripts/run_dense50k.py, averaged over the 3000-workgroup launches.   usage: scripts/pmc_k1g.sh
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/xl/pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python $GRAFT_REPO_ROOT/scripts/run_dense50k.py --replicas 1024 --sweeps 1 --start 0 > $out/p$i.log 2>&1
  for f in $(find $out/p$i -name '*counter_collection.csv'); do head -1 $f > $out/p$i.csv; grep "k_xg_panel<false>" $f | tail -400 >> $out/p$i.csv; done
  rm -rf $out/p$i
done
python3 - <<PY
import csv, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
for i in (1, 2):
    for r in csv.DictReader(open("$out/p%d.csv" % i)):
        if int(r["Grid_Size"]) < 100000: continue      # the 16-workgroup first parts
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(tot): print("%-28s %.4g per launch (%d launches)" % (k, tot[k] / n[k], n[k]))
cyc = tot["GRBM_GUI_ACTIVE"] / n["GRBM_GUI_ACTIVE"] / 8
print("launch duration %.1f us under the counters: clock %.3f GHz" % (sum(dur) / len(dur) / 1e3, cyc / (sum(dur) / len(dur))))
print("cycles per launch", cyc, "MFMA busy frac", tot["SQ_VALU_MFMA_BUSY_CYCLES"] / n["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc))
w = tot["SQ_WAVE_CYCLES"]
print("waves: active %.3f stalled-at-issue %.3f parked %.3f wait-LDS %.3f" % (tot["SQ_ACTIVE_INST_ANY"] / w, tot["SQ_WAIT_INST_ANY"] / w, tot["SQ_WAIT_ANY"] / w, tot["SQ_WAIT_INST_LDS"] / w))
print("LDS busy", tot["SQ_LDS_IDX_ACTIVE"] / n["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), "bank conflict share", tot["SQ_LDS_BANK_CONFLICT"] / max(tot["SQ_LDS_IDX_ACTIVE"], 1))
PY
