#!/bin/bash
# MFMA counters of K4 (k_energy_dense_mfma): separate rocprofv3 --pmc passes.  usage: scripts/pmc_k4.sh <tag>
set -u
tag=${1:-pmc_k4}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python $GRAFT_REPO_ROOT/scripts/pmc_k4.py > $out/p$i.log 2>&1 || echo "pass $i failed (rc=$?)" >> $out/status.txt
  for f in $(find $out/p$i -name '*counter_collection.csv'); do head -1 $f > $out/p$i.csv; grep energy_dense_mfma $f >> $out/p$i.csv; done
  rm -rf $out/p$i
done
ls -la $out; cat $out/status.txt 2>/dev/null; grep "path" $out/p1.log
