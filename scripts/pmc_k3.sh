#!/bin/bash
# SQ / LDS / cache counters of the Potts kernel (K3) on the bench graph, K = 8, 4096 replicas x 200 sweeps,
# slot-independent order: separate rocprofv3 --pmc passes over scripts/perf_k3.py.   usage: scripts/pmc_k3.sh <tag>
set -u
tag=${1:-pmc_k3}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export K3_ONLY_SLOTS=1
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python $GRAFT_REPO_ROOT/scripts/perf_k3.py > $out/p$i.log 2>&1 || echo "pass $i failed (rc=$?)" >> $out/status.txt
  for f in $(find $out/p$i -name '*counter_collection.csv'); do head -1 $f > $out/p$i.csv; grep anneal_potts $f >> $out/p$i.csv; done
  rm -rf $out/p$i
done
ls -la $out
