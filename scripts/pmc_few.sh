#!/bin/bash
# SQ / LDS counters of the few-replica kernel (K2w + threshold wavefront) at the reference's call shape: 500 reads x 1000
# sweeps on the bench graph, 128-seat layout -- separate rocprofv3 --pmc passes over scripts/perf_k2.py.
# usage: scripts/pmc_few.sh <tag>
set -u
tag=${1:-pmc_few}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_IFETCH" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python $GRAFT_REPO_ROOT/scripts/perf_k2.py --rounds 1 --order padded --block 128 --replicas 500 --sweeps 1000 k2_tw=1 > $out/p$i.log 2>&1 || echo "pass $i failed (rc=$?)" >> $out/status.txt
  for f in $(find $out/p$i -name '*counter_collection.csv'); do head -1 $f > $out/p$i.csv; grep anneal_csr $f >> $out/p$i.csv; done
  rm -rf $out/p$i
done
ls -la $out
