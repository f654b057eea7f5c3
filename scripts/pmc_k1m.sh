#!/bin/bash
# SQ / MFMA / LDS counters of K1m (dense chain on the matrix cores) at a constant hot temperature on the bench
# model: separate rocprofv3 --pmc passes over scripts/perf_k1m.py.   usage: scripts/pmc_k1m.sh <tag> [beta] [sweeps]
set -u
tag=${1:-pmc_k1m}
beta=${2:-0.01}
sweeps=${3:-100}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_IFETCH" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_MISC SQ_IFETCH_LEVEL SQ_LEVEL_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python $GRAFT_REPO_ROOT/scripts/perf_k1m.py --beta $beta --sweeps $sweeps debug=0 > $out/p$i.log 2>&1 || echo "pass $i failed (rc=$?)" >> $out/status.txt
  for f in $(find $out/p$i -name '*counter_collection.csv'); do head -1 $f > $out/p$i.csv; grep anneal_dense_mfma $f >> $out/p$i.csv; done
  rm -rf $out/p$i
done
ls -la $out
