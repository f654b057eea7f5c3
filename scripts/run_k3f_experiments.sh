# timing-only A/B of K3f's experiments (wrong chains; MI_K3F_DBG_* in potts_fast_kernels.hip), every arm in its own process
mkdir -p gpurun_out/s2
P=scrna_seq_qannealing_clustering_amd
(
echo "== shipped"; timeout -k 10 120 python scripts/perf_k3_fast.py 8 4096 200 | grep "fast=0"
for v in NOFETCH LINEAR NOATOMICS; do echo "== $v"; MI_SA_LIB=$PWD/$P/libmi_sa_k3dbg_$v.so timeout -k 10 120 python scripts/perf_k3_fast.py 8 4096 200 | grep "fast=0"; done
) > gpurun_out/s2/k3f_exp.log 2>&1
