# timing-only A/B of the pair kernel's experiments (wrong chains; see the MI_K2P_DBG_* switches in sparse_pair_kernels.hip);
# every arm in its own process and its own loop (the clock a kernel gets depends on what ran before it)
mkdir -p gpurun_out/s2
P=scrna_seq_qannealing_clustering_amd
(
echo "== shipped"; timeout -k 10 120 python scripts/perf_k2.py --sweeps 1000 --order padded --rounds 4 k2_tw=1
for v in "$@"; do echo "== $v"; MI_SA_LIB=$PWD/$P/libmi_sa_dbg_$v.so timeout -k 10 120 python scripts/perf_k2.py --sweeps 1000 --order padded --rounds 4 k2_tw=1; done
echo "== shipped, no threshold wavefront"; timeout -k 10 120 python scripts/perf_k2.py --sweeps 1000 --order padded --rounds 4 k2_tw=2
) > gpurun_out/s2/tw_exp.log 2>&1
