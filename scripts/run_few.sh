# few-replica kernels with / without the threshold wavefront (development helper; bit-exactness checked by --check)
mkdir -p gpurun_out/s2
(
for blk in 128 64; do
echo "== 500 replicas, n = 2638, block $blk"
timeout -k 10 120 python scripts/perf_k2.py --replicas 500 --sweeps 1000 --order padded --block $blk --rounds 3 --check k2_tw=2 k2_tw=1
done
echo "== 500 replicas, n = 342 (one cluster), block 64"
timeout -k 10 120 python scripts/perf_k2.py --replicas 500 --sweeps 1000 --order padded --block 64 --n 342 --rounds 3 --check k2_tw=2 k2_tw=1
echo "== 1000 replicas, n = 2638, block 128"
timeout -k 10 120 python scripts/perf_k2.py --replicas 1000 --sweeps 1000 --order padded --block 128 --rounds 2 --check k2_tw=2 k2_tw=1
) > gpurun_out/s2/few.log 2>&1
