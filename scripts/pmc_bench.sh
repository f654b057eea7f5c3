#!/bin/bash
# rocprofv3 passes over the bench workload itself: kernel trace + stats over 1 warm-up + 20 timed steps (the first
# launch after an idle GPU runs at ramping clocks, ~30 % slower: 21 launches keep it out of the average), then
# FETCH_SIZE and WRITE_SIZE in their own --pmc passes (never combined with trace domains).
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_bench}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MI_BENCH_SKIP_50K=1      # the 10 GB dense side run has its own passes (scripts/pmc_dense50k.sh)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 1 --no-cpu-baseline > $out/trace.log 2>&1
for f in $(find $out/trace -name '*kernel_stats.csv'); do cp $f $out/kernel_stats.csv; done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/$c.log 2>&1
  for f in $(find $out/$c -name '*counter_collection.csv'); do head -1 $f > $out/$c.csv; grep anneal $f >> $out/$c.csv; done
  rm -rf $out/$c
done
rm -rf $out/trace
ls -la $out; cat $out/kernel_stats.csv
