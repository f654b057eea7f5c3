#!/bin/bash
# Fabric-side read bytes of the dense 50k run (K1g / K1x) on the 10 GB dense model (bench.py's other_kernels.dense_xl_50k shape): one
# rocprofv3 --pmc FETCH_SIZE pass over scripts/run_dense50k.py.   usage: scripts/pmc_dense50k.sh <tag>
set -u
tag=${1:-pmc_dense50k}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- python $GRAFT_REPO_ROOT/scripts/run_dense50k.py --replicas 1024 --sweeps 4 --start 0 > $out/run.json 2> $out/run.err || echo "failed rc=$?" >> $out/status.txt
for f in $(find $out/f -name '*counter_collection.csv'); do head -1 $f > $out/FETCH_SIZE.csv; grep -E 'dense_xl|k_xg_' $f >> $out/FETCH_SIZE.csv; done
rm -rf $out/f
python - <<PY
import csv, json
d = json.loads([l for l in open("$out/run.json") if l.startswith("{")][-1])
kib = sum(float(r["Counter_Value"]) for r in csv.DictReader(open("$out/FETCH_SIZE.csv")))
d["pmc"] = {"FETCH_SIZE_KiB": kib, "fabric_read_bytes_x2": kib * 1024 * 2}
d["fabric_read_GBps_profiled_run"] = kib * 1024 * 2 / (d["kernel_ms"] * 1e-3) / 1e9   # (counter passes serialise the launches: bench.py divides the bytes by its own time)
d["method"] = "rocprofv3 --pmc FETCH_SIZE on scripts/run_dense50k.py --replicas 1024 --sweeps 4 --start 0; x2 per the gfx950 half-count note (MI355X_MICROARCH.md, HBM); kernel_ms of the profiled run"
json.dump(d, open("$out/r02_dense50k.json", "w"), indent=1)
print(json.dumps(d))
PY
