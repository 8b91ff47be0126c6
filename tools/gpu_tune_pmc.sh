#!/bin/bash
# FETCH_SIZE per SpMV variant of tools/spmv_tune (kernel names differ by template arguments)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tune_pmc
rm -rf $OUT; mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p tools/bin && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/spmv_tune.hip -o tools/bin/spmv_tune || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -- ./tools/bin/spmv_tune 216 6 > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    name = k.split("(")[0][-70:]
    print(f"{name:72s} n={len(v):3d} mean FETCH_SIZE={sum(v)/len(v):10.0f} KiB  -> x2 = {2*1024*sum(v)/len(v)/1e6:8.1f} MB")
PY
find $OUT -name '*.csv' -size +1M -delete
