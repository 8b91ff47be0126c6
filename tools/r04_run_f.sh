export HSA_ENABLE_IPC_MODE_LEGACY=0 OGL_CASE_CACHE_DIR=/tmp/cc
mkdir -p /tmp/cc gpurun_out
python tools/dump_pattern.py voronoi 3000000 /tmp/cc/vor3m.bin 2>&1 | tail -1
tools/bin/win_tune /tmp/cc/vor3m.bin 50 2>&1 | tee gpurun_out/r04f_win_tune_vor3m.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_r04f_win; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -- $GRAFT_REPO_ROOT/tools/bin/win_tune /tmp/cc/vor3m.bin 3 > $OUT/run.txt 2>&1
F=$(find $OUT -name '*counter_collection.csv' | head -1)
python3 - "$F" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r04f_win_fetch.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if r.get("Counter_Name") != "FETCH_SIZE": continue
        k = r["Kernel_Name"][:90]
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{k:92s} launches {n:4d}  FETCH_SIZE per launch {v / n / 1024:10.1f} MiB-units(KiB counter)  x2 corrected {2 * v / n * 1024 / 1e6:9.1f} MB")
PY
find $OUT -name '*.csv' -size +1M -delete
