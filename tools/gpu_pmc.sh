#!/bin/bash
# PMC passes (separate runs, counters only with --kernel-trace) for the bench command.
TAG=${1:-r01}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for CNT in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  NAME=$(echo $CNT | tr ' ' '+')
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$NAME
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 0 --iters 20 --cpu-iters 0 > $OUT/bench.json 2> $OUT/bench.err
  echo "$NAME rc=$?"
  F=$(find $OUT -name '*counter_collection.csv' | head -1)
  if [ -n "$F" ]; then
    python3 - "$F" <<'PY'
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0])
with open(f) as fh:
    for row in csv.DictReader(fh):
        k = (row["Kernel_Name"].split("(")[0][-60:], row["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(row["Counter_Value"])
for (kn, cn), (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{kn:60s} {cn:24s} dispatches={n:5d} mean={v / n:.6g}")
PY
    # keep a compact copy, drop the big raw file
    python3 - "$F" "$OUT/summary.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = (row["Kernel_Name"], row["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(row["Counter_Value"])
with open(sys.argv[2], "w") as out:
    out.write("kernel,counter,dispatches,mean_value\n")
    for (kn, cn), (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        out.write(f"\"{kn}\",{cn},{n},{v / n}\n")
PY
    find $OUT -name '*.csv' ! -name summary.csv -size +1M -delete
  fi
done
