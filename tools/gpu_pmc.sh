#!/bin/bash
# PMC passes for the bench command: one rocprofv3 run per counter set (FETCH_SIZE and WRITE_SIZE
# cannot share a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), --kernel-trace only.
#   gpu_pmc.sh TAG [extra bench.py flags, e.g. --no-compress]
TAG=${1:-r01}
shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
DIRS=""
for CNT in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  NAME=$(echo $CNT | tr ' ' '+')
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$NAME
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 0 --iters 20 --cpu-iters 0 "$@" > $OUT/bench.json 2> $OUT/bench.err
  echo "$NAME rc=$?"
  DIRS="$DIRS $OUT"
done
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_summary.json $DIRS
find gpurun_out -name '*counter_collection.csv' -size +1M -delete
find gpurun_out -name '*kernel_trace.csv' -size +1M -delete
