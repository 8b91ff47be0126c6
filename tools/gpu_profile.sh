#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command; summary CSVs land in gpurun_out/prof_<tag>
TAG=${1:-r01}
shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:?}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --cpu-iters 0 "$@" > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$?"
cat $OUT/bench.json | cut -c1-1500
find $OUT -name '*kernel_stats.csv' | head -3
F=$(find $OUT -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && cp $F $OUT/kernel_stats.csv && head -20 $F | cut -c1-250
# the full trace is big: keep only the stats
find $OUT -name '*kernel_trace.csv' -size +20M -delete
du -sh $OUT
