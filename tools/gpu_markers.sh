#!/bin/bash
# ROCTx ranges of the plug-in phases under rocprofv3 --marker-trace (no counters): end-to-end solve()
# calls incl. PCIe, as an OpenFOAM time step would issue them.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/markers
rm -rf $OUT; mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:?}"
rocprofv3 --marker-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --cpu-iters 0 > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$?"
F=$(find $OUT -name '*marker*stats*.csv' | head -1)
echo "stats file: $F"
[ -n "$F" ] && cp $F $OUT/marker_stats.csv && cat $F | cut -c1-200
find $OUT -name '*.csv' | head
