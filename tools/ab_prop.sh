#!/bin/bash
# A/B of one solver property on one box: bench.py alternately with --prop KEY=A and --prop KEY=B
#   tools/ab_prop.sh KEY A B [rounds] [bench.py flags ...]      (development tool; prints it/s, us per turn, SpMV us)
key=$1; a=$2; b=$3; rounds=${4:-3}; shift 4
for r in $(seq 1 $rounds); do
  for v in $a $b; do
    python bench.py --steps 5 --warmup 2 --cpu-iters 0 --no-general-legs --live-pmc off --prop $key=$v "$@" 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $key=$v  %9.1f it/s  %8.2f us/turn  spmv %7.2f us' % (d['value'], 1e3*d['ms_per_step']/d['config'].get('cg_iters_per_step', 100), 1e3*d['roofline']['avg_kernel_ms']))"
  done
done
