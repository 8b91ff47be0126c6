#!/bin/bash
# GKOCG turn shapes on one box, alternating: five / four (merged) launches against the leader finalisation (three launches)
#   tools/ab_lead.sh [bench.py flags ...]     (development tool)
run() { python bench.py --steps 5 --warmup 2 --cpu-iters 0 --no-general-legs --live-pmc off "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f it/s  %8.2f us/turn  spmv %7.2f us' % (d['value'], 1e3*d['ms_per_step']/d['config'].get('cg_iters_per_step', 100), 1e3*d['roofline']['avg_kernel_ms']))"; }
for r in 1 2 3; do
  echo -n "round $r five launches       "; run --prop leadFinalizers=0 --prop fusedTurnBig=0 "$@"
  echo -n "round $r default w/o lead    "; run --prop leadFinalizers=0 "$@"
  echo -n "round $r lead, three launches"; run --prop leadFinalizers=1 --prop fusedTurnBig=0 "$@"
  echo -n "round $r lead, two launches  "; run --prop leadFinalizers=1 --prop fusedTurnBig=1 "$@"
done
