#!/usr/bin/env python3
"""Time line of one solve from a rocprofv3 --kernel-trace CSV: what runs before the first turn, one turn kernel by kernel
with the gaps between dispatches, what runs after the last turn, and where the solve's wall time goes.

    python tools/turn_timeline.py <dir with *_kernel_trace.csv> [solve index, default: the last one]

A "turn" is recognised by the first kernel name that repeats most often (the in-loop SpMV); a solve by the
k_reset_scalars launch that opens it."""
import csv
import glob
import re
import sys
from collections import Counter

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_reset_scalars")]
if not starts:
    sys.exit("no k_reset_scalars launch in the trace")
lo = starts[which]
hi = starts[starts.index(lo) + 1] if starts.index(lo) + 1 < len(starts) else len(rows)
solve = rows[lo:hi]
names = Counter(r[2] for r in solve)
spmv = max((n for n in names if "spmv" in n or "turn_sym" in n), key=lambda n: names[n])
idx = [i for i, r in enumerate(solve) if r[2] == spmv]
# the SpMV also runs in the prologue (A xbar, the initial residual): the loop starts where its spacing becomes regular
gaps = [idx[i + 1] - idx[i] for i in range(len(idx) - 1)]
per = Counter(gaps).most_common(1)[0][0]
first = next(i for i in range(len(gaps)) if all(g == per for g in gaps[i:i + 3]))
t_first = idx[first] - (per - 1) if solve[idx[first] - 1][2] != spmv and per > 1 else idx[first]
# a turn = `per` consecutive launches; find the launch the turn starts with by looking at turn 10
def us(ns):
    return ns / 1e3
print(f"# solve #{starts.index(lo)} of {len(starts)}: {len(solve)} launches, {names[spmv]} x {spmv}, {per} launches per turn")
loop_lo, loop_hi = idx[first], idx[-1]
turns = (len([i for i in idx if i >= loop_lo]) - 1)
loop_span = solve[loop_hi][0] - solve[loop_lo][0]
print(f"# in-loop: {turns} turns in {us(loop_span):.1f} us = {us(loop_span) / max(turns, 1):.2f} us per turn")
print(f"# prologue (reset -> first in-loop SpMV): {us(solve[loop_lo][0] - solve[0][0]):.1f} us, {loop_lo} launches:")
for r in solve[:loop_lo]:
    print(f"    {us(r[0] - solve[0][0]):9.1f}  {us(r[1] - r[0]):8.1f} us  {r[2]}")
k = idx[first + min(10, len(idx) - first - 2)]
print(f"# one turn (the {min(10, len(idx) - first - 2) + 1}th), from the SpMV's dispatch:")
prev_end = None
for r in solve[k:k + per + 1]:
    gap = "" if prev_end is None else f"  (gap {us(r[0] - prev_end):5.2f} us)"
    print(f"    {us(r[0] - solve[k][0]):9.2f}  {us(r[1] - r[0]):8.2f} us  {r[2]}{gap}")
    prev_end = r[1]
# per-kernel averages over the in-loop launches + average gap in front of each
acc = {}
for i in range(loop_lo, loop_hi):
    r = solve[i]
    a = acc.setdefault(r[2], [0, 0.0, 0.0])
    a[0] += 1
    a[1] += r[1] - r[0]
    a[2] += max(0, r[0] - solve[i - 1][1])
print("# in-loop averages: launches, mean duration, mean gap in front")
for n, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"    {a[0]:6d}  {us(a[1] / a[0]):8.2f} us  gap {us(a[2] / a[0]):5.2f} us  {n}")
# the epilogue ends where the host takes over (a gap of more than 100 us: copy-back, the next set_matrix, other legs)
tail = []
prev = solve[loop_hi][1]
for r in solve[loop_hi + 1:]:
    if r[0] - prev > 100000:
        break
    tail.append(r)
    prev = r[1]
print(f"# epilogue (after the last in-loop SpMV, up to the first host-side pause): {len(tail)} launches, "
      f"{us((tail[-1][1] if tail else solve[loop_hi][1]) - solve[loop_hi][1]):.1f} us:")
for r in tail:
    print(f"    {us(r[0] - solve[loop_hi][1]):9.1f}  {us(r[1] - r[0]):8.1f} us  {r[2]}")
