#!/usr/bin/env python3
"""Merge the rocprofv3 --kernel-trace CSVs of several ranks that shared one GPU into one time line and print a
window of it (a few GKOCG turns): which kernel of which rank ran when, and what overlapped what.

  ranks_trace.py DIR [first_kernel_index [count]]      (DIR holds one sub-tree per rank, any depth)
"""
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][-40:]


def main():
    root = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else None
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    rows = []
    files = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))
    for rank, f in enumerate(files):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), rank, short(r["Kernel_Name"])))
    rows.sort()
    if not rows:
        print("no kernel traces under", root)
        return
    # default window: the middle of the run, starting at a step_1x of rank 0
    if first is None:
        mid = len(rows) // 2
        first = next((i for i in range(mid, len(rows)) if rows[i][2] == 0 and "k_cg_step1x" in rows[i][3]), mid)
    t0 = rows[first][0]
    print(f"# {len(files)} ranks, {len(rows)} kernel launches; window of {count} from launch {first}; times in us from its start")
    print("#   start     end    dur  rank  kernel                                   overlaps (rank:kernel)")
    win = rows[first:first + count]
    for i, (s, e, rk, k) in enumerate(win):
        ov = [f"{r2}:{k2.split('<')[0]}" for (s2, e2, r2, k2) in win if r2 != rk and s2 < e and e2 > s]
        print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:7.2f} {(e - s) / 1e3:6.2f}  {rk:4d}  {k[:40]:40s} {' '.join(ov)}")
    # overlap summary over the whole trace: time during which kernels of >= 2 ranks were running
    ev = []
    for s, e, rk, k in rows:
        ev += [(s, 1), (e, -1)]
    ev.sort()
    busy1 = busy2 = 0
    depth, last = 0, ev[0][0]
    for t, d in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        depth += d
        last = t
    print(f"# GPU busy with >= 1 kernel: {busy1 / 1e6:.2f} ms; with kernels of >= 2 launches at once: {busy2 / 1e6:.2f} ms "
          f"({100.0 * busy2 / max(1, busy1):.1f} %)")


if __name__ == "__main__":
    main()
