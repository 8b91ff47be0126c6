#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs (one pass per counter set) into one JSON:
per kernel, the mean counter value over the dispatches that did real work (the solver enqueues a
few gated no-op launches after the stop; they are dropped by a 5 %-of-max threshold).

  pmc_summary.py OUT.json DIR [DIR ...]       (each DIR holds one pass)
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    vals = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    vals[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    res = collections.defaultdict(dict)
    for (k, c), v in vals.items():
        mx = max(v)
        real = [x for x in v if x > 0.05 * mx] if mx > 0 else v
        # "void ogl::(anonymous namespace)::k_name<0, 1>(int, ogl::(anonymous namespace)::Type, ...)" -> "k_name<0, 1>"
        m = re.search(r"namespace\)::(k_\w+(?:<[^>]*>)?)\(", k)
        short = m.group(1) if m else k.split("(")[0].split("::")[-1]
        res[short][c] = {"mean": sum(real) / len(real), "dispatches": len(real),
                         "dropped_noop_dispatches": len(v) - len(real)}
    # which kernels these counters belong to: bench.py refuses a summary collected with other kernels
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import hashlib
    h = hashlib.sha256()  # = bench.py kernels_sha16(): device_common.hpp + kernels_*.hip in name order
    csrc = os.path.join(root, "ogl_amd", "csrc")
    for f in [os.path.join(csrc, "device_common.hpp")] + sorted(glob.glob(os.path.join(csrc, "kernels_*.hip"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    sha = h.hexdigest()[:16]
    head = None
    for d in dirs:  # the bench line of the profiled run carries nothing about git: the pass script exports it
        head = head or os.environ.get("OGL_GIT_HEAD")
    res["_meta"] = {"kernels_sha16": sha, "head": head, "passes": [os.path.basename(d) for d in dirs]}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    for k in sorted(res):
        if k == "_meta":
            continue
        print(k, {c: round(x["mean"], 1) for c, x in res[k].items()})


if __name__ == "__main__":
    main()
