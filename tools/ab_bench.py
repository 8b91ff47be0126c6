#!/usr/bin/env python3
"""A/B of two builds of libogl_amd.so on one box: runs bench.py alternately with the in-tree library
and with another one (e.g. tools/bin/libogl_amd_<commit>.so built from an earlier commit) and prints
the in-loop SpMV time and the turn rate of every run.  Development tool.

  python tools/ab_bench.py tools/bin/libogl_amd_2c97ef6.so [rounds] [bench.py flags ...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
flags = sys.argv[3:]
runner = ("import sys, runpy; sys.path.insert(0, %r); from ogl_amd import capi; capi.LIB_PATH = sys.argv[1]; "
          "sys.argv = ['bench.py'] + sys.argv[2:]; runpy.run_path(%r, run_name='__main__')"
          % (ROOT, os.path.join(ROOT, "bench.py")))
libs = [("in-tree", os.path.join(ROOT, "ogl_amd", "lib", "libogl_amd.so")), (os.path.basename(other), other)]
for r in range(rounds):
    for name, lib in libs:
        p = subprocess.run([sys.executable, "-c", runner, lib, "--steps", "3", "--warmup", "1", "--cpu-iters", "0", "--no-general-legs", *flags],
                           cwd=ROOT, capture_output=True, text=True)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
            print("round %d %-28s %8.1f it/s  spmv %6.1f us  layout %s" % (
                r, name, d["value"], 1e3 * d["roofline"]["avg_kernel_ms"], d["roofline"]["layout"]), flush=True)
        except Exception as e:
            print("round %d %-28s FAILED %s\n%s" % (r, name, e, p.stderr[-800:]), flush=True)
