#!/bin/bash
# A/B of the in-tree library against tools/bin/libogl_amd_base.so (default bench, then the shuffled one)
mkdir -p gpurun_out
python tools/ab_bench.py tools/bin/libogl_amd_base.so ${1:-2} 2>&1 | tee gpurun_out/ab_default.txt
python tools/ab_bench.py tools/bin/libogl_amd_base.so ${2:-1} --shuffle 65536 2>&1 | tee gpurun_out/ab_shuffle.txt
