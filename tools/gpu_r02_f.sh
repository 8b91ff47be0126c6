#!/bin/bash
mkdir -p gpurun_out
python tools/ab_bench.py tools/bin/libogl_amd_2c97ef6.so 3 2>&1 | tee gpurun_out/r02_ab_default.txt
python tools/ab_bench.py tools/bin/libogl_amd_2c97ef6.so 2 --shuffle 65536 2>&1 | tee gpurun_out/r02_ab_shuffle.txt
