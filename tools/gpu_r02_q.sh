#!/bin/bash
mkdir -p gpurun_out /tmp/cc
export OGL_CASE_CACHE_DIR=/tmp/cc
python tools/dump_pattern.py voronoi 3000000 /tmp/cc/vor.bin
timeout 600 tools/bin/csr_tune /tmp/cc/vor.bin 50 2>&1 | tee gpurun_out/r02q_csr_tune_vor3m.txt
python tools/dump_pattern.py poisson 160 /tmp/cc/box160.bin 0
timeout 600 tools/bin/csr_tune /tmp/cc/box160.bin 50 2>&1 | tee gpurun_out/r02q_csr_tune_box160.txt
