#!/bin/bash
mkdir -p gpurun_out /tmp/cc
export OGL_CASE_CACHE_DIR=/tmp/cc
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sell.py tests/test_gpu_renumber.py tests/test_gpu_formats.py -m gpu -q -x 2>&1 | tail -4
python tools/ab_bench.py tools/bin/libogl_amd_base.so 2 --voronoi 3000000 2>&1 | tee gpurun_out/r02r_ab_voronoi.txt
python tools/ab_bench.py tools/bin/libogl_amd_base.so 2 --no-compress 2>&1 | tee gpurun_out/r02r_ab_nocompress.txt
python tools/ab_bench.py tools/bin/libogl_amd_base.so 1 --long-rows 0.03 --no-compress 2>&1 | tee gpurun_out/r02r_ab_long_nocompress.txt
