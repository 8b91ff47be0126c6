#!/bin/bash
mkdir -p gpurun_out /tmp/cc; export OGL_CASE_CACHE_DIR=/tmp/cc
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sell.py tests/test_gpu_random_systems.py tests/test_gpu_formats.py tests/test_gpu_renumber.py -m gpu -q -x 2>&1 | tail -2
python tools/ab_bench.py tools/bin/libogl_amd_base.so 2 --voronoi 1000000 2>&1 | sed "s/^/vor1m /"
python tools/ab_bench.py tools/bin/libogl_amd_base.so 2 --solver GKOBiCGStab --asym --edge 128 2>&1 | sed "s/^/bicg asym 128 /"
python tools/ab_bench.py tools/bin/libogl_amd_base.so 2 --edge 128 --shuffle 65536 2>&1 | sed "s/^/shuffle 128 /"
python tools/ab_bench.py tools/bin/libogl_amd_base.so 1 --shuffle 65536 2>&1 | sed "s/^/shuffle 216 /"
python tools/ab_bench.py tools/bin/libogl_amd_base.so 1 --precond ISAI 2>&1 | sed "s/^/isai 216 /"
