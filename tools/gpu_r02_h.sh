#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_systems.py -m gpu -q -x -k "block or random" 2>&1 | tail -3
python tools/ab_bench.py tools/bin/libogl_amd_base.so 2 --block-size 4 2>&1 | tee gpurun_out/ab_bj4.txt
bash tools/gpu_bench_configs.sh r02
