#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sym.py tests/test_gpu_sell.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_formats.py tests/test_gpu_deferred_x.py -m gpu -q -x 2>&1 | tail -3
for E in 216 368 128 100 64; do python tools/ab_bench.py tools/bin/libogl_amd_base.so 2 --edge $E 2>&1 | sed "s/^/edge $E /"; done | tee gpurun_out/r02z_ab2.txt
python tools/ab_bench.py tools/bin/libogl_amd_base.so 1 --solver GKOGMRES --krylov-dim 30 --iters 60 2>&1 | sed "s/^/gmres30 /" | tee -a gpurun_out/r02z_ab2.txt
