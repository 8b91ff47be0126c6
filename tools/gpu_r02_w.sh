#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sell.py tests/test_gpu_fullsize.py tests/test_gpu_formats.py -m gpu -q -x 2>&1 | tail -3
for E in 216 368 128 100 64; do
python tools/ab_bench.py tools/bin/libogl_amd_base.so 2 --edge $E 2>&1 | sed "s/^/edge $E /"
done | tee gpurun_out/r02w_ab.txt
python tools/ab_bench.py tools/bin/libogl_amd_base.so 1 --solver GKOBiCGStab --asym 2>&1 | sed "s/^/bicg asym 216 /" | tee -a gpurun_out/r02w_ab.txt
python tools/ab_bench.py tools/bin/libogl_amd_base.so 1 --shuffle 65536 2>&1 | sed "s/^/shuffle /" | tee -a gpurun_out/r02w_ab.txt
