export OGL_CASE_CACHE_DIR=/tmp/cc HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p /tmp/cc gpurun_out
timeout 900 python -m pytest tests/test_gpu_exact_arbiter.py -q -x -s 2>&1 | tail -8
timeout 1200 python tools/parity_deviation.py > gpurun_out/r05_parity_deviation.txt 2> gpurun_out/r05_parity_deviation.err; tail -5 gpurun_out/r05_parity_deviation.txt
bash tools/gpu_pass.sh r05c prof:--iters+100+--edge+128+--shuffle+65536+--solver+GKOBiCGStab+--asym+--precond+ISAI+--prop+streamAboveBytes=100000000
