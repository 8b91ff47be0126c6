export OGL_CASE_CACHE_DIR=/tmp/cc HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p /tmp/cc gpurun_out
timeout 900 python -m pytest tests/test_gpu_proxy_meshes.py -m gpu -q -x -k "far_entries" 2>&1 | tail -15
timeout 1500 python tools/fuzz_gpu.py 1200 51 2>&1 | tail -8 | tee gpurun_out/r05_fuzz.txt
OGL_FUZZ_SCALE=2 timeout 1500 python tools/fuzz_gpu.py 400 52 2>&1 | tail -8 | tee -a gpurun_out/r05_fuzz.txt
