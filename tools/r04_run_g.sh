export HSA_ENABLE_IPC_MODE_LEGACY=0 OGL_CASE_CACHE_DIR=/tmp/cc
mkdir -p /tmp/cc gpurun_out
python tools/dump_pattern.py voronoi 3000000 /tmp/cc/vor3m.bin 2>&1 | tail -1
tools/bin/win_tune /tmp/cc/vor3m.bin 50 2>&1 | tee gpurun_out/r04g_win_tune_vor3m.txt
