export OGL_CASE_CACHE_DIR=/tmp/cc HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p /tmp/cc gpurun_out
timeout 3300 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/r05_pytest.txt
bash tools/gpu_pass.sh r05 smoke bench:default prof:default pmc:default pmc:fullstorage pmc:nocompress pmc:shuffle65536 pmc:--no-compress+--prop+spmvBandRows=46656 pmc:--full-storage+--prop+spmvBandRows=0 configs small markers bench:vor3m bench:vor1m bench:--voronoi+1000000+--no-centres bench:--voronoi+3000000+--no-centres prof:vor1m pmc:vor1m pmc:vor3m bench:blocks2 bench:long bench:oct15 ranks:2:216 table
