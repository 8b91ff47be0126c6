#!/bin/bash
# round 2 final evidence pass: full GPU suite, then tools/gpu_r02_g.sh (benches, configs, profiles, PMC)
mkdir -p gpurun_out /tmp/cc
export OGL_CASE_CACHE_DIR=/tmp/cc
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/r02_final_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/gpu_r02_g.sh
