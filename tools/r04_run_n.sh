export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_systems.py tests/test_gpu_renumber.py -m gpu -q -x -k "jacobi or random or renumber or caching" 2>&1 | tail -4
for ST in 1 0 1 0; do
python bench.py --steps 3 --warmup 2 --cpu-iters 0 --no-general-legs --iters 50 --edge 128 --shuffle 65536 --block-size 4 --prop bjStagedApply=$ST > gpurun_out/r04n_bj4_st$ST.json 2> gpurun_out/r04n_bj4_st$ST.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04n_bj4_st$ST.json")); print("cg+BJ4 128^3 shuffled, caller's blocks, bjStagedApply=$ST", d["value"], "turns/s, turn us", 1e3*d["solver_turn"]["ms"])
PY
done 2>&1 | tee gpurun_out/r04n_bj_staged.txt
python bench.py --steps 3 --warmup 2 --cpu-iters 0 --no-general-legs --iters 50 --edge 128 --shuffle 65536 --block-size 8 --prop bjStagedApply=1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BJ8 staged turn us', 1e3*d['solver_turn']['ms'])" | tee -a gpurun_out/r04n_bj_staged.txt
python bench.py --steps 3 --warmup 2 --cpu-iters 0 --no-general-legs --iters 50 --edge 128 --shuffle 65536 --block-size 8 --prop bjStagedApply=0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BJ8 direct turn us', 1e3*d['solver_turn']['ms'])" | tee -a gpurun_out/r04n_bj_staged.txt
