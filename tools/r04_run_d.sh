# round 4, GPU pass d: window-kernel harness on the Voronoi proxies + merged multi-rank turn at cache-resident sizes
export HSA_ENABLE_IPC_MODE_LEGACY=0 OGL_CASE_CACHE_DIR=/tmp/cc
mkdir -p /tmp/cc gpurun_out
for E in 100 108; do for M in 1 0 1 0; do
OGL_BENCH_PEER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 3 --warmup 1 --edge $E --iters 200 --cpu-iters 0 --prop fusedTurnMulti=$M > gpurun_out/r04d_ranks2_e${E}_merged$M.json 2> gpurun_out/r04d_ranks2_e${E}_merged$M.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04d_ranks2_e${E}_merged$M.json")); print("2 x $E^3 merged=$M", d["value"], "turn us", 1e3*d["cg_iteration"]["ms"], d["roofline"]["kernel"], (d["config"]["selfcheck"] or {}).get("ok"))
PY
done; done 2>&1 | tee gpurun_out/r04d_merged_ab.txt
python tools/dump_pattern.py voronoi 1000000 /tmp/cc/vor1m.bin 2>&1 | tail -1
tools/bin/win_tune /tmp/cc/vor1m.bin 50 2>&1 | tee gpurun_out/r04d_win_tune_vor1m.txt
python tools/dump_pattern.py voronoi 3000000 /tmp/cc/vor3m.bin 2>&1 | tail -1
tools/bin/win_tune /tmp/cc/vor3m.bin 50 2>&1 | tee gpurun_out/r04d_win_tune_vor3m.txt
