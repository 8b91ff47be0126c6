export HSA_ENABLE_IPC_MODE_LEGACY=0
(time timeout 1800 python -m pytest tests -m gpu -q -x --durations=10 2>&1 | tail -40) > gpurun_out/r04c_pytest.txt 2>&1
tail -n 8 gpurun_out/r04c_pytest.txt
for T in 8 4 16; do OGL_STAGE_THREADS=$T python bench.py --steps 5 --warmup 1 --cpu-iters 0 --no-general-legs > gpurun_out/r04c_bench_threads$T.json 2> gpurun_out/r04c_bench_threads$T.err; python - <<PY
import json
d=json.load(open("gpurun_out/r04c_bench_threads$T.json")); print("threads $T", d["value"], d["boundary"])
PY
done
OGL_STAGE_PLAIN_STORES=1 python bench.py --steps 5 --warmup 1 --cpu-iters 0 --no-general-legs > gpurun_out/r04c_bench_plain.json 2> gpurun_out/r04c_bench_plain.err; python -c "
import json
d=json.load(open('gpurun_out/r04c_bench_plain.json')); print('plain stores', d['value'], d['boundary'])"
for E in 136 128; do for M in 1 0; do
OGL_BENCH_PEER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 3 --warmup 1 --edge $E --iters 100 --cpu-iters 0 --prop fusedTurnMulti=$M > gpurun_out/r04c_ranks2_e${E}_merged$M.json 2> gpurun_out/r04c_ranks2_e${E}_merged$M.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04c_ranks2_e${E}_merged$M.json")); print("2 x $E^3 merged=$M", d["value"], "turn us", 1e3*d["cg_iteration"]["ms"], d["roofline"]["kernel"], (d["config"]["selfcheck"] or {}).get("ok"))
PY
done; done
