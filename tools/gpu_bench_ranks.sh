#!/bin/bash
# bench.py with R ranks sharing the single GPU of the box (RCCL refuses duplicate devices -> the
# agreed host-buffer fallback for the halo), with and without the peer-write all-reduce:
#   gpu_bench_ranks.sh R EDGE ITERS
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${1:-2}; EDGE=${2:-64}; ITERS=${3:-50}
mkdir -p gpurun_out
for PEER in 1 0; do
  OGL_BENCH_PEER=$PEER timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$R --master-addr 127.0.0.1 --master-port $((29520 + PEER)) bench.py --gpus $R --steps 3 --warmup 1 --edge $EDGE --iters $ITERS > gpurun_out/bench_${R}rank_peer$PEER.json 2> gpurun_out/bench_${R}rank_peer$PEER.err
  echo "peer=$PEER rc=$?"
  python - "$R" "$PEER" <<'PY'
import json,sys
try:
    d=json.load(open(f"gpurun_out/bench_{sys.argv[1]}rank_peer{sys.argv[2]}.json"))
    print("value %.1f it/s  ms/turn %.4f | %s" % (d["value"], d["cg_iteration"]["ms"], d["config"]["parallelism"]))
except Exception as e:
    print("no json:", e)
PY
  grep -v "amdgpu.ids" gpurun_out/bench_${R}rank_peer$PEER.err | grep -v "^$" | tail -4
done
