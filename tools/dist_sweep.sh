#!/bin/bash
# sweep of random partitions over rank counts / solvers / switches (dist_worker.py, ranks share the box's GPU)
export OMP_NUM_THREADS=1 HSA_ENABLE_IPC_MODE_LEGACY=0
ok=0; bad=0; port=29600
for seed in $(seq 40 75); do
  n=$(( 2 + seed % 5 ))
  extra=""
  case $(( seed % 6 )) in
    1) extra="--asym 1";; 2) extra="--gmres 12";; 3) extra="--renumber 1";; 4) extra="--halo-fused 0";; 5) extra="--precond 0";;
  esac
  mode=gpu-peer; [ $(( seed % 7 )) = 0 ] && mode=gpu-host
  port=$(( port + 1 ))
  out=$(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$n --master-addr 127.0.0.1 --master-port $port tests/dist_worker.py --mode $mode --random $seed $extra 2>&1)
  c=$(echo "$out" | grep -o "gpu-[a-z]* ok" | wc -l)
  if [ "$c" = "$n" ]; then ok=$(( ok + 1 )); else bad=$(( bad + 1 )); echo "FAIL seed $seed ranks $n $mode $extra"; echo "$out" | grep -v "^W\|Gloo\|amdgpu.ids" | tail -8; fi
done
echo "dist sweep: $ok ok, $bad bad"
