#!/bin/bash
# Round-4 one-off measurements (the standing passes are tools/gpu_pass.sh).  Run on the GPU box:
#   gpurun -- 'bash tools/r04_experiments.sh STAGE [STAGE ...]'        results under gpurun_out/r04x_*
#   merged_ab            two ranks on one device, merged 4-launch turn against the 5-launch turn at 100^3 ... 136^3
#   staging              coefficient refresh / plug-in call with 4, 8, 16 staging threads and plain stores
#   win_tune             tools/win_tune.hip on the Voronoi proxies (library's RCM numbering)  -> profiles/r04_win_tune.txt
#   win_fetch            FETCH_SIZE per variant of the same harness (rocprofv3 --pmc, own pass)
#   precond_numbering    BJ(4) / ISAI on shuffled cells: structures of the caller's against the backend's numbering
#   bj_staged            block Jacobi through the permutation: staged against direct apply
#   eight_way            2 x 2 x 2 cuts with eight ranks on one device: where the peer mesh starves, host-buffer transport
export HSA_ENABLE_IPC_MODE_LEGACY=0 OGL_CASE_CACHE_DIR=/tmp/cc OMP_NUM_THREADS=1
mkdir -p /tmp/cc gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], '%.1f it/s' % d['value'], 'turn %.1f us' % (1e3*d['solver_turn']['ms']), d['roofline']['kernel'], 'boundary', {k: round(v, 4) for k, v in d.get('boundary', {}).items()})" "$1" "$2"; }
two_ranks() { OGL_BENCH_PEER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 2 --steps 3 --warmup 1 --iters 200 --cpu-iters 0 "$@"; }
for STAGE in "$@"; do case $STAGE in
  merged_ab)
    for E in 100 108 128 136; do for M in 1 0 1 0; do
      two_ranks --edge $E --prop fusedTurnMulti=$M > gpurun_out/r04x_ranks2_e${E}_m$M.json 2> gpurun_out/r04x_ranks2_e${E}_m$M.err
      line gpurun_out/r04x_ranks2_e${E}_m$M.json "2 x $E^3 merged=$M"
    done; done | tee gpurun_out/r04x_merged_ab.txt;;
  staging)
    for V in "OGL_STAGE_THREADS=4" "OGL_STAGE_THREADS=8" "OGL_STAGE_THREADS=16" "OGL_STAGE_PLAIN_STORES=1"; do
      env $V python bench.py --steps 5 --warmup 1 --cpu-iters 0 --no-general-legs > gpurun_out/r04x_stage.json 2> gpurun_out/r04x_stage.err
      line gpurun_out/r04x_stage.json "$V"
    done | tee gpurun_out/r04x_staging.txt;;
  win_tune)
    for N in 3000000 1000000; do
      python tools/dump_pattern.py voronoi $N /tmp/cc/vor$N.bin 2>&1 | tail -1
      tools/bin/win_tune /tmp/cc/vor$N.bin 50 2>&1 | tee gpurun_out/r04x_win_tune_vor$N.txt
    done;;
  win_fetch)
    python tools/dump_pattern.py voronoi 3000000 /tmp/cc/vor3000000.bin 2>&1 | tail -1
    ( cd /tmp && export TMPDIR=/tmp
      OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_r04x_win; rm -rf $OUT; mkdir -p $OUT
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -- $GRAFT_REPO_ROOT/tools/bin/win_tune /tmp/cc/vor3000000.bin 3 > $OUT/run.txt 2>&1
      python3 - "$(find $OUT -name '*counter_collection.csv' | head -1)" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r04x_win_fetch.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if r.get("Counter_Name") == "FETCH_SIZE":
            k = r["Kernel_Name"][:90]; acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{k:92s} launches {n:4d}  FETCH_SIZE x 2 (gfx950 correction) per launch {2 * v / n * 1024 / 1e6:9.1f} MB")
PY
      find $OUT -name '*.csv' -size +1M -delete );;
  precond_numbering)
    for CN in 1 0 1 0; do
      python bench.py --steps 3 --warmup 2 --cpu-iters 0 --no-general-legs --iters 50 --solver GKOBiCGStab --asym --edge 128 --shuffle 65536 --precond ISAI --prop precondCallerNumbering=$CN > gpurun_out/r04x_pn.json 2> gpurun_out/r04x_pn.err
      line gpurun_out/r04x_pn.json "bicg+ISAI 128^3 shuffled, precondCallerNumbering=$CN"
      python bench.py --steps 3 --warmup 2 --cpu-iters 0 --no-general-legs --iters 50 --edge 128 --shuffle 65536 --block-size 4 --prop precondCallerNumbering=$CN > gpurun_out/r04x_pn.json 2> gpurun_out/r04x_pn.err
      line gpurun_out/r04x_pn.json "cg+BJ4 128^3 shuffled, precondCallerNumbering=$CN"
    done | tee gpurun_out/r04x_precond_numbering.txt;;
  bj_staged)
    for K in 4 8; do for ST in 1 0 1 0; do
      python bench.py --steps 3 --warmup 2 --cpu-iters 0 --no-general-legs --iters 50 --edge 128 --shuffle 65536 --block-size $K --prop bjStagedApply=$ST > gpurun_out/r04x_bj.json 2> gpurun_out/r04x_bj.err
      line gpurun_out/r04x_bj.json "cg+BJ$K 128^3 shuffled, caller's blocks, bjStagedApply=$ST"
    done; done | tee gpurun_out/r04x_bj_staged.txt;;
  eight_way)
    run() { echo "== $*"; OGL_PEER_TIMEOUT_S=5 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29577 \
      tests/dist_worker.py "$@" 2>&1 | grep -E " ok|Mismatched|timed out" | sort | uniq -c | head -12; }
    { run --mode gpu-peer --shape 128,128,128 --procs 2,2,2 --max-iter 10
      run --mode gpu-peer --shape 192,192,192 --procs 2,2,2 --max-iter 10 --no-global 1
      run --mode gpu-host --shape 272,272,272 --procs 2,2,2 --max-iter 10 --no-global 1
      run --mode gpu-host --shape 368,368,368 --procs 2,2,2 --gmres 30 --max-iter 10 --no-global 1 --shuffle 65536 --renumber 1; } | tee gpurun_out/r04x_eight_way.txt;;
  *) echo "unknown stage $STAGE";;
esac; done
