#!/bin/bash
# One GPU evidence pass, parameterised (replaces the per-letter tools/gpu_r02_?.sh scripts of round 2).
#
#   tools/gpu_pass.sh TAG STAGE [STAGE ...]
#
# Everything lands under gpurun_out/ with the prefix TAG.  Stages run in the order given:
#   tests[:K]          pytest -m gpu (optionally -k K)                      -> TAG_pytest.txt
#   smoke              __graft_entry__.smoke()
#   bench:V            bench.py on variant V (see variant())                -> TAG_bench_V.json
#   prof:V             rocprofv3 --kernel-trace --stats of the same         -> prof_TAG_V/kernel_stats.csv
#   pmc:V              separate --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC)  -> pmc_TAG_V_summary.json
#   small | configs | markers      tools/gpu_bench_small.sh / gpu_bench_configs.sh / gpu_markers.sh
#   ranks:R:EDGE       bench.py with R ranks sharing the box's one GPU (peer mesh on / off)
#   ranktrace:R:EDGE   rocprofv3 kernel trace of R ranks sharing the GPU, merged time line -> TAG_ranktrace_R_eEDGE.txt
#   table              one line per TAG_bench_*.json                        -> TAG_table.txt
# Variants: default fullstorage nocompress shuffle512 shuffle4096 shuffle65536 shuffle65536off
#           shuffle65536nc drop dropnc long longnc oct15 oct15nc oct4 oct4append vor1m vor3m vor3moff vor3mnc
#           n64 n32 n100 n128 (+ anything else is passed to bench.py verbatim, '+' for spaces: --edge+100)
TAG=${1:?tag}; shift
mkdir -p gpurun_out /tmp/cc
export OGL_CASE_CACHE_DIR=/tmp/cc HSA_ENABLE_IPC_MODE_LEGACY=0
cd "${GRAFT_REPO_ROOT:-.}"

variant() {
  case $1 in
    default) echo "";;
    fullstorage) echo "--full-storage";;
    nocompress) echo "--no-compress";;
    shuffle512|shuffle4096|shuffle65536) echo "--shuffle ${1#shuffle}";;
    shuffle65536off) echo "--shuffle 65536 --renumber off";;
    shuffle65536nc) echo "--shuffle 65536 --no-compress";;
    drop) echo "--drop-faces 0.3";;       dropnc) echo "--drop-faces 0.3 --no-compress";;
    long) echo "--long-rows 0.03";;       longnc) echo "--long-rows 0.03 --no-compress";;
    oct15) echo "--octree 1.5";;          oct15nc) echo "--octree 1.5 --no-compress";;
    oct4) echo "--octree 4";;             oct4append) echo "--octree 4 --octree-append";;
    vor1m) echo "--voronoi 1000000";;     vor3m) echo "--voronoi 3000000";;
    vor3moff) echo "--voronoi 3000000 --renumber off";;
    vor3mg*) echo "--voronoi 3000000 --prop xcdGroup=${1#vor3mg}";;
    vor1mg*) echo "--voronoi 1000000 --prop xcdGroup=${1#vor1mg}";;
    shuffle65536g*) echo "--shuffle 65536 --prop xcdGroup=${1#shuffle65536g}";;
    nocompressg*) echo "--no-compress --prop xcdGroup=${1#nocompressg}";;
    hostsetup) echo "--prop deviceSetup=0";;
    blocks2) echo "--blocks 120,96";;      blocks2full) echo "--blocks 120,96 --full-storage";;
    blocks3) echo "--blocks 70,100,46";;   blocks3full) echo "--blocks 70,100,46 --full-storage";;
    oct15full) echo "--octree 1.5 --full-storage";;  oct15append) echo "--octree 1.5 --octree-append";;
    oct15appendfull) echo "--octree 1.5 --octree-append --full-storage";;
    vor3mnc) echo "--voronoi 3000000 --no-compress";;
    n32|n64|n100|n128) echo "--edge ${1#n} --iters 200";;
    *) echo "${1//+/ }";;
  esac
}

for STAGE in "$@"; do
  IFS=: read -r KIND A1 A2 <<< "$STAGE"
  case $KIND in
    tests)
      timeout 3300 python -m pytest tests -m gpu -q -x ${A1:+-k "$A1"} 2>&1 | tail -6 | tee gpurun_out/${TAG}_pytest.txt;;
    smoke)
      python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2;;
    bench)
      EXTRA="--cpu-iters 0"; [ "$A1" = default ] && EXTRA=""
      python bench.py --steps 5 --warmup 1 $EXTRA $(variant $A1) > gpurun_out/${TAG}_bench_$A1.json 2> gpurun_out/${TAG}_bench_$A1.err \
        || { echo "bench $A1 FAILED"; tail -5 gpurun_out/${TAG}_bench_$A1.err; };;
    prof)
      ( cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
        OUT=$PWD/gpurun_out/prof_${TAG}_$A1; rm -rf $OUT; mkdir -p $OUT
        rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --cpu-iters 0 --no-general-legs $(variant $A1) > $OUT/bench.json 2> $OUT/bench.err
        echo "prof $A1 rc=$?"
        F=$(find $OUT -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $OUT/kernel_stats.csv && head -8 $F | cut -c1-220
        find $OUT -name '*kernel_trace.csv' -size +20M -delete );;
    pmc)
      ( cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
        DIRS=""
        for CNT in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
          NAME=$(echo $CNT | tr ' ' '+'); OUT=$PWD/gpurun_out/pmc_${TAG}_${A1}_$NAME; rm -rf $OUT; mkdir -p $OUT
          rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 0 --iters 20 --cpu-iters 0 --no-general-legs $(variant $A1) > $OUT/bench.json 2> $OUT/bench.err
          echo "pmc $A1 $NAME rc=$?"; DIRS="$DIRS $OUT"
        done
        python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_${A1}_summary.json $DIRS | grep k_spmv
        find gpurun_out -name '*counter_collection.csv' -size +1M -delete
        find gpurun_out -name '*kernel_trace.csv' -size +1M -delete );;
    small) bash tools/gpu_bench_small.sh 2>&1 | tee gpurun_out/${TAG}_small.txt;;
    small5) SMALL_EXTRA="--prop fusedFinalizers=0" bash tools/gpu_bench_small.sh 2>&1 | tee gpurun_out/${TAG}_small_five_launch.txt;;
    configs) bash tools/gpu_bench_configs.sh $TAG;;
    markers) bash tools/gpu_markers.sh > gpurun_out/markers_$TAG.log 2>&1;;
    ranks)
      for PEER in 1 0; do
        OGL_BENCH_PEER=$PEER timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$A1 --master-addr 127.0.0.1 \
          --master-port $((29520 + PEER)) bench.py --gpus $A1 --steps 3 --warmup 1 --edge ${A2:-128} --iters 100 --cpu-iters 0 \
          > gpurun_out/${TAG}_ranks${A1}_e${A2:-128}_peer$PEER.json 2> gpurun_out/${TAG}_ranks${A1}_e${A2:-128}_peer$PEER.err
        echo "ranks $A1 edge ${A2:-128} peer=$PEER rc=$?"
      done;;
    ranktrace)
      # kernel trace of R ranks sharing the GPU (each rank under its own rocprofv3: the profiler starts python
      # directly, nothing re-execs after the GPU was touched), merged into one time line
      ( cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
        OUT=$PWD/gpurun_out/ranktrace_${TAG}_${A1}_e${A2:-128}; rm -rf $OUT; mkdir -p $OUT
        timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$A1 --master-addr 127.0.0.1 --master-port 29533 \
          --no-python rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --gpus $A1 --steps 2 --warmup 1 \
          --edge ${A2:-128} --iters 40 --cpu-iters 0 --no-selfcheck > $OUT/bench.json 2> $OUT/bench.err
        echo "ranktrace $A1 rc=$?"
        python3 tools/ranks_trace.py $OUT > gpurun_out/${TAG}_ranktrace_${A1}_e${A2:-128}.txt 2>&1
        head -50 gpurun_out/${TAG}_ranktrace_${A1}_e${A2:-128}.txt | cut -c1-200
        find $OUT -name '*kernel_trace.csv' -size +8M -delete );;
    table)
      python - $TAG <<'PY' | tee gpurun_out/${TAG}_table.txt
import json, glob, sys
tag = sys.argv[1]
def line(name, d, r):
    print("%-24s %8.1f it/s  %-44s %7.1f us  frac %.3f (%.0f MB)  csr-equivalent %.3f  traffic/model %s  renumbered=%s first set_matrix %.2f s" % (
        name, d.get("value", r.get("cg_iters_per_sec", 0.0)), r["kernel"], 1e3 * r["avg_kernel_ms"], r["frac"], r["bytes_per_launch"] / 1e6,
        r["csr_equivalent_frac"], "-" if r.get("traffic_over_model") is None else "%.2f" % r["traffic_over_model"],
        d["config"]["renumbered"] if "config" in d else r.get("renumbered"), d["boundary"]["first_set_matrix_s"] if "boundary" in d else r["first_set_matrix_s"]))
for f in sorted(glob.glob(f"gpurun_out/{tag}_bench_*.json")) + sorted(glob.glob(f"gpurun_out/{tag}_ranks*.json")):
    name = f.split("/")[-1][len(tag) + 1:-5]
    try:
        d = json.load(open(f))
    except Exception as e:
        print(name, "unreadable", e); continue
    line(name, d, d["roofline"])
    for g in d.get("roofline_general") or []:
        line("  leg " + g["leg"], {"value": g["cg_iters_per_sec"]}, g)
    if "cpu_baseline" in d:
        print("    cpu_baseline", d["cpu_baseline"]); print("    cpu_baseline_omp", {k: v for k, v in d["cpu_baseline_omp"].items() if k != "placement_probe"})
    if d.get("n_gpus", 1) > 1:
        print("    ", d["config"]["parallelism"], "| turn", "%.1f us" % (1e3 * d["cg_iteration"]["ms"]), "| selfcheck", (d["config"]["selfcheck"] or {}).get("ok"))
PY
      ;;
    *) echo "unknown stage $STAGE";;
  esac
done
