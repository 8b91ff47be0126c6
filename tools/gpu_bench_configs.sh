#!/bin/bash
# Single-GPU measurements of the other BASELINE.json configs' solver / preconditioner pairs, each against
# the byte model of one solver turn (bench.py turn_model).  Writes gpurun_out/${TAG}_configs.txt
TAG=${1:-r05}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_configs.txt
echo "# tools/gpu_bench_configs.sh on one MI355X (bench.py --steps 3 --warmup 2, HBM-resident, fixed turn count)" > $OUT
echo "# turn = one solver iteration (BiCGStab: two SpMVs); spmv = in-loop SpMV (HIP event pairs); frac = bytes the turn's kernels move (SpMV at its layout's bytes) / time / 8 TB/s; csr-eq = the same with the SpMV priced at SURVEY 8(d)'s CSR bytes" >> $OUT
run() {
  T=$1; shift
  python bench.py --steps 3 --warmup 2 --cpu-iters 0 --no-general-legs "$@" > gpurun_out/cfg_$T.json 2> gpurun_out/cfg_$T.err || { echo "$T FAILED" | tee -a $OUT; tail -3 gpurun_out/cfg_$T.err; return; }
  python - "$T" <<'PY' | tee -a $OUT
import json,sys
d=json.load(open(f"gpurun_out/cfg_{sys.argv[1]}.json"))
r=d["roofline"]; t=d["solver_turn"]
print("%-18s turns/s=%8.1f  ms/turn=%.4f  spmv_us=%6.1f (frac %.3f, csr-eq %.3f, %s)  turn bytes=%.3f GB -> %.0f GB/s = %.3f of 8 TB/s (csr-eq %.3f) | %s" % (
    sys.argv[1], d["value"], t["ms"], 1e3*r["avg_kernel_ms"], r["frac"], r["csr_equivalent_frac"], r["layout"], t["bytes"]/1e9,
    t["achieved_GBps"], t["frac_of_peak"], t["csr_equivalent_frac_of_peak"], d["config"]["workload"][:96]))
PY
}
run cg_bj_216        --iters 100
run cg_none_216      --iters 100 --precond none
run cg_bj4_216       --iters 100 --block-size 4
run cg_isai_216      --iters 100 --precond ISAI
run bicg_bj_128a     --iters 100 --solver GKOBiCGStab --asym --edge 128
run bicg_gisai_128a  --iters 100 --solver GKOBiCGStab --asym --edge 128 --precond GISAI
run bicg_bj_216a     --iters 50 --solver GKOBiCGStab --asym
run gmres30_bj_216   --iters 60 --solver GKOGMRES --krylov-dim 30
run gmres30_bj_216s  --iters 60 --solver GKOGMRES --krylov-dim 30 --shuffle 65536
run gmres30_bj_368   --iters 60 --solver GKOGMRES --krylov-dim 30 --edge 368
run cg_bj_368        --iters 50 --edge 368
# proxies of the unstructured configs (cells renumbered at random in windows of 65536; the backend renumbers itself)
run c3_cg_bj_128s     --iters 100 --edge 128 --shuffle 65536
run c3_bicg_isai_128s --iters 100 --edge 128 --shuffle 65536 --solver GKOBiCGStab --asym --precond ISAI
# ... block Jacobi through the backend's renumbering (the blocks stay the caller's: applied through the permutation)
run c3_cg_bj4_128s    --iters 100 --edge 128 --shuffle 65536 --block-size 4
run c3_cg_bj8_128s    --iters 100 --edge 128 --shuffle 65536 --block-size 8
# ... and on the kind of mesh pitzDaily is (three blockMesh blocks of 60 / 90 / 40 x 104 x 104 cells, 2.06 M, numbered block by block)
run c3_cg_bj_blocks3     --iters 100 --edge 104 --blocks 60,90,40
run c3_bicg_isai_blocks3 --iters 100 --edge 104 --blocks 60,90,40 --solver GKOBiCGStab --asym --precond ISAI
run c4_cg_bj_136      --iters 100 --edge 136
run c5_gmres_csr_184s --iters 60 --edge 184 --shuffle 65536 --solver GKOGMRES --krylov-dim 30
run c5_gmres_ell_184s --iters 60 --edge 184 --shuffle 65536 --solver GKOGMRES --krylov-dim 30 --format Ell
