#!/usr/bin/env python3
"""bench.py's cpu_baseline leg, run as a CHILD process of rank 0 (N = 1 only):

  python -m oracle.cpu_baseline --edge 216 --precond BJ --iters 100 --seconds 15

The oracle ("port" of the reference's CPU path: Ginkgo's reference / omp executors cannot be built
here) on the same synthetic system as the GPU run: the sequential restatement on one
core and its OpenMP variant on all cores.  A child process because thread placement has to be decided
before an OpenMP runtime is loaded (OMP_PROC_BIND / OMP_PLACES are read once; the parent has torch's
libgomp in it already) and so that nothing here shares an address space with the GPU run.  Prints one
JSON object.
"""
import argparse
import json
import os
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edge", type=int, default=216)
    ap.add_argument("--precond", default="BJ", choices=["BJ", "none"])
    ap.add_argument("--iters", type=int, default=100, help="upper bound of CG iterations per leg")
    ap.add_argument("--seq-iters", type=int, default=-1, help="-1: sized for about --seconds of work")
    ap.add_argument("--seconds", type=float, default=15.0)
    ap.add_argument("--bind", type=int, default=1,
                    help="1: OMP_PROC_BIND=close OMP_PLACES=cores (set before libgomp loads); 2: spread; 0: leave "
                         "the threads to the scheduler")
    ap.add_argument("--probe", action="store_true",
                    help="only the thread-count probe: STREAM-like triad over a ladder of thread counts "
                         "under this binding; prints {threads, GBps} of the best")
    ap.add_argument("--threads", type=int, default=0, help="OpenMP threads of the omp leg (0: probe)")
    ap.add_argument("--omp-only", action="store_true", help="skip the sequential leg (placement trials)")
    args = ap.parse_args()
    cores = len(os.sched_getaffinity(0))
    if args.bind:
        # (2: one thread per core, spread over the sockets and core complexes -- what a thread count below the core
        #  count wants: every thread brings its own L3 slice and memory channel share)
        os.environ["OMP_PROC_BIND"] = "spread" if args.bind == 2 else "close"
        os.environ["OMP_PLACES"] = "cores"
    else:
        os.environ.pop("OMP_PROC_BIND", None)
        os.environ.pop("OMP_PLACES", None)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")     # a CPU quota below the CPU count must not spin

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import numpy as np
    from ogl_amd import synthetic          # numpy-only input generator (no device code)
    from oracle import oracle as orc
    orc.build()
    omp_build = orc.use_native_omp_build()     # the OpenMP legs (probe, CG) may use this host's vector units

    def probe():
        """The thread count that moves the most bytes here: affinity masks say nothing about cgroup CPU
        quotas, SMT siblings or how many cores it takes to saturate the memory controllers."""
        ladder = sorted({t for t in (4, 8, 16, 32, 48, 64, 96, 128, 192, 256, cores) if t <= cores})
        best = (0.0, 1)
        for t in ladder:
            gbps = orc.stream_triad_omp(1 << 26, 2, t)          # 3 x 512 MiB: past any L3
            best = max(best, (gbps, t))
            if gbps < 0.5 * best[0]:
                break                                           # far past the knee: stop paying for it
        return best

    if args.probe:
        gbps, t = probe()
        print(json.dumps({"threads": t, "GBps": gbps, "bind": args.bind, "host_cpus": cores, "build": omp_build}))
        return

    case = synthetic.poisson_case(args.edge)
    b, _ = synthetic.rhs_for_x_star(case)
    n, nnz = case.n_cells, case.nnz
    t0 = time.perf_counter()
    rows, cols, perm = orc.init_local_sparsity(n, case.upper_addr, case.lower_addr, True)
    vals = orc.update_local_matrix_data(case.diag, case.upper, None, [], perm)
    rowptr = orc.rowptr_from_rows(n, rows)
    A = orc.DistMatrix(rowptr, cols, vals)
    inv = orc.jacobi_generate_scalar(rowptr, cols, vals) if args.precond == "BJ" else None
    t_build = time.perf_counter() - t0
    b_cg = 12 * nnz + 20 * n + 4 + (88 if inv is not None else 72) * n

    out = {}
    seq_iters = 0 if args.omp_only else args.seq_iters
    if seq_iters < 0:                      # probe with 2 iterations
        t0 = time.perf_counter()
        orc.cg(A, b, np.zeros_like(b), inv, tolerance=0.0, rel_tol=0.0, max_iter=2, export_res=False)
        per_it = (time.perf_counter() - t0) / 3.0
        seq_iters = int(max(3, min(args.iters, args.seconds / max(per_it, 1e-6))))
    if not args.omp_only:
        t0 = time.perf_counter()
        r = orc.cg(A, b, np.zeros_like(b), inv, tolerance=0.0, rel_tol=0.0, max_iter=seq_iters, export_res=False)
        t_seq = time.perf_counter() - t0
        done = r.n_iterations - 1
        out["seq"] = {"value": done / t_seq, "unit": "iter/s", "cores": 1, "kind": "port",
                      "GBps": b_cg * done / t_seq / 1e9,
                      "sample": f"{done} CG iterations of the same {args.edge}^3 system, oracle (sequential "
                                f"reference-executor restatement), {t_seq:.1f} s (+{t_build:.1f} s LDU->CSR)"}

    if args.threads > 0:
        threads, triad = args.threads, None
    else:
        triad, threads = probe()
    # The triad names the thread count that streams best; CG also gathers (latency-bound), where SMT
    # siblings can hurt, and the hosts of this pool are shared (short trials mislead).  So the whole leg
    # is run for that count and for half of it, each for about half of --seconds, and the faster is kept.
    runs = {}
    for t in sorted({threads, max(1, threads // 2)}, reverse=True):
        _, _, t3 = orc.cg_omp_timed(A, b, np.zeros_like(b), inv, max_iter=3, threads=t)
        its = int(max(3, min(args.iters, 0.5 * args.seconds / max(t3 / 4.0, 1e-6))))
        runs[t] = orc.cg_omp_timed(A, b, np.zeros_like(b), inv, max_iter=its, threads=t) + (orc.cg_omp_phases(),)
    threads = max(runs, key=lambda t: (runs[t][0].n_iterations - 1) / runs[t][2])
    res, t_setup, t_loop, phases = runs[threads]
    trials = {t: (r[0].n_iterations - 1) / r[2] for t, r in runs.items()}
    if triad is None:
        triad = orc.stream_triad_omp(1 << 26, 3, threads)
    done = res.n_iterations - 1
    out["omp"] = {"value": done / t_loop, "unit": "iter/s", "cores": threads, "kind": "port",
                  "GBps": b_cg * done / t_loop / 1e9, "stream_triad_GBps": triad,
                  "host_cpus": cores, "build": omp_build,
                  "iters_per_s_by_thread_count": {str(k): v for k, v in trials.items()},
                  # where the loop's time goes, with the bytes each pass streams (scalar Jacobi: 56 N, 32 N, 12 nnz + 28 N)
                  "passes": {name: {"s_per_iter": sec / max(done, 1), "GBps": nbytes * done / max(sec, 1e-12) / 1e9}
                             for name, sec, nbytes in (("x_r_update_rho_norm", phases[0], 56 * n), ("p_update", phases[1], 32 * n),
                                                       ("spmv_pq", phases[2], 12 * nnz + 28 * n))},
                  "thread_binding": (f"OMP_PROC_BIND={os.environ['OMP_PROC_BIND']} OMP_PLACES="
                                     f"{os.environ['OMP_PLACES']} (set before libgomp loads)") if args.bind
                  else "none (scheduler)",
                  "sample": f"{done} iterations, OpenMP variant on {threads} threads (the faster of the count asked for and its "
                            f"half; {cores} CPUs in the affinity mask), loop {t_loop:.2f} s "
                            f"(first-touch copy of the matrix {t_setup:.2f} s, not counted)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
