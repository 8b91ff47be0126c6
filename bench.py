#!/usr/bin/env python3
"""bench.py -- the hot path on synthetic input, one JSON line on rank 0.

Workload (BASELINE.json configs[1]): 7-point Poisson lduMatrix on a 216^3 box per GPU
(10,077,696 rows / 70,263,936 nnz), GKOCG + block-Jacobi (maxBlockSize 1), fp64 values + int32
indices, persistent device CSR.  One "step" = one solver->apply(b, x) (what the reference times
as delta_t_solve, lduLduBase.H:275-276) of a fixed number of CG iterations from x0 = 0, with the
matrix, b and x already resident in HBM.

  value     = CG iterations per second; at N GPUs the global box is 216 x 216 x (216 N), cut into
              N z-slabs (weak scaling: 10M rows per GPU, halo exchange + scalar all-reduces over
              RCCL), and value counts 10M-row block iterations: N * iterations / time.
  roofline  = the in-loop SpMV: the bytes the kernel has to move for the layout it runs on (matrix data of
              that layout + x read once + y written; for the plain CSR-stream kernel that is SURVEY.md
              §8d's 12 nnz + 20 N + 4) over the kernel's mean duration, measured with HIP events on the
              solver's stream inside the timed steps (profile_kernels), against 8 TB/s: `achieved`, `frac`
              (always <= 1).  The default layouts move fewer bytes than a CSR would (half storage of a
              symmetric matrix; index-compressed copy): the rate on SURVEY §8d's CSR bytes over the same
              time is reported apart as `csr_equivalent_achieved` / `csr_equivalent_frac` (may exceed 1: it
              is a unit-of-work rate, not a bandwidth).
              `traffic` = HBM-side bytes per launch from the PMC counters (2*FETCH_SIZE + WRITE_SIZE, KiB; the
              gfx950 correction of MI355X_MICROARCH.md), collected in THIS run at N = 1: after the timed region
              two short child runs of this command under `rocprofv3 --kernel-trace --pmc <counter>` (separate
              passes, children of this process, nothing exec'ed in place); when that is not possible (no
              rocprofv3, this process itself profiled, a pass fails) the committed summary of exactly these
              kernels (profiles/r*_pmc*_summary.json, refused when the kernel sources have changed) is quoted instead
              and `traffic_measured_in_this_run` says so.
  roofline_general = the same measurement for the general layouts on the same system, after the headline
              and outside its timed region: --full-storage (k_spmv_sell, pattern codes), --no-compress
              (k_spmv_stream, the kernel north_star describes), --shuffle 65536 (irregular numbering: the
              library renumbers itself, k_spmv_sell with 16-bit delta codes or k_spmv_stream).
  cpu_baseline = the oracle (sequential restatement, 1 core, "port") on the same matrix for a
              bounded number of iterations; plus its OpenMP variant as `cpu_baseline_omp`.

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29500 bench.py --gpus 8 --steps 5 --warmup 1
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBPS = 6290.0        # measured float4-copy ceiling, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    # (--edge, not --n: torch.distributed.run abbreviates its own options and would swallow --n)
    ap.add_argument("--edge", "--n", dest="n", type=int, default=216,
                    help="box edge per GPU (216 = configs[1])")
    ap.add_argument("--iters", type=int, default=100, help="CG iterations per step")
    ap.add_argument("--precond", default="BJ", choices=["BJ", "none", "ISAI", "GISAI"])
    ap.add_argument("--solver", default="GKOCG", choices=["GKOCG", "GKOBiCGStab", "GKOGMRES"],
                    help="GKOCG = the headline; the others measure configs[2]/[4]-style runs")
    ap.add_argument("--block-size", type=int, default=1, help="BJ maxBlockSize")
    ap.add_argument("--krylov-dim", type=int, default=30, help="GKOGMRES restart length")
    ap.add_argument("--asym", action="store_true", help="non-symmetric coefficients (momentum-like)")
    ap.add_argument("--format", default="Csr", choices=["Csr", "Ell"],
                    help="matrixFormat (Csr = the headline; Ell = configs[4]'s comparison)")
    ap.add_argument("--force-compress", action="store_true",
                    help="compressIndices force: the compressed layout without the one-off timing against the "
                         "CSR-stream kernel that decides for irregular patterns")
    ap.add_argument("--full-storage", action="store_true",
                    help="symmetricStorage false: a symmetric matrix is expanded to full storage on the device "
                         "(the compressed layout of round 1/2) instead of keeping diagonal + upper planes")
    ap.add_argument("--no-compress", action="store_true",
                    help="compressIndices false: SpMV on the plain CSR arrays (CSR-stream kernel)")
    ap.add_argument("--shuffle", type=int, default=0,
                    help="renumber the cells at random inside windows of this many cells (a stand-in for "
                         "an unstructured mesh: the compressed layouts do not qualify, the CSR-stream "
                         "kernel runs); single rank only")
    ap.add_argument("--drop-faces", type=float, default=0.0,
                    help="remove this fraction of the internal faces at random (row lengths 1..7: a stand-in "
                         "for a mesh of mixed cell types); single rank only")
    ap.add_argument("--long-rows", type=float, default=0.0,
                    help="give this fraction of the cells five extra couplings (rows of 12 entries among rows "
                         "of 7: a stand-in for a hex-dominant mesh); single rank only")
    ap.add_argument("--voronoi", type=int, default=0,
                    help="instead of the box: the Voronoi cells of this many random points (a polyhedral mesh, "
                         "15.5 faces per cell on average, random numbering); single rank only; scipy Delaunay "
                         "takes about a minute per million points")
    ap.add_argument("--centres", action="store_true",
                    help="box cases: hand the cell centres over too (mesh.C(); with --shuffle the Hilbert order through them "
                         "then competes with reverse Cuthill-McKee for the library's own numbering); single rank only")
    ap.add_argument("--no-centres", action="store_true",
                    help="with --voronoi: do not hand the cell centres over (reverse Cuthill-McKee is then the only "
                         "candidate for the library's own numbering; with them the Hilbert order through the centres competes)")
    ap.add_argument("--blocks", default="",
                    help="instead of the plain box: a multi-block structured mesh -- comma-separated block lengths along "
                         "x (e.g. 120,96), each block n x n in y and z, numbered block after block as blockMesh does; "
                         "single rank only")
    ap.add_argument("--octree", type=float, default=0.0,
                    help="instead of the plain box: an octree (snappyHexMesh-like, hex-dominant) mesh -- the cells "
                         "of the n^3 box within this many cells of a sphere are split 2x2x2, their unsplit "
                         "neighbours get rows of 10..16 entries; single rank only")
    ap.add_argument("--octree-append", action="store_true",
                    help="with --octree: children 1..7 of a split cell are numbered at the end of the cell list "
                         "(what splitting in place leaves behind) instead of next to their parent")
    ap.add_argument("--rcm", action="store_true",
                    help="after --shuffle: renumber the CASE with reverse Cuthill-McKee (scipy), as "
                         "renumberMesh would, before the library sees it")
    ap.add_argument("--renumber", default="auto", choices=["auto", "on", "off"],
                    help="keyword `renumber`: the library's own RCM numbering of its device copy "
                         "(auto = only when the numbering it is handed gathers x badly)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay batches of GKOCG turns as a hipGraph (needs --no-profile: event-timed "
                         "SpMVs cannot be captured).  auto = the library's default: on for the 3-launch "
                         "turn of systems of <= 1024 chunks, where it takes 2 us off a turn; off above, "
                         "where it measured no faster than stream launches")
    ap.add_argument("--cpu-iters", type=int, default=-1,
                    help="oracle iterations for cpu_baseline (-1: sized for ~15 s, 0: skip)")
    ap.add_argument("--profile-stride", type=int, default=4,
                    help="event-time the in-loop SpMV of every k-th turn (an event pair costs ~2 us of "
                         "the stream's time)")
    ap.add_argument("--no-selfcheck", dest="selfcheck", action="store_false",
                    help="N > 1: skip the cross-rank self-check that runs before anything is timed")
    ap.add_argument("--selfcheck-edge", type=int, default=64, help="box edge per rank of the self-check")
    ap.add_argument("--prop", action="append", default=[], metavar="KEY=VALUE",
                    help="set a solver property before set_matrix (development switches: xcdGroup, streamAboveBytes, "
                         "deviceSetup, ...); repeatable")
    ap.add_argument("--no-general-legs", dest="general_legs", action="store_false",
                    help="skip the roofline_general legs (full storage, CSR-stream, shuffled cells) that follow the "
                         "headline measurement of the default run")
    ap.add_argument("--general-steps", type=int, default=5, help="timed steps of each roofline_general leg")
    ap.add_argument("--live-pmc", default="auto", choices=["auto", "on", "off"],
                    help="roofline.traffic measured in THIS run: after the timed region two short child runs of this "
                         "command under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, ~25 s each). "
                         "auto: the default headline run at N = 1 when rocprofv3 is there and this process is not "
                         "itself profiled; otherwise (and whenever a pass fails) the committed summary of the same "
                         "kernels is quoted")
    ap.add_argument("--config", type=int, default=0, choices=[0, 3, 4],
                    help="the 8-way configurations as BASELINE.json states them, STRONG scaling (the global system is "
                         "fixed, `value` = global solver turns per second): 3 = channel-flow proxy of 20 M cells "
                         "(272^3 box) cut into N blocks (8: 2 x 2 x 2), GKOCG + BJ; 4 = 50 M-cell unstructured proxy "
                         "(368^3 box, every rank's cells shuffled in windows of 65536) cut the same way, "
                         "GKOGMRES(30) + BJ, --format Csr | Ell.  0 = the headline workload (216^3 per GPU, weak scaling)")
    ap.add_argument("--no-rung-probes", dest="rung_probes", action="store_false",
                    help="N > 1: do not collect evidence for the transport rungs the timed run did not use (by default "
                         "every rung of the ladder -- peer mesh over RCCL, peer mesh over the host bootstrap, RCCL alone, "
                         "host buffers alone -- is brought up once in child processes after the timed region, self-checked "
                         "and timed for a few steps: config.transport.rungs)")
    ap.add_argument("--rung-timeout", type=float, default=120.0,
                    help="seconds a rung probe may take before its child processes are ended")
    ap.add_argument("--no-profile", action="store_true",
                    help="do not event-time the in-loop SpMV (roofline then comes from a "
                         "separate back-to-back SpMV loop)")
    return ap.parse_args()


def kernels_sha16():
    """Hash of the Krylov-loop kernel sources (csrc/device_common.hpp + kernels_*.hip, in name order): the tag
    tools/pmc_summary.py writes into a PMC summary."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "ogl_amd", "csrc")
    for f in [os.path.join(csrc, "device_common.hpp")] + sorted(glob.glob(os.path.join(csrc, "kernels_*.hip"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, variant=""):
    """HBM-side bytes per launch of exactly the instantiation `kernel` (what the library says it launched)
    from the committed PMC passes of this same command (tools/gpu_pmc.sh: FETCH_SIZE and WRITE_SIZE in
    separate rocprofv3 --pmc runs).  Units and the gfx950 correction per MI355X_MICROARCH.md "HBM":
    counters are KiB, FETCH_SIZE reads half the bytes of a wide coalesced stream (calibrated here on
    k_cg_step1: 2 x 118,111 KiB = 241.9 MB measured vs 24 N = 241.9 MB algorithmic).  The summary records
    the hash of the kernel sources it was collected with; a summary of other kernels yields None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc{variant}_summary.json")))
    if not files:
        return None, "no committed PMC summary for this command"
    with open(files[-1]) as fh:
        d = json.load(fh)
    rel = os.path.relpath(files[-1], ROOT)
    meta = d.get("_meta", {})
    if meta.get("kernels_sha16") != kernels_sha16():
        return None, f"{rel} was collected with other kernels (kernel sources {meta.get('kernels_sha16')}): stale"
    k = d.get(kernel)
    if not k or "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
        return None, f"{rel} holds no counters for {kernel}"
    total = (2.0 * k["FETCH_SIZE"]["mean"] + k["WRITE_SIZE"]["mean"]) * 1024.0
    return total, rel + " (2*FETCH_SIZE + WRITE_SIZE) KiB, separate passes, head " + str(meta.get("head"))


_LIVE_PMC_STATE = {"broken": None}


def live_pmc_traffic(kernel, argv_tail):
    """HBM-side bytes per launch of `kernel`, collected now: one child run of this script per counter under
    rocprofv3 (--pmc passes are separate, as MI355X_MICROARCH.md prescribes; --kernel-trace only, no other trace
    domain).  The children are started as ordinary child processes of this one (nothing is exec'ed in place), from
    /tmp.  Returns (bytes or None, how)."""
    import csv
    import glob
    import re
    import shutil
    import tempfile
    if _LIVE_PMC_STATE["broken"]:    # (one failed pass: the rest of the run quotes the committed summaries)
        return None, "skipped: " + _LIVE_PMC_STATE["broken"]
    exe = shutil.which("rocprofv3")
    if exe is None:
        _LIVE_PMC_STATE["broken"] = "rocprofv3 not found"
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="ogl_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", OGL_BENCH_CHILD="1")
    means = {}
    try:
        for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, cnt)
            cmd = [exe, "--kernel-trace", "--pmc", cnt, "--output-format", "csv", "-d", out, "--", "python3",
                   os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--iters", "20", "--cpu-iters", "0",
                   "--no-general-legs", "--live-pmc", "off"] + argv_tail
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=120)
            if p.returncode != 0:
                _LIVE_PMC_STATE["broken"] = f"rocprofv3 --pmc {cnt} pass failed (rc {p.returncode}): {p.stderr[-300:]}"
                return None, _LIVE_PMC_STATE["broken"]
            vals = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        m = re.search(r"namespace\)::(k_\w+(?:<[^>]*>)?)\(", row["Kernel_Name"])
                        if m and m.group(1) == kernel and row["Counter_Name"] == cnt:
                            vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, f"the --pmc {cnt} pass saw no dispatch of {kernel}"
            real = [v for v in vals if v > 0.05 * max(vals)]       # (gated no-op launches after the stop)
            means[cnt] = (sum(real) / len(real), len(real))
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        _LIVE_PMC_STATE["broken"] = f"live PMC passes failed: {e!r}"
        return None, _LIVE_PMC_STATE["broken"]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = (2.0 * means["FETCH_SIZE"][0] + means["WRITE_SIZE"][0]) * 1024.0
    return total, (f"measured in this run: two child runs of this command (--steps 1 --iters 20) under rocprofv3 "
                   f"--kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean over "
                   f"{means['FETCH_SIZE'][1]} / {means['WRITE_SIZE'][1]} working launches, (2*FETCH_SIZE + WRITE_SIZE) KiB")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            # started plainly (`python bench.py --gpus N`, the form the driver uses at N = 1): this process becomes the
            # launcher -- the N ranks are CHILD processes of torch.distributed.run, started before anything here has
            # touched the GPU (no import of torch, no HIP call), never a re-exec -- and leaves with their exit code; the
            # ranks' output (rank 0's one JSON line) passes straight through
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            # (this process has sent its own fd 1 to stderr, see the bottom of the file: the ranks get the REAL stdout)
            raise SystemExit(subprocess.call(cmd, env=env, stdout=_REAL_STDOUT))
        args.gpus = world

    import numpy as np
    import torch  # plumbing only: device sync, barrier, max-over-ranks
    import torch.distributed as dist

    from ogl_amd import capi, synthetic

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path in libogl_amd")
    # a rank that never shows up on the peer mesh ends a wait after this many seconds (library default
    # 60): the ladder below may have to sit out two failed rungs before it reaches a working transport
    os.environ.setdefault("OGL_PEER_TIMEOUT_S", "15")
    # one rank per GPU; on a box with fewer GPUs than ranks (development only) ranks share devices
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        import datetime
        # gloo is the control plane only (agreeing on transports, the final MAX over ranks); a rank that dies
        # must not leave the others waiting for the default 30 minutes
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(
            seconds=float(os.environ.get("OGL_BENCH_GLOO_TIMEOUT_S", "480"))))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    n = args.n
    procs = (1, 1, world)                       # the headline: z-slabs of n^3 cells each
    if args.config:
        # BASELINE.json configs[3] / configs[4]: one global box for every N, cut as cubically as N allows
        n = args.n = {3: 272, 4: 368}[args.config]
        procs = {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}.get(world)
        if procs is None:
            raise SystemExit("--config 3 | 4 runs on 1, 2, 4 or 8 ranks")
        if args.config == 4:
            args.solver, args.precond = "GKOGMRES", "BJ"
        else:
            args.solver, args.precond = "GKOCG", "BJ"
        case = synthetic.poisson_block(n, n, n, procs[0], procs[1], procs[2], rank)
        if args.config == 4:                    # an unstructured numbering per rank (interfaces keep their face order)
            case = synthetic.renumber_case(case, 65536, seed=20241016 + rank)
    elif args.asym:
        case = synthetic.poisson_block(n, n, n * world, pz=world, rank=rank, symmetric=False,
                                       off_upper=-0.9, off_lower=-1.1, with_centres=args.centres)
    else:
        case = synthetic.poisson_block(n, n, n * world, pz=world, rank=rank, with_centres=args.centres)
    if args.voronoi:
        assert world == 1, "--voronoi is a single-rank option"
        # (the generating points stand in for mesh.C(): the plug-in passes the cell centres, ogl_ldu_view::cell_centres)
        case = synthetic.voronoi_case(args.voronoi, with_centres=not args.no_centres)
    if args.blocks:
        assert world == 1 and not args.voronoi, "--blocks is a single-rank option"
        case = synthetic.multi_block_case([int(v) for v in args.blocks.split(",")], n, n)
        if args.asym:       # (momentum-like coefficients on the same addressing, as tests/test_gpu_fullsize_configs.py)
            import dataclasses
            case = dataclasses.replace(case, upper=np.full(case.n_faces, -0.9), lower=np.full(case.n_faces, -1.1))
    if args.octree:
        assert world == 1 and not args.voronoi and not args.asym, "--octree is a single-rank, symmetric option"
        case = synthetic.octree_case(n, args.octree, args.octree_append)
    if args.drop_faces:
        assert world == 1, "--drop-faces is a single-rank option"
        case = synthetic.drop_faces_case(case, args.drop_faces)
    if args.long_rows:
        assert world == 1, "--long-rows is a single-rank option"
        case = synthetic.long_rows_case(case, args.long_rows, n)
    if args.shuffle:
        assert world == 1, "--shuffle is a single-rank option"
        case = synthetic.renumber_case(case, args.shuffle)
        if args.rcm:
            case = synthetic.rcm_case(case)
    N, nnz = case.n_cells, case.nnz
    # b = A x* with x*_i = sin(2 pi i / N) on the global numbering, x0 = 0
    b, _ = synthetic.rhs_for_x_star(case)

    precond = {"BJ": capi.PRECOND_BJ, "none": capi.PRECOND_NONE, "ISAI": capi.PRECOND_ISAI,
               "GISAI": capi.PRECOND_GISAI}[args.precond]
    solver_kind = {"GKOCG": capi.SOLVER_CG, "GKOBiCGStab": capi.SOLVER_BICGSTAB,
                   "GKOGMRES": capi.SOLVER_GMRES}[args.solver]
    cfg = capi.default_config(solver=solver_kind, preconditioner=precond,
                              max_block_size=args.block_size, krylov_dim=args.krylov_dim,
                              tolerance=0.0, rel_tol=0.0, max_iter=args.iters, min_iter=0,
                              eval_frequency=1, adapt_min_iter=0,
                              matrix_format=capi.FORMAT_ELL if args.format == "Ell" else capi.FORMAT_CSR,
                              export_res=0, profile_kernels=0 if args.no_profile else args.profile_stride,
                              compress_indices=0 if args.no_compress else (2 if args.force_compress else 1),
                              symmetric_half=0 if args.full_storage else 1,
                              renumber={"off": 0, "on": 1, "auto": 2}[args.renumber])

    def all_ok(ok):
        """True iff every rank says ok (gloo)."""
        if world == 1:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def _allreduce(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    def _exchange(neighbours, counts, send):
        recv = np.zeros_like(send)
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(int)
        reqs, bufs = [], []
        for i, nb in enumerate(neighbours):
            s_t = torch.from_numpy(np.ascontiguousarray(send[offs[i]:offs[i + 1]]))
            r_t = torch.zeros(int(counts[i]), dtype=torch.float64)
            reqs += [dist.isend(s_t, int(nb)), dist.irecv(r_t, int(nb))]
            bufs.append((i, r_t, s_t))
        for q in reqs:
            q.wait()
        for i, r_t, _ in bufs:
            recv[offs[i]:offs[i + 1]] = r_t.numpy()
        return recv

    def connect(kind, use_peer):
        """Registry + transport (N > 1), every step agreed by all ranks over gloo.  kind "rccl": RCCL
        over xGMI; "host": the host-buffer callbacks (gloo).  use_peer: on top, the peer mesh -- halo
        values put straight into the neighbours' receive blocks and all-reduces inside the finaliser
        kernels (hipIpc).  Returns (registry, description) or None when a rank could not follow."""
        reg = capi.Registry(device_id=local_rank, hip_stream=torch.cuda.current_stream().cuda_stream)
        if world == 1:
            return reg, "single GPU"
        ok, why = True, ""
        if kind == "rccl" and not all_ok(reg.rccl_ready()):
            # init_rccl is collective and has no time-out: nobody enters it unless everybody can
            print(f"bench.py: rank {rank}: RCCL not loadable on every rank", file=sys.stderr)
            reg.close()
            return None
        try:
            if kind == "rccl":
                uid = [capi.rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                reg.init_rccl(rank, world, uid[0])
            else:
                reg.set_host_comm(rank, world, _allreduce, _exchange)
        except capi.OglError as e:
            ok, why = False, str(e)
        if not all_ok(ok):
            if why:
                print(f"bench.py: rank {rank}: transport {kind} unavailable ({why})", file=sys.stderr)
            reg.close()
            return None
        desc = ("RCCL" if kind == "rccl" else "host-buffer (gloo)") + " halo + all-reduce"
        if use_peer:
            mine = None
            try:
                mine = reg.peer_handle()
            except capi.OglError as e:
                print(f"bench.py: rank {rank}: no peer mailbox ({e})", file=sys.stderr)
            handles = [None] * world
            dist.all_gather_object(handles, mine)          # every rank takes part, handle or not
            ok = all(h is not None for h in handles)
            if ok:
                try:
                    reg.peer_connect(rank, world, handles)   # collective self-test inside
                except capi.OglError as e:
                    ok = False
                    print(f"bench.py: rank {rank}: peer mesh unavailable ({e})", file=sys.stderr)
            if not all_ok(ok):
                reg.close()
                return None
            desc = desc.replace("halo + all-reduce", "halo, peer-write all-reduce (hipIpc)")
        return reg, desc

    def selfcheck(reg):
        """N > 1, before anything is timed: 20 GKOCG + BJ turns on a 64^3-per-rank
        slab case through the transport just brought up.  The residual history must be bit-identical on
        every rank and agree with a single-GPU solve of the assembled system (rank 0) to 1e-11; the
        solution slices to 1e-12."""
        e = args.selfcheck_edge
        sc = synthetic.poisson_block(e * procs[0], e * procs[1], e * procs[2], procs[0], procs[1], procs[2], rank)
        sb, _ = synthetic.rhs_for_x_star(sc)
        c = capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, tolerance=0.0,
                                rel_tol=0.0, max_iter=20, export_res=1, adapt_min_iter=0,
                                matrix_format=capi.FORMAT_CSR)
        rep = {"edge": e, "ok": False}
        hist, x = None, None
        try:
            # both multi-rank GKOCG turns go through the transport: the merged 4-launch one (what a share of this size
            # runs by default with the peer mesh: z put by step_2r) and the 5-launch one (p put by step_1x: what the
            # timed 216^3 shares run) -- same bits required
            sv5 = reg.solver("selfcheck5", c)
            sv5.set_property("fusedTurnMulti", 0.0)
            sv5.set_matrix(sc)
            x5, _ = sv5.solve(sb, np.zeros_like(sb))
            sv = reg.solver("selfcheck", c).set_matrix(sc)
            x, perf = sv.solve(sb, np.zeros_like(sb))
            hist = sv.history()
            rep["peer_halo"] = sv.get_property("peerHalo") == 1.0
            rep["merged_turn_checked"] = sv.get_property("fusedTurnInUse") == 1.0
            rep["five_launch_turn_bit_identical"] = bool(np.array_equal(sv5.history(), hist) and np.array_equal(x5, x))
            if not rep["five_launch_turn_bit_identical"]:
                hist = None
                rep["error"] = "the 4- and the 5-launch turn disagree"
        except capi.OglError as ex:
            rep["error"] = str(ex)
        if not all_ok(hist is not None):
            return rep
        every = [None] * world
        dist.all_gather_object(every, hist.tobytes())
        rep["history_bit_identical_on_all_ranks"] = all(h == every[0] for h in every)
        # single-GPU solve of the assembled system on rank 0's device, broadcast to all
        glob_n = e * e * e * world
        href = torch.zeros(hist.size, dtype=torch.float64)
        xref = torch.zeros(glob_n, dtype=torch.float64)
        if rank == 0:
            r1 = capi.Registry(device_id=local_rank)
            g = synthetic.poisson_block(e * procs[0], e * procs[1], e * procs[2])
            gb, _ = synthetic.rhs_for_x_star(g)
            s1 = r1.solver("selfcheck_global", c).set_matrix(g)
            gx, _ = s1.solve(gb, np.zeros_like(gb))
            href = torch.from_numpy(s1.history().copy())
            xref = torch.from_numpy(gx.copy())
            r1.close()
        dist.broadcast(href, src=0)
        dist.broadcast(xref, src=0)
        href, xref = href.numpy(), xref.numpy()
        dev_h = float(np.max(np.abs(hist - href) / np.abs(href)))
        dev_x = float(np.max(np.abs(x - xref[sc.global_index])))
        worst = torch.tensor([dev_h, dev_x], dtype=torch.float64)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        rep["max_rel_dev_history_vs_single_gpu"] = float(worst[0])
        rep["max_abs_dev_x_vs_single_gpu"] = float(worst[1])
        rep["ok"] = bool(rep["history_bit_identical_on_all_ranks"] and worst[0] <= 1e-11 and worst[1] <= 1e-12)
        return rep

    def load(reg):
        """The benchmark's own system on a connected registry: matrix and b resident, warm-up done."""
        s = reg.solver("p", cfg)
        if args.graph != "auto":
            s.set_property("hipGraph", 1.0 if args.graph == "on" else 0.0)
        for kv in args.prop:
            k, v = kv.split("=", 1)
            s.set_property(k, float(v))
        t0 = time.perf_counter()
        s.set_matrix(case)                       # pattern + H2D + device permutation (not timed below)
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        s.set_matrix(case)                       # values-only refresh, as every later time step
        t_refresh = time.perf_counter() - t0
        s.upload_rhs(b)
        for _ in range(args.warmup):
            s.upload_solution(None)
            s.apply_resident()
        return s, t_first, t_refresh

    # Transport ladder at N > 1: peer mesh over RCCL bootstrap -> peer mesh over the host bootstrap ->
    # RCCL alone -> host-buffer alone.  A rung is kept only if every rank connects, the self-check
    # passes and the benchmark system loads and warms up on it; otherwise all ranks step down together.
    # (All of it in this process: nothing is re-exec'ed after the GPU has been touched.)
    ladder = [("rccl", True), ("host", True), ("rccl", False), ("host", False)]
    want = os.environ.get("OGL_BENCH_TRANSPORT", "")
    if want:
        ladder = [r for r in ladder if r[0] == want]
    if os.environ.get("OGL_BENCH_PEER", "1") != "1":
        ladder = [r for r in ladder if not r[1]]
    full_ladder = [("rccl", True), ("host", True), ("rccl", False), ("host", False)]
    rung_name = lambda r: r[0] + ("+peer" if r[1] else "")
    if os.environ.get("OGL_BENCH_RUNG"):     # a rung probe (child of a bench.py run): exactly this rung or nothing
        ladder = [r for r in full_ladder if rung_name(r) == os.environ["OGL_BENCH_RUNG"]]
    if world == 1:
        ladder = [("none", False)]
    state, selfcheck_report, tried, chosen_rung = None, None, [], None
    for kind, use_peer in ladder:
        got = connect(kind, use_peer)
        if got is None:
            tried.append(f"{kind}{'+peer' if use_peer else ''}: no connection")
            continue
        reg, transport = got
        try:
            info = reg.comm_info()
            if world > 1 and args.selfcheck:
                selfcheck_report = selfcheck(reg)
                selfcheck_report["transport"] = transport
                selfcheck_report["rccl_ranks_seen"] = info.ranks_seen if info.transport == 2 else None
                if not all_ok(selfcheck_report["ok"]):
                    tried.append(f"{kind}{'+peer' if use_peer else ''}: self-check failed {selfcheck_report}")
                    if rank == 0:
                        print(f"bench.py: self-check FAILED on {transport}: {selfcheck_report}", file=sys.stderr)
                    reg.close()
                    continue
            s, t_first_matrix, t_refresh_matrix = None, 0.0, 0.0
            err = None
            try:
                s, t_first_matrix, t_refresh_matrix = load(reg)
            except capi.OglError as e:
                err = e
            if not all_ok(s is not None):
                if world == 1:
                    raise err
                tried.append(f"{kind}{'+peer' if use_peer else ''}: load/warm-up failed ({err})")
                reg.close()
                continue
            state = (reg, s, transport, t_first_matrix, t_refresh_matrix)
            chosen_rung = rung_name((kind, use_peer))
            break
        except Exception:
            reg.close()
            raise
    if state is None:
        raise SystemExit(f"bench.py: rank {rank}: no transport survived: {tried}")
    reg, s, transport, t_first_matrix, t_refresh_matrix = state
    comm_info = reg.comm_info()

    def probe_rungs(timed_rung, timed_us_per_turn):
        """Evidence for the rungs the timed run did not use (RCCL with more than one rank must not
        need the peer mesh to fail before it is ever exercised).  Each remaining rung: every rank starts ONE child
        process -- this script, restricted to that rung (OGL_BENCH_RUNG), own gloo rendezvous on a port of its own --
        which connects, runs the cross-rank self-check and two short timed steps of this workload.  Children are
        ordinary child processes (nothing is exec'ed in place), watched with a time limit and ended by PID when it
        passes: a rung that hangs costs its time limit, not the run."""
        res = []
        base_port = int(os.environ.get("MASTER_PORT", "29500"))
        for idx, r in enumerate(full_ladder):
            name = rung_name(r)
            if name == timed_rung:
                res.append({"rung": name, "ok": True, "timed_run": True, "us_per_turn": timed_us_per_turn,
                            "selfcheck_ok": None if not selfcheck_report else bool(selfcheck_report.get("ok"))})
                continue
            env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC")}
            env.update({"MASTER_PORT": str(base_port + 31 + idx), "OGL_BENCH_RUNG": name, "OGL_BENCH_CHILD": "1",
                        "OGL_BENCH_GLOO_TIMEOUT_S": str(int(args.rung_timeout))})
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                   "--iters", str(min(args.iters, 50)), "--edge", str(n), "--precond", args.precond, "--solver", args.solver,
                   "--block-size", str(args.block_size), "--krylov-dim", str(args.krylov_dim), "--format", args.format,
                   "--cpu-iters", "0", "--no-general-legs", "--no-rung-probes", "--live-pmc", "off",
                   "--selfcheck-edge", str(args.selfcheck_edge)] + (["--asym"] if args.asym else []) + \
                  (["--config", str(args.config)] if getattr(args, "config", 0) else []) + \
                  sum((["--prop", kv] for kv in args.prop), [])
            t0 = time.perf_counter()
            rec = {"rung": name, "ok": False, "timed_run": False}
            p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            try:
                so, se = p.communicate(timeout=args.rung_timeout)
                rec["rc"] = p.returncode
                if p.returncode == 0 and rank == 0:
                    d = json.loads(so.strip().splitlines()[-1])
                    sc = d["config"].get("selfcheck") or {}
                    rec.update({"ok": bool(sc.get("ok", False)) if args.selfcheck else True,
                                "selfcheck_ok": sc.get("ok"), "us_per_turn": 1e3 * d["solver_turn"]["ms"],
                                "value": d["value"], "transport": d["config"]["parallelism"],
                                "rccl_ranks_seen": d["config"]["transport"]["rccl_ranks_seen"],
                                "wait_us": d["config"]["transport"]["wait_us"]})
                elif p.returncode != 0:
                    said = [ln for ln in se.splitlines() if ln.startswith("bench.py:")]
                    rec["error"] = " | ".join(dict.fromkeys(ln[-300:] for ln in said[-3:])) if said else \
                        (se.strip().splitlines()[-1][-400:] if se.strip() else "child failed")
            except subprocess.TimeoutExpired:
                p.kill()
                p.communicate()
                rec["error"] = f"no result within {args.rung_timeout:.0f} s: children ended"
            rec["seconds"] = time.perf_counter() - t0
            # every rank's child has ended before the next rung starts (the children of one rung talk to each other)
            every = [None] * world
            dist.all_gather_object(every, rec.get("rc", -9))
            rec["child_rc_by_rank"] = every
            if any(c != 0 for c in every):
                rec["ok"] = False
            res.append(rec)
        return res

    def step():
        s.upload_solution(None)              # x0 = 0, device memset
        return s.apply_resident()

    barrier()
    t0 = time.perf_counter()
    perfs = [step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # turns actually done: CG / GMRES report checks (= turns + 1), BiCGStab reports checks / 2
    bicg = args.solver == "GKOBiCGStab"
    iters = sum(p.n_iterations - (0 if bicg else 1) for p in perfs)
    expect = args.iters if bicg else args.iters + 1
    assert all(p.n_iterations == expect for p in perfs), [p.n_iterations for p in perfs]
    # headline: weak scaling, 10 M-row block iterations of all ranks; --config 3 | 4: ONE global system, global turns
    value = iters / elapsed if args.config else world * iters / elapsed

    # ---- roofline of the dominant kernel: the in-loop SpMV -----------------------------------
    def prop_or(sv, name, default):
        try:
            return sv.get_property(name)
        except capi.OglError:
            return default

    def spmv_roofline(sv, n_rows, n_nnz, spmv_ms, timing, pmc_variant, in_loop=True):
        """roofline object of the in-loop SpMV of solver `sv`: `achieved` / `frac` on the bytes the layout in
        use has to move (matrix data + x read once + y written), the CSR-equivalent rate apart."""
        b_csr = 12 * n_nnz + 20 * n_rows + 4            # values + columns + row pointers + x + y
        if args.format == "Ell":                        # 7 slots/row -> 84 N + 16 N
            b_csr = (12 * 7 + 16) * n_rows
        layout = {0.0: "csr", 1.0: "ell", 2.0: "sell", 3.0: "csr21"}[sv.get_property("spmvLayout")]
        if layout == "sell" and prop_or(sv, "symmetricHalf", 0.0) == 1.0:
            layout = "sym"   # half storage of a symmetric matrix on a banded pattern (diagonal + upper planes)
            if prop_or(sv, "symmetricHalfPerChunk", 0.0) == 1.0:
                layout = "symx"  # ... with per-chunk distances and explicit exceptions (multi-block meshes)
        stream = "true" if prop_or(sv, "spmvStream", 0.0) == 1.0 else "false"
        kernel = {"csr": f"k_spmv_stream<0, 1, {stream}, {int(prop_or(sv, 'spmvLdsRounds', 1.0))}>", "csr21": f"k_spmv_stream21<0, 1, {stream}>",
                  "ell": f"k_spmv_ell<0, 1, {stream}>",
                  "sell": f"k_spmv_sell<0, 1, {stream}>",
                  # (the lean instantiation -- chunks without explicit entries or with simple ones; the general one runs
                  #  in a second launch on the chunks property symxGeneralChunks counts)
                  "symx": f"k_spmv_symx<0, 1, {stream}, "
                          f"{'true' if prop_or(sv, 'spmvSymFast', 0.0) == 1.0 else 'false'}, false>",
                  "sym": f"k_spmv_sym<0, 1, {int(prop_or(sv, 'spmvSymPlanes', 0))}, "
                         f"{'true' if prop_or(sv, 'spmvSymFast', 0.0) == 1.0 else 'false'}, {stream}>"}[layout]
        # bytes the kernel has to move for the layout it runs on (matrix + x read once + y written)
        b_moved = ((sv.get_property("sellMatrixBytes") + 16 * n_rows) if layout in ("sell", "sym", "symx") else
                   (sv.get_property("csr21MatrixBytes") + 16 * n_rows) if layout == "csr21" else b_csr)
        merged = None
        if (in_loop and layout == "sym" and args.solver == "GKOCG" and prop_or(sv, "fusedTurnInUse", 0.0) == 1.0):
            # the in-loop kernel on half storage is step_1x + SpMV in one launch (k_cg_turn_sym / _big, kernels_spmv_sym.hip):
            # besides the SpMV's bytes it reads z and x and writes x and the new p of its rows
            nd, fast = int(prop_or(sv, 'spmvSymPlanes', 0)), 'true' if prop_or(sv, 'spmvSymFast', 0.0) == 1.0 else 'false'
            small = prop_or(sv, "fusedFinalizersInUse", 0.0) == 1.0
            kernel = f"k_cg_turn_sym<{nd}, {fast}>" if small else f"k_cg_turn_sym_big<{nd}, {fast}, {stream}>"
            merged = {"with": "step_1x" + ("_fin (check of the previous turn, partial sums)" if small else ""),
                      "extra_bytes_per_launch": 32 * n_rows}
            b_moved += 32 * n_rows
            b_csr += 32 * n_rows
        traffic, traffic_src = pmc_traffic(kernel, pmc_variant) if pmc_variant is not None else (None, None)
        moved = b_moved / (spmv_ms * 1e-3) / 1e9
        return layout, b_moved, b_csr, {
            "kernel": kernel, "layout": layout, "merged": merged, "bound": "hbm",
            "achieved": moved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": moved / HBM_PEAK_GBPS,
            "bytes_per_launch": b_moved,
            "frac_of_measured_copy_peak": moved / HBM_COPY_GBPS,
            # (the committed summary of exactly this kernel instantiation; replaced in main() by counters collected
            #  in this run when the live PMC passes are on)
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_measured_in_this_run": False,
            "traffic_over_model": None if traffic is None else traffic / b_moved,
            # the same time priced in SURVEY.md 8(d)'s CSR bytes (the unit of work `matrixFormat Csr` names):
            # a rate of work, not a bandwidth -- above 1 when the layout moves fewer bytes than a CSR
            "csr_equivalent_bytes_per_launch": b_csr,
            "csr_equivalent_achieved": b_csr / (spmv_ms * 1e-3) / 1e9,
            "csr_equivalent_frac": b_csr / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "avg_kernel_ms": spmv_ms, "timing": timing,
        }

    if args.no_profile:
        spmv_ms = s.time_spmv(100)
        spmv_src = "100 back-to-back launches, HIP events"
    else:
        launches = sum(p.spmv_launches for p in perfs)
        spmv_ms = sum(p.spmv_avg_ms * p.spmv_launches for p in perfs) / max(1, launches)
        spmv_src = (f"{launches} in-loop launches of the timed steps (every "
                    f"{args.profile_stride}th turn), HIP event pairs")
    plain_box = not (args.voronoi or args.octree or args.drop_faces or args.long_rows or args.asym or args.blocks)
    pmc_variant = None
    if n == 216 and args.format == "Csr" and plain_box and args.renumber == "auto" and not args.rcm \
            and args.shuffle in (0, 65536):
        pmc_variant = ("_shuffle65536" if args.shuffle else "_fullstorage" if args.full_storage else
                       "_nocompress" if args.no_compress else "")
    layout, b_moved, b_spmv, roofline = spmv_roofline(s, N, nnz, spmv_ms, spmv_src, pmc_variant,
                                                      in_loop=not args.no_profile)
    profiled = any(k in os.environ.get("LD_PRELOAD", "") for k in ("rocprof", "roctracer")) or \
        any(k.startswith("ROCPROF") for k in os.environ)
    live_tail = ["--edge", str(n), "--precond", args.precond, "--solver", args.solver]
    want_live = args.live_pmc == "on" or (args.live_pmc == "auto" and pmc_variant == "" and not args.no_profile
                                          and not args.prop and not profiled)
    if want_live and world == 1 and rank == 0 and not os.environ.get("OGL_BENCH_CHILD"):
        live, how = live_pmc_traffic(roofline["kernel"], live_tail)
        roofline["traffic_committed_summary"] = {"bytes": roofline["traffic"], "source": roofline["traffic_source"]}
        if live is not None:
            roofline.update({"traffic": live, "traffic_source": how, "traffic_measured_in_this_run": True,
                             "traffic_over_model": live / roofline["bytes_per_launch"]})
        else:
            roofline["traffic_live_attempt"] = how
    renumbered = s.get_property("renumbered") == 1.0
    if world > 1 and s.get_property("peerHalo") == 1.0:
        # the transport above was only the bootstrap: halo values are put straight into the
        # neighbours' receive blocks (hipIpc-mapped) by the pack kernel
        transport = "peer-put halo + peer-write all-reduce over xGMI (hipIpc), bootstrap: " + \
                    ("RCCL" if transport.startswith("RCCL") else "gloo")
    # one CG turn: the SpMV + the vector passes of the fused kernels (x update deferred into step_1x: p is
    # read once per turn): 80 N with scalar Jacobi, 64 N without; SURVEY.md §8d's model has 88 N / 72 N
    cg_headline = (args.solver == "GKOCG" and precond in (capi.PRECOND_BJ, capi.PRECOND_NONE)
                   and args.block_size == 1)
    b_cg = b_moved + (80 if precond == capi.PRECOND_BJ else 64) * N if cg_headline else None
    b_cg_csr = b_spmv + (88 if precond == capi.PRECOND_BJ else 72) * N if cg_headline else None
    if cg_headline and roofline["merged"]:
        # step_1x merged into the SpMV kernel (its 32 N are in b_moved already): what is left is step_2r, which
        # also keeps z for the next turn's gathers -- 72 N / 56 N per turn besides the SpMV's own bytes
        b_cg = b_moved + (40 if precond == capi.PRECOND_BJ else 24) * N
        b_cg_csr = b_spmv + (56 if precond == capi.PRECOND_BJ else 40) * N   # (SURVEY's unit of work is unchanged)

    def turn_model(b_mat):
        """Bytes of ONE solver turn: the SpMV(s) priced at `b_mat` + every vector pass the Ginkgo step order
        needs when neighbouring element-wise steps are fused (what the kernels here do), each vector counted
        once per read or write.  GKOCG: SURVEY.md §8d.  Others: DESIGN.md §4."""
        k, m = args.block_size, args.krylov_dim
        nnz_w = (nnz + N) // 2 if args.precond == "ISAI" else nnz          # tril(A) resp. pattern of A
        if args.precond == "none":
            apply_b, materialised = 0, False
        elif args.precond == "BJ" and k == 1:
            apply_b, materialised = 8 * N, False                          # inv_diag read, fused
        elif args.precond == "BJ":
            apply_b, materialised = (8 * k + 20) * N, True                # block rows + in + out + row->block
        else:
            n_spmv = 2 if args.precond == "ISAI" else 1                   # W^T (W r)  resp.  W r
            apply_b, materialised = n_spmv * (12 * nnz_w + 20 * N), True
        if args.solver == "GKOCG":
            # fused: step_1 24 N (+ deferred x update 16 N) , step_2 24 N + partials; materialised z: +16 N
            vec = 72 * N if not materialised else (24 + 48 + 0) * N
            return b_mat + vec + (apply_b if apply_b != 8 * N else 16 * N), "B_spmv + vector passes + M^-1"
        if args.solver == "GKOBiCGStab":
            vec = (24 + 24 + 64) * N + (2 * 16 * N if apply_b == 8 * N else 0)   # step_1/2/3 (+ y, z written)
            return 2 * b_mat + vec + 2 * (apply_b if materialised else 0), "2 B_spmv + step_1/2/3 + 2 M^-1"
        # GKOGMRES(m): column `it` of a cycle costs it + 1 modified Gram-Schmidt links of 32 N (w read and
        # written, v_k-1 and v_k read), the closing link 24 N, the scaling 16 N; per cycle once: update of
        # x (8 m N + 24 N), the residual SpMV and the restart (24 N)
        per_col = b_mat + (apply_b + 16 * N if not materialised else apply_b) + 32 * N * (m + 1) / 2 + 40 * N
        per_cycle = b_mat + (8 * m + 48) * N + (apply_b if materialised else 8 * N)
        return per_col + per_cycle / m, "B_spmv + M^-1 + (m+1)/2 MGS links of 32 N + cycle overhead / m"
    merged_extra = roofline["merged"]["extra_bytes_per_launch"] if roofline["merged"] else 0
    b_turn, b_turn_what = turn_model(b_moved - merged_extra)   # (the model prices the SpMV alone)
    b_turn_csr, _ = turn_model(b_spmv - merged_extra)
    # end-to-end plug-in call incl. PCIe (reported, never `value`): one solve() with H2D/D2H
    # (host arrays prepared outside the timed region: the C ABI borrows the caller's arrays, it allocates nothing)
    ldu_arrays = capi.LduArrays(case)
    psi_io = np.zeros_like(b)
    t0 = time.perf_counter()
    s.set_matrix(ldu_arrays)
    _, p_e2e = s.solve(b, psi_io, inplace=True)
    t_e2e = time.perf_counter() - t0

    # Momentum components (ogl_solver_set_matrix_like): three solvers on ONE set of host arrays, as solveSegregated
    # builds them for Ux, Uy, Uz -- the second and third take the first one's device copy of upper / lower
    components = None
    if world == 1 and not args.config and not os.environ.get("OGL_BENCH_CHILD") and plain_box and not args.shuffle \
            and cg_headline:
        cs = [reg.solver("U" + c, cfg) for c in "xyz"]
        for k, c in enumerate(cs):                       # (patterns: not timed)
            c.set_matrix(ldu_arrays, like=cs[k - 1] if k else None)
        torch.cuda.synchronize()
        ms, reused = [], []
        for k, c in enumerate(cs):
            t0 = time.perf_counter()
            c.set_matrix(ldu_arrays, like=cs[k - 1] if k else None)
            ms.append(1e3 * (time.perf_counter() - t0))
            reused.append(c.get_property("offDiagReused") == 1.0)
        components = {"refresh_ms": ms, "off_diagonals_taken_from_the_sibling": reused,
                      "what": "values-only set_matrix of Ux, Uy, Uz on one lduMatrix (same upper / lower host arrays): "
                              "Uy and Uz upload their diagonal only and copy upper / lower device to device"}
        del cs

    # ---- the general layouts on the same system (N = 1, the plain benchmark box only), outside the timed
    # region: full storage (pattern-coded compressed copy), plain CSR-stream, cells shuffled (irregular
    # numbering: the library renumbers its device copy itself) -- each a short run of its own solver
    general = None
    if world == 1 and args.general_legs and plain_box and not args.shuffle and args.format == "Csr" and not args.config \
            and not (args.full_storage or args.no_compress or args.force_compress) and cg_headline \
            and not args.no_profile:
        general = []
        legs = [("full_storage", dict(symmetric_half=0), None, "_fullstorage"),
                ("no_compress", dict(compress_indices=0), None, "_nocompress"),
                ("shuffle65536", dict(), 65536, "_shuffle65536")]
        for name, over, shuffle, variant in legs:
            leg_case, leg_b = case, b
            if shuffle:
                leg_case = synthetic.renumber_case(case, shuffle)
                leg_b, _ = synthetic.rhs_for_x_star(leg_case)
            d = {f: getattr(cfg, f) for f, _ in cfg._fields_}
            d.update(over)
            leg_cfg = type(cfg)(**d)
            sv = reg.solver("p_" + name, leg_cfg)
            t0 = time.perf_counter()
            sv.set_matrix(leg_case)
            t_first = time.perf_counter() - t0
            sv.upload_rhs(leg_b)
            sv.upload_solution(None)
            sv.apply_resident()                                    # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lp = []
            for _ in range(args.general_steps):
                sv.upload_solution(None)
                lp.append(sv.apply_resident())
            torch.cuda.synchronize()
            t_leg = time.perf_counter() - t0
            ln = sum(p.spmv_launches for p in lp)
            l_ms = sum(p.spmv_avg_ms * p.spmv_launches for p in lp) / max(1, ln)
            _, _, _, r = spmv_roofline(sv, leg_case.n_cells, leg_case.nnz, l_ms,
                                       f"{ln} in-loop launches, HIP event pairs",
                                       variant if (n == 216 and args.renumber == "auto") else None)
            r.update({"leg": name, "steps": args.general_steps,
                      "cg_iters_per_sec": sum(p.n_iterations - 1 for p in lp) / t_leg,
                      "first_set_matrix_s": t_first,
                      "renumbered": sv.get_property("renumbered") == 1.0,
                      "moved_frac": r["frac"]})
            if want_live and world == 1 and rank == 0 and not os.environ.get("OGL_BENCH_CHILD"):
                leg_flags = {"full_storage": ["--full-storage"], "no_compress": ["--no-compress"],
                             "shuffle65536": ["--shuffle", "65536"]}[name]
                live, how = live_pmc_traffic(r["kernel"], live_tail + leg_flags)
                r["traffic_committed_summary"] = {"bytes": r["traffic"], "source": r["traffic_source"]}
                if live is not None:
                    r.update({"traffic": live, "traffic_source": how, "traffic_measured_in_this_run": True,
                              "traffic_over_model": live / r["bytes_per_launch"]})
                else:
                    r["traffic_live_attempt"] = how
            general.append(r)
            del sv

    # ---- N > 1: where the turns waited, and evidence for EVERY rung of the transport ladder ------------------
    waits, rungs = None, None
    if world > 1:
        turns = max(1, perfs[-1].n_iterations - (0 if bicg else 1))
        mine = {"rank": rank,
                # halo: sum over the workgroups that waited for the neighbours' puts (boundary workgroups of the SpMV; ONE
                # with the single waiter) of their longest flag wait; all-reduce: the finaliser's longest mailbox wait,
                # summed over the turn's all-reduces -- device clock, last timed step, per turn (DevScalars, kernels.hpp)
                "halo_wait_us_per_turn": prop_or(s, "haloWaitUs", 0.0) / turns,
                "halo_waiting_workgroups_per_turn": prop_or(s, "haloWaits", 0.0) / turns,
                "allreduce_wait_us_per_turn": prop_or(s, "allreduceWaitUs", 0.0) / turns,
                "allreduces_per_turn": prop_or(s, "allreduceWaits", 0.0) / turns,
                "single_waiter": prop_or(s, "peerSafeWaitInUse", 0.0) == 1.0,
                "shares_its_device": prop_or(s, "peerSharedDevice", 0.0) == 1.0}
        waits = [None] * world
        dist.all_gather_object(waits, mine)
        if args.rung_probes and not os.environ.get("OGL_BENCH_RUNG"):
            rungs = probe_rungs(chosen_rung, 1e3 * elapsed / max(1, iters) * 1e3)

    if general:
        # north_star's graded figure -- the in-loop CSR SpMV on SURVEY.md 8(d)'s bytes (12 nnz + 20 N + 4) -- where the
        # record's reader looks for it: the `no_compress` leg's kernel (k_spmv_stream, the kernel north_star sketches)
        g = next(x for x in general if x["leg"] == "no_compress")
        roofline["csr"] = {k: g[k] for k in ("kernel", "frac", "achieved", "avg_kernel_ms", "bytes_per_launch", "traffic",
                                              "traffic_over_model", "traffic_measured_in_this_run", "cg_iters_per_sec")}
        roofline["csr"]["what"] = ("compressIndices false on the same system, 5 steps after the timed region: the plain CSR "
                                   "arrays through k_spmv_stream; bytes_per_launch = SURVEY.md 8(d)'s 12 nnz + 20 N + 4")

    out = {
        "metric": "cg_iters_per_sec", "value": value, "unit": "iter/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / max(1, args.steps),
        "higher_is_better": True, "scaling": "strong" if args.config else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": ({3: f"BASELINE.json configs[3]: channel-flow proxy, {n}^3 = {n ** 3:,} cells cut into "
                             f"{procs[0]} x {procs[1]} x {procs[2]} blocks (strong scaling), ",
                          4: f"BASELINE.json configs[4]: unstructured proxy, {n}^3 = {n ** 3:,} cells cut into "
                             f"{procs[0]} x {procs[1]} x {procs[2]} blocks, every rank's cells shuffled in windows of 65536 "
                             f"(strong scaling), matrixFormat {args.format}, "}.get(args.config, "")) +
                        (f"Voronoi mesh of {args.voronoi} random points (polyhedral cells, random numbering) lduMatrix"
                         if args.voronoi else
                         f"octree mesh: {n}^3 hexahedra, those within {args.octree} cells of a sphere split 2x2x2"
                         f"{' (children appended to the cell list)' if args.octree_append else ''} lduMatrix"
                         if args.octree else
                         f"multi-block mesh: blocks of {args.blocks} x {n} x {n} hexahedra lduMatrix"
                         if args.blocks else f"{n}^3-per-GPU 7-pt Poisson lduMatrix") +
                        f"{' (non-symmetric)' if args.asym else ''}"
                        f"{f' ({args.drop_faces:.0%} of the faces removed at random)' if args.drop_faces else ''}"
                        f"{f' ({args.long_rows:.0%} of the cells with 5 extra couplings)' if args.long_rows else ''}"
                        f"{f' (cells shuffled within windows of {args.shuffle}' + (', then RCM' if args.rcm else '') + ')' if args.shuffle else ''}, {args.solver}"
                        f"{'(' + str(args.krylov_dim) + ')' if args.solver == 'GKOGMRES' else ''} + "
                        f"{args.precond + ('(maxBlockSize ' + str(args.block_size) + ')' if args.precond == 'BJ' else '') if precond else 'no preconditioner'}, "
                        + {"sell": "fp64 SpMV on the index-compressed SELL copy (1-byte / 16-bit column codes) of "
                                   "the persistent fp64/int32 device CSR",
                           "sym": "fp64 SpMV on the half storage of the symmetric matrix (diagonal + upper planes, "
                                  "as the lduMatrix holds it; lower entries read where their twins live) next to "
                                  "the persistent fp64/int32 device CSR",
                           "symx": "fp64 SpMV on the half storage of the symmetric matrix with per-chunk distances and "
                                   "explicit entries for what breaks the bands (multi-block mesh, refinement shell)",
                           "csr": "fp64/int32 persistent device CSR (CSR-stream SpMV)",
                           "csr21": "fp64 persistent device CSR values, columns as 21-bit offsets packed six to a "
                                    "16-byte word (CSR-stream SpMV)",
                           "ell": "fp64/int32 ELL copy of the persistent device CSR"}[layout]
                        + (", device copy renumbered by the library (RCM)" if renumbered else "")
                        + ("" if args.config else
                           " (BASELINE.json configs[1])" if not (args.voronoi or args.shuffle or args.drop_faces
                                                                or args.long_rows or args.octree or args.blocks)
                           else " (proxy of a multi-block / unstructured mesh)"),
            "rows_per_gpu": N, "nnz_per_gpu": nnz, "cg_iters_per_step": args.iters,
            "renumber": args.renumber, "renumbered": renumbered,
            "rows_sorted_by_length": prop_or(s, "rowsSortedByLength", 0.0) == 1.0,
            "numbering": {"along_hilbert_curve": prop_or(s, "renumberedAlongCurve", 0.0) == 1.0,
                          "gather_sectors_rcm": prop_or(s, "gatherSectorRatioRcm", None),
                          "gather_sectors_curve": prop_or(s, "gatherSectorRatioCurve", None)},
            # irregular patterns: both SpMV kernels timed once per pattern at set_matrix, the faster one runs
            "layout_tuned_us": ({"csr": prop_or(s, "spmvTunedCsrUs", None), "sell": prop_or(s, "spmvTunedSellUs", None),
                                 "csr21": prop_or(s, "spmvTunedCsr21Us", None), "symx": prop_or(s, "spmvTunedSymxUs", None)}
                                if (prop_or(s, "spmvTunedCsrUs", None) is not None
                                    or prop_or(s, "spmvTunedSymxUs", None) is not None) else None),
            "spilled_entries": prop_or(s, "sellSpilledEntries", 0.0) if layout == "sell" else 0.0,
            "gather_sectors_per_entry": {"as_given": s.get_property("gatherSectorRatioNatural"),
                                         "in_use": s.get_property("gatherSectorRatio")},
            "parallelism": (f"rows sharded into {procs[0]} x {procs[1]} x {procs[2]} blocks, {transport}" if args.config else
                            f"rows sharded into {world} z-slab(s), {transport}") if world > 1 else "single GPU",
            "transport": {"kind": {0: "none", 1: "host-buffer", 2: "rccl"}[comm_info.transport],
                          "rccl_ranks_seen": comm_info.ranks_seen if comm_info.transport == 2 else None,
                          "peer_mesh": bool(comm_info.peer_mesh), "stepped_down_from": tried,
                          "timed_rung": chosen_rung,
                          # every rung of the ladder brought up once (the ones the timed run did not use: in child
                          # processes after the timed region), self-checked and timed on this workload
                          "rungs": rungs,
                          "wait_us": waits},
            "selfcheck": selfcheck_report,
        },
        "roofline": dict(roofline, note=(
            "achieved / frac: the bytes the kernel has to move for the layout it runs on (bytes_per_launch: matrix "
            "data of that layout + x read once + y written; layout model, checked against the PMC traffic) over "
            "avg_kernel_ms -- "
            + {"sym": "diagonal + upper coefficients only: the matrix is symmetric and every lower entry is read "
                      "where its upper twin lives (same bits in y)",
               "symx": "diagonal + upper planes per chunk and explicit entries where the pattern breaks its bands",
               "sell": "an index-compressed copy of the CSR arrays",
               "ell": "slot-major planes of the CSR arrays",
               "csr21": "the CSR values and row pointers with the columns packed to 21 bits",
               "csr": "the CSR arrays themselves (= SURVEY.md 8(d)'s figure)"}[layout]
            + "; csr_equivalent_*: the same time priced in SURVEY.md 8(d)'s CSR bytes, a rate of work that exceeds "
              "the bandwidth when the layout moves fewer bytes than a CSR would")),
        "roofline_general": general,
        "cg_iteration": {
            "bytes": b_cg, "ms": 1e3 * elapsed / max(1, iters),
            "achieved_GBps": None if b_cg is None else b_cg * iters / elapsed / 1e9,
            "frac_of_peak": None if b_cg is None else b_cg * iters / elapsed / 1e9 / HBM_PEAK_GBPS,
            "model": ("bytes the kernels of one turn move: SpMV layout bytes + 16 N + 72 N (scalar Jacobi; 56 N "
                      "without) -- step_1x merged into the SpMV kernel, z kept by step_2r" if roofline["merged"] else
                      "bytes the kernels of one turn move: SpMV layout bytes + 16 N + 80 N (scalar Jacobi; 64 N without)"),
            "csr_equivalent_bytes": b_cg_csr,
            "csr_equivalent_frac_of_peak": None if b_cg_csr is None else
                                           b_cg_csr * iters / elapsed / 1e9 / HBM_PEAK_GBPS,
        },
        "solver_turn": {
            "bytes": b_turn, "model": b_turn_what + " (SpMV priced at the bytes its layout moves)",
            "ms": 1e3 * elapsed / max(1, iters),
            "achieved_GBps": b_turn * iters / elapsed / 1e9,
            "frac_of_peak": b_turn * iters / elapsed / 1e9 / HBM_PEAK_GBPS,
            "csr_equivalent_bytes": b_turn_csr,
            "csr_equivalent_frac_of_peak": b_turn_csr * iters / elapsed / 1e9 / HBM_PEAK_GBPS,
        },
        "boundary": {
            "first_set_matrix_s": t_first_matrix, "refresh_set_matrix_s": t_refresh_matrix,
            "solve_incl_pcie_s": t_e2e,
            "iters_per_sec_incl_pcie": (p_e2e.n_iterations - 1) / t_e2e,
            # the legs of that one plug-in call as the library timed them (ogl_perf, lduLduBase.H:296-305's block):
            # coefficients up (upper [+ lower] + diag through the pinned ring, then the gathers), b [+ x] up,
            # the Krylov loop, x down
            "update_matrix_ms": p_e2e.t_update_matrix_ms, "upload_ms": p_e2e.t_upload_ms,
            "solve_ms": p_e2e.t_solve_ms, "copy_back_ms": p_e2e.t_copy_back_ms,
            "h2d_GBps": 8.0 * (case.n_faces * (1 if case.lower is None else 2) + N) / 1e9 /
                        max(1e-9, t_refresh_matrix),
            "d2h_GBps": 8.0 * N / 1e6 / max(1e-9, p_e2e.t_copy_back_ms),
            "momentum_components": components,
        },
    }

    # ---- CPU baseline (rank 0, N=1 only): the oracle on the same matrix ----------------------
    headline = (args.solver == "GKOCG" and args.precond in ("BJ", "none") and args.block_size == 1
                and not args.asym)
    if rank == 0 and world == 1 and args.cpu_iters != 0 and headline and not args.shuffle and not args.config:
        # a child process: thread placement must be fixed before an OpenMP runtime loads (this process
        # already carries torch's), and the oracle shares nothing with the GPU run
        env = {k: v for k, v in os.environ.items() if not k.startswith("OMP_")}
        base = [sys.executable, "-m", "oracle.cpu_baseline", "--edge", str(n), "--precond", args.precond,
                "--iters", str(args.iters), "--seq-iters", str(args.cpu_iters)]

        def child(extra, timeout):
            p = subprocess.run(base + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
            if p.returncode != 0:
                raise SystemExit("cpu_baseline leg failed:\n" + p.stderr[-2000:])
            return json.loads(p.stdout.strip().splitlines()[-1])
        # Which placement and thread count run the CG loop fastest on this host.  The triad probe (bound one per
        # core, or left to the scheduler) names a thread count; but a container with a CPU quota (cgroup cpu.max)
        # below its CPU count is throttled for most of every period when all cores run -- there a thread per quota
        # CPU, spread over the sockets and core complexes, is what a user would configure.  Decided by short trials
        # of the loop itself, all reported.
        probes = [child(["--probe", "--bind", str(bnd)], 300) for bnd in (1, 0)]
        best = max(probes, key=lambda q: q["GBps"])
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q == "max" else max(1, int(round(int(q) / int(per))))
        except (OSError, ValueError):
            pass
        candidates = [(best["bind"], best["threads"])]
        host_cpus = best["host_cpus"]
        if quota is not None and quota < host_cpus:
            candidates += [(2, t) for t in (quota, 2 * quota) if t <= host_cpus]
        trials = []
        for bnd, thr in candidates:
            t = child(["--omp-only", "--bind", str(bnd), "--threads", str(thr), "--seconds", "4"], 600)["omp"]
            trials.append({"bind": bnd, "threads_tried": t["iters_per_s_by_thread_count"], "iter_per_s": t["value"],
                           "threads": t["cores"]})
        pick = max(trials, key=lambda t: t["iter_per_s"])
        legs = child(["--bind", str(pick["bind"]), "--threads", str(pick["threads"])], 900)
        legs["omp"]["placement_probe"] = probes
        legs["omp"]["placement_trials"] = trials
        legs["omp"]["cpu_quota_cpus"] = quota
        out["cpu_baseline"] = legs["seq"]
        out["cpu_baseline_omp"] = legs["omp"]

    reg.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(_REAL_STDOUT, (json.dumps(out) + "\n").encode())


# The contract is ONE JSON line on stdout: gloo and RCCL print banners to file descriptor 1 from
# native code, so everything but the final line is sent to stderr.
_REAL_STDOUT = 1

if __name__ == "__main__":
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    main()
