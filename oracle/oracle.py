"""ctypes front-end of the CPU oracle (oracle/ogl_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package ``ogl_amd``.

Parity status (see ogl_oracle.h): LDU conversion pinned by the reference's gtest
vectors; Krylov arithmetic "parity unpinned" (Ginkgo is not in the reference tree).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

label = np.int32
scalar = np.float64
_LP = C.POINTER(C.c_int32)
_SP = C.POINTER(C.c_double)

IFACE_PROCESSOR = 0
IFACE_CYCLIC = 1
REDUCE_SEQUENTIAL = 0
REDUCE_BLOCKED = 1
REDUCE_EXACT = 2


def build(force=False):
    """Compile liboracle.so / liboracle_omp.so with oracle/Makefile (gcc)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


class _CIface(C.Structure):
    _fields_ = [("kind", C.c_int), ("neighb_proc", C.c_int32), ("neighb_patch", C.c_int32),
                ("size", C.c_int32), ("face_cells", _LP), ("bou_coeffs", _SP)]


EXCHANGE_FN = C.CFUNCTYPE(None, C.c_void_p, _SP, _SP)
ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, _SP, C.c_int32)


class _CDistMatrix(C.Structure):
    _fields_ = [("n", C.c_int32), ("rowptr", _LP), ("cols", _LP), ("vals", _SP),
                ("n_halo", C.c_int32), ("nl_rowptr", _LP), ("nl_cols", _LP), ("nl_vals", _SP),
                ("n_send", C.c_int32), ("send_idxs", _LP),
                ("exchange", EXCHANGE_FN), ("allreduce", ALLREDUCE_FN),
                ("user", C.c_void_p), ("global_n", C.c_int64)]


class _CCriterion(C.Structure):
    _fields_ = [("tolerance", C.c_double), ("rel_tol", C.c_double), ("min_iter", C.c_int32),
                ("max_iter", C.c_int32), ("frequency", C.c_int32), ("export_res", C.c_int)]


class _CState(C.Structure):
    _fields_ = [("init_residual", C.c_double), ("residual", C.c_double),
                ("norm_factor", C.c_double), ("iter", C.c_int32), ("n_evals", C.c_int32),
                ("history", _SP)]


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.orc_count_interface_nnz.restype = C.c_int32
    lib.orc_create_communication_pattern.restype = C.c_int32
    for f in ("orc_dot", "orc_norm1", "orc_sum", "orc_compute_normfactor"):
        getattr(lib, f).restype = C.c_double
    for f in ("orc_cg", "orc_bicgstab", "orc_cg_omp", "orc_cg_p", "orc_bicgstab_p"):
        getattr(lib, f).restype = C.c_int32
    lib.orc_omp_max_threads.restype = C.c_int
    lib.orc_cg_omp_timed.restype = C.c_int32
    lib.orc_stream_triad_omp.restype = C.c_double
    return lib


_libs = {}


def lib(omp=False):
    key = "liboracle_omp.so" if omp else "liboracle.so"
    if key not in _libs:
        _libs[key] = _load(key)
    return _libs[key]


def use_native_omp_build():
    """CPU-baseline leg only: rebuild the OpenMP variant for THIS host's CPU (-march=native) into the
    temporary directory and use it as lib(True) from now on.  The in-tree liboracle_omp.so is generic
    x86-64 because it travels between machines; a baseline timed on a host should be allowed that
    host's vector units.  Returns a description of what is in use."""
    import tempfile
    if "liboracle_omp.so" in _libs:
        return "generic x86-64 build (already loaded)"
    out = os.path.join(tempfile.gettempdir(), f"liboracle_omp_native_{os.getpid()}.so")
    cmd = ["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c11", "-fopenmp",
           "-shared", "-o", out, os.path.join(_HERE, "ogl_oracle.c"), "-lm"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        native = C.CDLL(out)
    except (subprocess.CalledProcessError, OSError):
        return "generic x86-64 build (native build failed)"
    finally:
        if os.path.exists(out):
            os.unlink(out)                      # the mapping stays valid; nothing is left behind
    generic = _load("liboracle_omp.so")         # restype declarations
    for f in ("orc_cg_omp_timed", "orc_omp_max_threads"):
        getattr(native, f).restype = getattr(generic, f).restype
    native.orc_stream_triad_omp.restype = C.c_double
    _libs["liboracle_omp.so"] = native
    return "gcc -O3 -march=native build for this host"


def _l(a):
    a = np.ascontiguousarray(a, dtype=label)
    return a, a.ctypes.data_as(_LP)


def _s(a):
    a = np.ascontiguousarray(a, dtype=scalar)
    return a, a.ctypes.data_as(_SP)


class Iface:
    """Plain view of one lduInterfaceField (what HostMatrix.C reads from it)."""

    def __init__(self, kind, face_cells, bou_coeffs, neighb_proc=-1, neighb_patch=-1):
        self.kind = kind
        self.face_cells = np.ascontiguousarray(face_cells, dtype=label)
        self.bou_coeffs = np.ascontiguousarray(bou_coeffs, dtype=scalar)
        self.neighb_proc = neighb_proc
        self.neighb_patch = neighb_patch
        assert self.face_cells.shape == self.bou_coeffs.shape


def _ifaces(ifaces):
    arr = (_CIface * max(1, len(ifaces)))()
    for i, f in enumerate(ifaces):
        arr[i] = _CIface(f.kind, f.neighb_proc, f.neighb_patch, f.face_cells.size,
                         f.face_cells.ctypes.data_as(_LP), f.bou_coeffs.ctypes.data_as(_SP))
    return arr, C.c_int32(len(ifaces))


# ---------------------------------------------------------------- HostMatrixFreeFunctions.C

def init_local_sparsity(nrows, upper, lower, symmetric=True):
    upper, pu = _l(upper)
    lower, pl = _l(lower)
    F = upper.size
    nnz = nrows + 2 * F
    rows, cols, perm = (np.zeros(nnz, label) for _ in range(3))
    lib().orc_init_local_sparsity(C.c_int32(nrows), C.c_int32(F), C.c_int(bool(symmetric)), pu, pl,
                                  rows.ctypes.data_as(_LP), cols.ctypes.data_as(_LP),
                                  perm.ctypes.data_as(_LP))
    return rows, cols, perm


def symmetric_update(permute, scale, diag, upper):
    permute, pp = _l(permute)
    diag, pd = _s(diag)
    upper, pu = _s(upper)
    out = np.zeros(permute.size, scalar)
    lib().orc_symmetric_update(C.c_int32(permute.size), C.c_int32(upper.size), pp,
                               C.c_double(scale), pd, pu, out.ctypes.data_as(_SP))
    return out


def non_symmetric_update(permute, scale, diag, upper, lower):
    permute, pp = _l(permute)
    diag, pd = _s(diag)
    upper, pu = _s(upper)
    lower, pl = _s(lower)
    out = np.zeros(permute.size, scalar)
    lib().orc_non_symmetric_update(C.c_int32(permute.size), C.c_int32(upper.size), pp,
                                   C.c_double(scale), pd, pu, pl, out.ctypes.data_as(_SP))
    return out


def symmetric_update_w_interface(permute, scale, diag, upper, iface):
    permute, pp = _l(permute)
    diag, pd = _s(diag)
    upper, pu = _s(upper)
    iface, pi = _s(iface)
    out = np.zeros(permute.size, scalar)
    lib().orc_symmetric_update_w_interface(C.c_int32(permute.size), C.c_int32(diag.size),
                                           C.c_int32(upper.size), pp, C.c_double(scale), pd, pu,
                                           pi, out.ctypes.data_as(_SP))
    return out


def non_symmetric_update_w_interface(permute, scale, diag, upper, lower, iface):
    permute, pp = _l(permute)
    diag, pd = _s(diag)
    upper, pu = _s(upper)
    lower, pl = _s(lower)
    iface, pi = _s(iface)
    out = np.zeros(permute.size, scalar)
    lib().orc_non_symmetric_update_w_interface(C.c_int32(permute.size), C.c_int32(diag.size),
                                               C.c_int32(upper.size), pp, C.c_double(scale), pd,
                                               pu, pl, pi, out.ctypes.data_as(_SP))
    return out


# ---------------------------------------------------------------- HostMatrix.C

def count_interface_nnz(ifaces, proc):
    arr, n = _ifaces(ifaces)
    return int(lib().orc_count_interface_nnz(arr, n, C.c_int(bool(proc))))


def collect_interface_coeffs(ifaces, local):
    arr, n = _ifaces(ifaces)
    out = np.zeros(count_interface_nnz(ifaces, not local), scalar)
    lib().orc_collect_interface_coeffs(arr, n, C.c_int(bool(local)), out.ctypes.data_as(_SP))
    return out


def create_communication_pattern(ifaces):
    arr, n = _ifaces(ifaces)
    ids = np.zeros(max(1, len(ifaces)), label)
    sizes = np.zeros(max(1, len(ifaces)), label)
    send = np.zeros(count_interface_nnz(ifaces, True), label)
    k = lib().orc_create_communication_pattern(arr, n, ids.ctypes.data_as(_LP),
                                               sizes.ctypes.data_as(_LP),
                                               send.ctypes.data_as(_LP))
    return ids[:k].copy(), sizes[:k].copy(), send


def init_non_local_sparsity(ifaces):
    arr, n = _ifaces(ifaces)
    nnz = count_interface_nnz(ifaces, True)
    rows, cols, perm = (np.zeros(nnz, label) for _ in range(3))
    lib().orc_init_non_local_sparsity(arr, n, rows.ctypes.data_as(_LP), cols.ctypes.data_as(_LP),
                                      perm.ctypes.data_as(_LP))
    return rows, cols, perm


def init_local_sparsity_pattern(nrows, upper, lower, symmetric, ifaces):
    upper, pu = _l(upper)
    lower, pl = _l(lower)
    arr, n = _ifaces(ifaces)
    nnz = nrows + 2 * upper.size + count_interface_nnz(ifaces, False)
    rows, cols, perm = (np.zeros(nnz, label) for _ in range(3))
    lib().orc_init_local_sparsity_pattern(C.c_int32(nrows), C.c_int32(upper.size),
                                          C.c_int(bool(symmetric)), pu, pl, arr, n,
                                          rows.ctypes.data_as(_LP), cols.ctypes.data_as(_LP),
                                          perm.ctypes.data_as(_LP))
    return rows, cols, perm


def update_local_matrix_data(diag, upper, lower, ifaces, permute, host_path=False, scaling=1.0):
    diag, pd = _s(diag)
    upper, pu = _s(upper)
    sym = lower is None
    lower, pl = _s(upper if sym else lower)
    permute, pp = _l(permute)
    arr, n = _ifaces(ifaces)
    out = np.zeros(permute.size, scalar)
    if host_path:
        lib().orc_update_local_matrix_data_host(C.c_int32(diag.size), C.c_int32(upper.size),
                                                C.c_int(sym), C.c_double(scaling), pd, pu, pl, arr,
                                                n, pp, C.c_int32(permute.size),
                                                out.ctypes.data_as(_SP))
    else:
        lib().orc_update_local_matrix_data(C.c_int32(diag.size), C.c_int32(upper.size),
                                           C.c_int(sym), pd, pu, pl, arr, n, pp,
                                           C.c_int32(permute.size), out.ctypes.data_as(_SP))
    return out


def update_non_local_matrix_data(ifaces, permute):
    permute, pp = _l(permute)
    arr, n = _ifaces(ifaces)
    out = np.zeros(permute.size, scalar)
    lib().orc_update_non_local_matrix_data(arr, n, pp, C.c_int32(permute.size),
                                           out.ctypes.data_as(_SP))
    return out


# ---------------------------------------------------------------- arithmetic

def rowptr_from_rows(nrows, rows):
    rows, pr = _l(rows)
    rp = np.zeros(nrows + 1, label)
    lib().orc_rowptr_from_rows(C.c_int32(nrows), C.c_int32(rows.size), pr, rp.ctypes.data_as(_LP))
    return rp


def permute_csr(rowptr, cols, vals, new_id):
    """P A P^T of a row-major CSR matrix: cell c becomes row/column new_id[c]; a row keeps its
    entries, ordered by NEW column (stable, so entries with equal columns keep their stored order).
    Not a reference function: it is the INPUT of a renumbered run (ogl_config.renumber) -- the
    product reports the permutation it chose, the oracle solves the system permuted by it, and the
    comparison stays bit for bit (oracle/README: the permutation is an explicit input)."""
    rowptr = np.asarray(rowptr, np.int64)
    cols = np.asarray(cols, np.int64)
    new_id = np.asarray(new_id, np.int64)
    n = rowptr.size - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    nr, nc = new_id[rows], new_id[cols]
    order = np.lexsort((np.arange(cols.size), nc, nr))
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, nr + 1, 1)
    return (np.cumsum(rp).astype(label), nc[order].astype(label),
            np.asarray(vals, scalar)[order].copy(), order)


def permute_non_local(nl_rows, nl_cols, nl_vals, new_id):
    """Rows of the non-local triplets renamed and row-sorted again (stable); columns address the
    receive buffer and do not change."""
    nr = np.asarray(new_id, np.int64)[np.asarray(nl_rows, np.int64)]
    order = np.argsort(nr, kind="stable")
    return (nr[order].astype(label), np.asarray(nl_cols)[order].astype(label),
            np.asarray(nl_vals, scalar)[order].copy(), order)


def set_reduction(mode, chunk_rows=512, omp=False):
    lib(omp).orc_set_reduction(C.c_int(mode), C.c_int32(chunk_rows))


def spmv(rowptr, cols, vals, x):
    rowptr, prp = _l(rowptr)
    cols, pc = _l(cols)
    vals, pv = _s(vals)
    x, px = _s(x)
    y = np.zeros(rowptr.size - 1, scalar)
    lib().orc_spmv(C.c_int32(y.size), prp, pc, pv, px, y.ctypes.data_as(_SP))
    return y


def spmv_adv(rowptr, cols, vals, alpha, x, beta, y):
    rowptr, prp = _l(rowptr)
    cols, pc = _l(cols)
    vals, pv = _s(vals)
    x, px = _s(x)
    y = np.array(y, dtype=scalar, copy=True)
    lib().orc_spmv_adv(C.c_int32(y.size), prp, pc, pv, C.c_double(alpha), px, C.c_double(beta),
                       y.ctypes.data_as(_SP))
    return y


def dot(a, b):
    a, pa = _s(a)
    b, pb = _s(b)
    return float(lib().orc_dot(C.c_int32(a.size), pa, pb))


def norm1(a):
    a, pa = _s(a)
    return float(lib().orc_norm1(C.c_int32(a.size), pa))


def vsum(a):
    a, pa = _s(a)
    return float(lib().orc_sum(C.c_int32(a.size), pa))


def jacobi_generate_scalar(rowptr, cols, vals):
    rowptr, prp = _l(rowptr)
    cols, pc = _l(cols)
    vals, pv = _s(vals)
    out = np.zeros(rowptr.size - 1, scalar)
    lib().orc_jacobi_generate_scalar(C.c_int32(out.size), prp, pc, pv, out.ctypes.data_as(_SP))
    return out


class _CPrecond(C.Structure):
    _fields_ = [("kind", C.c_int), ("inv_diag", _SP), ("n_blocks", C.c_int32),
                ("block_ptrs", _LP), ("blocks", _SP), ("stride", C.c_int32),
                ("w_rowptr", _LP), ("w_cols", _LP), ("w_vals", _SP),
                ("wt_rowptr", _LP), ("wt_cols", _LP), ("wt_vals", _SP), ("block_rows", _LP)]


class Precond:
    """Jacobi preconditioner object: scalar (maxBlockSize 1) or block (maxBlockSize k > 1)."""

    def __init__(self, rowptr, cols, vals, max_block_size=1, isai=None, sparsity_power=1, caller=None):
        """isai: None (Jacobi), "spd" (keyword ISAI) or "general" (keyword GISAI).
        caller = (rowptr0, cols0, vals0, new_id): (rowptr, cols, vals) is the system the backend renumbered by
        new_id (new_id[caller row] = row here) and the preconditioner is the one the reference generates on the
        matrix OpenFOAM hands over (Preconditioner.H:91-105, :225-241), expressed in the new numbering: block-Jacobi
        blocks are found and inverted on the caller's matrix and applied through new_id; ISAI(spd) takes the
        triangle by the caller's index."""
        if caller is not None and isai is None and int(max_block_size) > 1:
            rp0, c0, v0, new_id = caller
            base = Precond(rp0, c0, v0, max_block_size)
            self.__dict__.update(base.__dict__)
            self.block_rows, pbr = _l(new_id)
            self.c = _CPrecond(2, None, base.c.n_blocks, self.block_ptrs.ctypes.data_as(_LP),
                               self.blocks.ctypes.data_as(_SP), int(max_block_size), None, None, None, None, None,
                               None, pbr)
            return
        key = None
        if caller is not None and isai == "spd":
            old_of = np.empty(len(caller[3]), label)
            old_of[np.asarray(caller[3])] = np.arange(len(caller[3]), dtype=label)
            self.tri_key, key = _l(old_of)
        rowptr, prp = _l(rowptr)
        cols, pc = _l(cols)
        vals, pv = _s(vals)
        n = rowptr.size - 1
        self.max_block_size = int(max_block_size)
        if isai is not None:
            spd = C.c_int(isai == "spd")
            lib().orc_isai_generate_pk.restype = C.c_int32
            pw = C.c_int(int(sparsity_power))
            self.w_rowptr = np.zeros(n + 1, label)
            nnz = lib().orc_isai_generate_pk(C.c_int32(n), prp, pc, pv, spd, pw, key,
                                             self.w_rowptr.ctypes.data_as(_LP), None, None)
            if nnz < 0:
                raise ValueError("ISAI row wider than 2048")
            self.w_cols, self.w_vals = np.zeros(max(1, nnz), label), np.zeros(max(1, nnz), scalar)
            lib().orc_isai_generate_pk(C.c_int32(n), prp, pc, pv, spd, pw, key, self.w_rowptr.ctypes.data_as(_LP),
                                       self.w_cols.ctypes.data_as(_LP), self.w_vals.ctypes.data_as(_SP))
            self.wt_rowptr = np.zeros(n + 1, label)
            self.wt_cols, self.wt_vals = np.zeros_like(self.w_cols), np.zeros_like(self.w_vals)
            lib().orc_csr_transpose(C.c_int32(n), self.w_rowptr.ctypes.data_as(_LP),
                                    self.w_cols.ctypes.data_as(_LP), self.w_vals.ctypes.data_as(_SP),
                                    self.wt_rowptr.ctypes.data_as(_LP),
                                    self.wt_cols.ctypes.data_as(_LP),
                                    self.wt_vals.ctypes.data_as(_SP))
            self.c = _CPrecond(3 if isai == "spd" else 4, None, 0, None, None, 0,
                               self.w_rowptr.ctypes.data_as(_LP), self.w_cols.ctypes.data_as(_LP),
                               self.w_vals.ctypes.data_as(_SP), self.wt_rowptr.ctypes.data_as(_LP),
                               self.wt_cols.ctypes.data_as(_LP), self.wt_vals.ctypes.data_as(_SP), None)
        elif self.max_block_size == 1:
            self.inv_diag = jacobi_generate_scalar(rowptr, cols, vals)
            self.c = _CPrecond(1, self.inv_diag.ctypes.data_as(_SP), 0, None, None, 0, None, None,
                               None, None, None, None, None)
        else:
            k = self.max_block_size
            bp = np.zeros(n + 1, label)
            lib().orc_jacobi_find_blocks.restype = C.c_int32
            nb = lib().orc_jacobi_find_blocks(C.c_int32(n), prp, pc, C.c_int32(k),
                                              bp.ctypes.data_as(_LP))
            self.block_ptrs = np.ascontiguousarray(bp[:nb + 1])
            self.blocks = np.zeros(nb * k * k, scalar)
            lib().orc_jacobi_generate_blocks(C.c_int32(n), prp, pc, pv, C.c_int32(nb),
                                             self.block_ptrs.ctypes.data_as(_LP), C.c_int32(k),
                                             self.blocks.ctypes.data_as(_SP))
            self.c = _CPrecond(2, None, nb, self.block_ptrs.ctypes.data_as(_LP),
                               self.blocks.ctypes.data_as(_SP), k, None, None, None, None, None, None, None)


class DistMatrix:
    """local CSR (+ optional non-local CSR, halo exchange and all-reduce callbacks)."""

    def __init__(self, rowptr, cols, vals, nl_rowptr=None, nl_cols=None, nl_vals=None,
                 send_idxs=None, n_halo=0, exchange=None, allreduce=None, global_n=None):
        self.rowptr, _ = _l(rowptr)
        self.cols, _ = _l(cols)
        self.vals, _ = _s(vals)
        self.n = self.rowptr.size - 1
        self.n_halo = int(n_halo)
        z = np.zeros(1, label)
        self.nl_rowptr, _ = _l(nl_rowptr if nl_rowptr is not None else np.zeros(self.n + 1, label))
        self.nl_cols, _ = _l(nl_cols if nl_cols is not None else z)
        self.nl_vals, _ = _s(nl_vals if nl_vals is not None else np.zeros(1))
        self.send_idxs, _ = _l(send_idxs if send_idxs is not None else z)
        self.n_send = 0 if send_idxs is None else int(np.asarray(send_idxs).size)
        self._exchange = exchange
        self._allreduce = allreduce
        self.global_n = int(global_n if global_n is not None else self.n)

        def _ex(_user, send, recv):
            s = np.ctypeslib.as_array(send, shape=(max(1, self.n_send),))[:self.n_send]
            r = np.ctypeslib.as_array(recv, shape=(max(1, self.n_halo),))[:self.n_halo]
            r[:] = self._exchange(s.copy())

        def _ar(_user, v, n):
            a = np.ctypeslib.as_array(v, shape=(n,))
            a[:] = self._allreduce(a.copy())

        self._cb_ex = EXCHANGE_FN(_ex) if exchange is not None else EXCHANGE_FN()
        self._cb_ar = ALLREDUCE_FN(_ar) if allreduce is not None else ALLREDUCE_FN()
        self.c = _CDistMatrix(self.n, self.rowptr.ctypes.data_as(_LP),
                              self.cols.ctypes.data_as(_LP), self.vals.ctypes.data_as(_SP),
                              self.n_halo, self.nl_rowptr.ctypes.data_as(_LP),
                              self.nl_cols.ctypes.data_as(_LP), self.nl_vals.ctypes.data_as(_SP),
                              self.n_send, self.send_idxs.ctypes.data_as(_LP), self._cb_ex,
                              self._cb_ar, None, self.global_n)

    def apply(self, x):
        x, px = _s(x)
        y = np.zeros(self.n, scalar)
        lib().orc_dist_spmv(C.byref(self.c), px, y.ctypes.data_as(_SP))
        return y


class Result:
    def __init__(self, x, st, hist, export_res):
        self.x = x
        self.n_iterations = int(st.iter)       # number of check_impl calls (CG steps + 1)
        self.n_evals = int(st.n_evals)
        self.initial_residual = float(st.init_residual)
        self.final_residual = float(st.residual)
        self.norm_factor = float(st.norm_factor)
        self.history = hist[:st.iter].copy() if export_res else None


def adapt_criterion(min_iter, frequency, export_res, prev_solve_iters, adapt_min_iter=True,
                    relaxation_factor=0.6, norm_eval_limit=100, prev_rel_cost=0.0):
    mi, fr = C.c_int32(0), C.c_int32(0)
    lib().orc_adapt_criterion(C.c_int32(min_iter), C.c_int32(frequency), C.c_int(bool(export_res)),
                              C.c_int32(prev_solve_iters), C.c_int(bool(adapt_min_iter)),
                              C.c_double(relaxation_factor), C.c_int32(norm_eval_limit),
                              C.c_double(prev_rel_cost), C.byref(mi), C.byref(fr))
    return mi.value, fr.value


def compute_normfactor(A, r, x, b):
    r, pr = _s(r)
    x, px = _s(x)
    b, pb = _s(b)
    return float(lib().orc_compute_normfactor(C.byref(A.c), pr, px, pb))


def _solve(fn_name, A, b, x0, inv_diag, tolerance, rel_tol, min_iter, max_iter, frequency,
           export_res, omp_threads=None):
    b, pb = _s(b)
    x = np.array(x0, dtype=scalar, copy=True)
    pinv = None
    if isinstance(inv_diag, Precond):
        fn_name += "_p"
        pinv = C.byref(inv_diag.c)
    elif inv_diag is not None:
        inv_diag, pinv = _s(inv_diag)
    crit = _CCriterion(tolerance, rel_tol, min_iter, max_iter, frequency, int(bool(export_res)))
    # the criterion stops at the first evaluated check at or after max(maxIter, minIter)
    hist = np.zeros(2 * max(max_iter, min_iter) + frequency + 4, scalar)   # (x2: orc_bicgstab doubles maxIter)
    st = _CState()
    st.history = hist.ctypes.data_as(_SP)
    if omp_threads is None:
        getattr(lib(), fn_name)(C.byref(A.c), pb, x.ctypes.data_as(_SP), pinv, C.byref(crit),
                                C.byref(st))
    else:
        rc = lib(True).orc_cg_omp(C.byref(A.c), pb, x.ctypes.data_as(_SP), pinv, C.byref(crit),
                                  C.byref(st), C.c_int(omp_threads))
        if rc < 0:
            raise RuntimeError("orc_cg_omp unavailable")
    return Result(x, st, hist, export_res)


def cg(A, b, x0, inv_diag=None, tolerance=1e-6, rel_tol=1e-6, min_iter=0, max_iter=1000,
       frequency=1, export_res=True):
    """GKOCG semantics; defaults are the code's defaults (StoppingCriterion.H:165-169)."""
    return _solve("orc_cg", A, b, x0, inv_diag, tolerance, rel_tol, min_iter, max_iter, frequency,
                  export_res)


def bicgstab(A, b, x0, inv_diag=None, tolerance=1e-6, rel_tol=1e-6, min_iter=0, max_iter=1000,
             frequency=1, export_res=True):
    """GKOBiCGStab semantics: max_iter is doubled here like StoppingCriterion.H:188."""
    return _solve("orc_bicgstab", A, b, x0, inv_diag, tolerance, rel_tol, min_iter, 2 * max_iter,
                  frequency, export_res)


def gmres(A, b, x0, precond=None, tolerance=1e-6, rel_tol=1e-6, min_iter=0, max_iter=1000,
          frequency=1, export_res=True, krylov_dim=0):
    """GKOGMRES semantics; precond is a Precond object or None."""
    b, pb = _s(b)
    x = np.array(x0, dtype=scalar, copy=True)
    crit = _CCriterion(tolerance, rel_tol, min_iter, max_iter, frequency, int(bool(export_res)))
    hist = np.zeros(max(max_iter, min_iter) + frequency + 4, scalar)
    st = _CState()
    st.history = hist.ctypes.data_as(_SP)
    lib().orc_gmres_p.restype = C.c_int32
    lib().orc_gmres_p(C.byref(A.c), pb, x.ctypes.data_as(_SP),
                      C.byref(precond.c) if precond is not None else None, C.byref(crit),
                      C.byref(st), C.c_int32(krylov_dim))
    return Result(x, st, hist, export_res)


def cg_omp(A, b, x0, inv_diag=None, tolerance=1e-6, rel_tol=1e-6, min_iter=0, max_iter=1000,
           frequency=1, export_res=True, threads=0):
    return _solve("orc_cg_omp", A, b, x0, inv_diag, tolerance, rel_tol, min_iter, max_iter,
                  frequency, export_res, omp_threads=threads)


def bind_omp_threads():
    """Thread placement for the OpenMP baseline: one thread per core, neighbours close.  libgomp reads
    these when it is loaded, so call this BEFORE the first lib(True)."""
    if "liboracle_omp.so" in _libs:
        return False
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    return True


def cg_omp_timed(A, b, x0, inv_diag=None, tolerance=0.0, rel_tol=0.0, max_iter=100, threads=0):
    """(Result, t_setup_s, t_loop_s): set-up (allocation + first-touch copy, once per matrix) and the
    solve loop timed apart inside the C code."""
    b, pb = _s(b)
    x = np.array(x0, dtype=scalar, copy=True)
    pinv = None
    if inv_diag is not None:
        inv_diag, pinv = _s(inv_diag)
    crit = _CCriterion(tolerance, rel_tol, 0, max_iter, 1, 0)
    st = _CState()
    t_setup, t_loop = C.c_double(), C.c_double()
    rc = lib(True).orc_cg_omp_timed(C.byref(A.c), pb, x.ctypes.data_as(_SP), pinv, C.byref(crit),
                                    C.byref(st), C.c_int(threads), C.byref(t_setup), C.byref(t_loop))
    if rc < 0:
        raise RuntimeError("orc_cg_omp_timed unavailable")
    return Result(x, st, np.zeros(1), False), t_setup.value, t_loop.value


def cg_omp_phases():
    """Seconds the last cg_omp_timed spent in (x/r update + reductions, p update, SpMV + p.q)."""
    out = (C.c_double * 3)()
    lib(True).orc_cg_omp_phases(out)
    return float(out[0]), float(out[1]), float(out[2])


def stream_triad_omp(n, reps=5, threads=0):
    return float(lib(True).orc_stream_triad_omp(C.c_long(n), C.c_int(reps), C.c_int(threads)))


def omp_max_threads():
    return int(lib(True).orc_omp_max_threads())
