"""ctypes binding of include/ogl_amd.h (libogl_amd.so) -- the driver the tests and bench.py use.

This is plumbing over the C ABI, not a second implementation: every compute call lands in the
HIP library.  If the library is missing or no gfx950 device is visible the calls raise; there is
no CPU path behind them.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (OGL_AMD_LIB: another build of the same library, for A/B runs during development)
LIB_PATH = os.environ.get("OGL_AMD_LIB") or os.path.join(_HERE, "lib", "libogl_amd.so")

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_COMM, ERR_STATE, ERR_UNSUPPORTED, ERR_COMM_SELFTEST = -1, -2, -3, -4, -5, -6, -7
SOLVER_CG, SOLVER_BICGSTAB, SOLVER_GMRES = 0, 1, 2
PRECOND_NONE, PRECOND_BJ, PRECOND_ISAI, PRECOND_GISAI = 0, 1, 2, 3
FORMAT_COO, FORMAT_CSR, FORMAT_ELL = 0, 1, 2
RENUMBER_OFF, RENUMBER_ON, RENUMBER_AUTO = 0, 1, 2
IFACE_PROCESSOR, IFACE_CYCLIC = 0, 1
RCCL_ID_BYTES = 128
PEER_HANDLE_BYTES = 64

_LP = C.POINTER(C.c_int32)
_SP = C.POINTER(C.c_double)


class Config(C.Structure):
    _fields_ = [
        ("solver", C.c_int32), ("preconditioner", C.c_int32), ("max_block_size", C.c_int32),
        ("caching", C.c_int32), ("tolerance", C.c_double), ("rel_tol", C.c_double),
        ("max_iter", C.c_int32), ("min_iter", C.c_int32), ("eval_frequency", C.c_int32),
        ("norm_eval_limit", C.c_int32), ("relaxation_factor", C.c_double),
        ("adapt_min_iter", C.c_int32), ("matrix_format", C.c_int32), ("regenerate", C.c_int32),
        ("update_sys_matrix", C.c_int32), ("update_rhs", C.c_int32),
        ("update_init_guess", C.c_int32), ("scaling", C.c_double), ("reorder_on_host", C.c_int32),
        ("export_res", C.c_int32), ("verbose", C.c_int32), ("force_host_buffer", C.c_int32),
        ("ranks_per_gpu", C.c_int32), ("krylov_dim", C.c_int32), ("sparsity_power", C.c_int32),
        ("profile_kernels", C.c_int32), ("compress_indices", C.c_int32),
        ("renumber", C.c_int32), ("symmetric_half", C.c_int32),
    ]


class Interface(C.Structure):
    _fields_ = [("kind", C.c_int32), ("neighb_proc", C.c_int32), ("neighb_patch", C.c_int32),
                ("size", C.c_int32), ("face_cells", _LP), ("bou_coeffs", _SP)]


class LduView(C.Structure):
    _fields_ = [("n_cells", C.c_int32), ("n_faces", C.c_int32), ("lower_addr", _LP),
                ("upper_addr", _LP), ("diag", _SP), ("upper", _SP), ("lower", _SP),
                ("n_interfaces", C.c_int32), ("interfaces", C.POINTER(Interface)), ("cell_centres", _SP)]


class Perf(C.Structure):
    _fields_ = [("initial_residual", C.c_double), ("final_residual", C.c_double),
                ("n_iterations", C.c_int32), ("n_norm_evals", C.c_int32),
                ("norm_factor", C.c_double), ("t_update_matrix_ms", C.c_double),
                ("t_upload_ms", C.c_double), ("t_solve_ms", C.c_double),
                ("t_copy_back_ms", C.c_double), ("spmv_avg_ms", C.c_double),
                ("spmv_launches", C.c_int32), ("reserved0", C.c_int32), ("t_res_norm_us", C.c_double),
                ("n_global_rows", C.c_double)]


class MatrixDims(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("local_nnz", C.c_int32), ("non_local_nnz", C.c_int32),
                ("n_halo", C.c_int32), ("n_neighbours", C.c_int32), ("n_send", C.c_int32)]


class CommInfo(C.Structure):
    _fields_ = [("transport", C.c_int32), ("rank", C.c_int32), ("n_ranks", C.c_int32),
                ("ranks_seen", C.c_int32), ("peer_mesh", C.c_int32), ("device", C.c_int32)]


ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, _SP, C.c_int32)
EXCHANGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, _LP, _LP, _SP, _SP)

# every symbol include/ogl_amd.h declares
EXPORTED_SYMBOLS = [
    "ogl_last_error", "ogl_abi_version", "ogl_config_default", "ogl_registry_create",
    "ogl_registry_destroy", "ogl_registry_set_host_comm", "ogl_rccl_unique_id",
    "ogl_registry_rccl_ready", "ogl_registry_init_rccl", "ogl_registry_peer_handle", "ogl_registry_peer_connect",
    "ogl_registry_peer_disable", "ogl_solver_get_or_create", "ogl_solver_set_matrix", "ogl_solver_set_matrix_like",
    "ogl_solver_solve", "ogl_solver_history", "ogl_solver_export_system",
    "ogl_solver_get_property",
    "ogl_solver_set_property", "ogl_solver_apply_resident", "ogl_solver_upload_solution",
    "ogl_solver_upload_rhs", "ogl_solver_download_solution", "ogl_solver_spmv",
    "ogl_solver_time_spmv", "ogl_solver_reduce", "ogl_reduction_chunk_rows",
    "ogl_solver_matrix_dims", "ogl_solver_get_local_matrix", "ogl_solver_get_non_local_matrix",
    "ogl_solver_get_comm_pattern", "ogl_host_init_local_sparsity", "ogl_host_symmetric_update",
    "ogl_host_symmetric_update_w_interface", "ogl_host_non_symmetric_update_w_interface",
    "ogl_host_non_symmetric_update", "ogl_host_pattern", "ogl_host_adapt_criterion",
    "ogl_host_sell_check", "ogl_host_sym_check", "ogl_host_symx_check", "ogl_solver_get_renumbering", "ogl_host_rcm", "ogl_host_hilbert_order",
    "ogl_host_gather_sector_ratio", "ogl_host_pattern_renumbered",
    "ogl_host_addressing_fingerprint", "ogl_registry_comm_info", "ogl_memory_ledger_read", "ogl_registry_mem_info",
]


class MemoryLedger(C.Structure):
    """ogl_memory_ledger: what the library holds, process-wide (csrc/ledger.hpp)."""
    _fields_ = [(n, C.c_int64) for n in (
        "device_bytes", "device_blocks", "device_peak_bytes", "device_alloc_calls",
        "pinned_bytes", "pinned_blocks", "pinned_peak_bytes", "pinned_alloc_calls",
        "streams", "events", "graph_execs", "events_created", "graph_execs_created", "unknown_frees")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def memory_ledger() -> MemoryLedger:
    out = MemoryLedger()
    _check(lib().ogl_memory_ledger_read(C.byref(out)))
    return out


class OglError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libogl_amd status {status}: {message}")
        self.status = status


_lib = None


def lib():
    """Load libogl_amd.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (there is no fallback path)")
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _lib.ogl_last_error.restype = C.c_char_p
        _lib.ogl_registry_destroy.restype = None
        _lib.ogl_config_default.restype = None
        for name in ("ogl_host_init_local_sparsity", "ogl_host_symmetric_update",
                     "ogl_host_symmetric_update_w_interface",
                     "ogl_host_non_symmetric_update_w_interface", "ogl_host_non_symmetric_update",
                     "ogl_host_adapt_criterion"):
            getattr(_lib, name).restype = None
        _lib.ogl_host_gather_sector_ratio.restype = C.c_double
        _lib.ogl_host_addressing_fingerprint.restype = C.c_uint64
    return _lib


def _check(rc):
    if rc < 0:
        raise OglError(rc, lib().ogl_last_error().decode())
    return rc


def _l(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _s(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pl(a):
    return a.ctypes.data_as(_LP)


def _ps(a):
    return a.ctypes.data_as(_SP)


def default_config(**kw):
    """ogl_config with the reference code's defaults, overridden by keyword."""
    cfg = Config()
    lib().ogl_config_default(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise KeyError(k)
        setattr(cfg, k, v)
    return cfg


class LduArrays:
    """Keeps the numpy arrays of an lduMatrix view alive next to the C struct."""

    def __init__(self, case):
        self.lower_addr, self.upper_addr = _l(case.lower_addr), _l(case.upper_addr)
        self.diag, self.upper = _s(case.diag), _s(case.upper)
        self.lower = None if case.lower is None else _s(case.lower)
        self.ifaces = [(_l(f.face_cells), _s(f.bou_coeffs), f) for f in case.interfaces]
        self.c_ifaces = (Interface * max(1, len(self.ifaces)))()
        for i, (fc, bc, f) in enumerate(self.ifaces):
            self.c_ifaces[i] = Interface(f.kind, f.neighb_proc, f.neighb_patch, fc.size, _pl(fc),
                                         _ps(bc))
        centres = getattr(case, "centres", None)
        self.centres = None if centres is None else _s(np.asarray(centres).reshape(case.n_cells, 3))
        self.view = LduView(case.n_cells, self.upper_addr.size, _pl(self.lower_addr),
                            _pl(self.upper_addr), _ps(self.diag), _ps(self.upper),
                            None if self.lower is None else _ps(self.lower), len(self.ifaces),
                            self.c_ifaces, None if self.centres is None else _ps(self.centres))


class Registry:
    """objectRegistry analogue: owns the device context and the per-field solvers."""

    def __init__(self, device_id=-1, hip_stream=None):
        self._h = C.c_void_p()
        _check(lib().ogl_registry_create(C.byref(self._h), C.c_int(device_id),
                                         C.c_void_p(hip_stream)))
        self._cbs = None

    def close(self):
        if self._h:
            lib().ogl_registry_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_host_comm(self, rank, n_ranks, allreduce, exchange):
        """allreduce(np.ndarray) -> np.ndarray ; exchange(neighbours, counts, send) -> recv."""
        def _ar(_u, v, n):
            a = np.ctypeslib.as_array(v, shape=(n,))
            a[:] = allreduce(a.copy())

        def _ex(_u, nn, nbr, cnt, send, recv):
            nb = np.ctypeslib.as_array(nbr, shape=(nn,)).copy()
            ct = np.ctypeslib.as_array(cnt, shape=(nn,)).copy()
            tot = int(ct.sum())
            s = np.ctypeslib.as_array(send, shape=(tot,)).copy()
            np.ctypeslib.as_array(recv, shape=(tot,))[:] = exchange(nb, ct, s)

        self._cbs = (ALLREDUCE_FN(_ar), EXCHANGE_FN(_ex))
        _check(lib().ogl_registry_set_host_comm(self._h, rank, n_ranks, self._cbs[0], self._cbs[1],
                                                None))

    def rccl_ready(self):
        """Local check to agree on BEFORE the collective init_rccl (no time-out in there)."""
        return lib().ogl_registry_rccl_ready(self._h) == OK

    def init_rccl(self, rank, n_ranks, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, RCCL_ID_BYTES)
        _check(lib().ogl_registry_init_rccl(self._h, rank, n_ranks, buf))

    def peer_handle(self) -> bytes:
        """64-byte IPC handle of this rank's all-reduce mailbox (to be all-gathered by the host)."""
        buf = C.create_string_buffer(PEER_HANDLE_BYTES)
        _check(lib().ogl_registry_peer_handle(self._h, buf))
        return buf.raw

    def peer_connect(self, rank, n_ranks, handles):
        """Collective: map the other ranks' mailboxes and self-test the peer-write all-reduce."""
        blob = b"".join(handles)
        assert len(blob) == PEER_HANDLE_BYTES * n_ranks
        buf = C.create_string_buffer(blob, len(blob))
        _check(lib().ogl_registry_peer_connect(self._h, rank, n_ranks, buf))

    def mem_info(self):
        """(free, total) bytes of the registry's device as the driver sees them (hipMemGetInfo)."""
        free, total = C.c_int64(), C.c_int64()
        _check(lib().ogl_registry_mem_info(self._h, C.byref(free), C.byref(total)))
        return free.value, total.value

    def comm_info(self):
        info = CommInfo()
        _check(lib().ogl_registry_comm_info(self._h, C.byref(info)))
        return info

    def peer_disable(self):
        _check(lib().ogl_registry_peer_disable(self._h))

    def solver(self, field_name, cfg):
        return Solver(self, field_name, cfg)


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(RCCL_ID_BYTES)
    _check(lib().ogl_rccl_unique_id(buf))
    return buf.raw


class Solver:
    """One GKOCG / GKOBiCGStab / GKOGMRES construction (lookup-or-create by field name)."""

    def __init__(self, registry, field_name, cfg):
        self.registry = registry
        self.cfg = cfg
        self._h = C.c_void_p()
        _check(lib().ogl_solver_get_or_create(registry._h, field_name.encode(), C.byref(cfg),
                                              C.byref(self._h)))
        self._ldu = None

    def set_matrix(self, case, like=None):
        """like: another Solver of this registry whose device copy of upper / lower may be taken instead of uploading
        them again (ogl_solver_set_matrix_like: momentum components; `case` must then be LduArrays built on the very
        host arrays `like` uploaded from)."""
        self._ldu = case if isinstance(case, LduArrays) else LduArrays(case)
        if like is None:
            _check(lib().ogl_solver_set_matrix(self._h, C.byref(self._ldu.view)))
        else:
            _check(lib().ogl_solver_set_matrix_like(self._h, C.byref(self._ldu.view), like._h))
        return self

    def solve(self, source, psi, inplace=False):
        """Returns (psi_out, Perf).  psi is the initial guess (honoured per updateInitGuess).  inplace: psi
        (a contiguous float64 array) is overwritten with the solution, as the C ABI does -- no copy on the
        Python side (bench.py's PCIe-inclusive timing)."""
        b = _s(source)
        if inplace:
            assert isinstance(psi, np.ndarray) and psi.dtype == np.float64 and psi.flags.c_contiguous
        x = psi if inplace else np.array(psi, dtype=np.float64, copy=True)
        perf = Perf()
        _check(lib().ogl_solver_solve(self._h, _ps(b), _ps(x), C.byref(perf)))
        return x, perf

    def history(self, capacity=1 << 20):
        out = np.zeros(capacity, np.float64)
        n = _check(lib().ogl_solver_history(self._h, _ps(out), capacity))
        return out[:n].copy()

    def export_system(self, directory):
        os.makedirs(directory, exist_ok=True)
        _check(lib().ogl_solver_export_system(self._h, directory.encode()))

    def get_property(self, key):
        v = C.c_double()
        _check(lib().ogl_solver_get_property(self._h, key.encode(), C.byref(v)))
        return v.value

    def set_property(self, key, value):
        _check(lib().ogl_solver_set_property(self._h, key.encode(), C.c_double(value)))

    def apply_resident(self):
        perf = Perf()
        _check(lib().ogl_solver_apply_resident(self._h, C.byref(perf)))
        return perf

    def upload_solution(self, psi=None):
        p = None if psi is None else _ps(_s(psi))
        _check(lib().ogl_solver_upload_solution(self._h, p))

    def upload_rhs(self, source=None):
        p = None if source is None else _ps(_s(source))
        _check(lib().ogl_solver_upload_rhs(self._h, p))

    def download_solution(self):
        out = np.zeros(self.dims().n_rows, np.float64)
        _check(lib().ogl_solver_download_solution(self._h, _ps(out)))
        return out

    def spmv(self, x):
        x = _s(x)
        y = np.zeros_like(x)
        _check(lib().ogl_solver_spmv(self._h, _ps(x), _ps(y)))
        return y

    def time_spmv(self, repeats):
        ms = C.c_double()
        _check(lib().ogl_solver_time_spmv(self._h, repeats, C.byref(ms)))
        return ms.value

    def reduce(self, op, a, b=None):
        a = _s(a)
        out = C.c_double()
        pb = None
        if b is not None:
            b = _s(b)
            pb = _ps(b)
        _check(lib().ogl_solver_reduce(self._h, {"dot": 0, "norm1": 1, "sum": 2}[op], _ps(a), pb,
                                       C.byref(out)))
        return out.value

    def dims(self):
        d = MatrixDims()
        _check(lib().ogl_solver_matrix_dims(self._h, C.byref(d)))
        return d

    def local_matrix(self):
        d = self.dims()
        rp = np.zeros(d.n_rows + 1, np.int32)
        cols = np.zeros(d.local_nnz, np.int32)
        mp = np.zeros(d.local_nnz, np.int32)
        vals = np.zeros(d.local_nnz, np.float64)
        _check(lib().ogl_solver_get_local_matrix(self._h, _pl(rp), _pl(cols), _pl(mp), _ps(vals)))
        return rp, cols, mp, vals

    def non_local_matrix(self):
        d = self.dims()
        rows, cols, mp = (np.zeros(d.non_local_nnz, np.int32) for _ in range(3))
        vals = np.zeros(d.non_local_nnz, np.float64)
        _check(lib().ogl_solver_get_non_local_matrix(self._h, _pl(rows), _pl(cols), _pl(mp),
                                                     _ps(vals)))
        return rows, cols, mp, vals

    def renumbering(self):
        """new_id (cell c of the lduMatrix is row new_id[c] on the device) or None if the caller's
        numbering is in use."""
        new_id = np.zeros(self.dims().n_rows, np.int32)
        return new_id if _check(lib().ogl_solver_get_renumbering(self._h, _pl(new_id))) == 1 else None

    def comm_pattern(self):
        d = self.dims()
        ids, sizes = np.zeros(d.n_neighbours, np.int32), np.zeros(d.n_neighbours, np.int32)
        send = np.zeros(d.n_send, np.int32)
        _check(lib().ogl_solver_get_comm_pattern(self._h, _pl(ids), _pl(sizes), _pl(send)))
        return ids, sizes, send


# ------------------------------------------------------------------ pure host logic (no GPU)

def host_init_local_sparsity(nrows, upper, lower, symmetric=True):
    upper, lower = _l(upper), _l(lower)
    nnz = nrows + 2 * upper.size
    rows, cols, perm = (np.zeros(nnz, np.int32) for _ in range(3))
    lib().ogl_host_init_local_sparsity(C.c_int32(nrows), C.c_int32(upper.size),
                                       C.c_int(bool(symmetric)), _pl(upper), _pl(lower), _pl(rows),
                                       _pl(cols), _pl(perm))
    return rows, cols, perm


def host_symmetric_update(permute, scale, diag, upper):
    permute, diag, upper = _l(permute), _s(diag), _s(upper)
    out = np.zeros(permute.size)
    lib().ogl_host_symmetric_update(C.c_int32(permute.size), C.c_int32(upper.size), _pl(permute),
                                    C.c_double(scale), _ps(diag), _ps(upper), _ps(out))
    return out


def host_non_symmetric_update(permute, scale, diag, upper, lower):
    permute, diag, upper, lower = _l(permute), _s(diag), _s(upper), _s(lower)
    out = np.zeros(permute.size)
    lib().ogl_host_non_symmetric_update(C.c_int32(permute.size), C.c_int32(upper.size),
                                        _pl(permute), C.c_double(scale), _ps(diag), _ps(upper),
                                        _ps(lower), _ps(out))
    return out


def host_symmetric_update_w_interface(permute, scale, diag, upper, iface):
    permute, diag, upper, iface = _l(permute), _s(diag), _s(upper), _s(iface)
    out = np.zeros(permute.size)
    lib().ogl_host_symmetric_update_w_interface(C.c_int32(permute.size), C.c_int32(diag.size),
                                                C.c_int32(upper.size), _pl(permute),
                                                C.c_double(scale), _ps(diag), _ps(upper),
                                                _ps(iface), _ps(out))
    return out


def host_non_symmetric_update_w_interface(permute, scale, diag, upper, lower, iface):
    permute, diag, upper, lower, iface = _l(permute), _s(diag), _s(upper), _s(lower), _s(iface)
    out = np.zeros(permute.size)
    lib().ogl_host_non_symmetric_update_w_interface(
        C.c_int32(permute.size), C.c_int32(diag.size), C.c_int32(upper.size), _pl(permute),
        C.c_double(scale), _ps(diag), _ps(upper), _ps(lower), _ps(iface), _ps(out))
    return out


def host_pattern(case):
    """(dims, local (rows, cols, map), non-local (rows, cols, map), (ids, sizes, send_idxs))."""
    arr = case if isinstance(case, LduArrays) else LduArrays(case)
    d = MatrixDims()
    none = None
    _check(lib().ogl_host_pattern(C.byref(arr.view), C.byref(d), none, none, none, none, none,
                                  none, none, none, none))
    loc = [np.zeros(d.local_nnz, np.int32) for _ in range(3)]
    nl = [np.zeros(d.non_local_nnz, np.int32) for _ in range(3)]
    ids, sizes = np.zeros(d.n_neighbours, np.int32), np.zeros(d.n_neighbours, np.int32)
    send = np.zeros(d.n_send, np.int32)
    _check(lib().ogl_host_pattern(C.byref(arr.view), C.byref(d), *[_pl(a) for a in loc],
                                  *[_pl(a) for a in nl], _pl(ids), _pl(sizes), _pl(send)))
    return d, tuple(loc), tuple(nl), (ids, sizes, send)


def host_pattern_renumbered(case, mode=RENUMBER_ON, compress_indices=1):
    """host_pattern in the numbering config `renumber` = mode chooses, plus (renumbered, new_id)."""
    arr = case if isinstance(case, LduArrays) else LduArrays(case)
    d = MatrixDims()
    none = None
    _check(lib().ogl_host_pattern_renumbered(C.byref(arr.view), C.c_int32(mode),
                                             C.c_int32(compress_indices), C.byref(d), none, none,
                                             none, none, none, none, none, none, none, none))
    loc = [np.zeros(d.local_nnz, np.int32) for _ in range(3)]
    nl = [np.zeros(d.non_local_nnz, np.int32) for _ in range(3)]
    ids, sizes = np.zeros(d.n_neighbours, np.int32), np.zeros(d.n_neighbours, np.int32)
    send = np.zeros(d.n_send, np.int32)
    new_id = np.zeros(d.n_rows, np.int32)
    rc = _check(lib().ogl_host_pattern_renumbered(
        C.byref(arr.view), C.c_int32(mode), C.c_int32(compress_indices), C.byref(d),
        *[_pl(a) for a in loc], *[_pl(a) for a in nl], _pl(ids), _pl(sizes), _pl(send), _pl(new_id)))
    return d, tuple(loc), tuple(nl), (ids, sizes, send), (rc == 1, new_id)


def host_addressing_fingerprint(case):
    arr = case if isinstance(case, LduArrays) else LduArrays(case)
    return int(lib().ogl_host_addressing_fingerprint(C.byref(arr.view)))


def host_rcm(row_ptrs, cols):
    """Reverse Cuthill-McKee order of a row-major pattern: new_id[old] = new."""
    rp, cc = _l(row_ptrs), _l(cols)
    new_id = np.zeros(len(rp) - 1, np.int32)
    _check(lib().ogl_host_rcm(C.c_int32(len(rp) - 1), _pl(rp), _pl(cc), _pl(new_id)))
    return new_id


def host_hilbert_order(centres):
    """Cells along the Hilbert curve through their centres: new_id[old] = new."""
    c = _s(np.asarray(centres).reshape(-1, 3))
    new_id = np.zeros(c.shape[0], np.int32)
    _check(lib().ogl_host_hilbert_order(C.c_int32(c.shape[0]), _ps(c), _pl(new_id)))
    return new_id


def host_gather_sector_ratio(row_ptrs, cols, new_id=None):
    rp, cc = _l(row_ptrs), _l(cols)
    ni = None if new_id is None else _l(new_id)
    return float(lib().ogl_host_gather_sector_ratio(C.c_int32(len(rp) - 1), _pl(rp), _pl(cc),
                                                    None if ni is None else _pl(ni)))


def host_sell_check(row_ptrs, cols):
    """Index-compressed chunked ELL layout of a CSR pattern: (qualifies, padded_slots, dict_entries,
    code_bytes); raises if the layout does not decode back to the pattern."""
    rp, cc = _l(row_ptrs), _l(cols)
    stats = (C.c_int64 * 8)()
    _check(lib().ogl_host_sell_check(C.c_int32(len(rp) - 1), rp.ctypes.data_as(C.c_void_p),
                                     cc.ctypes.data_as(C.c_void_p), stats))
    return bool(stats[0]), int(stats[1]), int(stats[2]), int(stats[3])


def host_sym_check(row_ptrs, cols):
    """Half storage of a symmetric matrix on a banded pattern: (qualifies, distances from the diagonal
    incl. 0, plane slots, slots in use); raises if walking the layout does not reproduce the pattern."""
    rp, cc = _l(row_ptrs), _l(cols)
    stats = (C.c_int64 * 8)()
    _check(lib().ogl_host_sym_check(C.c_int32(len(rp) - 1), rp.ctypes.data_as(C.c_void_p),
                                    cc.ctypes.data_as(C.c_void_p), stats))
    return bool(stats[0]), [int(stats[2 + j]) for j in range(int(stats[1]))], int(stats[6]), int(stats[7])


def host_symx_check(row_ptrs, cols):
    """(qualifies, plane_slots, planar_entries, explicit_entries, chunks_with_explicit, chunks) of the half storage
    with per-chunk distances and explicit exceptions; raises when the layout does not decode to the input."""
    rp = np.ascontiguousarray(row_ptrs, np.int32)
    cc = np.ascontiguousarray(cols, np.int32)
    st = (C.c_int64 * 8)()
    _check(lib().ogl_host_symx_check(C.c_int32(len(rp) - 1), rp.ctypes.data_as(C.c_void_p),
                                     cc.ctypes.data_as(C.c_void_p), st))
    return bool(st[0]), int(st[1]), int(st[2]), int(st[3]), int(st[4]), int(st[5])


def host_symx_kernels(row_ptrs, cols):
    """(every chunk takes the pair-load instantiation, chunks the general kernel takes) of the same layout."""
    rp = np.ascontiguousarray(row_ptrs, np.int32)
    cc = np.ascontiguousarray(cols, np.int32)
    st = (C.c_int64 * 8)()
    _check(lib().ogl_host_symx_check(C.c_int32(len(rp) - 1), rp.ctypes.data_as(C.c_void_p),
                                     cc.ctypes.data_as(C.c_void_p), st))
    return bool(st[6]), int(st[7])


def host_sell_modes(row_ptrs, cols):
    """(qualifies, chunks in 16-bit delta mode, chunks in 32-bit column mode)."""
    rp, cc = _l(row_ptrs), _l(cols)
    stats = (C.c_int64 * 8)()
    _check(lib().ogl_host_sell_check(C.c_int32(len(rp) - 1), rp.ctypes.data_as(C.c_void_p),
                                     cc.ctypes.data_as(C.c_void_p), stats))
    return bool(stats[0]), int(stats[4]), int(stats[5])


def host_sell_read_slots(row_ptrs, cols):
    """(qualifies, allocated value slots, value slots the kernel reads)."""
    rp, cc = _l(row_ptrs), _l(cols)
    stats = (C.c_int64 * 8)()
    _check(lib().ogl_host_sell_check(C.c_int32(len(rp) - 1), rp.ctypes.data_as(C.c_void_p),
                                     cc.ctypes.data_as(C.c_void_p), stats))
    return bool(stats[0]), int(stats[1]), int(stats[6])


def host_sell_spilled(row_ptrs, cols):
    """(qualifies, value slots read, entries spilled into the second pass)."""
    rp, cc = _l(row_ptrs), _l(cols)
    stats = (C.c_int64 * 8)()
    _check(lib().ogl_host_sell_check(C.c_int32(len(rp) - 1), rp.ctypes.data_as(C.c_void_p),
                                     cc.ctypes.data_as(C.c_void_p), stats))
    return bool(stats[0]), int(stats[6]), int(stats[7])


def host_adapt_criterion(cfg, prev_solve_iters, prev_rel_cost):
    mi, fr = C.c_int32(), C.c_int32()
    lib().ogl_host_adapt_criterion(C.byref(cfg), C.c_int32(prev_solve_iters),
                                   C.c_double(prev_rel_cost), C.byref(mi), C.byref(fr))
    return mi.value, fr.value
