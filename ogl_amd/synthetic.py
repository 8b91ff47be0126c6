"""Synthetic lduMatrix inputs (numpy only): the 7-point Poisson box of BASELINE.md §3.

Cells are numbered x-fastest; internal faces are emitted in OpenFOAM's upper-triangular
order (owner ascending, neighbour ascending within an owner): for cell c the faces to
c+1, c+nx, c+nx*ny, in that order.  ``upper[f] = -1``,
``diag[i] = (#neighbours of i in the GLOBAL mesh) + 1e-3 * (1 + (gi mod 7) / 7)`` with gi the
global cell index, so a decomposed case assembles exactly the global matrix.

A decomposed case carries OpenFOAM-style coupled interfaces:
  processor patch : faceCells + neighbProcNo + bouCoeffs (= -offdiag, see HostMatrix.C:204)
  cyclic patch    : faceCells + neighbPatchID + bouCoeffs
ordered by ascending neighbour rank, both sides enumerating the shared faces identically
(what HostMatrix.C:251-306 / :412-436 rely on).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

IFACE_PROCESSOR = 0
IFACE_CYCLIC = 1


@dataclass
class Interface:
    kind: int
    face_cells: np.ndarray           # int32
    bou_coeffs: np.ndarray           # float64
    neighb_proc: int = -1
    neighb_patch: int = -1
    neighb_global: Optional[np.ndarray] = None   # processor patch: global id of the cell across each face


@dataclass
class LduCase:
    n_cells: int
    lower_addr: np.ndarray           # int32 [F]  (owner)
    upper_addr: np.ndarray           # int32 [F]  (neighbour)
    diag: np.ndarray                 # float64 [N]
    upper: np.ndarray                # float64 [F]
    lower: Optional[np.ndarray]      # float64 [F] or None (symmetric)
    interfaces: List[Interface] = field(default_factory=list)
    global_index: Optional[np.ndarray] = None   # int64 [N] global cell id of each local cell
    global_n: int = 0
    centres: Optional[np.ndarray] = None        # float64 [N, 3] cell centres (mesh.C()), optional

    @property
    def symmetric(self):
        return self.lower is None

    @property
    def n_faces(self):
        return int(self.upper_addr.size)

    @property
    def nnz(self):
        return self.n_cells + 2 * self.n_faces


def box_faces(lx, ly, lz):
    """(lower_addr, upper_addr) of an lx*ly*lz box in upper-triangular order."""
    n = lx * ly * lz
    c = np.arange(n, dtype=np.int32)
    i = c % lx
    j = (c // lx) % ly
    k = c // (lx * ly)
    mask = np.stack([i < lx - 1, j < ly - 1, k < lz - 1], axis=1)
    nb = np.stack([c + 1, c + lx, c + lx * ly], axis=1)
    own = np.broadcast_to(c[:, None], nb.shape)
    return np.ascontiguousarray(own[mask]), np.ascontiguousarray(nb[mask])


def _delta(gi):
    return 1e-3 * (1.0 + (gi % 7) / 7.0)


def poisson_block(gx, gy, gz, px=1, py=1, pz=1, rank=0, symmetric=True, periodic_x=False,
                  off_upper=-1.0, off_lower=-1.0, with_centres=False):
    """Rank `rank`'s share of a gx*gy*gz Poisson box cut into px*py*pz equal blocks.

    rank = bx + px*(by + py*bz).  periodic_x adds a cyclic patch pair (only with px == 1).
    With symmetric=False the upper/lower coefficients are off_upper/off_lower.
    """
    assert gx % px == 0 and gy % py == 0 and gz % pz == 0
    assert not (periodic_x and px != 1)
    lx, ly, lz = gx // px, gy // py, gz // pz
    bx, by, bz = rank % px, (rank // px) % py, rank // (px * py)
    ox, oy, oz = bx * lx, by * ly, bz * lz
    n = lx * ly * lz
    lower_addr, upper_addr = box_faces(lx, ly, lz)
    F = lower_addr.size
    c = np.arange(n, dtype=np.int64)
    i, j, k = c % lx, (c // lx) % ly, c // (lx * ly)
    gi = (ox + i) + gx * ((oy + j) + gy * (oz + k))
    gI, gJ, gK = ox + i, oy + j, oz + k
    nnb = ((gI > 0).astype(np.float64) + (gI < gx - 1) + (gJ > 0) + (gJ < gy - 1) + (gK > 0)
           + (gK < gz - 1))
    if periodic_x:
        nnb = nnb + (gI == 0) + (gI == gx - 1)
    diag = nnb + _delta(gi)
    if symmetric:
        upper = np.full(F, off_upper, dtype=np.float64)
        lower = None
    else:
        upper = np.full(F, off_upper, dtype=np.float64)
        lower = np.full(F, off_lower, dtype=np.float64)

    # coupled interfaces, ascending neighbour rank: -z, -y, -x, +x, +y, +z
    cl = np.arange(n, dtype=np.int32)
    ifaces = []

    def add_proc(cond, nrank, to_higher, g_step):
        cells = np.ascontiguousarray(cl[cond])
        # true off-diagonal entry = -bouCoeffs; row owner < column owner => "upper" coefficient
        coeff = off_upper if (to_higher or symmetric) else off_lower
        ifaces.append(Interface(IFACE_PROCESSOR, cells, np.full(cells.size, -coeff), nrank, -1,
                                gi[cond] + g_step))

    if bz > 0:
        add_proc(k == 0, rank - px * py, False, -gx * gy)
    if by > 0:
        add_proc(j == 0, rank - px, False, -gx)
    if bx > 0:
        add_proc(i == 0, rank - 1, False, -1)
    if bx < px - 1:
        add_proc(i == lx - 1, rank + 1, True, 1)
    if by < py - 1:
        add_proc(j == ly - 1, rank + px, True, gx)
    if bz < pz - 1:
        add_proc(k == lz - 1, rank + px * py, True, gx * gy)
    if periodic_x:
        left = np.ascontiguousarray(cl[i == 0])
        right = np.ascontiguousarray(cl[i == lx - 1])
        p0 = len(ifaces)
        # row in `left`, column in `right` (> row): upper-type coefficient, and vice versa
        ifaces.append(Interface(IFACE_CYCLIC, left, np.full(left.size, -off_upper), -1, p0 + 1))
        ifaces.append(Interface(IFACE_CYCLIC, right,
                                np.full(right.size, -(off_upper if symmetric else off_lower)), -1,
                                p0))
    centres = np.stack([gI + 0.5, gJ + 0.5, gK + 0.5], axis=1).astype(np.float64) if with_centres else None
    return LduCase(n, lower_addr, upper_addr, diag, upper, lower, ifaces, gi, gx * gy * gz, centres)


def poisson_case(n, symmetric=True, **kw):
    """Single-rank n^3 case of BASELINE.md §3 (asymmetric variant: upper -0.9, lower -1.1)."""
    if not symmetric:
        kw.setdefault("off_upper", -0.9)
        kw.setdefault("off_lower", -1.1)
    return poisson_block(n, n, n, symmetric=symmetric, **kw)


def x_star(global_index, global_n):
    """x*_i = sin(2 pi i / N) on the global numbering."""
    return np.sin(2.0 * np.pi * global_index.astype(np.float64) / float(global_n))


def apply_case(case: LduCase, x, halo_fn=None):
    """y = A x straight from the LDU form (numpy; input generation only, not an oracle).

    halo_fn(iface_index, send_values) -> neighbour values, for processor interfaces.
    """
    y = case.diag * x
    lo, up = case.lower_addr, case.upper_addr
    lower = case.upper if case.lower is None else case.lower
    np.add.at(y, lo, case.upper * x[up])
    np.add.at(y, up, lower * x[lo])
    for idx, itf in enumerate(case.interfaces):
        if itf.kind == IFACE_CYCLIC:
            nb = case.interfaces[itf.neighb_patch].face_cells
            np.add.at(y, itf.face_cells, -itf.bou_coeffs * x[nb])
        elif halo_fn is not None:
            np.add.at(y, itf.face_cells, -itf.bou_coeffs * halo_fn(idx, x[itf.face_cells]))
    return y


def rhs_for_x_star(case: LduCase):
    """(b, x*) with b = A x* (BASELINE.md §3); on a decomposed case the values of x* across the
    processor patches come from the analytic global formula, so no communication is needed."""
    xs = x_star(case.global_index, case.global_n)

    def halo(idx, _send):
        return x_star(case.interfaces[idx].neighb_global, case.global_n)
    return apply_case(case, xs, halo), xs


def permute_case(case: LduCase, new_id) -> LduCase:
    """The same system with cell c renamed new_id[c].  Faces are put back into OpenFOAM's
    upper-triangular order (owner < neighbour, sorted by owner then neighbour)."""
    new_id = np.asarray(new_id, dtype=np.int64)
    n = case.n_cells
    a, b = new_id[case.lower_addr], new_id[case.upper_addr]
    swapped = a > b
    lo, up = np.where(swapped, b, a), np.where(swapped, a, b)
    order = np.lexsort((up, lo))
    upper = case.upper if case.lower is None else np.where(swapped, case.lower, case.upper)
    lower = None if case.lower is None else np.where(swapped, case.upper, case.lower)[order]
    diag = np.empty_like(case.diag)
    diag[new_id] = case.diag
    gi = None
    if case.global_index is not None:
        gi = np.empty_like(case.global_index)
        gi[new_id] = case.global_index
    ifaces = [Interface(f.kind, new_id[f.face_cells].astype(np.int32), f.bou_coeffs, f.neighb_proc,
                        f.neighb_patch, f.neighb_global) for f in case.interfaces]
    centres = None
    if case.centres is not None:
        centres = np.empty_like(case.centres)
        centres[new_id] = case.centres
    return LduCase(n, lo[order].astype(np.int32), up[order].astype(np.int32), diag, upper[order], lower,
                   ifaces, gi, case.global_n, centres)


def renumber_case(case: LduCase, window: int, seed: int = 20241016) -> LduCase:
    """Cells renumbered at random inside consecutive windows of `window` cells -- a stand-in for an
    unstructured mesh: the diagonals of the box dissolve into a band of irregular offsets."""
    rng = np.random.default_rng(seed)
    n = case.n_cells
    new_id = np.empty(n, dtype=np.int64)
    for start in range(0, n, window):
        stop = min(n, start + window)
        new_id[start:stop] = start + rng.permutation(stop - start)
    return permute_case(case, new_id)


def drop_faces_case(case: LduCase, fraction: float, seed: int = 20241016) -> LduCase:
    """A stand-in for a mesh of mixed cell types: a random `fraction` of the internal faces is removed
    (row lengths then vary from 1 to 7 on the box) and the diagonal is reduced by the couplings that
    went, so the matrix stays diagonally dominant.  Single-rank cases only."""
    assert not case.interfaces
    rng = np.random.default_rng(seed)
    keep = rng.random(case.n_faces) >= fraction
    gone = ~keep
    diag = case.diag.copy()
    low = case.upper if case.lower is None else case.lower
    np.add.at(diag, case.lower_addr[gone], -np.abs(case.upper[gone]))
    np.add.at(diag, case.upper_addr[gone], -np.abs(low[gone]))
    return LduCase(case.n_cells, case.lower_addr[keep].copy(), case.upper_addr[keep].copy(), diag,
                   case.upper[keep].copy(), None if case.lower is None else case.lower[keep].copy(), [],
                   case.global_index, case.global_n)


def long_rows_case(case: LduCase, fraction: float, nx: int, seed: int = 20241016) -> LduCase:
    """A stand-in for a hex-dominant mesh (snappyHexMesh): a random `fraction` of the cells of an nx-wide
    box gets five extra couplings (offsets 2, nx+1, 2 nx, nx^2+1, nx^2+nx: none of them a stencil leg), so
    that a few rows have 12 entries among many with 7.  Coefficient -0.25 each, the diagonal grows
    accordingly.  Single-rank cases only."""
    assert not case.interfaces
    rng = np.random.default_rng(seed)
    n = case.n_cells
    cells = np.flatnonzero(rng.random(n) < fraction).astype(np.int64)
    offs = np.array([2, nx + 1, 2 * nx, nx * nx + 1, nx * nx + nx], dtype=np.int64)
    a = np.repeat(cells, offs.size)
    b = a + np.tile(offs, cells.size)
    ok = b < n
    a, b = a[ok], b[ok]
    lo = np.concatenate([case.lower_addr.astype(np.int64), a])
    up = np.concatenate([case.upper_addr.astype(np.int64), b])
    coef = np.concatenate([case.upper, np.full(a.size, -0.25)])
    low = None if case.lower is None else np.concatenate([case.lower, np.full(a.size, -0.25)])
    order = np.lexsort((up, lo))
    diag = case.diag.copy()
    np.add.at(diag, a, 0.25)
    np.add.at(diag, b, 0.25)
    return LduCase(n, lo[order].astype(np.int32), up[order].astype(np.int32), diag, coef[order],
                   None if low is None else low[order], [], case.global_index, case.global_n)


def voronoi_case(n_points: int, seed: int = 20241016, with_centres: bool = False) -> LduCase:
    """A genuinely unstructured finite-volume pattern: the cells are the Voronoi cells of `n_points`
    random points in the unit cube, two cells share a face when their points share a Delaunay edge
    (scipy.spatial.Delaunay) -- a polyhedral mesh with 15.5 faces per cell on average (5 ... ~60), numbered
    in the random order of the points.  upper = -1, diag = #faces + delta.  Single-rank cases only.
    The triangulation takes ~1 min per million points: with OGL_CASE_CACHE_DIR set the addressing is kept
    there between calls (profiling scripts run the same case several times)."""
    import os
    cache = os.environ.get("OGL_CASE_CACHE_DIR")
    path = os.path.join(cache, f"voronoi_{n_points}_{seed}.npz") if cache else None
    if path and os.path.exists(path):
        z = np.load(path)
        lo, up, n_faces = z["lo"], z["up"], z["n_faces"]
    else:
        from scipy.spatial import Delaunay
        rng = np.random.default_rng(seed)
        tri = Delaunay(rng.random((n_points, 3)))
        indptr, indices = tri.vertex_neighbor_vertices
        own = np.repeat(np.arange(n_points, dtype=np.int64), np.diff(indptr))
        keep = own < indices                                 # every face once, owner < neighbour
        lo, up = own[keep], indices[keep].astype(np.int64)
        order = np.lexsort((up, lo))
        lo, up, n_faces = lo[order].astype(np.int32), up[order].astype(np.int32), np.diff(indptr)
        if path:
            np.savez(path, lo=lo, up=up, n_faces=n_faces)
    gi = np.arange(n_points, dtype=np.int64)
    diag = n_faces.astype(np.float64) + _delta(gi)
    # (the generating points stand in for the cell centres: same seed, same first draw as the triangulation's)
    centres = np.random.default_rng(seed).random((n_points, 3)) if with_centres else None
    return LduCase(n_points, lo, up, diag, np.full(lo.size, -1.0), None, [], gi, n_points, centres)


def octree_case(n: int, band: float = 1.5, append_children: bool = False) -> LduCase:
    """A hex-dominant mesh as snappyHexMesh / an octree mesher makes it: an n^3 box of hexahedra in which
    every cell within `band` cells of a sphere (radius n/3, centred) is split into 2x2x2 children.  An
    unsplit cell next to a split one shares FOUR faces with it on that side, so the rows of the matrix have
    7 entries in the bulk and 10, 13 or 16 along the two surfaces of the refined shell.  Numbering: cell by
    cell in the box's x-fastest order, the 8 children of a split cell consecutively (append_children: child
    0 keeps the parent's place and the other seven go to the end of the list, which is what cell splitting
    in place leaves behind).  upper = -1, diag = #faces + delta.  Single-rank cases only."""
    ii, jj, kk = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")   # [i, j, k], i = x
    c = (n - 1) / 2.0
    dist = np.sqrt((ii - c) ** 2 + (jj - c) ** 2 + (kk - c) ** 2)
    split = np.abs(dist - n / 3.0) <= band                                             # [x, y, z]
    flat = lambda a: np.transpose(a, (2, 1, 0)).reshape(-1)                            # x-fastest order
    sp = flat(split)
    ncoarse = n ** 3
    if append_children:
        first = np.arange(ncoarse, dtype=np.int64)                 # parent's place = child 0 / the unsplit cell
        rest = ncoarse + 7 * (np.cumsum(sp) - sp).astype(np.int64)  # children 1..7 of a split cell
        n_cells = ncoarse + 7 * int(sp.sum())

        def child(cells, q):                                       # q = a + 2 b + 4 c
            return np.where(q == 0, first[cells], rest[cells] + q - 1)
    else:
        first = np.cumsum(np.where(sp, 8, 1)) - np.where(sp, 8, 1)
        first = first.astype(np.int64)
        n_cells = int(first[-1] + (8 if sp[-1] else 1))

        def child(cells, q):
            return first[cells] + q
    cid = np.arange(ncoarse, dtype=np.int64).reshape(n, n, n)      # [z, y, x]
    pairs = []
    strides = (1, n, n * n)
    for d in range(3):                                             # faces between box neighbours along axis d
        sl_lo = [slice(None)] * 3
        sl_hi = [slice(None)] * 3
        sl_lo[2 - d] = slice(0, n - 1)
        sl_hi[2 - d] = slice(1, n)
        a = cid[tuple(sl_lo)].reshape(-1)
        b = cid[tuple(sl_hi)].reshape(-1)
        sa, sb = sp[a], sp[b]
        o1, o2 = [x for x in range(3) if x != d]                   # the two axes of the shared face
        both = ~sa & ~sb
        pairs.append((first[a[both]], first[b[both]]))
        for u in (0, 1):
            for v in (0, 1):
                q_hi = (1 << d) | (u << o1) | (v << o2)            # child of the lower cell on its + side
                q_lo = (u << o1) | (v << o2)                       # child of the upper cell on its - side
                m = sa & sb
                pairs.append((child(a[m], q_hi), child(b[m], q_lo)))
                m = sa & ~sb
                pairs.append((child(a[m], q_hi), first[b[m]]))
                m = ~sa & sb
                pairs.append((first[a[m]], child(b[m], q_lo)))
    cells = np.flatnonzero(sp)
    for d in range(3):                                             # the 12 faces inside a split cell
        for q in range(8):
            if not q & (1 << d):
                pairs.append((child(cells, q), child(cells, q | (1 << d))))
    lo = np.concatenate([np.minimum(p, q) for p, q in pairs])
    up = np.concatenate([np.maximum(p, q) for p, q in pairs])
    order = np.lexsort((up, lo))
    lo, up = lo[order], up[order]
    n_faces = np.bincount(lo, minlength=n_cells) + np.bincount(up, minlength=n_cells)
    gi = np.arange(n_cells, dtype=np.int64)
    diag = n_faces.astype(np.float64) + _delta(gi)
    return LduCase(n_cells, lo.astype(np.int32), up.astype(np.int32), diag, np.full(lo.size, -1.0), None, [],
                   gi, n_cells)


def multi_block_case(blocks, ny: int, nz: int) -> LduCase:
    """A multi-block structured mesh as blockMesh numbers it: boxes of blocks[b] x ny x nz cells glued along x,
    the cells of block b numbered (x fastest) after those of block b - 1.  Inside a block the neighbours sit at
    distances 1, nx_b, nx_b * ny; the faces between two blocks couple cells at distances that vary from cell to
    cell -- a pattern that is banded block by block, not as a whole."""
    start = np.concatenate([[0], np.cumsum([nx * ny * nz for nx in blocks])]).astype(np.int64)
    lo, up = [], []
    for b, nx in enumerate(blocks):
        i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
        cell = start[b] + i + nx * (j + ny * k)
        for (di, dj, dk) in ((1, 0, 0), (0, 1, 0), (0, 0, 1)):
            ok = (i + di < nx) & (j + dj < ny) & (k + dk < nz)
            lo.append(cell[ok])
            up.append((cell + di + nx * (dj + ny * dk))[ok])
        if b + 1 < len(blocks):   # x-face to the next block: (nx - 1, j, k) <-> (0, j, k)
            nx2 = blocks[b + 1]
            jj, kk = np.meshgrid(np.arange(ny), np.arange(nz), indexing="ij")
            lo.append((start[b] + (nx - 1) + nx * (jj + ny * kk)).ravel())
            up.append((start[b + 1] + nx2 * (jj + ny * kk)).ravel())
    lo, up = np.concatenate([a.ravel() for a in lo]), np.concatenate([a.ravel() for a in up])
    order = np.lexsort((up, lo))
    lo, up = lo[order].astype(np.int32), up[order].astype(np.int32)
    n = int(start[-1])
    gi = np.arange(n, dtype=np.int64)
    deg = np.bincount(lo, minlength=n) + np.bincount(up, minlength=n)
    return LduCase(n, lo, up, deg + _delta(gi), np.full(lo.size, -1.0), None, [], gi, n)


def rcm_case(case: LduCase) -> LduCase:
    """What OpenFOAM's renumberMesh does: reverse Cuthill-McKee ordering of the cell graph (scipy)."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    n = case.n_cells
    ones = np.ones(case.lower_addr.size, dtype=np.int8)
    g = sp.coo_matrix((ones, (case.lower_addr, case.upper_addr)), shape=(n, n)).tocsr()
    order = reverse_cuthill_mckee(g + g.T, symmetric_mode=True)   # order[k] = old id of new cell k
    new_id = np.empty(n, dtype=np.int64)
    new_id[order] = np.arange(n)
    return permute_case(case, new_id)


def random_global_case(n, per_row, reach, symmetric=True, seed=0):
    """Diagonally dominant random lduMatrix: every cell couples to `per_row` random cells among the
    next `reach` ones (faces in upper-triangular order)."""
    rng = np.random.default_rng(seed)
    pairs = set()
    for i in range(n - 1):
        for j in rng.integers(i + 1, min(n, i + 1 + reach), per_row):
            pairs.add((i, int(j)))
    pairs = np.array(sorted(pairs), dtype=np.int32).reshape(-1, 2)
    f = len(pairs)
    upper = rng.uniform(-1, -0.1, f)
    lower = None if symmetric else rng.uniform(-1, -0.1, f)
    diag = np.full(n, 1.0)
    np.add.at(diag, pairs[:, 0], np.abs(upper))
    np.add.at(diag, pairs[:, 1], np.abs(upper if symmetric else lower))
    return LduCase(n, pairs[:, 0].copy(), pairs[:, 1].copy(), diag, upper, lower, [],
                   np.arange(n, dtype=np.int64), n)


def partition_rows(glob: LduCase, bounds, rank) -> LduCase:
    """Rank `rank`'s rows [bounds[rank], bounds[rank+1]) of a global case: faces inside the block stay
    faces, faces that cross into rank q become one processor interface per neighbour (ascending q),
    ordered by (global owner, global neighbour) on both sides, bouCoeffs = -(off-diagonal entry)."""
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    a, b = glob.lower_addr.astype(np.int64), glob.upper_addr.astype(np.int64)
    owner = np.searchsorted(np.asarray(bounds[1:]), np.arange(glob.n_cells), side="right")
    ra, rb = owner[a], owner[b]
    inside = (ra == rank) & (rb == rank)
    low_coeff = glob.upper if glob.lower is None else glob.lower
    ifaces = []
    for q in sorted(set(rb[(ra == rank) & (rb != rank)]) | set(ra[(rb == rank) & (ra != rank)])):
        mine_low = (ra == rank) & (rb == q)      # my cell is the face owner: entry (a, b) = upper
        mine_up = (rb == rank) & (ra == q)       # my cell is the neighbour: entry (b, a) = lower
        sel = np.flatnonzero(mine_low | mine_up)  # already sorted by (owner, neighbour)
        cells = np.where(mine_low[sel], a[sel], b[sel]) - lo
        coeff = np.where(mine_low[sel], glob.upper[sel], low_coeff[sel])
        ifaces.append(Interface(IFACE_PROCESSOR, cells.astype(np.int32), -coeff, int(q), -1))
    return LduCase(hi - lo, (a[inside] - lo).astype(np.int32), (b[inside] - lo).astype(np.int32),
                   glob.diag[lo:hi].copy(), glob.upper[inside].copy(),
                   None if glob.lower is None else glob.lower[inside].copy(), ifaces,
                   np.arange(lo, hi, dtype=np.int64), glob.n_cells)
