// Host logic of libogl_amd (host_matrix.cpp, common.cpp: no HIP) compiled with g++ under
// AddressSanitizer + UndefinedBehaviorSanitizer and driven through the ogl_host_* C ABI on generated
// lduMatrix views: LDU -> CSR pattern, update functions, halo / communication pattern, compressed
// layout (build + decode).  Sanitizers run on the CPU build only (the GPU pool has no ASan).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "ogl_amd.h"

static int failures = 0;
#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #c); \
            ++failures;                                                 \
        }                                                               \
    } while (0)

struct Box {
    int nx, ny, nz;
    std::vector<ogl_label> lower, upper;
    std::vector<ogl_scalar> diag, up, lo;
    int n() const { return nx * ny * nz; }
};

static Box make_box(int nx, int ny, int nz)
{
    Box b{nx, ny, nz, {}, {}, {}, {}, {}};
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const int c = i + nx * (j + ny * k);
                if (i + 1 < nx) { b.lower.push_back(c); b.upper.push_back(c + 1); }
                if (j + 1 < ny) { b.lower.push_back(c); b.upper.push_back(c + nx); }
                if (k + 1 < nz) { b.lower.push_back(c); b.upper.push_back(c + nx * ny); }
            }
    std::mt19937 g(7);
    std::uniform_real_distribution<double> u(-1, 1);
    b.diag.resize(b.n());
    for (auto &d : b.diag) d = 7 + u(g);
    b.up.resize(b.lower.size());
    b.lo.resize(b.lower.size());
    for (auto &v : b.up) v = u(g);
    for (auto &v : b.lo) v = u(g);
    return b;
}

static void run_case(int nx, int ny, int nz, bool symmetric, bool with_interfaces)
{
    Box b = make_box(nx, ny, nz);
    const int N = b.n(), F = (int)b.lower.size();
    // processor interface on the z = nz-1 plane (neighbour rank 3) and a cyclic pair on x = 0 / nx-1
    std::vector<ogl_label> top, left, right;
    for (int c = 0; c < N; ++c) {
        if (c / (nx * ny) == nz - 1) top.push_back(c);
        if (c % nx == 0) left.push_back(c);
        if (c % nx == nx - 1) right.push_back(c);
    }
    std::vector<ogl_scalar> ctop(top.size(), 0.5), cl(left.size(), 0.25), cr(right.size(), 0.125);
    std::vector<ogl_interface> ifs;
    if (with_interfaces) {
        ifs.push_back({OGL_IFACE_PROCESSOR, 3, -1, (ogl_label)top.size(), top.data(), ctop.data()});
        ifs.push_back({OGL_IFACE_CYCLIC, -1, 2, (ogl_label)left.size(), left.data(), cl.data()});
        ifs.push_back({OGL_IFACE_CYCLIC, -1, 1, (ogl_label)right.size(), right.data(), cr.data()});
    }
    ogl_ldu_view v{};
    v.n_cells = N;
    v.n_faces = F;
    v.lower_addr = b.lower.data();
    v.upper_addr = b.upper.data();
    v.diag = b.diag.data();
    v.upper = b.up.data();
    v.lower = symmetric ? nullptr : b.lo.data();
    v.n_interfaces = (ogl_label)ifs.size();
    v.interfaces = ifs.data();

    ogl_matrix_dims d{};
    CHECK(ogl_host_pattern(&v, &d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr) == OGL_OK);
    const int cyc = with_interfaces ? (int)(left.size() + right.size()) : 0;
    CHECK(d.n_rows == N && d.local_nnz == N + 2 * F + cyc);
    std::vector<ogl_label> rows(d.local_nnz), cols(d.local_nnz), map(d.local_nnz), nr(d.non_local_nnz + 1),
        nc(d.non_local_nnz + 1), nm(d.non_local_nnz + 1), tid(d.n_neighbours + 1), tsz(d.n_neighbours + 1),
        snd(d.n_send + 1);
    CHECK(ogl_host_pattern(&v, &d, rows.data(), cols.data(), map.data(), nr.data(), nc.data(), nm.data(),
                           tid.data(), tsz.data(), snd.data()) == OGL_OK);
    for (int e = 1; e < d.local_nnz; ++e)
        CHECK(rows[e] > rows[e - 1] || (rows[e] == rows[e - 1] && cols[e] >= cols[e - 1]));
    // CSR row pointers -> compressed layout build + decode
    std::vector<ogl_label> rp(N + 1, 0);
    for (int e = 0; e < d.local_nnz; ++e) ++rp[rows[e] + 1];
    for (int r = 0; r < N; ++r) rp[r + 1] += rp[r];
    int64_t stats[8];
    CHECK(ogl_host_sell_check(N, rp.data(), cols.data(), stats) == OGL_OK);
    CHECK(stats[0] == 1 && stats[1] >= d.local_nnz);
    {   // half storage of the symmetric pattern: builds, walks, must qualify on the plain box
        int64_t sym[8];
        CHECK(ogl_host_sym_check(N, rp.data(), cols.data(), sym) == OGL_OK);
        if (!with_interfaces && N > 1) CHECK(sym[0] == 1);  // (a cyclic pair adds distances: full storage stays)
        if (sym[0]) CHECK(sym[1] >= 2 && sym[1] <= 4 && sym[2] == 0 && sym[7] <= sym[6]);
    }
    // renumbering: forced (RCM) and auto; the outputs must be a permutation and a sorted pattern
    for (int mode = 1; mode <= 2; ++mode) {
        std::vector<ogl_label> new_id(N + 1, -1), rr(d.local_nnz), cc(d.local_nnz), mm(d.local_nnz);
        ogl_matrix_dims d2{};
        const int rc = ogl_host_pattern_renumbered(&v, mode, 1, &d2, rr.data(), cc.data(), mm.data(), nr.data(),
                                                   nc.data(), nm.data(), tid.data(), tsz.data(), snd.data(),
                                                   new_id.data());
        CHECK(rc == 0 || rc == 1);
        std::vector<int> seen(N, 0);
        for (int c = 0; c < N; ++c) {
            CHECK(new_id[c] >= 0 && new_id[c] < N);
            if (new_id[c] >= 0 && new_id[c] < N) ++seen[new_id[c]];
        }
        for (int c = 0; c < N; ++c) CHECK(seen[c] == 1);
        for (int e = 1; e < d2.local_nnz; ++e)
            CHECK(rr[e] > rr[e - 1] || (rr[e] == rr[e - 1] && cc[e] >= cc[e - 1]));
        std::vector<ogl_label> rp2(N + 1, 0);
        for (int e = 0; e < d2.local_nnz; ++e) ++rp2[rr[e] + 1];
        for (int r = 0; r < N; ++r) rp2[r + 1] += rp2[r];
        int64_t st2[8];
        CHECK(ogl_host_sell_check(N, rp2.data(), cc.data(), st2) == OGL_OK);
        std::vector<ogl_label> rcm(N + 1);
        CHECK(ogl_host_rcm(N, rp.data(), cols.data(), rcm.data()) == OGL_OK);
        (void)ogl_host_gather_sector_ratio(N, rp.data(), cols.data(), rcm.data());
    }
    (void)ogl_host_addressing_fingerprint(&v);
    // update functions on the plain (interface-free) pattern
    if (!with_interfaces) {
        std::vector<ogl_label> r2(N + 2 * F), c2(N + 2 * F), p2(N + 2 * F);
        ogl_host_init_local_sparsity(N, F, symmetric, b.upper.data(), b.lower.data(), r2.data(), c2.data(),
                                     p2.data());
        std::vector<ogl_scalar> out(N + 2 * F);
        if (symmetric)
            ogl_host_symmetric_update(N + 2 * F, F, p2.data(), 1.0, b.diag.data(), b.up.data(), out.data());
        else
            ogl_host_non_symmetric_update(N + 2 * F, F, p2.data(), 1.0, b.diag.data(), b.up.data(),
                                          b.lo.data(), out.data());
        for (int e = 0; e < N + 2 * F; ++e)
            if (r2[e] == c2[e]) CHECK(out[e] == b.diag[r2[e]]);
    }
}

int main()
{
    for (int sym = 0; sym < 2; ++sym)
        for (int ifs = 0; ifs < 2; ++ifs) {
            run_case(1, 1, 1, sym, false);
            run_case(5, 4, 3, sym, ifs);
            run_case(33, 17, 9, sym, ifs);   // 5049 cells: ten chunks, the last one partial
            run_case(600, 1, 1, sym, false);
            run_case(40, 30, 20, sym, ifs);  // 24,000 cells: past the auto policy's size threshold
        }
    // bad input must be refused, not read
    ogl_ldu_view bad{};
    bad.n_cells = 4;
    bad.n_faces = 1;
    ogl_label lo = 0, up = 9;
    ogl_scalar one = 1.0;
    std::vector<ogl_scalar> dg(4, 1.0);
    bad.lower_addr = &lo;
    bad.upper_addr = &up;
    bad.diag = dg.data();
    bad.upper = &one;
    ogl_matrix_dims d{};
    CHECK(ogl_host_pattern(&bad, &d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr) == OGL_ERR_INVALID);
    std::printf("%d failure(s)\n", failures);
    return failures ? 1 : 0;
}
