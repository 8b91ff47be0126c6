// host_sanitize.cpp -- the host-side layout builders (per-chunk half storage, compressed chunked ELL) and numbering code on patterns read
// from files, meant for an AddressSanitizer / UBSan build (tests/test_cpp_host.py builds and runs it):
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -I include -I ogl_amd/csrc tools/host_sanitize.cpp \
//       ogl_amd/csrc/host_matrix.cpp ogl_amd/csrc/common.cpp -lpthread -ldl -o host_sanitize && ./host_sanitize pattern.bin ...
// pattern.bin: int32 n_rows, int32 nnz, int32 row_ptrs[n_rows + 1], int32 cols[nnz]
#include <cstdio>
#include <cstdint>
#include <vector>
#include "ogl_amd.h"
int main(int argc, char **argv) {
    for (int i = 1; i < argc; ++i) {
        FILE *f = fopen(argv[i], "rb");
        int32_t hdr[2];
        if (!f || fread(hdr, 4, 2, f) != 2) return 2;
        std::vector<int32_t> rp(hdr[0] + 1), cols(hdr[1]);
        if (fread(rp.data(), 4, rp.size(), f) != rp.size() || fread(cols.data(), 4, cols.size(), f) != cols.size()) return 3;
        fclose(f);
        int64_t st[8];
        int rc = ogl_host_symx_check(hdr[0], rp.data(), cols.data(), st);
        printf("%s symx rc=%d ok=%lld planar=%lld explicit=%lld fast=%lld general=%lld\n", argv[i], rc, (long long)st[0], (long long)st[2], (long long)st[3], (long long)st[6], (long long)st[7]);
        rc = ogl_host_sell_check(hdr[0], rp.data(), cols.data(), st);
        printf("   sell rc=%d ok=%lld\n", rc, (long long)st[0]);
        // the numbering candidates (reverse Cuthill-McKee, Hilbert curve through made-up centres) and the whole
        // renumbered pattern of the lduMatrix view that has this pattern (faces = its upper entries, owner-sorted)
        const int32_t n = hdr[0];
        std::vector<int32_t> nid(n > 0 ? n : 1);
        rc = ogl_host_rcm(n, rp.data(), cols.data(), nid.data());
        const double r_rcm = ogl_host_gather_sector_ratio(n, rp.data(), cols.data(), nid.data());
        std::vector<double> centres(3 * (size_t)(n > 0 ? n : 1));
        uint64_t lcg = 88172645463325252ull + (uint64_t)n;
        for (double &c : centres) {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            c = (double)(lcg >> 11) / 9007199254740992.0;
        }
        const int rc_h = ogl_host_hilbert_order(n, centres.data(), nid.data());
        const double r_h = ogl_host_gather_sector_ratio(n, rp.data(), cols.data(), nid.data());
        std::vector<int32_t> lo, up;
        for (int32_t r = 0; r < n; ++r)
            for (int32_t e = rp[r]; e < rp[r + 1]; ++e)
                if (cols[e] > r) {
                    lo.push_back(r);
                    up.push_back(cols[e]);
                }
        std::vector<double> diag(n > 0 ? n : 1, 6.0), off(lo.size() + 1, -1.0);
        ogl_ldu_view v{};
        v.n_cells = n;
        v.n_faces = (int32_t)lo.size();
        v.lower_addr = lo.data();
        v.upper_addr = up.data();
        v.diag = diag.data();
        v.upper = off.data();
        v.cell_centres = centres.data();
        int renumbered = 0;
        for (int with_centres = 0; with_centres < 2; ++with_centres) {
            v.cell_centres = with_centres ? centres.data() : nullptr;
            ogl_matrix_dims d{};
            if (ogl_host_pattern_renumbered(&v, 1, 1, &d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                            nullptr, nullptr) < 0)
                return 4;
            std::vector<int32_t> a(d.local_nnz + 1), b(d.local_nnz + 1), c(d.local_nnz + 1), nl(1), ids(1), sz(1), snd(1);
            renumbered += ogl_host_pattern_renumbered(&v, 1, 1, &d, a.data(), b.data(), c.data(), nl.data(), nl.data(), nl.data(),
                                                      ids.data(), sz.data(), snd.data(), nid.data());
        }
        printf("   numbering rcm rc=%d ratio %.3f  hilbert rc=%d ratio %.3f  renumbered %d\n", rc, r_rcm, rc_h, r_h, renumbered);
    }
    return 0;
}
