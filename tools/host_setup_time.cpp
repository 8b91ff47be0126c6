// host_setup_time.cpp -- where the first set_matrix of a field spends its host time (development tool).
//   g++ -O3 -std=c++17 -I include -I ogl_amd/csrc tools/host_setup_time.cpp ogl_amd/csrc/host_matrix.cpp \
//       ogl_amd/csrc/common.cpp -lpthread -ldl -o tools/bin/host_setup_time && tools/bin/host_setup_time 216 [shuffle_window]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#include "host_matrix.hpp"

using namespace ogl;
static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 216;
    const int window = argc > 2 ? atoi(argv[2]) : 0;
    const int64_t N = (int64_t)n * n * n;
    std::vector<int32_t> new_id((size_t)N);
    std::iota(new_id.begin(), new_id.end(), 0);
    if (window) {
        std::mt19937 rng(20241016);
        for (int64_t s = 0; s < N; s += window)
            std::shuffle(new_id.begin() + s, new_id.begin() + std::min<int64_t>(N, s + window), rng);
    }
    // faces of the box in the (possibly renamed) numbering, upper-triangular order
    std::vector<std::pair<int32_t, int32_t>> faces;
    faces.reserve((size_t)3 * N);
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                const int64_t c = i + (int64_t)n * (j + (int64_t)n * k);
                auto add = [&](int64_t d) {
                    int32_t a = new_id[(size_t)c], b = new_id[(size_t)d];
                    if (a > b) std::swap(a, b);
                    faces.emplace_back(a, b);
                };
                if (i < n - 1) add(c + 1);
                if (j < n - 1) add(c + n);
                if (k < n - 1) add(c + (int64_t)n * n);
            }
    if (window) std::sort(faces.begin(), faces.end());
    const int64_t F = (int64_t)faces.size();
    std::vector<int32_t> lo((size_t)F), up((size_t)F);
    for (int64_t f = 0; f < F; ++f) {
        lo[(size_t)f] = faces[(size_t)f].first;
        up[(size_t)f] = faces[(size_t)f].second;
    }
    faces.clear();
    faces.shrink_to_fit();
    std::vector<double> diag((size_t)N, 6.0), upper((size_t)F, -1.0);
    ogl_ldu_view v{};
    v.n_cells = (int32_t)N;
    v.n_faces = (int32_t)F;
    v.lower_addr = lo.data();
    v.upper_addr = up.data();
    v.diag = diag.data();
    v.upper = upper.data();
    double t0 = now();
    HostPattern p;
    if (build_host_pattern(v, p) != OGL_OK) return 1;
    double t1 = now();
    printf("build_host_pattern      %.3f s  (N %ld nnz %d)\n", t1 - t0, (long)N, p.local_nnz);
    SellLayout pre;
    bool pre_built = false;
    RenumberReport rep;
    choose_numbering(p, 2, true, &pre, &pre_built, rep);
    double t2 = now();
    printf("choose_numbering(auto)  %.3f s  (applied %d sorted %d sell_built %d sell_used %d ratio %.3f -> %.3f)\n", t2 - t1,
           rep.applied, rep.sorted_by_length, pre_built, rep.sell_used, rep.ratio_natural, rep.ratio_used);
    SymLayout sym;
    const bool ok = !p.renumbered() && build_sym_layout(p.n_rows, p.row_ptrs.data(), p.cols.data(), sym);
    double t3 = now();
    printf("build_sym_layout        %.3f s  (qualifies %d nd %d)\n", t3 - t2, ok, sym.nd);
    std::vector<int32_t> dpos((size_t)p.n_rows, -1);
    for (int32_t r = 0; r < p.n_rows; ++r)
        for (int32_t k = p.row_ptrs[r]; k < p.row_ptrs[r + 1]; ++k)
            if (p.cols[k] == r) {
                dpos[(size_t)r] = k;
                break;
            }
    double t4 = now();
    printf("diag positions          %.3f s\n", t4 - t3);
    printf("fingerprint             ");
    double t5 = now();
    (void)addressing_fingerprint(v);
    printf("%.3f s\n", now() - t5);
    return 0;
}
