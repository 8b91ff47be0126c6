// launch_key.hpp -- what a captured batch of turns bakes in, as ONE number.  A hipGraph replays kernel launches with the
// arguments they had at capture time; the arguments of this library's kernels are the view structs below (built from the
// solver's buffers right before every launch) plus a handful of vectors and scalars.  The key of a captured graph is
// the hash of exactly those views, field by field -- not a hand-kept list of pointers beside them: a view that gains a
// field changes its size, the assertion next to its visitor fails to compile, and the field gets hashed.
// (tests/test_gpu_graph_key.py swaps every layout under a live graph and compares with the uncaptured turns.)
#pragma once
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "kernels.hpp"

namespace ogl {

struct KeyHasher {
    uint64_t h = 1469598103934665603ull;
    void bytes(const void *p, size_t n)
    {
        const unsigned char *c = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 1099511628211ull;
    }
    template <class T>
    std::enable_if_t<std::is_arithmetic<T>::value || std::is_pointer<T>::value || std::is_enum<T>::value> operator()(const T &v)
    {
        bytes(&v, sizeof(v));
    }
};

#define OGL_VIEW_FIELDS(T, n)                                                                                        \
    static_assert(sizeof(T) == n, "a field of " #T " was added, removed or resized: bring visit(KeyHasher &, const " #T \
                                  " &) below in line with the struct, then update this size")

OGL_VIEW_FIELDS(DevCsr, 88);
inline void visit(KeyHasher &h, const DevCsr &a)
{
    h(a.n_rows), h(a.nnz), h(a.row_ptrs), h(a.cols), h(a.vals), h(a.stream), h(a.xcd_group), h(a.chunks21), h(a.codes21);
    h(a.far_idx21), h(a.far_col21), h(a.block_order), h(a.n_blocks), h(a.lds_rounds);
}
OGL_VIEW_FIELDS(DevEll, 40);
inline void visit(KeyHasher &h, const DevEll &a)
{
    h(a.n_rows), h(a.width), h(a.stride), h(a.cols), h(a.vals), h(a.stream);
}
OGL_VIEW_FIELDS(DevSell, 112);
inline void visit(KeyHasher &h, const DevSell &a)
{
    h(a.n_rows), h(a.chunks), h(a.dict), h(a.codes), h(a.vals), h(a.spill_chunk_ptr), h(a.spill_rows), h(a.spill_ptrs);
    h(a.spill_cols), h(a.spill_vals), h(a.stream), h(a.xcd_group), h(a.rmap), h(a.block_order), h(a.n_blocks);
}
OGL_VIEW_FIELDS(DevSym, 64);
inline void visit(KeyHasher &h, const DevSym &a)
{
    h(a.n_rows), h(a.nd);
    for (int j = 0; j < 4; ++j) h(a.d[j]);
    h(a.mask), h(a.planes), h(a.stream), h(a.block_order), h(a.n_blocks);
}
OGL_VIEW_FIELDS(DevSymx, 96);
inline void visit(KeyHasher &h, const DevSymx &a)
{
    h(a.n_rows), h(a.chunks), h(a.chunks_general), h(a.n_blocks_general), h(a.mask), h(a.planes), h(a.ex_rowptr), h(a.ex_cols);
    h(a.ex_lrow), h(a.ex_vals), h(a.stream), h(a.fast), h(a.xcd_group), h(a.n_blocks);
}
OGL_VIEW_FIELDS(LeadBox, 24);
inline void visit(KeyHasher &h, const LeadBox &a) { h(a.box), h(a.timeout_ticks), h(a.early_loads); }

#undef OGL_VIEW_FIELDS

}  // namespace ogl
