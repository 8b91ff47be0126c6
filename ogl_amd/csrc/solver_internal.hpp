// solver_internal.hpp -- what the translation units of the solver share besides solver.hpp: small helpers with internal
// linkage.  solver.cpp: registry, staging, layouts, HostMatrixWrapper (set_matrix); solver_transport.cpp: the peer mesh
// and the all-reduce; solver_precond.cpp: Preconditioner.H:353-431; solver_krylov.cpp: the distributed product, the
// finalisers and the Krylov drivers (lduLduBase.H:189-308).
#pragma once
#include <chrono>
#include <cstddef>

#include "solver.hpp"

namespace ogl {
namespace {

inline double now_ms()
{
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}

inline double *sums_ptr(DevScalars *s)
{
    return reinterpret_cast<double *>(reinterpret_cast<char *>(s) + offsetof(DevScalars, sums));
}

struct EventPair {  // destroyed on every return path
    hipEvent_t e[2] = {nullptr, nullptr};
    ~EventPair()
    {
        for (auto &x : e)
            if (x) ev_destroy(x);
    }
    hipEvent_t &operator[](int i) { return e[i]; }
};

}  // namespace
}  // namespace ogl
