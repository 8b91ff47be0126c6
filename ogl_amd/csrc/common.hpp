// common.hpp -- error plumbing and compile-time geometry shared by host and device code.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "ogl_amd.h"

namespace ogl {

// Thread-local message behind ogl_last_error().  The reference aborts the process through
// OpenFOAM's FatalError (e.g. HostMatrix.C:339-341, ExecutorHandler.H:107-110); a C ABI cannot,
// so every failure becomes a status code + message and the adapter raises FatalError from it.
std::string &last_error();
int fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Geometry of the deterministic reduction tree and of the CSR-stream SpMV (kernels.hip).
// One workgroup = BLOCK threads = one chunk of CHUNK_ROWS consecutive rows; thread t owns the
// ROWS_PER_THREAD consecutive rows starting at chunk_start + t * ROWS_PER_THREAD.
constexpr int BLOCK = 256;
constexpr int WAVE = 64;
constexpr int CHUNK_ROWS = 512;
constexpr int ROWS_PER_THREAD = CHUNK_ROWS / BLOCK;
// Non-zeros staged through LDS per pass of the SpMV (products, 8 B each).
constexpr int SPMV_TILE = 4096;
// Vector loads read up to 3 entries past a tile end: value/column arrays carry this much padding.
constexpr int NNZ_PAD = 8;

inline int64_t n_chunks(int64_t n_rows) { return (n_rows + CHUNK_ROWS - 1) / CHUNK_ROWS; }

}  // namespace ogl
