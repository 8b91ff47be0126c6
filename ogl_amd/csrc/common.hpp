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

// Geometry of the deterministic reduction tree and of the CSR-stream SpMV (device_common.hpp, kernels_*.hip).
// One workgroup = BLOCK threads = one chunk of CHUNK_ROWS consecutive rows; thread t owns the
// ROWS_PER_THREAD consecutive rows starting at chunk_start + t * ROWS_PER_THREAD.
#if defined(__HIPCC__)
#define OGL_HD_FWD __host__ __device__
#else
#define OGL_HD_FWD
#endif
constexpr int N_XCD = 8;  // MI355X: 8 XCDs, workgroup b is observed on XCD b % 8 (speed only)
constexpr int BLOCK = 256;
constexpr int WAVE = 64;
constexpr int CHUNK_ROWS = 512;
constexpr int ROWS_PER_THREAD = CHUNK_ROWS / BLOCK;
// Non-zeros staged through LDS per pass of the SpMV (products, 8 B each).
constexpr int SPMV_TILE = 4096;
// CSR-stream with packed columns (DevCsr::codes21; irregular patterns whose rows are too long / too uneven for the
// chunked ELL, e.g. polyhedral meshes): the values stay the plain CSR array, the columns of a chunk are kept as
// 21-bit offsets from the chunk's smallest column, six to a 16-byte word, in the order the lanes of the kernel
// consume them -- 10.67 instead of 12 bytes per entry.  A chunk whose columns span 2^21 or more (a numbering along a
// space-filling curve above 2 M rows: most neighbours in the chunk's own blob, a few in blobs anywhere) takes a window of
// 2^21 columns around its own rows; the few entries outside it ("far") are coded as offset 0 and listed per chunk with
// their entry index and column -- the kernel's workgroup overwrites their products before the rows are summed.
constexpr int STREAM21_TILE = 3072;                 // entries per pass: 256 lanes x 2 groups x 6 entries
constexpr int STREAM21_GROUPS = STREAM21_TILE / (BLOCK * 6);
constexpr int STREAM21_BITS = 21;
struct Stream21Chunk {  // per chunk (16 bytes: one scalar load)
    int32_t base;       // smallest column of the chunk / first column of its window
    int32_t word_off;   // first 16-byte code word of the chunk
    int32_t far_off;    // first entry of the chunk in the far lists
    int32_t far_n;      // entries of the chunk outside [base, base + 2^21)
};
// where a chunk's window starts when its columns span 2^21 or more: 2^20 columns before its first row, inside [0, n)
OGL_HD_FWD inline int32_t stream21_window_base(int32_t first_row, int32_t n_rows)
{
    const int64_t top = (int64_t)n_rows - ((int64_t)1 << STREAM21_BITS);
    int64_t b = (int64_t)first_row - ((int64_t)1 << (STREAM21_BITS - 1));
    if (b > top) b = top;
    if (b < 0) b = 0;
    return (int32_t)b;
}
constexpr double STREAM21_MAX_FAR = 0.02;  // ... and the layout is given up when more than this share of the entries is far
// Vector loads read up to 3 entries past a tile end: value/column arrays carry this much padding.
constexpr int NNZ_PAD = 8;

// ROCTx range around a phase of the plug-in call -- the analogue of the nvtxRangePushA/Pop inside
// the reference's TIME_WITH_FIELDNAME (common/common.H:54-87).  libroctx64 is bound at run time;
// without it (or without a profiler attached) the ranges cost a function-pointer test.
// `rocprofv3 --marker-trace` shows them as "ogl:<phase>:<field>".
class TraceRange {
public:
    TraceRange(const char *phase, const std::string &field);
    ~TraceRange();
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;

private:
    bool pushed_ = false;
};

inline int64_t n_chunks(int64_t n_rows) { return (n_rows + CHUNK_ROWS - 1) / CHUNK_ROWS; }

// Index-compressed chunked ELL ("SELL-512 with diagonal codes"): the layout the Coo/Csr-format path
// runs on when the sparsity pattern allows it.  Per chunk of CHUNK_ROWS rows: `width` = its longest
// row, values slot-major [width][CHUNK_ROWS], and the columns coded in one of two ways:
//   pattern mode (SELL_MODE_PATTERN): ONE BYTE PER ROW naming one of <= 256 row
//     patterns of the chunk; the table holds `width` (column - row) offsets per pattern,
//     SELL_PAD_OFFSET in the unused tail slots.  8.1 bytes per stored entry on a 7-point stencil;
//   offset mode (SELL_MODE_OFFSET8): one byte per (row, slot) naming an entry of the
//     chunk's ascending dictionary of <= 255 offsets (255 = padding slot).  9 bytes per entry.
//   delta mode (mode == SELL_MODE_DELTA16; irregular patterns, e.g. an unstructured mesh in
//     RCM numbering): 16 bits per (row, slot): the first code of a row is (first column - row) -
//     dict_off, every later one the distance to the previous column of the row (rows are stored in
//     ascending column order); 0xFFFF = padding.  10 bytes per entry.  Needs every distance and the
//     spread of the first offsets over the chunk below 65535;
//   column mode (mode == SELL_MODE_COL32): plain 32-bit columns, -1 = padding.  12 bytes per
//     entry; the fallback that every chunk can take.
//   Delta / column codes are stored as 16-byte words, group-major: word (g, t) holds the codes of
//   SELL_D16_GROUP (SELL_C32_GROUP) consecutive slots of thread t's two rows, [slot][row of the pair];
// CSR moves 12 + 4 per row.  Rows are still summed in stored column order, so y and the fused dot
// partials are bit-identical to the CSR kernel's.
#if defined(__HIPCC__)
#define OGL_HD __host__ __device__
#else
#define OGL_HD
#endif
constexpr int SELL_WAVE_ROWS = WAVE * ROWS_PER_THREAD;  // rows one wavefront of the SpMV owns
constexpr int SELL_WAVES = CHUNK_ROWS / SELL_WAVE_ROWS;  // wavefronts of a chunk's workgroup
enum SellMode : int { SELL_MODE_PATTERN = 0, SELL_MODE_OFFSET8 = 1, SELL_MODE_DELTA16 = 2, SELL_MODE_COL32 = 3 };
struct SellChunk {  // 32 bytes of 32/64-bit words: the whole header arrives by scalar loads (a 16-bit
                    // member would cost a vector load and a full memory round trip in the prologue)
    int64_t val_off;      // first value of the chunk (doubles); planes of CHUNK_ROWS, width() of them
    int64_t code_off;     // first code byte of the chunk (the row lengths sit at code_off - SELL_LEN_BYTES)
    int32_t dict_off;     // first table entry of the chunk (delta mode: the base of the first codes)
    uint32_t mode_len;    // SellMode | table ints << 16 (pattern mode: patterns x width; offset mode:
                          // <= SELL_MAX_DICT)
    // slots each wavefront (SELL_WAVE_ROWS rows) runs to = its own longest row; the planes beyond,
    // up to the chunk's longest row, are allocated but never read.  Within a wavefront every lane loads
    // up to the longer of ITS two rows only (lengths: SELL_LEN_BYTES in front of the codes; not in
    // pattern mode, where the rows of a chunk are (nearly) equally long)
    uint32_t w01, w23;    // wavefront 0 | wavefront 1 << 16,  wavefront 2 | wavefront 3 << 16
    OGL_HD int mode() const { return (int)(mode_len & 0xffffu); }
    OGL_HD int dict_len() const { return (int)(mode_len >> 16); }
    OGL_HD int wave_width(int wv) const { return (int)(((wv & 2 ? w23 : w01) >> (16 * (wv & 1))) & 0xffffu); }
    OGL_HD int width() const  // slots allocated per row = the chunk's longest row
    {
        int w = wave_width(0);
        for (int i = 1; i < SELL_WAVES; ++i) w = wave_width(i) > w ? wave_width(i) : w;
        return w;
    }
    // code bytes per thread: ROWS_PER_THREAD (pattern mode), ROWS_PER_THREAD x width rounded up to 16
    // (offset mode), 16 x groups of SELL_D16_GROUP / SELL_C32_GROUP slots (delta / column mode)
    OGL_HD int code_stride() const
    {
        const int w = width();
        if (mode() == SELL_MODE_PATTERN) return ROWS_PER_THREAD;
        if (mode() == SELL_MODE_OFFSET8) return (ROWS_PER_THREAD * w + 15) / 16 * 16;
        if (mode() == SELL_MODE_DELTA16) return 16 * ((w + 3) / 4);
        return 16 * ((w + 1) / 2);
    }
    void set(int mode, int dict_len, const int32_t (&wave_w)[SELL_WAVES])
    {
        mode_len = (uint32_t)mode | ((uint32_t)dict_len << 16);
        w01 = (uint32_t)wave_w[0] | ((uint32_t)wave_w[1] << 16);
        w23 = (uint32_t)wave_w[2] | ((uint32_t)wave_w[3] << 16);
    }
};
static_assert(sizeof(SellChunk) == 32, "SellChunk is read as two 16-byte words");
constexpr int SELL_D16_GROUP = 4, SELL_C32_GROUP = 2;  // slots per 16-byte code word
constexpr int SELL_MAX_DELTA16 = 65534;                // 0xFFFF marks a padding slot
constexpr int SELL_MAX_DICT = 255;        // offset mode: code 255 marks a padding slot
constexpr int SELL_TABLE_INTS = 2048;     // LDS table of the SpMV kernel (8 KB)
constexpr int32_t SELL_PAD_OFFSET = INT32_MIN;  // pattern mode: unused slot of a pattern
// half storage of a symmetric matrix on a banded pattern (SymLayout, host_matrix.hpp)
constexpr int SYM_MAX_OFFSETS = 4;        // diagonal + 3 legs: up to a 7-point stencil in 3-D
constexpr double SYM_MAX_PADDING = 1.15;  // plane slots / (diagonal + upper entries) above which full storage stays
// Half storage with exceptions (SymxLayout, host_matrix.hpp): the distances are chosen PER CHUNK (the three
// most frequent ones of its rows) and whatever does not fit -- the couplings across a block interface of a
// multi-block mesh, the long rows of a refinement shell -- is kept as explicit entries that the row sum merges in
// by column.  Header of one chunk, read by scalar loads:
struct SymxChunk {
    int64_t val_off;        // first value of the chunk's planes (doubles): [nd][CHUNK_ROWS]
    int64_t lo_base[3][2];  // lower entries at distance d[j]: A(r, r - d) = planes[lo_base[j][w] + ((r - d) % CHUNK_ROWS)],
                            // w = 0: r - d lies in the chunk of (first row - d), w = 1: in the next one; -1: not held there
    int32_t nd;             // planes: diagonal + distances (1..4)
    int32_t d[3];           // the distances of planes 1..nd-1, ascending
    int32_t ex_rp_off;      // this chunk's CHUNK_ROWS + 1 row pointers into the explicit entries, -1: it has none
    int32_t merge;          // 0: every row has at most one explicit entry before its first planar entry and at most one
                            // behind its last (the coupling across a block face): the lean kernel adds them ahead of /
                            // behind the plane walk; 1: anything else -- the general kernel merges by column
    int32_t chunk;          // device copy (kept in dispatch order: workgroup b reads header b): the chunk this header
                            // describes, -1: workgroup b has nothing to do
    int32_t ex_begin;       // its explicit entries: [ex_begin, ex_begin + ex_count) of the explicit arrays
    int32_t ex_count;
    int32_t pad_[1];
};
static_assert(sizeof(SymxChunk) == 96, "SymxChunk is read as six 16-byte words");
// mask byte of a row: bit 3 = diagonal, bit 3 - j / 3 + j = the entry at -d[j] / +d[j] (j = 1..3), bit 7 = the row
// has explicit entries
constexpr unsigned SYMX_EXTRAS_BIT = 0x80u;
constexpr int32_t SYMX_BEHIND_BIT = 1 << 16;  // SymxLayout::ex_lrow: the entry comes behind the row's last planar entry
constexpr int SYMX_LDS_ENTRIES = 1024;   // explicit entries of a chunk staged through LDS by its workgroup (more: read from memory)
constexpr double SYMX_MIN_PLANAR = 0.8;  // share of the entries that must live in planes for the layout to be used

// Matrix data that is read once per launch is streamed past the caches when matrix + the turn's five vectors do
// not fit the 256 MB Infinity Cache (measured: +3-4 % turn rate at 6-10 M rows; at 2 M rows / 1 M polyhedral
// cells, where everything fits, the hint costs 5-12 %: profiles/spmv_tune_r02.txt section 9)
constexpr double STREAM_MATRIX_ABOVE_BYTES = 288e6;
constexpr int SPMV_TUNE_MIN_ROWS = 65536;     // smaller systems: launch-bound, the compressed layout stays
constexpr int RENUMBER_AUTO_MIN_ROWS = 16384;  // config renumber = auto: smaller systems keep their numbering
// Padding is laid out but not read: every lane stops loading at the longer of its two rows (the row
// lengths of the chunk, one byte each, sit in front of its codes -- SELL_LEN_BYTES), so a padded slot
// costs HBM traffic only where it shares a 128-byte line (SELL_LINE_ROWS rows of a plane) with a slot in
// use, plus the instruction issue of the wavefront that runs to its longest row.  With `renumber` the
// rows of a wavefront are put longest first (host_matrix.cpp, choose_numbering), which keeps the lines in
// use dense; the gather does not change (a wavefront still works on the same SELL_WAVE_ROWS rows).
constexpr int SELL_LEN_BYTES = CHUNK_ROWS;  // row lengths in the planes, bytes, in front of the chunk's codes
constexpr int SELL_MAX_WIDTH = 255;         // (so that a length fits a byte; longer rows spill)
constexpr int SELL_LINE_ROWS = 16;          // rows of a plane in one 128-byte line
constexpr double SELL_SORT_MAX_SLOT_RATIO = 0.55;  // slot-major gather sectors per entry above which the rows
                                                 // are not sorted for the compressed layout (measured: a
                                                 // Voronoi mesh, 0.66, is faster on CSR-stream; mixed row
                                                 // lengths on a hex mesh, 0.44, on the compressed layout)
constexpr double SELL_ISSUE_COST = 0.1;     // a slot a wavefront steps over without loading, in units of a slot read
constexpr double SELL_MAX_PADDING = 1.15;  // (value slots in lines READ + 4 x spilled) / nnz above which the
                                           // CSR-stream kernel is at least as fast (12 B per entry, no padding)
constexpr double SELL_SPILL_COST = 4.0;     // cost of one spilled entry in units of one plane slot read
constexpr double SELL_MAX_ALLOC = 4.0;     // value slots ALLOCATED / nnz (planes no wavefront reads)

}  // namespace ogl
