// host_sanitize.cpp -- the host-side layout builders (per-chunk half storage, compressed chunked ELL) on patterns read
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
    }
    return 0;
}
