// tests/cpp/io_fuzz.cpp -- random and half-valid files through every loader of include/icet_io.h; built with ASan + UBSan by tests/test_io.py
// (CPU only: sanitizers are not available for the GPU build on this pool).
#include "icet_io.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
int main(int argc, char** argv) {
    unsigned seed = 1;
    for (int it = 0; it < 4000; it++) {
        seed = seed * 1664525u + 1013904223u;
        const int fmt = 1 + (seed >> 8) % 4;
        size_t len = (seed >> 12) % 600;
        std::vector<unsigned char> buf(len);
        for (size_t i = 0; i < len; i++) { seed = seed * 1664525u + 1013904223u; buf[i] = (unsigned char)(seed >> 24); }
        const char* tmpl[] = {"", "\x93NUMPY\x01\x00\x76\x00{'descr': '<f8', 'fortran_order': False, 'shape': (7, 3), }", "a,b\nc,d\n1,2,3,4,5,6,7,8,9,10,11\n", "1\t2\t3\n4\t5\t6\n", ""};
        if ((seed >> 5) & 1) { size_t tl = std::strlen(tmpl[fmt]); if (fmt == 1) tl = 60; for (size_t i = 0; i < tl && i < len; i++) buf[i] = (unsigned char)tmpl[fmt][i]; }
        FILE* f = std::fopen("fz.bin", "wb"); if (len) std::fwrite(buf.data(), 1, len, f); std::fclose(f);
        float* out = nullptr; int64_t n = -1;
        icet_status st = icet_load_scan("fz.bin", fmt, &out, &n);
        if (st == ICET_OK) { volatile float s = 0; for (int64_t i = 0; i < 3 * n; i++) s += out[i]; }
        icet_free_scan(out);
    }
    std::puts("fuzz ok");
    return 0;
}
