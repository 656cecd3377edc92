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
    // Hostile .npy headers: shapes whose byte count wraps size_t, and shapes larger than the payload.
    // Every one of them must be REFUSED without touching memory (this binary runs under ASan + UBSan).
    const char* shapes[] = {"(4611686018427387904, 3)", "(1537228672809129302, 3)", "(9223372036854775807, 3)", "(1000000, 3)", "(3, 3)", "(-5, 3)", "(2, 4)"};
    const char* descrs[] = {"<f4", "<f8"};
    const char* orders[] = {"False", "True"};
    for (const char* sh : shapes) for (const char* de : descrs) for (const char* fo : orders) {
        char dict[256];
        int len = std::snprintf(dict, sizeof(dict), "{'descr': '%s', 'fortran_order': %s, 'shape': %s, }", de, fo, sh);
        std::vector<unsigned char> file = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(len & 0xff), (unsigned char)(len >> 8)};
        file.insert(file.end(), dict, dict + len);
        for (int i = 0; i < 6; i++) { const float v = 1.5f * (float)i; const unsigned char* b = reinterpret_cast<const unsigned char*>(&v); file.insert(file.end(), b, b + 4); }   // 6-float payload
        FILE* f = std::fopen("fz.bin", "wb"); std::fwrite(file.data(), 1, file.size(), f); std::fclose(f);
        float* out = nullptr; int64_t n = -1;
        icet_status st = icet_load_scan("fz.bin", ICET_FMT_NPY, &out, &n);
        if (st == ICET_OK) { std::printf("accepted a header that claims more rows than the file holds: %s %s\n", sh, de); return 1; }
        if (out != nullptr || n != 0) { std::puts("outputs not reset on failure"); return 1; }
    }
    std::puts("fuzz ok");
    return 0;
}
