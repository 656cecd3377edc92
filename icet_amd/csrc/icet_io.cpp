// icet_amd/csrc/icet_io.cpp -- host-side scan loaders of include/icet_io.h.
// Plain C++ (no HIP, no third-party parser): the reference's loaders are utils::loadPointCloudCSV
// (/root/reference/src/utils.cpp:12-91, on top of the vendored include/csv.hpp) plus NumPy / KITTI files read by its
// Python side (README.md:28, src/fake_lidar.py).  The row-skipping behaviour of the two CSV modes was pinned against
// the reference's own csv.hpp (oracle/ref_csv_harness.cpp -> oracle/_ref/csv_ref, tests/test_io.py).
#include "../../include/icet_io.h"

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

bool read_file(const char* path, std::vector<char>& buf) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz < 0) { std::fclose(f); return false; }
    buf.resize((size_t)sz);
    const size_t got = sz ? std::fread(buf.data(), 1, (size_t)sz, f) : 0;
    std::fclose(f);
    return got == (size_t)sz;
}

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// split the buffer into lines (LF or CRLF); a trailing newline does not make an extra empty line
void split_lines(const std::vector<char>& buf, std::vector<std::pair<const char*, const char*>>& lines) {
    const char* p = buf.data(); const char* end = p + buf.size();
    while (p < end) {
        const char* q = static_cast<const char*>(std::memchr(p, '\n', (size_t)(end - p)));
        const char* e = q ? q : end;
        const char* stop = (e > p && e[-1] == '\r') ? e - 1 : e;
        lines.emplace_back(p, stop);
        p = q ? q + 1 : end;
    }
}

// the k-th field of a delimiter-separated line (no quoting: lidar dumps are plain numbers); false if absent
bool field(const char* b, const char* e, char delim, int k, const char*& fb, const char*& fe) {
    const char* p = b;
    for (int i = 0; i < k; i++) {
        const char* q = static_cast<const char*>(std::memchr(p, delim, (size_t)(e - p)));
        if (!q) return false;
        p = q + 1;
    }
    const char* q = static_cast<const char*>(std::memchr(p, delim, (size_t)(e - p)));
    fb = p; fe = q ? q : e;
    return true;
}

icet_status alloc_out(int64_t n, float** out) {
    *out = nullptr;
    if (n < 0 || (uint64_t)n > SIZE_MAX / (3 * sizeof(float))) return ICET_ERR_NOMEM;      // 12 * n must not wrap
    *out = static_cast<float*>(std::malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1)));
    return *out ? ICET_OK : ICET_ERR_NOMEM;
}

icet_status load_ouster(const std::vector<char>& buf, float** out, int64_t* n) {
    std::vector<std::pair<const char*, const char*>> lines; split_lines(buf, lines);
    // utils.cpp:21 header_row(1): lines 0 and 1 never reach the caller; :26-29 two read_row calls: lines 2 and 3 discarded
    const size_t first = 4;
    std::vector<size_t> rows;
    for (size_t i = first; i < lines.size(); i++) if (lines[i].second > lines[i].first) rows.push_back(i);     // the parser skips empty lines
    icet_status s = alloc_out((int64_t)rows.size(), out); if (s != ICET_OK) return s;
    const int64_t N = (int64_t)rows.size();
    for (int64_t r = 0; r < N; r++) {
        for (int c = 0; c < 3; c++) {
            const char *fb, *fe;
            if (!field(lines[rows[r]].first, lines[rows[r]].second, ',', 8 + c, fb, fe)) { std::free(*out); *out = nullptr; return ICET_ERR_UNSUPPORTED; }
            std::string tok(fb, fe);
            char* endp = nullptr; errno = 0;
            const long v = std::strtol(tok.c_str(), &endp, 10);
            if (endp == tok.c_str() || errno) { std::free(*out); *out = nullptr; return ICET_ERR_UNSUPPORTED; }
            (*out)[c * N + r] = static_cast<float>(static_cast<int>(v)) / 1000.0f;       // utils.cpp:35-37,51: float(int) / 1000
        }
    }
    *n = N;
    return ICET_OK;
}

icet_status load_tsv(const std::vector<char>& buf, float** out, int64_t* n) {
    std::vector<std::pair<const char*, const char*>> lines; split_lines(buf, lines);
    std::vector<size_t> rows;
    bool header_taken = false;                                       // utils.cpp:65: default format -> first non-empty line is the header
    for (size_t i = 0; i < lines.size(); i++) {
        if (lines[i].second == lines[i].first) continue;
        if (!header_taken) { header_taken = true; continue; }
        rows.push_back(i);
    }
    icet_status s = alloc_out((int64_t)rows.size(), out); if (s != ICET_OK) return s;
    const int64_t N = (int64_t)rows.size();
    for (int64_t r = 0; r < N; r++) {
        for (int c = 0; c < 3; c++) {
            const char *fb, *fe;
            if (!field(lines[rows[r]].first, lines[rows[r]].second, '\t', c, fb, fe)) { std::free(*out); *out = nullptr; return ICET_ERR_UNSUPPORTED; }
            std::string tok(fb, fe);
            char* endp = nullptr;
            const float v = std::strtof(tok.c_str(), &endp);           // utils.cpp:70-72: stof
            if (endp == tok.c_str()) { std::free(*out); *out = nullptr; return ICET_ERR_UNSUPPORTED; }
            (*out)[c * N + r] = v;
        }
    }
    *n = N;
    return ICET_OK;
}

icet_status load_kitti(const std::vector<char>& buf, float** out, int64_t* n) {
    if (buf.size() % 16 != 0) return ICET_ERR_UNSUPPORTED;
    const int64_t N = (int64_t)(buf.size() / 16);
    icet_status s = alloc_out(N, out); if (s != ICET_OK) return s;
    const float* rec = reinterpret_cast<const float*>(buf.data());
    for (int64_t r = 0; r < N; r++) for (int c = 0; c < 3; c++) (*out)[c * N + r] = rec[4 * r + c];
    *n = N;
    return ICET_OK;
}

icet_status load_npy(const std::vector<char>& buf, float** out, int64_t* n) {
    if (buf.size() < 10 || std::memcmp(buf.data(), "\x93NUMPY", 6) != 0) return ICET_ERR_UNSUPPORTED;
    const int major = (unsigned char)buf[6];
    size_t hlen, hoff;
    if (major == 1) { hlen = (unsigned char)buf[8] | ((size_t)(unsigned char)buf[9] << 8); hoff = 10; }
    else if (major == 2 || major == 3) {
        if (buf.size() < 12) return ICET_ERR_UNSUPPORTED;
        hlen = 0; for (int i = 0; i < 4; i++) hlen |= (size_t)(unsigned char)buf[8 + i] << (8 * i);
        hoff = 12;
    } else return ICET_ERR_UNSUPPORTED;
    if (hoff + hlen > buf.size()) return ICET_ERR_UNSUPPORTED;
    const std::string hdr(buf.data() + hoff, hlen);
    auto value_after = [&](const char* key) -> std::string {
        const size_t k = hdr.find(key);
        if (k == std::string::npos) return "";
        size_t p = hdr.find(':', k);
        if (p == std::string::npos) return "";
        p++;
        while (p < hdr.size() && hdr[p] == ' ') p++;
        return hdr.substr(p);
    };
    const std::string descr = value_after("'descr'"), order = value_after("'fortran_order'"), shape = value_after("'shape'");
    int width = 0;
    if (descr.compare(0, 5, "'<f4'") == 0) width = 4; else if (descr.compare(0, 5, "'<f8'") == 0) width = 8; else return ICET_ERR_UNSUPPORTED;
    const bool fortran = order.compare(0, 4, "True") == 0;
    long long d0 = -1, d1 = -1;
    if (std::sscanf(shape.c_str(), "(%lld, %lld", &d0, &d1) != 2 || d0 < 0 || d1 != 3) return ICET_ERR_UNSUPPORTED;
    // The shape comes from the file: bound it by DIVISION against the payload actually present before any multiply
    // (a header claiming 2^62 rows would otherwise wrap 12 * N to 0 and pass a product-based check).
    const uint64_t avail = (uint64_t)(buf.size() - hoff - hlen);
    if ((uint64_t)d0 > avail / (3u * (uint64_t)width)) return ICET_ERR_UNSUPPORTED;
    const int64_t N = d0;
    const char* data = buf.data() + hoff + hlen;
    icet_status s = alloc_out(N, out); if (s != ICET_OK) return s;
    for (int64_t r = 0; r < N; r++)
        for (int c = 0; c < 3; c++) {
            const size_t idx = fortran ? (size_t)c * N + r : (size_t)r * 3 + c;
            float v;
            if (width == 4) { std::memcpy(&v, data + idx * 4, 4); }
            else { double d; std::memcpy(&d, data + idx * 8, 8); v = static_cast<float>(d); }      // what `astype(float32)` does
            (*out)[c * N + r] = v;
        }
    *n = N;
    return ICET_OK;
}

}  // namespace

extern "C" {

icet_status icet_load_scan(const char* path, int32_t format, float** out, int64_t* n) {
    if (!path || !out || !n) return ICET_ERR_BAD_ARG;
    *out = nullptr; *n = 0;
    if (format == ICET_FMT_AUTO) {
        const std::string p(path);
        format = ends_with(p, ".npy") ? ICET_FMT_NPY : ends_with(p, ".bin") ? ICET_FMT_KITTI_BIN : ends_with(p, ".csv") ? ICET_FMT_OUSTER_CSV : ICET_FMT_XYZ_TSV;
    }
    if (format < ICET_FMT_NPY || format > ICET_FMT_KITTI_BIN) return ICET_ERR_BAD_ARG;
    // std::vector / std::string allocations below may throw: nothing may unwind across the C ABI
    try {
        std::vector<char> buf;
        if (!read_file(path, buf)) return ICET_ERR_BAD_ARG;
        switch (format) {
            case ICET_FMT_NPY: return load_npy(buf, out, n);
            case ICET_FMT_OUSTER_CSV: return load_ouster(buf, out, n);
            case ICET_FMT_XYZ_TSV: return load_tsv(buf, out, n);
            default: return load_kitti(buf, out, n);
        }
    } catch (const std::bad_alloc&) {
        if (*out) { std::free(*out); *out = nullptr; }
        *n = 0;
        return ICET_ERR_NOMEM;
    } catch (...) {
        if (*out) { std::free(*out); *out = nullptr; }
        *n = 0;
        return ICET_ERR_UNSUPPORTED;
    }
}

void icet_free_scan(float* scan) { std::free(scan); }

icet_status icet_save_scan_npy(const char* path, const float* scan, int64_t n, int64_t ld) {
    if (!path || n < 0 || ld < n || (n > 0 && !scan)) return ICET_ERR_BAD_ARG;
    if ((uint64_t)n > SIZE_MAX / (3 * sizeof(float))) return ICET_ERR_NOMEM;
    try {
    FILE* f = std::fopen(path, "wb");
    if (!f) return ICET_ERR_BAD_ARG;
    char dict[128];
    int len = std::snprintf(dict, sizeof(dict), "{'descr': '<f4', 'fortran_order': False, 'shape': (%lld, 3), }", (long long)n);
    std::string hdr(dict, (size_t)len);
    while ((10 + hdr.size() + 1) % 64 != 0) hdr.push_back(' ');
    hdr.push_back('\n');
    const unsigned char pre[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(hdr.size() & 0xff), (unsigned char)(hdr.size() >> 8)};
    bool ok = std::fwrite(pre, 1, 10, f) == 10 && std::fwrite(hdr.data(), 1, hdr.size(), f) == hdr.size();
    std::vector<float> row(3 * (size_t)(n > 0 ? n : 1));
    for (int64_t r = 0; r < n; r++) for (int c = 0; c < 3; c++) row[(size_t)r * 3 + c] = scan[c * ld + r];
    if (n) ok = ok && std::fwrite(row.data(), sizeof(float), 3 * (size_t)n, f) == 3 * (size_t)n;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? ICET_OK : ICET_ERR_BAD_ARG;
    } catch (const std::bad_alloc&) { return ICET_ERR_NOMEM; }
}

}  // extern "C"
