/* include/icet_io.h -- scan file formats on the way into the hot path.
 *
 * Host-side loaders that produce exactly the matrix the reference hands to the ICET constructor, in the layout of
 * icet_hip.h (N x 3 float32 COLUMN-MAJOR, x[N] | y[N] | z[N], leading dimension N):
 *   ICET_FMT_OUSTER_CSV  `utils::loadPointCloudCSV(file, "ouster")`  /root/reference/src/utils.cpp:19-52
 *                        (third-party parser: include/csv.hpp, vendored by the reference).  Quirk kept: with
 *                        `header_row(1)` the parser drops lines 0-1, and the two `read_row` calls drop the next two
 *                        DATA rows as well, so the matrix starts at line 4; columns 8,9,10 are integers in mm, each
 *                        converted with float(int) / 1000.
 *   ICET_FMT_XYZ_TSV     the generic branch, utils.cpp:63-88: tab separated x y z, `stof` per field.  Quirk kept:
 *                        the parser's default format takes line 0 as a header, so the FIRST POINT IS LOST.
 *   ICET_FMT_NPY         NumPy .npy v1/v2/v3, shape (N, 3), '<f4' or '<f8', C or Fortran order -- the format of
 *                        src/sample_data/frame_80{4,5}.npy and the .npy clouds under python/point_clouds (README.md:28).
 *   ICET_FMT_KITTI_BIN   KITTI velodyne .bin: float32 records x, y, z, reflectance; the first three are kept
 *                        (README.md:28, src/fake_lidar.py:101-102).
 *   ICET_FMT_AUTO        by file name: .npy, .bin, .csv (Ouster), anything else tab separated.
 * Errors: unreadable file -> ICET_ERR_BAD_ARG (the reference prints to stderr and carries on with an empty matrix,
 * utils.cpp:15-17); malformed content -> ICET_ERR_UNSUPPORTED.  `*out` is malloc'ed; release it with icet_free_scan.
 */
#ifndef ICET_IO_H
#define ICET_IO_H
#include "icet_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef enum icet_scan_format {
    ICET_FMT_AUTO = 0, ICET_FMT_NPY = 1, ICET_FMT_OUSTER_CSV = 2, ICET_FMT_XYZ_TSV = 3, ICET_FMT_KITTI_BIN = 4
} icet_scan_format;

icet_status icet_load_scan(const char* path, int32_t format, float** out, int64_t* n);
void        icet_free_scan(float* scan);

/* Writes N x 3 column-major float32 (leading dimension ld) as a C-order '<f4' (N, 3) .npy -- the inverse of ICET_FMT_NPY. */
icet_status icet_save_scan_npy(const char* path, const float* scan, int64_t n, int64_t ld);

#ifdef __cplusplus
}
#endif
#endif /* ICET_IO_H */
