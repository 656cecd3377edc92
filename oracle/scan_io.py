"""oracle/scan_io.py -- TEST INFRASTRUCTURE.  NumPy / pure-Python restatement of the reference's scan loaders (SURVEY.md
section 8, row f2); only tests/ may import it.  Each function returns the N x 3 float32 matrix the reference would hand to
the ICET constructor.

  load_ouster_csv   utils::loadPointCloudCSV(file, "ouster")   /root/reference/src/utils.cpp:19-52
  load_xyz_tsv      the generic branch                          /root/reference/src/utils.cpp:63-88
  load_npy          np.load(...) as the Python side does        /root/reference/README.md:28, src/sample_data/*.npy
  load_kitti_bin    pykitti's velodyne reader: float32 (N, 4)   /root/reference/src/fake_lidar.py:101-102

The two CSV functions depend on the behaviour of the reference's third-party parser (include/csv.hpp); that behaviour is
PINNED by running the parser itself (oracle/ref_csv_harness.cpp -> oracle/_ref/csv_ref) in tests/test_io.py.
"""
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_BIN = os.path.join(_HERE, "_ref", "csv_ref")


def _lines(path):
    with open(path, "rb") as f:
        txt = f.read().decode("ascii", "replace")
    out = txt.split("\n")
    if out and out[-1] == "":
        out.pop()
    return [ln[:-1] if ln.endswith("\r") else ln for ln in out]


def load_ouster_csv(path):
    rows = []
    for ln in _lines(path)[4:]:                  # header_row(1) hides lines 0-1; two read_row calls drop lines 2-3 (utils.cpp:21-29)
        if ln == "":
            continue
        f = ln.split(",")
        rows.append([int(f[8]), int(f[9]), int(f[10])])                  # utils.cpp:35-37
    a = np.asarray(rows, np.int64).reshape(-1, 3)
    return a.astype(np.float32) / np.float32(1000)                       # utils.cpp:51


def load_xyz_tsv(path):
    rows, header = [], False
    for ln in _lines(path):
        if ln == "":
            continue
        if not header:                           # default CSVFormat: the first line is the header (utils.cpp:65)
            header = True
            continue
        f = ln.split("\t")
        rows.append([np.float32(f[0]), np.float32(f[1]), np.float32(f[2])])   # stof
    return np.asarray(rows, np.float32).reshape(-1, 3)


def load_npy(path):
    return np.load(path).astype(np.float32)


def load_kitti_bin(path):
    return np.fromfile(path, dtype=np.float32).reshape(-1, 4)[:, :3].copy()


def reference_parser_rows(path, mode):
    """Rows as the reference's own csv.hpp yields them (oracle/_ref/csv_ref); None if the harness is not built."""
    if not os.path.exists(REF_BIN):
        return None
    out = subprocess.run([REF_BIN, path, mode], check=True, capture_output=True, text=True).stdout
    rows = [ln.split() for ln in out.splitlines() if ln.strip()]
    return np.asarray(rows, np.float64).reshape(-1, 3)
