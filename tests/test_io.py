"""Scan file formats: include/icet_io.h against oracle/scan_io.py, and both against the
reference's own third-party CSV parser (oracle/_ref/csv_ref, built from /root/reference/include/csv.hpp when that tree is
present).  Bit-exact: the loaders only parse and convert."""
import os
import numpy as np
import pytest

from conftest import GOLDEN


def _ouster_file(path, n=300, seed=3, crlf=False, blank_lines=False, trailing_newline=True):
    rs = np.random.RandomState(seed)
    nl = "\r\n" if crlf else "\n"
    lines = ["# ouster pcap dump", ",".join("col%d" % i for i in range(14))]
    vals = rs.randint(-120000, 120000, size=(n, 14))
    for k, r in enumerate(vals):
        lines.append(",".join(str(int(v)) for v in r))
        if blank_lines and k % 50 == 7:
            lines.append("")
    txt = nl.join(lines) + (nl if trailing_newline else "")
    open(path, "w", newline="").write(txt)
    return vals


def _tsv_file(path, n=200, seed=4):
    rs = np.random.RandomState(seed)
    pts = (rs.normal(size=(n, 3)) * 20).astype(np.float32)
    with open(path, "w") as f:
        for p in pts:
            f.write("%.6f\t%.7g\t%.4e\n" % (p[0], p[1], p[2]))
    return pts


@pytest.mark.parametrize("crlf,blank,trail", [(False, False, True), (True, False, True), (False, True, False)])
def test_ouster_csv_matches_oracle_and_reference_parser(tmp_path, crlf, blank, trail):
    from icet_amd import api
    from oracle import scan_io
    p = str(tmp_path / "pcap_out_000261.csv")
    vals = _ouster_file(p, crlf=crlf, blank_lines=blank, trailing_newline=trail)
    got = api.load_scan(p)                                   # .csv -> Ouster
    ref = scan_io.load_ouster_csv(p)
    assert got.dtype == np.float32 and np.array_equal(got, ref)
    # the quirk: two header lines AND the first two data rows are gone (utils.cpp:21-29)
    assert len(got) == len(vals) - 2 and np.array_equal(got[0], vals[2, 8:11].astype(np.float32) / np.float32(1000))
    rows = scan_io.reference_parser_rows(p, "ouster")
    if rows is not None:                                     # the reference's own csv.hpp
        assert np.array_equal(rows.astype(np.int64).astype(np.float32) / np.float32(1000), got)


def test_xyz_tsv_matches_oracle_and_reference_parser(tmp_path):
    from icet_amd import api
    from oracle import scan_io
    p = str(tmp_path / "desk_test_20.txt")
    pts = _tsv_file(p)
    got = api.load_scan(p)
    assert np.array_equal(got, scan_io.load_xyz_tsv(p))
    assert len(got) == len(pts) - 1                          # the quirk: the first point is taken for a header (utils.cpp:65)
    rows = scan_io.reference_parser_rows(p, "xyz")
    if rows is not None:
        assert np.array_equal(rows.astype(np.float32), got)


@pytest.mark.parametrize("dtype,order", [("<f8", "C"), ("<f8", "F"), ("<f4", "C"), ("<f4", "F")])
def test_npy_all_layouts(tmp_path, dtype, order):
    from icet_amd import api
    from oracle import scan_io
    d = np.load(os.path.join(GOLDEN, "scans_frame_804_805.npz"))["scan1"][:5000].astype(np.float64)
    d[3] = [np.nan, -0.0, 1e-30]
    a = np.asarray(d.astype(dtype), order=order)
    p = str(tmp_path / "frame.npy")
    np.save(p, a)
    got = api.load_scan(p)
    assert np.array_equal(got.view(np.uint32), scan_io.load_npy(p).view(np.uint32))
    # and back: icet_save_scan_npy writes what np.load reads
    q = str(tmp_path / "out.npy")
    api.save_scan_npy(q, got)
    assert np.array_equal(np.load(q).view(np.uint32), got.view(np.uint32))


def test_npy_v2_header_and_kitti_bin(tmp_path):
    from icet_amd import api
    from oracle import scan_io
    import numpy.lib.format as fmt
    a = np.arange(30, dtype=np.float64).reshape(10, 3)
    p = str(tmp_path / "v2.npy")
    with open(p, "wb") as f:
        fmt.write_array(f, a, version=(2, 0))
    assert np.array_equal(api.load_scan(p), a.astype(np.float32))
    rs = np.random.RandomState(5)
    velo = rs.normal(size=(1234, 4)).astype(np.float32)
    b = str(tmp_path / "0000000001.bin")
    velo.tofile(b)
    assert np.array_equal(api.load_scan(b), scan_io.load_kitti_bin(b)) and np.array_equal(api.load_scan(b), velo[:, :3])


def test_loader_errors(tmp_path):
    from icet_amd import api
    with pytest.raises(api.IcetError) as e:
        api.load_scan(str(tmp_path / "missing.npy"))
    assert e.value.status == api.ICET_ERR_BAD_ARG
    bad = tmp_path / "bad.npy"; bad.write_bytes(b"not a numpy file")
    with pytest.raises(api.IcetError) as e:
        api.load_scan(str(bad))
    assert e.value.status == api.ICET_ERR_UNSUPPORTED
    i4 = str(tmp_path / "ints.npy"); np.save(i4, np.zeros((4, 3), np.int32))
    with pytest.raises(api.IcetError):
        api.load_scan(i4)
    short = tmp_path / "short.csv"; short.write_text("a\nb\n1,2,3\n4,5,6\n7,8,9\n")      # fewer than 11 columns
    with pytest.raises(api.IcetError):
        api.load_scan(str(short))
    empty = tmp_path / "empty.txt"; empty.write_text("")
    assert api.load_scan(str(empty)).shape == (0, 3)
    assert api.load_scan(str(tmp_path / "nothing.bin") if (tmp_path / "nothing.bin").write_bytes(b"") == 0 else "").shape == (0, 3)


@pytest.mark.skipif(not os.path.exists("/root/reference/src/sample_data/frame_804.npy"), reason="reference tree not present (GPU box)")
def test_reference_sample_data_loads_to_the_committed_fixture():
    """The reference's own sample files (float64, C order) through the C loader == the float32 scans the golden fixtures hold."""
    from icet_amd import api
    d = np.load(os.path.join(GOLDEN, "scans_frame_804_805.npz"))
    for name, key in (("frame_804.npy", "scan1"), ("frame_805.npy", "scan2")):
        got = api.load_scan("/root/reference/src/sample_data/" + name)
        assert np.array_equal(got.view(np.uint32), d[key].view(np.uint32))


def test_loaders_survive_garbage_under_sanitizers(tmp_path):
    """4000 random / half-valid files through every loader, compiled with AddressSanitizer + UBSan (host code only)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "io_fuzz")
    r = subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(root, "include"),
                        os.path.join(root, "tests", "cpp", "io_fuzz.cpp"), os.path.join(root, "icet_amd", "csrc", "icet_io.cpp"), "-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.gpu
def test_files_to_solution_end_to_end(tmp_path):
    """File -> icet_load_scan -> icet_solve on the GPU box (the demo's path, /root/reference/src/icet_cpp_demo.cpp:25-38,
    src/utils.cpp:12-91): the golden pair written as Ouster CSV (integer millimetres in columns 8-10), tab-separated xyz, .npy (float64,
    Fortran order -- the layout of the reference's sample data is float64) and KITTI .bin; every pair of files must load to exactly the
    rows the format keeps (the CSV branches drop 4 lines / the first point) and solve to the bits of the direct-array solve of those rows;
    the millimetre-quantised pair is also held to the oracle."""
    import icet_amd
    from icet_amd import api
    from oracle import pyoracle as po
    d = np.load(os.path.join(GOLDEN, "scans_frame_804_805.npz"))
    scans = [d["scan1"], d["scan2"]]
    ctx = icet_amd.Context(0)

    def solve(p, q):
        return ctx.solve(p, q, 7, np.zeros(6), 24, 75)

    def check(paths, expected, fmt=api.FMT_AUTO):
        loaded = [api.load_scan(p, fmt) for p in paths]
        for got, want in zip(loaded, expected):
            assert got.dtype == np.float32 and got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
        r, r0 = solve(*loaded), solve(*expected)
        for key in ("X", "pred_stds", "cov"):
            assert np.array_equal(r[key], r0[key]) and np.isfinite(r[key]).all(), key
        return r

    # Ouster CSV: 14 integer columns, x / y / z in millimetres in columns 8-10; two header lines, and the reader drops two more rows
    paths, want = [], []
    for k, s in enumerate(scans):
        mm = np.rint(s.astype(np.float64) * 1000.0).astype(np.int64)
        p = str(tmp_path / ("pcap_out_%06d.csv" % k))
        with open(p, "w") as f:
            f.write("# ouster dump\n" + ",".join("c%d" % i for i in range(14)) + "\n")
            for row in mm:
                f.write("0,0,0,0,0,0,0,0,%d,%d,%d,0,0,0\n" % (row[0], row[1], row[2]))
        paths.append(p); want.append((mm[2:].astype(np.float32) / np.float32(1000)))
    r_mm = check(paths, want)
    ref = po.solve(want[0], want[1])
    assert np.abs(r_mm["X"][:3] - ref["X"][:3]).max() <= 2e-4 and np.abs(r_mm["X"][3:] - ref["X"][3:]).max() <= 2e-5
    # tab-separated xyz: %.9g round-trips float32; the first point is taken for a header
    paths, want = [], []
    for k, s in enumerate(scans):
        p = str(tmp_path / ("desk_%d.txt" % k))
        with open(p, "w") as f:
            for row in s:
                f.write("%.9g\t%.9g\t%.9g\n" % (row[0], row[1], row[2]))
        paths.append(p); want.append(s[1:].copy())
    check(paths, want)
    # .npy float64 Fortran order, and KITTI .bin (x, y, z, reflectance)
    paths = []
    for k, s in enumerate(scans):
        p = str(tmp_path / ("frame_%d.npy" % k)); np.save(p, np.asfortranarray(s.astype(np.float64))); paths.append(p)
    r_npy = check(paths, scans)
    paths = []
    for k, s in enumerate(scans):
        p = str(tmp_path / ("%010d.bin" % k)); np.concatenate([s, np.full((len(s), 1), 0.5, np.float32)], 1).astype(np.float32).tofile(p); paths.append(p)
    r_bin = check(paths, scans)
    g = dict(np.load(os.path.join(GOLDEN, "golden_frame_804_805.npz")))
    assert np.array_equal(r_npy["X"], r_bin["X"]) and np.abs(r_npy["X"][:3] - g["X"][:3]).max() <= 2e-4
    ctx.close()
