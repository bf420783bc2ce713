"""util::Cloud PCD reader of the host mirror (pcl::io::loadPCDFile, cloud.cpp:643-660): ASCII and
uncompressed binary layouts, extra fields, NaN rows removed."""
import numpy as np

from gpd_amd import hostlib


def _write(path, xyz, normals, binary, extra_rgb=False):
    n = len(xyz)
    fields = ["x", "y", "z"] + (["rgb"] if extra_rgb else []) + (["normal_x", "normal_y", "normal_z"] if normals is not None else [])
    sizes = ["4"] * len(fields)
    types = ["F", "F", "F"] + (["U"] if extra_rgb else []) + (["F"] * 3 if normals is not None else [])
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH %d\nHEIGHT 1\n"
           "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n" % (" ".join(fields), " ".join(sizes), " ".join(types),
                                                              " ".join(["1"] * len(fields)), n, n, "binary" if binary else "ascii"))
    with open(path, "wb") as f:
        f.write(hdr.encode())
        for i in range(n):
            if binary:
                rec = xyz[i].astype("<f4").tobytes()
                if extra_rgb:
                    rec += np.uint32(0x00ff8040 + i).tobytes()
                if normals is not None:
                    rec += normals[i].astype("<f4").tobytes()
                f.write(rec)
            else:
                vals = ["%.9g" % v for v in xyz[i]] + (["%d" % (4286578688 + i)] if extra_rgb else [])
                if normals is not None:
                    vals += ["%.9g" % v for v in normals[i]]
                f.write((" ".join(vals) + "\n").encode())


def test_pcd_ascii_and_binary_roundtrip(tmp_path):
    rng = np.random.RandomState(0)
    xyz = rng.normal(0, 0.3, (500, 3)).astype(np.float32)
    nrm = rng.normal(0, 1, (500, 3)).astype(np.float32)
    xyz[17] = np.nan  # removed on load (Cloud::removeNans)
    keep = np.ones(500, bool)
    keep[17] = False
    for binary in (False, True):
        for normals in (None, nrm):
            for rgb in (False, True):
                p = tmp_path / ("c_%d_%d_%d.pcd" % (binary, normals is not None, rgb))
                _write(str(p), xyz, normals, binary, rgb)
                gx, gn = hostlib.load_pcd(p)
                assert np.array_equal(gx, xyz[keep]), (binary, rgb)
                if normals is None:
                    assert gn is None
                else:
                    assert np.array_equal(gn, nrm[keep])


def test_pcd_unsupported_or_missing_is_empty(tmp_path):
    p = tmp_path / "z.pcd"
    p.write_bytes(b"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary_compressed\n\x00\x00")
    gx, _ = hostlib.load_pcd(p)
    assert len(gx) == 0
    gx, _ = hostlib.load_pcd(tmp_path / "does_not_exist.pcd")
    assert len(gx) == 0


def test_normals_csv(tmp_path):
    import ctypes as C
    rng = np.random.RandomState(1)
    xyz = rng.normal(0, 0.3, (40, 3)).astype(np.float32)
    nrm = rng.normal(0, 1, (40, 3)).astype(np.float32)
    pcd, csv = tmp_path / "c.pcd", tmp_path / "n.csv"
    _write(str(pcd), xyz, None, False)
    csv.write_text("\n".join(",".join("%.9g" % v for v in nrm[:, r]) for r in range(3)) + "\n")
    out = np.zeros((40, 3), np.float32)
    L = hostlib.lib()
    L.gpd_host_load_normals_csv.restype = C.c_int
    n = L.gpd_host_load_normals_csv(str(pcd).encode(), str(csv).encode(), out.ctypes.data_as(C.c_void_p), 40)
    assert n == 40 and np.array_equal(out, nrm)


def test_cli_argument_errors_without_a_gpu(tmp_path):
    """The host tools check their arguments and inputs before they touch the device, with the reference's
    messages and its -1 exit status (src/tests/test_grasp_image.cpp:20-44, src/detect_grasps.cpp)."""
    import os
    import subprocess
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpd_amd", "host")
    tool = os.path.join(host, "test_grasp_image")
    assert os.path.exists(tool), "run __graft_entry__.build()"
    out = subprocess.run([tool], capture_output=True, text=True, timeout=60)
    assert out.returncode == 255 and "ERROR: Not enough arguments given!" in out.stdout
    out = subprocess.run([tool, str(tmp_path / "missing.pcd"), "0", "0"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 255 and "Input point cloud is empty or does not exist" in out.stdout
    pcd = tmp_path / "three.pcd"
    _write(str(pcd), np.eye(3, dtype=np.float32), None, False)
    out = subprocess.run([tool, str(pcd), "7", "0"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 255 and "Sample index is larger than the number of points" in out.stdout
    for name in ("detect_grasps", "generate_candidates", "label_grasps", "cem_detect_grasps"):
        out = subprocess.run([os.path.join(host, name)], capture_output=True, text=True, timeout=60)
        assert out.returncode == 255 and "Not enough input arguments" in out.stdout, name


def test_pcd_reader_refuses_malformed_headers(tmp_path):
    """An untrusted PCD header (negative / huge SIZE, COUNT 0, absurd POINTS, mismatched field lists) yields an empty
    cloud and a message, never a crash or a giant allocation."""
    from gpd_amd import hostlib
    base = "VERSION 0.7\nFIELDS x y z\n%s\nTYPE F F F\n%s\nWIDTH 2\nHEIGHT 1\nPOINTS %s\nDATA %s\n"
    bad = [
        base % ("SIZE -4 4 4", "COUNT 1 1 1", "2", "binary"),
        base % ("SIZE 4 4 400000000", "COUNT 1 1 1", "2", "binary"),
        base % ("SIZE 4 4 4", "COUNT 0 1 1", "2", "binary"),
        base % ("SIZE 4 4 4", "COUNT 1 1 1", "99999999999999", "binary"),
        base % ("SIZE 4 4", "COUNT 1 1 1", "2", "binary"),
    ]
    for i, head in enumerate(bad):
        path = tmp_path / ("bad%d.pcd" % i)
        path.write_bytes(head.encode() + b"\x00" * 24)
        xyz, nrm = hostlib.load_pcd(str(path))
        assert len(xyz) == 0, i
    # an ASCII field with COUNT 2 takes two columns
    path = tmp_path / "count2.pcd"
    path.write_text("VERSION 0.7\nFIELDS x y z extra\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 2\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA ascii\n"
                    "1 2 3 9 9\n4 5 6 8 8\n")
    xyz, nrm = hostlib.load_pcd(str(path))
    assert xyz.tolist() == [[1, 2, 3], [4, 5, 6]] and nrm is None
