"""util::Cloud PCD reader of the host mirror (pcl::io::loadPCDFile, cloud.cpp:643-660): ASCII, binary and
binary_compressed (LZF, fields-major) layouts, extra fields, NaN rows removed."""
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


def _lzf_compress(data):
    """A small greedy LZF encoder (liblzf's stream format): literal runs of up to 32 bytes, back references of 3..264
    bytes at distances up to 8192 found through a 3-byte hash of the last position."""
    out = bytearray()
    lit = bytearray()
    last = {}
    i, n = 0, len(data)

    def flush():
        for k in range(0, len(lit), 32):
            run = lit[k:k + 32]
            out.append(len(run) - 1)
            out.extend(run)
        lit.clear()

    while i < n:
        key = bytes(data[i:i + 3])
        j = last.get(key, -1) if len(key) == 3 else -1
        if len(key) == 3:
            last[key] = i
        if j >= 0 and i - j <= 8192:
            length = 3
            while i + length < n and length < 264 and data[j + length] == data[i + length]:
                length += 1
            flush()
            dist, l2 = i - j - 1, length - 2
            if l2 < 7:
                out.append((l2 << 5) | (dist >> 8))
            else:
                out.append((7 << 5) | (dist >> 8))
                out.append(l2 - 7)
            out.append(dist & 255)
            i += length
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def _lzf_decompress(comp):
    out = bytearray()
    i = 0
    while i < len(comp):
        c = comp[i]
        i += 1
        if c < 32:
            out += comp[i:i + c + 1]
            i += c + 1
        else:
            ln = c >> 5
            if ln == 7:
                ln += comp[i]
                i += 1
            d = ((c & 31) << 8 | comp[i]) + 1
            i += 1
            for _ in range(ln + 2):
                out.append(out[-d])
    return bytes(out)


def _write_compressed(path, columns, fields, sizes, types, counts, n, comp=None):
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH %d\nHEIGHT 1\n"
           "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary_compressed\n" % (" ".join(fields), " ".join(map(str, sizes)), " ".join(types),
                                                                             " ".join(map(str, counts)), n, n))
    raw = b"".join(columns)  # fields-major: every field's values of all points back to back
    if comp is None:
        comp = _lzf_compress(raw)
        assert _lzf_decompress(comp) == raw
    with open(path, "wb") as f:
        f.write(hdr.encode() + np.array([len(comp), len(raw)], "<u4").tobytes() + comp)
    return raw, comp


def test_pcd_binary_compressed(tmp_path):
    """DATA binary_compressed as PCDWriter::writeBinaryCompressed lays it out (what pcl_viewer / ROS tools save by
    default): sizes, one LZF stream, fields-major.  A lattice cloud compresses through real back references (the
    stream is checked to be shorter than the raw bytes), an rgb field and a COUNT-2 field sit between the
    coordinates and the normals, NaN rows go."""
    rng = np.random.RandomState(3)
    n = 700
    xyz = (rng.randint(-40, 40, (n, 3)) * 0.003).astype(np.float32)  # 3 mm lattice: few distinct values per column
    nrm = np.zeros((n, 3), np.float32)
    nrm[:, rng.randint(0, 3)] = 1.0
    xyz[5] = np.nan
    keep = np.ones(n, bool)
    keep[5] = False
    rgb = (0x00ff8040 + np.arange(n)).astype("<u4")
    extra = rng.randint(0, 9, (n, 2)).astype("<i2")
    cols = [xyz[:, 0].astype("<f4").tobytes(), xyz[:, 1].astype("<f4").tobytes(), xyz[:, 2].astype("<f4").tobytes(), rgb.tobytes(),
            extra.tobytes(), nrm[:, 0].astype("<f4").tobytes(), nrm[:, 1].astype("<f4").tobytes(), nrm[:, 2].astype("<f4").tobytes()]
    p = tmp_path / "comp.pcd"
    raw, comp = _write_compressed(str(p), cols, ["x", "y", "z", "rgb", "pair", "normal_x", "normal_y", "normal_z"], [4, 4, 4, 4, 2, 4, 4, 4],
                                  ["F", "F", "F", "U", "I", "F", "F", "F"], [1, 1, 1, 1, 2, 1, 1, 1], n)
    assert len(comp) < len(raw) // 2 and any(c >= 32 for c in comp[:64])
    gx, gn = hostlib.load_pcd(p)
    assert np.array_equal(gx, xyz[keep]) and np.array_equal(gn, nrm[keep])
    # a literal-only stream (incompressible data) and an overlapping back reference (run-length form: distance 1)
    p2 = tmp_path / "lit.pcd"
    xyz2 = rng.normal(0, 0.3, (9, 3)).astype(np.float32)
    raw2 = b"".join(xyz2[:, k].astype("<f4").tobytes() for k in range(3))
    lit = b"".join(bytes([len(raw2[k:k + 32]) - 1]) + raw2[k:k + 32] for k in range(0, len(raw2), 32))
    _write_compressed(str(p2), [raw2], ["x", "y", "z"], [4, 4, 4], ["F", "F", "F"], [1, 1, 1], 9, comp=lit)
    gx, gn = hostlib.load_pcd(p2)
    assert np.array_equal(gx, xyz2) and gn is None
    p3 = tmp_path / "rle.pcd"
    zeros = np.zeros((50, 3), np.float32)
    rle = bytes([0, 0]) + bytes([(7 << 5) | 0, 255, 0]) * 2 + bytes([(7 << 5) | 0, 600 - 1 - 2 * 264 - 9, 0])
    assert _lzf_decompress(rle) == zeros.tobytes()
    _write_compressed(str(p3), [zeros.tobytes()], ["x", "y", "z"], [4, 4, 4], ["F", "F", "F"], [1, 1, 1], 50, comp=rle)
    gx, _ = hostlib.load_pcd(p3)
    assert np.array_equal(gx, zeros)


def test_pcd_binary_compressed_random_streams(tmp_path):
    """Round trips of the LZF reader on clouds drawn from small alphabets (many back references of every length class,
    overlapping ones included) and from noise (literal runs only): 60 files."""
    rng = np.random.RandomState(11)
    for case in range(60):
        n = int(rng.randint(1, 400))
        kind = case % 3
        if kind == 0:
            xyz = rng.choice(np.array([0.0, 0.003, -0.006, 1.5], np.float32), (n, 3))   # long repeats
        elif kind == 1:
            xyz = (rng.randint(-3, 3, (n, 3)) * 0.25).astype(np.float32)
            xyz[n // 2:] = xyz[: n - n // 2]                                         # a far back reference
        else:
            xyz = rng.normal(0, 1, (n, 3)).astype(np.float32)                        # incompressible
        cols = [xyz[:, k].astype("<f4").tobytes() for k in range(3)]
        p = tmp_path / ("r%d.pcd" % case)
        _write_compressed(str(p), cols, ["x", "y", "z"], [4, 4, 4], ["F", "F", "F"], [1, 1, 1], n)
        gx, gn = hostlib.load_pcd(p)
        assert np.array_equal(gx, xyz) and gn is None, case


def test_pcd_corrupt_or_missing_is_empty(tmp_path):
    head = b"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary_compressed\n"
    sizes = lambda c, r: np.array([c, r], "<u4").tobytes()
    good = bytes([11]) + np.array([1, 2, 3], "<f4").tobytes()
    cases = {
        "truncated_header": b"\x00\x00",
        "size_mismatch": sizes(len(good), 16) + good,               # 16 bytes announced for one 12-byte point
        "truncated_block": sizes(len(good), 12) + good[:5],
        "reference_before_start": sizes(3, 12) + bytes([(3 << 5), 4, 0]),
        "runs_past_the_end": sizes(len(good) + 2, 12) + good + bytes([0, 7]),
        "short_output": sizes(5, 12) + bytes([3, 1, 2, 3, 4]),
    }
    for name, tail in cases.items():
        p = tmp_path / (name + ".pcd")
        p.write_bytes(head + tail)
        gx, _ = hostlib.load_pcd(p)
        assert len(gx) == 0, name
    p = tmp_path / "ok.pcd"
    p.write_bytes(head + sizes(len(good), 12) + good)
    gx, _ = hostlib.load_pcd(p)
    assert gx.tolist() == [[1, 2, 3]]
    p = tmp_path / "z.pcd"
    p.write_bytes(head.replace(b"binary_compressed", b"binary_zipped") + b"\x00\x00")
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


def test_unsupported_preprocessing_options_are_refused(tmp_path):
    """remove_outliers / sample_above_plane / refine_normals_k are PCL algorithms of their own (candidates_generator.cpp:28-34)
    that the mirror does not have: a cfg that asks for one is refused with a message — before the device is touched —
    instead of yielding grasps on a silently different cloud."""
    import os
    import subprocess
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpd_amd", "host")
    pcd = tmp_path / "three.pcd"
    _write(str(pcd), np.eye(3, dtype=np.float32), np.eye(3, dtype=np.float32), False)
    for key in ("remove_outliers", "sample_above_plane", "refine_normals_k"):
        cfg = tmp_path / (key + ".cfg")
        cfg.write_text("num_samples = 5\nimage_num_channels = 15\n%s = 1\n" % key)
        out = subprocess.run([os.path.join(host, "detect_grasps"), str(cfg), str(pcd)], capture_output=True, text=True, timeout=60)
        assert out.returncode == 255, (key, out.stdout[-300:])
        assert ("ERROR: %s = 1 asks for" % key) in out.stdout and "hip" not in out.stdout.lower().split("asks for")[0], out.stdout[-300:]


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
