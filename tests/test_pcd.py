"""Row N3: PCD v0.7 reader / writer of the C-ABI (host only, no GPU) against independent Python
encoders / decoders of the same format and, when the reference tree is mounted, its own fixtures."""
import os
import struct

import numpy as np
import pytest

REF_PCD = "/root/reference/ndt_omp/data/251370668.pcd"


@pytest.fixture(scope="module")
def ndt(built_lib):
    from toyslam_amd import ndt as m
    return m


def lzf_compress(data):
    """Independent LZF encoder (literal runs + back references) used only to make test inputs."""
    out = bytearray()
    lit = bytearray()
    table = {}
    i, n = 0, len(data)

    def flush():
        for k in range(0, len(lit), 32):
            chunk = lit[k:k + 32]
            out.append(len(chunk) - 1)
            out.extend(chunk)
        lit.clear()

    while i < n:
        key = bytes(data[i:i + 3])
        j = table.get(key, -1)
        if len(key) == 3:
            table[key] = i
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
            out.append(dist & 0xFF)
            i += length
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def header(fields, sizes, types, counts, n, kind):
    return ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\n"
            "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n"
            % (" ".join(fields), " ".join(map(str, sizes)), " ".join(types), " ".join(map(str, counts)), n, n, kind)).encode()


def test_binary_round_trip_and_python_reader(ndt, tmp_path):
    from toyslam_amd import clouds
    rng = np.random.default_rng(3)
    xyz = (rng.standard_normal((5000, 3)) * [30, 30, 3]).astype(np.float32)
    p = str(tmp_path / "a.pcd")
    ndt.pcd_write_xyz(p, xyz)
    got, dense = ndt.pcd_read_xyz(p)
    assert dense and np.array_equal(got, xyz)
    py, names = clouds.read_pcd(p)
    assert names == ["x", "y", "z"] and np.array_equal(py, xyz)
    # written by the Python writer, read by the library
    q = str(tmp_path / "b.pcd")
    clouds.write_pcd_xyz(q, xyz)
    assert open(p, "rb").read() == open(q, "rb").read()
    # 32-byte input records (XYZI-shaped cloud in memory)
    wide = np.zeros((len(xyz), 8), np.float32)
    wide[:, :3] = xyz
    ndt.pcd_write_xyz(q, wide)
    assert np.array_equal(ndt.pcd_read_xyz(q)[0], xyz)
    # empty cloud
    ndt.pcd_write_xyz(q, np.zeros((0, 3), np.float32))
    assert ndt.pcd_read_xyz(q)[0].shape == (0, 3)
    # the chunked reader (32 768 records per chunk; the last record of a chunk takes the scalar path): non-finite values
    # anywhere clear the dense flag, every coordinate survives
    big = (rng.standard_normal((70001, 3)) * [30, 30, 3]).astype(np.float32)
    ndt.pcd_write_xyz(q, big)
    got, dense = ndt.pcd_read_xyz(q)
    assert dense and np.array_equal(got, big)
    for where, col, bad in ((0, 0, np.nan), (32767, 2, np.inf), (32768, 1, -np.inf), (40000, 2, np.nan), (70000, 0, np.nan)):
        c = big.copy()
        c[where, col] = bad
        ndt.pcd_write_xyz(q, c)
        got, dense = ndt.pcd_read_xyz(q)
        assert not dense and np.array_equal(got, c, equal_nan=True), where


def test_extra_fields_orders_and_types(ndt, tmp_path):
    rng = np.random.default_rng(4)
    n = 777
    xyz = rng.standard_normal((n, 3)).astype(np.float32)
    inten = rng.uniform(0, 255, n).astype(np.float32)
    ring = rng.integers(0, 64, n).astype(np.uint16)
    t64 = rng.standard_normal(n)
    # fields out of order, mixed sizes, a COUNT 2 field in front of x
    rec = np.zeros(n, dtype=[("pad", "<f4", (2,)), ("z", "<f4"), ("ring", "<u2"), ("x", "<f4"), ("t", "<f8"),
                             ("intensity", "<f4"), ("y", "<f4")])
    rec["z"], rec["x"], rec["y"] = xyz[:, 2], xyz[:, 0], xyz[:, 1]
    rec["ring"], rec["t"], rec["intensity"] = ring, t64, inten
    p = str(tmp_path / "m.pcd")
    with open(p, "wb") as f:
        f.write(header(["pad", "z", "ring", "x", "t", "intensity", "y"], [4, 4, 2, 4, 8, 4, 4],
                       ["F", "F", "U", "F", "F", "F", "F"], [2, 1, 1, 1, 1, 1, 1], n, "binary"))
        f.write(rec.tobytes())
    got, dense = ndt.pcd_read_xyz(p)
    assert dense and np.array_equal(got, xyz)
    # f64 coordinates are narrowed like PCL's field mapping does
    rec64 = np.zeros(n, dtype=[("x", "<f8"), ("y", "<f8"), ("z", "<f8")])
    rec64["x"], rec64["y"], rec64["z"] = t64, t64 * 2, t64 * 3
    with open(p, "wb") as f:
        f.write(header(["x", "y", "z"], [8, 8, 8], ["F", "F", "F"], [1, 1, 1], n, "binary"))
        f.write(rec64.tobytes())
    got, _ = ndt.pcd_read_xyz(p)
    assert np.array_equal(got, np.stack([t64, t64 * 2, t64 * 3], 1).astype(np.float32))


def test_ascii(ndt, tmp_path):
    rng = np.random.default_rng(5)
    xyz = (rng.standard_normal((300, 3)) * 10).astype(np.float32)
    xyz[7, 1] = np.nan
    p = str(tmp_path / "t.pcd")
    ndt.pcd_write_xyz(p, xyz, binary=False)
    got, dense = ndt.pcd_read_xyz(p)
    assert not dense and np.isnan(got[7, 1])
    ok = np.isfinite(xyz)
    assert np.allclose(got[ok], xyz[ok], rtol=1e-7, atol=0)  # PCDWriter's 8 significant digits
    # hand-written ascii with an intensity column and blank / comment lines
    with open(p, "wb") as f:
        f.write(header(["x", "y", "z", "intensity"], [4, 4, 4, 4], ["F"] * 4, [1] * 4, 3, "ascii"))
        f.write(b"1 2 3 0.5\n\n-1.5e1 2.25 nan 7\n0.1 0.2 0.3 9\n")
    got, dense = ndt.pcd_read_xyz(p)
    assert not dense
    assert np.array_equal(got[[0, 2]], np.array([[1, 2, 3], [0.1, 0.2, 0.3]], np.float32))
    assert got[1, 0] == -15.0 and got[1, 1] == 2.25 and np.isnan(got[1, 2])


def test_binary_compressed(ndt, tmp_path):
    rng = np.random.default_rng(6)
    n = 600
    xyz = np.round(rng.standard_normal((n, 3)) * 4, 1).astype(np.float32)  # coarse values: the stream has back references
    inten = np.repeat(np.float32(3.0), n)
    soa = xyz[:, 0].tobytes() + xyz[:, 1].tobytes() + xyz[:, 2].tobytes() + inten.tobytes()
    comp = lzf_compress(soa)
    assert len(comp) < len(soa)
    p = str(tmp_path / "c.pcd")
    with open(p, "wb") as f:
        f.write(header(["x", "y", "z", "intensity"], [4] * 4, ["F"] * 4, [1] * 4, n, "binary_compressed"))
        f.write(struct.pack("<II", len(comp), len(soa)))
        f.write(comp)
    got, dense = ndt.pcd_read_xyz(p)
    assert dense and np.array_equal(got, xyz)


def test_errors(ndt, tmp_path):
    from toyslam_amd import NdtError
    with pytest.raises(NdtError):
        ndt.pcd_read_xyz(str(tmp_path / "missing.pcd"))
    p = str(tmp_path / "bad.pcd")
    with open(p, "wb") as f:
        f.write(header(["x", "y", "z"], [4, 4, 4], ["F"] * 3, [1] * 3, 10, "binary"))
        f.write(b"\0" * 50)  # truncated
    with pytest.raises(NdtError):
        ndt.pcd_read_xyz(p)
    with open(p, "wb") as f:
        f.write(header(["a", "b", "c"], [4, 4, 4], ["F"] * 3, [1] * 3, 1, "binary"))
        f.write(b"\0" * 12)
    with pytest.raises(NdtError):
        ndt.pcd_read_xyz(p)
    with open(p, "wb") as f:
        f.write(b"VERSION 0.7\nFIELDS x y z\n")  # no DATA line
    with pytest.raises(NdtError):
        ndt.pcd_read_xyz(p)


@pytest.mark.skipif(not os.path.exists(REF_PCD), reason="reference tree not mounted")
def test_reference_fixture(ndt):
    """The reference's own scan (FIELDS x y z intensity, binary, 69 088 points)."""
    from toyslam_amd import clouds
    got, dense = ndt.pcd_read_xyz(REF_PCD)
    py, names = clouds.read_pcd(REF_PCD)
    assert names[:3] == ["x", "y", "z"] and got.shape == (69088, 3) and dense
    assert np.array_equal(got, py[:, :3])


REF_PCD2 = "/root/reference/ndt_omp/data/251371071.pcd"


@pytest.mark.skipif(not (os.path.exists(REF_PCD) and os.path.exists(REF_PCD2)), reason="reference tree not mounted")
def test_committed_pair_is_the_reference_pair_after_align_cpp_downsample(ndt, pair):
    """tests/golden/pair_0p1.npz (what every parity test registers) is the reference's bundled PCD pair
    after apps/align.cpp's 0.1 m VoxelGrid (align.cpp:36-69: argv[1] = target = 251370668, argv[2] =
    source = 251371071): rebuilt here from the raw files through the library's PCD reader and the
    oracle's VoxelGrid restatement."""
    from oracle import pyoracle as po
    from toyslam_amd import clouds
    t, s = pair
    for path, fixture in ((REF_PCD, t), (REF_PCD2, s)):
        raw, dense = ndt.pcd_read_xyz(path)
        assert dense
        # the generator (oracle/gen_golden.py) used the numpy down-sample: f64 centroid sums
        assert np.array_equal(clouds.voxel_downsample(raw, 0.1), fixture)
        # [PCL]'s own f32 centroid accumulation (the oracle's restatement, what N1 reproduces on the
        # GPU) gives the same voxels in the same order; coordinates agree to the last f32 bit or two
        down, overflow = po.voxel_grid_filter(raw, 0.1)
        assert not overflow and down.shape == fixture.shape
        assert np.abs(down - fixture).max() <= 2e-5


def test_reader_survives_mutated_files_under_sanitizers(tmp_path):
    """tests/pcd_fuzz.cpp: 5000 mutated binary / ascii / compressed files through the reader, built with
    ASan + UBSan (CPU build only): parsed or rejected, never a crash or an out-of-bounds access."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pcd_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(root, "toyslam_amd", "csrc"), os.path.join(root, "tests", "pcd_fuzz.cpp"),
                           os.path.join(root, "toyslam_amd", "csrc", "ndt_pcd.cpp"), "-o", exe])
    out = subprocess.run([exe, "5000", str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "no crash" in out.stdout


def test_mapped_and_read_bodies_parse_alike(ndt, tmp_path):
    """A binary body parsed from a mapping of the file (the default) and through the chunk buffer (NDT_PCD_MMAP=0) gives the
    same records: 12-byte and wider records, more than one 32768-point chunk, a NaN in the last record."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(5)
    xyz = rng.standard_normal((70001, 3)).astype(np.float32)
    xyz[-1, 1] = np.nan
    plain = str(tmp_path / "plain.pcd")
    ndt.pcd_write_xyz(plain, xyz)
    wide = str(tmp_path / "wide.pcd")
    rec = np.zeros(len(xyz), dtype=[("i", "<f4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("t", "<f8")])
    rec["x"], rec["y"], rec["z"], rec["i"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], 7.0
    with open(wide, "wb") as f:
        f.write(("# .PCD v0.7\nVERSION 0.7\nFIELDS intensity x y z stamp\nSIZE 4 4 4 4 8\nTYPE F F F F F\nCOUNT 1 1 1 1 1\n"
                 "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (len(xyz), len(xyz))).encode())
        f.write(rec.tobytes())
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from toyslam_amd import ndt; "
            "a, d = ndt.pcd_read_xyz(sys.argv[1]); sys.stdout.buffer.write(bytes([int(d)]) + np.ascontiguousarray(a).tobytes())" % root)
    for path in (plain, wide):
        outs = [subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, NDT_PCD_MMAP=m), capture_output=True, check=True).stdout
                for m in ("1", "0")]
        assert outs[0] == outs[1] and outs[0][0] == 0  # (not dense: the NaN)
        got = np.frombuffer(outs[0][1:], dtype=np.float32).reshape(len(xyz), -1)[:, :3]
        assert np.array_equal(got, xyz, equal_nan=True)


# ------------------------------------------------------------------ numbered scans of a directory (the mapping node's input)
def test_extract_file_number_follows_the_node(ndt):
    """extract_file_number (ndt_omp_mapping_node.cpp:231-239): std::stoi of what follows the last underscore, -1 otherwise."""
    cases = {"cloud_12": 12, "cloud_0007": 7, "a_b_42": 42, "scan_3extra": 3, "cloud": -1, "cloud_": -1, "cloud_x9": -1,
             "x_-3": -3, "x_+8": 8, "x_ 5": 5, "x_99999999999": -1, "_1": 1}
    for stem, want in cases.items():
        assert ndt.extract_file_number(stem) == want, stem


def test_sequence_order_polling_and_bad_files(ndt, tmp_path):
    """process_new_clouds (:110-136): only *.pcd, only numbers >= loaded + 1, ascending by NUMBER (not by name); later
    polls pick up files that have appeared since; a file that cannot be parsed is reported and skipped."""
    rng = np.random.default_rng(5)
    clouds = {k: rng.uniform(-5, 5, (50 + k, 3)).astype(np.float32) for k in (1, 2, 3, 10, 11)}
    for k in (10, 2, 1):  # written out of order; "cloud_10" sorts before "cloud_2" by name
        ndt.pcd_write_xyz(str(tmp_path / ("cloud_%d.pcd" % k)), clouds[k], binary=(k != 2))
    (tmp_path / "notes_5.txt").write_text("not a scan")
    ndt.pcd_write_xyz(str(tmp_path / "unnumbered.pcd"), clouds[1])
    (tmp_path / "cloud_4.pcd").write_bytes(b"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 9\nHEIGHT 1\nPOINTS 9\nDATA binary\nxx")
    seq = ndt.PcdSequence(str(tmp_path))
    assert seq.next() is None  # nothing queued before the first poll
    assert seq.poll(0) == 4  # 1, 2, 4 (broken), 10
    got = []
    from toyslam_amd import NdtError
    for _ in range(4):
        try:
            xyz, dense, number = seq.next()
            assert dense and np.array_equal(xyz, clouds[number])
            got.append(number)
        except NdtError as e:
            assert "cloud_4.pcd" in str(e)
            got.append("bad")
    assert got == [1, 2, "bad", 10] and seq.next() is None
    # three clouds are loaded: the next poll takes numbers >= 4 -- the node's rule re-lists 4 and 10 (its count of loaded
    # clouds, not the last number, is the threshold) together with the newcomers
    ndt.pcd_write_xyz(str(tmp_path / "cloud_3.pcd"), clouds[3])
    ndt.pcd_write_xyz(str(tmp_path / "cloud_11.pcd"), clouds[11])
    assert seq.poll(3) == 3
    numbers = []
    for _ in range(3):
        try:
            numbers.append(seq.next()[2])
        except NdtError:
            numbers.append("bad")
    assert numbers == ["bad", 10, 11]
    # with contiguous numbering from 1 (what lidar_subscriber_node writes) every file is delivered exactly once
    assert seq.poll(11) == 0 and seq.next() is None
    with pytest.raises(NdtError):
        ndt.PcdSequence(str(tmp_path / "missing")).poll(0)


def test_sequence_read_ahead_keeps_scans_intact(ndt, tmp_path):
    """the two buffers alternate: a scan handed out stays intact while the next file is being read, for files of very
    different sizes (buffers grow) in all three encodings."""
    rng = np.random.default_rng(6)
    sizes = [5, 40000, 17, 90000, 1, 30000, 30000, 64]
    clouds = [rng.normal(0, 30, (n, 3)).astype(np.float32) for n in sizes]
    for k, c in enumerate(clouds, 1):
        ndt.pcd_write_xyz(str(tmp_path / ("cloud_%d.pcd" % k)), c, binary=(k % 3 != 0))
    seq = ndt.PcdSequence(str(tmp_path))
    assert seq.poll(0) == len(sizes)
    for k, c in enumerate(clouds, 1):
        xyz, dense, number = seq.next()
        assert number == k and xyz.shape == c.shape
        if k % 3 != 0:
            assert np.array_equal(xyz, c)
        else:  # ascii: decimal text of the writer's precision
            assert np.allclose(xyz, c, rtol=2e-6, atol=0)
    assert seq.next() is None


def test_pointcloud2_style_records(ndt):
    """ndt_host_repack_fields: what pcl::fromROSMsg does for the rosbag node (ndt_rosbag_mapping_node.cpp:45-50) -- x, y, z
    picked by byte offset out of records of any step, here the unaligned 22-byte Velodyne layout and a reordered one."""
    rng = np.random.default_rng(8)
    n = 1000
    xyz = rng.normal(0, 20, (n, 3)).astype(np.float32)
    velodyne = np.dtype({"names": ["x", "y", "z", "intensity", "ring", "time"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u2", "<f4"],
                         "offsets": [0, 4, 8, 12, 16, 18], "itemsize": 22})
    rec = np.zeros(n, dtype=velodyne)
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rec["ring"] = rng.integers(0, 32, n)
    out, dense = ndt.repack_fields(rec.tobytes(), n, 22)
    assert dense and np.array_equal(out[:, :3], xyz) and np.all(out[:, 3] == 1)
    odd = np.dtype({"names": ["t", "z", "pad", "x", "y"], "formats": ["<f8", "<f4", "u1", "<f4", "<f4"], "offsets": [0, 8, 12, 13, 17], "itemsize": 21})
    rec2 = np.zeros(n, dtype=odd)
    rec2["x"], rec2["y"], rec2["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rec2["x"][7] = np.nan
    out2, dense2 = ndt.repack_fields(rec2.tobytes(), n, 21, off_x=13, off_y=17, off_z=8)
    assert not dense2 and np.array_equal(out2[:, :3], np.c_[rec2["x"], rec2["y"], rec2["z"]], equal_nan=True)
    from toyslam_amd import NdtError
    with pytest.raises(NdtError):
        ndt.repack_fields(rec.tobytes(), n, 22, off_x=20)
