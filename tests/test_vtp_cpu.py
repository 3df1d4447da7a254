"""Feature files (VTK XML PolyData, the reference's `<dir>_00.vtk`) read without VTK: the C++ reader
include/lrr_amd/vtp_features.h against the committed fixtures and against files the independent Python writer
produces in every encoding; the conversion to features against cdcl/io/vtkPolyDataToFeaturesICPFilter.cxx:129-237
restated in numpy."""
import os

import numpy as np
import pytest

from lrr_amd import synth, vtp
from tests.common import GOLDEN_DIR

EXPECTED = np.load(os.path.join(GOLDEN_DIR, "features_vtp_expected.npz"))
FIXTURES = ["features_ascii", "features_binary", "features_appended_zlib", "features_appended_raw64"]


def _same(got, want):
    for k in ("points", "errorProjectors", "strengths", "shapes", "normals", "binormals"):
        w = want[k]
        assert got[k] is not None, k
        assert np.array_equal(got[k].reshape(w.shape), w), k


@pytest.mark.parametrize("name", FIXTURES)
def test_committed_fixture_reads_back(name):
    _same(vtp.read_vtp_arrays(os.path.join(GOLDEN_DIR, name + ".vtp")), EXPECTED)


@pytest.mark.parametrize("fmt", vtp.FORMATS)
@pytest.mark.parametrize("compress", [False, True])
@pytest.mark.parametrize("header_type", ["UInt32", "UInt64"])
def test_every_encoding_round_trips(tmp_path, fmt, compress, header_type):
    pair = synth.make_pair(3000, 10, 77, kind="affine")  # 3000 x 9 floats: several zlib blocks
    arrays = vtp.disk_arrays(pair.fixed)
    path = tmp_path / "f.vtp"
    vtp.write_vtp(path, arrays, fmt=fmt, compress=compress, header_type=header_type)
    want = {k: np.asarray(v, dtype=np.float32 if k != "shapes" else np.uint8).astype(np.float64) for k, v in arrays.items()}
    _same(vtp.read_vtp_arrays(path), want)
    if fmt in ("binary", "appended-base64") and not compress:
        vtp.write_vtp(path, arrays, fmt=fmt, header_type=header_type, joint_header=True)  # length + data in one base64 unit
        _same(vtp.read_vtp_arrays(path), want)


def test_features_follow_the_reference_filter():
    """directions_ per shape: CORNER {normal x binormal, binormal, normal}, TUBE {binormal, normal}, SHEET {normal};
    strength float; projector row-major"""
    fs, ndirs, dirs9 = vtp.read_features(os.path.join(GOLDEN_DIR, "features_binary.vtp"), return_all_directions=True)
    a = EXPECTED
    assert np.array_equal(fs.loc, a["points"]) and np.array_equal(fs.proj, a["errorProjectors"])
    assert np.array_equal(fs.strength, a["strengths"]) and np.array_equal(fs.shape, a["shapes"].astype(np.uint8))
    n, b = a["normals"], a["binormals"]
    corner, tube, sheet = fs.shape == 0, fs.shape == 1, fs.shape == 2
    assert corner.any() and tube.any() and sheet.any()
    assert np.array_equal(ndirs[corner], np.full(corner.sum(), 3)) and np.array_equal(ndirs[tube], np.full(tube.sum(), 2))
    assert np.array_equal(ndirs[sheet], np.full(sheet.sum(), 1))
    assert np.array_equal(dirs9[sheet, 0:3], n[sheet]) and not dirs9[sheet, 3:].any()
    assert np.array_equal(dirs9[tube, 0:3], b[tube]) and np.array_equal(dirs9[tube, 3:6], n[tube])
    assert np.allclose(dirs9[corner, 0:3], np.cross(n[corner], b[corner]), rtol=0, atol=1e-15)
    assert np.array_equal(dirs9[corner, 3:6], b[corner]) and np.array_equal(dirs9[corner, 6:9], n[corner])
    # the tube tangent the angle diagnostics use (rrl .txx:351: directions_[0] x directions_[1]) survives the file
    src = synth.make_pair(400, 400, 424242, kind="affine").fixed
    t_file = np.cross(fs.dirs[tube, 0:3], fs.dirs[tube, 3:6])
    t_src = np.cross(src.dirs[tube, 0:3], src.dirs[tube, 3:6])
    assert np.allclose(t_file, t_src, atol=1e-6)


def test_optional_arrays_and_errors(tmp_path):
    pair = synth.make_pair(50, 10, 5)
    arrays = vtp.disk_arrays(pair.fixed)
    del arrays["strengths"], arrays["shapes"]
    p = tmp_path / "nostrength.vtp"
    vtp.write_vtp(p, arrays, fmt="ascii")
    got = vtp.read_vtp_arrays(p)
    assert got["strengths"] is None and got["shapes"] is None
    fs = vtp.read_features(p)  # strengths / shapes are optional in the filter (:181-186): 0 and CORNER
    assert not fs.strength.any() and not fs.shape.any()
    del arrays["errorProjectors"]
    vtp.write_vtp(p, arrays, fmt="ascii")
    with pytest.raises(RuntimeError, match="errorProjectors"):
        vtp.read_features(p)
    with pytest.raises(RuntimeError, match="cannot open"):
        vtp.read_vtp_arrays(tmp_path / "missing.vtp")
    bad = tmp_path / "bad.vtp"
    bad.write_text('<?xml version="1.0"?><VTKFile type="UnstructuredGrid"></VTKFile>')
    with pytest.raises(RuntimeError, match="PolyData"):
        vtp.read_vtp_arrays(bad)
    # truncated appended section
    arrays = vtp.disk_arrays(pair.fixed)
    vtp.write_vtp(p, arrays, fmt="appended-raw")
    data = p.read_bytes()
    cut = tmp_path / "cut.vtp"
    cut.write_bytes(data[: len(data) - 2000])
    with pytest.raises(RuntimeError, match="vtp:"):
        vtp.read_vtp_arrays(cut)


def test_empty_file(tmp_path):
    e = dict(points=np.zeros((0, 3)), errorProjectors=np.zeros((0, 9)), strengths=np.zeros(0), shapes=np.zeros(0, np.uint8),
             normals=np.zeros((0, 3)), binormals=np.zeros((0, 3)))
    for fmt in vtp.FORMATS:
        p = tmp_path / "empty.vtp"
        vtp.write_vtp(p, e, fmt=fmt)
        assert vtp.read_features(p).n == 0


@pytest.fixture(scope="module")
def vtp_probe(tmp_path_factory):
    """tests/cpp/vtp_probe.cpp built with AddressSanitizer + UBSan (host build: g++, no GPU code)"""
    import subprocess
    out = tmp_path_factory.mktemp("probe") / "vtp_probe"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "vtp_probe.cpp"), "-lz", "-o", str(out)])
    return str(out)


def _probe(exe, path):
    import subprocess
    r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=4096"))
    assert r.returncode == 0, f"sanitizer report or crash on {path}:\n{r.stderr[-2000:]}"
    return r.stdout.strip()


def _patch_word(data: bytes, pos: int, value: int, size: int) -> bytes:
    return data[:pos] + int(value).to_bytes(size, "little") + data[pos + size:]


@pytest.mark.parametrize("header_type", ["UInt32", "UInt64"])
@pytest.mark.parametrize("compress", [False, True])
def test_corrupt_size_words_end_in_an_error(tmp_path, vtp_probe, header_type, compress):
    """ADVICE (round 2): every size word of a binary block is attacker-controlled.  Raw appended blocks with their size words
    replaced by huge values (sums that would wrap 64 bits), by values just past the end of the file, and truncated files:
    the reader must throw its runtime_error — run under ASan/UBSan so that an out-of-bounds read would be seen"""
    pair = synth.make_pair(600, 10, 91, kind="affine")
    arrays = vtp.disk_arrays(pair.fixed)
    p = tmp_path / "f.vtp"
    vtp.write_vtp(p, arrays, fmt="appended-raw", compress=compress, header_type=header_type)
    data = p.read_bytes()
    assert _probe(vtp_probe, p) == "ok 600"
    h = 8 if header_type == "UInt64" else 4
    start = data.index(b"<AppendedData")
    us = data.index(b"_", data.index(b">", start)) + 1
    big = [2**(8 * h) - 1, 2**(8 * h) - h, 2**(8 * h - 1), len(data), len(data) - us, 2**31]
    words = 1 if not compress else 5  # nbytes | nblocks, block size, last size, first compressed sizes
    for w in range(words):
        for v in big:
            q = tmp_path / f"c_{w}_{v}.vtp"
            q.write_bytes(_patch_word(data, us + w * h, v, h))
            got = _probe(vtp_probe, q)
            # (word 1 of a compressed block is the writer's nominal block size: unused when the array is one short block)
            assert got.startswith("error vtp:") or (compress and w == 1 and got == "ok 600"), (w, v, got)
    for cut in (us + 1, us + h, us + 3 * h + 1, len(data) - 300, len(data) // 2):
        q = tmp_path / f"t_{cut}.vtp"
        q.write_bytes(data[:cut])
        assert _probe(vtp_probe, q).startswith("error vtp:"), cut


def test_corrupt_markup_and_base64_end_in_an_error(tmp_path, vtp_probe):
    pair = synth.make_pair(300, 10, 92, kind="affine")
    arrays = vtp.disk_arrays(pair.fixed)
    p = tmp_path / "f.vtp"
    for fmt in ("binary", "appended-base64"):
        for compress in (False, True):
            vtp.write_vtp(p, arrays, fmt=fmt, compress=compress)
            data = p.read_bytes()
            assert _probe(vtp_probe, p) == "ok 300"
            for cut in (40, len(data) // 3, len(data) // 2, len(data) - 200):
                q = tmp_path / "cut.vtp"
                q.write_bytes(data[:cut])
                assert _probe(vtp_probe, q).startswith("error vtp:"), (fmt, compress, cut)
    # tags without their closing '>'
    for text in ('<VTKFile type="PolyData"', '<VTKFile type="PolyData"><PolyData><Piece NumberOfPoints="3"',
                 '<VTKFile type="PolyData"><AppendedData encoding="raw"',
                 '<VTKFile type="PolyData"><PolyData><Piece NumberOfPoints="99999999999999999999"><Points><DataArray type="Float32" '
                 'NumberOfComponents="3" format="ascii">1 2 3</DataArray></Points></Piece></PolyData></VTKFile>'):
        q = tmp_path / "m.vtp"
        q.write_text(text)
        assert _probe(vtp_probe, q).startswith("error vtp:"), text
