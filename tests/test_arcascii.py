"""ArcASCII fast paths (wdpm_amd/csrc/arcascii.c): the integer "%f" formatter must produce exactly
glibc's printf("%f") text and the decimal parser exactly strtod's double, on random and edge-case
values; a written grid must read back to the values printf's 6 decimals define."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def asc(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("asc") / "libasc.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "wdpm_amd", "csrc", "arcascii.c")])
    lib = C.CDLL(so)
    lib.asc_format_f6.argtypes = [C.c_double, C.c_char_p]
    lib.asc_format_f6.restype = C.c_int
    lib.asc_parse_double.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]
    lib.asc_parse_double.restype = C.c_double
    return lib


def edge_values():
    v = [0.0, -0.0, 1 / 128, 3 / 128, -5 / 128, 0.0000005, 0.00000049999, 0.9999995, 0.99999949, 1e-7, 2.4e-7, 2.5e-7,
         123456789.1234565, 1e15, 9.007199254740992e15, 1e18, 9.2e18, 1e19, 1e300, 5e-324, -1e-9, 0.5, 1.5, 2.5,
         float("inf"), float("-inf"), float("nan"), 491.5992, -99999.0, 0.420810, 97577.54, 2 ** 53 - 1, 2.0 ** 62]
    v += [k / 128 + 1e5 for k in range(1, 64, 2)] + [(2 * k + 1) / 256 for k in range(40)]
    return v


def test_format_f6_matches_printf(asc):
    rng = np.random.default_rng(3)
    vals = edge_values()
    vals += list(rng.uniform(-2000, 2000, 100000))
    vals += list(10.0 ** rng.uniform(-12, 17, 50000) * rng.choice([-1, 1], 50000))
    vals += list(np.round(rng.uniform(0, 3, 50000), 6))                 # typical depths, incl. near-ties
    vals += list((rng.integers(0, 10 ** 9, 50000) + 0.5) / 1e6)          # decimal "ties" that are not binary ties
    buf = C.create_string_buffer(400)
    for x in vals:
        n = asc.asc_format_f6(float(x), buf)
        assert buf.raw[:n].decode() == "%f" % x, repr(x)


def test_parse_double_matches_strtod(asc):
    rng = np.random.default_rng(4)
    texts = ["0", "-0", "+1.5", "1e5", "1E-5", "-99999.0000", "491.5992", "  12.5 ", "\n7", ".5", "5.", "1e", "1e+",
             "1.7976931348623157e308", "4.9e-324", "123456789012345678", "0.000000000000000000001", "1e23", "1e22",
             "9007199254740993", "0x1p3", "nan", "inf", "-inf", "abc", "", "1.2.3", "12abc", "1e400", "00012.50",
             "3.141592653589793238462643383279", "1234567890123456", "123456789012345", "0.1", "0.3", "1e-22", "1e-23"]
    texts += ["%.4f" % v for v in rng.uniform(-1000, 1000, 20000)]
    texts += ["%.6f" % v for v in rng.uniform(0, 5, 20000)]
    texts += ["%.*e" % (int(p), v) for p, v in zip(rng.integers(0, 17, 20000), rng.uniform(-1e6, 1e6, 20000))]
    texts += [repr(float(v)) for v in 10.0 ** rng.uniform(-30, 30, 20000)]
    for t in texts:
        b = t.encode() + b" tail"
        end = C.c_char_p()
        got = asc.asc_parse_double(b, C.byref(end))
        buf = C.create_string_buffer(b)
        want_end = C.c_char_p()
        libc = C.CDLL(None)
        libc.strtod.restype = C.c_double
        libc.strtod.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]
        want = libc.strtod(buf, C.byref(want_end))
        assert (got == want or (got != got and want != want)) and np.signbit(got) == np.signbit(want), t
        # same number of characters consumed
        assert len(end.value or b"") == len(want_end.value or b""), t


class Header(C.Structure):
    _fields_ = [("name", (C.c_char * 32) * 6), ("value", C.c_double * 6)]


def test_threaded_grid_io_equals_single_thread(asc, tmp_path):
    """a raster above the threading threshold is written / read by several host threads: the file
    must be byte-identical to the single-threaded one and read back identically"""
    rng = np.random.default_rng(5)
    R, Cc = 1100, 1000
    a = np.round(rng.uniform(0, 3, (R, Cc)), 7)
    a[rng.random((R, Cc)) < 0.3] = 0.0
    a[5, 7] = -99999.0
    h = Header()
    for i, (nm, v) in enumerate([("NCOLS", Cc), ("NROWS", R), ("XLLCORNER", 1.5), ("YLLCORNER", 2.5),
                                 ("CELLSIZE", 10.0), ("NODATA_VALUE", -99999.0)]):
        h.name[i].value = nm.encode()
        h.value[i] = v
    asc.asc_write_grid.argtypes = [C.c_char_p, C.POINTER(Header), C.c_int, C.c_int, C.c_void_p]
    asc.asc_read_grid.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    files = {}
    for t in ("1", "5"):
        os.environ["WDPM_IO_THREADS"] = t
        path = str(tmp_path / f"g{t}.asc").encode()
        assert asc.asc_write_grid(path, C.byref(h), R, Cc, a.ctypes.data) == 0
        files[t] = open(path, "rb").read()
        back = np.zeros_like(a)
        assert asc.asc_read_grid(path, R, Cc, back.ctypes.data) == 0
        want = np.vectorize(lambda v: float("%f" % v))(a[::97])
        assert np.array_equal(back[::97], want)
    os.environ.pop("WDPM_IO_THREADS")
    assert files["1"] == files["5"]
    # big files are mapped, not read - except when their size is a whole number of pages (no NUL behind
    # the text then): pad one to such a size and read it again
    padded = files["1"] + b"\n" * (-len(files["1"]) % 4096)
    assert len(padded) % 4096 == 0 and len(files["1"]) % 4096 != 0
    ppath = str(tmp_path / "padded.asc").encode()
    open(ppath, "wb").write(padded)
    again = np.zeros_like(a)
    assert asc.asc_read_grid(ppath, R, Cc, again.ctypes.data) == 0
    assert np.array_equal(again, back)
    first = files["1"].split(b"\n")
    assert first[0] == b"NCOLS 1000" and first[5] == b"NODATA_VALUE  -99999.000000"
    assert first[6].startswith(("%f " % a[0, 0]).encode())
