"""gz file calls (after the reference's test/test_gz.c: nx_gzwrite -> gzip reads it, gzip writes ->
nx_gzread), nx_inflateCopy / nx_inflateResetKeep, and the 64-bit combine aliases."""
import ctypes as C
import gzip
import os
import zlib

import pytest

import zstream as Z
from datagen import make_block


@pytest.fixture(params=["model", pytest.param("gpu", marks=pytest.mark.gpu)])
def L(request):
    return Z.load(request.param)


@pytest.mark.parametrize("mode", [b"w", b"w1", b"w9"])
def test_gzwrite_is_read_by_gzip(L, tmp_path, mode):
    data = make_block("alice", 1 << 20, 3) + make_block("random", 70000, 4) + make_block("zeros", 5000, 5)
    path = str(tmp_path / "a.gz").encode()
    f = L.nx_gzopen(path, mode)
    assert f
    off = 0
    for step in (1, 7, 4096, 65536, 300000, len(data)):       # uneven writes
        k = min(step, len(data) - off)
        if k:
            assert L.nx_gzwrite(f, data[off:off + k], k) == k
            off += k
    assert off == len(data)
    assert L.nx_gzclose(f) == Z.Z_OK
    assert gzip.open(path.decode(), "rb").read() == data


def test_gzread_of_a_gzip_made_file(L, tmp_path):
    data = make_block("lz", 1 << 20, 9) + make_block("text33", 12345, 10)
    path = str(tmp_path / "b.gz")
    with gzip.open(path, "wb", compresslevel=6) as g:
        g.write(data)
    fd = os.open(path, os.O_RDONLY)
    f = L.nx_gzdopen(fd, b"r")
    assert f
    got = bytearray()
    buf = C.create_string_buffer(100000)
    for want in (1, 10, 1000, 100000, 100000, 100000):
        n = L.nx_gzread(f, buf, want)
        got += buf.raw[:n]
    while True:
        n = L.nx_gzread(f, buf, 100000)
        if n <= 0:
            break
        got += buf.raw[:n]
    assert L.nx_gzclose(f) == Z.Z_OK
    assert bytes(got) == data


def test_gz_argument_errors(L, tmp_path):
    # a strategy the engine does not do is refused like in the reference (lib/nx_deflate.c:626-629)
    assert not L.nx_gzopen(str(tmp_path / "f.gz").encode(), b"wf")
    assert not L.nx_gzopen(str(tmp_path / "missing" / "x.gz").encode(), b"r")
    assert L.nx_gzclose(None) == Z.Z_STREAM_ERROR
    assert L.nx_gzwrite(None, b"x", 1) == 0
    assert L.nx_gzread(None, C.create_string_buffer(4), 4) == 0


def test_inflate_copy_continues_independently(L):
    data = make_block("alice", 300000, 21)
    comp = zlib.compress(data, 6)
    half = len(comp) // 2
    s = Z.ZStream()
    assert L.nx_inflateInit2_(C.byref(s), 15, zlib.ZLIB_VERSION.encode(), C.sizeof(Z.ZStream)) == Z.Z_OK
    src = C.create_string_buffer(comp, len(comp))
    out1 = C.create_string_buffer(len(data) + 64)
    s.next_in = C.cast(src, C.c_void_p).value; s.avail_in = half
    s.next_out = C.cast(out1, C.c_void_p).value; s.avail_out = len(data) + 64
    assert L.nx_inflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_OK
    produced = s.total_out
    # copy in the middle of the stream, then finish both from the same remaining input
    c = Z.ZStream()
    assert L.nx_inflateCopy(C.byref(c), C.byref(s)) == Z.Z_OK
    out2 = C.create_string_buffer(len(data) + 64)
    C.memmove(out2, out1, produced)
    for strm, out in ((s, out1), (c, out2)):
        strm.next_in = C.cast(src, C.c_void_p).value + half - strm.avail_in if False else C.cast(src, C.c_void_p).value + (half - strm.avail_in)
        strm.avail_in = strm.avail_in + (len(comp) - half)
        strm.next_out = C.cast(out, C.c_void_p).value + produced
        strm.avail_out = len(data) + 64 - produced
        rc = L.nx_inflate(C.byref(strm), Z.Z_FINISH)
        assert rc == Z.Z_STREAM_END, rc
        assert strm.total_out == len(data)
        assert out.raw[:len(data)] == data
        assert L.nx_inflateEnd(C.byref(strm)) == Z.Z_OK
    assert L.nx_inflateCopy(None, None) == Z.Z_STREAM_ERROR


def test_inflate_reset_keep_reuses_the_stream(L):
    s = Z.ZStream()
    assert L.nx_inflateInit2_(C.byref(s), 15, zlib.ZLIB_VERSION.encode(), C.sizeof(Z.ZStream)) == Z.Z_OK
    for seed in (1, 2):
        data = make_block("lz", 50000, seed)
        comp = zlib.compress(data, 1)
        src = C.create_string_buffer(comp, len(comp)); out = C.create_string_buffer(len(data) + 16)
        s.next_in = C.cast(src, C.c_void_p).value; s.avail_in = len(comp)
        s.next_out = C.cast(out, C.c_void_p).value; s.avail_out = len(data) + 16
        assert L.nx_inflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
        assert out.raw[:len(data)] == data
        assert L.nx_inflateResetKeep(C.byref(s)) == Z.Z_OK
        assert s.total_out == 0
    assert L.nx_inflateEnd(C.byref(s)) == Z.Z_OK


def test_combine64_aliases(L):
    a, b = make_block("random", 1000, 1), make_block("random", 3333, 2)
    assert L.nx_crc32_combine64(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    assert L.nx_adler32_combine64(zlib.adler32(a), zlib.adler32(b), len(b)) == zlib.adler32(a + b)
