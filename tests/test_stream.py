"""The nx_* stream layer (power-gzip_amd/csrc/nxz_stream.cpp).  CPU runs use the host sources over
the CPU engine model; the same cases run over the HIP engine with -m gpu.  Cases follow the
reference's own tests (test/test_deflate.c, test_inflate.c, test_zeroinput.c, test_buf_error.c,
test_inflatesyncpoint.c, deflate/compress.c) with seeded data."""
import ctypes as C
import json
import os
import zlib

import pytest

import zstream as Z
from datagen import make_block

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(params=["model", pytest.param("gpu", marks=pytest.mark.gpu)])
def L(request):
    return Z.load(request.param)


def test_empty_streams_golden_bytes(L):
    # SURVEY Appendix B, derived from lib/nx_deflate.c:1418-1459,1594-1604,220-243,428-468
    out, rcs, _ = Z.deflate_all(L, b"", wbits=15)
    assert out.hex() == "7801" + "010000ffff" + "00000001" and rcs == [Z.Z_STREAM_END]
    out, _, _ = Z.deflate_all(L, b"", wbits=31)
    assert out.hex() == "1f8b0800000000000403" + "010000ffff" + "00000000" + "00000000"
    out, _, _ = Z.deflate_all(L, b"", wbits=-15)
    assert out.hex() == "010000ffff"
    assert zlib.decompress(bytes.fromhex("7801010000ffff00000001")) == b""


@pytest.mark.parametrize("level,hdr", [(-1, "7801"), (1, "7801"), (5, "785e"), (6, "789c"), (9, "78da")])
def test_zlib_header_flevel(L, level, hdr):
    out, _, _ = Z.deflate_all(L, b"hello hello hello hello", level=level)
    assert out[:2].hex() == hdr
    assert zlib.decompress(out) == b"hello hello hello hello"


SIZES = [1, 14, 100, 4096, 8192, 8193, 65535, 65536, 65537, 200000]


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("wbits,strategy", [(15, Z.Z_DEFAULT_STRATEGY), (31, Z.Z_FIXED), (-15, Z.Z_DEFAULT_STRATEGY)])
def test_deflate_roundtrip_through_zlib(L, n, wbits, strategy):
    data = make_block("lz" if n % 2 else "alice", n, seed=n)
    out, _, adler = Z.deflate_all(L, data, wbits=wbits, strategy=strategy)
    assert zlib.decompress(out, wbits) == data
    if wbits == 15:
        assert adler == zlib.adler32(data)


@pytest.mark.parametrize("step_in,step_out", [(1, 1), (7, 3), (100, 1), (5000, 64), (70000, 100000)])
def test_deflate_small_steps(L, step_in, step_out):
    # test/test_utils.c:232-307: feed `step` bytes in/out per call, then an avail_out=1 finish loop
    data = make_block("alice", 20000 if step_in < 100 else 150000, seed=step_in)
    out, rcs, _ = Z.deflate_all(L, data, step_in=step_in, step_out=step_out)
    assert zlib.decompress(out) == data


@pytest.mark.parametrize("level", [1, 5, 6, 9])
def test_deflate_levels_history_carry(L, level):
    data = make_block("alice", 300000, seed=3)
    out, _, _ = Z.deflate_all(L, data, level=level, step_in=50000)
    assert zlib.decompress(out) == data


@pytest.mark.parametrize("flush", [Z.Z_SYNC_FLUSH, Z.Z_FULL_FLUSH, Z.Z_PARTIAL_FLUSH])
def test_deflate_flush_modes(L, flush):
    data = make_block("lz", 100000, seed=11)
    out, _, _ = Z.deflate_all(L, data, step_in=9000, flush=flush, wbits=-15)
    assert zlib.decompress(out, -15) == data
    if flush != Z.Z_PARTIAL_FLUSH:
        assert out.count(b"\x00\x00\xff\xff") >= 11


def test_incompressible_goes_to_stored_blocks(L):
    data = make_block("random", 150000, seed=5)
    out, _, _ = Z.deflate_all(L, data, wbits=31)
    assert zlib.decompress(out, 31) == data
    assert len(out) < len(data) + 5 * (len(data) // 60000 + 4) + 64


def test_zero_input_return_codes(L):
    # test/test_zeroinput.c:37-58 and :61-88
    for flush in range(Z.Z_NO_FLUSH, Z.Z_FINISH + 1):
        st = Z.ZStream()
        assert L.nx_deflateInit2_(C.byref(st), -1, 8, 15, 8, 0, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_OK
        dst = C.create_string_buffer(8192)
        src = C.create_string_buffer(make_block("text33", 2048, 1), 2048)
        st.next_in = C.addressof(src); st.next_out = C.addressof(dst); st.avail_in = 0; st.avail_out = 8192
        assert L.nx_deflate(C.byref(st), flush) == (Z.Z_STREAM_END if flush == Z.Z_FINISH else Z.Z_OK) or flush == Z.Z_NO_FLUSH
        if flush != Z.Z_FINISH:
            assert L.nx_deflate(C.byref(st), Z.Z_FINISH) == Z.Z_STREAM_END
        assert L.nx_deflateEnd(C.byref(st)) == Z.Z_OK
    for flush in range(Z.Z_NO_FLUSH, Z.Z_FINISH + 1):
        st = Z.ZStream()
        assert L.nx_deflateInit2_(C.byref(st), -1, 8, 15, 8, 0, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_OK
        dst = C.create_string_buffer(8192)
        src = C.create_string_buffer(make_block("text33", 2048, 1), 2048)
        st.next_in = C.addressof(src); st.next_out = C.addressof(dst); st.avail_in = 1024; st.avail_out = 8192
        assert L.nx_deflate(C.byref(st), Z.Z_NO_FLUSH) == Z.Z_OK
        st.avail_in = 0
        exp = Z.Z_BUF_ERROR if flush == Z.Z_NO_FLUSH else Z.Z_STREAM_END if flush == Z.Z_FINISH else Z.Z_OK
        assert L.nx_deflate(C.byref(st), flush) == exp, flush
        if flush != Z.Z_FINISH:
            assert L.nx_deflate(C.byref(st), Z.Z_FINISH) == Z.Z_STREAM_END
        assert zlib.decompress(dst.raw[:st.total_out]) == src.raw[:1024]
        assert L.nx_deflateEnd(C.byref(st)) == Z.Z_OK


def test_deflate_argument_errors(L):
    st = Z.ZStream()
    assert L.nx_deflateInit2_(None, 6, 8, 15, 8, 0, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_STREAM_ERROR
    for wb in (8, 14, 16, 30, -14, 0, 47):
        assert L.nx_deflateInit2_(C.byref(st), 6, 8, wb, 8, 0, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_STREAM_ERROR
    assert L.nx_deflateInit2_(C.byref(st), 6, 8, 15, 8, 1, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_STREAM_ERROR   # Z_FILTERED
    assert L.nx_deflateInit2_(C.byref(st), 10, 8, 15, 8, 0, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_STREAM_ERROR
    assert L.nx_deflateInit2_(C.byref(st), 6, 8, 15, 8, 0, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_OK
    dst = C.create_string_buffer(100)
    st.next_out = C.addressof(dst)
    st.avail_out = 0
    assert L.nx_deflate(C.byref(st), Z.Z_NO_FLUSH) == Z.Z_BUF_ERROR
    st.avail_out = 100
    assert L.nx_deflate(C.byref(st), 6) == Z.Z_STREAM_ERROR
    assert L.nx_deflate(C.byref(st), Z.Z_FINISH) == Z.Z_STREAM_END
    assert L.nx_deflate(C.byref(st), Z.Z_NO_FLUSH) == Z.Z_STREAM_ERROR        # after BFINAL only Z_FINISH is legal
    assert L.nx_deflateEnd(C.byref(st)) == Z.Z_OK
    assert L.nx_deflateEnd(C.byref(st)) == Z.Z_STREAM_ERROR


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("wbits", [15, 31, -15])
def test_inflate_of_zlib_streams(L, level, wbits):
    data = make_block("alice", 180000, seed=level) + make_block("random", 3000, 2) + make_block("zeros", 70000, 0)
    co = zlib.compressobj(level, zlib.DEFLATED, wbits)
    comp = co.compress(data) + co.flush()
    out, rc, tin, adler = Z.inflate_all(L, comp, wbits=wbits if wbits < 0 else 47)
    assert rc == Z.Z_STREAM_END and out == data and tin == len(comp)
    if wbits == 15:
        assert adler == zlib.adler32(data)
    if wbits == 31:
        assert adler == zlib.crc32(data)


@pytest.mark.parametrize("n", [1, 2, 5, 17, 64, 100])
def test_inflate_every_step_small(L, n):
    # test/inflate/random_buffer.c:47-63: lengths 1..100 x every step
    data = make_block("text33", n, seed=n)
    comp = zlib.compress(data, 6)
    for step in range(1, len(comp) + 1):
        out, rc, tin, _ = Z.inflate_all(L, comp, step_in=step, step_out=step)
        assert rc == Z.Z_STREAM_END and out == data, (n, step)


@pytest.mark.parametrize("flush", [Z.Z_NO_FLUSH, Z.Z_PARTIAL_FLUSH])
def test_inflate_steps_large(L, flush):
    data = make_block("lz", 400000, seed=8)
    comp = zlib.compress(data, 6)
    for step_in, step_out in [(1000, 1000), (7, 100000), (100000, 13), (len(comp), 1 << 20)]:
        out, rc, _, _ = Z.inflate_all(L, comp, step_in=step_in, step_out=step_out, flush=flush)
        assert rc == Z.Z_STREAM_END and out == data


def test_inflate_golden_scp_stream(L, golden_dir):
    # test/test_buf_error.c:107-229: 611 B -> 603 B, then 92 B -> 117 B, stream not ended
    g = json.load(open(os.path.join(golden_dir, "zlib_stream_buf_error.json")))
    compr, compr2 = bytes.fromhex(g["compr"]), bytes.fromhex(g["compr2"])
    st = Z.ZStream()
    assert L.nx_inflateInit2_(C.byref(st), 15, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_OK
    src = C.create_string_buffer(compr + compr2, len(compr) + len(compr2))
    dst = C.create_string_buffer(4096)
    st.next_in = C.addressof(src); st.avail_in = len(compr); st.next_out = C.addressof(dst); st.avail_out = 4096
    rc = L.nx_inflate(C.byref(st), Z.Z_NO_FLUSH)
    while rc == Z.Z_OK and st.total_out < 603:
        rc = L.nx_inflate(C.byref(st), Z.Z_NO_FLUSH)
    assert st.total_out == 603
    assert L.nx_inflate(C.byref(st), Z.Z_NO_FLUSH) == Z.Z_BUF_ERROR      # no progress possible (bug #74 in the reference)
    st.avail_in += len(compr2)
    rc = L.nx_inflate(C.byref(st), Z.Z_NO_FLUSH)
    while rc == Z.Z_OK and st.total_out < 720:
        rc = L.nx_inflate(C.byref(st), Z.Z_NO_FLUSH)
    assert st.total_out == 720
    ref = zlib.decompressobj()
    assert dst.raw[:720] == ref.decompress(compr) + ref.decompress(compr2)
    L.nx_inflateEnd(C.byref(st))


def test_inflate_errors(L):
    data = make_block("alice", 5000, 1)
    comp = bytearray(zlib.compress(data))
    bad = bytes(comp[:-1]) + bytes([comp[-1] ^ 0xff])
    out, rc, _, _ = Z.inflate_all(L, bad)
    assert rc == Z.Z_DATA_ERROR                   # deliberate fix of SURVEY quirk Q9
    out, rc, _, _ = Z.inflate_all(L, b"\x12\x34" + bytes(comp[2:]))
    assert rc == Z.Z_DATA_ERROR
    st = Z.ZStream()
    assert L.nx_inflateInit2_(C.byref(st), 47, b"2.0", C.sizeof(Z.ZStream)) == Z.Z_VERSION_ERROR
    assert L.nx_inflateInit2_(C.byref(st), 16, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_STREAM_ERROR
    assert L.nx_inflateInit2_(C.byref(st), 47, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_OK
    assert L.nx_inflate(C.byref(st), Z.Z_BLOCK) == Z.Z_STREAM_ERROR
    L.nx_inflateEnd(C.byref(st))


def test_one_shot_compress_uncompress(L):
    # test/deflate/compress.c:87-121 (sizes reduced), incl. zeros
    for data in (make_block("alice", 5 * 1024, 1), make_block("lz", 300000, 2), bytes(200000)):
        bound = L.nx_compressBound(len(data))
        dst = C.create_string_buffer(bound)
        dl = C.c_ulong(bound)
        assert L.nx_compress(dst, C.byref(dl), data, len(data)) == Z.Z_OK
        assert zlib.decompress(dst.raw[:dl.value]) == data
        back = C.create_string_buffer(len(data) + 16)
        bl = C.c_ulong(len(data) + 16)
        assert L.nx_uncompress(back, C.byref(bl), dst.raw[:dl.value], dl.value) == Z.Z_OK
        assert back.raw[:bl.value] == data
        z = zlib.compress(data, 6)
        bl = C.c_ulong(len(data) + 16)
        assert L.nx_uncompress(back, C.byref(bl), z, len(z)) == Z.Z_OK and back.raw[:bl.value] == data
        bl = C.c_ulong(len(data) // 2)
        assert L.nx_uncompress(back, C.byref(bl), z, len(z)) == Z.Z_BUF_ERROR


def test_dictionary_roundtrip(L):
    dic = make_block("alice", 40000, 7)
    data = dic[1000:9000] + make_block("alice", 3000, 9)
    out, _, _ = Z.deflate_all(L, data, dictionary=dic)
    assert out[1] & 0x20                                       # FDICT
    assert int.from_bytes(out[2:6], "big") == zlib.adler32(dic)
    d = zlib.decompressobj(zdict=dic)
    assert d.decompress(out) == data
    back, rc, _, _ = Z.inflate_all(L, out, dictionary=dic)
    assert rc == Z.Z_STREAM_END and back == data
    co = zlib.compressobj(6, zlib.DEFLATED, -15, zdict=dic)
    raw = co.compress(data) + co.flush()
    back, rc, _, _ = Z.inflate_all(L, raw, wbits=-15, dictionary=dic)
    assert rc == Z.Z_STREAM_END and back == data


def test_inflate_sync_point(L):
    # test/test_inflatesyncpoint.c: chunks ended with Z_SYNC_FLUSH; the sync point is 4 bytes before
    # the end of each flushed chunk (after the empty stored block header)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    chunks = [co.compress(make_block("text33", 256, i)) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(5)]
    st = Z.ZStream()
    assert L.nx_inflateInit2_(C.byref(st), -15, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_OK
    dst = C.create_string_buffer(8192)
    st.next_out = C.addressof(dst); st.avail_out = 8192
    for ch in chunks:
        head = C.create_string_buffer(ch[:-4], len(ch) - 4)
        st.next_in = C.addressof(head); st.avail_in = len(ch) - 4
        rc = L.nx_inflate(C.byref(st), Z.Z_SYNC_FLUSH)
        assert rc in (Z.Z_OK, Z.Z_BUF_ERROR)
        assert L.nx_inflateSyncPoint(C.byref(st)) == 1
        tail = C.create_string_buffer(ch[-4:], 4)
        st.next_in = C.addressof(tail); st.avail_in = 4
        L.nx_inflate(C.byref(st), Z.Z_SYNC_FLUSH)
    assert dst.raw[:st.total_out] == b"".join(make_block("text33", 256, i) for i in range(5))
    L.nx_inflateEnd(C.byref(st))


def test_software_checksums(L, golden_dir):
    for name, fn in (("crc32_kat.json", L.nx_crc32), ("adler32_kat.json", L.nx_adler32)):
        for r in json.load(open(os.path.join(golden_dir, name))):
            buf = None if r["buf"] is None else bytes.fromhex(r["buf"])
            assert fn(r["init"], buf, r["len"] if buf is not None else 0) == r["expect"], r
    a, b = make_block("random", 1000, 1), make_block("random", 70001, 2)
    assert L.nx_crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    assert L.nx_adler32_combine(zlib.adler32(a), zlib.adler32(b), len(b)) == zlib.adler32(a + b)


@pytest.mark.parametrize("size,step_out", [(3 << 20, 65536), (3 << 20, 1 << 20), (200000, 4096), (40000, 1000)])
def test_next_in_stands_right_behind_the_stream_at_stream_end(L, size, step_out):
    """zlib's contract (and lib/nx_inflate.c:1614-1623 update_stream_in: next_in moves only over what the engine
    consumed): at Z_STREAM_END next_in / avail_in / total_in point just past the stream, so a caller that reads
    concatenated gzip members (inflateEnd + inflateInit per member, CPython's unused_data) finds the next member.
    Round-2 advisor finding: with a large avail_in and a small avail_out up to a megabyte of the caller's input
    was swallowed into the stream state and the start of the next member was lost."""
    a = make_block("alice", size, 3)
    b = make_block("lz", 70000, 4)
    m1 = zlib.compressobj(6, zlib.DEFLATED, 31)
    m1 = m1.compress(a) + m1.flush()
    m2 = zlib.compressobj(6, zlib.DEFLATED, 31)
    m2 = m2.compress(b) + m2.flush()
    both = m1 + m2
    src = C.create_string_buffer(both, len(both))
    dst = C.create_string_buffer(step_out)
    st = Z.ZStream()
    assert L.nx_inflateInit2_(C.byref(st), 31, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_OK
    st.next_in = C.addressof(src)
    st.avail_in = len(both)                                    # everything at once: both members
    got = bytearray()
    rc = Z.Z_OK
    for _ in range(1000000):
        st.next_out = C.addressof(dst)
        st.avail_out = step_out
        rc = L.nx_inflate(C.byref(st), Z.Z_NO_FLUSH)
        got += dst.raw[:step_out - st.avail_out]
        if rc != Z.Z_OK:
            break
    assert rc == Z.Z_STREAM_END and bytes(got) == a
    assert st.avail_in == len(m2) and st.total_in == len(m1)
    assert st.next_in == C.addressof(src) + len(m1)
    L.nx_inflateEnd(C.byref(st))
    # the second member is read from where next_in stands
    st2 = Z.ZStream()
    assert L.nx_inflateInit2_(C.byref(st2), 31, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_OK
    st2.next_in = st.next_in
    st2.avail_in = st.avail_in
    out2 = C.create_string_buffer(len(b) + 64)
    st2.next_out = C.addressof(out2)
    st2.avail_out = len(out2)
    assert L.nx_inflate(C.byref(st2), Z.Z_FINISH) == Z.Z_STREAM_END
    assert out2.raw[:st2.total_out] == b and st2.avail_in == 0
    L.nx_inflateEnd(C.byref(st2))


@pytest.mark.parametrize("n", [0, 10, 4096, 300000])
def test_canonical_zpipe_loop_sees_the_end_of_short_streams_and_short_last_chunks(L, n):
    """The canonical client loop of zlib's zpipe.c (the reference ships it: samples/zpipe.c:100-146): feed what a
    read gave with Z_NO_FLUSH, stop at Z_STREAM_END, treat end of input before that as a truncated stream.  A
    stream shorter than the reference's input cache threshold, or whose last chunk is, must still end in
    Z_STREAM_END within the call that brings its last byte (found by tests/test_gpu_oct.py)."""
    data = make_block("alice", n, 5) if n else b""
    comp = zlib.compress(data, 6)
    chunk = 4096 if n < 100000 else len(comp) - 200            # the long one: a 200-byte last chunk
    st = Z.ZStream()
    assert L.nx_inflateInit2_(C.byref(st), 15, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_OK
    out = C.create_string_buffer(1 << 19)
    got = bytearray()
    rc = Z.Z_OK
    pos = 0
    while rc != Z.Z_STREAM_END:
        piece = comp[pos:pos + chunk]
        pos += len(piece)
        assert piece, "end of input before Z_STREAM_END"
        src = C.create_string_buffer(piece, len(piece))
        st.next_in = C.addressof(src)
        st.avail_in = len(piece)
        while True:
            st.next_out = C.addressof(out)
            st.avail_out = len(out)
            rc = L.nx_inflate(C.byref(st), Z.Z_NO_FLUSH)
            assert rc in (Z.Z_OK, Z.Z_STREAM_END, Z.Z_BUF_ERROR), rc
            got += out.raw[:len(out) - st.avail_out]
            if st.avail_out != 0:
                break
    L.nx_inflateEnd(C.byref(st))
    assert bytes(got) == data
