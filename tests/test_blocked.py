"""Blocked gzip files (include/nxz_blocked.h) and the nxz_gzip tool: the on-disk step either side
of the batched hot path (reference counterparts lib/nx_gzlib.c, samples/nx_gzip.c).
CPU: the member walker and the end marker (no device work).  GPU: buffers and files through the
engine, checked with Python's gzip/zlib (any gzip reader must accept the output)."""
import ctypes as C
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from datagen import make_block

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "power-gzip_amd", "libnxz_amd.so")
CLI = os.path.join(ROOT, "power-gzip_amd", "nxz_gzip")
EOF_MARKER = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)


class Opts(C.Structure):
    _fields_ = [("device", C.c_int), ("fixed", C.c_int), ("block_size", C.c_uint32), ("chunk_blocks", C.c_uint32),
                ("group", C.c_uint32), ("reserved", C.c_uint32 * 3)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB)          # finds libnxz_engine.so through its run path; nothing is made global
        L.nxz_blocked_deflate.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Opts), SINK, C.c_void_p, C.POINTER(C.c_uint64)]
        L.nxz_blocked_inflate.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Opts), SINK, C.c_void_p, C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_size_t)]
        L.nxz_blocked_scan.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]
        L.nxz_blocked_end_marker.argtypes = [SINK, C.c_void_p]
        _lib = L
    return _lib


def collector():
    parts = []

    def cb(user, buf, n):
        parts.append(C.string_at(buf, n))
        return 0
    return parts, SINK(cb)


def py_member(data, level=6):
    """a BGZF member made with zlib: what bgzip writes"""
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    pay = co.compress(data) + co.flush()
    size = 18 + len(pay) + 8
    return (b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, size - 1) +
            pay + struct.pack("<II", zlib.crc32(data), len(data)))


def members_of(image):
    """[(offset, size)] by hopping over the BC subfields"""
    out, pos = [], 0
    while pos < len(image):
        assert image[pos:pos + 4] == b"\x1f\x8b\x08\x04", pos
        assert image[pos + 12:pos + 16] == b"BC\x02\x00"
        size = struct.unpack_from("<H", image, pos + 16)[0] + 1
        out.append((pos, size))
        pos += size
    assert pos == len(image)
    return out


def sample(n_bytes, seed=0):
    kinds = ("alice", "lz", "text33", "random", "zeros", "periodic", "binary")
    out = bytearray()
    i = 0
    while len(out) < n_bytes:
        out += make_block(kinds[i % len(kinds)], 65536 if i % 3 else 40000 + 17 * i, seed + i)
        i += 1
    return bytes(out[:n_bytes])


# ---------------------------------------------------------------------------------------- CPU
def test_scan_walks_members_and_stops_at_foreign_data():
    L = lib()
    blocks = [make_block("alice", 65280, 1), make_block("lz", 65280, 2), make_block("text33", 1234, 3)]
    image = b"".join(py_member(b) for b in blocks) + EOF_MARKER
    m, u, used = C.c_uint64(), C.c_uint64(), C.c_size_t()
    assert L.nxz_blocked_scan(image, len(image), C.byref(m), C.byref(u), C.byref(used)) == 0
    assert (m.value, u.value, used.value) == (4, sum(map(len, blocks)), len(image))
    # an ordinary gzip member after the blocked ones, and a cut-off member, end the walk
    tail = gzip.compress(b"plain member")
    for extra in (tail, image[:40]):
        buf = image + extra
        assert L.nxz_blocked_scan(buf, len(buf), C.byref(m), C.byref(u), C.byref(used)) == 0
        assert (m.value, used.value) == (4, len(image))
    assert L.nxz_blocked_scan(tail, len(tail), C.byref(m), C.byref(u), C.byref(used)) == 0
    assert (m.value, u.value, used.value) == (0, 0, 0)


def test_end_marker_is_the_bgzf_one():
    parts, cb = collector()
    assert lib().nxz_blocked_end_marker(cb, None) == 0
    assert b"".join(parts) == EOF_MARKER
    assert gzip.decompress(EOF_MARKER) == b""


def test_blocked_symbols_are_exported():
    import re
    hdr = open(os.path.join(ROOT, "include", "nxz_blocked.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(nxz_blocked_\w+)\s*\(", hdr))
    assert names == {"nxz_blocked_deflate", "nxz_blocked_end_marker", "nxz_blocked_scan", "nxz_blocked_inflate"}
    for n in names:
        assert hasattr(lib(), n)
    eng = C.CDLL(os.path.join(ROOT, "power-gzip_amd", "libnxz_engine.so"))
    for n in ("nxz_batch_pack_gzip", "nxz_dev_malloc", "nxz_dev_free", "nxz_pinned_malloc", "nxz_pinned_free",
              "nxz_stream_create", "nxz_stream_destroy", "nxz_copy_to_device", "nxz_copy_to_host"):
        assert hasattr(eng, n), n
    assert os.access(CLI, os.X_OK)


# ---------------------------------------------------------------------------------------- GPU
def deflate(data, **kw):
    parts, cb = collector()
    o = Opts(device=-1, **kw)
    n = C.c_uint64()
    rc = lib().nxz_blocked_deflate(data, len(data), C.byref(o), cb, None, C.byref(n))
    assert rc == 0, rc
    image = b"".join(parts)
    assert n.value == len(image)
    return image


def inflate(image, **kw):
    parts, cb = collector()
    o = Opts(device=-1, **kw)
    n, used = C.c_uint64(), C.c_size_t()
    rc = lib().nxz_blocked_inflate(image, len(image), C.byref(o), cb, None, C.byref(n), C.byref(used))
    return rc, b"".join(parts), n.value, used.value


@pytest.mark.gpu
@pytest.mark.parametrize("fixed", [0, 1])
def test_buffer_to_members_any_gzip_reader_accepts(fixed):
    data = sample(3_000_000 + 4321, seed=3)
    image = deflate(data, fixed=fixed, chunk_blocks=7)          # several batches, two in flight
    mem = members_of(image)
    assert len(mem) == (len(data) + 65279) // 65280 and max(s for _, s in mem) <= 65536
    assert gzip.decompress(image) == data                       # multi-member gzip, checked member by member by zlib
    # each member is its block, alone
    for k in (0, 1, len(mem) // 2, len(mem) - 1):
        off, size = mem[k]
        assert gzip.decompress(image[off:off + size]) == data[k * 65280:(k + 1) * 65280]
    # one batch gives the same bytes as many (dynamic: table groups restart with every batch)
    one = deflate(data, fixed=fixed)
    assert gzip.decompress(one) == data and (not fixed or one == image)
    if not fixed:
        fixed_image = deflate(data, fixed=1, chunk_blocks=7)
        assert len(image) < len(fixed_image)                    # dynamic tables pay on this mix
    # and back through the GPU, members in parallel
    rc, back, n, used = inflate(image + EOF_MARKER, chunk_blocks=11)
    assert (rc, n, used) == (0, len(data), len(image) + 28) and back == data


@pytest.mark.gpu
def test_incompressible_blocks_become_stored_members_and_small_inputs():
    rnd = np.random.default_rng(1).integers(0, 256, 200_000, dtype=np.uint8).tobytes()
    image = deflate(rnd, fixed=1)
    for off, size in members_of(image):
        assert image[off + 18] == 0x01                          # BFINAL=1 BTYPE=00
    assert len(image) == len(rnd) + len(members_of(image)) * (26 + 5)
    assert gzip.decompress(image) == rnd
    assert inflate(image)[1] == rnd
    for n in (1, 15, 16, 17, 65279, 65280, 65281):
        d = sample(n, seed=n)
        for fixed in (0, 1):
            img = deflate(d, fixed=fixed)
            assert gzip.decompress(img) == d
            assert inflate(img)[:3] == (0, d, n)
    parts, cb = collector()
    assert lib().nxz_blocked_deflate(b"", 0, None, cb, None, None) == 0 and parts == []
    # other block sizes
    d = sample(500_000, seed=9)
    img = deflate(d, block_size=4096, group=8)
    assert len(members_of(img)) == (len(d) + 4095) // 4096 and gzip.decompress(img) == d


@pytest.mark.gpu
def test_inflate_of_foreign_blocked_files_and_damage():
    data = sample(1_500_000, seed=21)
    blocks = [data[i:i + 65280] for i in range(0, len(data), 65280)]
    blocks.insert(3, b"")                                       # an empty member in the middle
    blocks.insert(5, data[:777])                                # a short one: later outputs lose their alignment
    image = b"".join(py_member(b, level=(1, 6, 9)[i % 3]) for i, b in enumerate(blocks)) + EOF_MARKER
    want = b"".join(blocks)
    rc, back, n, used = inflate(image, chunk_blocks=5)
    assert (rc, n, used) == (0, len(want), len(image)) and back == want
    # a flipped payload bit, a wrong CRC and a wrong ISIZE are all refused
    mem = members_of(image)
    off, size = mem[7]
    for pos, what in ((off + 18 + 100, "payload"), (off + size - 8, "crc"), (off + size - 4, "isize")):
        bad = bytearray(image)
        bad[pos] ^= 0x10
        assert inflate(bytes(bad))[0] == -84, what              # -EILSEQ
    # an absurd ISIZE is refused before anything is allocated for it
    bad = bytearray(image)
    bad[off + size - 1] = 0x7f
    assert inflate(bytes(bad))[0] == -84
    # ordinary gzip data is not ours to decode here
    assert inflate(gzip.compress(data[:100000]))[0] == 1
    # whole members only: a cut-off tail is reported through `consumed`
    rc, back, n, used = inflate(image[:mem[4][0] + 10])
    assert rc == 0 and used == mem[4][0] and back == b"".join(blocks[:4])


def run_cli(args, stdin=None):
    return subprocess.run([CLI] + args, input=stdin, capture_output=True, check=False)


@pytest.mark.gpu
def test_cli_round_trips_and_reads_ordinary_gzip(tmp_path):
    data = sample(2_345_678, seed=33)
    f = tmp_path / "data.bin"
    f.write_bytes(data)
    r = run_cli(["-k", "-v", str(f)])
    assert r.returncode == 0, r.stderr
    gz = tmp_path / "data.bin.gz"
    image = gz.read_bytes()
    assert f.exists() and image.endswith(EOF_MARKER)
    assert gzip.decompress(image) == data
    assert len(image) < 0.75 * len(data)
    assert run_cli(["-t", str(gz)]).returncode == 0
    lst = run_cli(["-l", str(gz)])
    assert lst.returncode == 0 and str(len(data)).encode() in lst.stdout
    # refuses to overwrite, then -f; the original goes away without -k
    assert run_cli([str(f)]).returncode == 1
    assert run_cli(["-f", "-F", str(f)]).returncode == 0 and not f.exists()
    assert gzip.decompress(gz.read_bytes()) == data
    assert run_cli(["-d", str(gz)]).returncode == 0 and f.read_bytes() == data and not gz.exists()
    # pipes
    r = run_cli(["-c"], stdin=data[:300000])
    assert r.returncode == 0 and gzip.decompress(r.stdout) == data[:300000]
    r2 = run_cli(["-dc"], stdin=r.stdout)
    assert r2.returncode == 0 and r2.stdout == data[:300000]
    # ordinary gzip files (one member, or several) go through the stream layer
    plain = gzip.compress(data[:400000], 6) + gzip.compress(data[400000:500000], 1)
    r3 = run_cli(["-dc"], stdin=plain)
    assert r3.returncode == 0 and r3.stdout == data[:500000]
    # blocked members followed by an ordinary one
    r4 = run_cli(["-dc"], stdin=r.stdout[:-28] + gzip.compress(b"tail"))
    assert r4.returncode == 0 and r4.stdout == data[:300000] + b"tail"
    # an empty input is one empty member (the end marker) and comes back empty
    e = run_cli(["-c"], stdin=b"")
    assert e.returncode == 0 and e.stdout == EOF_MARKER and gzip.decompress(e.stdout) == b""
    e2 = run_cli(["-dc"], stdin=e.stdout)
    assert e2.returncode == 0 and e2.stdout == b""
    # several files in one call
    a, b = tmp_path / "a.txt", tmp_path / "b.txt"
    a.write_bytes(data[:70000]); b.write_bytes(data[70000:70001])
    assert run_cli([str(a), str(b)]).returncode == 0
    assert gzip.decompress((tmp_path / "a.txt.gz").read_bytes()) == data[:70000]
    assert gzip.decompress((tmp_path / "b.txt.gz").read_bytes()) == data[70000:70001]
    assert run_cli(["-d", str(tmp_path / "a.txt.gz"), str(tmp_path / "b.txt.gz")]).returncode == 0
    assert a.read_bytes() == data[:70000] and b.read_bytes() == data[70000:70001]
    # damage is reported
    bad = bytearray(image)
    bad[5000] ^= 1
    assert run_cli(["-tc"], stdin=bytes(bad)).returncode == 1


@pytest.mark.gpu
def test_plain_c_host_of_the_batch_interface(tmp_path):
    # the boundary is a C ABI: a C program with host buffers and no HIP of its own (tests/native/batch_host.c)
    exe = tmp_path / "batch_host"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "native", "batch_host.c"), "-o", str(exe),
                    "-L", os.path.join(ROOT, "power-gzip_amd"), "-lnxz_engine",
                    "-Wl,-rpath," + os.path.join(ROOT, "power-gzip_amd")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr


def test_scan_survives_garbage():
    """random bytes and mutated images: the walker never reads outside the buffer it was given
    (exact-size ctypes buffers), never reports more than it was given, and stays consistent"""
    import random
    rnd = random.Random(5)
    L = lib()
    good = b"".join(py_member(make_block("alice", 3000 + 500 * i, i)) for i in range(6)) + EOF_MARKER
    m, u, used = C.c_uint64(), C.c_uint64(), C.c_size_t()
    for it in range(3000):
        if it % 3 == 0:
            buf = bytearray(rnd.randbytes(rnd.randrange(0, 200)))
        else:
            buf = bytearray(good[:rnd.randrange(0, len(good) + 1)])
            for _ in range(rnd.randrange(0, 4)):
                if buf:
                    buf[rnd.randrange(len(buf))] = rnd.randrange(256)
        if it % 5 == 0:
            buf[:0] = b"\x1f\x8b\x08\x04" + bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 30)))
        raw = bytes(buf)
        cbuf = C.create_string_buffer(raw, len(raw)) if raw else C.create_string_buffer(1)
        assert L.nxz_blocked_scan(cbuf, len(raw), C.byref(m), C.byref(u), C.byref(used)) == 0
        assert used.value <= len(raw)
        if used.value:
            assert len(members_of_prefix(raw[:used.value])) == m.value


def members_of_prefix(image):
    out, pos = [], 0
    while pos < len(image):
        xlen = struct.unpack_from("<H", image, pos + 10)[0]
        q, size = 0, None
        while q + 4 <= xlen:
            si, slen = image[pos + 12 + q:pos + 14 + q], struct.unpack_from("<H", image, pos + 14 + q)[0]
            if si == b"BC" and slen == 2:
                size = struct.unpack_from("<H", image, pos + 16 + q)[0] + 1
                break
            q += 4 + slen
        assert size
        out.append((pos, size))
        pos += size
    assert pos == len(image)
    return out
