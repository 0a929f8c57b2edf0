"""GPU tests of the parallel decode of ONE long deflate stream (nxz_inflate_stream: block-boundary
speculation, -m gpu): zlib-made streams of several levels and strategies, stored / fixed stretches
in between, a history in front, damage, streams the engine must decline; and the same through the
zlib-style API (nx_uncompress / nx_inflate on a whole .gz), where it replaces the job-after-job loop."""
import ctypes as C
import importlib
import os
import sys
import zlib

import numpy as np
import pytest

import corpus
import zstream as Z

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("power-gzip_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    e = pkg.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def data():
    _, blocks, _ = corpus.load(65536)
    raw = b"".join(b for _, _, b in blocks)
    return (raw * 2)[:24 << 20]


def _run(eng, comp, cap, first_bit=0, hist=None):
    import torch
    src = torch.from_numpy(np.frombuffer(comp, np.uint8).copy()).to(eng.dev)
    dst = torch.zeros(cap, dtype=torch.uint8, device=eng.dev)
    h = torch.from_numpy(np.frombuffer(hist, np.uint8).copy()).to(eng.dev) if hist else None
    rc, info = eng.inflate_stream(src, len(comp), dst, first_bit=first_bit, hist=h)
    torch.cuda.synchronize()
    return rc, info, dst


@pytest.mark.parametrize("level,strategy", [(1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (6, zlib.Z_FILTERED), (6, zlib.Z_HUFFMAN_ONLY)])
def test_one_stream_bit_exact(eng, data, level, strategy):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    comp = c.compress(data) + c.flush()
    rc, info, dst = _run(eng, comp, len(data) + 4096)
    assert rc == 0, (rc, info)
    assert info["out_len"] == len(data) and info["crc"] == zlib.crc32(data) and info["adler"] == zlib.adler32(data)
    assert dst[:len(data)].cpu().numpy().tobytes() == data
    assert info["pieces"] >= 16 and (info["end_bit"] + 7) // 8 == len(comp)


def test_stored_and_fixed_stretches_and_flush_points(eng, data):
    """random bytes (stored blocks), a Z_FIXED stretch and full-flush points inside one stream: block
    starts that the search does not look for are decoded as part of the piece in front"""
    rnd = np.random.default_rng(5).integers(0, 256, 3 << 20, dtype=np.uint8).tobytes()
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = [data[:6 << 20], rnd, data[6 << 20:12 << 20]]
    comp = b""
    for p in parts:
        comp += c.compress(p) + c.flush(zlib.Z_FULL_FLUSH)
    # a fixed-Huffman stretch: a second compressor continues the same raw stream (Z_FULL_FLUSH left it byte aligned)
    f = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
    tailpart = data[12 << 20:14 << 20]
    comp += f.compress(tailpart) + f.flush()
    plain = b"".join(parts) + tailpart
    rc, info, dst = _run(eng, comp, len(plain) + 4096)
    assert rc == 0, (rc, info)
    assert info["out_len"] == len(plain) and info["crc"] == zlib.crc32(plain)
    assert dst[:len(plain)].cpu().numpy().tobytes() == plain


def test_a_block_header_that_is_not_one(eng, data):
    """Deflate output carried as DATA inside the stream (it does not compress, so zlib stores it): the
    stored bytes start with a perfectly good dynamic block header that is no block start.  The piece in
    front of it does not end there, the start is dropped and only the merged piece is decoded again."""
    c0 = zlib.compressobj(6, zlib.DEFLATED, -15)
    inner = c0.compress(data[:1 << 20]) + c0.flush()          # a raw stream: its first bits are a dynamic block header
    assert (inner[0] & 7) == 4                                 # BFINAL 0, BTYPE 10
    plain = data[:5 << 20] + inner[:200000] + data[5 << 20:9 << 20] + inner[:70000] + data[9 << 20:12 << 20]
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(plain) + c.flush()
    rc, info, dst = _run(eng, comp, len(plain) + 4096)
    assert rc == 0, (rc, info)
    assert info["out_len"] == len(plain) and info["crc"] == zlib.crc32(plain)
    assert dst[:len(plain)].cpu().numpy().tobytes() == plain


@pytest.mark.parametrize("off", [1, 3, 8, 13])
def test_a_stream_that_does_not_start_on_a_16_byte_boundary(eng, data, off):
    """The pieces read the caller's stream in place when it starts on a 16-byte boundary of device memory; one that
    does not gets aligned copies of its pieces (nxz_pinflate.cpp `direct`): same bytes out either way."""
    import torch
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    plain = data[:6 << 20]
    comp = c.compress(plain) + c.flush()
    buf = torch.zeros(len(comp) + 64, dtype=torch.uint8, device=eng.dev)
    buf[off:off + len(comp)] = torch.from_numpy(np.frombuffer(comp, np.uint8).copy()).to(eng.dev)
    src = buf[off:off + len(comp)]
    assert src.data_ptr() % 16 == off % 16
    dst = torch.zeros(len(plain) + 4096, dtype=torch.uint8, device=eng.dev)
    rc, info = eng.inflate_stream(src, len(comp), dst)
    torch.cuda.synchronize()
    assert rc == 0, (rc, info)
    assert info["out_len"] == len(plain) and info["crc"] == zlib.crc32(plain)
    assert dst[:len(plain)].cpu().numpy().tobytes() == plain


def _stream_with_headers_as_stored_data(data):
    """data[:5 MiB] deflated | 6.5 MiB of STORED blocks that hold a 15-byte dynamic block header (zlib's own, of a
    block of four letters) every 20 bytes | data[5 MiB:11 MiB] deflated: one raw deflate stream of > 8 MiB"""
    rng = np.random.default_rng(3)
    four = bytes(rng.choice([97, 98, 99, 100], size=4000, p=[.5, .25, .15, .1]).astype(np.uint8))
    c0 = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_HUFFMAN_ONLY)
    hdr = (c0.compress(four) + c0.flush())[:15]
    assert (hdr[0] & 6) == 4                                    # BTYPE 10
    junk = b"".join(hdr + rng.integers(0, 256, 5, dtype=np.uint8).tobytes() for _ in range((6500 << 10) // 20))
    c1 = zlib.compressobj(6, zlib.DEFLATED, -15)
    part1 = c1.compress(data[:5 << 20]) + c1.flush(zlib.Z_SYNC_FLUSH)            # ends on a byte boundary, not final
    stored = b"".join(b"\x00" + len(ch).to_bytes(2, "little") + (len(ch) ^ 0xffff).to_bytes(2, "little") + ch
                      for ch in (junk[o:o + 65535] for o in range(0, len(junk), 65535)))
    c2 = zlib.compressobj(6, zlib.DEFLATED, -15)
    part2 = c2.compress(data[5 << 20:11 << 20]) + c2.flush()
    plain = data[:5 << 20] + junk + data[5 << 20:11 << 20]
    comp = part1 + stored + part2
    assert zlib.decompress(comp, -15) == plain and len(comp) > (8 << 20)
    return comp, plain


def test_hundreds_of_headers_that_are_none_in_one_segment(eng, data):
    """A short, perfectly good dynamic block header every 20 bytes of stored data: 400 candidates in every 8 KiB
    segment of the search pass every test -- more than its first kernel hands to the second (LEFT_MAX = 126: such a
    segment is done by the first kernel itself), and every segment of 6.5 MiB reports a block start that is none.
    The stream must still come out bit exact (the pieces in front do not end at those starts; stored_walk_kernel
    carries the piece over the run of stored blocks)."""
    comp, plain = _stream_with_headers_as_stored_data(data)
    rc, info, dst = _run(eng, comp, len(plain) + 4096)
    assert rc == 0, (rc, info)
    assert info["out_len"] == len(plain) and info["crc"] == zlib.crc32(plain)
    assert dst[:len(plain)].cpu().numpy().tobytes() == plain


def test_history_in_front_and_bit_offset(eng, data):
    """a stream that starts in the middle of a byte and refers to a preset dictionary"""
    dic = data[100000:100000 + 32768]
    c = zlib.compressobj(6, zlib.DEFLATED, -15, zdict=dic)
    comp = c.compress(data[:8 << 20]) + c.flush()
    # shift the whole stream by 3 bits
    v = int.from_bytes(comp, "little") << 3
    shifted = v.to_bytes(len(comp) + 1, "little")
    rc, info, dst = _run(eng, shifted, (8 << 20) + 4096, first_bit=3, hist=dic)
    assert rc == 0, (rc, info)
    assert dst[:8 << 20].cpu().numpy().tobytes() == data[:8 << 20] and info["crc"] == zlib.crc32(data[:8 << 20])


def test_declined_and_damaged(eng, data):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data[:8 << 20]) + c.flush()
    # too short a stream: the ordinary loop is the right tool
    rc, _, _ = _run(eng, comp[:30000], 8 << 20)
    assert rc == -95                                   # ENOTSUP
    rc, _, _ = _run(eng, comp[:500000], 8 << 20)       # long enough, but cut off: no final block
    assert rc in (-95, -84)
    # no final block inside the source
    rc, _, _ = _run(eng, comp[:len(comp) - 4000], 9 << 20)
    assert rc in (-95, -84)                            # ENOTSUP / EILSEQ
    # target too small: says how much it needs
    rc, info, _ = _run(eng, comp, 1 << 20)
    assert rc == -7 and info["out_len"] == 8 << 20     # E2BIG
    # damage in the middle: either refused or caught by the checksum the caller compares
    bad = bytearray(comp)
    for k in range(len(bad) // 2, len(bad) // 2 + 64):
        bad[k] ^= 0x5a
    rc, info, _ = _run(eng, bytes(bad), 9 << 20)
    assert rc != 0 or info["crc"] != zlib.crc32(data[:8 << 20])


def test_through_the_zlib_style_api(data):
    """nx_uncompress on a whole zlib stream and nx_inflate(Z_FINISH) on a whole .gz: megabytes of input
    at once take the parallel path (NX_GZIP_TRACE-free check: the result and the trailer verification)"""
    L = Z.load("gpu")
    plain = data[:16 << 20]
    z = zlib.compress(plain, 6)
    out = C.create_string_buffer(len(plain) + 64)
    n = C.c_ulong(len(out))
    assert L.nx_uncompress(out, C.byref(n), z, len(z)) == 0
    assert n.value == len(plain) and out.raw[:n.value] == plain
    # gzip wrapper through the streaming call, everything at once
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz = co.compress(plain) + co.flush()
    got, rc, total_in, _ = Z.inflate_all(L, gz, wbits=31, cap=len(plain) + 64)
    assert got == plain and rc == Z.Z_STREAM_END and total_in == len(gz)
    # a damaged trailer is still a data error
    badgz = gz[:-5] + bytes([gz[-5] ^ 1]) + gz[-4:]
    got, rc, _, _ = Z.inflate_all(L, badgz, wbits=31, cap=len(plain) + 64)
    assert rc == Z.Z_DATA_ERROR


@pytest.mark.parametrize("kib", [96, 200, 700, 3000])
def test_streams_of_a_few_blocks(eng, data, kib):
    """streams of a few hundred KiB: a few blocks, cut at token boundaries inside them.  The engine takes every
    stream of 12 KiB or more (nxz_pinflate.cpp: below that one wavefront is as fast); a stream it is documented
    to take must be taken (rc 0, three pieces or more), so the test cannot pass without the path having run."""
    plain = data[1 << 20:(1 << 20) + (kib << 10)]
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(plain) + c.flush()
    if len(comp) < (12 << 10):
        # too short for the parallel path by its own rule: take as much more of the data as makes 12 KiB
        rc, _, _ = _run(eng, comp, len(plain) + 4096)
        assert rc == -95
        while len(comp) < (13 << 10):
            plain = data[1 << 20:(1 << 20) + len(plain) + (64 << 10)]
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = c.compress(plain) + c.flush()
    rc, info, dst = _run(eng, comp, len(plain) + 4096)
    assert rc == 0, (rc, info)
    assert info["pieces"] >= 3, info
    assert info["out_len"] == len(plain) and info["crc"] == zlib.crc32(plain)
    assert dst[:len(plain)].cpu().numpy().tobytes() == plain


@pytest.mark.parametrize("step_in,step_out", [(1 << 20, 1 << 20), (256 << 10, 64 << 10), (100000, 1 << 20), (3 << 20, 200000), (64 << 10, 4 << 20)])
def test_inflate_in_steps_takes_parts_of_the_stream(data, step_in, step_out):
    """inflate() with avail_in / avail_out of 64 KiB - 3 MiB: every call hands the engine a PART
    of the stream -- it begins wherever the last one stopped (inside a block, with the resume fields) and
    ends where the source ends; the output beyond avail_out waits for the next call"""
    L = Z.load("gpu")
    plain = data[:12 << 20]
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz = co.compress(plain) + co.flush()
    got, rc, total_in, _ = Z.inflate_all(L, gz, wbits=31, cap=len(plain) + 64, step_in=step_in, step_out=step_out)
    assert rc == Z.Z_STREAM_END and total_in == len(gz)
    assert got == plain


def test_inflate_in_steps_of_a_mixed_stream(data):
    """stored stretches, a fixed-Huffman stretch, flush points, huge and tiny blocks, a preset dictionary"""
    L = Z.load("gpu")
    rnd = np.random.default_rng(11).integers(0, 256, 2 << 20, dtype=np.uint8).tobytes()
    dic = data[200000:200000 + 32768]
    c = zlib.compressobj(6, zlib.DEFLATED, -15, zdict=dic)
    parts = [data[:3 << 20], rnd, data[3 << 20:5 << 20], b"\0" * (3 << 20), data[5 << 20:(5 << 20) + 70000]]
    comp = b""
    for k, p in enumerate(parts):
        comp += c.compress(p) + c.flush(zlib.Z_FULL_FLUSH if k & 1 else zlib.Z_SYNC_FLUSH)
    f = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
    more = data[6 << 20:8 << 20]
    comp += f.compress(more) + f.flush()
    plain = b"".join(parts) + more
    for step_in, step_out in ((512 << 10, 512 << 10), (150000, 70000), (len(comp), len(plain) + 64)):
        got, rc, total_in, _ = Z.inflate_all(L, comp, wbits=-15, cap=len(plain) + 64, step_in=step_in, step_out=step_out, dictionary=dic)
        assert rc == Z.Z_STREAM_END and total_in == len(comp), (step_in, step_out, rc)
        assert got == plain, (step_in, step_out)


def test_inflate_in_steps_stops_at_the_end_of_the_member(data):
    """two gzip members back to back: inflate() ends with the first; what follows stays with the caller"""
    L = Z.load("gpu")
    a, b = data[:5 << 20], data[5 << 20:7 << 20]
    ga = zlib.compressobj(6, zlib.DEFLATED, 31); gza = ga.compress(a) + ga.flush()
    gb = zlib.compressobj(9, zlib.DEFLATED, 31); gzb = gb.compress(b) + gb.flush()
    for step_in in (len(gza) + len(gzb), 700000):
        got, rc, total_in, _ = Z.inflate_all(L, gza + gzb, wbits=31, cap=len(a) + 64, step_in=step_in, step_out=1 << 20)
        assert rc == Z.Z_STREAM_END and got == a and total_in == len(gza), (step_in, rc, total_in, len(gza))


def test_damage_is_still_a_data_error_in_steps(data):
    L = Z.load("gpu")
    plain = data[:6 << 20]
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz = bytearray(co.compress(plain) + co.flush())
    for k in range(len(gz) // 2, len(gz) // 2 + 48):
        gz[k] ^= 0xa5
    got, rc, _, _ = Z.inflate_all(L, bytes(gz), wbits=31, cap=len(plain) + (1 << 20), step_in=1 << 20, step_out=1 << 20)
    assert rc == Z.Z_DATA_ERROR


def test_streams_made_by_this_engine_are_cut_at_every_block(eng, data):
    """The NX table generator (and this engine's, which makes the same tables) sends all 286 + 30 code lengths
    with all 19 code-length-code lengths, used or not: headers no other encoder writes.  The block search
    takes them as they come, so a stream made by nx_deflate / the reference is decoded side by side too."""
    plain = data[:8 << 20]
    rc, stream, crc, _ = eng.deflate_host(plain)
    assert rc == 0 and zlib.decompress(stream, -15) == plain
    rc, info, dst = _run(eng, stream, len(plain) + 4096)
    assert rc == 0, (rc, info)
    assert info["out_len"] == len(plain) and info["crc"] == crc == zlib.crc32(plain)
    assert dst[:len(plain)].cpu().numpy().tobytes() == plain
    assert info["pieces"] >= len(plain) // 65536          # a piece per 64 KiB block at least (cuts inside the blocks come on top)


def test_the_part_interface_walks_a_stream_on_the_device(eng, data):
    """nxz_inflate_stream_part itself (include/nxz_engine.h), device buffers only: a stream is fed in parts of
    odd sizes; each call gets the state the last one returned (inside a block, as a rule), the last 32 KiB of
    output as history and the byte the last part stopped in as its first; CRCs combine to the stream's"""
    import torch
    plain = data[2 << 20:11 << 20]
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(plain) + c.flush()
    L = Z.load("gpu")
    got = bytearray()
    crc = 0
    pos, first_bit, state = 0, 0, None
    hist = None
    sizes = [300000, 70000, 1 << 20, 15000, 555555]
    k = 0
    while True:
        n = min(sizes[k % len(sizes)], len(comp) - pos); k += 1
        src = torch.from_numpy(np.frombuffer(comp[pos:pos + n], np.uint8).copy()).to(eng.dev)
        dst = torch.zeros(n * 12 + (1 << 20), dtype=torch.uint8, device=eng.dev)
        rc, info = eng.inflate_stream_part(src, n, dst, state=state, first_bit=first_bit, hist=hist)
        torch.cuda.synchronize()
        assert rc == 0, (rc, pos, n, info)
        out = dst[:info["out_len"]].cpu().numpy().tobytes()
        assert info["crc"] == zlib.crc32(out)
        crc = L.nx_crc32_combine(crc, info["crc"], len(out))
        got += out
        state = info["state"]
        if state.final:
            assert pos + (info["end_bit"] + 7) // 8 == len(comp)
            break
        assert info["end_bit"] > first_bit
        pos += info["end_bit"] >> 3                       # whole bytes used; the byte it stopped in comes again
        first_bit = info["end_bit"] & 7
        tail = bytes(got[-32768:])
        hist = torch.from_numpy(np.frombuffer(tail, np.uint8).copy()).to(eng.dev)
        assert pos < len(comp)
    assert bytes(got) == plain and crc == zlib.crc32(plain)


def test_parts_beyond_the_pinned_staging_from_several_threads_at_once():
    """Callers in company stage their copies through pinned memory (nxz_stream.cpp parallel_inflate) up to 8 MiB of source and
    16 MiB of output; beyond, the copies go straight from and to the caller's pages.  Four threads at once: nx_uncompress of
    40 MiB (21 MiB of stream), and inflate() of the same stream with all the input and room for 30 MiB of the output (the rest
    waits and comes with the next calls): every byte and the checksum as zlib has them."""
    import threading
    from datagen import make_block
    L = Z.load("gpu")
    kinds = ("alice", "text33", "binary", "random", "lz")
    base = [make_block(kinds[k % 5], 65536, 8800 + k) for k in range(160)]
    datas = [b"".join(base[(k * (1, 3, 7, 9)[t] + t) % 160] for k in range(640)) for t in range(4)]       # (steps coprime to 160: no block comes again within 10 MiB)
    comps = [zlib.compress(d, 1) for d in datas]
    assert min(len(c) for c in comps) > (9 << 20)
    bad = []

    def worker(t):
        d, c = datas[t], comps[t]
        back = C.create_string_buffer(len(d))
        for rep in range(2):
            n = C.c_ulong(len(back))
            if L.nx_uncompress(back, C.byref(n), c, len(c)) != Z.Z_OK or n.value != len(d) or back.raw != d:
                bad.append((t, rep, "uncompress"))
                return
        st = Z.ZStream()
        if L.nx_inflateInit2_(C.byref(st), 15, Z.VERSION, C.sizeof(Z.ZStream)) != Z.Z_OK:
            bad.append((t, "init"))
            return
        src = C.create_string_buffer(c, len(c))
        st.next_in = C.addressof(src); st.avail_in = len(c)
        got = bytearray()
        room = 30 << 20
        rc = Z.Z_OK
        for _ in range(64):
            st.next_out = C.addressof(back); st.avail_out = room
            rc = L.nx_inflate(C.byref(st), Z.Z_NO_FLUSH)
            got += back.raw[:room - st.avail_out]
            if rc != Z.Z_OK:
                break
            room = 3 << 20
        if rc != Z.Z_STREAM_END or bytes(got) != d or st.adler != zlib.adler32(d) or st.total_in != len(c):
            bad.append((t, "inflate", rc, len(got)))
        L.nx_inflateEnd(C.byref(st))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not bad, bad


_IDLE_BUDGET_SCRIPT = r"""
import ctypes as C, os, sys, threading, zlib
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import zstream as Z
from datagen import make_block
L = Z.load("gpu")
E = C.CDLL(os.path.join(ROOT, "power-gzip_amd", "libnxz_engine.so"))
E.nxz_pinflate_trim.restype = C.c_size_t
kinds = ("alice", "lz", "text33", "binary")
T = 6
datas = [b"".join(make_block(kinds[(t + k) % 4], 65536, 61 * t + k) for k in range(64)) for t in range(T)]     # 4 MiB each
comps = [zlib.compress(d, 6) for d in datas]
bad = []
def worker(t):
    back = C.create_string_buffer(len(datas[t]))
    for _ in range(4):
        n = C.c_ulong(len(back))
        if L.nx_uncompress(back, C.byref(n), comps[t], len(comps[t])) != 0 or back.raw[:n.value] != datas[t]:
            bad.append(t)
            return
th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
for x in th: x.start()
for x in th: x.join()
print("bad", len(bad), "idle", E.nxz_pinflate_trim())
"""


def test_idle_workspaces_stay_inside_their_budget_and_busy_ones_are_left_alone():
    """NXZ_PINFLATE_IDLE_MB: what the one-stream workspaces hold BETWEEN calls.  Six threads of 4 MiB nx_uncompress calls with a
    budget of 300 MiB (a workspace of such a call is some 350 MiB): every call gets its bytes, and what nxz_pinflate_trim() finds
    to give back when all is over -- all that lay idle -- is no more than the budget and one workspace (two calls that end at the
    same moment: the one that leaves the device idle cannot take the other's, which is still locked), where six untrimmed
    workspaces would be 2 GiB.  (The limits are about idle workspaces, not about those that calls are working in, and while
    others work a workspace stays for two seconds: nxz_pinflate.cpp TrimOnExit.)"""
    import subprocess
    env = dict(os.environ, NXZ_PINFLATE_IDLE_MB="300")
    p = subprocess.run([sys.executable, "-c", _IDLE_BUDGET_SCRIPT, ROOT], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("bad ")]
    assert p.returncode == 0 and line, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    f = line[-1].split()
    assert int(f[1]) == 0
    assert int(f[3]) <= (300 << 20) + (512 << 20), line[-1]
