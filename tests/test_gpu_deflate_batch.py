"""nx_deflate / nx_compress2 with hundreds of kilobytes available at once: at the levels that keep no
history between jobs (lib/nx_deflate.c:654-680) all full 64 KiB blocks of a call go to the engine as
ONE batch (nxz_stream.cpp deflate_batch).  What comes out must be an ordinary zlib/gzip/raw stream:
zlib reads it back, checksums and totals agree, every flush mode and every framing still works, and a
block that does not shrink is stored."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

import zstream as Z
from datagen import make_block

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    return Z.load("gpu")


def mixed(n, seed):
    kinds = ["alice", "lz", "random", "zeros", "text33", "alice"]
    out = bytearray()
    i = 0
    while len(out) < n:
        out += make_block(kinds[i % len(kinds)], 65536 + (i % 3) * 4096, seed + i)
        i += 1
    return bytes(out[:n])


@pytest.mark.parametrize("wbits", [15, 31, -15])
@pytest.mark.parametrize("strategy", [Z.Z_DEFAULT_STRATEGY, Z.Z_FIXED])
def test_one_shot_deflate_of_megabytes(L, wbits, strategy):
    data = mixed((3 << 20) + 12345, 7)
    out, rcs, adler = Z.deflate_all(L, data, level=-1, wbits=wbits, strategy=strategy)
    assert zlib.decompress(out, wbits) == data
    if wbits == 15:
        assert adler == zlib.adler32(data)
    if wbits == 31:
        assert adler == zlib.crc32(data)
    # few calls: the whole input went in batches, not in one call per 64 KiB job
    assert len(rcs) <= 4, len(rcs)
    assert len(out) < 0.8 * len(data)


def test_exact_multiple_of_the_block_size_ends_in_a_final_block(L):
    data = mixed(16 * 65536, 3)
    out, _, _ = Z.deflate_all(L, data, level=1, wbits=-15)
    d = zlib.decompressobj(-15)
    assert d.decompress(out) == data and d.eof and d.unused_data == b""


@pytest.mark.parametrize("step_out", [4096, 1 << 20])
@pytest.mark.parametrize("flush", [Z.Z_NO_FLUSH, Z.Z_SYNC_FLUSH, Z.Z_FULL_FLUSH])
def test_streaming_with_large_feeds_and_small_outputs(L, step_out, flush):
    data = mixed((2 << 20) + 777, 11)
    out, _, adler = Z.deflate_all(L, data, level=-1, wbits=15, step_in=700000, step_out=step_out, flush=flush)
    assert zlib.decompress(out) == data
    assert adler == zlib.adler32(data)


def test_incompressible_blocks_are_stored(L):
    data = os.urandom(1 << 20)
    out, _, _ = Z.deflate_all(L, data, level=-1, wbits=15)
    assert zlib.decompress(out) == data
    assert len(out) < len(data) + 16 * 10 + 64


def test_compress2_round_trip_through_zlib_and_nx_uncompress(L):
    data = mixed(5 << 20, 23)
    cap = C.c_ulong(L.nx_compressBound(len(data)))
    dst = C.create_string_buffer(cap.value)
    assert L.nx_compress2(dst, C.byref(cap), data, len(data), 1) == Z.Z_OK
    comp = dst.raw[:cap.value]
    assert zlib.decompress(comp) == data
    n = C.c_ulong(len(data))
    back = C.create_string_buffer(len(data))
    assert L.nx_uncompress(back, C.byref(n), comp, len(comp)) == Z.Z_OK
    assert n.value == len(data) and back.raw == data


@pytest.mark.parametrize("level", [5, 6, 9])
def test_levels_with_history_go_in_one_batch_with_the_input_as_window(L, level):
    """Levels 5..9 carry 4..32 KiB of the earlier INPUT into the next job (lib/nx_deflate.c:654-680,845-862).
    The history of a block is just the bytes in front of it, so a large call still goes to the engine as one
    batch (nxz_deflate_host_hist: blocks of 64 KiB - history, each with its window): zlib reads the stream,
    matches reach across block boundaries (better ratio than the level-1 batch on text), and it is fast."""
    import time
    data = (open(os.path.join(os.path.dirname(__file__), "golden", "alice29.txt"), "rb").read() * 120)[:16 << 20]
    out, rcs, adler = Z.deflate_all(L, data, level=level, wbits=15)
    assert zlib.decompress(out) == data
    assert adler == zlib.adler32(data)
    assert len(rcs) <= 4, len(rcs)                               # one batch, not 256+ jobs
    out1, _, _ = Z.deflate_all(L, data, level=1, wbits=15)
    assert len(out) < len(out1)                                  # cross-block matches
    # the first block of a later call sees the tail of the call before (the stream's history)
    half = len(data) // 2 + 12345
    out2, _, _ = Z.deflate_all(L, data, level=level, wbits=15, step_in=half)
    assert zlib.decompress(out2) == data
    # host buffer to host buffer through the one-shot call (buffers made once: the call alone is timed)
    big = data * 4                                               # 64 MiB
    cap = C.c_ulong(L.nx_compressBound(len(big)))
    dst = C.create_string_buffer(cap.value)
    best = 1e9
    for _ in range(4):
        cap.value = len(dst)
        t = time.perf_counter()
        assert L.nx_compress2(dst, C.byref(cap), big, len(big), level) == Z.Z_OK
        best = min(best, time.perf_counter() - t)
    assert zlib.decompress(dst.raw[:cap.value]) == big
    rate = len(big) / best / 2 ** 30
    print("nx_compress2 level %d, 64 MiB host to host: %.2f GiB/s (%.2f ms), ratio %.3f" % (level, rate, best * 1e3, len(big) / cap.value))
    assert rate >= 15.0, "level %d: %.2f GiB/s host to host" % (level, rate)


@pytest.mark.parametrize("flush,tail", [(Z.Z_SYNC_FLUSH, b"\x00\x00\xff\xff"), (Z.Z_FULL_FLUSH, b"\x00\x00\xff\xff"), (Z.Z_PARTIAL_FLUSH, None)])
@pytest.mark.parametrize("kind", ["alice", "zeros", "mixed"])
def test_a_flush_request_is_honoured_when_the_batch_takes_all_the_input(L, flush, tail, kind):
    """Round-2 advisor finding: 4 x 64 KiB with Z_SYNC_FLUSH went to the batch, which consumed everything and
    returned without the flush rules of lib/nx_deflate.c:1081-1176.  The output of the flush call must end in the
    00 00 FF FF marker (sync / full), a partial flush in its empty fixed block; the stream goes on afterwards."""
    n = 4 * 65536
    data = mixed(n, 77) if kind == "mixed" else make_block(kind, n, 9)
    st = Z.ZStream()
    assert L.nx_deflateInit2_(C.byref(st), 1, Z.Z_DEFLATED, -15, 8, Z.Z_DEFAULT_STRATEGY, Z.VERSION, C.sizeof(Z.ZStream)) == Z.Z_OK
    src = C.create_string_buffer(data + data[:1000], n + 1000)
    dst = C.create_string_buffer(2 * n + 4096)
    st.next_in = C.addressof(src); st.avail_in = n
    st.next_out = C.addressof(dst); st.avail_out = len(dst)
    assert L.nx_deflate(C.byref(st), flush) == Z.Z_OK
    assert st.avail_in == 0
    first = dst.raw[:st.total_out]
    if tail is not None:
        assert first.endswith(tail), first[-8:]
    # everything fed so far is decodable from what has been written (that is what a flush promises)
    d = zlib.decompressobj(-15)
    assert d.decompress(first) == data
    # and the stream continues
    st.avail_in = 1000
    assert L.nx_deflate(C.byref(st), Z.Z_FINISH) == Z.Z_STREAM_END
    whole = dst.raw[:st.total_out]
    L.nx_deflateEnd(C.byref(st))
    d = zlib.decompressobj(-15)
    assert d.decompress(whole) == data + data[:1000] and d.eof


def test_dictionary_then_batch(L):
    data = mixed(1 << 20, 31)
    dic = data[:20000]
    out, _, _ = Z.deflate_all(L, data, level=-1, wbits=15, dictionary=dic)
    d = zlib.decompressobj(zdict=dic)
    assert d.decompress(out) == data


def test_sixteen_threads_share_the_engine(L):
    """T threads x one nx_compress2 / nx_uncompress call per 64 KiB buffer, all at once (the reference's
    samples/compdecomp_th.c): callers that arrive together go out as one launch of each kernel
    (nxz_engine.cpp round_submit); every thread must get its own bytes back."""
    import threading
    T, per = 16, 24
    bufs = [[make_block(("alice", "lz", "text33", "random", "zeros")[(t + i) % 5], 65536 - 17 * i, 100 * t + i) for i in range(per)] for t in range(T)]
    bad = []

    def worker(t):
        cap = C.c_ulong()
        dst = C.create_string_buffer(L.nx_compressBound(65536))
        back = C.create_string_buffer(65536)
        for b in bufs[t]:
            cap.value = len(dst)
            if L.nx_compress2(dst, C.byref(cap), b, len(b), 1) != Z.Z_OK or zlib.decompress(dst.raw[:cap.value]) != b:
                bad.append((t, "compress"))
                return
            n = C.c_ulong(65536)
            if L.nx_uncompress(back, C.byref(n), dst.raw[:cap.value], cap.value) != Z.Z_OK or back.raw[:n.value] != b:
                bad.append((t, "uncompress"))
                return

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not bad, bad


def test_threads_with_large_calls_work_side_by_side(L):
    """eight threads, each nx_compress2 / nx_uncompress of buffers of 1 - 3 MiB at once: the large-call paths
    (nxz_deflate_host on one of four pairs of lanes, nxz_inflate_stream_part on one of eight workspaces)
    run side by side; every thread must get its own bytes back, and zlib must read what was written"""
    import threading
    T, per = 8, 4
    kinds = ("alice", "lz", "text33", "random", "zeros", "binary")
    bufs = [[b"".join(make_block(kinds[(t + i + k) % 6], 65536, 1000 * t + 10 * i + k) for k in range(16 + 8 * ((t + i) % 5))) for i in range(per)] for t in range(T)]
    bad = []

    def worker(t):
        cap = C.c_ulong()
        dst = C.create_string_buffer(L.nx_compressBound(4 << 20))
        back = C.create_string_buffer(4 << 20)
        for b in bufs[t]:
            cap.value = len(dst)
            if L.nx_compress2(dst, C.byref(cap), b, len(b), 1) != Z.Z_OK or zlib.decompress(dst.raw[:cap.value]) != b:
                bad.append((t, "compress"))
                return
            z6 = zlib.compress(b, 6)
            for stream in (dst.raw[:cap.value], z6):
                n = C.c_ulong(len(back))
                if L.nx_uncompress(back, C.byref(n), stream, len(stream)) != Z.Z_OK or back.raw[:n.value] != b:
                    bad.append((t, "uncompress"))
                    return

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not bad, bad


def test_calls_of_a_few_blocks_from_many_threads_go_out_as_one_batch_and_stay_apart():
    """nxz_deflate_host calls of up to 32 blocks are merged: the callers that are there together share one launch of each
    kernel (nxz_engine.cpp merged_deflate).  Sixteen threads, calls of 1 .. 32 blocks and odd lengths, both function
    codes, final and not: every caller gets the stream of ITS bytes (zlib reads it, checksums match), and the stream is
    byte for byte what the same call makes alone on its own pair of lanes (NXZ_MERGE_MAX_BLOCKS=0)."""
    import importlib, threading
    pkg = importlib.import_module("power-gzip_amd")
    eng = pkg.Engine(0)
    kinds = ("alice", "lz", "text33", "random", "zeros", "binary")
    T, per = 16, 6
    rng = np.random.default_rng(5)
    sizes = [[int(rng.integers(1, 33)) * 65536 - int(rng.integers(0, 3)) * int(rng.integers(1, 60000)) for _ in range(per)] for _ in range(T)]
    bufs = [[b"".join(make_block(kinds[(t + i + k) % 6], 65536, 777 * t + 13 * i + k) for k in range((n + 65535) // 65536))[:n] for i, n in enumerate(sizes[t])] for t in range(T)]
    got = [[None] * per for _ in range(T)]
    bad = []

    def worker(t):
        for i, b in enumerate(bufs[t]):
            fc = pkg.FC_COMPRESS_DHTGEN if (t + i) % 3 else pkg.FC_COMPRESS_FHT
            final = (t + i) % 4 != 0
            rc, comp, crc, adler = eng.deflate_host(b, fc=fc, final=final)
            if rc != 0 or crc != zlib.crc32(b) or adler != zlib.adler32(b):
                bad.append((t, i, rc, "checksums"))
                return
            d = zlib.decompressobj(-15)
            if d.decompress(comp) != b or d.eof != final:
                bad.append((t, i, "zlib"))
                return
            got[t][i] = (fc, final, comp)

    try:
        th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not bad, bad
        os.environ["NXZ_MERGE_MAX_BLOCKS"] = "0"
        try:
            for t in range(0, T, 3):
                for i, b in enumerate(bufs[t]):
                    fc, final, comp = got[t][i]
                    rc, alone, _, _ = eng.deflate_host(b, fc=fc, final=final)
                    assert rc == 0 and alone == comp, (t, i, len(b))
        finally:
            del os.environ["NXZ_MERGE_MAX_BLOCKS"]
    finally:
        eng.close()


def test_merged_calls_with_a_window_match_the_calls_made_alone(L):
    """Levels 5..9 through the merged calls (blocks of 64 KiB - history with the input in front of them as window; the call's
    first block sees the tail of the call before): eight threads, each a stream of three 1 MiB steps at its own level, all at
    once.  zlib reads every stream, and every stream is byte for byte what the same steps make when no call is merged."""
    import threading
    T = 8
    kinds = ("alice", "lz", "text33", "binary")
    datas = [b"".join(make_block(kinds[(t + k) % 4], 65536, 31 * t + k) for k in range(40))[:(5 << 19) - 777 * t] for t in range(T)]
    levels = [5, 6, 7, 9, 6, 9, 5, 8]
    res = [None] * T
    bad = []

    def worker(t):
        try:
            out, _, adler = Z.deflate_all(L, datas[t], level=levels[t], wbits=15, step_in=1 << 20)
            if zlib.decompress(out) != datas[t] or adler != zlib.adler32(datas[t]):
                bad.append((t, "zlib"))
            res[t] = out
        except AssertionError as e:
            bad.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not bad, bad
    os.environ["NXZ_MERGE_MAX_BLOCKS"] = "0"
    try:
        for t in range(T):
            alone, _, _ = Z.deflate_all(L, datas[t], level=levels[t], wbits=15, step_in=1 << 20)
            assert alone == res[t], (t, levels[t])
    finally:
        del os.environ["NXZ_MERGE_MAX_BLOCKS"]


def test_calls_beyond_the_merge_limit_go_through_the_merges_in_slices(L):
    """nxz_deflate_host(_hist) calls of more than 128 blocks while other callers are about are cut into slices of 128 blocks that
    join the merges one after the other (nxz_engine.cpp): six threads at once, calls of 130 .. 300 blocks and odd lengths, both
    function codes, final and not, and two threads of level-6 / level-9 streams in steps of 12 MiB (blocks with the input in front
    of them as window).  zlib reads every stream, the checksums are the data's, and every stream is byte for byte what the same
    call makes on lanes of its own (NXZ_MERGE_SLICES=0)."""
    import importlib, threading
    pkg = importlib.import_module("power-gzip_amd")
    eng = pkg.Engine(0)
    kinds = ("alice", "lz", "text33", "random", "zeros", "binary")
    T, per = 6, 2
    nblocks = [[130, 257], [300, 129], [191, 256], [140, 222], [], []]
    cut = [[0, 12345], [1, 0], [40000, 7], [0, 0], [], []]
    bufs = [[b"".join(make_block(kinds[(t + i + k) % 6], 65536, 91 * t + 17 * i + k) for k in range(n))[:n * 65536 - cut[t][i]] for i, n in enumerate(nblocks[t])] for t in range(T)]
    streams = {4: (6, b"".join(make_block(kinds[k % 4], 65536, 4000 + k) for k in range(400))[:(25 << 20) - 4321]),
               5: (9, b"".join(make_block(kinds[(k + 2) % 4], 65536, 5000 + k) for k in range(330))[:(20 << 20) + 99])}
    got = [[None] * per for _ in range(T)]
    bad = []

    def worker(t):
        try:
            if t in streams:
                level, data = streams[t]
                out, _, adler = Z.deflate_all(L, data, level=level, wbits=15, step_in=12 << 20)
                if zlib.decompress(out) != data or adler != zlib.adler32(data):
                    bad.append((t, "zlib"))
                got[t][0] = out
                return
            for i, b in enumerate(bufs[t]):
                fc = pkg.FC_COMPRESS_DHTGEN if (t + i) % 2 else pkg.FC_COMPRESS_FHT
                final = (t + i) % 3 != 0
                rc, comp, crc, adler = eng.deflate_host(b, fc=fc, final=final)
                if rc != 0 or crc != zlib.crc32(b) or adler != zlib.adler32(b):
                    bad.append((t, i, rc, "checksums"))
                    return
                d = zlib.decompressobj(-15)
                if d.decompress(comp) != b or d.eof != final:
                    bad.append((t, i, "zlib"))
                    return
                got[t][i] = (fc, final, comp)
        except AssertionError as e:
            bad.append((t, repr(e)))

    try:
        th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not bad, bad
        os.environ["NXZ_MERGE_SLICES"] = "0"
        try:
            for t in range(T):
                if t in streams:
                    level, data = streams[t]
                    alone, _, _ = Z.deflate_all(L, data, level=level, wbits=15, step_in=12 << 20)
                    assert alone == got[t][0], (t, level)
                    continue
                for i, b in enumerate(bufs[t]):
                    fc, final, comp = got[t][i]
                    rc, alone, _, _ = eng.deflate_host(b, fc=fc, final=final)
                    assert rc == 0 and alone == comp, (t, i, len(b))
        finally:
            del os.environ["NXZ_MERGE_SLICES"]
    finally:
        eng.close()


def test_deflate_host_entry_point():
    """nxz_deflate_host itself (include/nxz_engine.h): any length, final or not, both function codes;
    runs that are not final end on a byte boundary and are continued by the next run."""
    import importlib
    pkg = importlib.import_module("power-gzip_amd")
    eng = pkg.Engine(0)
    try:
        data = mixed((2 << 20) + 4711, 41)
        for fc in (pkg.FC_COMPRESS_FHT, pkg.FC_COMPRESS_DHTGEN):
            rc, comp, crc, adler = eng.deflate_host(data, fc=fc, final=True)
            assert rc == 0 and zlib.decompress(comp, -15) == data
            assert crc == zlib.crc32(data) and adler == zlib.adler32(data)
        # two runs make one stream
        a, b = data[:1 << 20], data[1 << 20:]
        rc1, c1, crc1, _ = eng.deflate_host(a, final=False)
        rc2, c2, crc2, _ = eng.deflate_host(b, final=True)
        assert rc1 == 0 and rc2 == 0
        d = zlib.decompressobj(-15)
        assert d.decompress(c1 + c2) == data and d.eof
        assert crc1 == zlib.crc32(a) and crc2 == zlib.crc32(b)
        # small inputs: a single short block; refusals
        rc, comp, _, _ = eng.deflate_host(b"x" * 100)
        assert rc == 0 and zlib.decompress(comp, -15) == b"x" * 100
        assert eng.deflate_host(data, cap=1000)[0] == -7                  # E2BIG: the bound is what it asks for
        assert eng.deflate_host(data, fc=pkg.FC_COMPRESS_DHT)[0] == -22   # EINVAL: a caller's table makes no sense here
    finally:
        eng.close()
