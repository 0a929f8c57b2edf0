"""Concurrency, after the reference's test/test_stress.c (threads released together, 64 KiB each) and
test/test_multithread_stress.c (random buffer sizes 4 KiB .. 1 MiB, compress + uncompress): many
threads share one engine through the one-shot API; every result must round-trip through zlib and
through the library itself.  ctypes drops the GIL during the calls, so the threads really overlap
inside the engine (job slots, streams, staging buffers)."""
import ctypes as C
import threading
import zlib

import pytest

import zstream as Z
from datagen import make_block

SIZES = [4096, 16384, 65536, 65537, 262144, 1048576, 70000, 5000, 131072, 33333]


@pytest.fixture(params=["model", pytest.param("gpu", marks=pytest.mark.gpu)])
def L(request):
    return Z.load(request.param)


def _worker(L, tid, rounds, barrier, errors):
    try:
        barrier.wait()
        for r in range(rounds):
            n = SIZES[(tid * 7 + r * 3) % len(SIZES)]
            kind = ("alice", "lz", "text33", "random", "zeros")[(tid + r) % 5]
            data = make_block(kind, n, seed=tid * 100 + r)
            bound = L.nx_compressBound(n)
            out = C.create_string_buffer(bound)
            olen = C.c_ulong(bound)
            rc = L.nx_compress2(out, C.byref(olen), data, n, 6)
            assert rc == Z.Z_OK, ("compress", tid, r, rc)
            comp = out.raw[:olen.value]
            assert zlib.decompress(comp) == data, ("zlib inflate", tid, r)
            back = C.create_string_buffer(n + 16)
            blen = C.c_ulong(n + 16)
            rc = L.nx_uncompress(back, C.byref(blen), comp, len(comp))
            assert rc == Z.Z_OK and blen.value == n and back.raw[:n] == data, ("uncompress", tid, r, rc)
            # and a zlib-made stream through the library
            zc = zlib.compress(data, 6)
            blen = C.c_ulong(n + 16)
            rc = L.nx_uncompress(back, C.byref(blen), zc, len(zc))
            assert rc == Z.Z_OK and back.raw[:blen.value] == data, ("uncompress zlib", tid, r, rc)
    except BaseException as e:      # noqa: BLE001 - collected and re-raised in the main thread
        errors.append(e)


@pytest.mark.parametrize("nthreads,rounds", [(16, 6), (60, 2)])
def test_threads_share_the_engine(L, nthreads, rounds):
    errors = []
    barrier = threading.Barrier(nthreads)
    ts = [threading.Thread(target=_worker, args=(L, i, rounds, barrier, errors)) for i in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
        assert not t.is_alive(), "worker hung"
    assert not errors, errors[:3]
