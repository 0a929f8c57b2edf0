"""nx_uncompress of ONE zlib -6 stream by size of the data (host buffers): where the parallel path pays.
usage: python tools/bench_uncompress_sizes.py"""
import ctypes as C, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import zstream as Z
import corpus
L = Z.load("gpu")
base = b"".join(b for _, _, b in corpus.load()[1])
for kib in (128, 256, 512, 1024, 2048, 4096, 16384):
    data = base[1 << 20:(1 << 20) + (kib << 10)] if (kib << 10) + (1 << 20) <= len(base) else (base * 2)[:kib << 10]
    z6 = zlib.compress(data, 6)
    back = C.create_string_buffer(len(data))
    best = 1e9
    for it in range(4):
        n = C.c_ulong(len(data))
        t = time.perf_counter()
        rc = L.nx_uncompress(back, C.byref(n), z6, len(z6))
        dt = time.perf_counter() - t
        assert rc == 0 and n.value == len(data) and back.raw == data, (kib, rc)
        if it:
            best = min(best, dt)
    t = time.perf_counter(); zlib.decompress(z6); tz = time.perf_counter() - t
    print("%6d KiB (%7d compressed): nx_uncompress %8.2f ms = %7.3f GiB/s   zlib one thread %6.2f ms" % (kib, len(z6), best * 1e3, len(data) / best / 2**30, tz * 1e3), flush=True)
