"""inflate() in steps (avail_in / avail_out of 64 KiB - 1 MiB and more) over a zlib -6 .gz of the
corpus: GiB/s of output per step size, nx_inflate on the GPU engine against system zlib's inflate on one
host thread.  usage: bench_inflate_steps.py [MiB of plain data, default 64]"""
import ctypes as C
import importlib
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus
import zstream as Z

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
_, blocks, _ = corpus.load(65536)
only = [c for c in os.environ.get("CLASSES", "").split(",") if c]                  # CLASSES=text,xml: those classes of the corpus only
raw = b"".join(b for cls, _, b in blocks if not only or cls in only)
plain = (raw * (1 + (mib << 20) // len(raw)))[:mib << 20]
co = zlib.compressobj(6, zlib.DEFLATED, 31)
gz = co.compress(plain) + co.flush()
L = Z.load("gpu")
print("%d MiB of the corpus, zlib -6 .gz of %d bytes" % (mib, len(gz)))
for step in (64 << 10, 256 << 10, 1 << 20, 4 << 20, 16 << 20):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        got, rc, total_in, _ = Z.inflate_all(L, gz, wbits=31, cap=len(plain) + 64, step_in=step, step_out=step)
        dt = time.perf_counter() - t0
        assert rc == Z.Z_STREAM_END and len(got) == len(plain)
        best = min(best, dt)
    assert got == plain
    d = zlib.decompressobj(31)
    t0 = time.perf_counter()
    n = 0
    for o in range(0, len(gz), step):
        n += len(d.decompress(gz[o:o + step]))
    tz = time.perf_counter() - t0
    print("  steps of %6d KiB in and out: nx_inflate %8.1f ms = %6.3f GiB/s    zlib, one thread %8.1f ms = %6.3f GiB/s" %
          (step >> 10, best * 1e3, len(plain) / best / 2**30, tz * 1e3, len(plain) / tz / 2**30), flush=True)
