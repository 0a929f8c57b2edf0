"""Soak run (not collected by pytest): many threads, each a long series of nxz_deflate_host / nx_compress2 / nx_deflate calls of
random sizes (one block to a few hundred), function codes, levels (with and without history) and `final` flags, all at once --
the merged calls of nxz_engine.cpp (merged_deflate) with callers joining and leaving merges at every moment, next to calls
too large to merge on their own pairs of lanes.  Every stream is read back by zlib and its checksums compared.
python tests/soak_merged_gpu.py [seconds] [threads]"""
import ctypes as C, importlib, os, random, sys, threading, time, zlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from datagen import make_block
import zstream as Z
L = Z.load("gpu")
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
kinds = ("alice", "lz", "text33", "random", "zeros", "binary", "periodic", "sparse")
pool = [make_block(k, 65536, seed=100 + i) for i, k in enumerate(kinds * 4)]
bad, calls, nbytes = [], [0] * T, [0] * T
stop = time.time() + seconds


def data_of(rnd, n):
    out = b"".join(rnd.choice(pool) for _ in range((n + 65535) // 65536))
    return out[:n]


def worker(t):
    rnd = random.Random(7 * t + 1)
    dst = C.create_string_buffer(L.nx_compressBound(24 << 20))
    while time.time() < stop and not bad:
        how = rnd.random()
        n = rnd.choice([1, 100, 65535, 65536, 65537]) if rnd.random() < 0.1 else rnd.randrange(1, 64 * 65536) if rnd.random() < 0.85 else rnd.randrange(64 * 65536, 300 * 65536)
        d = data_of(rnd, n)
        try:
            if how < 0.4:
                fc = pkg.FC_COMPRESS_DHTGEN if rnd.random() < 0.7 else pkg.FC_COMPRESS_FHT
                final = rnd.random() < 0.7
                rc, comp, crc, adler = eng.deflate_host(d, fc=fc, final=final)
                o = zlib.decompressobj(-15)
                ok = rc == 0 and o.decompress(comp) == d and o.eof == final and crc == zlib.crc32(d) and adler == zlib.adler32(d)
            elif how < 0.7:
                cap = C.c_ulong(len(dst))
                level = rnd.choice([1, 1, 3, 5, 6, 9])
                ok = L.nx_compress2(dst, C.byref(cap), d, len(d), level) == Z.Z_OK and zlib.decompress(dst.raw[:cap.value]) == d
            else:
                level = rnd.choice([1, 4, 5, 6, 7, 9])
                step = rnd.choice([None, 1 << 20, 300000, 65536 * 3])
                out, _, adler = Z.deflate_all(L, d, level=level, wbits=15, step_in=step)
                ok = zlib.decompress(out) == d and adler == zlib.adler32(d)
        except AssertionError as e:
            ok = False
            bad.append((t, "assert", repr(e)[:200]))
        if not ok:
            bad.append((t, how, n))
            return
        calls[t] += 1; nbytes[t] += n


th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
t0 = time.time()
for x in th: x.start()
for x in th: x.join()
dt = time.time() - t0
print("soak merged: %d threads, %.0f s, %d calls, %.1f GiB in: %s" % (T, dt, sum(calls), sum(nbytes) / 2.0 ** 30, "BAD %r" % bad[:3] if bad else "SOAK OK"))
sys.exit(1 if bad else 0)
