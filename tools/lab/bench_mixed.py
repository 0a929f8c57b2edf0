"""BASELINE configs[4] shape on one GPU: a batch of 64 KiB blocks of mixed entropy (zeros, text,
makedata-style LZ copies, English-like text, random bytes), compressed and decompressed again.
Incompressible blocks complete with CC=64 (tpbc > spbc, the engine's own answer); they are re-run as
stored copies with the WRAP function code -- what the library does per job (lib/nx_deflate.c:1274-1282,
1763-1790) done here for the whole batch.  Every block is checked bit for bit after the round trip.
usage: python tools/bench_mixed.py [blocks]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from datagen import make_block
pkg = importlib.import_module("power-gzip_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = pkg.Engine(0)
src = bench.gen_blocks(torch, eng.dev, n, 0)                       # text + LZ copies
g = torch.Generator(device=eng.dev); g.manual_seed(12345)
kind = torch.randint(0, 5, (n,), device=eng.dev, generator=g)      # 0,1: keep  2: zeros  3: random  4: English-like
src[kind == 2] = 0
nr = int((kind == 3).sum())
src[kind == 3] = torch.randint(0, 256, (nr, 65536), device=eng.dev, dtype=torch.uint8, generator=g)
alice = torch.from_numpy(np.stack([np.frombuffer(make_block("alice", 65536, s), np.uint8) for s in range(64)])).to(eng.dev)
idx = torch.nonzero(kind == 4).flatten()
src[idx] = alice[torch.arange(len(idx), device=eng.dev) % 64]
STR = 73856
comp = torch.empty((n, STR), dtype=torch.uint8, device=eng.dev)
lens = np.full(n, 65536, np.uint32)
jobs = eng.jobs_strided(src, 65536, lens, comp, STR, STR)
res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)

def compress_all():
    eng.compress(pkg.FC_COMPRESS_FHT, jobs, n, results=res)
    r = eng.results_to_host(res)
    bad = np.nonzero(r["cc"] == 64)[0]
    if len(bad):                                                   # stored copies for what did not shrink
        bsrc = src[torch.from_numpy(bad).to(eng.dev)]
        bdst = torch.empty((len(bad), 65536), dtype=torch.uint8, device=eng.dev)
        jb = eng.jobs_strided(bsrc, 65536, np.full(len(bad), 65536, np.uint32), bdst, 65536, 65536)
        rw = eng.results_to_host(eng.wrap(jb, len(bad)))
        assert (rw["cc"] == 0).all()
        return r, bad, bdst, rw
    return r, bad, None, None

compress_all(); torch.cuda.synchronize()
t0 = time.perf_counter(); r, bad, stored, rw = compress_all(); torch.cuda.synchronize(); tc = time.perf_counter() - t0
ok = np.nonzero(r["cc"] == 0)[0]
assert len(ok) + len(bad) == n, np.unique(r["cc"])
# decompress the deflate blocks; the stored ones are plain copies already
back = torch.empty((n, 65536), dtype=torch.uint8, device=eng.dev)
ok_t = torch.from_numpy(ok).to(eng.dev)
csel = comp[ok_t]
bsel = torch.empty((len(ok), 65536), dtype=torch.uint8, device=eng.dev)
jd = eng.jobs_strided(csel, STR, r["tpbc"][ok].astype(np.uint32), bsel, 65536, 65536)
rd = torch.empty(len(ok) * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
eng.decompress(jd, len(ok), results=rd); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.decompress(jd, len(ok), results=rd); torch.cuda.synchronize(); td = time.perf_counter() - t0
rdh = eng.results_to_host(rd)
assert (rdh["tpbc"] == 65536).all() and (rdh["crc"] == r["crc"][ok]).all()
back[ok_t] = bsel
if len(bad):
    back[torch.from_numpy(bad).to(eng.dev)] = stored
    assert (rw["crc"] == r["crc"][bad]).all()
assert torch.equal(back, src)
out_bytes = int(r["tpbc"][ok].sum()) + len(bad) * (65536 + 5)
gib = n * 65536 / 2**30
print("%d blocks (%.1f GiB): %d deflate blocks + %d stored; compress %.1f GiB/s, decompress %.1f GiB/s (of the deflate part), ratio %.3f; round trip bit exact"
      % (n, gib, len(ok), len(bad), gib / tc, len(ok) * 65536 / 2**30 / td, n * 65536 / out_bytes))
