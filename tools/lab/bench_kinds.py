"""Deflate kernel throughput by kind of data (batches of identical-kind 64 KiB blocks)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from datagen import make_block
pkg = importlib.import_module("power-gzip_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
eng = pkg.Engine(0)
dst = torch.empty((n, 73856), dtype=torch.uint8, device=eng.dev)
for kind in ("zeros", "random", "text33", "alice", "lz", "periodic", "binary", "sparse"):
    m = 64
    host = np.stack([np.frombuffer(make_block(kind, 65536, s), np.uint8) for s in range(m)])
    src = torch.from_numpy(host).to(eng.dev).repeat((n + m - 1) // m, 1)[:n].contiguous()
    jobs = eng.jobs_strided(src, 65536, np.full(n, 65536, np.uint32), dst, 73856, 73856)
    res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
    eng.compress(pkg.FC_COMPRESS_FHT, jobs, n, results=res); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.compress(pkg.FC_COMPRESS_FHT, jobs, n, results=res)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    r = eng.results_to_host(res)
    okb = r["cc"] == 0
    line = "%-9s deflate %7.1f GiB/s   ratio %6.2f   cc=0 for %d of %d" % (kind, n * 65536 / dt / 2**30, (okb.sum() * 65536) / max(1, int(r["tpbc"][okb].sum())), okb.sum(), n)
    if okb.all():
        back = torch.empty((n, 65536), dtype=torch.uint8, device=eng.dev)
        jd = eng.jobs_strided(dst, 73856, r["tpbc"].astype(np.uint32), back, 65536, 65536)
        rd = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
        eng.decompress(jd, n, results=rd); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): eng.decompress(jd, n, results=rd)
        torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / 3
        assert torch.equal(back, src)
        line += "   inflate %7.1f GiB/s (round trip ok)" % (n * 65536 / dt2 / 2**30)
    print(line)
