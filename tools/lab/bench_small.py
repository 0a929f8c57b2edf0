import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from datagen import make_block
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
for size in (1024, 4096, 16384, 65536):
    n = 262144
    m = 64
    host = np.stack([np.frombuffer(make_block("alice", size, s), np.uint8) for s in range(m)])
    src = torch.from_numpy(host).to(eng.dev).repeat(n // m, 1).contiguous()
    so = size + size // 8 + 128
    dst = torch.empty((n, so), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, size, np.full(n, size, np.uint32), dst, so, so)
    res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
    eng.compress(pkg.FC_COMPRESS_FHT, jobs, n, results=res); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.compress(pkg.FC_COMPRESS_FHT, jobs, n, results=res)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("block %6d B: %6.1f GiB/s  (%.2f us per job per CU)" % (size, n * size / dt / 2**30, dt / n * 256 * 1e6))
