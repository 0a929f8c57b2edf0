"""Batched inflate rate against the number of streams in the batch, for both kernels: a stream per
lane (nxz_inflate_lanes.hip) and a stream per wave (nxz_inflate.hip).  The engine switches between
them at NXZ_LANES_MIN streams (nxz_engine.cpp; NXZ_INFLATE_LANES_MIN overrides it for this sweep).
usage: python tools/bench_inflate_sizes.py   (spawns one child per kernel)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import importlib, os, sys, time, zlib
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, bench
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
N = int(os.environ.get('NXZ_BENCH_N', '65536'))
src = bench.gen_blocks(torch, eng.dev, N, 0)
comp = torch.empty((N, 73856), dtype=torch.uint8, device=eng.dev)
jobs = eng.jobs_strided(src, 65536, np.full(N, 65536, np.uint32), comp, 73856, 73856)
r = eng.results_to_host(eng.compress(pkg.FC_COMPRESS_FHT, jobs, N)[0])
m = 1024
host = src[:m].cpu().numpy()
cs = []
for i in range(m):
    z = zlib.compressobj(6, zlib.DEFLATED, -15); cs.append(z.compress(host[i].tobytes()) + z.flush())
stride = (max(map(len, cs)) + 31) & ~15
buf = np.zeros((m, stride), np.uint8)
for i, c in enumerate(cs): buf[i, :len(c)] = np.frombuffer(c, np.uint8)
zc = torch.from_numpy(buf).to(eng.dev).repeat(N // m, 1)
zl = np.tile(np.array([len(c) for c in cs], np.uint32), N // m)
back = torch.empty((N, 65536), dtype=torch.uint8, device=eng.dev)
for n in [k for k in (32, 128, 512, 1024, 2048, 4096, 8192, 16384, 65536, 262144) if k <= N]:
    out = []
    for name, s, st, ln in (("own fixed-Huffman output", comp, 73856, r["tpbc"].astype(np.uint32)), ("zlib -6 streams", zc, stride, zl)):
        j = eng.jobs_strided(s, st, ln[:n], back, 65536, 65536)
        res = torch.empty(n * 32, dtype=torch.uint8, device=eng.dev)
        eng.decompress(j, n, results=res); torch.cuda.synchronize()
        reps = 3 if n >= 4096 else 10
        t0 = time.perf_counter()
        for _ in range(reps): eng.decompress(j, n, results=res)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        assert torch.equal(back[:min(n, m)], src[:min(n, m)])
        out.append("%s %7.2f GiB/s (%7.2f ms)" % (name, n * 65536 / dt / 2**30, dt * 1e3))
    print("%6d streams: %s" % (n, "   ".join(out)), flush=True)
'''
for label, thr, lds in (("a stream per lane", "1", "0"), ("a stream per wave, window in LDS", "1000000000", "1000000000"),
                        ("a stream per wave, the target as window", "1000000000", "0")):
    if len(sys.argv) > 1 and sys.argv[1] not in label:
        continue
    print("--- " + label, flush=True)
    subprocess.run([sys.executable, "-c", CHILD, ROOT], env=dict(os.environ, NXZ_INFLATE_LANES_MIN=thr, NXZ_INFLATE_LDS_MAX=lds), check=True)
