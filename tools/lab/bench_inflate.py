"""Throughput of the inflate engine on the deflate engine's own output (and on zlib level-6 streams)."""
import importlib, os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
pkg = importlib.import_module("power-gzip_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
eng = pkg.Engine(0)
src = bench.gen_blocks(torch, eng.dev, n, 0)
comp = torch.empty((n, 73856), dtype=torch.uint8, device=eng.dev)
jobs = eng.jobs_strided(src, 65536, np.full(n, 65536, np.uint32), comp, 73856, 73856)
r = eng.results_to_host(eng.compress(pkg.FC_COMPRESS_FHT, jobs, n)[0])
back = torch.empty((n, 65536), dtype=torch.uint8, device=eng.dev)
jobs2 = eng.jobs_strided(comp, 73856, r["tpbc"].astype(np.uint32), back, 65536, 65536)
res = torch.empty(n * 32, dtype=torch.uint8, device=eng.dev)
eng.decompress(jobs2, n, results=res); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng.decompress(jobs2, n, results=res)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
assert torch.equal(back, src)
print("inflate of own FHT output: %.2f GiB/s out (%d blocks, %.1f ms)" % (n * 65536 / dt / 2**30, n, dt * 1e3))
# zlib level 6 dynamic streams of the same blocks (first 2048 blocks compressed on the host)
m = min(n, 2048)
host = src[:m].cpu().numpy()
zs = [zlib.compressobj(6, zlib.DEFLATED, -15) for _ in range(m)]
cs = [z.compress(host[i].tobytes()) + z.flush() for i, z in enumerate(zs)]
stride = (max(map(len, cs)) + 31) & ~15
buf = np.zeros((m, stride), np.uint8)
for i, c in enumerate(cs): buf[i, :len(c)] = np.frombuffer(c, np.uint8)
zc = torch.from_numpy(buf).to(eng.dev)
reps = max(1, n // m)
zc = zc.repeat(reps, 1); lens = np.tile(np.array([len(c) for c in cs], np.uint32), reps)
k = m * reps
back2 = torch.empty((k, 65536), dtype=torch.uint8, device=eng.dev)
jobs3 = eng.jobs_strided(zc, stride, lens, back2, 65536, 65536)
res3 = torch.empty(k * 32, dtype=torch.uint8, device=eng.dev)
eng.decompress(jobs3, k, results=res3); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng.decompress(jobs3, k, results=res3)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
assert torch.equal(back2[:m], src[:m])
print("inflate of zlib -6 streams:  %.2f GiB/s out (%d blocks, %.1f ms, ratio %.2f)" % (k * 65536 / dt / 2**30, k, dt * 1e3, m * 65536 / sum(map(len, cs))))
# the engine's own dynamic-Huffman output: one table per 64 consecutive blocks (what bench.py's dht leg and
# the blocked gzip layer write) -- the lanes of a wave then share one table
import ctypes as C
H = C.CDLL(os.path.join(ROOT, "power-gzip_amd", "libnxz_amd.so"))
H.nxz_dhtgen_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
G = 64
ng = (n + G - 1) // G
lead = eng.jobs_strided(src, 65536 * G, np.full(ng, 65536, np.uint32), comp, 73856 * G, 73856)
cnt = torch.empty(ng * 316, dtype=torch.int32, device=eng.dev)
eng.compress(pkg.FC_COMPRESS_FHT_COUNT, lead, ng, counts=cnt)
tabs = np.zeros(ng, pkg.DHT_DTYPE)
c = cnt.cpu().numpy().view(np.uint32)
assert H.nxz_dhtgen_batch(c.ctypes.data, ng, tabs.ctypes.data, bench.usable_cores()) == 0
jd = eng.jobs_strided(src, 65536, np.full(n, 65536, np.uint32), comp, 73856, 73856, dht_index=(np.arange(n) // G).astype(np.uint32))
rd = eng.results_to_host(eng.compress(pkg.FC_COMPRESS_DHT, jd, n, dht=eng.to_device(tabs), ntables=ng)[0])
assert (rd["cc"] == 0).all()
back.zero_()
jobs4 = eng.jobs_strided(comp, 73856, rd["tpbc"].astype(np.uint32), back, 65536, 65536)
eng.decompress(jobs4, n, results=res); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng.decompress(jobs4, n, results=res)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
assert torch.equal(back, src)
print("inflate of own dynamic output (a table per 64 blocks): %.2f GiB/s out (%d blocks, %.1f ms, ratio %.2f)"
      % (n * 65536 / dt / 2**30, n, dt * 1e3, n * 65536 / float(rd["tpbc"].astype(np.float64).sum())))
