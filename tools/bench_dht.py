"""Dynamic-Huffman pipeline on device-resident 64 KiB blocks (BASELINE configs[2] shape):
   pass 1  COMPRESS_FHT_COUNT  -> LZ77 symbol counts per block (the output bytes are not used)
   host    nxz_dhtgen_batch    -> one table per block (libnxz_amd.so, host threads)
   pass 2  COMPRESS_DHT        -> dynamic-Huffman blocks
Reports the rate of each stage, the end-to-end rate, the ratio next to zlib -1 (default strategy)
on a sample, and checks that sampled outputs inflate to the input with zlib.
usage: python tools/bench_dht.py [blocks] [alice|synthetic]"""
import ctypes as C, importlib, os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
pkg = importlib.import_module("power-gzip_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kind = sys.argv[2] if len(sys.argv) > 2 else "synthetic"
eng = pkg.Engine(0)
src = bench.gen_blocks(torch, eng.dev, n, 0)
if kind == "alice":
    from datagen import make_block
    m = 256
    host = np.stack([np.frombuffer(make_block("alice", 65536, s), np.uint8) for s in range(m)])
    src[:] = torch.from_numpy(host).to(eng.dev).repeat((n + m - 1) // m, 1)[:n]
dst = torch.empty((n, 73856), dtype=torch.uint8, device=eng.dev)
lens = np.full(n, 65536, np.uint32)
jobs = eng.jobs_strided(src, 65536, lens, dst, 73856, 73856, dht_index=np.arange(n, dtype=np.uint32))
counts = torch.empty(n * 316, dtype=torch.int32, device=eng.dev)
H = C.CDLL(os.path.join(ROOT, "power-gzip_amd", "libnxz_amd.so"))
H.nxz_dhtgen_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
nthreads = bench.usable_cores()

def timed(f, reps=2):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

t1 = timed(lambda: eng.compress(pkg.FC_COMPRESS_FHT_COUNT, jobs, n, counts=counts))
cnt = counts.cpu().numpy().view(np.uint32)
tables = np.zeros(n, pkg.DHT_DTYPE)
t0 = time.perf_counter()
assert H.nxz_dhtgen_batch(cnt.ctypes.data, n, tables.ctypes.data, nthreads) == 0
th = time.perf_counter() - t0
dht = eng.to_device(tables)
res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
t2 = timed(lambda: eng.compress(pkg.FC_COMPRESS_DHT, jobs, n, results=res, dht=dht, ntables=n))
r = eng.results_to_host(res)
assert (r["cc"] == 0).all(), np.unique(r["cc"])
gib = n * 65536 / 2**30
csize = int(r["tpbc"].sum())
print("blocks %d (%s)  pass1 counts %.1f GiB/s  host tables %.0f /s on %d threads (%.2f GiB/s of input)  pass2 encode %.1f GiB/s"
      % (n, kind, gib / t1, n / th, nthreads, gib / th, gib / t2))
print("end to end (no overlap) %.1f GiB/s   GPU stages only %.1f GiB/s" % (gib / (t1 + th + t2), gib / (t1 + t2)))
m = min(n, 256)
hs = src[:m].cpu().numpy(); out = dst[:m].cpu().numpy()
z1 = 0
for i in range(m):
    dz = zlib.decompressobj(-15)
    assert dz.decompress(out[i, :r["tpbc"][i]].tobytes()) == hs[i].tobytes() and dz.eof, i
    c = zlib.compressobj(1, zlib.DEFLATED, -15); z1 += len(c.compress(hs[i].tobytes()) + c.flush())
ours = int(r["tpbc"][:m].sum())
print("ratio %.4f (whole batch)   sample of %d: ours %.4f  zlib -1 %.4f  -> %.3f x zlib -1;  all sampled blocks inflate with zlib"
      % (n * 65536 / csize, m, m * 65536 / ours, m * 65536 / z1, z1 / ours))

# ---- one pass: a table per group of G consecutive blocks, made from the counts of the group's
# first block (zero counts raised to 1, so every symbol has a code) -- the reference's policy of
# reusing a table over neighbouring data (lib/nx_dht.c:568-676), batched
G = 64
ng = (n + G - 1) // G
lead = np.arange(0, n, G)
jobs_lead = eng.jobs_strided(src, 65536 * G, np.full(ng, 65536, np.uint32), dst, 73856 * G, 73856)
cl = torch.empty(ng * 316, dtype=torch.int32, device=eng.dev)
gt = np.zeros(ng, pkg.DHT_DTYPE)
def one_pass():
    eng.compress(pkg.FC_COMPRESS_FHT_COUNT, jobs_lead, ng, counts=cl)
    c = cl.cpu().numpy().view(np.uint32)
    assert H.nxz_dhtgen_batch(c.ctypes.data, ng, gt.ctypes.data, nthreads) == 0
    d = eng.to_device(gt)
    eng.compress(pkg.FC_COMPRESS_DHT, jobs_g, n, results=res, dht=d, ntables=ng)
jobs_g = eng.jobs_strided(src, 65536, lens, dst, 73856, 73856, dht_index=(np.arange(n) // G).astype(np.uint32))
t3 = timed(one_pass)
r = eng.results_to_host(res)
assert (r["cc"] == 0).all(), np.unique(r["cc"])
out = dst[:m].cpu().numpy()
for i in range(m):
    dz = zlib.decompressobj(-15)
    assert dz.decompress(out[i, :r["tpbc"][i]].tobytes()) == hs[i].tobytes() and dz.eof, i
print("one pass, table per %d blocks from the first block's counts: %.1f GiB/s end to end, ratio %.4f (%.3f x zlib -1 on the sample)"
      % (G, gib / t3, n * 65536 / int(r["tpbc"].sum()), z1 / int(r["tpbc"][:m].sum())))
