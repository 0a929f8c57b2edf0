"""PCIe-inclusive rate of the batched path: blocks start and end in pinned HOST memory; chunks of
blocks go host -> device, through the deflate kernel, and back, on three streams so that the copies
overlap the kernels.  (The bench `value` is the device-resident rate; this is the note DESIGN.md owes.)
usage: python tools/bench_host.py [blocks] [chunk]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
pkg = importlib.import_module("power-gzip_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
eng = pkg.Engine(0)
dev = eng.dev
STR = 73856
h_src = torch.empty((n, 65536), dtype=torch.uint8).pin_memory()
h_src.copy_(bench.gen_blocks(torch, dev, n, 0).cpu())
h_dst = torch.empty((n, STR), dtype=torch.uint8).pin_memory()
h_res = torch.empty((n, pkg.RESULT_DTYPE.itemsize), dtype=torch.uint8).pin_memory()
NS = 3
streams = [torch.cuda.Stream() for _ in range(NS)]
d_src = [torch.empty((chunk, 65536), dtype=torch.uint8, device=dev) for _ in range(NS)]
d_dst = [torch.empty((chunk, STR), dtype=torch.uint8, device=dev) for _ in range(NS)]
d_res = [torch.empty(chunk * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(NS)]
jobs = [eng.jobs_strided(d_src[k], 65536, np.full(chunk, 65536, np.uint32), d_dst[k], STR, STR) for k in range(NS)]

def run():
    for c, lo in enumerate(range(0, n, chunk)):
        k = c % NS
        m = min(chunk, n - lo)
        with torch.cuda.stream(streams[k]):
            d_src[k][:m].copy_(h_src[lo:lo + m], non_blocking=True)
            eng.compress(pkg.FC_COMPRESS_FHT, jobs[k], m, results=d_res[k])
            h_res[lo:lo + m].copy_(d_res[k].view(chunk, -1)[:m], non_blocking=True)
            h_dst[lo:lo + m].copy_(d_dst[k][:m], non_blocking=True)           # whole output slots (73 856 B each)
    torch.cuda.synchronize()

run()
t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
r = h_res.numpy().view(pkg.RESULT_DTYPE).reshape(-1)
assert (r["cc"] == 0).all()
print("%d blocks from/to pinned host memory, %d-block chunks on %d streams, whole output slots copied back: "
      "%.1f GiB/s of input (%.1f GB/s over PCIe in, %.1f GB/s out)"
      % (n, chunk, NS, n * 65536 / dt / 2**30, n * 65536 / dt / 1e9, n * STR / dt / 1e9))
