"""Wave-kernel inflate time of 64 KiB pieces of a FILE (zlib -6 raw streams), a stream per wave:
window in LDS against the target as window.  usage: bench_inflate_file.py <file> [offset]"""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import importlib, os, sys, time, zlib
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
data = open(sys.argv[2], "rb").read()
off = int(sys.argv[3])
n = 256
blocks = [data[off + i * 65536: off + (i + 1) * 65536] for i in range(16)]
blocks = [b for b in blocks if len(b) == 65536]
cs = []
for b in blocks:
    z = zlib.compressobj(6, zlib.DEFLATED, -15); cs.append(z.compress(b) + z.flush())
stride = (max(map(len, cs)) + 31) & ~15
buf = np.zeros((n, stride), np.uint8)
lens = np.zeros(n, np.uint32)
for i in range(n):
    c = cs[i % len(cs)]; buf[i, :len(c)] = np.frombuffer(c, np.uint8); lens[i] = len(c)
src = torch.from_numpy(buf).to(eng.dev)
back = torch.empty((n, 65536), dtype=torch.uint8, device=eng.dev)
j = eng.jobs_strided(src, stride, lens, back, 65536, 65536)
res = torch.empty(n * 32, dtype=torch.uint8, device=eng.dev)
eng.decompress(j, n, results=res); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): eng.decompress(j, n, results=res)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
assert back[0].cpu().numpy().tobytes() == blocks[0]
print("  %s: %.2f ms per round of %d streams of 64 KiB (mean compressed %d bytes)" % (sys.argv[4], dt * 1e3, n, int(lens.mean())), flush=True)
'''
f = sys.argv[1]
off = sys.argv[2] if len(sys.argv) > 2 else "0"
for label, lds in (("window in LDS", "1000000000"), ("target as window", "0")):
    subprocess.run([sys.executable, "-c", CHILD, ROOT, f, off, label], env=dict(os.environ, NXZ_INFLATE_LANES_MIN="1000000000", NXZ_INFLATE_LDS_MAX=lds), check=True)
