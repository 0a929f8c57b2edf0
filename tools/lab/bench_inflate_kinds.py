import ctypes as C, importlib, os, sys, time, zlib
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, bench
from datagen import make_block
os.environ["NXZ_INFLATE_LANES_MIN"] = "1000000000"
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
n = 512
KINDS = [("zeros", 6), ("text33", 6), ("text33", 0), ("alice", 6), ("alice", 1), ("lz", 6), ("random", 6)]
if len(sys.argv) > 1:            # a file: its first 64 KiB (or at the offset given) at zlib -6
    KINDS = [(sys.argv[1], 6)]
for kind, level in KINDS:
    d = open(kind, "rb").read()[int(sys.argv[2]) if len(sys.argv) > 2 else 0:][:65536] if os.path.exists(kind) else make_block(kind, 65536, 1)
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, zlib.Z_FIXED if kind == "text33" and level == 6 else zlib.Z_DEFAULT_STRATEGY)
    c = co.compress(d) + co.flush()
    stride = (len(c) + 31) & ~15
    buf = np.zeros((n, stride), np.uint8); buf[:, :len(c)] = np.frombuffer(c, np.uint8)
    src = torch.from_numpy(buf).to(eng.dev)
    back = torch.empty((n, 65536), dtype=torch.uint8, device=eng.dev)
    j = eng.jobs_strided(src, stride, np.full(n, len(c), np.uint32), back, 65536, 65536)
    res = torch.empty(n * 32, dtype=torch.uint8, device=eng.dev)
    eng.decompress(j, n, results=res); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): eng.decompress(j, n, results=res)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    assert back[0].cpu().numpy().tobytes() == d
    prof = torch.zeros(16, dtype=torch.int64, device=eng.dev)
    eng.L.nxz_inflate_prof_set.argtypes = [C.c_void_p]
    eng.L.nxz_inflate_prof_set(prof.data_ptr())
    eng.decompress(j, n, results=res); torch.cuda.synchronize()
    eng.L.nxz_inflate_prof_set(None)
    p = prof.cpu().numpy() / n
    print("%-8.40s level %d: %6d compressed bytes, %.2f ms per 64 KiB stream (512 streams at once)" % (kind, level, len(c), dt * 1e3))
    print("    ticks per stream: other %.0f  flush %.0f  step set-up + look-ups + token decode %.0f  chain walk %.0f  prefix sum + limits %.0f  literals + matches %.0f | steps %.0f, bytes/step %.1f, bits/step %.1f, one-token path %.0f"
          % (p[0], p[1], p[2], p[8], p[9], p[3], p[4], p[5] / max(p[4], 1), p[6] / max(p[4], 1), p[7]))
