"""Lane-per-stream inflate on runs of short periods (copies whose source overlaps the target).
usage: python tools/bench_inflate_periods.py"""
import importlib, os, sys, time, zlib
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
n = 131072
for name, d in (("period 3", (b"\x10\x80\xf0" * 21846)[:65536]), ("period 3 with breaks", b"".join(bytes([i & 255, 7, 9]) * 40 + b"xy" for i in range(600))[:65536].ljust(65536, b"z")),
                ("period 5", (b"abcde" * 13108)[:65536])):
    co = zlib.compressobj(6, zlib.DEFLATED, -15); c = co.compress(d) + co.flush()
    stride = (len(c) + 31) & ~15
    buf = np.zeros((1, stride), np.uint8); buf[0, :len(c)] = np.frombuffer(c, np.uint8)
    src = torch.from_numpy(buf).to(eng.dev).repeat(n, 1).contiguous()
    back = torch.empty((n, 65536), dtype=torch.uint8, device=eng.dev)
    j = eng.jobs_strided(src, stride, np.full(n, len(c), np.uint32), back, 65536, 65536)
    res = torch.empty(n * 32, dtype=torch.uint8, device=eng.dev)
    eng.decompress(j, n, results=res); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.decompress(j, n, results=res)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    assert back[0].cpu().numpy().tobytes() == d and back[n - 1].cpu().numpy().tobytes() == d
    print("%-22s %d streams (%d bytes each): %.1f GiB/s out (%.1f ms)" % (name, n, len(c), n * 65536 / dt / 2**30, dt * 1e3))
