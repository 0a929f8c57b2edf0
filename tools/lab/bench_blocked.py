"""Host-buffer rate of the blocked gzip layer (include/nxz_blocked.h): a buffer in host memory ->
gzip members in host memory (memcpy to pinned staging, PCIe both ways, kernels, packing), and back.
usage: python tools/bench_blocked.py [MiB] [fixed]"""
import ctypes as C, os, sys, time, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import test_blocked as TB

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
fixed = 1 if len(sys.argv) > 2 and sys.argv[2] == "fixed" else 0
n = mib * 16
data = bench.gen_blocks(torch, torch.device("cuda", 0), n, 0).cpu().numpy().tobytes()
L = TB.lib()
total = [0]
keep = []
def cb(user, buf, ln):
    total[0] += ln
    if keep is not None: keep.append(C.string_at(buf, ln))
    return 0
sink = TB.SINK(cb)
o = TB.Opts(device=-1, fixed=fixed)
for rep in range(2):
    total[0] = 0; keep.clear()
    t0 = time.perf_counter()
    rc = L.nxz_blocked_deflate(data, len(data), C.byref(o), sink, None, None)
    dt = time.perf_counter() - t0
    assert rc == 0
image = b"".join(keep)
print("deflate (%s): %d MiB host buffer -> %d members, %.1f MiB: %.2f GiB/s of input, ratio %.3f"
      % ("fixed" if fixed else "dynamic", mib, (len(data) + 65279) // 65280, len(image) / 2**20, len(data) / dt / 2**30, len(data) / len(image)))
assert gzip.decompress(image[:TB.members_of(image)[63][0]]) == data[:63 * 65280]      # the first 63 members, by zlib
keep = None
for rep in range(2):
    total[0] = 0
    t0 = time.perf_counter()
    rc = L.nxz_blocked_inflate(image, len(image), C.byref(o), sink, None, None, None)
    dt = time.perf_counter() - t0
    assert rc == 0 and total[0] == len(data)
print("inflate: %.2f GiB/s of output (CRC32 and ISIZE of every member checked)" % (len(data) / dt / 2**30))
