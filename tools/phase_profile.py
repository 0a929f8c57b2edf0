"""Diagnostic: per-phase cycle shares of the LZ77 kernel (thread 0 of every workgroup)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
pkg = importlib.import_module("power-gzip_amd")
NAMES = ["load", "cksum", "seed", "hash", "chain (wave 0)", "M3b dist-1 runs", "pass1", "mark", "pass2 (token walk)", "out (records, counts, bitmaps)", "e-flags + piece links", "M3a members", "match after the chain", "(count) positions queued for M2", "(count) open after 24 bytes", "(count) tails"]
FC = {"fht": pkg.FC_COMPRESS_FHT, "dhtgen": pkg.FC_COMPRESS_DHTGEN}[os.environ.get("FC", "fht")]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = pkg.Engine(0)
src = bench.gen_blocks(torch, eng.dev, n, 0)
if len(sys.argv) > 2 and sys.argv[2].startswith("corpus"):       # "corpus", or "corpus:<class>" (elf, msgpack, text, ...)
    import corpus
    _, blocks, _ = corpus.load(65536)
    want = sys.argv[2][7:]
    full = [np.frombuffer(b, np.uint8) for cls, _, b in blocks if len(b) == 65536 and (not want or cls == want)]
    host = np.stack([full[i % len(full)] for i in range(n)])
    src = torch.from_numpy(host).to(eng.dev)
elif len(sys.argv) > 2:
    from datagen import make_block
    b = np.frombuffer(make_block(sys.argv[2], 65536, 1), np.uint8)
    src[:] = torch.from_numpy(b.copy()).to(eng.dev)
dst = torch.empty((n, 73856), dtype=torch.uint8, device=eng.dev)
size = int(sys.argv[3]) if len(sys.argv) > 3 else 65536             # bytes of each block that are compressed
jobs = eng.jobs_strided(src, 65536, np.full(n, size, np.uint32), dst, 73856, 73856)
prof = torch.zeros(64, dtype=torch.int64, device=eng.dev)
eng.L.nxz_lz77_prof_set.argtypes = [C.c_void_p]
eng.compress(FC, jobs, n)
torch.cuda.synchronize()
assert eng.L.nxz_lz77_prof_set(prof.data_ptr()) == 0
eng.compress(FC, jobs, n)
torch.cuda.synchronize()
eng.L.nxz_lz77_prof_set(None)
p = prof.cpu().numpy().astype(np.float64) / n
tot = p[:13].sum()
for i, name in enumerate(NAMES):
    print("%-20s %10.0f cycles/block  %5.1f%%" % (name, p[i], 100 * p[i] / tot))
print("match waves, wave-cycles per block: waiting for the chain %.0f, M1 %.0f, M2 in the loop %.0f, M2 leftovers %.0f" % (p[16], p[17], p[18], p[19]))
print("total %.0f cycles/block (s_memtime ticks = 100 MHz? see guide)" % tot)
if p[32:48].any():
    N2 = ["level2 calls", "level2 entries", "group rounds", "tails settled by a group", "rounds that found no group", "batches to the lane-per-tail loop", "its trips",
          "tails in it", "batches to the 16-lanes-per-tail path", "its steps", "level1 calls", "level1 entries", "members found in level 2", "batches the gate turned away", "tails in the 16-lane path"]
    for i, name in enumerate(N2):
        print("%-40s %10.1f per block" % (name, p[32 + i]))
