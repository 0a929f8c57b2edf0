"""Soak run (not collected by pytest): 8 seeds x 1200 random blocks (kinds, sizes at tile / piece edges,
histories) through the HIP deflate engine, compared bit for bit with the oracle.  python tests/soak_gpu.py [seeds [blocks per seed]]"""
import importlib, os, random, sys, zlib
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as O
from datagen import make_block
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
SI, SO = 65536 + 16, 73856
kinds = ["zeros", "random", "text33", "alice", "lz", "periodic", "binary", "sparse"]
edges = [0, 1, 3, 4, 5, 15, 16, 17, 63, 64, 65, 511, 512, 513, 16383, 16384, 16385, 32767, 32768, 32769, 49152, 65535, 65536]
bad = 0
NSEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PER_SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
for seed in range(1, NSEEDS + 1):
    rnd = random.Random(seed * 7919)
    blocks, hl_ = [], []
    for i in range(PER_SEED):
        hl = rnd.choice([0, 0, 0, 16, 48, 4096, 16384, 32768]); room = 65536 - hl
        n = rnd.choice(edges) if rnd.random() < 0.4 else rnd.randrange(0, room + 1); n = min(n, room)
        body = make_block(rnd.choice(kinds), n, seed=seed * 100000 + i)
        hist = make_block(rnd.choice(kinds), hl, seed=seed * 100000 + 50000 + i) if hl else b""
        if hl and rnd.random() < 0.5 and n:
            k = min(hl, n); body = hist[-k:] + body[k:]
        blocks.append(hist + body); hl_.append(hl)
    host = np.zeros((len(blocks), SI), np.uint8)
    for i, b in enumerate(blocks): host[i, :len(b)] = np.frombuffer(b, np.uint8)
    src = torch.from_numpy(host).to(eng.dev)
    dst = torch.zeros((len(blocks), SO), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, SI, np.array([len(b) for b in blocks], np.uint32), dst, SO, SO, hist_len=np.array(hl_, np.uint32), in_crc=seed, in_adler=seed + 1)
    res, _ = eng.compress(pkg.FC_COMPRESS_RESUME_FHT, jobs, len(blocks))
    r = eng.results_to_host(res); out = dst.cpu().numpy()
    for i, (b, hl) in enumerate(zip(blocks, hl_)):
        exp, bits = O.deflate_fixed(b, hist=hl)
        if len(exp) > len(b):
            ok = r["cc"][i] == 64
        else:
            ok = r["cc"][i] == 0 and r["tpbc"][i] == len(exp) and out[i, :len(exp)].tobytes() == exp and r["crc"][i] == zlib.crc32(b[hl:], seed) and r["adler"][i] == zlib.adler32(b[hl:], seed + 1)
        if not ok:
            bad += 1; print("MISMATCH seed", seed, "block", i, len(b), hl, r["cc"][i])
    print("seed", seed, "done, mismatches so far", bad, flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
