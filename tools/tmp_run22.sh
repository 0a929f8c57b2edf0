cd /root/repo
timeout 90 python - <<'PY' 2>&1 | grep -v amdgpu | tail -5
import sys, importlib, zlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from datagen import make_block
import oracle_lib as O
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
for nb in (1, 3, 700):
    blocks = [make_block(["alice", "lz", "binary", "text33", "zeros", "random"][i % 6], 65536 if i % 7 else 30000 + i, seed=i) for i in range(nb)]
    host = np.zeros((nb, 65536), np.uint8)
    for i, b in enumerate(blocks): host[i, :len(b)] = np.frombuffer(b, np.uint8)
    src = torch.from_numpy(host).to(eng.dev)
    dst = torch.zeros((nb, 73856), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, 65536, np.array([len(b) for b in blocks], np.uint32), dst, 73856, 73856)
    res, _ = eng.compress(pkg.FC_COMPRESS_DHTGEN, jobs, nb)
    torch.cuda.synchronize()
    r = eng.results_to_host(res)
    out = dst.cpu().numpy()
    bad = 0
    for i, b in enumerate(blocks[:40]):
        tok, nt = O.lz77(b); ll, d = O.counts(tok, nt); dht, dhtlen = O.dhtgen(ll, d); exp, bits = O.deflate_dynamic(b, dht, dhtlen)
        if not (r["cc"][i] in (0, 64) and r["tpbc"][i] == len(exp) and out[i, :len(exp)].tobytes() == exp): bad += 1
    print("blocks", nb, "cc", np.unique(r["cc"]), "mismatches vs oracle among the first 40:", bad)
PY
echo "exit: $?"
