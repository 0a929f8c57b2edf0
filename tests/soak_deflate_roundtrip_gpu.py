"""Soak (not collected by pytest; run on the GPU box: python tests/soak_deflate_roundtrip_gpu.py <blocks> <seconds>):
the same batch of synthetic blocks (bench.gen_blocks: SURVEY 8(d) C2 recipe) is compressed pass after pass; every pass'
output must equal the first pass' byte for byte (a data race shows as an unstable block), the first pass' equals the
oracle's on the blocks that differ, and every pass inflates back to the source on the device.  Round 3: this found a
race in the parse's second pass (one block in 65536, one pass in sixteen) that no parity test had met; the last
results are kept in profiles/rNN_soak.txt."""
import importlib, os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
import oracle_lib as O
pkg = importlib.import_module("power-gzip_amd")
n = int(sys.argv[1]); secs = float(sys.argv[2]); fc = pkg.FC_COMPRESS_FHT
eng = pkg.Engine(0)
src = bench.gen_blocks(torch, eng.dev, n, 0)
dst = torch.zeros((n, 73856), dtype=torch.uint8, device=eng.dev)
ref = None
lens = np.full(n, 65536, np.uint32)
jobs = eng.jobs_strided(src, 65536, lens, dst, 73856, 73856)
back = torch.zeros((n, 65536), dtype=torch.uint8, device=eng.dev)
t0 = time.time(); it = 0; nbad_def = nbad_inf = 0
while time.time() - t0 < secs:
    dst.zero_()
    r = eng.results_to_host(eng.compress(fc, jobs, n)[0]).copy()
    if ref is None:
        ref = dst.clone(); rref = r.copy()
    else:
        d = (dst != ref).any(dim=1)
        if bool(d.any()) or not (r["tpbc"] == rref["tpbc"]).all():
            idx = torch.nonzero(d).flatten().tolist()
            nbad_def += len(idx)
            for i in idx[:3]:
                a = dst[i].cpu().numpy(); b = ref[i].cpu().numpy(); k = np.nonzero(a != b)[0]
                exp, bits = O.deflate_fixed(src[i].cpu().numpy().tobytes())
                print("iter %d: DEFLATE output of block %d differs from the first pass: %d bytes, first at %d, tpbc %d vs %d; first pass == oracle %s, this pass == oracle %s"
                      % (it, i, len(k), k[0], r["tpbc"][i], rref["tpbc"][i], ref[i, :len(exp)].cpu().numpy().tobytes() == exp, dst[i, :len(exp)].cpu().numpy().tobytes() == exp), flush=True)
    back.zero_()
    jobs2 = eng.jobs_strided(dst, 73856, r["tpbc"].astype(np.uint32), back, 65536, 65536)
    r2 = eng.results_to_host(eng.decompress(jobs2, n))
    d = (back != src).any(dim=1)
    if bool(d.any()):
        idx = torch.nonzero(d).flatten().tolist(); nbad_inf += len(idx)
        for i in idx[:3]:
            a = back[i].cpu().numpy(); b = src[i].cpu().numpy(); k = np.nonzero(a != b)[0]
            print("iter %d: INFLATE output of block %d differs: %d bytes, first at %d (cc %d)" % (it, i, len(k), k[0], r2["cc"][i]), flush=True)
    it += 1
print("soak %s: %d iterations of %d blocks, deflate-unstable blocks %d, round-trip failures %d" % (os.environ.get("NXZ_ENGINE_LIB", "libnxz_engine.so"), it, n, nbad_def, nbad_inf))
