"""A/B of engine builds (kernel variants) on one box: tools/ab_lz77.py libnxz_engine.so libnxz_engine_base.so ...
Every build runs in a process of its own (NXZ_ENGINE_LIB), rounds interleaved; prints the LZ77 / entropy kernel
times per launch set by the engine's own HIP events, for the corpus (COMPRESS_DHTGEN) and the synthetic blocks (FHT)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import numpy as np
    import torch
    import bench, corpus
    pkg = importlib.import_module("power-gzip_amd")
    eng = pkg.Engine(0)
    n = int(os.environ.get("AB_JOBS", "32768"))
    out = {}
    _, blocks, _ = corpus.load(65536)
    legs = []
    for cls in os.environ.get("AB_CLASSES", "").split(","):          # "" = the whole corpus; "msgpack,json": those classes, a leg each
        full = [np.frombuffer(b, np.uint8) for c, _, b in blocks if len(b) == 65536 and (not cls or c == cls)]
        host = np.stack([full[i % len(full)] for i in range(n)])
        legs.append((cls or "corpus", pkg.FC_COMPRESS_DHTGEN, torch.from_numpy(host).to(eng.dev)))
    if os.environ.get("AB_SYNTH", "1") == "1":
        legs.append(("synth", pkg.FC_COMPRESS_FHT, bench.gen_blocks(torch, eng.dev, n, 0)))
    for name, fc, src in legs:
        dst = torch.empty((n, 73856), dtype=torch.uint8, device=eng.dev)
        jobs = eng.jobs_strided(src, 65536, np.full(n, 65536, np.uint32), dst, 73856, 73856)
        res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
        ms, st, _ = bench.timed_compress(torch, eng, fc, jobs, n, res, 4, 1)
        r = eng.results_to_host(res)
        out[name] = {"GiB_s": round(n * 65536 / ms / 1e-3 / 2 ** 30, 2), "ms": round(ms, 3), "lz77_ms": round(st[0], 3), "entropy_ms": round(st[2], 3),
                     "csum": int(r["tpbc"].astype(np.uint64).sum()), "crcx": int(np.bitwise_xor.reduce(r["crc"]))}
        del dst, src
    print("AB " + json.dumps(out))
    sys.exit(0)
libs = sys.argv[1:]
ref = None
for rnd in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for lib in libs:
        env = dict(os.environ, NXZ_ENGINE_LIB=lib)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("AB ")]
        if not line:
            print("%-28s FAILED: %s" % (lib, p.stderr[-400:]), flush=True)
            continue
        res = json.loads(line[0][3:])
        if os.environ.get("AB_VERBOSE"):
            print("%-28s %s" % (lib, line[0][3:]), flush=True)
        sums = {k: (v["csum"], v["crcx"]) for k, v in res.items()}
        ref = ref or sums
        print("%-28s %s  lz77 ms: %s%s" % (lib, " ".join("%s %.1f" % (k, v["GiB_s"]) for k, v in res.items()),
                                          " ".join("%.2f" % v["lz77_ms"] for v in res.values()), "" if sums == ref else "  OUTPUT DIFFERS from the first build's"), flush=True)
