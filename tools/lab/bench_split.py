#!/usr/bin/env python3
"""Deflate pipeline by function code on bench.py's synthetic blocks: fixed Huffman (LZ77 kernel ->
entropy kernel) and the DHTGEN code (LZ77 + counts -> device dhtgen -> entropy)."""
import importlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def run(n):
    import torch
    import bench
    pkg = importlib.import_module("power-gzip_amd")
    eng = pkg.Engine(0)
    dev = eng.dev
    src = bench.gen_blocks(torch, dev, n, 0)
    dst = torch.empty((n, 73856), dtype=torch.uint8, device=dev)
    jobs = eng.jobs_strided(src, 65536, np.full(n, 65536, np.uint32), dst, 73856, 73856)
    res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    out = {}
    for name, fc in (("fht", pkg.FC_COMPRESS_FHT), ("dhtgen", pkg.FC_COMPRESS_DHTGEN)):
        eng.compress(fc, jobs, n, results=res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            eng.compress(fc, jobs, n, results=res)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        r = res.cpu().numpy().view(pkg.RESULT_DTYPE)
        assert ((r["cc"] == 0) | (r["cc"] == 64)).all(), np.unique(r["cc"])
        out[name] = {"GiB_s": round(n * 65536 / ms / 1e-3 / 2 ** 30, 2), "ms": round(ms, 3),
                     "ratio": round(n * 65536.0 / float(r["tpbc"].astype(np.float64).sum()), 4)}
    print(json.dumps({"blocks": n, **out}))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    run(n)
