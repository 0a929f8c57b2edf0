#!/usr/bin/env python3
"""One leg of the bench workload for profiling runs (rocprofv3 --kernel-trace --stats / --pmc ... -- python3 tools/prof_workload.py <leg>):
  fht            fixed-Huffman deflate of 65536 synthetic blocks (bench.gen_blocks), 3 passes
  dhtgen         COMPRESS_DHTGEN of the real-data corpus replicated to >= 65536 jobs, 3 passes
  inflate_zlib6  zlib -6 streams of the corpus blocks, >= 262144 streams as the bench leg runs them, 3 passes
  inflate_wg     the same streams, >= 65536 of them (both: a stream per workgroup, nxz_inflate_wg.hip), 3 passes
  inflate_own    the engine's own fixed-Huffman output, 262144 streams (a stream per lane), 2 passes
  inflate_stream ONE 256 MiB zlib -6 stream, 3 passes
  c5             BASELINE configs[4]: 163840 mixed blocks (10 GiB), bench.c5_prepare's step (compress + wrap + decompress + wrap + compare), 2 passes;
                 c5:<kind> (zeros, text, lz, random): 40960 blocks of that kind only, compress and decompress
Prints one JSON line: the leg, units per pass (blocks / streams / 64 KiB of output), passes, algorithmic bytes per unit."""
import importlib, json, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench, corpus
pkg = importlib.import_module("power-gzip_amd")
leg = sys.argv[1]
eng = pkg.Engine(0)
dev = eng.dev
B, S = 65536, 73856


def corpus_blocks():
    _, blocks, _ = corpus.load(B)
    return [b for _, _, b in blocks]


if leg in ("fht", "inflate_own"):
    n = 65536 if leg == "fht" else 262144
    src = bench.gen_blocks(torch, dev, n, 0)
    dst = torch.empty((n, S), dtype=torch.uint8, device=dev)
    jobs = eng.jobs_strided(src, B, np.full(n, B, np.uint32), dst, S, S)
    res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    passes = 3 if leg == "fht" else 1
    for _ in range(passes):
        eng.compress(pkg.FC_COMPRESS_FHT, jobs, n, results=res)
    r = eng.results_to_host(res)
    c = float(r["tpbc"].astype(np.float64).sum())
    if leg == "inflate_own":
        back = torch.empty((n, B), dtype=torch.uint8, device=dev)
        jobs2 = eng.jobs_strided(dst, S, r["tpbc"].astype(np.uint32), back, B, B)
        passes = 2
        for _ in range(passes):
            eng.decompress(jobs2, n)
        torch.cuda.synchronize()
    print(json.dumps({"leg": leg, "units": n, "passes": passes, "algorithmic_bytes_per_unit": (n * B + c) / n}))
elif leg in ("dhtgen", "inflate_zlib6", "inflate_wg"):
    raw = corpus_blocks()
    rep = -(-(262144 if leg == "inflate_zlib6" else 65536) // len(raw))        # (inflate_zlib6: the bench leg's own stream count and routes)
    n = len(raw) * rep
    lens = np.tile(np.array([len(b) for b in raw], np.uint32), rep)
    if leg == "dhtgen":
        host = np.zeros((len(raw), B), np.uint8)
        for i, b in enumerate(raw):
            host[i, :len(b)] = np.frombuffer(b, np.uint8)
        src = torch.from_numpy(host).to(dev).repeat(rep, 1)
        dst = torch.empty((n, S), dtype=torch.uint8, device=dev)
        jobs = eng.jobs_strided(src, B, lens, dst, S, S)
        res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        for _ in range(3):
            eng.compress(pkg.FC_COMPRESS_DHTGEN, jobs, n, results=res)
        r = eng.results_to_host(res)
        print(json.dumps({"leg": leg, "units": n, "passes": 3, "algorithmic_bytes_per_unit": (float(lens.sum()) + float(r["tpbc"].astype(np.float64).sum())) / n}))
    else:
        streams = []
        for b in raw:
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            streams.append(c.compress(b) + c.flush())
        cs = (max(len(s) for s in streams) + 64 + 15) & ~15
        host = np.zeros((len(raw), cs), np.uint8)
        for i, s in enumerate(streams):
            host[i, :len(s)] = np.frombuffer(s, np.uint8)
        src = torch.from_numpy(host).to(dev).repeat(rep, 1)
        clen = np.tile(np.array([len(s) for s in streams], np.uint32), rep)
        dst = torch.zeros((n, B), dtype=torch.uint8, device=dev)
        jobs = eng.jobs_strided(src, cs, clen, dst, B, B)
        for _ in range(3):
            eng.decompress(jobs, n)
        torch.cuda.synchronize()
        print(json.dumps({"leg": leg, "units": n, "passes": 3, "algorithmic_bytes_per_unit": (float(lens.sum()) + float(clen.sum())) / n}))
elif leg == "inflate_stream":
    raw = corpus_blocks()
    mib = 256
    data = (b"".join(raw) * ((mib << 20) // sum(len(b) for b in raw) + 1))[:mib << 20]
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    src = torch.from_numpy(np.frombuffer(comp, np.uint8).copy()).to(dev)
    dst = torch.zeros(len(data) + 4096, dtype=torch.uint8, device=dev)
    for _ in range(3):
        rc, info = eng.inflate_stream(src, len(comp), dst)
        assert rc == 0 and info["out_len"] == len(data)
    torch.cuda.synchronize()
    print(json.dumps({"leg": leg, "units": len(data) // B, "passes": 3, "algorithmic_bytes_per_unit": (len(data) + len(comp)) / (len(data) // B)}))
elif leg.startswith("c5"):
    kind = leg[3:]
    n = 163840 if not kind else 4 * 40960
    src = bench.gen_mixed(torch, dev, n, 0)
    if kind:
        src = src[{"zeros": 0, "text": 1, "lz": 2, "random": 3}[kind]::4].contiguous()
        n = src.shape[0]
    step, info = bench.c5_prepare(torch, eng, pkg, src)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    assert int(info["flag"].item()) == 0
    print(json.dumps({"leg": leg, "units": n, "passes": 2, "algorithmic_bytes_per_unit": 2 * (n * B + info["c_bytes"]) / n, "stored": info["stored"]}))
eng.close()
