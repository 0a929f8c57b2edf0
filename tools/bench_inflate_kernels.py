"""Which batched inflate kernel for which batch: a stream per wave (the target as window) against a stream per
lane, by batch size and by kind of stream (zlib -6 dynamic streams of the corpus blocks; the engine's own
fixed-Huffman output of the synthetic blocks; its own exact-table output of the corpus).  Forces the kernel with
NXZ_INFLATE_LANES_MIN; prints GiB/s of uncompressed bytes out.  -> profiles/rNN_inflate_by_batch_size.txt"""
import importlib, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench, corpus
pkg = importlib.import_module("power-gzip_amd")
B, S = 65536, 73856
sizes = [int(x) for x in os.environ.get("SIZES", "4096,16384,65536,131072,262144").split(",")]
_, blocks, _ = corpus.load(B)
only = [c for c in os.environ.get("CLASSES", "").split(",") if c]                  # CLASSES=text,xml: those classes of the corpus only
raw = [b for cls, _, b in blocks if len(b) == B and (not only or cls in only)]


def timed(eng, jobs, n):
    eng.decompress(jobs, n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2):
        res = eng.decompress(jobs, n)
    e1.record()
    torch.cuda.synchronize()
    r = eng.results_to_host(res)
    assert (r["cc"] == 0).all()
    return e0.elapsed_time(e1) / 2


KERNELS = [k for k in os.environ.get("KERNELS", "wg,waves,cut,lanes").split(",") if k]    # KERNELS=wg: the workgroup-per-stream kernel only
print("%-34s %8s | %s" % ("streams", "n", " ".join("%12s" % {"wg": "per workgroup", "waves": "per wave", "cut": "cut", "lanes": "per lane", "old": "as round 5", "engine": "the engine"}[k] for k in KERNELS)))
for kind in ("zlib -6 of the corpus blocks", "own fixed-Huffman (synthetic)", "own fixed-Huffman (corpus)", "own exact tables (corpus)"):
    for n in sizes:
        row = []
        for kernel in KERNELS:
            if kernel == "cut" and n > 24576:
                row.append(float("nan"))
                continue
            os.environ["NXZ_INFLATE_LANES_MIN"] = "1" if kernel == "lanes" else "1000000000"
            os.environ["NXZ_INFLATE_WG"] = "1" if kernel == "wg" else "0"
            os.environ["NXZ_INFLATE_CUT"] = "1" if kernel == "cut" else "0"                 # every stream cut inside its first block (nxz_inflate_cut.hip)
            os.environ["NXZ_LANES_FIXED"] = "2" if kind.startswith("own fixed") else "0"   # (the fixed-code-only kernel in front, as the engine's sampling would choose)
            os.environ["NXZ_INFLATE_LDS_MAX"] = os.environ.get("LDS_MAX", "0")          # (LDS_MAX=1024: small batches as the engine runs them, the window in LDS)
            if kernel in ("old", "engine"):
                # the engine's own choice: with the workgroup kernel out of it (what round 5 did), or as it stands
                for k in ("NXZ_INFLATE_LANES_MIN", "NXZ_INFLATE_CUT", "NXZ_LANES_FIXED", "NXZ_INFLATE_LDS_MAX", "NXZ_INFLATE_WG"):
                    os.environ.pop(k, None)
                if kernel == "old":
                    os.environ["NXZ_INFLATE_WG"] = "0"
            eng = pkg.Engine(0)
            if kind.startswith("zlib"):
                streams = []
                for b in raw:
                    c = zlib.compressobj(6, zlib.DEFLATED, -15)
                    streams.append(c.compress(b) + c.flush())
                cs = (max(len(s) for s in streams) + 64 + 15) & ~15
                host = np.zeros((len(raw), cs), np.uint8)
                for i, s in enumerate(streams):
                    host[i, :len(s)] = np.frombuffer(s, np.uint8)
                rep = -(-n // len(raw))
                src = torch.from_numpy(host).to(eng.dev).repeat(rep, 1)[:n]
                clen = np.tile(np.array([len(s) for s in streams], np.uint32), rep)[:n]
                dst = torch.zeros((n, B), dtype=torch.uint8, device=eng.dev)
                jobs = eng.jobs_strided(src, cs, clen, dst, B, B)
                data = None
            else:
                if "synthetic" in kind:
                    data = bench.gen_blocks(torch, eng.dev, n, 0)
                    fc = pkg.FC_COMPRESS_FHT
                else:
                    host = np.stack([np.frombuffer(raw[i % len(raw)], np.uint8) for i in range(n)])
                    data = torch.from_numpy(host).to(eng.dev)
                    fc = pkg.FC_COMPRESS_FHT if "fixed" in kind else pkg.FC_COMPRESS_DHTGEN
                comp = torch.empty((n, S), dtype=torch.uint8, device=eng.dev)
                j1 = eng.jobs_strided(data, B, np.full(n, B, np.uint32), comp, S, S)
                r = eng.results_to_host(eng.compress(fc, j1, n)[0])
                dst = torch.zeros((n, B), dtype=torch.uint8, device=eng.dev)
                jobs = eng.jobs_strided(comp, S, r["tpbc"].astype(np.uint32), dst, B, B)
            ms = timed(eng, jobs, n)
            row.append(n * B / ms / 1e-3 / 2 ** 30)
            if kernel == "wg":
                # the bytes, not just the completion codes: against the blocks themselves
                if data is not None:
                    assert torch.equal(dst, data), "workgroup kernel: output differs"
                else:
                    got = dst[:len(raw)].cpu().numpy()
                    for i, b in enumerate(raw[:n]):
                        assert got[i].tobytes() == b, "workgroup kernel: stream %d differs" % i
                why = eng.wg_reasons()
                if why and why.get("handed_back"):
                    print("    (handed back: %s)" % why, flush=True)
                if os.environ.get("NXZ_WG_PROF"):
                    pr = eng.wg_prof()
                    print("    (cycles a stream: %s)" % ", ".join(("%s %.0f" % (k, v) if v >= 100 else "%s %.2f" % (k, v)) if not isinstance(v, list) else "%s %s" % (k, v) for k, v in pr.items()), flush=True)
            eng.close()
            del jobs, dst
            torch.cuda.empty_cache()
        print("%-34s %8d | %s" % (kind, n, " ".join("%12.1f" % v for v in row)), flush=True)
