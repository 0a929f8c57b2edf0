"""Streams longer than LDS through the workgroup-per-stream kernel against the engine's older routes: N streams of S KiB each
(zlib -6 of corpus text run together), one nxz_batch_decompress call.  SIZES (KiB), COUNTS, KERNELS=wg,old."""
import importlib, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import corpus
pkg = importlib.import_module("power-gzip_amd")
_, blocks, _ = corpus.load(65536)
raw = [b for cls, _, b in blocks if len(b) == 65536]
sizes = [int(x) for x in os.environ.get("SIZES", "64,256,1024,4096").split(",")]
counts = [int(x) for x in os.environ.get("COUNTS", "1,16,256").split(",")]
KERNELS = os.environ.get("KERNELS", "wg,old").split(",")
print("%8s %6s | %s" % ("KiB", "n", " ".join("%14s" % k for k in KERNELS)))
for kib in sizes:
    for n in counts:
        row = []
        for kernel in KERNELS:
            os.environ["NXZ_INFLATE_WG"] = "1" if kernel == "wg" else "0"
            eng = pkg.Engine(0)
            per = kib // 64
            plains, streams = [], []
            for i in range(min(n, 8)):
                d = b"".join(raw[(i * per + k) % len(raw)] for k in range(per))
                c = zlib.compressobj(6, zlib.DEFLATED, -15)
                plains.append(d); streams.append(c.compress(d) + c.flush())
            cs = (max(len(s) for s in streams) + 64 + 15) & ~15
            host = np.zeros((n, cs), np.uint8)
            for i in range(n):
                s = streams[i % len(streams)]
                host[i, :len(s)] = np.frombuffer(s, np.uint8)
            src = torch.from_numpy(host).to(eng.dev)
            clen = np.array([len(streams[i % len(streams)]) for i in range(n)], np.uint32)
            B = kib * 1024
            dst = torch.zeros((n, B + 64), dtype=torch.uint8, device=eng.dev)
            jobs = eng.jobs_strided(src, cs, clen, dst, B + 64, B + 64)
            eng.decompress(jobs, n)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                res = eng.decompress(jobs, n)
            e1.record()
            torch.cuda.synchronize()
            r = eng.results_to_host(res)
            assert (r["cc"] == 0).all(), r["cc"]
            got = dst.cpu().numpy()
            for i in range(min(n, 8)):
                assert got[i, :B].tobytes() == plains[i], (kernel, kib, i)
            ms = e0.elapsed_time(e1) / 3
            row.append("%7.3f ms %5.1f" % (ms, n * B / ms / 1e-3 / 2 ** 30))
            if kernel == "wg":
                why = eng.wg_reasons()
                if why and why.get("handed_back"):
                    print("    (handed back: %s)" % why, flush=True)
                if os.environ.get("NXZ_WG_PROF"):
                    pr = eng.wg_prof()
                    print("    (cycles a stream: %s)" % ", ".join(("%s %.0f" % (k, v) if v >= 100 else "%s %.2f" % (k, v)) if not isinstance(v, list) else "%s %s" % (k, v) for k, v in pr.items()), flush=True)
            eng.close()
            del jobs, dst, src
            torch.cuda.empty_cache()
        print("%8d %6d | %s" % (kib, n, " ".join("%14s" % v for v in row)), flush=True)
