"""The inflate half of BASELINE configs[4] alone: the mixed 64 KiB blocks (zeros / 33-symbol text / copies / random by index)
through the fixed-code deflate, what shrank through the batched inflate; milliseconds per pass.  NXZ_ENGINE_LIB picks another
build of the engine (tools/build_variant.sh) for an A/B.  usage: python tools/c5_inflate_time.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, bench
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
n = 163840
src = bench.gen_mixed(torch, eng.dev, n, 0)
comp = torch.empty((n, 73856), dtype=torch.uint8, device=eng.dev)
jobs = eng.jobs_strided(src, 65536, np.full(n, 65536, np.uint32), comp, 73856, 73856)
res, _ = eng.compress(pkg.FC_COMPRESS_FHT, jobs, n)
r = eng.results_to_host(res)
ok = np.nonzero(r["cc"] == 0)[0]
back = torch.empty((n, 65536), dtype=torch.uint8, device=eng.dev)
j = np.zeros(len(ok), pkg.JOB_DTYPE)
j["src"] = np.uint64(comp.data_ptr()) + ok.astype(np.uint64) * np.uint64(73856)
j["dst"] = np.uint64(back.data_ptr()) + ok.astype(np.uint64) * np.uint64(65536)
j["src_len"] = r["tpbc"][ok]; j["dst_cap"] = 65536; j["in_adler"] = 1
jd = eng.to_device(j)
eng.decompress(jd, len(ok)); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): eng.decompress(jd, len(ok))
e1.record(); torch.cuda.synchronize()
print("c5 inflate of %d streams: %.1f ms" % (len(ok), e0.elapsed_time(e1) / 3))
