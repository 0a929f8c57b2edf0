"""Are the lane-per-stream inflate kernels bound by what their lanes fetch, or by the number of memory instructions their
wavefronts issue?  40 960 fixed-code text streams a lane each, once all different and once all the same bytes (one cache line
for all 64 lanes of a load, and the lanes run dry together): round 4, before the kernels were rebuilt, 99.8 ms against 57.3
(DESIGN.md section 6).  usage: python tools/lanes_lockstep.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ["NXZ_INFLATE_LANES_MIN"] = "1"
import numpy as np, torch, bench
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
n = 40960
for what in ("distinct", "identical"):
    src = bench.gen_text(torch, eng.dev, n, 0)
    if what == "identical":
        src[:] = src[0].clone()
    comp = torch.empty((n, 73856), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, 65536, np.full(n, 65536, np.uint32), comp, 73856, 73856)
    res, _ = eng.compress(pkg.FC_COMPRESS_FHT, jobs, n)
    r = eng.results_to_host(res)
    assert (r["cc"] == 0).all()
    back = torch.empty((n, 65536), dtype=torch.uint8, device=eng.dev)
    jd = eng.jobs_strided(comp, 73856, r["tpbc"].astype(np.uint32), back, 65536, 65536)
    eng.decompress(jd, n); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): eng.decompress(jd, n)
    e1.record(); torch.cuda.synchronize()
    assert torch.equal(back, src)
    print("%s text blocks, %d streams a lane each: %.1f ms (%.1f GiB/s)" % (what, n, e0.elapsed_time(e1) / 3, n * 65536 / (e0.elapsed_time(e1) / 3 * 1e-3) / 2**30))
