#!/usr/bin/env python3
"""bench.py -- throughput of the DEFLATE hot path on MI355X (BASELINE.json metric:
"GiB/s uncompressed in (deflate) + out (inflate), Silesia 64KiB blocks").

  python bench.py --gpus N --steps K --warmup W [--config corpus|c2|c5] [--blocks B] [--corpus-jobs J]
                  [--no-c2] [--no-c5] [--no-inflate] [--no-api] [--no-cpu-baseline]

config corpus (default; BASELINE.json configs[2], the metric's own kind of data): the real-data corpus --
  Silesia from $SILESIA_DIR (sha256-checked against the pins of the reference's oct/silesia-*.source) or
  the recorded real-data fallback of this image (tests/corpus.py) -- cut at 64 KiB and replicated to J
  jobs per GPU, resident in HBM.  A *step* is one pass of the deflate engine with an exact dynamic-Huffman
  table per block built on the device (function code COMPRESS_DHTGEN: LZ77 kernel -> dhtgen kernel ->
  entropy kernel).  Every rank owns its own J jobs (independent units, no data-path collective: weak
  scaling); the only collectives are the barrier and the reductions of {bytes, elapsed}.  Rank 0 prints
  ONE JSON line with `roofline` and `cpu_baseline` (zlib -1 on the same chunks, the reference's software
  path) and, at N = 1, outside the timed region:
    "inflate_zlib6", "inflate_stream"  the inflate engine on zlib -6 streams of the corpus chunks (batch)
               and on ONE long zlib -6 stream (block-boundary speculation, BASELINE configs[3])
    "api"      the reference's own API (nx_compress2 / nx_uncompress / inflate() in steps) on host buffers
    "c2"       BASELINE configs[1]: fixed-Huffman deflate of synthetic 64 KiB blocks (SURVEY 8(d) C2 recipe)
               with its own roofline / cpu_baseline, and the inflate engine on that output
    "c5"       BASELINE configs[4]: the mixed-entropy batch, compress + stored fallback + decompress + compare
config c2: the "c2" object as the headline (2^20 blocks per GPU by default), weak scaling over N GPUs.
config c5: 10 GiB of mixed-entropy 64 KiB blocks (index mod 4: zeros / 33-symbol text / makedata-style
  LZ copies / random bytes), cut into contiguous shards over the N ranks (strong scaling).  A step =
  compress (FHT; what does not shrink is stored through the WRAP function code, inside the timed region)
  + decompress + compare on the device.

Inside the timed region: the engine launches only.  Outside: data generation, zlib verification of
a sample, the CPU baselines (with the oracle parity check).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BLOCK = 65536
STRIDE_OUT = 73856
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured there with a float4 copy)
_copy_peak = None


def copy_peak_gbs(torch, dev, eng=None):
    """the achievable HBM figure, measured on this box with a copy kernel (1 GiB device to device, read +
    written bytes over the time of the copy by HIP events on the copy's stream).  The engine's own copy kernel -- 16 bytes a
    lane, six workgroups a CU, nontemporal -- when an engine is at hand, an elementwise torch kernel, and the runtime's
    device-to-device memcpy (which rounds 1-5 took alone: 4.8-5.5 TB/s on these boxes, the kernels 6.0); the best of the three.
    (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; tools/lab/copy_peak.py: no grid or unrolling reached more than
    6.04 here.)"""
    global _copy_peak
    if _copy_peak is None:
        n = 1 << 30
        a = torch.empty(n, dtype=torch.uint8, device=dev)
        b = torch.empty(n, dtype=torch.uint8, device=dev)
        a.zero_()
        best = 0.0
        a32, b32 = a.view(torch.float32), b.view(torch.float32)
        for copy in ([lambda: b.copy_(a), lambda: torch.add(a32, 0.0, out=b32)] + ([lambda: eng.copy_device(b, a)] if eng is not None else [])):
            copy()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                copy()
            e1.record()
            torch.cuda.synchronize(dev)
            best = max(best, 2.0 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        _copy_peak = round(best, 1)
        del a, b
    return _copy_peak


def roof(achieved, traffic, kernel, peak_measured=None, **extra):
    """the roofline object of the contract (+ the figure against the copy kernel's rate on this box)"""
    out = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "kernel": kernel}
    if traffic is not None:
        out["traffic_source"] = ("bytes per block of the committed rocprofv3 --pmc pass of this workload (profiles/%s; FETCH_SIZE / WRITE_SIZE "
                                 "in runs of their own, corrected as MI355X_MICROARCH.md prescribes) x the blocks of this step -- not counters of this run"
                                 % "|".join(sorted(set(PMC_SOURCE.values()))))
    if peak_measured:
        out["peak_measured"] = peak_measured
        out["frac_of_measured"] = round(achieved / peak_measured, 5)
    out.update(extra)
    return out



def dev_equal(torch, a, b, rows=32768):
    """torch.equal in slices of `rows` blocks: the comparison's temporary is as large as its operands, and at configs[1]'s
    full size (64 GiB a side) the device has no room for a third copy"""
    for i in range(0, a.shape[0], rows):
        if not torch.equal(a[i:i + rows], b[i:i + rows]):
            return False
    return True


def shard(nblocks_per_rank, rank, world):
    """Block index range of a rank (contiguous); weak scaling: every rank gets B."""
    lo = rank * nblocks_per_rank
    return lo, lo + nblocks_per_rank


def shard_strong(total_blocks, rank, world):
    """Contiguous shard of a fixed total (config c5): the first `total % world` ranks get one more."""
    per, extra = divmod(total_blocks, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def reduce_totals(torch, dist, dev, u_bytes, c_bytes, wall, distributed):
    """The only collectives of the run: SUM of {uncompressed, compressed} bytes, MAX of elapsed."""
    tot = torch.tensor([u_bytes, c_bytes], dtype=torch.float64, device=dev)
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        if os.environ.get("NXZ_BENCH_TRACE_COLLECTIVES") and dist.get_rank() == 0:
            print("collectives: backend %s, all_reduce SUM + MAX on %s, world %d" % (dist.get_backend(), tot.device.type, dist.get_world_size()), file=sys.stderr)
    return float(tot[0].item()), float(tot[1].item()), float(tmax.item())


GOLD = 0x9E3779B97F4A7C15
ALPHABET33 = b"abcdefghijklmnopqrstuvwxyz .,;!?\n"


def _i64(x):
    """a 64-bit pattern as the signed value torch's int64 holds"""
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >> 63 else x


def _mix(torch, x):
    """splitmix64's finaliser on int64 tensors (wrapping multiplies, logical shifts): a counter-based
    generator, so that block i depends on nothing but its own seed GOLD ^ i"""
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * _i64(0xBF58476D1CE4E5B9)
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * _i64(0x94D049BB133111EB)
    return x ^ ((x >> 31) & 0x1FFFFFFFF)


def _rand31(torch, seed, counter):
    """31 random bits per element: hash of (block seed, counter)"""
    step = _i64(counter * GOLD) if isinstance(counter, int) else counter * _i64(GOLD)
    return (_mix(torch, seed + step) >> 33) & 0x7FFFFFFF


def block_seeds(torch, dev, n, first_index):
    idx = torch.arange(first_index, first_index + n, device=dev, dtype=torch.int64)
    return _mix(torch, idx ^ _i64(GOLD)).unsqueeze(1)              # (n, 1): seed 0x9E3779B97F4A7C15 ^ i, whitened


def gen_text(torch, dev, n, first_index, length=BLOCK):
    """uniform text over a 33-symbol alphabet (test/test_utils.c:22-28,152-161), block i from its own seed"""
    alphabet = torch.tensor(list(ALPHABET33), dtype=torch.uint8, device=dev)
    seed = block_seeds(torch, dev, n, first_index)
    j = torch.arange(length, device=dev, dtype=torch.int64).unsqueeze(0)
    return alphabet[_rand31(torch, seed, j) % 33]


def gen_blocks(torch, dev, n, first_index, chunk=1024):
    """Seeded synthetic 64 KiB blocks, generated on the device.

    Block i (global index) uses the seed 0x9E3779B97F4A7C15 ^ i and nothing else.  Recipe = the
    reference's own generators: the first half of the block is uniform text over a 33-symbol
    alphabet (test/test_utils.c:22-28,152-161); the rest is filled the way samples/makedata.c:51-68
    does it: per block len_max in [10, 249] and dist_max in [1, 65536], then copy after copy of
    len in [16, len_max + 15] bytes from dist in [1, min(dist_max, bytes so far)] bytes back, byte
    by byte (a copy may overlap itself and may read earlier copies).  On the device a copy is a
    source index per byte, and copies of copies are resolved by pointer doubling.
    """
    out = torch.empty((n, BLOCK), dtype=torch.uint8, device=dev)
    half = BLOCK // 2
    K = half // 16                                                   # copies are >= 16 bytes long
    pos = torch.arange(BLOCK, device=dev, dtype=torch.int64).unsqueeze(0)
    k = torch.arange(K, device=dev, dtype=torch.int64).unsqueeze(0)
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        seed = block_seeds(torch, dev, m, first_index + c0)
        text = gen_text(torch, dev, m, first_index + c0, half)
        len_max = 10 + _rand31(torch, seed, 1 << 20) % 240           # (m, 1)
        dist_max = 1 + _rand31(torch, seed, (1 << 20) + 1) % 65536
        ln = 16 + _rand31(torch, seed, (2 << 20) + k) % len_max      # (m, K)
        start = half + torch.cumsum(ln, dim=1) - ln                  # where copy k begins
        dist = 1 + _rand31(torch, seed, (3 << 20) + k) % torch.minimum(dist_max, start)
        # the copy a position belongs to: count the copies that begin at or before it
        mark = torch.zeros((m, BLOCK + 1), dtype=torch.int32, device=dev)
        mark.scatter_(1, start.clamp(max=BLOCK), 1)
        run = (torch.cumsum(mark[:, :BLOCK], dim=1) - 1).clamp_(min=0).to(torch.int64)
        s = pos - torch.gather(dist, 1, run)                         # source index of every byte ...
        s = torch.where(pos < half, pos, s)                          # ... the text stands for itself
        del mark, run, ln, start, dist
        for _ in range(15):                                          # 2^15 hops: as deep as a chain of copies gets
            s = torch.gather(s, 1, s)
        out[c0:c0 + m] = torch.gather(text, 1, s)
        del text, s
    return out


def gen_mixed(torch, dev, n, first_index):
    """config c5: block i (global index) is zeros / 33-symbol text / text + makedata copies / random
    bytes by i mod 4, each from its own seed."""
    out = gen_blocks(torch, dev, n, first_index)
    idx = torch.arange(first_index, first_index + n, device=dev)
    kind = idx % 4
    out[kind == 0] = 0
    j = torch.arange(BLOCK, device=dev, dtype=torch.int64).unsqueeze(0)
    for c0 in range(0, n, 2048):
        m = min(2048, n - c0)
        k = kind[c0:c0 + m]
        sub = out[c0:c0 + m]
        if bool((k == 1).any()):
            sub[k == 1] = gen_text(torch, dev, m, first_index + c0)[k == 1]
        if bool((k == 3).any()):
            seed = block_seeds(torch, dev, m, first_index + c0)[k == 3]
            sub[k == 3] = ((_rand31(torch, seed, (5 << 20) + j) >> 7) & 0xFF).to(torch.uint8)
    return out


PMC_SOURCE = {}


def pmc_traffic(n_blocks, which="fht"):
    """HBM bytes per step from the committed PMC pass (profiles/*pmc_traffic_<which>.json, collected
    with separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same workload and corrected as
    MI355X_MICROARCH.md prescribes), scaled to this step's block count; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*pmc_traffic_%s.json" % which)))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        if d.get("block_bytes") != BLOCK:
            return None
        PMC_SOURCE[which] = os.path.basename(files[-1])
        return float(d["traffic_bytes_per_block"]) * n_blocks
    except (OSError, ValueError, KeyError):
        return None


def usable_cores():
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _cpu_run(bufs, mode, threads, budget_s):
    """oracle/nxz_bench.c harness on a list of byte strings: calibrates on a few, then runs a sample
    sized for about budget_s.  Returns (GiB/s of the bytes that matter, bytes in, bytes out, items)."""
    import oracle_lib as O
    L = O.lib()
    L.nxo_bench_run.restype = C.c_double
    L.nxo_bench_run.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                C.POINTER(C.c_uint64), C.POINTER(C.c_int)]

    def run(items):
        buf = b"".join(items)
        lens = np.array([len(b) for b in items], np.uint32)
        off = np.zeros(len(items), np.uint64)
        off[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        ob, bad = C.c_uint64(), C.c_int()
        t = L.nxo_bench_run(buf, off.ctypes.data, lens.ctypes.data, len(items), threads, mode, C.byref(ob), C.byref(bad))
        if bad.value:
            raise SystemExit("cpu baseline: zlib reported %d errors" % bad.value)
        return t, int(lens.sum()), ob.value

    ncal = min(len(bufs), 4 * threads)
    t, _, _ = run(bufs[:ncal])
    n = int(max(ncal, ncal * budget_s / max(t, 1e-4)))
    items = [bufs[i % len(bufs)] for i in range(n)]
    t, nin, nout = run(items)
    moved = nout if mode == 2 else nin                       # inflate: uncompressed bytes OUT
    return moved / t / 2.0 ** 30, nin, nout, n


def cpu_baseline_deflate(blocks, strategy, budget_s):
    """The reference's CPU path for this call is system zlib (lib/sw_zlib.c:283-324 dlopens it):
    deflate level 1, raw, fixed or default strategy, per 64 KiB block with deflateReset between
    blocks, T = 1 and T = all usable cores; the oracle port as a side key."""
    cores = usable_cores()
    zmode = 1 if strategy == "fixed" else 3
    pmode = 0 if strategy == "fixed" else 4
    z1, _, _, n1 = _cpu_run(blocks, zmode, 1, budget_s / 4)
    zt, nin, nout, nt = _cpu_run(blocks, zmode, cores, budget_s / 4)
    pt, pin, pout, npt = _cpu_run(blocks, pmode, cores, budget_s / 2)
    return {
        "value": round(zt, 4), "unit": "GiB/s", "cores": cores, "kind": "reference",
        "what": "system zlib 1.2.11 deflate level 1 (%s strategy), raw, deflateReset per block: the library the "
                "reference's software path dlopens (lib/sw_zlib.c:283-324)" % strategy,
        "sample": "%d of the same blocks on %d pthreads (oracle/nxz_bench.c)" % (nt, cores),
        "one_thread_GiB_s": round(z1, 4), "zlib_ratio": round(nin / nout, 4),
        "oracle_port_GiB_s": round(pt, 4), "oracle_port_ratio": round(pin / pout, 4),
        "oracle_port_sample": "%d blocks, oracle/nxz_lz77.c + %s" % (npt, "fixed code" if strategy == "fixed" else "nxo_dhtgen per block"),
    }


def verify_sample(src_rows, out_rows, tpbc, k=48):
    """zlib inflates the first k outputs to the inputs (outside the timed region)."""
    import zlib
    for i in range(min(k, len(src_rows))):
        z = zlib.decompressobj(-15)
        if z.decompress(out_rows[i][:tpbc[i]].tobytes()) != src_rows[i] or not z.eof:
            raise SystemExit("ROUND TRIP FAILURE: block %d" % i)


def oracle_parity(blocks, out_rows, tpbc, dynamic, k=32):
    """cpu_baseline leg: the engine's first k outputs equal the oracle's, byte for byte."""
    import oracle_lib as O
    k = min(k, len(blocks))
    for i in range(k):
        b = blocks[i]
        if dynamic:
            tok, nt = O.lz77(b)
            ll, d = O.counts(tok, nt)
            dht, dhtlen = O.dhtgen(ll, d)
            exp, bits = O.deflate_dynamic(b, dht, dhtlen)
        else:
            exp, bits = O.deflate_fixed(b)
        if out_rows[i][:tpbc[i]].tobytes() != exp:
            raise SystemExit("PARITY FAILURE: block %d differs from the oracle" % i)
    return k


def stage_times(eng):
    ms = (C.c_double * 3)()
    n = C.c_uint()
    eng.L.nxz_ctx_stage_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint)]
    eng.L.nxz_ctx_stage_ms(eng.ctx, ms, C.byref(n))
    return [ms[0], ms[1], ms[2]], n.value


def timed_compress(torch, eng, fc, jobs, n, results, steps, warmup):
    """(ms per step by torch events, [lz77, dhtgen, entropy] ms per step by the engine's own HIP events
    on the launch stream, launches per step)"""
    eng.L.nxz_ctx_stage_timing.argtypes = [C.c_void_p, C.c_int]
    for _ in range(warmup):
        eng.compress(fc, jobs, n, results=results)
    torch.cuda.synchronize(eng.dev)
    eng.L.nxz_ctx_stage_timing(eng.ctx, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        eng.compress(fc, jobs, n, results=results)
    e1.record()
    torch.cuda.synchronize(eng.dev)
    eng.L.nxz_ctx_stage_timing(eng.ctx, 0)
    st, launches = stage_times(eng)
    return e0.elapsed_time(e1) / steps, [x / steps for x in st], launches // max(steps, 1)


def what_binds(kernel_prefix="nxzl77::lz77_kernel<true, false", leg="dhtgen"):
    """The path moves few bytes per instruction: what binds the LZ77 kernel is the issue of vector instructions and
    the LDS pipe, from the latest committed counter pass (profiles/r*_pmc_counters.json, tools/prof_report.py: keys
    "<kernel> [<leg>]", figures per unit = per 64 KiB block)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_counters.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        for key, k in d.items():
            if not (key.startswith(kernel_prefix) and key.endswith("[%s]" % leg)):
                continue
            try:
                valu = k.get("SQ_INSTS_VALU_per_unit", k.get("SQ_INSTS_VALU_per_job"))
                cyc = k.get("cu_cycles_per_unit", k.get("cu_cycles_per_job"))
                split = k.get("wave_cycles_split", {})
                return {"kernel": key, "source": os.path.basename(f),
                        "vector_issue_frac": round(valu * 4 / 4 / cyc, 3), "lds_busy_frac": k.get("lds_busy_share_of_kernel_time"),
                        "lds_bank_conflict_share": k.get("lds_bank_conflict_share_of_lds_cycles"),
                        "wave_cycles_active": split.get("SQ_ACTIVE_INST_ANY"), "wave_cycles_waiting": split.get("SQ_WAIT_ANY"),
                        "note": "from the committed counter pass of the same workload, not this run; a wave64 vector instruction occupies one "
                                "of the CU's four SIMDs for four cycles: wave-instructions x 4 / 4 SIMDs / CU-cycles per block"}
            except (TypeError, ZeroDivisionError):
                continue
    return None


def roofline(u_bytes, c_bytes, stage_ms, launches, traffic, kernel, peak_measured=None, leg="dhtgen"):
    """algorithmic bytes U + C over the time of the dominant kernel (the LZ77 kernel:
    it reads U; the entropy kernel writes C and is accounted with it: both are needed to move U + C),
    so achieved = (U + C) / (lz77 + dhtgen + entropy kernel time), measured by HIP events on the
    launch stream around every launch of the timed region."""
    kern_ms = sum(stage_ms)
    achieved = (u_bytes + c_bytes) / (kern_ms * 1e-3) / 1e9
    return roof(achieved, traffic, kernel, peak_measured,
                what_binds=what_binds("nxzl77::lz77_kernel<false, true", "fht") if leg == "fht" else what_binds(),
                kernel_ms=round(kern_ms, 3), lz77_ms=round(stage_ms[0], 3), dhtgen_ms=round(stage_ms[1], 3),
                entropy_ms=round(stage_ms[2], 3), launches_per_step=launches,
                avg_lz77_launch_ms=round(stage_ms[0] / max(launches, 1), 4),
                algorithmic_bytes_per_step=u_bytes + c_bytes)


def run_corpus(torch, dist, args, rank, world, dev, distributed, pkg, eng):
    """The headline: real data (the metric's corpus), exact dynamic-Huffman table per block (BASELINE configs[2])."""
    import zlib
    import corpus
    name, blocks, report = corpus.load(BLOCK)
    uniq = len(blocks)
    rep = max(1, -(-args.corpus_jobs // uniq))
    n = uniq * rep                                             # jobs of THIS rank (every rank its own: weak scaling)
    host = np.zeros((uniq, BLOCK), np.uint8)
    lens_u = np.array([len(b) for _, _, b in blocks], np.uint32)
    for i, (_, _, b) in enumerate(blocks):
        host[i, :len(b)] = np.frombuffer(b, np.uint8)
    src = torch.from_numpy(host).to(dev).repeat(rep, 1)
    lens = np.tile(lens_u, rep)
    dst = torch.empty((n, STRIDE_OUT), dtype=torch.uint8, device=dev)
    jobs = eng.jobs_strided(src, BLOCK, lens, dst, STRIDE_OUT, STRIDE_OUT)
    res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    eng.L.nxz_ctx_stage_timing.argtypes = [C.c_void_p, C.c_int]

    def step():
        eng.compress(pkg.FC_COMPRESS_DHTGEN, jobs, n, results=res)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    eng.L.nxz_ctx_stage_timing(eng.ctx, 1)                 # HIP events on the launch stream around every kernel
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    eng.L.nxz_ctx_stage_timing(eng.ctx, 0)
    st, launches = stage_times(eng)
    st = [x / args.steps for x in st]
    launches //= args.steps

    r = eng.results_to_host(res)
    if not ((r["cc"] == 0) | (r["cc"] == 64)).all():
        raise SystemExit("corpus: engine reported errors %s" % np.unique(r["cc"]))
    # a block that did not shrink (cc 64) leaves the library as a stored block, 5 bytes of header (lib/nx_deflate.c:1274-1282)
    out_bytes = np.where(r["cc"] == 64, lens.astype(np.int64) + 5, r["tpbc"].astype(np.int64))
    u_bytes, c_bytes = float(lens.astype(np.float64).sum()), float(out_bytes.astype(np.float64).sum())
    # every output inflates back (on the device, full size) -- on every rank
    back = torch.zeros((n, BLOCK), dtype=torch.uint8, device=dev)
    jobs2 = eng.jobs_strided(dst, STRIDE_OUT, r["tpbc"].astype(np.uint32), back, BLOCK, BLOCK)
    r2 = eng.results_to_host(eng.decompress(jobs2, n))
    if not (bool((r2["cc"] == 0).all()) and bool((r2["tpbc"] == lens).all()) and dev_equal(torch, back, src)):
        raise SystemExit("corpus: ROUND TRIP FAILURE on rank %d (inflate of the deflate output != source)" % rank)
    del back
    tot_u, tot_c, wall_max = reduce_totals(torch, dist, dev, u_bytes, c_bytes, wall, distributed)
    if rank != 0:
        return
    out_u = dst[:uniq].cpu().numpy()
    raw = [b for _, _, b in blocks]
    verify_sample(raw, out_u, r["tpbc"], k=uniq if uniq <= 400 else 400)      # ... and a sample through zlib
    # ratio per class against zlib -1 (default strategy), on the unique blocks
    per = {}
    for i, (cls, _, b) in enumerate(blocks):
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        z = len(c.compress(b) + c.flush())
        a = per.setdefault(cls, [0, 0, 0])
        a[0] += len(b); a[1] += int(out_bytes[i]); a[2] += z
    classes = {k: {"bytes": v[0], "ratio": round(v[0] / v[1], 4), "zlib1_ratio": round(v[0] / v[2], 4),
                   "vs_zlib1": round(v[2] / v[1], 4)} for k, v in sorted(per.items())}
    tot = [sum(v[i] for v in per.values()) for i in range(3)]
    # the rate by class (the kernel's time depends on the data), outside the timed region: the headline's own buffers and job
    # count filled with the full blocks of one class at a time, so that a class is launched in the same chunks as the headline
    # (round 4 timed 4096 jobs in one 8192-job chunk: the tail of that one short launch read 4 % low)
    src_u = src[:uniq].clone()
    m = n if args.class_jobs <= 0 else min(args.class_jobs, n)
    full_lens = np.full(m, BLOCK, np.uint32)
    for cls in classes:
        idx = [i for i, (c, _, b) in enumerate(blocks) if c == cls and len(b) == BLOCK]
        if not idx:
            continue
        sel = torch.from_numpy(np.array([idx[i % len(idx)] for i in range(m)], np.int64)).to(dev)
        torch.index_select(src_u, 0, sel, out=src[:m])
        jobs_c = eng.jobs_strided(src[:m], BLOCK, full_lens, dst[:m], STRIDE_OUT, STRIDE_OUT)
        ms_c, st_c, _ = timed_compress(torch, eng, pkg.FC_COMPRESS_DHTGEN, jobs_c, m, res, 2, 1)
        classes[cls]["GiB_s"] = round(m * BLOCK / (ms_c * 1e-3) / 2.0 ** 30, 1)
        classes[cls]["lz77_ms"] = round(st_c[0], 3); classes[cls]["jobs_timed"] = m
        del jobs_c, sel
    del src_u
    # what these rates say about the metric's own corpus: the twelve Silesia files by their published sizes, each at the rate
    # of the class of this corpus that stands in for it (tests/corpus.py: SILESIA_AS_FALLBACK_CLASS)
    silesia = None
    if name != "silesia" and all(classes.get(c, {}).get("GiB_s") for c in set(corpus.SILESIA_AS_FALLBACK_CLASS.values())):
        tot_b = float(sum(corpus.SILESIA_BYTES.values()))
        secs = sum(corpus.SILESIA_BYTES[f] / classes[corpus.SILESIA_AS_FALLBACK_CLASS[f]]["GiB_s"] for f in corpus.SILESIA_BYTES)
        silesia = {"value": round(tot_b / secs, 1), "unit": "GiB/s", "how": "12 files x published size / rate of the stand-in class (time-weighted)",
                   "class_of_file": corpus.SILESIA_AS_FALLBACK_CLASS}
    peak_m = copy_peak_gbs(torch, dev, eng)
    line = {
        "metric": "GiB/s uncompressed in (deflate), real-data corpus in 64 KiB blocks, exact dynamic-Huffman table per block",
        "value": round(tot_u * args.steps / wall_max / 2.0 ** 30, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(wall_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8",
        "data": "Silesia ($SILESIA_DIR, sha256-checked)" if name == "silesia" else
                "real files of this image (recorded fallback corpus: Silesia is not on the box), replicated",
        "corpus_skipped": report.get("skipped", []) if isinstance(report, dict) else [],
        "data_note": "the %d unique blocks (%.0f MB) are replicated %d x: the SOURCE set fits the 256 MB Infinity Cache, which does not matter at "
                     "1-2 %% of the HBM roof (the kernels are bound by instruction issue, roofline.what_binds)" % (uniq, uniq * BLOCK / 1e6, rep),
        "config": {"workload": "BASELINE configs[2]: dynamic-Huffman deflate (COMPRESS_DHTGEN: LZ77 + histogram, dhtgen on the device, "
                               "entropy) of the %s corpus cut at 64 KiB, %d jobs per GPU (%d unique blocks), device resident" % (name, n, uniq),
                   "corpus": name, "corpus_files": report, "unique_blocks": uniq, "jobs_per_gpu": n, "block_bytes": BLOCK,
                   "ratio": round(tot[0] / tot[1], 4), "zlib1_ratio": round(tot[0] / tot[2], 4), "ratio_vs_zlib1": round(tot[2] / tot[1], 4),
                   "min_class_vs_zlib1": min(v["vs_zlib1"] for v in classes.values()),
                   "min_class_GiB_s": min((v["GiB_s"] for v in classes.values() if "GiB_s" in v), default=None),
                   "silesia_weighted_GiB_s": silesia, "classes": classes,
                   "roundtrip_bit_exact": True, "zlib_inflated_sample": min(uniq, 400), "parallelism": "shard%d" % world},
        "roofline": roofline(u_bytes, c_bytes, st, launches, pmc_traffic(n, "dhtgen"),
                             "nxzl77::lz77_kernel<true, false, false> (dominant) + nxzd::dhtgen_kernel + nxze::encode_kernel<true, false>", peak_m),
    }
    if world > 1:
        # a scaling line: the legs that belong to one GPU's report are left out, and the line says so
        line["legs_skipped"] = ["cpu_baseline", "inflate_zlib6", "inflate_stream", "api", "c2", "c5"]
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_deflate(raw, "default", 10.0)
        line["cpu_baseline"]["parity_checked_blocks"] = oracle_parity(raw, out_u, r["tpbc"], True, k=64)
    if world == 1:
        del src, dst, jobs, res
        torch.cuda.empty_cache()
        # inflate of zlib -6 streams of the same corpus (what configs[3] feeds the inflate engine)
        if not args.no_inflate:
            line["inflate_zlib6"] = inflate_leg(torch, eng, pkg, raw, rep, args)
            line["inflate_stream"] = stream_leg(torch, eng, raw, args)
            # the same leg on the classes the corpus had up to round 4 (without the image-like and the packed kind, which zlib
            # turns into long all-literal and stored blocks: nothing there for decoders at neighbouring bits to fall in step on)
            raw_r04 = [b for cls, _, b in blocks if cls not in ("image", "packed")]
            if len(raw_r04) != len(raw):
                s4 = stream_leg(torch, eng, raw_r04, args)
                line["inflate_stream"]["without_image_and_packed_classes"] = {k: s4[k] for k in ("value", "unit", "ms", "pieces", "bit_exact") if k in s4}
                z4 = inflate_leg(torch, eng, pkg, raw_r04, max(1, -(-args.corpus_jobs // len(raw_r04))), dict_args(args, no_cpu_baseline=True))
                line["inflate_zlib6"]["without_image_and_packed_classes"] = {k: z4[k] for k in ("value", "unit", "ms_per_pass", "streams", "bit_exact") if k in z4}
        if not args.no_api:
            raw_r04 = [b for cls, _, b in blocks if cls not in ("image", "packed")]
            line["api"] = api_leg(raw, args, raw_r04=raw_r04 if len(raw_r04) != len(raw) else None)
        if not args.no_c2:
            torch.cuda.empty_cache()
            # configs[1] at its stated size (2^20 blocks: 136 GiB of buffers + scratch) when the device has the room, else 2^18
            n_c2 = args.blocks or c2_default_blocks(torch, dev)
            line["c2"] = c2_measure(torch, dist, args, 0, 1, dev, False, pkg, eng, n_c2, max(2, args.steps // 2), 1)
        if not args.no_c5:
            torch.cuda.empty_cache()
            line["c5"] = c5_measure(torch, dist, args, 0, 1, dev, False, pkg, eng, 163840, max(2, args.steps // 2), 1)
    print(json.dumps(line), flush=True)


def api_leg(raw, args, mib=256, nthreads=16, raw_r04=None):
    """Through the reference's own API (libnxz_amd.so: nx_compress2 / nx_uncompress, HOST buffers, PCIe and
    host copies inside the timed region): one call over `mib` MiB, and `nthreads` threads each compressing
    64 KiB buffers one call after the other (the shape of the reference's samples/compdecomp_th.c)."""
    import ctypes as C
    import threading
    import zlib
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    import zstream as Z
    L = Z.load("gpu")
    base = b"".join(raw)
    data = (base * ((mib << 20) // len(base) + 1))[:mib << 20]
    out = {"note": "host buffers in, host buffers out; level 1 (no history between jobs, lib/nx_deflate.c:654-680)"}
    cap = C.c_ulong(L.nx_compressBound(len(data)))
    dst = C.create_string_buffer(cap.value)
    best = 1e9
    for it in range(4):
        cap.value = len(dst)
        t = time.perf_counter()
        rc = L.nx_compress2(dst, C.byref(cap), data, len(data), 1)
        dt = time.perf_counter() - t
        if rc != 0:
            return {"error": "nx_compress2 returned %d" % rc}
        if it:
            best = min(best, dt)
    comp = dst.raw[:cap.value]
    if zlib.decompress(comp) != data:
        raise SystemExit("api leg: zlib does not read nx_compress2's stream back")
    out["compress2_one_shot"] = {"value": round(len(data) / best / 2.0 ** 30, 2), "unit": "GiB/s uncompressed in", "MiB": mib,
                                 "ms": round(best * 1e3, 2), "ratio": round(len(data) / len(comp), 3), "zlib_reads_it_back": True}
    def inflate_side(data):
        out = {}
        z6 = zlib.compress(data, 6)
        back = C.create_string_buffer(len(data))
        best = 1e9
        for it in range(4):
            n = C.c_ulong(len(data))
            t = time.perf_counter()
            rc = L.nx_uncompress(back, C.byref(n), z6, len(z6))
            dt = time.perf_counter() - t
            if rc != 0 or n.value != len(data):
                return dict(out, error="nx_uncompress returned %d" % rc)
            if it:
                best = min(best, dt)
        if back.raw != data:
            raise SystemExit("api leg: nx_uncompress of a zlib -6 stream differs from the source")
        out["uncompress_one_shot"] = {"value": round(len(data) / best / 2.0 ** 30, 2), "unit": "GiB/s uncompressed out", "MiB": mib,
                                      "ms": round(best * 1e3, 2), "stream": "zlib level 6, one zlib stream"}
        # the same data as a .gz through inflate() with avail_in / avail_out steps of 64 KiB and 1 MiB
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        gz = co.compress(data[:64 << 20]) + co.flush()
        gsrc = C.create_string_buffer(gz, len(gz))
        gdst = C.create_string_buffer(1 << 20)
        for step in (64 << 10, 1 << 20):
            best = 1e9
            for it in range(2):
                st = Z.ZStream()
                if L.nx_inflateInit2_(C.byref(st), 31, Z.VERSION, C.sizeof(Z.ZStream)) != 0:
                    return dict(out, error="nx_inflateInit2_ failed")
                fed = total = 0
                crc = 0
                rc = 0
                t = time.perf_counter()
                while rc != Z.Z_STREAM_END:
                    if st.avail_in == 0 and fed < len(gz):
                        k = min(step, len(gz) - fed)
                        st.next_in = C.addressof(gsrc) + fed
                        st.avail_in = k
                        fed += k
                    st.next_out = C.addressof(gdst)
                    st.avail_out = step
                    rc = L.nx_inflate(C.byref(st), Z.Z_NO_FLUSH)
                    if rc not in (Z.Z_OK, Z.Z_STREAM_END, Z.Z_BUF_ERROR) or (rc == Z.Z_BUF_ERROR and st.avail_out == step and fed == len(gz)):
                        return dict(out, error="nx_inflate in steps returned %d" % rc)
                    total += step - st.avail_out
                dt = time.perf_counter() - t
                crc = st.adler
                L.nx_inflateEnd(C.byref(st))
                best = min(best, dt)
            if total != (64 << 20) or crc != zlib.crc32(data[:64 << 20]):
                raise SystemExit("api leg: nx_inflate in steps of %d made %d bytes, crc %08x" % (step, total, crc))
            out["inflate_in_steps_%dKiB" % (step >> 10)] = {"value": round(total / best / 2.0 ** 30, 3), "unit": "GiB/s uncompressed out", "MiB": 64,
                                                            "ms": round(best * 1e3, 1), "stream": "zlib level 6 .gz, avail_in = avail_out = the step"}
        return out

    side = inflate_side(data)
    if "error" in side:
        return dict(out, **side)
    out.update(side)
    if raw_r04 is not None:
        # the same on the classes the corpus had up to round 4 (see run_corpus: the image-like and packed classes are all-literal and stored blocks under zlib -6)
        b4 = b"".join(raw_r04)
        side4 = inflate_side((b4 * ((mib << 20) // len(b4) + 1))[:mib << 20])
        for k, v in side4.items():
            if isinstance(v, dict) and k in out:
                out[k]["without_image_and_packed_classes"] = v.get("value")
    blocks = [data[i * BLOCK:(i + 1) * BLOCK] for i in range(min(1024, len(data) // BLOCK))]

    def worker(res, k):
        c = C.c_ulong()
        d = C.create_string_buffer(L.nx_compressBound(BLOCK))
        tot = 0
        for b in blocks:
            c.value = len(d)
            if L.nx_compress2(d, C.byref(c), b, len(b), 1) != 0:
                tot = -1
                break
            tot += c.value
        res[k] = tot

    for T in (1, nthreads):
        res = [0] * T
        th = [threading.Thread(target=worker, args=(res, k)) for k in range(T)]
        t = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t
        if min(res) < 0:
            return dict(out, error="nx_compress2 failed in a thread")
        out["threads_%d_x_64KiB" % T] = {"value": round(T * len(blocks) * BLOCK / dt / 2.0 ** 30, 3), "unit": "GiB/s uncompressed in, all threads",
                                         "us_per_call": round(dt / len(blocks) * 1e6, 1), "calls_per_thread": len(blocks)}
    # the same harness shape at a size where the engine is meant to win: every thread 1 MiB buffers, one call per buffer,
    # compress and decompress (the whole sizes x threads table: tools/api_sweep.py, profiles/r03_api_sweep.txt)
    # ... and at 16 MiB, beyond what one merge takes from a caller: compress calls go through the merges in slices, the one-stream
    # workspaces of the inflate side stay while other callers work (round 5's last session; before: 10-14 and 7 GiB/s)
    for size, ncalls, nz in ((1 << 20, 48, 16), (16 << 20, 12, 4)):
        big = [data[i * size:(i + 1) * size] for i in range(min(ncalls, len(data) // size))]
        if len(big) < 2:
            continue
        zbig = [zlib.compress(b, 6) for b in big[:nz]]

        # (every thread's first call -- which makes the thread's buffer set and streams -- before the clock, as the reference's harness has it,
        # samples/compdecomp_th.c:161-185; what those first calls take is reported beside the rate: first_calls_ms)
        def worker_big(res, k, inflate, gate, t_first, size=size, big=big, zbig=zbig):
            tot = 0
            if inflate:
                d = C.create_string_buffer(size)
                n = C.c_ulong(size)
                t = time.perf_counter()
                ok = L.nx_uncompress(d, C.byref(n), zbig[k % len(zbig)], len(zbig[k % len(zbig)])) == 0
                t_first[k] = time.perf_counter() - t
                gate.wait(timeout=120)
                for i in range(len(big)):
                    z = zbig[(i + k) % len(zbig)]
                    n = C.c_ulong(size)
                    if not ok or L.nx_uncompress(d, C.byref(n), z, len(z)) != 0 or n.value != size:
                        tot = -1
                        break
                    tot += n.value
            else:
                c = C.c_ulong()
                d = C.create_string_buffer(L.nx_compressBound(size))
                c.value = len(d)
                t = time.perf_counter()
                ok = L.nx_compress2(d, C.byref(c), big[k % len(big)], size, 1) == 0
                t_first[k] = time.perf_counter() - t
                gate.wait(timeout=120)
                for i in range(len(big)):
                    b = big[(i + k) % len(big)]
                    c.value = len(d)
                    if not ok or L.nx_compress2(d, C.byref(c), b, len(b), 1) != 0:
                        tot = -1
                        break
                    tot += len(b)
            res[k] = tot

        for inflate in (False, True):
            res = [0] * nthreads
            t_first = [0.0] * nthreads
            clock = [0.0]
            gate = threading.Barrier(nthreads, action=lambda: clock.__setitem__(0, time.perf_counter()))
            def guarded(*a, gate=gate, res=res):
                # (a worker that raises before the gate must not leave the others waiting there)
                try:
                    worker_big(*a)
                except Exception:                                           # noqa: BLE001 -- any failure: the leg reports it
                    gate.abort()
                    res[a[1]] = -1
            th = [threading.Thread(target=guarded, args=(res, k, inflate, gate, t_first)) for k in range(nthreads)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            dt = time.perf_counter() - clock[0]
            if min(res) < 0:
                return dict(out, error="a %d MiB call failed in a thread" % (size >> 20))
            out["threads_%d_x_%dMiB_%s" % (nthreads, size >> 20, "uncompress" if inflate else "compress2")] = {
                "value": round(sum(res) / dt / 2.0 ** 30, 3), "unit": "GiB/s uncompressed, all threads", "us_per_call": round(dt / len(big) * 1e6, 1),
                "calls_per_thread": len(big), "first_calls_ms": round(max(t_first) * 1e3, 1), "warm": True,
                "value_incl_first_calls": round(sum(res) * (len(big) + 1) / len(big) / (dt + max(t_first)) / 2.0 ** 30, 3)}
    # what these calls left idle on the device (the one-stream workspaces of sixteen callers, the hosts' lanes) goes back before the legs
    # that size themselves by the device's free memory (c2 at its stated 2^20 blocks wants 212 GiB free)
    try:
        E = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "power-gzip_amd", "libnxz_engine.so"))
        E.nxz_trim.restype = C.c_size_t
        out["trimmed_MiB"] = int(E.nxz_trim()) >> 20
    except (OSError, AttributeError):
        pass
    if not args.no_cpu_baseline:
        t = time.perf_counter()
        zlib.compress(data[:32 << 20], 1)
        out["cpu_baseline"] = {"value": round((32 << 20) / (time.perf_counter() - t) / 2.0 ** 30, 4), "unit": "GiB/s uncompressed in", "cores": 1,
                               "kind": "reference", "what": "system zlib compress(level 1) of 32 MiB of the same data, one thread"}
    return out


def stream_leg(torch, eng, raw, args, mib=256):
    """BASELINE configs[3]: ONE zlib-made deflate stream (the corpus repeated to `mib` MiB, zlib -6),
    inflated by block-boundary speculation (nxz_inflate_stream), bit-exact check; zlib on one host
    thread beside it (a single stream does not spread over threads there either)."""
    import zlib
    data = (b"".join(raw) * ((mib << 20) // sum(len(b) for b in raw) + 1))[:mib << 20]
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    src = torch.from_numpy(np.frombuffer(comp, np.uint8).copy()).to(eng.dev)
    dst = torch.zeros(len(data) + 4096, dtype=torch.uint8, device=eng.dev)
    rc, info = eng.inflate_stream(src, len(comp), dst)
    if rc != 0:
        return {"error": "nxz_inflate_stream declined the stream (%d)" % rc}
    torch.cuda.synchronize(eng.dev)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        rc, info = eng.inflate_stream(src, len(comp), dst)
        torch.cuda.synchronize(eng.dev)
        best = min(best, time.perf_counter() - t0)
    ok = rc == 0 and info["out_len"] == len(data) and info["crc"] == zlib.crc32(data) and bool(
        torch.equal(dst[:len(data)], torch.from_numpy(np.frombuffer(data, np.uint8).copy()).to(eng.dev)))
    if not ok:
        raise SystemExit("inflate_stream leg: the stream did not inflate to its source")
    out = {"value": round(len(data) / best / 2.0 ** 30, 3), "unit": "GiB/s uncompressed out, one stream, device resident",
           "ms": round(best * 1e3, 2), "stream": "%d MiB of the corpus as ONE raw deflate stream, zlib level 6 (%.1f MiB compressed)" % (mib, len(comp) / 2.0 ** 20),
           "pieces": info["pieces"], "bit_exact": True,
           "roofline": roof((len(data) + len(comp)) / best / 1e9, pmc_traffic(len(data) / BLOCK, "inflate_stream"),
                            "find_blocks + inflate of the pieces into 16-bit elements + window chain + resolve + checksums, wall clock",
                            copy_peak_gbs(torch, eng.dev, eng))}
    if not args.no_cpu_baseline:
        t0 = time.perf_counter()
        zlib.decompress(comp, -15)
        out["cpu_baseline"] = {"value": round(len(data) / (time.perf_counter() - t0) / 2.0 ** 30, 4), "unit": "GiB/s uncompressed out", "cores": 1,
                               "kind": "reference", "what": "system zlib inflate of the same stream, one thread"}
    return out


def dict_args(args, **over):
    """a copy of the parsed arguments with some of them replaced"""
    import copy
    a = copy.copy(args)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def inflate_leg(torch, eng, pkg, raw, rep, args):
    """zlib -6 raw streams of the corpus blocks through the batched inflate engine."""
    import zlib
    streams = []
    for b in raw:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        streams.append(c.compress(b) + c.flush())
    uniq = len(raw)
    cstride = (max(len(s) for s in streams) + 64 + 15) & ~15
    host = np.zeros((uniq, cstride), np.uint8)
    for i, s in enumerate(streams):
        host[i, :len(s)] = np.frombuffer(s, np.uint8)
    n = uniq * rep
    src = torch.from_numpy(host).to(eng.dev).repeat(rep, 1)
    clen = np.tile(np.array([len(s) for s in streams], np.uint32), rep)
    ulen = np.tile(np.array([len(b) for b in raw], np.uint32), rep)
    dst = torch.zeros((n, BLOCK), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, cstride, clen, dst, BLOCK, BLOCK)
    res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
    eng.decompress(jobs, n, results=res)
    torch.cuda.synchronize(eng.dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(2, args.steps // 2)
    e0.record()
    for _ in range(reps):
        eng.decompress(jobs, n, results=res)
    e1.record()
    torch.cuda.synchronize(eng.dev)
    ms = e0.elapsed_time(e1) / reps
    r = eng.results_to_host(res)
    got = dst[:uniq].cpu().numpy()
    ok = bool((r["cc"] == 0).all()) and bool((r["tpbc"] == ulen).all())
    for i, b in enumerate(raw):
        ok = ok and got[i, :len(b)].tobytes() == b and int(r["crc"][i]) == zlib.crc32(b)
    if not ok:
        raise SystemExit("inflate leg: zlib -6 streams did not inflate to their sources")
    u, cb = float(ulen.astype(np.float64).sum()), float(clen.astype(np.float64).sum())
    # (the engine's route for streams that bring tables, nxz_engine.cpp batch_decompress: a stream per workgroup at every batch size)
    wg_kernel = "nxzw::inflate_wg_kernel<false> (a stream per workgroup: source, output and tables in LDS) + nxzl::cksum_kernel"
    out = {"value": round(u / (ms * 1e-3) / 2.0 ** 30, 3), "unit": "GiB/s uncompressed out", "ms_per_pass": round(ms, 3),
           "streams": n, "made_by": "zlib level 6, raw deflate, one stream per 64 KiB block", "bit_exact": True,
           "roofline": roof((u + cb) / (ms * 1e-3) / 1e9, pmc_traffic(n, "inflate_wg"), wg_kernel, copy_peak_gbs(torch, eng.dev, eng),
                            kernel_ms=round(ms, 3))}
    # the same call on the first k streams of the batch: what a caller with fewer streams at hand gets
    by_size = {}
    for k in (256, 1024, 4096, 16384, 65536):
        if k >= n:
            continue
        eng.decompress(jobs, k, results=res)
        torch.cuda.synchronize(eng.dev)
        e0.record()
        for _ in range(3):
            eng.decompress(jobs, k, results=res)
        e1.record()
        torch.cuda.synchronize(eng.dev)
        mk = e0.elapsed_time(e1) / 3
        rk = eng.results_to_host(res)
        if not (bool((rk["cc"][:k] == 0).all()) and bool((rk["tpbc"][:k] == ulen[:k]).all())):
            raise SystemExit("inflate leg: a batch of %d streams did not inflate" % k)
        by_size[str(k)] = {"value": round(float(ulen[:k].astype(np.float64).sum()) / (mk * 1e-3) / 2.0 ** 30, 3), "ms_per_pass": round(mk, 3)}
        if k == 65536:
            by_size[str(k)]["roofline"] = roof((float(ulen[:k].astype(np.float64).sum()) + float(clen[:k].astype(np.float64).sum())) / (mk * 1e-3) / 1e9,
                                               pmc_traffic(k, "inflate_wg"), wg_kernel, copy_peak_gbs(torch, eng.dev, eng), kernel_ms=round(mk, 3))
    if by_size:
        out["by_batch_size"] = dict(by_size, unit="GiB/s uncompressed out", kernel=wg_kernel)
    if not args.no_cpu_baseline:
        cores = usable_cores()
        z1, _, _, _ = _cpu_run(streams, 2, 1, 2.0)
        zt, _, _, nt = _cpu_run(streams, 2, cores, 2.0)
        out["cpu_baseline"] = {"value": round(zt, 4), "unit": "GiB/s uncompressed out", "cores": cores, "kind": "reference",
                               "what": "system zlib inflate of the same streams (the reference's software path)",
                               "sample": "%d streams on %d pthreads" % (nt, cores), "one_thread_GiB_s": round(z1, 4)}
    return out


def c5_prepare(torch, eng, pkg, src):
    """One step of BASELINE configs[4] on the mixed blocks `src` (n x 64 KiB, device): returns (step, info).
    step() = FHT compress + WRAP of what does not shrink + decompress (WRAP back for the stored ones) +
    compare on the device; info: buffers, the first pass' results, the indices of the stored blocks and
    the device flag that counts mismatching passes."""
    dev = src.device
    n = src.shape[0]
    comp = torch.empty((n, STRIDE_OUT), dtype=torch.uint8, device=dev)
    back = torch.empty((n, BLOCK), dtype=torch.uint8, device=dev)
    lens = np.full(n, BLOCK, np.uint32)
    jobs = eng.jobs_strided(src, BLOCK, lens, comp, STRIDE_OUT, STRIDE_OUT)
    res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    # blocks that do not shrink (every fourth: random bytes) complete with CC 64; they are stored
    # through the WRAP function code, the fallback the library applies per job
    # (lib/nx_deflate.c:1274-1282,1763-1790).  Which ones: known after the first pass, the job
    # arrays of the wrap and of the decompress pass are built once (outside the timed region)
    eng.compress(pkg.FC_COMPRESS_FHT, jobs, n, results=res)
    r = eng.results_to_host(res).copy()
    if not ((r["cc"] == 0) | (r["cc"] == 64)).all():
        raise SystemExit("engine reported errors: %s" % np.unique(r["cc"]))
    stored = np.nonzero(r["cc"] == 64)[0]
    ok = np.nonzero(r["cc"] == 0)[0]
    base_s, base_c, base_b = src.data_ptr(), comp.data_ptr(), back.data_ptr()

    def jobs_at(idx, a_base, a_stride, a_len, b_base, b_stride, b_cap):
        j = np.zeros(len(idx), pkg.JOB_DTYPE)
        j["src"] = np.uint64(a_base) + idx.astype(np.uint64) * np.uint64(a_stride)
        j["dst"] = np.uint64(b_base) + idx.astype(np.uint64) * np.uint64(b_stride)
        j["src_len"] = a_len
        j["dst_cap"] = b_cap
        j["in_adler"] = 1
        return eng.to_device(j)
    jw = jobs_at(stored, base_s, BLOCK, BLOCK, base_c, STRIDE_OUT, STRIDE_OUT) if len(stored) else None
    jd = jobs_at(ok, base_c, STRIDE_OUT, r["tpbc"][ok], base_b, BLOCK, BLOCK)
    ju = jobs_at(stored, base_c, STRIDE_OUT, BLOCK, base_b, BLOCK, BLOCK) if len(stored) else None   # stored blocks come back by WRAP too
    resw = torch.empty(max(1, len(stored)) * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    resd = torch.empty(max(1, len(ok)) * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    back64, src64 = back.view(torch.int64), src.view(torch.int64)

    def step():
        eng.compress(pkg.FC_COMPRESS_FHT, jobs, n, results=res)
        if jw is not None:
            eng.wrap(jw, len(stored), results=resw)
        eng.decompress(jd, len(ok), results=resd)
        if ju is not None:
            eng.wrap(ju, len(stored), results=resw)
        flag.add_((back64 != src64).any().to(torch.int32))       # compare on the device (eight bytes an element: an eighth of the temporary)

    c_bytes = float(r["tpbc"][ok].astype(np.float64).sum()) + len(stored) * float(BLOCK + 5)
    return step, {"comp": comp, "back": back, "results": r, "stored": int(len(stored)), "stored_index": stored,
                  "flag": flag, "c_bytes": c_bytes, "keep": (jobs, res, jw, jd, ju, resw, resd)}


def c5_measure(torch, dist, args, rank, world, dev, distributed, pkg, eng, total, steps, warmup):
    """BASELINE configs[4]: 10 GiB mixed entropy, contiguous strong-scaled shards; the JSON line (rank 0) or None."""
    lo, hi = shard_strong(total, rank, world)
    n = hi - lo
    src = gen_mixed(torch, dev, n, lo)
    step, info = c5_prepare(torch, eng, pkg, src)
    flag = info["flag"]
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    flag.zero_()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    if int(flag.item()) != 0:
        raise SystemExit("ROUND TRIP FAILURE (c5): decompressed data != source on rank %d" % rank)
    u_bytes = float(n) * BLOCK
    tot_u, tot_c, wall_max = reduce_totals(torch, dist, dev, u_bytes, info["c_bytes"], wall, distributed)
    if rank != 0:
        return None
    value = 2.0 * tot_u * steps / wall_max / 2.0 ** 30
    line = {
        "metric": "GiB/s uncompressed in (deflate) + out (inflate), 10 GiB mixed-entropy 64 KiB blocks",
        "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(wall_max / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: %d x 64 KiB blocks (zeros / 33-symbol text / text + makedata copies / random by "
                               "index mod 4), contiguous shards over %d GPUs; step = FHT compress + WRAP of what does not "
                               "shrink + decompress + compare, device resident" % (total, world),
                   "total_blocks": total, "block_bytes": BLOCK, "ratio": round(tot_u / tot_c, 4),
                   "stored_blocks_rank0": info["stored"], "roundtrip_bit_exact": True, "parallelism": "shard%d" % world},
        "roofline": roof(2 * (tot_u + tot_c) * steps / wall_max / 1e9 / world, pmc_traffic(n, "c5"),
                         "whole step (deflate + wrap + inflate kernels), wall clock, per GPU", copy_peak_gbs(torch, dev, eng)),
    }
    if world == 1 and not args.no_cpu_baseline:
        # the reference's software path for the same step: zlib level 1 (fixed code) per block, stored when it does not
        # shrink, and zlib's inflate back -- a bounded sample of the same mix
        import zlib
        sample = [row.tobytes() for row in src[:256].cpu().numpy()]
        cores = usable_cores()
        zd, nin, nout, nd = _cpu_run(sample, 1, cores, 3.0)
        streams = []
        for b in sample:
            c = zlib.compressobj(1, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
            z = c.compress(b) + c.flush()
            if len(z) >= len(b):                                  # stored: one block header per 65535 bytes
                c = zlib.compressobj(0, zlib.DEFLATED, -15)
                z = c.compress(b) + c.flush()
            streams.append(z)
        zi, _, _, ni = _cpu_run(streams, 2, cores, 3.0)
        both = 2.0 / (1.0 / zd + 1.0 / zi)                        # in + out over the time of both passes
        line["cpu_baseline"] = {"value": round(both, 4), "unit": "GiB/s uncompressed in + out", "cores": cores, "kind": "reference",
                                "what": "system zlib 1.2.11: deflate level 1 Z_FIXED per block (stored when it does not shrink), then inflate of those streams",
                                "sample": "%d + %d blocks of the same mix on %d pthreads" % (nd, ni, cores),
                                "deflate_GiB_s": round(zd, 4), "inflate_GiB_s": round(zi, 4)}
    return line


def c2_default_blocks(torch, dev):
    """configs[1] at its stated size (2^20 blocks: source + target + the inflate leg's buffer + scratch = 212 GiB) when THIS
    device has that much free, else 2^18 (a shared or smaller device must not die in torch.empty)"""
    free_b, _ = torch.cuda.mem_get_info(dev)
    return (1 << 20) if free_b > (212 << 30) else (1 << 18)


def c2_measure(torch, dist, args, rank, world, dev, distributed, pkg, eng, n, steps, warmup):
    """BASELINE configs[1]: fixed-Huffman deflate of n synthetic 64 KiB blocks per GPU (weak scaling); the JSON
    line (rank 0) or None."""
    lo, hi = shard(n, rank, world)
    src = gen_blocks(torch, dev, n, lo)
    dst = torch.empty((n, STRIDE_OUT), dtype=torch.uint8, device=dev)
    lens = np.full(n, BLOCK, np.uint32)
    jobs = eng.jobs_strided(src, BLOCK, lens, dst, STRIDE_OUT, STRIDE_OUT)
    results = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    eng.L.nxz_ctx_stage_timing.argtypes = [C.c_void_p, C.c_int]

    def step():
        eng.compress(pkg.FC_COMPRESS_FHT, jobs, n, results=results)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)

    eng.L.nxz_ctx_stage_timing(eng.ctx, 1)                 # HIP events on the launch stream around every kernel
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    eng.L.nxz_ctx_stage_timing(eng.ctx, 0)
    st, launches = stage_times(eng)
    st = [x / steps for x in st]
    launches //= steps

    res = results.cpu().numpy().view(pkg.RESULT_DTYPE)
    if not ((res["cc"] == 0) | (res["cc"] == 64)).all():
        raise SystemExit("engine reported errors: %s" % np.unique(res["cc"]))
    peak_m = copy_peak_gbs(torch, dev, eng)

    # second leg of the metric (uncompressed bytes OUT of inflate), measured after the timed deflate
    # region on the same device-resident data: the inflate engine decodes the deflate engine's output
    # and the result is compared with the source on the device (bit-exact round trip at full size).
    inflate_info = None
    if not args.no_inflate and rank == 0:
        back = torch.empty((n, BLOCK), dtype=torch.uint8, device=dev)
        jobs2 = eng.jobs_strided(dst, STRIDE_OUT, res["tpbc"].astype(np.uint32), back, BLOCK, BLOCK)
        res2 = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        eng.decompress(jobs2, n, results=res2)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(max(1, steps // 2)):
            eng.decompress(jobs2, n, results=res2)
        e1.record()
        torch.cuda.synchronize(dev)
        inf_ms = e0.elapsed_time(e1) / max(1, steps // 2)
        r2 = res2.cpu().numpy().view(pkg.RESULT_DTYPE)
        ok = dev_equal(torch, back, src) and bool((r2["cc"] == 0).all()) and bool((r2["crc"] == res["crc"]).all())
        if not ok:
            badrows = (back != src).any(dim=1)
            nbad = int(badrows.sum().item())
            import zlib
            for i in torch.nonzero(badrows).flatten()[:4].tolist():
                b = src[i].cpu().numpy().tobytes()
                o = dst[i, :int(res["tpbc"][i])].cpu().numpy().tobytes()
                g = back[i].cpu().numpy().tobytes()
                try:
                    z = zlib.decompressobj(-15)
                    zok = z.decompress(o) == b
                except zlib.error as e:
                    zok = "zlib error %s" % e
                first = next((k for k in range(BLOCK) if g[k] != b[k]), -1)
                ndiff = sum(1 for k in range(BLOCK) if g[k] != b[k])
                print("bad block %d: zlib inflates the deflate output to the source: %s; first differing byte %d, %d bytes differ; tpbc %d; crc deflate %08x inflate %08x zlib %08x"
                      % (i, zok, first, ndiff, int(res["tpbc"][i]), int(res["crc"][i]), int(r2["crc"][i]), zlib.crc32(b)), file=sys.stderr)
            raise SystemExit("ROUND TRIP FAILURE at full size (inflate of the deflate output != source): %d of %d blocks differ, cc != 0 in %d (%s), "
                             "crc differs in %d" % (nbad, n, int((r2["cc"] != 0).sum()), np.unique(r2["cc"]), int((r2["crc"] != res["crc"]).sum())))
        ub, cb = float(n) * BLOCK, float(res["tpbc"].astype(np.float64).sum())
        inflate_info = {"value": round(ub / (inf_ms * 1e-3) / 2.0 ** 30, 3), "unit": "GiB/s uncompressed out",
                        "ms_per_pass": round(inf_ms, 3), "kernel": "batched inflate (a stream per lane, fixed-code blocks) + cksum_kernel",
                        "what": "the fixed-Huffman output of the timed region", "scope": "one GPU (rank 0)", "roundtrip_bit_exact": True,
                        "roofline": roof((ub + cb) / (inf_ms * 1e-3) / 1e9, pmc_traffic(n, "inflate_own"), "nxzl::inflate_lanes_fixed_kernel + cksum_kernel", peak_m)}
        del back
    u_bytes = float(n) * BLOCK
    c_bytes = float(res["tpbc"].astype(np.float64).sum())
    tot_u, tot_c, wall_max = reduce_totals(torch, dist, dev, u_bytes, c_bytes, wall, distributed)
    if rank != 0:
        return None
    k = min(n, 48)
    src_rows = [row.tobytes() for row in src[:k].cpu().numpy()]
    out_rows = dst[:k].cpu().numpy()
    verify_sample(src_rows, out_rows, res["tpbc"], k)
    value = tot_u * steps / wall_max / 2.0 ** 30
    line = {
        "metric": "GiB/s uncompressed in (deflate), fixed-Huffman level 1, synthetic 64 KiB blocks",
        "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(wall_max / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: fixed-Huffman deflate (FC 0x00), %d x 64 KiB synthetic blocks per GPU (block i from seed "
                               "0x9E3779B97F4A7C15 ^ i: 33-symbol text, then makedata copies with per-block len_max / dist_max), device resident" % n,
                   "blocks_per_gpu": n, "block_bytes": BLOCK, "ratio": round(tot_u / tot_c, 4),
                   "parallelism": "shard%d" % world},
        "roofline": roofline(u_bytes, c_bytes, st, launches, pmc_traffic(n, "fht"),
                             "nxzl77::lz77_kernel<false, true, false> (one kernel: LZ77 + the fixed code)", peak_m, leg="fht"),
    }
    if inflate_info:
        line["inflate"] = inflate_info
    if world == 1 and not args.no_cpu_baseline:
        sample = [row.tobytes() for row in src[:min(n, 2048)].cpu().numpy()]
        line["cpu_baseline"] = cpu_baseline_deflate(sample, "fixed", 10.0)
        line["cpu_baseline"]["parity_checked_blocks"] = oracle_parity(src_rows, out_rows, res["tpbc"], False)
        line["config"]["ratio_vs_zlib1_fixed"] = round(line["cpu_baseline"]["zlib_ratio"] and line["config"]["ratio"] / line["cpu_baseline"]["zlib_ratio"], 4)
    return line


def launcher_command(n_gpus, argv, port):
    """the command line that runs this bench on n_gpus ranks of one node (what the driver uses for N > 1)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(n_gpus, argv, visible=None, run=None):
    """`python bench.py --gpus N` without a launcher: one rank per GPU as FRESH child processes (this process has not
    touched the GPU and never does: it relays rank 0's JSON line and the exit code).  Independent blocks shard over the
    ranks, RCCL carries the two all-reduces of reduce_totals (the reference's analogue is a process that
    opens whichever NX unit is nearest, lib/nx_zlib.c:568-576,1281-1287).  Returns the exit code."""
    import socket
    import subprocess
    if visible is None:
        import torch
        visible = torch.cuda.device_count()            # (counting devices does not initialise the GPU)
    if visible < n_gpus:
        print("bench.py: %d GPUs requested, %d visible" % (n_gpus, visible), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launcher_command(n_gpus, argv, port)
    p = (run or subprocess.run)(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(p.stdout)
    sys.stdout.flush()
    return p.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["corpus", "c3", "c2", "c5"], default="corpus")
    ap.add_argument("--blocks", type=int, default=0, help="c2: 64 KiB blocks per GPU (default 2^20 = 64 GiB; as the side object of the default "
                                                         "config 2^18); c5: total blocks (default 163840 = 10 GiB)")
    ap.add_argument("--corpus-jobs", type=int, default=262144, help="jobs per GPU of the corpus config (the corpus is replicated)")
    ap.add_argument("--class-jobs", type=int, default=0, help="corpus config: jobs of the per-class legs (0 = as many as the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inflate", action="store_true", help="skip the inflate legs")
    ap.add_argument("--no-api", action="store_true", help="skip the host-buffer API leg")
    ap.add_argument("--no-c2", action="store_true", help="corpus config: skip the c2 side object")
    ap.add_argument("--no-c5", action="store_true", help="corpus config: skip the c5 side object")
    ap.add_argument("--no-corpus", action="store_true", help="(kept for old command lines: no effect)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the ranks ourselves, BEFORE anything in this process touches the GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree" % (args.gpus, world))
    distributed = world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ)   # under a launcher the collectives run, even on one rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pkg = importlib.import_module("power-gzip_amd")
    eng = pkg.Engine(local_rank)

    if args.config == "c5":
        line = c5_measure(torch, dist, args, rank, world, dev, distributed, pkg, eng, args.blocks or 163840, args.steps, args.warmup)
        if line:
            print(json.dumps(line), flush=True)
    elif args.config == "c2":
        n_c2 = args.blocks or c2_default_blocks(torch, dev)
        if distributed and not args.blocks:
            # every rank the same count (weak scaling): the smallest any rank has room for
            t = torch.tensor([n_c2], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            n_c2 = int(t.item())
        line = c2_measure(torch, dist, args, rank, world, dev, distributed, pkg, eng, n_c2, args.steps, args.warmup)
        if line:
            print(json.dumps(line), flush=True)
    else:
        run_corpus(torch, dist, args, rank, world, dev, distributed, pkg, eng)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
