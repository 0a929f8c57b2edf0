#!/usr/bin/env python3
"""bench.py -- throughput of the DEFLATE hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--blocks B] [--no-inflate] [--no-cpu-baseline] [--no-dht]

A *step* is one pass of the fixed-Huffman deflate engine (LZ77 + bit-pack kernel, function
code COMPRESS_FHT) over one batch of B synthetic 64 KiB blocks that are already resident in
HBM (BASELINE.json configs[1]: "Fixed-Huffman deflate (level 1), 1xMI355X, 1 M synthetic
64 KiB blocks").  Every rank owns its own B blocks (independent units, no data-path
collective: weak scaling); the only collectives are the barrier and the reductions of
{bytes, elapsed}.  Rank 0 prints ONE JSON line.

Inside the timed region: the engine launches only (kernel + tiny result buffer).  Outside:
data generation, zlib verification of a sample, the CPU baseline (with the oracle parity check).
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
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def shard(nblocks_per_rank, rank, world):
    """Block index range of a rank (contiguous); weak scaling: every rank gets B."""
    lo = rank * nblocks_per_rank
    return lo, lo + nblocks_per_rank


def reduce_totals(torch, dist, dev, u_bytes, c_bytes, wall, distributed):
    """The only collectives of the run: SUM of {uncompressed, compressed} bytes, MAX of elapsed."""
    tot = torch.tensor([u_bytes, c_bytes], dtype=torch.float64, device=dev)
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return float(tot[0].item()), float(tot[1].item()), float(tmax.item())


def gen_blocks(torch, dev, n, first_index, chunk=4096):
    """Seeded synthetic 64 KiB blocks, generated on the device.

    Recipe (after the reference's generators): first half of every block =
    uniform text over a 33-symbol alphabet (test/test_utils.c:22-28); second half = LZ copies
    (samples/makedata.c:51-68): runs (mean length 96) copied from the first half at a per-run
    distance of 300..32768 bytes.  Chunk c of 4096 blocks uses seed 0x9E3779B97F4A7C15 ^ index.
    """
    out = torch.empty((n, BLOCK), dtype=torch.uint8, device=dev)
    alphabet = torch.tensor(list(b"abcdefghijklmnopqrstuvwxyz .,;!?\n"), dtype=torch.uint8, device=dev)
    half = BLOCK // 2
    pos = torch.arange(half, device=dev, dtype=torch.int32).unsqueeze(0)
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        g = torch.Generator(device=dev)
        g.manual_seed((0x9E3779B97F4A7C15 ^ (first_index + c0)) & 0x7FFFFFFFFFFFFFFF)
        text = alphabet[torch.randint(0, 33, (m, half), device=dev, generator=g)]
        boundary = torch.rand((m, half), device=dev, generator=g) < (1.0 / 96)
        boundary[:, 0] = True
        runstart = torch.cummax(torch.where(boundary, pos, torch.zeros_like(pos)), dim=1).values
        runid = torch.cumsum(boundary.to(torch.int32), dim=1).clamp_(max=1023).to(torch.int64)
        frac = torch.gather(torch.rand((m, 1024), device=dev, generator=g), 1, runid)
        off = (frac * (half - 300 - runstart).clamp_(min=0).to(torch.float32)).to(torch.int32)
        src_idx = (pos + off).clamp_(max=half - 1).to(torch.int64)     # distance = half - off in [300, 32768]
        out[c0:c0 + m, :half] = text
        out[c0:c0 + m, half:] = torch.gather(text, 1, src_idx)
        del text, boundary, runstart, runid, frac, off, src_idx
    return out


def pmc_traffic(n_blocks):
    """HBM bytes per launch from the committed PMC pass (profiles/*pmc_traffic.json, collected with
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same workload and corrected as
    MI355X_MICROARCH.md prescribes), scaled to this launch's block count; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        if d.get("block_bytes") != BLOCK:
            return None
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


def cpu_baseline(blocks_host, budget_s=12.0):
    """oracle ('port') timed on all host cores on a bounded sample; zlib -1 Z_FIXED beside it."""
    import oracle_lib as O
    L = O.lib()
    L.nxo_bench_deflate.restype = C.c_double
    L.nxo_bench_deflate.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    cores = usable_cores()
    buf = blocks_host.tobytes()
    nmax = len(buf) // BLOCK
    ob = C.c_uint64()
    # calibrate on a few blocks, then size the sample for ~budget_s/2 per library
    ncal = min(nmax, 8 * cores)
    t = L.nxo_bench_deflate(buf, ncal, BLOCK, cores, 0, C.byref(ob))
    n = int(max(ncal, min(nmax, ncal * (budget_s / 2) / max(t, 1e-3))))
    t_port = L.nxo_bench_deflate(buf, n, BLOCK, cores, 0, C.byref(ob))
    port_out = ob.value
    t_z = L.nxo_bench_deflate(buf, n, BLOCK, cores, 1, C.byref(ob))
    z_out = ob.value
    gib = n * BLOCK / 2.0 ** 30
    return {
        "value": round(gib / t_port, 4), "unit": "GiB/s", "cores": cores, "kind": "port",
        "sample": "%d of the same 64 KiB blocks, oracle/nxz_lz77.c fixed-Huffman deflate, %d pthreads" % (n, cores),
        "ratio": round(n * BLOCK / port_out, 4),
        "zlib1_fixed_GiB_s": round(gib / t_z, 4), "zlib1_fixed_ratio": round(n * BLOCK / z_out, 4),
    }


def verify_sample(eng, pkg, src, dst, res_host, stride_out, k=48):
    """zlib inflates the first k outputs to the inputs (outside the timed region; the bit-for-bit
    comparison with the oracle is part of the cpu_baseline leg, the only place bench.py uses it)."""
    import zlib
    k = min(k, src.shape[0])
    s = src[:k].cpu().numpy()
    d = dst[:k].cpu().numpy()
    for i in range(k):
        b = s[i].tobytes()
        got = d[i, :res_host["tpbc"][i]].tobytes()
        z = zlib.decompressobj(-15)
        if z.decompress(got) != b or not z.eof:
            raise SystemExit("ROUND TRIP FAILURE: block %d" % i)


def oracle_parity(blocks_host, dst, res_host, k=48):
    """cpu_baseline leg: the engine's first k outputs equal the oracle's, byte for byte."""
    import oracle_lib as O
    k = min(k, blocks_host.shape[0])
    d = dst[:k].cpu().numpy()
    for i in range(k):
        exp, bits = O.deflate_fixed(blocks_host[i].tobytes())
        if d[i, :res_host["tpbc"][i]].tobytes() != exp:
            raise SystemExit("PARITY FAILURE: block %d differs from the oracle" % i)
    return k


def dht_leg(torch, eng, pkg, src, dst, n, stride_out, group=64, sample=256):
    """Dynamic-Huffman leg (BASELINE configs[2] shape, same blocks): one table per `group` consecutive
    blocks, built by the host generator (nxz_dhtgen_batch in libnxz_amd.so, the part the reference
    also keeps on the host, lib/nx_dhtgen.c) from the LZ77 symbol counts of the group's first block;
    then every block is encoded with its group's table.  Timed end to end (count pass + copy of the
    counts + host tables + copy of the tables + encode pass); ratio next to zlib -1 (default
    strategy) on a sample whose outputs zlib inflates back to the input."""
    import zlib
    H = C.CDLL(os.path.join(ROOT, "power-gzip_amd", "libnxz_amd.so"))
    H.nxz_dhtgen_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    H.nxz_dhtgen_batch.restype = C.c_int
    nthreads = usable_cores()
    ng = (n + group - 1) // group
    lens = np.full(n, BLOCK, np.uint32)
    jobs_lead = eng.jobs_strided(src, BLOCK * group, np.full(ng, BLOCK, np.uint32), dst, stride_out * group, stride_out)
    jobs_all = eng.jobs_strided(src, BLOCK, lens, dst, stride_out, stride_out,
                                dht_index=(np.arange(n) // group).astype(np.uint32))
    counts = torch.empty(ng * 316, dtype=torch.int32, device=eng.dev)
    tables = np.zeros(ng, pkg.DHT_DTYPE)
    res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)

    def one_pass():
        eng.compress(pkg.FC_COMPRESS_FHT_COUNT, jobs_lead, ng, counts=counts)
        c = counts.cpu().numpy().view(np.uint32)
        if H.nxz_dhtgen_batch(c.ctypes.data, ng, tables.ctypes.data, nthreads) != 0:
            raise SystemExit("nxz_dhtgen_batch failed")
        eng.compress(pkg.FC_COMPRESS_DHT, jobs_all, n, results=res, dht=eng.to_device(tables), ntables=ng)

    one_pass()
    torch.cuda.synchronize(eng.dev)
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        one_pass()
    torch.cuda.synchronize(eng.dev)
    dt = (time.perf_counter() - t0) / reps
    r = eng.results_to_host(res)
    if not (r["cc"] == 0).all():
        raise SystemExit("dynamic-Huffman leg: engine reported errors %s" % np.unique(r["cc"]))
    m = min(n, sample)
    hs = src[:m].cpu().numpy()
    out = dst[:m].cpu().numpy()
    z1 = 0
    for i in range(m):
        b = hs[i].tobytes()
        z = zlib.decompressobj(-15)
        if z.decompress(out[i, :r["tpbc"][i]].tobytes()) != b or not z.eof:
            raise SystemExit("ROUND TRIP FAILURE (dynamic Huffman): block %d" % i)
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        z1 += len(c.compress(b) + c.flush())
    ours = int(r["tpbc"][:m].sum())
    return {"value": round(n * BLOCK / dt / 2.0 ** 30, 3), "unit": "GiB/s uncompressed in, end to end",
            "ms_per_pass": round(dt * 1e3, 3), "ratio": round(n * float(BLOCK) / float(r["tpbc"].astype(np.float64).sum()), 4),
            "zlib1_ratio_sample": round(m * BLOCK / z1, 4), "ratio_vs_zlib1": round(z1 / ours, 4),
            "tables": "one per %d blocks from the first block's counts, %d host threads" % (group, nthreads),
            "sample": "%d blocks inflated with zlib and compressed with zlib -1" % m}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=1 << 20, help="64 KiB blocks per GPU (default 2^20 = 64 GiB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inflate", action="store_true", help="skip the inflate leg (round trip of the output)")
    ap.add_argument("--no-dht", action="store_true", help="skip the dynamic-Huffman leg (N=1 only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." %
                             (args.gpus, args.gpus))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pkg = importlib.import_module("power-gzip_amd")
    eng = pkg.Engine(local_rank)

    n = args.blocks
    lo, hi = shard(n, rank, world)
    stride_out = 73856
    src = gen_blocks(torch, dev, n, lo)
    dst = torch.empty((n, stride_out), dtype=torch.uint8, device=dev)
    lens = np.full(n, BLOCK, np.uint32)
    jobs = eng.jobs_strided(src, BLOCK, lens, dst, stride_out, stride_out)
    results = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)

    def step():
        eng.compress(pkg.FC_COMPRESS_FHT, jobs, n, results=results)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps          # the engine kernel is the only work on the stream

    res = results.cpu().numpy().view(pkg.RESULT_DTYPE)
    if not ((res["cc"] == 0) | (res["cc"] == 64)).all():
        raise SystemExit("engine reported errors: %s" % np.unique(res["cc"]))

    # second leg of the metric (uncompressed bytes OUT of inflate), measured after the timed deflate
    # region on the same device-resident data: the inflate engine decodes the deflate engine's output
    # and the result is compared with the source on the device (bit-exact round trip at full size).
    inflate_info = None
    if not args.no_inflate:
        back = torch.empty((n, BLOCK), dtype=torch.uint8, device=dev)
        jobs2 = eng.jobs_strided(dst, stride_out, res["tpbc"].astype(np.uint32), back, BLOCK, BLOCK)
        res2 = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        eng.decompress(jobs2, n, results=res2)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(max(1, args.steps // 2)):
            eng.decompress(jobs2, n, results=res2)
        e1.record()
        torch.cuda.synchronize(dev)
        inf_ms = e0.elapsed_time(e1) / max(1, args.steps // 2)
        r2 = res2.cpu().numpy().view(pkg.RESULT_DTYPE)
        ok = bool(torch.equal(back, src)) and bool((r2["cc"] == 0).all()) and bool((r2["crc"] == res["crc"]).all())
        if not ok:
            raise SystemExit("ROUND TRIP FAILURE at full size (inflate of the deflate output != source)")
        inflate_info = {"value": round(float(n) * BLOCK / (inf_ms * 1e-3) / 2.0 ** 30, 3), "unit": "GiB/s uncompressed out",
                        "ms_per_pass": round(inf_ms, 3), "kernel": "nxzl::inflate_lanes_kernel + cksum_kernel",
                        "scope": "one GPU (rank 0)", "roundtrip_bit_exact": True}
        del back
    u_bytes = float(n) * BLOCK
    c_bytes = float(res["tpbc"].astype(np.float64).sum())

    tot_u, tot_c, wall_max = reduce_totals(torch, dist, dev, u_bytes, c_bytes, wall, distributed)

    if rank == 0:
        verify_sample(eng, pkg, src, dst, res, stride_out)
        value = tot_u * args.steps / wall_max / 2.0 ** 30
        achieved = (u_bytes + c_bytes) / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "GiB/s uncompressed in (deflate), fixed-Huffman level 1, synthetic 64 KiB blocks",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: fixed-Huffman deflate (FC 0x00), %d x 64 KiB synthetic blocks "
                                   "per GPU (33-symbol text + makedata-style LZ copies), device resident" % n,
                       "blocks_per_gpu": n, "block_bytes": BLOCK, "ratio": round(tot_u / tot_c, 4),
                       "parallelism": "shard%d" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(n),
                         "kernel": "nxz::deflate_kernel<false,false>", "kernel_ms": round(kern_ms, 3),
                         "algorithmic_bytes_per_launch": u_bytes + c_bytes},
        }
        if inflate_info:
            line["inflate"] = inflate_info
        if world == 1 and not args.no_cpu_baseline:
            sample = src[:min(n, 32768)].cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(sample)
            line["cpu_baseline"]["parity_checked_blocks"] = oracle_parity(sample, dst, res)
        if world == 1 and not args.no_dht:
            line["dht"] = dht_leg(torch, eng, pkg, src, dst, n, stride_out)      # overwrites dst: last leg
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
