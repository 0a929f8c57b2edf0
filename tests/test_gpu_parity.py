"""GPU parity tests (run with -m gpu on the MI355X box): the HIP engine, called through the
C ABI (libnxz_engine.so), against the CPU oracle on the same seeded inputs -- bit exact."""
import ctypes as C
import importlib
import json
import os
import zlib

import numpy as np
import pytest

import oracle_lib as O
from datagen import make_block

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("power-gzip_amd")
crb = importlib.import_module("power-gzip_amd.crb")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRIDE_IN = 65536
STRIDE_OUT = 73856  # nxz_compress_bound(65536) rounded up


@pytest.fixture(scope="module")
def eng():
    # batches of >= 32 streams go to the stream-per-lane inflate kernel in these tests (the engine's
    # own switch-over is at 4096 streams); `inflate_kernel` below runs a test under both kernels
    os.environ["NXZ_INFLATE_LANES_MIN"] = "32"
    e = pkg.Engine(0)
    yield e
    e.close()
    os.environ.pop("NXZ_INFLATE_LANES_MIN", None)


@pytest.fixture(params=["wg", "wg-128", "lanes", "lanes-fixed", "waves", "waves-global-window", "waves-by-length", "cut", "cut-3"])
def inflate_kernel(request):
    """the inflate kernels: a stream per workgroup with everything in LDS (what every batch gets; what that kernel does not
    do it hands to the stream-per-wavefront kernel), the same with pieces of 128 bits; a stream per lane; the same with the fixed-code-only kernel in front (which
    hands a batch with a dynamic block in it back to the first); a stream per wave with its window in
    LDS, with the target buffer as its window (what mid-size batches get), and that with the jobs taken
    in the order of their lengths (what batches of more than one round of wavefronts get); every stream
    cut inside its first block into up to 32 pieces (what small batches get: nxz_inflate_cut.hip), or 3"""
    old = os.environ.get("NXZ_INFLATE_LANES_MIN")
    os.environ["NXZ_INFLATE_LANES_MIN"] = "32" if request.param.startswith("lanes") else "1000000000"
    os.environ["NXZ_LANES_FIXED"] = "2" if request.param == "lanes-fixed" else "0"
    os.environ["NXZ_INFLATE_CUT"] = "1" if request.param.startswith("cut") else "0"
    os.environ["NXZ_INFLATE_WG"] = "1" if request.param.startswith("wg") else "0"
    if request.param == "wg-128":
        os.environ["NXZ_WG_PMIN"] = "128"
    if request.param == "cut-3":
        os.environ["NXZ_INFLATE_CUT_PIECES"] = "3"
    if request.param in ("waves-global-window", "waves-by-length"):
        os.environ["NXZ_INFLATE_LDS_MAX"] = "0"
    if request.param == "waves-by-length":                  # (the jobs in the order of their lengths, as batches beyond one round of wavefronts go)
        os.environ["NXZ_INFLATE_ORDER"] = "1"
    yield request.param
    os.environ.pop("NXZ_INFLATE_ORDER", None)
    os.environ.pop("NXZ_INFLATE_WG", None)
    os.environ.pop("NXZ_WG_PMIN", None)
    os.environ.pop("NXZ_INFLATE_CUT", None)
    os.environ.pop("NXZ_INFLATE_CUT_PIECES", None)
    os.environ["NXZ_INFLATE_LANES_MIN"] = old if old is not None else "32"
    os.environ.pop("NXZ_INFLATE_LDS_MAX", None)
    os.environ.pop("NXZ_LANES_FIXED", None)


def pack_blocks(eng, blocks, stride):
    import torch
    host = np.zeros((len(blocks), stride), np.uint8)
    for i, b in enumerate(blocks):
        host[i, :len(b)] = np.frombuffer(b, np.uint8)
    return torch.from_numpy(host).to(eng.dev)


BLOCK_CASES = [("zeros", 65536), ("text33", 65536), ("lz", 65536), ("random", 65536), ("alice", 65536),
               ("lz", 0), ("lz", 1), ("lz", 2), ("lz", 3), ("lz", 4), ("lz", 5), ("text33", 15), ("text33", 16),
               ("text33", 17), ("text33", 63), ("text33", 64), ("text33", 65), ("lz", 2047), ("lz", 2048),
               ("lz", 2049), ("lz", 16383), ("lz", 16384), ("lz", 16385), ("alice", 40000), ("zeros", 300),
               ("zeros", 16384), ("random", 5000), ("lz", 65535), ("alice", 32768), ("lz", 49152)]
# long matches, broken chains, piece (512) and tile (16384) boundaries
BLOCK_CASES += [("periodic", 65536)] * 12 + [("binary", 65536)] * 6 + [("sparse", 65536)] * 12 + \
               [("periodic", 16384 + 511), ("periodic", 513), ("binary", 16385), ("sparse", 33000), ("sparse", 511), ("sparse", 1025)]


def test_fixed_huffman_bit_exact(eng):
    import torch
    blocks = [make_block(k, n, seed=i) for i, (k, n) in enumerate(BLOCK_CASES)]
    blocks += [make_block("lz", 65536, seed=100 + i) for i in range(40)]
    src = pack_blocks(eng, blocks, STRIDE_IN)
    dst = torch.zeros((len(blocks), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    lens = np.array([len(b) for b in blocks], np.uint32)
    jobs = eng.jobs_strided(src, STRIDE_IN, lens, dst, STRIDE_OUT, STRIDE_OUT)
    res, _ = eng.compress(pkg.FC_COMPRESS_FHT, jobs, len(blocks))
    r = eng.results_to_host(res)
    out = dst.cpu().numpy()
    for i, b in enumerate(blocks):
        exp, bits = O.deflate_fixed(b)
        assert r["tpbc"][i] == len(exp), (i, BLOCK_CASES[i] if i < len(BLOCK_CASES) else "lz64k")
        assert r["tebc"][i] == bits % 8
        assert out[i, :len(exp)].tobytes() == exp, i
        assert r["spbc"][i] == len(b)
        assert r["crc"][i] == zlib.crc32(b) and r["adler"][i] == zlib.adler32(b), i
        assert r["cc"][i] == (64 if len(exp) > len(b) else 0), i
        d = zlib.decompressobj(-15)
        assert d.decompress(exp) == b and d.eof


def test_history_and_running_checksums(eng):
    import torch
    hist = make_block("text33", 32768, 5)
    blocks, hlens = [], []
    for hl, n, seed in [(32768, 20000, 1), (16, 100, 2), (4096, 32768, 3), (32768, 32768, 4), (1024, 0, 5)]:
        body = hist[:n] if seed % 2 else make_block("lz", n, seed)
        blocks.append(hist[-hl:] + body)
        hlens.append(hl)
    src = pack_blocks(eng, blocks, STRIDE_IN)
    dst = torch.zeros((len(blocks), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, STRIDE_IN, np.array([len(b) for b in blocks], np.uint32), dst, STRIDE_OUT,
                            STRIDE_OUT, hist_len=np.array(hlens, np.uint32), in_crc=0xdeadbeef, in_adler=0x00c0ffee)
    res, _ = eng.compress(pkg.FC_COMPRESS_RESUME_FHT, jobs, len(blocks))
    r = eng.results_to_host(res)
    out = dst.cpu().numpy()
    for i, (b, hl) in enumerate(zip(blocks, hlens)):
        exp, bits = O.deflate_fixed(b, hist=hl)
        assert out[i, :len(exp)].tobytes() == exp and r["tpbc"][i] == len(exp), i
        assert r["spbc"][i] == len(b)
        assert r["crc"][i] == zlib.crc32(b[hl:], 0xdeadbeef), i
        assert r["adler"][i] == zlib.adler32(b[hl:], 0x00c0ffee), i


def _dht_array(tables):
    arr = np.zeros(len(tables), pkg.DHT_DTYPE)
    for i, (bits, n) in enumerate(tables):
        arr["dhtlen"][i] = n
        arr["dht"][i, :len(bits)] = np.frombuffer(bits, np.uint8)
    return arr


def test_dynamic_huffman_and_counts_bit_exact(eng):
    import torch
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "builtin_dht.json")))
    kinds = [("alice", 65536), ("lz", 65536), ("text33", 30000), ("zeros", 65536), ("random", 4096), ("lz", 7)]
    blocks = [make_block(k, n, seed=50 + i) for i, (k, n) in enumerate(kinds)]
    tables = [(bytes.fromhex(g[0]["dht"]), g[0]["dhtlen"]), (bytes.fromhex(g[7]["dht"]), g[7]["dhtlen"])]
    # plus an exact table per block (may miss codes for other blocks)
    for b in blocks[:2]:
        tok, nt = O.lz77(b)
        ll, d = O.counts(tok, nt)
        tables.append(O.dhtgen(ll, d))
    dht = eng.to_device(_dht_array(tables))
    use = [0, 1, 2, 0, 1, 3]          # job 5 uses block 1's exact table on other data -> may be CC 66
    src = pack_blocks(eng, blocks, STRIDE_IN)
    dst = torch.zeros((len(blocks), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, STRIDE_IN, np.array([len(b) for b in blocks], np.uint32), dst, STRIDE_OUT,
                            STRIDE_OUT, dht_index=np.array(use, np.uint32))
    res, cnt = eng.compress(pkg.FC_COMPRESS_DHT_COUNT, jobs, len(blocks), dht=dht, ntables=len(tables))
    r = eng.results_to_host(res)
    out = dst.cpu().numpy()
    cnt = cnt.cpu().numpy().view(np.uint32).reshape(len(blocks), 316)
    for i, b in enumerate(blocks):
        bits, n = tables[use[i]]
        exp, nbits = O.deflate_dynamic(b, bits, n)
        if exp is None:
            assert r["cc"][i] == 66, i
            continue
        assert r["tpbc"][i] == len(exp) and r["tebc"][i] == nbits % 8, i
        assert out[i, :len(exp)].tobytes() == exp, i
        tok, nt = O.lz77(b)
        ll, d = O.counts(tok, nt)
        assert list(cnt[i]) == list(ll) + list(d), i
        dz = zlib.decompressobj(-15)
        assert dz.decompress(exp) == b and dz.eof


def test_invalid_dht_is_cc68(eng):
    import torch
    b = make_block("alice", 1000, 1)
    dht = eng.to_device(_dht_array([(b"\xff" * 40, 300)]))
    src = pack_blocks(eng, [b], STRIDE_IN)
    dst = torch.zeros((1, STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, STRIDE_IN, np.array([len(b)], np.uint32), dst, STRIDE_OUT, STRIDE_OUT)
    res, _ = eng.compress(pkg.FC_COMPRESS_DHT, jobs, 1, dht=dht, ntables=1)
    assert eng.results_to_host(res)["cc"][0] == 68


def test_target_too_small_is_cc13(eng):
    import torch
    b = make_block("random", 65536, 1)
    src = pack_blocks(eng, [b], STRIDE_IN)
    dst = torch.zeros((1, STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, STRIDE_IN, np.array([len(b)], np.uint32), dst, STRIDE_OUT, 32768)
    res, _ = eng.compress(pkg.FC_COMPRESS_FHT, jobs, 1)
    r = eng.results_to_host(res)
    assert r["cc"][0] == 13 and r["tpbc"][0] == 0
    assert not dst[0, 32768:].any()          # nothing written past the capacity


def test_wrap(eng):
    import torch
    blocks = [make_block("random", n, n) for n in (0, 1, 3, 4, 5, 255, 256, 4097, 60000)]
    src = pack_blocks(eng, blocks, STRIDE_IN)
    dst = torch.zeros((len(blocks), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, STRIDE_IN, np.array([len(b) for b in blocks], np.uint32), dst, STRIDE_OUT, STRIDE_OUT)
    r = eng.results_to_host(eng.wrap(jobs, len(blocks)))
    out = dst.cpu().numpy()
    for i, b in enumerate(blocks):
        assert r["cc"][i] == 0 and r["tpbc"][i] == len(b)
        assert out[i, :len(b)].tobytes() == b
        assert r["crc"][i] == zlib.crc32(b) and r["adler"][i] == zlib.adler32(b), i


def _zstreams():
    data = {k: make_block(k, n, seed=7) for k, n in [("alice", 65536), ("lz", 65536), ("random", 20000),
                                                      ("zeros", 65536), ("text33", 3000)]}
    out = []
    for name, d in data.items():
        for level, strat in [(1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                             (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_HUFFMAN_ONLY)]:
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
            out.append((d, co.compress(d) + co.flush()))
    big = make_block("lz", 300000, 11) + make_block("alice", 200000, 12)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    out.append((big, co.compress(big[:250000]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(big[250000:]) + co.flush()))
    return out


@pytest.mark.parametrize("copies", [1, 3])          # 31 streams: wave-per-stream kernel; 93: lane-per-stream kernel
def test_inflate_zlib_streams_bit_exact(eng, copies, inflate_kernel):
    import torch
    streams = _zstreams() * copies
    cstride = max(len(c) for _, c in streams) + 64
    cstride = (cstride + 15) & ~15
    ostride = (max(len(d) for d, _ in streams) + 15) & ~15
    src = pack_blocks(eng, [c + b"TRAILER8" for _, c in streams], cstride)
    dst = torch.zeros((len(streams), ostride), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, cstride, np.array([len(c) + 8 for _, c in streams], np.uint32), dst, ostride, ostride)
    r = eng.results_to_host(eng.decompress(jobs, len(streams)))
    out = dst.cpu().numpy()
    for i, (d, c) in enumerate(streams):
        assert r["cc"][i] == 3 and (r["sfbt"][i] & 0xf) == 0 and r["sfbt"][i] & 0x100, i   # trailer follows the final EOB
        assert r["tpbc"][i] == len(d), i
        assert out[i, :len(d)].tobytes() == d, i
        assert r["crc"][i] == zlib.crc32(d) and r["adler"][i] == zlib.adler32(d), i
        assert r["subc"][i] // 8 == 8, i


def test_inflate_suspend_state_matches_oracle(eng, inflate_kernel):
    """cut streams at arbitrary bytes: CC 3, SFBT/SUBC/rembytecnt/tpbc/dht must equal the CPU model."""
    import random
    import torch
    rnd = random.Random(3)
    streams = _zstreams()
    cases = []
    for d, c in streams[:18]:
        for _ in range(6):
            cases.append((d, c[:rnd.randrange(0, len(c))]))
    cstride = (max(len(c) for _, c in cases) + 31) & ~15
    ostride = (max(len(d) for d, _ in cases) + 15) & ~15
    src = pack_blocks(eng, [c for _, c in cases], cstride)
    dst = torch.zeros((len(cases), ostride), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, cstride, np.array([len(c) for _, c in cases], np.uint32), dst, ostride, ostride)
    dht_io = torch.zeros(len(cases) * pkg.DHT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
    r = eng.results_to_host(eng.decompress(jobs, len(cases), dht_io=dht_io))
    out = dst.cpu().numpy()
    dio = dht_io.cpu().numpy().view(pkg.DHT_DTYPE)
    for i, (d, c) in enumerate(cases):
        exp, st = O.inflate(c, ostride)
        assert st.err == 0
        assert r["tpbc"][i] == st.tpbc and out[i, :st.tpbc].tobytes() == exp, i
        assert (r["sfbt"][i] & 0xf) == st.out_sfbt and r["subc"][i] == st.out_subc, i
        if (st.out_sfbt & 0xe) == 0x8:
            assert r["tebc"][i] == st.out_rembytecnt
        if (st.out_sfbt & 0xe) == 0xc:
            assert dio["dhtlen"][i] == st.out_dhtlen
            nb = (st.out_dhtlen + 7) // 8
            assert dio["dht"][i, :nb].tobytes() == bytes(st.out_dht)[:nb], i


def test_damaged_streams_match_the_oracle(eng, inflate_kernel):
    """Bit flips, byte swaps, cuts and garbage: both inflate kernels take the oracle's decision
    (error code, or where the stream suspends and what was produced before), never write past the
    target and never hang."""
    import random
    import torch
    rnd = random.Random(11)
    base = _zstreams()[:24]
    cases = []
    for k in range(420):
        d, c = base[k % len(base)]
        b = bytearray(c)
        how = k % 5
        if how == 0 and b:                                # one to three flipped bits
            for _ in range(rnd.randrange(1, 4)):
                i = rnd.randrange(len(b)); b[i] ^= 1 << rnd.randrange(8)
        elif how == 1 and len(b) > 8:                     # a damaged header region
            for i in range(rnd.randrange(1, 6)):
                b[rnd.randrange(0, min(len(b), 48))] = rnd.randrange(256)
        elif how == 2 and len(b) > 4:                     # cut and flipped
            del b[rnd.randrange(1, len(b)):]
            b[rnd.randrange(len(b))] ^= 0x40
        elif how == 3:                                    # pure noise
            b = bytearray(rnd.randbytes(rnd.randrange(1, 3000)))
        elif b:                                           # a zeroed or saturated span
            i = rnd.randrange(len(b)); n = rnd.randrange(1, 64)
            b[i:i + n] = bytes([rnd.choice([0, 0xff])]) * len(b[i:i + n])
        cases.append(bytes(b))
    cap = 65536 + 4096                                    # damaged streams may also run long: CC 13 is a valid verdict
    cstride = (max(map(len, cases)) + 31) & ~15
    ostride = cap + 64
    src = pack_blocks(eng, cases, cstride)
    dst = torch.full((len(cases), ostride), 0xAA, dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, cstride, np.array([len(c) for c in cases], np.uint32), dst, ostride, cap)
    r = eng.results_to_host(eng.decompress(jobs, len(cases)))
    out = dst.cpu().numpy()
    verdicts = {}
    for i, c in enumerate(cases):
        exp, st = O.inflate(c, cap)
        assert (out[i, cap:] == 0xAA).all(), i           # nothing beyond the target
        if st.err:
            assert r["cc"][i] == st.err, (i, r["cc"][i], st.err)
        else:
            assert r["cc"][i] in (0, 3), (i, r["cc"][i])
            assert r["tpbc"][i] == st.tpbc and out[i, :st.tpbc].tobytes() == exp, i
            subc = st.out_subc
            if st.final_eob and subc > 0xfff8:            # SUBC is a 16-bit field: whole excess bytes stay unread
                subc -= 8 * ((subc - 0xfff8 + 7) // 8)
            assert (r["sfbt"][i] & 0xf) == st.out_sfbt and r["subc"][i] == subc, i
            assert bool(r["sfbt"][i] & 0x100) == bool(st.final_eob), i
        verdicts[int(r["cc"][i])] = verdicts.get(int(r["cc"][i]), 0) + 1
    assert len(verdicts) >= 4, verdicts                   # the set really exercises the error paths


def test_lanes_with_equal_tables_share_one_and_others_do_not(eng, inflate_kernel):
    """Streams with identical dynamic tables sit next to streams with other tables and with several
    dynamic blocks each (the wave's spare table slot is taken, busy, free again): every output is its
    own stream's data."""
    import torch
    datas = [make_block(k, 65536, seed=400 + i) for i, k in enumerate(("alice", "lz", "text33", "alice"))]
    whole = []
    for d in datas:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        whole.append(co.compress(d) + co.flush())
    pieces = []
    for d in datas[:2]:                                   # three dynamic blocks per stream
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        pieces.append(co.compress(d[:20000]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(d[20000:45000]) +
                      co.flush(zlib.Z_SYNC_FLUSH) + co.compress(d[45000:]) + co.flush())
    streams = []
    for i in range(64 * 5 + 17):                          # the last wave is not full
        k = (i * 7 + i // 64) % 6
        streams.append((datas[k], whole[k]) if k < 4 else (datas[k - 4], pieces[k - 4]))
    streams[64:128] = [(datas[0], whole[0])] * 64         # a wave whose lanes all hold the same table
    cstride = (max(len(c) for _, c in streams) + 31) & ~15
    src = pack_blocks(eng, [c for _, c in streams], cstride)
    dst = torch.zeros((len(streams), 65536), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, cstride, np.array([len(c) for _, c in streams], np.uint32), dst, 65536, 65536)
    for rep in range(2):
        r = eng.results_to_host(eng.decompress(jobs, len(streams)))
        out = dst.cpu().numpy()
        for i, (d, c) in enumerate(streams):
            assert r["cc"][i] == 0 and r["tpbc"][i] == 65536 and r["crc"][i] == zlib.crc32(d), i
            assert out[i].tobytes() == d, i
        dst.zero_()


def test_inflate_resume_chain(eng, inflate_kernel):
    """feed a stream in pieces through resume jobs with history, like lib/nx_inflate.c:1464-1609 does."""
    import torch
    d, c = _zstreams()[-1]
    c = c + b"12345678"
    piece = 40000
    ostride = 1 << 20
    out = b""
    pos = 0
    resume = 0
    dht_state = None
    crc, adler = 0, 1
    for _ in range(100):
        chunk = c[pos:pos + piece]
        hist = out[-32768:]
        hpad = (-len(hist)) % 16
        srcbuf = bytes(hpad) + hist + chunk      # history length must be a multiple of 16
        src = pack_blocks(eng, [srcbuf], (len(srcbuf) + 31) & ~15)
        dst = torch.zeros((1, ostride), dtype=torch.uint8, device=eng.dev)
        jobs = eng.jobs_strided(src, 0, np.array([len(srcbuf)], np.uint32), dst, 0, ostride,
                                hist_len=len(hist) + hpad, in_crc=crc, in_adler=adler, resume=resume)
        dht_io = torch.zeros(pkg.DHT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
        if dht_state is not None:
            dht_io = eng.to_device(dht_state)
        r = eng.results_to_host(eng.decompress(jobs, 1, dht_io=dht_io))[0]
        assert r["cc"] in (0, 3), r
        out += dst[0, :r["tpbc"]].cpu().numpy().tobytes()
        crc, adler = int(r["crc"]), int(r["adler"])
        sfbt = int(r["sfbt"]) & 0xf
        if r["sfbt"] & 0x100:
            consumed = pos + len(chunk) - int(r["subc"]) // 8
            assert c[consumed:] == b"12345678"
            break
        back = (int(r["subc"]) + 7) // 8
        pos += len(chunk) - back
        resume = (sfbt << 16) | ((int(r["subc"]) % 8) << 20) | (int(r["tebc"]) if (sfbt & 0xe) == 8 else 0)
        dht_state = dht_io.cpu().numpy().view(pkg.DHT_DTYPE).copy() if (sfbt & 0xe) == 0xc else None
    assert out == d
    assert crc == zlib.crc32(d) and adler == zlib.adler32(d)


# ---------------------------------------------------------------------------
# the six transport symbols: nxu_run_job on host buffers vs the CPU engine model
# ---------------------------------------------------------------------------
def _run_both(eng, handle, setup_kwargs, src_bufs_bytes, dst_sizes):
    """returns (gpu_job, gpu_dst_bytes, cpu_job, cpu_dst_bytes)"""
    res = []
    for which in ("gpu", "cpu"):
        j = crb.Job()
        srcs = [C.create_string_buffer(b, len(b)) for b in src_bufs_bytes]
        dsts = [C.create_string_buffer(n) for n in dst_sizes]
        j.setup(src_bufs=srcs, dst_bufs=dsts, **setup_kwargs)
        if which == "gpu":
            rc = eng.L.nxu_run_job(C.c_void_p(j.addr), C.byref(handle))
        else:
            rc = O.lib().nxo_run_job(C.c_void_p(j.addr))
        assert rc == 0 and j.valid == 1
        res.append((j, b"".join(d.raw for d in dsts)))
    return res[0][0], res[0][1], res[1][0], res[1][1]


@pytest.fixture(scope="module")
def handle(eng):
    h = crb.DevHandle()
    assert eng.L.nx_function_begin(2, -1, C.byref(h)) == 0
    assert h.paste_addr and h.fd > 0 and h.function == 2
    yield h
    eng.L.nx_function_end(C.byref(h))


def test_nxu_run_job_compress_matches_model(eng, handle):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "builtin_dht.json")))
    dht0, dhtlen0 = bytes.fromhex(g[0]["dht"]), g[0]["dhtlen"]
    hist = make_block("alice", 4096, 1)
    body = make_block("alice", 30000, 2)
    cases = [
        dict(fc=0x08, src=[hist, body], hist_qw=256, dst=[70000]),
        dict(fc=0x08, src=[body[:100], body[100:5000], body[5000:]], hist_qw=0, dst=[1000, 80000]),  # gather + scatter
        dict(fc=0x0e, src=[hist, body], hist_qw=256, dst=[70000], dht=(dht0, dhtlen0)),
        dict(fc=0x0c, src=[body], hist_qw=0, dst=[70000]),
        dict(fc=0x08, src=[make_block("random", 3000, 3)], hist_qw=0, dst=[8000]),         # expands -> CC 64
        dict(fc=0x08, src=[body], hist_qw=0, dst=[512]),                                    # CC 13
        dict(fc=0x08, src=[make_block("lz", 100000, 4)], hist_qw=0, dst=[200000]),          # > 64 KiB -> CC 3 partial
        dict(fc=0x1e, src=[body[:7000]], hist_qw=0, dst=[60000]),                           # wrap
    ]
    for k, cse in enumerate(cases):
        kw = dict(fc=cse["fc"], histlen_qw=cse["hist_qw"], in_crc=0x1234 if cse["fc"] != 0x1e else 0,
                  in_adler=77 if cse["fc"] != 0x1e else 1)
        if "dht" in cse:
            kw.update(dht=cse["dht"][0], dhtlen=cse["dht"][1])
        gj, gd, cj, cd = _run_both(eng, handle, kw, cse["src"], cse["dst"])
        assert gj.cc == cj.cc and gj.ce3 == cj.ce3, (k, gj.cc, cj.cc)
        if cj.cc in (0, 3, 64):
            assert gj.tpbc == cj.tpbc and gd[:cj.tpbc] == cd[:cj.tpbc], k
            assert gj.out_crc == cj.out_crc and gj.out_adler == cj.out_adler, k
            if cse["fc"] != 0x1e:
                assert gj.out_tebc == cj.out_tebc, k
            if cse["fc"] & 0x4 and cse["fc"] != 0x1e:
                assert gj.lzcounts == cj.lzcounts and gj.out_spbc_count == cj.out_spbc_count, k
            else:
                assert gj.out_spbc == cj.out_spbc, k


def test_nxu_run_job_decompress_matches_model(eng, handle):
    d, c = _zstreams()[1]
    trailer = b"\x01\x02\x03\x04\x05\x06\x07\x08"
    cases = [dict(fc=0x10, src=[c + trailer], dst=[len(d) + 100]),
             dict(fc=0x10, src=[c[:1000], c[1000:] + trailer], dst=[1000, len(d)]),
             dict(fc=0x10, src=[c], dst=[len(d)]),                      # ends exactly at EOB -> CC 0
             dict(fc=0x10, src=[c[:len(c) // 2]], dst=[len(d)]),        # suspended
             dict(fc=0x10, src=[c], dst=[len(d) // 2])]                 # CC 13
    for k, cse in enumerate(cases):
        gj, gd, cj, cd = _run_both(eng, handle, dict(fc=cse["fc"]), cse["src"], cse["dst"])
        assert gj.cc == cj.cc and gj.ce3 == cj.ce3, (k, gj.cc, cj.cc)
        if cj.cc in (0, 3):
            assert gj.tpbc == cj.tpbc and gd[:cj.tpbc] == cd[:cj.tpbc], k
            assert (gj.out_sfbt, gj.out_subc, gj.out_spbc_decomp) == (cj.out_sfbt, cj.out_subc, cj.out_spbc_decomp), k
            assert gj.out_crc == cj.out_crc and gj.out_adler == cj.out_adler, k
            if (cj.out_sfbt & 0xe) == 0xc:
                assert gj.out_dhtlen == cj.out_dhtlen
                assert gj.out_dht[:(cj.out_dhtlen + 7) // 8] == cj.out_dht[:(cj.out_dhtlen + 7) // 8]


def test_full_size_roundtrip_property(eng):
    """BASELINE-sized blocks at scale: compress on the GPU, inflate on the GPU, compare on the GPU
    (size-independent property; the oracle is not in the loop)."""
    import torch
    n = 2048
    g = torch.Generator(device="cpu").manual_seed(1)
    base = torch.from_numpy(np.frombuffer(make_block("lz", 1 << 20, 9) + make_block("alice", 1 << 20, 10), np.uint8).copy())
    idx = torch.randint(0, base.numel() - 65536, (n,), generator=g)
    src = torch.stack([base[i:i + 65536] for i in idx.tolist()]).to(eng.dev)
    src[::7] ^= torch.randint(0, 3, (src[::7].shape[0], 65536), dtype=torch.uint8, generator=g).to(eng.dev) // 2
    comp = torch.zeros((n, STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    lens = np.full(n, 65536, np.uint32)
    jobs = eng.jobs_strided(src, 65536, lens, comp, STRIDE_OUT, STRIDE_OUT)
    r = eng.results_to_host(eng.compress(pkg.FC_COMPRESS_FHT, jobs, n)[0])
    assert (r["cc"] == 0).all()
    back = torch.zeros((n, 65536), dtype=torch.uint8, device=eng.dev)
    jobs2 = eng.jobs_strided(comp, STRIDE_OUT, r["tpbc"].astype(np.uint32), back, 65536, 65536)
    r2 = eng.results_to_host(eng.decompress(jobs2, n))
    assert (r2["cc"] == 0).all() and (r2["tpbc"] == 65536).all()
    assert torch.equal(back, src)
    assert (r2["crc"] == r["crc"]).all() and (r2["adler"] == r["adler"]).all()
    ratio = 65536.0 * n / r["tpbc"].sum()
    assert ratio > 1.5


def test_a_large_batch_of_streams_with_tables_goes_to_both_kernels_and_is_exact(eng):
    """163 840 streams or more that bring tables are shared out: a stream per lane on the caller's HIP stream, a stream
    per wavefront on a second one, side by side (nxz_engine.cpp, NXZ_INFLATE_SPLIT_PCT).  200 000 zlib -6 streams of
    500 different short texts, lengths 200 .. 3000: every result (bytes, checksums, where the stream stood) is what
    zlib says, whichever kernel took the stream -- and the same with the share at 0."""
    import torch
    rng = np.random.default_rng(3)
    text = make_block("alice", 200000, 5)
    uniq, comps = [], []
    for k in range(500):
        n = int(rng.integers(200, 3000)); o = int(rng.integers(0, len(text) - n))
        d = text[o:o + n]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        c = co.compress(d) + co.flush()
        assert (c[0] >> 1) & 3 == 2                          # a block that brings its table
        uniq.append(d); comps.append(c)
    n = 200000
    SI, SO = 2048, 3072
    which = rng.integers(0, 500, n)
    src = np.zeros((500, SI), np.uint8)
    for k, c in enumerate(comps):
        src[k, :len(c)] = np.frombuffer(c, np.uint8)
    d_src = torch.from_numpy(src).to(eng.dev)[torch.from_numpy(which).to(eng.dev)].contiguous()
    lens = np.array([len(comps[k]) for k in which], np.uint32)
    want_len = np.array([len(uniq[k]) for k in which], np.uint32)
    want_crc = np.array([zlib.crc32(uniq[k]) for k in range(500)], np.uint32)[which]
    expect = np.zeros((500, SO), np.uint8)
    for k, d in enumerate(uniq):
        expect[k, :len(d)] = np.frombuffer(d, np.uint8)
    d_expect = torch.from_numpy(expect).to(eng.dev)[torch.from_numpy(which).to(eng.dev)]
    for _ in range(2):
        out = torch.zeros((n, SO), dtype=torch.uint8, device=eng.dev)
        jobs = eng.jobs_strided(d_src, SI, lens, out, SO, SO)
        r = eng.results_to_host(eng.decompress(jobs, n))
        assert (r["cc"] == 0).all()
        assert (r["tpbc"] == want_len).all() and (r["crc"] == want_crc).all()
        assert torch.equal(out, d_expect)


def test_batches_on_two_streams_at_once(eng):
    """Two batched compress launches in flight on different streams (each draws its jobs from its own
    counter) give what they give one after the other; more jobs than workgroups, fewer, and one."""
    import torch
    res_seq, res_par = [], []
    for n in (1, 7, 300, 2000):
        blocks = [make_block(("alice", "lz", "zeros", "random", "periodic")[i % 5], 65536 if i % 3 else 30000 + i, seed=i) for i in range(n)]
        src = pack_blocks(eng, blocks, STRIDE_IN)
        lens = np.array([len(b) for b in blocks], np.uint32)
        outs = []
        for mode in ("seq", "par"):
            d1 = torch.zeros((n, STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
            d2 = torch.zeros((n, STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
            j1 = eng.jobs_strided(src, STRIDE_IN, lens, d1, STRIDE_OUT, STRIDE_OUT)
            j2 = eng.jobs_strided(src, STRIDE_IN, lens, d2, STRIDE_OUT, STRIDE_OUT)
            r1 = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
            r2 = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
            torch.cuda.synchronize()
            if mode == "seq":
                eng.compress(pkg.FC_COMPRESS_FHT, j1, n, results=r1)
                torch.cuda.synchronize()
                eng.compress(pkg.FC_COMPRESS_FHT, j2, n, results=r2)
            else:
                s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
                with torch.cuda.stream(s1):
                    eng.compress(pkg.FC_COMPRESS_FHT, j1, n, results=r1)
                with torch.cuda.stream(s2):
                    eng.compress(pkg.FC_COMPRESS_FHT, j2, n, results=r2)
            torch.cuda.synchronize()
            outs.append((d1.cpu().numpy(), d2.cpu().numpy(), eng.results_to_host(r1).copy(), eng.results_to_host(r2).copy()))
        (a1, a2, ra1, ra2), (b1, b2, rb1, rb2) = outs
        for x, y in ((ra1, rb1), (ra2, rb2), (ra1, ra2)):
            assert (x["cc"] == y["cc"]).all() and (x["tpbc"] == y["tpbc"]).all() and (x["crc"] == y["crc"]).all()
        assert (a1 == b1).all() and (a2 == b2).all() and (a1 == a2).all()
        for i in (0, n // 2, n - 1):
            if ra1["cc"][i] == 0:
                exp, _ = O.deflate_fixed(blocks[i])
                assert a1[i, :len(exp)].tobytes() == exp


def test_dynamic_and_inflate_batches_on_two_streams_at_once(eng):
    """Launches on different streams may overlap on the device: each stream has its own prepared
    tables and its own decode workspace, so two dynamic-Huffman batches with different tables, and
    two inflate batches of different dynamic streams, give what they give alone."""
    import torch
    n = 1500
    kinds = ("alice", "lz", "text33", "periodic")
    blocks_a = [make_block(kinds[i % 4], 65536, seed=900 + i % 7) for i in range(n)]
    blocks_b = [make_block(kinds[(i + 1) % 4], 65536, seed=950 + i % 5) for i in range(n)]

    def table_for(b):
        tok, nt = O.lz77(b)
        ll, d = O.counts(tok, nt)
        for i in range(286):
            ll[i] = max(ll[i], 1)                   # a universal table: every symbol has a code
        for i in range(30):
            d[i] = max(d[i], 1)
        return O.dhtgen(ll, d)
    ta, tb = table_for(blocks_a[0]), table_for(blocks_b[1])
    lens = np.full(n, 65536, np.uint32)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    sets = []
    for blocks, tab in ((blocks_a, ta), (blocks_b, tb)):
        src = pack_blocks(eng, blocks, STRIDE_IN)
        dst = torch.zeros((n, STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
        jobs = eng.jobs_strided(src, STRIDE_IN, lens, dst, STRIDE_OUT, STRIDE_OUT, dht_index=np.zeros(n, np.uint32))
        res = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
        sets.append((blocks, tab, src, dst, jobs, res, eng.to_device(_dht_array([tab]))))
    torch.cuda.synchronize()
    for rep in range(3):
        for st, (blocks, tab, src, dst, jobs, res, dht) in zip((s1, s2), sets):
            with torch.cuda.stream(st):
                eng.compress(pkg.FC_COMPRESS_DHT, jobs, n, results=res, dht=dht, ntables=1)
    torch.cuda.synchronize()
    comp = []
    for blocks, tab, src, dst, jobs, res, dht in sets:
        r = eng.results_to_host(res).copy()
        out = dst.cpu().numpy()
        # CC 64: complete output that is larger than the source (33-symbol noise under a table made from text)
        assert ((r["cc"] == 0) | (r["cc"] == 64)).all(), np.unique(r["cc"], return_counts=True)
        for i in (0, 1, 2, 3, n // 2, n - 1):
            exp, nbits = O.deflate_dynamic(blocks[i], tab[0], tab[1])
            assert out[i, :len(exp)].tobytes() == exp, i
        for i in range(0, n, 97):
            dz = zlib.decompressobj(-15)
            assert dz.decompress(out[i, :r["tpbc"][i]].tobytes()) == blocks[i] and dz.eof, i
        comp.append((dst, r))
    # inflate both outputs at the same time on the two streams
    back = []
    for (dst, r), st in zip(comp, (s1, s2)):
        o = torch.zeros((n, 65536), dtype=torch.uint8, device=eng.dev)
        j = eng.jobs_strided(dst, STRIDE_OUT, r["tpbc"].astype(np.uint32), o, 65536, 65536)
        rr = torch.empty(n * pkg.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=eng.dev)
        back.append((o, j, rr))
    torch.cuda.synchronize()
    for rep in range(3):
        for st, (o, j, rr) in zip((s1, s2), back):
            with torch.cuda.stream(st):
                eng.decompress(j, n, results=rr)
    torch.cuda.synchronize()
    for (o, j, rr), (blocks, *_), (dst, r) in zip(back, sets, comp):
        r2 = eng.results_to_host(rr)
        assert (r2["cc"] == 0).all() and (r2["crc"] == r["crc"]).all()
        got = o.cpu().numpy()
        for i in range(0, n, 41):
            assert got[i].tobytes() == blocks[i], i


def test_seeded_fuzz_against_oracle(eng):
    """1500 blocks of random kind, random size (0 .. 64 KiB, many at tile / piece / chunk edges) and
    random history (multiple of 16, window + block <= 64 KiB), checked bit for bit and by checksum."""
    import random
    import torch
    rnd = random.Random(20261003)
    kinds = ["zeros", "random", "text33", "alice", "lz", "periodic", "binary", "sparse"]
    edges = [0, 1, 3, 4, 5, 15, 16, 17, 63, 64, 65, 511, 512, 513, 16383, 16384, 16385, 32767, 32768, 32769, 49152, 65535, 65536]
    blocks, hlens = [], []
    for i in range(1500):
        hl = rnd.choice([0, 0, 0, 16, 48, 4096, 16384, 32768])
        room = 65536 - hl
        n = rnd.choice(edges) if rnd.random() < 0.4 else rnd.randrange(0, room + 1)
        n = min(n, room)
        body = make_block(rnd.choice(kinds), n, seed=1000 + i)
        hist = make_block(rnd.choice(kinds), hl, seed=5000 + i) if hl else b""
        if hl and rnd.random() < 0.5 and n:                     # the block repeats part of its window
            k = min(hl, n)
            body = hist[-k:] + body[k:]
        blocks.append(hist + body)
        hlens.append(hl)
    src = pack_blocks(eng, blocks, STRIDE_IN)
    dst = torch.zeros((len(blocks), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, STRIDE_IN, np.array([len(b) for b in blocks], np.uint32), dst, STRIDE_OUT,
                            STRIDE_OUT, hist_len=np.array(hlens, np.uint32), in_crc=0x1234abcd, in_adler=0x00010001)
    res, _ = eng.compress(pkg.FC_COMPRESS_RESUME_FHT, jobs, len(blocks))
    r = eng.results_to_host(res)
    out = dst.cpu().numpy()
    for i, (b, hl) in enumerate(zip(blocks, hlens)):
        exp, bits = O.deflate_fixed(b, hist=hl)
        if len(exp) > len(b):                                       # did not shrink: the engine says so
            assert r["cc"][i] == 64, (i, len(b), hl)
            continue
        assert r["cc"][i] == 0 and r["tpbc"][i] == len(exp), (i, len(b), hl, r["cc"][i])
        assert out[i, :len(exp)].tobytes() == exp, (i, len(b), hl)
        assert r["crc"][i] == zlib.crc32(b[hl:], 0x1234abcd) and r["adler"][i] == zlib.adler32(b[hl:], 0x00010001), i


def test_seeded_inflate_fuzz(eng, inflate_kernel):
    """600 zlib-made raw deflate streams (levels 0-9, default / fixed / Huffman-only / RLE strategies,
    sizes 0 .. 64 KiB, several blocks per stream through Z_FULL_FLUSH) through the batched inflate:
    output, length and both checksums."""
    import random
    import torch
    rnd = random.Random(7)
    kinds = ["zeros", "random", "text33", "alice", "lz", "periodic", "binary", "sparse"]
    streams = []
    for i in range(600):
        n = rnd.choice([0, 1, 2, 15, 16, 17, 255, 4096, 65535, 65536]) if rnd.random() < 0.3 else rnd.randrange(0, 65537)
        d = make_block(rnd.choice(kinds), n, seed=300 + i)
        co = zlib.compressobj(rnd.randrange(0, 10), zlib.DEFLATED, -15, 8,
                              rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
        c = b""
        pos = 0
        while pos < n and rnd.random() < 0.5:                       # a few flushed pieces -> several blocks
            k = rnd.randrange(1, n - pos + 1)
            c += co.compress(d[pos:pos + k]) + co.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
            pos += k
        c += co.compress(d[pos:]) + co.flush()
        streams.append((d, c))
    cstride = (max(len(c) for _, c in streams) + 64 + 15) & ~15
    ostride = 65536 + 16
    src = pack_blocks(eng, [c for _, c in streams], cstride)
    dst = torch.zeros((len(streams), ostride), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, cstride, np.array([len(c) for _, c in streams], np.uint32), dst, ostride, ostride)
    r = eng.results_to_host(eng.decompress(jobs, len(streams)))
    out = dst.cpu().numpy()
    for i, (d, c) in enumerate(streams):
        assert r["cc"][i] == 0 and r["sfbt"][i] & 0x100, (i, r["cc"][i], len(d))
        assert r["tpbc"][i] == len(d), i
        assert out[i, :len(d)].tobytes() == d, i
        assert r["crc"][i] == zlib.crc32(d) and r["adler"][i] == zlib.adler32(d), i


def test_fixed_and_stored_only_streams_whole_and_cut(eng, inflate_kernel):
    """what the fixed-code-only lane kernel keeps for itself: 300 streams of fixed-Huffman and stored blocks (zlib Z_FIXED
    at levels 0-9, several blocks each), whole and cut at a random byte, sources at every alignment: output, suspend
    state and checksums equal the CPU model's"""
    import random
    import torch
    rnd = random.Random(11)
    kinds = ["zeros", "random", "text33", "alice", "lz", "periodic", "binary", "sparse"]
    cases = []
    for i in range(300):
        n = rnd.choice([0, 1, 3, 4, 5, 17, 258, 259, 4096, 40000]) if rnd.random() < 0.3 else rnd.randrange(0, 40001)
        d = make_block(rnd.choice(kinds), n, seed=1300 + i)
        co = zlib.compressobj(rnd.randrange(0, 10), zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
        c, pos = b"", 0
        while pos < n and rnd.random() < 0.5:
            k = rnd.randrange(1, n - pos + 1)
            c += co.compress(d[pos:pos + k]) + co.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
            pos += k
        c += co.compress(d[pos:]) + co.flush()
        if i % 3 == 0 and c:
            c = c[:rnd.randrange(0, len(c))]
        cases.append((d, c))
    cstride = (max(len(c) for _, c in cases) + 64 + 15) & ~15
    ostride = 40000 + 32
    # sources at byte offsets 0..3 of a dword, targets too
    host = np.zeros((len(cases), cstride), np.uint8)
    for i, (_, c) in enumerate(cases):
        host[i, i % 4:i % 4 + len(c)] = np.frombuffer(c, np.uint8)
    src = torch.from_numpy(host).to(eng.dev)
    dst = torch.zeros((len(cases), ostride), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, cstride, np.array([len(c) for _, c in cases], np.uint32), dst, ostride, ostride - 8)
    jh = jobs.cpu().numpy().view(pkg.JOB_DTYPE).copy()
    for i in range(len(cases)):
        jh["src"][i] += i % 4
        jh["dst"][i] += (i // 4) % 4
    jobs = eng.to_device(jh)
    r = eng.results_to_host(eng.decompress(jobs, len(cases)))
    out = dst.cpu().numpy()
    for i, (d, c) in enumerate(cases):
        exp, st = O.inflate(c, ostride - 8)
        assert st.err == 0, i
        got = out[i, (i // 4) % 4:(i // 4) % 4 + st.tpbc].tobytes()
        assert r["tpbc"][i] == st.tpbc and got == exp, i
        assert (r["sfbt"][i] & 0xf) == st.out_sfbt and r["subc"][i] == st.out_subc, (i, r["sfbt"][i], st.out_sfbt)
        if (st.out_sfbt & 0xe) == 0x8:
            assert r["tebc"][i] == st.out_rembytecnt, i
        assert r["crc"][i] == zlib.crc32(exp) and r["adler"][i] == zlib.adler32(exp), i


def test_a_dynamic_block_behind_the_sample_sends_the_batch_back_to_the_general_kernel(eng):
    """the engine's own routing, no knob set: 49 152 fixed-code streams are sampled (256 of them, every 192nd), found free of
    dynamic blocks and given to the fixed-code-only lane kernel; three streams the sample does not see hold a dynamic block
    (one at its head, two behind a fixed-code block), so that kernel raises its flag and the general one does the batch
    again -- every stream's output, length and checksums are right"""
    import random
    import torch
    rnd = random.Random(5)
    n, size = 49152, 1024
    kinds = ["text33", "alice", "lz", "binary"]
    payloads = [make_block(kinds[i % 4], size, seed=2000 + i % 97) for i in range(97)]
    fixed = []
    for d in payloads:
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
        fixed.append(co.compress(d) + co.flush())
    streams = [(payloads[i % 97], fixed[i % 97]) for i in range(n)]
    odd = make_block("alice", 3000, seed=77)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    streams[7] = (odd, co.compress(odd) + co.flush())                                   # dynamic from the first byte
    for at in (12345, 49151):
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
        head = co.compress(odd[:500]) + co.flush(zlib.Z_FULL_FLUSH)                     # a fixed-code block, then a dynamic one
        tail = zlib.compressobj(6, zlib.DEFLATED, -15)
        streams[at] = (odd[:500] + odd, head + tail.compress(odd) + tail.flush())
    assert all(i % 192 for i in (7, 12345, 49151))                                      # (the sample takes every 192nd stream)
    cstride = (max(len(c) for _, c in streams) + 31) & ~15
    ostride = 3520
    src = pack_blocks(eng, [c for _, c in streams], cstride)
    dst = torch.zeros((n, ostride), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, cstride, np.array([len(c) for _, c in streams], np.uint32), dst, ostride, ostride)
    saved = {k: os.environ.pop(k, None) for k in ("NXZ_INFLATE_LANES_MIN", "NXZ_LANES_FIXED")}
    os.environ["NXZ_INFLATE_WG"] = "0"            # (a batch of this size goes a stream per workgroup since round 6: this test is the lane kernels')
    try:
        r = eng.results_to_host(eng.decompress(jobs, n))
    finally:
        os.environ.pop("NXZ_INFLATE_WG", None)
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v
    out = dst.cpu().numpy()
    assert (r["cc"] == 0).all(), np.unique(r["cc"])
    for i in list(range(0, n, 509)) + [7, 8, 12344, 12345, 49150, 49151]:
        d = streams[i][0]
        assert r["tpbc"][i] == len(d) and out[i, :len(d)].tobytes() == d, i
        assert r["crc"][i] == zlib.crc32(d) and r["adler"][i] == zlib.adler32(d), i
    want = np.array([len(d) for d, _ in streams], np.uint32)
    assert (r["tpbc"] == want).all()
    # ... and the general kernel did those three streams and no others (up to round 4 it did the whole batch again)
    back = C.c_uint32(0xffffffff)
    eng.L.nxz_ctx_lanes_handed_back.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    assert eng.L.nxz_ctx_lanes_handed_back(eng.ctx, eng.stream_handle(), n, C.byref(back)) == 0
    assert back.value == 3, back.value
    # one stream in a hundred with a dynamic block (none where the sample looks): handed back one by one, all results right
    dyn_at = [i for i in range(n) if i % 100 == 7]
    for i in dyn_at:
        streams[i] = streams[7]
    src = pack_blocks(eng, [c for _, c in streams], cstride)
    jobs = eng.jobs_strided(src, cstride, np.array([len(c) for _, c in streams], np.uint32), dst, ostride, ostride)
    saved = {k: os.environ.pop(k, None) for k in ("NXZ_INFLATE_LANES_MIN", "NXZ_LANES_FIXED")}
    os.environ["NXZ_INFLATE_WG"] = "0"            # (a batch of this size goes a stream per workgroup since round 6: this test is the lane kernels')
    try:
        r = eng.results_to_host(eng.decompress(jobs, n))
        assert eng.L.nxz_ctx_lanes_handed_back(eng.ctx, eng.stream_handle(), n, C.byref(back)) == 0
    finally:
        os.environ.pop("NXZ_INFLATE_WG", None)
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v
    assert back.value == len(set(dyn_at) | {12345, 49151}), back.value
    assert (r["cc"] == 0).all() and (r["tpbc"] == np.array([len(d) for d, _ in streams], np.uint32)).all()
    out = dst.cpu().numpy()
    for i in dyn_at[:40] + [8, 12345]:
        d = streams[i][0]
        assert out[i, :len(d)].tobytes() == d and r["crc"][i] == zlib.crc32(d), i


def test_all_35_canned_tables_encode_bit_exact(eng):
    """a11: every canned table of the reference (lib/nx_dht_builtin.c:104-840; all of them code every
    symbol) through COMPRESS_DHT on a block each of four kinds: bytes == the oracle's with the same table"""
    import torch
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "builtin_dht.json")))
    tables = [(bytes.fromhex(e["dht"]), int(e["dhtlen"])) for e in g]
    assert len(tables) == 35
    kinds = ["alice", "binary", "lz", "sparse"]
    blocks, use = [], []
    for ti in range(len(tables)):
        for k in kinds:
            blocks.append(make_block(k, 65536 if k != "binary" else 40000, seed=900 + ti))
            use.append(ti)
    dht = eng.to_device(_dht_array(tables))
    src = pack_blocks(eng, blocks, STRIDE_IN)
    dst = torch.zeros((len(blocks), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, STRIDE_IN, np.array([len(b) for b in blocks], np.uint32), dst, STRIDE_OUT,
                            STRIDE_OUT, dht_index=np.array(use, np.uint32))
    res, _ = eng.compress(pkg.FC_COMPRESS_DHT, jobs, len(blocks), dht=dht, ntables=len(tables))
    r = eng.results_to_host(res)
    out = dst.cpu().numpy()
    for i, b in enumerate(blocks):
        bits, n = tables[use[i]]
        exp, nbits = O.deflate_dynamic(b, bits, n)
        assert exp is not None, (i, use[i])
        assert r["cc"][i] in (0, 64) and r["tpbc"][i] == len(exp) and r["tebc"][i] == nbits % 8, (i, use[i], r["cc"][i])
        assert out[i, :len(exp)].tobytes() == exp, (i, use[i])


def test_second_entries_switch_follows_the_first_tile(eng):
    """oracle/nxz_lz77.c step 3b: the later tiles of a sub-block use the second bucket entries only if the first tile
    made 3072 tokens or more.  Blocks whose first tile is easy and whose rest is hard, the other way round, a first tile
    right at the threshold (token counts 2900..3300 made by isolated bytes in zeros), with and without history: fixed
    code and exact dynamic table, byte for byte the oracle's."""
    import torch
    rnd = np.random.RandomState(4)
    text = make_block("alice", 65536, seed=3) + make_block("text33", 65536, seed=4)
    json_like = (b'{"name": "entry", "id": %d, "tags": ["a", "b", "c"], "value": 3.14159},\n' * 2000)
    blocks = []
    blocks.append(bytes(16384) + text[:49152])                                    # easy first tile, hard rest
    blocks.append(text[:16384] + (json_like % ((7,) * 2000))[:49152])             # hard first tile, easy rest
    blocks.append((json_like % ((9,) * 2000))[:65536])                            # easy all the way
    blocks.append(text[1000:66536])                                               # hard all the way
    for k, lits in enumerate(range(2050, 2350, 40)):                              # first tiles around the threshold (a literal and a short run each: 1.4 tokens)
        t0 = bytearray(16384)
        pos = rnd.choice(16384 - 8, lits, replace=False)
        t0[:] = bytes(16384)
        for p in pos:
            t0[p] = 1 + rnd.randint(250)                                          # isolated bytes in a sea of zeros: a literal and a short run each
        blocks.append(bytes(t0) + text[20000 + 100 * k:20000 + 100 * k + 49152])
    hist = text[5000:5000 + 16384]
    src = pack_blocks(eng, blocks, STRIDE_IN)
    lens = np.array([len(b) for b in blocks], np.uint32)
    for fc in (pkg.FC_COMPRESS_FHT, pkg.FC_COMPRESS_DHTGEN):
        dst = torch.zeros((len(blocks), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
        jobs = eng.jobs_strided(src, STRIDE_IN, lens, dst, STRIDE_OUT, STRIDE_OUT)
        res, _ = eng.compress(fc, jobs, len(blocks))
        r = eng.results_to_host(res)
        out = dst.cpu().numpy()
        first_tile_tokens = []
        for i, b in enumerate(blocks):
            tok, nt = O.lz77(b)
            t0, _ = O.lz77(b[:16384])
            first_tile_tokens.append(_)
            if fc == pkg.FC_COMPRESS_FHT:
                exp, bits = O.deflate_fixed(b)
            else:
                ll, d = O.counts(tok, nt)
                dht, dhtlen = O.dhtgen(ll, d)
                exp, bits = O.deflate_dynamic(b, dht, dhtlen)
            assert r["cc"][i] == 0 and r["tpbc"][i] == len(exp), (fc, i)
            assert out[i, :len(exp)].tobytes() == exp, (fc, i)
            dz = zlib.decompressobj(-15)
            assert dz.decompress(exp) == b and dz.eof
        # both sides of the threshold are among the cases (else the test proves nothing)
        assert min(first_tile_tokens) < 3072 <= max(first_tile_tokens), sorted(first_tile_tokens)
    # and with a window in front of the block (the first TILE OF THE BLOCK decides, not the window)
    withh = [hist + b[:65536 - len(hist)] for b in blocks[:4]]
    src = pack_blocks(eng, withh, STRIDE_IN)
    dst = torch.zeros((len(withh), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, STRIDE_IN, np.full(len(withh), 65536, np.uint32), dst, STRIDE_OUT, STRIDE_OUT,
                            hist_len=np.full(len(withh), len(hist), np.uint32))
    res, _ = eng.compress(pkg.FC_COMPRESS_RESUME_FHT, jobs, len(withh))
    r = eng.results_to_host(res)
    out = dst.cpu().numpy()
    for i, b in enumerate(withh):
        exp, bits = O.deflate_fixed(b, hist=len(hist))
        assert r["tpbc"][i] == len(exp) and out[i, :len(exp)].tobytes() == exp, i


def test_text_rule_follows_the_first_tile(eng):
    """oracle/nxz_lz77.c step 3c: a sub-block whose first tile has fewer than one byte in 16 with its top bit set and
    made 3072 tokens or more does without the second entries and without the lazy step in its later tiles.  Blocks on
    both sides of the byte threshold (1023 / 1024 such bytes in 16384), bytes above 0x7f only behind the first tile, a
    block shorter than a tile, easy text (few tokens), with and without history: byte for byte the oracle's."""
    import torch
    rnd = np.random.RandomState(9)
    text = make_block("alice", 65536, seed=3) + make_block("text33", 65536, seed=4) + make_block("alice", 65536, seed=8)

    def with_high(b, k, lo, hi):
        b = bytearray(b)
        for p in rnd.choice(hi - lo, k, replace=False):
            b[lo + p] |= 0x80
        return bytes(b)

    blocks = [text[:65536], with_high(text[300:65836], 1023, 0, 16384), with_high(text[300:65836], 1024, 0, 16384),
              with_high(text[700:66236], 1100, 0, 16384), with_high(text[900:66436], 9000, 16384, 65536),
              text[2000:12000], with_high(text[2000:12000], 624, 0, 10000), with_high(text[2000:12000], 625, 0, 10000),
              (b"key = value; " * 6000)[:65536], text[40000:40000 + 16384 + 5], text[50000:50000 + 32768]]
    src = pack_blocks(eng, blocks, STRIDE_IN)
    lens = np.array([len(b) for b in blocks], np.uint32)
    for fc in (pkg.FC_COMPRESS_FHT, pkg.FC_COMPRESS_DHTGEN):
        dst = torch.zeros((len(blocks), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
        jobs = eng.jobs_strided(src, STRIDE_IN, lens, dst, STRIDE_OUT, STRIDE_OUT)
        res, _ = eng.compress(fc, jobs, len(blocks))
        r = eng.results_to_host(res)
        out = dst.cpu().numpy()
        for i, b in enumerate(blocks):
            if fc == pkg.FC_COMPRESS_FHT:
                exp, bits = O.deflate_fixed(b)
            else:
                tok, nt = O.lz77(b)
                ll, d = O.counts(tok, nt)
                dht, dhtlen = O.dhtgen(ll, d)
                exp, bits = O.deflate_dynamic(b, dht, dhtlen)
            assert r["cc"][i] == 0 and r["tpbc"][i] == len(exp), (fc, i)
            assert out[i, :len(exp)].tobytes() == exp, (fc, i)
            dz = zlib.decompressobj(-15)
            assert dz.decompress(exp) == b and dz.eof
    hist = text[5000:5000 + 16384]
    withh = [hist + b[:65536 - len(hist)] for b in blocks[:5]]
    src = pack_blocks(eng, withh, STRIDE_IN)
    dst = torch.zeros((len(withh), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, STRIDE_IN, np.full(len(withh), 65536, np.uint32), dst, STRIDE_OUT, STRIDE_OUT,
                            hist_len=np.full(len(withh), len(hist), np.uint32))
    res, _ = eng.compress(pkg.FC_COMPRESS_RESUME_FHT, jobs, len(withh))
    r = eng.results_to_host(res)
    out = dst.cpu().numpy()
    for i, b in enumerate(withh):
        exp, bits = O.deflate_fixed(b, hist=len(hist))
        assert r["tpbc"][i] == len(exp) and out[i, :len(exp)].tobytes() == exp, i


def test_trim_gives_the_token_scratch_back_and_the_next_batch_works(eng):
    """nxz_trim(): the scratch a dynamic-table batch left on its stream (tokens, tables, counts of a chunk) goes back to
    the device; the next batch allocates again and gives the same bytes."""
    import torch
    blocks = [make_block(k, 65536, seed=70 + i) for i, k in enumerate(["alice", "lz", "binary", "text33"] * 8)]
    src = pack_blocks(eng, blocks, STRIDE_IN)
    lens = np.full(len(blocks), 65536, np.uint32)
    outs = []
    for rnd in range(2):
        dst = torch.zeros((len(blocks), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
        jobs = eng.jobs_strided(src, STRIDE_IN, lens, dst, STRIDE_OUT, STRIDE_OUT)
        res, _ = eng.compress(pkg.FC_COMPRESS_DHTGEN, jobs, len(blocks))
        r = eng.results_to_host(res)
        assert (r["cc"] == 0).all()
        outs.append((r["tpbc"].copy(), dst.cpu().numpy().copy()))
        if rnd == 0:
            eng.L.nxz_trim.restype = C.c_size_t
            freed = eng.L.nxz_trim()
            assert freed >= len(blocks) * 106496, freed
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()


def test_fused_dhtgen_kernel_equals_the_three_kernels(eng):
    """NXZ_FUSED_GEN=1: LZ77, table and encode in one kernel (nxz_lz77.hip gen::: the table of a block is made by one wavefront
    while the others encode the block before it; a workgroup's last block behind its loop).  The same bytes, completion codes
    and symbol counts as the three kernels, for batches of one job per workgroup, of several, and with ragged and empty blocks;
    the oracle's for a sample."""
    import torch
    kinds = ["alice", "lz", "binary", "text33", "zeros", "random", "periodic", "sparse"]
    for nb in (1, 5, 700):
        sizes = [65536 if i % 7 else [0, 1, 30000 + i, 16384, 16385, 65535][i % 6] for i in range(nb)]
        blocks = [make_block(kinds[i % len(kinds)], sizes[i], seed=900 + i) for i in range(nb)]
        src = pack_blocks(eng, blocks, STRIDE_IN)
        lens = np.array([len(b) for b in blocks], np.uint32)
        got = {}
        for mode in ("0", "1"):
            os.environ["NXZ_FUSED_GEN"] = mode
            try:
                dst = torch.zeros((nb, STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
                jobs = eng.jobs_strided(src, STRIDE_IN, lens, dst, STRIDE_OUT, STRIDE_OUT)
                res, cnt = eng.compress(pkg.FC_COMPRESS_DHTGEN_COUNT, jobs, nb)
                r = eng.results_to_host(res)
                got[mode] = (r.copy(), dst.cpu().numpy().copy(), cnt.cpu().numpy().copy())
            finally:
                os.environ.pop("NXZ_FUSED_GEN", None)
        a, b = got["0"], got["1"]
        for f in ("cc", "tpbc", "tebc", "spbc", "crc", "adler"):
            assert (a[0][f] == b[0][f]).all(), (nb, f)
        assert (a[2] == b[2]).all(), nb
        for i in range(nb):
            n = int(a[0]["tpbc"][i])
            assert a[1][i, :n].tobytes() == b[1][i, :n].tobytes(), (nb, i)
        for i in range(0, nb, max(1, nb // 12)):
            tok, nt = O.lz77(blocks[i])
            ll, d = O.counts(tok, nt)
            dht, dhtlen = O.dhtgen(ll, d)
            exp, bits = O.deflate_dynamic(blocks[i], dht, dhtlen)
            assert b[0]["tpbc"][i] == len(exp) and b[1][i, :len(exp)].tobytes() == exp, (nb, i)


def test_missing_code_check_never_fires_on_device_made_tables(eng):
    """NXZ_ENCODE_CHECK=1 sends the tables the device made of a block's own counts through the kernel form that looks for
    symbols without a code (encode_kernel<true, true>, which a caller's table always takes; the default form has the check
    compiled out -- round 4's advisor finding).  The LZ77 kernel's counts and its tokens agree and dhtgen gives every counted
    symbol a code (lib/nx_dhtgen.c:252-270): no job answers cc 11, and the blocks are byte for byte those of the default form."""
    import torch
    kinds = ["alice", "lz", "binary", "text33", "zeros", "random", "periodic", "sparse"]
    sizes = [65536 if i % 5 else [1, 2, 258, 30000 + i, 65535][(i // 5) % 5] for i in range(600)]
    blocks = [make_block(kinds[i % len(kinds)], sizes[i], seed=4200 + i) for i in range(600)]
    nb = len(blocks)
    src = pack_blocks(eng, blocks, STRIDE_IN)
    lens = np.array([len(b) for b in blocks], np.uint32)
    got = {}
    for mode in ("0", "1"):
        os.environ["NXZ_ENCODE_CHECK"] = mode
        try:
            dst = torch.zeros((nb, STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
            jobs = eng.jobs_strided(src, STRIDE_IN, lens, dst, STRIDE_OUT, STRIDE_OUT)
            res, cnt = eng.compress(pkg.FC_COMPRESS_DHTGEN_COUNT, jobs, nb)
            r = eng.results_to_host(res)
            got[mode] = (r.copy(), dst.cpu().numpy().copy())
        finally:
            os.environ.pop("NXZ_ENCODE_CHECK", None)
    a, b = got["0"], got["1"]
    assert (b[0]["cc"] != 11).all()
    for f in ("cc", "tpbc", "tebc"):
        assert (a[0][f] == b[0][f]).all(), f
    for i in range(nb):
        n = int(a[0]["tpbc"][i])
        assert a[1][i, :n].tobytes() == b[1][i, :n].tobytes(), i
