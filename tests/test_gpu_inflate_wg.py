"""GPU tests of the workgroup-per-stream inflate kernel (nxz_inflate_wg.hip) through the C ABI: bytes against the source data and
system zlib's checksums, result records against the CPU oracle (oracle/nxz_inflate.c) -- also for the streams the kernel hands back
to the stream-per-wavefront kernel --, the reasons it gives, and the engine's routing of batches by size."""
import importlib
import os
import random
import zlib

import numpy as np
import pytest

import oracle_lib as O
from datagen import make_block

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("power-gzip_amd")


@pytest.fixture(scope="module")
def eng():
    saved = {k: os.environ.pop(k, None) for k in ("NXZ_INFLATE_LANES_MIN", "NXZ_INFLATE_CUT", "NXZ_INFLATE_WG", "NXZ_WG_PMIN")}
    e = pkg.Engine(0)
    yield e
    e.close()
    for k, v in saved.items():
        if v is not None:
            os.environ[k] = v


def _run(eng, cases, cap=65536, offsets=None, force=True):
    """cases: list of (plain, stream).  Returns results (host) and the outputs."""
    import torch
    n = len(cases)
    cstride = (max(len(c) for _, c in cases) + 64 + 15) & ~15
    host = np.full((n, cstride), 0xa5, np.uint8)
    offs = offsets or [0] * n
    for i, (_, c) in enumerate(cases):
        host[i, offs[i]:offs[i] + len(c)] = np.frombuffer(c, np.uint8)
    src = torch.from_numpy(host).to(eng.dev)
    ostride = (cap + 15) & ~15
    dst = torch.full((n, ostride + 16), 0xcd, dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, cstride, np.array([len(c) for _, c in cases], np.uint32), dst, ostride + 16, cap)
    if offsets:
        jh = jobs.cpu().numpy().view(pkg.JOB_DTYPE).copy()
        jh["src"] += np.array(offs, np.uint64)
        jobs = eng.to_device(jh)
    if force:
        os.environ["NXZ_INFLATE_WG"] = "1"
    try:
        r = eng.results_to_host(eng.decompress(jobs, n))
        why = eng.wg_reasons()
    finally:
        os.environ.pop("NXZ_INFLATE_WG", None)
    return r, dst.cpu().numpy(), why


def _streams(seed, count):
    rnd = random.Random(seed)
    kinds = ["alice", "lz", "text33", "zeros", "random", "periodic", "binary", "sparse"]
    out = []
    for i in range(count):
        kind = kinds[i % len(kinds)]
        n = rnd.choice([0, 1, 2, 5, 63, 64, 65, 300, 4095, 4096, 16384, 40000, 65535, 65536]) if i % 3 == 0 else rnd.randrange(1, 65537)
        d = make_block(kind, n, seed=seed * 1000 + i)
        level, strat = rnd.choice([(1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED),
                                   (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (4, zlib.Z_FILTERED)])
        co = zlib.compressobj(level, zlib.DEFLATED, -15, rnd.choice([8, 9, 5]), strat)
        out.append((d, co.compress(d) + co.flush()))
    return out


def test_fresh_streams_of_every_kind_bit_exact_and_not_handed_back(eng):
    cases = _streams(7, 400)
    r, out, why = _run(eng, cases)
    for i, (d, c) in enumerate(cases):
        assert r["cc"][i] == 0 and r["sfbt"][i] == 0x100 and r["tpbc"][i] == len(d) and r["spbc"][i] == len(c) and r["subc"][i] < 8, (i, len(d))
        assert out[i, :len(d)].tobytes() == d, i
        assert (out[i, len(d):len(d) + 8] == 0xcd).all() or len(d) % 16, i             # nothing written behind a whole output
        assert r["crc"][i] == zlib.crc32(d) and r["adler"][i] == zlib.adler32(d), i
    # streams whose source does not fit the LDS image (stored random blocks: 64 KiB and the block headers) are the only ones handed back
    assert why is not None and set(why) <= {"handed_back", "job", "rounds"}, why
    assert why["handed_back"] <= sum(1 for d, c in cases if len(c) > 65000) + why.get("rounds", 0), why


def test_unaligned_sources_and_multi_block_streams(eng):
    rnd = random.Random(5)
    cases, offs = [], []
    for i in range(96):
        d = make_block(["alice", "lz", "binary"][i % 3], rnd.randrange(20000, 65537), seed=i)
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        k = len(d) // 3
        c = co.compress(d[:k]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(d[k:2 * k]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(d[2 * k:]) + co.flush()
        cases.append((d, c)); offs.append(i % 16)
    r, out, why = _run(eng, cases, offsets=offs)
    for i, (d, c) in enumerate(cases):
        assert r["cc"][i] == 0 and r["tpbc"][i] == len(d) and out[i, :len(d)].tobytes() == d, i
        assert r["crc"][i] == zlib.crc32(d), i
    assert why["handed_back"] == 0, why


def test_what_the_kernel_hands_back_comes_out_as_the_oracle_says(eng):
    """streams cut short, damaged, with bytes behind the final block, with a target that is too small: the kernel takes none of them
    on trust -- every one is either decoded to the bytes and the record of the oracle by the kernel itself (trailing bytes) or
    handed back and decoded by the stream-per-wavefront kernel, whose results are the oracle's."""
    rnd = random.Random(9)
    base = _streams(3, 40)
    cases = []
    for k in range(240):
        d, c = base[k % len(base)]
        b = bytearray(c)
        how = k % 4
        if how == 0 and len(b) > 4:
            del b[rnd.randrange(1, len(b)):]
        elif how == 1 and b:
            for _ in range(rnd.randrange(1, 4)):
                i = rnd.randrange(len(b)); b[i] ^= 1 << rnd.randrange(8)
        elif how == 2:
            b += bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 40)))
        cases.append((d, bytes(b)))
    cap = 65536
    r, out, why = _run(eng, cases, cap=cap)
    for i, (d, c) in enumerate(cases):
        exp, st = O.inflate(c, cap)
        if st.err:
            assert r["cc"][i] == st.err, (i, r["cc"][i], st.err)
            continue
        assert r["cc"][i] in (0, 3), i
        assert r["tpbc"][i] == st.tpbc and out[i, :st.tpbc].tobytes() == exp, i
        assert (r["sfbt"][i] & 0xf) == st.out_sfbt and r["subc"][i] == st.out_subc, i
    assert why["handed_back"] > 0


def test_targets_that_are_too_small_are_handed_back(eng):
    cases = [(d, c) for d, c in _streams(11, 60) if len(d) > 100]
    r, out, why = _run(eng, cases, cap=96)
    for i, (d, c) in enumerate(cases):
        assert r["cc"][i] == 13, i                                                 # NXZ_CC_TARGET_SPACE
        assert (out[i, 96:112] == 0xcd).all(), i
    assert why["space"] + why.get("job", 0) + why.get("stored", 0) == why["handed_back"] == len(cases), why


def test_the_engine_sends_batches_up_to_its_limit_through_this_kernel(eng):
    cases = _streams(13, 300)
    r, out, why = _run(eng, cases, force=False)
    assert why is not None, "a batch of 300 streams did not go a stream per workgroup"
    for i, (d, c) in enumerate(cases):
        assert r["cc"][i] == 0 and out[i, :len(d)].tobytes() == d, i
    os.environ["NXZ_INFLATE_WG_MAX"] = "100"
    try:
        r2, out2, _ = _run(eng, cases, force=False)                                # (beyond the limit: the older kernels, same results)
        for i, (d, c) in enumerate(cases):
            assert r2["cc"][i] == 0 and out2[i, :len(d)].tobytes() == d, i
        assert (r2["crc"] == r["crc"]).all() and (r2["tpbc"] == r["tpbc"]).all()
    finally:
        os.environ.pop("NXZ_INFLATE_WG_MAX", None)


def test_streams_longer_than_lds_go_in_spans(eng):
    """a stream of any length is this kernel's too: the source through a 64 KiB window, the output flushed 32 KiB and more at a
    time, the last 32 KiB staying in LDS as the window of distances.  Every kind of data, level and block type; multi-block;
    stored blocks that lie across windows; output sizes round the edges of the halves."""
    rnd = random.Random(21)
    cases = []
    sizes = [65537, 70000, 98304, 98305, 131072, 200000, 262144, 300001, 524288, 1 << 20, (1 << 20) + 17, 2 << 20]
    kinds = ["alice", "lz", "text33", "zeros", "random", "periodic", "binary", "sparse"]
    for i in range(48):
        n = sizes[i % len(sizes)] if i < 36 else rnd.randrange(65537, 1 << 20)
        d = make_block(kinds[i % len(kinds)], n, seed=400 + i)
        level, strat = [(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY),
                        (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)][i % 7]
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
        if i % 5 == 4:
            k = n // 2
            c = co.compress(d[:k]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(d[k:]) + co.flush()
        else:
            c = co.compress(d) + co.flush()
        cases.append((d, c))
    r, out, why = _run(eng, cases, cap=(2 << 20) + 64, offsets=[i % 16 for i in range(len(cases))])
    for i, (d, c) in enumerate(cases):
        assert r["cc"][i] == 0 and r["sfbt"][i] == 0x100 and r["tpbc"][i] == len(d) and r["spbc"][i] == len(c), (i, len(d), r["cc"][i], r["tpbc"][i])
        assert out[i, :len(d)].tobytes() == d, i
        assert (out[i, len(d):len(d) + 8] == 0xcd).all(), i
        assert r["crc"][i] == zlib.crc32(d) and r["adler"][i] == zlib.adler32(d), i
    print("long streams handed back:", why)
    assert why["handed_back"] <= 4, why
