"""GPU parity on the metric's kind of data (BASELINE.json configs[2] / configs[3]):
every unique 64 KiB chunk of the real-data corpus (Silesia through $SILESIA_DIR, sha256-checked against the
reference's oct/silesia-*.source pins, else the recorded fallback: alice29.txt + system files of every class,
tests/corpus.py) through the engine, called through the C ABI.

  * COMPRESS_DHTGEN (LZ77 kernel -> device dhtgen -> entropy kernel): bytes == the oracle's, zlib inflates
    every block, compressed size per class >= 0.95 x zlib -1 on the identical chunks (the target's floor);
  * the same chunks as zlib -6 raw streams through all three inflate kernels: bytes == source, CRC == zlib's.
"""
import ctypes as C
import importlib
import os
import zlib

import numpy as np
import pytest

import corpus
import oracle_lib as O

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("power-gzip_amd")

BLOCK = 65536
STRIDE_OUT = 73856


@pytest.fixture(scope="module")
def eng():
    e = pkg.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def blocks():
    name, b, report = corpus.load(BLOCK)
    assert len(b) >= 3, report
    return name, b


def _oracle_dhtgen(b):
    tok, nt = O.lz77(b)
    ll, d = O.counts(tok, nt)
    dht, dhtlen = O.dhtgen(ll, d)
    cap = 2 * len(b) + 2048
    out = C.create_string_buffer(cap)
    bits = O.lib().nxo_encode_dynamic(tok, nt, dht, dhtlen, out, cap)
    assert bits < (1 << 62)
    return out.raw[:(bits + 7) // 8], bits


def test_every_corpus_block_dhtgen_equals_oracle_and_keeps_the_ratio_floor(eng, blocks):
    import torch
    name, bl = blocks
    raw = [b for _, _, b in bl]
    host = np.zeros((len(raw), BLOCK), np.uint8)
    for i, b in enumerate(raw):
        host[i, :len(b)] = np.frombuffer(b, np.uint8)
    src = torch.from_numpy(host).to(eng.dev)
    dst = torch.zeros((len(raw), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    lens = np.array([len(b) for b in raw], np.uint32)
    jobs = eng.jobs_strided(src, BLOCK, lens, dst, STRIDE_OUT, STRIDE_OUT)
    r = eng.results_to_host(eng.compress(pkg.FC_COMPRESS_DHTGEN, jobs, len(raw))[0])
    out = dst.cpu().numpy()
    per = {}
    for i, (cls, fname, b) in enumerate(bl):
        exp, bits = _oracle_dhtgen(b)
        assert r["cc"][i] in (0, 64), (i, cls, fname, r["cc"][i])
        assert r["tpbc"][i] == len(exp) and r["tebc"][i] == bits % 8, (i, cls, fname)
        got = out[i, :len(exp)].tobytes()
        assert got == exp, (i, cls, fname)
        assert r["crc"][i] == zlib.crc32(b) and r["adler"][i] == zlib.adler32(b), (i, cls, fname)
        z = zlib.decompressobj(-15)
        assert z.decompress(got) == b and z.eof, (i, cls, fname)
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        a = per.setdefault(cls, [0, 0])
        a[0] += len(exp)
        a[1] += len(c.compress(b) + c.flush())
    # the target's floor: >= 0.95 x zlib -1's ratio, on every class of data
    for cls, (ours, z1) in per.items():
        assert z1 / ours >= 0.95, (name, cls, z1 / ours)
    assert sum(v[1] for v in per.values()) / sum(v[0] for v in per.values()) >= 0.97


@pytest.mark.parametrize("kernel", ["lanes", "waves", "waves-global-window"])
def test_every_corpus_block_as_a_zlib6_stream_inflates_on_every_kernel(blocks, kernel):
    import torch
    old = {k: os.environ.get(k) for k in ("NXZ_INFLATE_LANES_MIN", "NXZ_INFLATE_LDS_MAX")}
    os.environ["NXZ_INFLATE_LANES_MIN"] = "32" if kernel == "lanes" else "1000000000"
    if kernel == "waves-global-window":
        os.environ["NXZ_INFLATE_LDS_MAX"] = "0"
    e = pkg.Engine(0)
    try:
        _, bl = blocks
        raw = [b for _, _, b in bl]
        streams = []
        for b in raw:
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            streams.append(c.compress(b) + c.flush())
        cstride = (max(len(s) for s in streams) + 64 + 15) & ~15
        host = np.zeros((len(raw), cstride), np.uint8)
        for i, s in enumerate(streams):
            host[i, :len(s)] = np.frombuffer(s, np.uint8)
        src = torch.from_numpy(host).to(e.dev)
        dst = torch.zeros((len(raw), BLOCK), dtype=torch.uint8, device=e.dev)
        jobs = e.jobs_strided(src, cstride, np.array([len(s) for s in streams], np.uint32), dst, BLOCK, BLOCK)
        r = e.results_to_host(e.decompress(jobs, len(raw)))
        got = dst.cpu().numpy()
        for i, (cls, fname, b) in enumerate(bl):
            assert r["cc"][i] == 0 and r["tpbc"][i] == len(b), (kernel, i, cls, fname, r["cc"][i])
            assert got[i, :len(b)].tobytes() == b, (kernel, i, cls, fname)
            assert r["crc"][i] == zlib.crc32(b) and r["adler"][i] == zlib.adler32(b), (kernel, i, cls, fname)
    finally:
        e.close()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_c5_mixed_batch_round_trip():
    """BASELINE configs[4] as one step of bench.py --config c5 on >= 16384 blocks: zeros / text /
    makedata copies / random bytes by block index mod 4 -> FHT -> what does not shrink (the random quarter,
    CC 64) stored through WRAP -> decompress (WRAP for the stored ones) -> compare on the device."""
    import torch
    import bench
    e = pkg.Engine(0)
    try:
        n = 16384
        src = bench.gen_mixed(torch, e.dev, n, 0)
        step, info = bench.c5_prepare(torch, e, pkg, src)
        step()
        torch.cuda.synchronize(e.dev)
        assert info["stored"] == n // 4                           # exactly the random quarter is stored
        assert (info["stored_index"] % 4 == 3).all()
        assert int(info["flag"].item()) == 0                      # decompressed == source, every byte
        assert torch.equal(info["back"], src)
        # the stored blocks really are what WRAP makes of them: the source bytes
        i = int(info["stored_index"][0])
        assert torch.equal(info["comp"][i, :BLOCK], src[i])
        # and a sample of the compressed ones inflates with zlib
        r = info["results"]
        comp = info["comp"]
        for i in (0, 1, 2, 4, 5, 6, n - 4, n - 3, n - 2):
            z = zlib.decompressobj(-15)
            assert z.decompress(comp[i, :int(r["tpbc"][i])].cpu().numpy().tobytes()) == src[i].cpu().numpy().tobytes() and z.eof, i
    finally:
        e.close()


def test_parse_pass2_does_not_read_what_other_walks_write():
    """Regression (round 3): in pass 2 of the parse a walk that has stepped over the start of the speculative
    walk's last match emits the rest of that match; its distance must be the one pass 1 saw -- another entered
    segment's walk may put the rest of ITS last match on that very position meanwhile.  Block 43729 of the
    synthetic recipe (bench.gen_blocks) hit this in about one pass out of sixteen: many copies of it and of its
    neighbours, several passes, every output equal to the oracle's."""
    import torch
    import bench
    e = pkg.Engine(0)
    try:
        base = bench.gen_blocks(torch, e.dev, 8, 43726)              # 43726 .. 43733
        src = base.repeat(512, 1)                                     # 4096 jobs
        n = src.shape[0]
        dst = torch.zeros((n, STRIDE_OUT), dtype=torch.uint8, device=e.dev)
        jobs = e.jobs_strided(src, BLOCK, np.full(n, BLOCK, np.uint32), dst, STRIDE_OUT, STRIDE_OUT)
        exp = [O.deflate_fixed(base[i].cpu().numpy().tobytes())[0] for i in range(8)]
        want = torch.zeros((8, STRIDE_OUT), dtype=torch.uint8, device=e.dev)
        for i, x in enumerate(exp):
            want[i, :len(x)] = torch.from_numpy(np.frombuffer(x, np.uint8).copy()).to(e.dev)
        for _ in range(12):
            dst.zero_()
            r = e.results_to_host(e.compress(pkg.FC_COMPRESS_FHT, jobs, n)[0])
            assert (r["tpbc"].reshape(512, 8) == np.array([len(x) for x in exp], np.uint32)).all()
            assert torch.equal(dst, want.repeat(512, 1))
    finally:
        e.close()
