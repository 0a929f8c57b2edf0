"""GPU tests of the device-side table generator and of the compress function codes that use it
(-m gpu): nxz_batch_dhtgen against the golden vectors made with the reference's own nx_dhtgen.c
and against the oracle's restatement on thousands of count arrays; NXZ_FC_COMPRESS_DHTGEN against
the oracle's LZ77 + nxo_dhtgen + encoder, bit for bit, with zlib inflating every block."""
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
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRIDE_IN = 65536
STRIDE_OUT = 73856


@pytest.fixture(scope="module")
def eng():
    e = pkg.Engine(0)
    yield e
    e.close()


def _oracle_table(ll, d):
    a = (C.c_uint32 * 286)(*[int(x) for x in ll])
    b = (C.c_uint32 * 30)(*[int(x) for x in d])
    return O.dhtgen(a, b)


def _device_tables(eng, counts):
    import torch
    n = counts.shape[0]
    dev = torch.from_numpy(counts.astype(np.uint32).view(np.int32).reshape(-1).copy()).to(eng.dev)
    t = eng.dhtgen(dev, n)
    torch.cuda.synchronize(eng.dev)
    return t.cpu().numpy().view(pkg.DHT_DTYPE)


def test_device_dhtgen_golden_vectors(eng):
    vecs = json.load(open(os.path.join(ROOT, "tests", "golden", "dhtgen_vectors.json")))
    rows, exp = [], []
    for v in vecs:
        ll, d = list(v["ll"]), list(v["d"])
        ll += [0] * (286 - len(ll))
        d += [0] * (30 - len(d))
        if v["flag"] == "-f":
            ll = [c or 1 for c in ll]
            d = [c or 1 for c in d]
        elif v["flag"] == "-g":
            ll = ll[:257] + [c or 1 for c in ll[257:]]
            d = [c or 1 for c in d]
        num_ll = max(257, max(i + 1 for i, c in enumerate(ll) if c))
        num_d = max([i + 1 for i, c in enumerate(d) if c] or [0])
        if num_ll != 286 or num_d != 30:
            continue                      # the device builds full-size tables (HLIT 286, HDIST 30) only
        rows.append(ll + d)
        exp.append((v["name"], v["dht"], int(v["dhtlen"])))
    assert len(rows) >= 8
    t = _device_tables(eng, np.array(rows, np.uint32))
    for i, (name, dht, dhtlen) in enumerate(exp):
        assert t["dhtlen"][i] == dhtlen, name
        nb = (dhtlen + 7) // 8
        assert t["dht"][i, :nb].tobytes().hex() == dht[:nb * 2], name


def _random_counts(rng, n):
    counts = np.zeros((n, 316), np.uint32)
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    for i in range(n):
        kind = i % 7
        if kind == 0:
            counts[i, :286] = rng.integers(0, 300, 286); counts[i, 286:] = rng.integers(0, 200, 30)
        elif kind == 1:
            k = rng.integers(1, 60)
            counts[i, rng.choice(286, k, replace=False)] = rng.integers(1, 5000, k)
            kd = rng.integers(0, 9)
            if kd:
                counts[i, 286 + rng.choice(30, kd, replace=False)] = rng.integers(1, 3000, kd)
        elif kind == 2:
            idx = rng.choice(286, 30, replace=False)
            counts[i, idx] = fib[:30]
            counts[i, 286:286 + 24] = fib[:24]
        elif kind == 3:
            v = (60000 * 0.9 ** np.arange(286)).astype(np.uint32) + rng.integers(0, 2, 286).astype(np.uint32)
            counts[i, :286] = rng.permutation(v)
            counts[i, 286:] = rng.choice([0, 1, 1, 2, 50], 30)
        elif kind == 4:
            counts[i, :286] = rng.choice([0, 0, 0, 1, 2], 286)
        elif kind == 5:
            counts[i, :286] = rng.integers(1, 4, 286); counts[i, 286:] = 1
        else:
            counts[i, :286] = rng.integers(0, 1 << 24, 286); counts[i, 286:] = rng.integers(0, 1 << 24, 30)   # saturated-size counts
        counts[i, 256] = 1
    return counts


def test_device_dhtgen_equals_oracle_on_many_count_arrays(eng):
    rng = np.random.default_rng(11)
    n = 2100
    counts = _random_counts(rng, n)
    t = _device_tables(eng, counts)
    for i in range(n):
        exp, dhtlen = _oracle_table(counts[i, :286], counts[i, 286:])
        assert t["dhtlen"][i] == dhtlen, (i, i % 7)
        assert t["dht"][i, :len(exp)].tobytes() == exp, (i, i % 7)


def _oracle_dhtgen_block(b, hist=0):
    tok, nt = O.lz77(b, hist)
    ll, d = O.counts(tok, nt)
    cnt = np.array(list(ll) + list(d), np.uint32)
    dht, dhtlen = O.dhtgen(ll, d)
    cap = 2 * len(b) + 2048
    out = C.create_string_buffer(cap)
    bits = O.lib().nxo_encode_dynamic(tok, nt, dht, dhtlen, out, cap)
    assert bits < (1 << 62)
    return out.raw[:(bits + 7) // 8], bits, cnt


CASES = [("zeros", 65536), ("text33", 65536), ("lz", 65536), ("random", 65536), ("alice", 65536), ("periodic", 65536),
         ("binary", 65536), ("sparse", 65536), ("lz", 0), ("lz", 1), ("lz", 5), ("text33", 17), ("alice", 40000),
         ("lz", 16385), ("zeros", 300), ("random", 5000), ("binary", 33333), ("alice", 65535)]


def test_dhtgen_function_code_bit_exact(eng):
    import torch
    blocks = [make_block(k, n, seed=50 + i) for i, (k, n) in enumerate(CASES)]
    blocks += [make_block(k, 65536, seed=200 + i) for i, k in enumerate(["alice", "binary", "sparse", "lz"] * 8)]
    host = np.zeros((len(blocks), STRIDE_IN), np.uint8)
    for i, b in enumerate(blocks):
        host[i, :len(b)] = np.frombuffer(b, np.uint8)
    src = torch.from_numpy(host).to(eng.dev)
    dst = torch.zeros((len(blocks), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    lens = np.array([len(b) for b in blocks], np.uint32)
    jobs = eng.jobs_strided(src, STRIDE_IN, lens, dst, STRIDE_OUT, STRIDE_OUT)
    res, cnt = eng.compress(pkg.FC_COMPRESS_DHTGEN_COUNT, jobs, len(blocks))
    r = eng.results_to_host(res)
    out = dst.cpu().numpy()
    cnt = cnt.cpu().numpy().view(np.uint32).reshape(len(blocks), 316)
    for i, b in enumerate(blocks):
        exp, bits, ocnt = _oracle_dhtgen_block(b)
        assert (cnt[i] == ocnt).all(), i
        assert r["cc"][i] in (0, 64), (i, r["cc"][i])
        assert r["tpbc"][i] == len(exp), i
        assert r["tebc"][i] == bits % 8, i
        assert out[i, :len(exp)].tobytes() == exp, i
        assert r["crc"][i] == zlib.crc32(b) and r["adler"][i] == zlib.adler32(b), i
        z = zlib.decompressobj(-15)
        assert z.decompress(exp) == b and z.eof, i
    # the code without the COUNT bit gives the same bytes
    dst2 = torch.zeros_like(dst)
    jobs2 = eng.jobs_strided(src, STRIDE_IN, lens, dst2, STRIDE_OUT, STRIDE_OUT)
    r2 = eng.results_to_host(eng.compress(pkg.FC_COMPRESS_DHTGEN, jobs2, len(blocks))[0])
    assert (r2["tpbc"] == r["tpbc"]).all() and (r2["cc"] == r["cc"]).all()
    o2 = dst2.cpu().numpy()
    for i in range(len(blocks)):
        assert o2[i, :r["tpbc"][i]].tobytes() == out[i, :r["tpbc"][i]].tobytes(), i


def test_dhtgen_with_history(eng):
    import torch
    hist = 32768
    blocks = [make_block("alice", 65536, seed=300 + i) for i in range(6)]
    host = np.stack([np.frombuffer(b, np.uint8) for b in blocks])
    src = torch.from_numpy(host.copy()).to(eng.dev)
    dst = torch.zeros((len(blocks), STRIDE_OUT), dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, STRIDE_IN, np.full(len(blocks), 65536, np.uint32), dst, STRIDE_OUT, STRIDE_OUT, hist_len=hist)
    r = eng.results_to_host(eng.compress(pkg.FC_COMPRESS_RESUME_DHTGEN, jobs, len(blocks))[0])
    out = dst.cpu().numpy()
    for i, b in enumerate(blocks):
        exp, bits, _ = _oracle_dhtgen_block(b, hist)
        assert r["tpbc"][i] == len(exp) and out[i, :len(exp)].tobytes() == exp, i
        z = zlib.decompressobj(-15, zdict=b[:hist])
        assert z.decompress(exp) == b[hist:], i
