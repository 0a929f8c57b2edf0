"""The product's host table builder (power-gzip_amd/csrc/nxz_dht.cpp, built into the CPU model library
for these tests) against the golden vectors made with the reference's own nx_dhtgen.c, and the
batched entry against the single one."""
import ctypes as C
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    L = C.CDLL(os.path.join(ROOT, "oracle", "libnxz_amd_model.so"))
    L.nxz_dhtgen.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_char_p,
                             C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.nxz_dhtgen_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    L.nxz_fill_zero_lzcounts.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32]
    return L


def _gen(L, ll, d, num_ll, num_d):
    buf = C.create_string_buffer(320)
    nb, vb = C.c_int(0), C.c_int(0)
    L.nxz_dhtgen(ll, num_ll, d, num_d, buf, C.byref(nb), C.byref(vb))
    return buf.raw[:nb.value], nb.value * 8 - (8 - vb.value if vb.value else 0)


def test_product_dhtgen_matches_reference_vectors():
    L = _lib()
    vecs = json.load(open(os.path.join(ROOT, "tests", "golden", "dhtgen_vectors.json")))
    assert len(vecs) >= 20
    for v in vecs:
        ll = (C.c_uint32 * 286)(*v["ll"])
        d = (C.c_uint32 * 30)(*v["d"])
        num_ll = max(257, max(i + 1 for i, c in enumerate(v["ll"]) if c))
        num_d = max([i + 1 for i, c in enumerate(v["d"]) if c] or [0])
        if v["flag"] == "-f":
            L.nxz_fill_zero_lzcounts(ll, d, 1)
            num_ll, num_d = 286, 30
        elif v["flag"] == "-g":
            for i in range(257, 286):
                ll[i] = ll[i] or 1
            for i in range(30):
                d[i] = d[i] or 1
            num_ll, num_d = 286, 30
        got, dhtlen = _gen(L, ll, d, num_ll, num_d)
        assert dhtlen == v["dhtlen"], v["name"]
        assert got.hex() == v["dht"][:len(got) * 2], v["name"]


def test_batched_dhtgen_equals_single_calls():
    L = _lib()
    rng = np.random.default_rng(7)
    n = 37
    counts = np.zeros((n, 316), np.uint32)
    for i in range(n):
        k = rng.integers(1, 286)
        counts[i, rng.choice(286, k, replace=False)] = rng.integers(1, 5000, k)
        kd = rng.integers(0, 30)
        if kd:
            counts[i, 286 + rng.choice(30, kd, replace=False)] = rng.integers(1, 3000, kd)
    dt = np.dtype([("dhtlen", "<u4"), ("dht", "u1", (292,))])
    tables = np.zeros(n, dt)
    for threads in (1, 3):
        tables[:] = 0
        assert L.nxz_dhtgen_batch(counts.ctypes.data, n, tables.ctypes.data, threads) == 0
        for i in range(n):
            ll = (C.c_uint32 * 286)(*counts[i, :286]); d = (C.c_uint32 * 30)(*counts[i, 286:])
            ll[256] = 1
            L.nxz_fill_zero_lzcounts(ll, d, 1)
            exp, bits = _gen(L, ll, d, 286, 30)
            assert tables["dhtlen"][i] == bits, i
            assert tables["dht"][i, :len(exp)].tobytes() == exp, i


def test_canned_tables_are_the_references_and_serve_as_in_dht_lookup():
    """a11: the 35 canned tables (lib/nx_dht_builtin.c:104-840) are carried as data; the first one is a
    stream's default table (lib/nx_dht.c:578-583); a job whose two most frequent literal/length symbols are
    a canned table's keys gets that table (dht_search_builtin, lib/nx_dht.c:401-432)."""
    import ctypes as C
    L = _lib()
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "builtin_dht.json")))
    L.nxz_dht_builtin_get.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int * 3)]
    assert L.nxz_dht_builtin_count() == len(gold) == 35
    for i, g in enumerate(gold):
        buf = C.create_string_buffer(320)
        n = C.c_uint32()
        key = (C.c_int * 3)()
        assert L.nxz_dht_builtin_get(i, buf, C.byref(n), C.byref(key)) == 0
        assert n.value == int(g["dhtlen"]) and list(key) == [int(x) for x in g["litlen"]]
        assert buf.raw[:(n.value + 7) // 8].hex() == g["dht"][:2 * ((n.value + 7) // 8)]
    L.nxz_dht_begin.restype = C.c_void_p
    L.nxz_dht_lookup.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_char_p, C.POINTER(C.c_uint32)]
    L.nxz_dht_end.argtypes = [C.c_void_p]
    st = L.nxz_dht_begin()
    out = C.create_string_buffer(320)
    n = C.c_uint32()
    L.nxz_dht_lookup(st, None, 0, out, C.byref(n))                  # first job: entry 0
    assert n.value == int(gold[0]["dhtlen"]) and out.raw[:(n.value + 7) // 8].hex() == gold[0]["dht"][:2 * ((n.value + 7) // 8)]
    # counts whose top two literals are the keys of canned table 2 (0 and 1): served by it, not generated
    k0, k1 = [int(x) for x in gold[2]["litlen"][:2]]
    counts = np.ones(316, np.uint32)
    counts[k0], counts[k1] = 5000, 4000
    L.nxz_dht_lookup(st, counts.ctypes.data, 1 << 20, out, C.byref(n))
    assert n.value == int(gold[2]["dhtlen"]) and out.raw[:(n.value + 7) // 8].hex() == gold[2]["dht"][:2 * ((n.value + 7) // 8)]
    # other top symbols: a generated table (no canned table has these keys)
    counts = np.ones(316, np.uint32)
    counts[200], counts[201] = 5000, 4000
    L.nxz_dht_lookup(st, counts.ctypes.data, 1 << 20, out, C.byref(n))
    assert all(out.raw[:(n.value + 7) // 8].hex() != g["dht"][:2 * ((n.value + 7) // 8)] for g in gold)
    L.nxz_dht_end(st)
