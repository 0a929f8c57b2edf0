"""CPU tests: the oracle (oracle/) against the reference's golden vectors (tests/golden/)."""
import ctypes as C
import json
import os
import zlib

import pytest

import oracle_lib as O


def load(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


def test_crc32_kat(golden_dir):
    # values of /root/reference/test/test_crc32.c:38-180
    rows = load(golden_dir, "crc32_kat.json")
    assert len(rows) >= 140
    for r in rows:
        buf = None if r["buf"] is None else bytes.fromhex(r["buf"])
        assert O.lib().nxo_crc32(r["init"], buf, r["len"] if buf is not None else 0) == r["expect"], r


def test_adler32_kat(golden_dir):
    # values of /root/reference/test/test_adler32.c:38-179
    rows = load(golden_dir, "adler32_kat.json")
    assert len(rows) >= 139
    for r in rows:
        buf = None if r["buf"] is None else bytes.fromhex(r["buf"])
        assert O.lib().nxo_adler32(r["init"], buf, r["len"] if buf is not None else 0) == r["expect"], r


def test_checksum_combine_matches_zlib():
    import random
    rnd = random.Random(7)
    for _ in range(50):
        a = rnd.randbytes(rnd.randrange(0, 5000))
        b = rnd.randbytes(rnd.randrange(0, 70000))
        assert O.lib().nxo_crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
        assert O.lib().nxo_adler32_combine(zlib.adler32(a), zlib.adler32(b), len(b)) == zlib.adler32(a + b)


def test_dhtgen_matches_reference(golden_dir):
    """bit-for-bit against the reference's nx_dhtgen.c (compiled in place -> dhtgen_ref)."""
    vecs = load(golden_dir, "dhtgen_vectors.json")
    assert len(vecs) >= 20
    for v in vecs:
        ll = (C.c_uint32 * 286)(*v["ll"])
        d = (C.c_uint32 * 30)(*v["d"])
        # what the reference's test main does before calling dhtgen (nx_dhtgen.c:1188-1235,1297-1309)
        num_ll = max(257, max(i + 1 for i, c in enumerate(v["ll"]) if c))
        num_d = max([i + 1 for i, c in enumerate(v["d"]) if c] or [0])
        if v["flag"] == "-f":
            O.lib().nxo_fill_zero_lzcounts(ll, d, 1)
            num_ll, num_d = 286, 30
        elif v["flag"] == "-g":
            for i in range(257, 286):
                if not ll[i]:
                    ll[i] = 1
            for i in range(30):
                if not d[i]:
                    d[i] = 1
            num_ll, num_d = 286, 30
        got, dhtlen = O.dhtgen(ll, d, num_ll, num_d)
        assert dhtlen == v["dhtlen"], v["name"]
        assert got.hex() == v["dht"][:len(got) * 2], v["name"]


def test_builtin_dht_parse(golden_dir):
    """the 35 canned tables of lib/nx_dht_builtin.c parse to exactly in_dhtlen bits, are complete."""
    tabs = load(golden_dir, "builtin_dht.json")
    assert len(tabs) == 35
    for t in tabs:
        c = O.Codes()
        used = O.lib().nxo_dht_parse(bytes.fromhex(t["dht"]), t["dhtlen"], C.byref(c))
        assert used == t["dhtlen"]
        assert all(c.ll_len[i] for i in range(286)) and all(c.d_len[i] for i in range(30))
        assert sum(2.0 ** -c.ll_len[i] for i in range(286)) == 1.0
        assert sum(2.0 ** -c.d_len[i] for i in range(30)) == 1.0


def test_reference_zlib_stream_inflates(golden_dir):
    """golden stream of test/test_buf_error.c:107-183 (611 B -> 603 B) and :217-229 (92 B -> 117 B)."""
    g = load(golden_dir, "zlib_stream_buf_error.json")
    compr, compr2 = bytes.fromhex(g["compr"]), bytes.fromhex(g["compr2"])
    ref = zlib.decompressobj()
    exp1 = ref.decompress(compr)
    exp2 = ref.decompress(compr2)
    assert len(exp1) == 603 and len(exp2) == 117
    # first call: raw deflate after the 2-byte zlib header; the stream does not end
    out1, st = O.inflate(compr[2:], 4096)
    assert out1 == exp1 and not st.final_eob and st.err == 0
    # resume with the engine's suspend state, history = everything produced so far
    tail = compr[2:][len(compr) - 2 - (st.out_subc + 7) // 8:]
    kw = dict(subc=st.out_subc % 8, sfbt=st.out_sfbt, rembytecnt=st.out_rembytecnt)
    if (st.out_sfbt & 0xe) == 0xc:
        kw.update(dht=bytes(st.out_dht), dhtlen=st.out_dhtlen)
    out2, st2 = O.inflate(tail + compr2, 4096, hist=out1, **kw)
    assert out2 == exp2 and st2.err == 0
