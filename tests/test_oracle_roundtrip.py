"""CPU tests: oracle deflate/inflate round trips through system zlib (RFC1951 validity)."""
import os
import random
import zlib

import pytest

import oracle_lib as O
from datagen import make_block, ALICE_LIKE


def zinflate_raw(b, n):
    d = zlib.decompressobj(-15)
    out = d.decompress(b, n + 16)
    assert d.eof, "zlib did not reach final EOB"
    return out


CASES = [("zeros", 65536), ("text33", 65536), ("lz", 65536), ("random", 65536), ("lz", 1), ("lz", 2),
         ("lz", 3), ("lz", 4), ("lz", 5), ("text33", 63), ("text33", 64), ("text33", 65), ("lz", 16383),
         ("lz", 16384), ("lz", 16385), ("zeros", 300), ("lz", 200000), ("alice", 152089)]


@pytest.mark.parametrize("kind,n", CASES)
def test_fixed_roundtrip(kind, n):
    data = make_block(kind, n, seed=n)
    comp, bits = O.deflate_fixed(data)
    assert comp[0] & 7 == 0b011           # BFINAL=1, BTYPE=01, block starts at bit 0
    assert zinflate_raw(comp, n) == data


@pytest.mark.parametrize("kind,n", CASES)
def test_dynamic_roundtrip_own_table(kind, n):
    data = make_block(kind, n, seed=n + 1)
    tok, nt = O.lz77(data)
    ll, d = O.counts(tok, nt)
    dht, dhtlen = O.dhtgen(ll, d)
    comp, bits = O.deflate_dynamic(data, dht, dhtlen)
    assert comp is not None and comp[0] & 7 == 0b101
    assert zinflate_raw(comp, n) == data


def test_dynamic_missing_code_is_reported():
    # table made from zeros only cannot code text -> engine CC 66 (UM 2.5.9.5)
    z = make_block("zeros", 4096, 0)
    tok, nt = O.lz77(z)
    ll, d = O.counts(tok, nt)
    dht, dhtlen = O.dhtgen(ll, d)
    comp, bits = O.deflate_dynamic(make_block("text33", 4096, 1), dht, dhtlen)
    assert comp is None


def test_history_is_used_and_not_emitted():
    hist = make_block("text33", 32768, 5)
    data = hist[:20000]                     # source repeats the history
    comp, bits = O.deflate_fixed(hist + data, hist=len(hist))
    d = zlib.decompressobj(-15, zdict=hist)
    assert d.decompress(comp) == data
    assert len(comp) < 6000          # single-probe table keeps ~1/4 of a 32 KiB random window


@pytest.mark.parametrize("level", [1, 6, 9])
def test_inflate_of_zlib_streams_with_random_suspends(level):
    rnd = random.Random(level)
    data = make_block("lz", 150000, 3) + make_block("random", 70000, 4) + make_block("zeros", 5000, 0)
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data[:100000]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(data[100000:]) + co.flush()
    comp += b"TRAILER!"                       # 8 trailer bytes after the final EOB
    out = b""
    pos = 0
    kw = {}
    while True:
        take = rnd.choice([1, 2, 7, 300, 5000, 70000])
        chunk = comp[pos:pos + take]
        hist = out[-32768:]
        got, st = O.inflate(chunk, 1 << 20, hist=hist, **kw)
        assert st.err == 0
        out += got
        if st.final_eob:
            # everything after the EOB byte is reported as unprocessed bits
            assert st.out_sfbt == 0
            consumed = pos + len(chunk) - st.out_subc // 8
            assert comp[consumed:] == b"TRAILER!"[: len(comp) - consumed] or consumed <= len(comp) - 8
            break
        back = (st.out_subc + 7) // 8
        pos = pos + len(chunk) - back
        kw = dict(subc=st.out_subc % 8, sfbt=st.out_sfbt, rembytecnt=st.out_rembytecnt)
        if (st.out_sfbt & 0xe) == 0xc:
            kw.update(dht=bytes(st.out_dht), dhtlen=st.out_dhtlen)
        assert pos < len(comp)
    assert out == data


def test_inflate_target_too_small_is_cc13():
    data = make_block("lz", 10000, 9)
    comp = zlib.compress(data, 6)[2:-4]
    got, st = O.inflate(comp, 5000)
    assert st.err == 13


def test_ratio_gate_on_alice_like_text():
    # >= 0.95 x zlib -1 on identical 64 KiB chunks
    data = ALICE_LIKE(65536)
    comp, _ = O.deflate_fixed(data)
    co = zlib.compressobj(1, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
    z = co.compress(data) + co.flush()
    assert len(z) / len(comp) >= 0.95
