#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference tree.

Run ONLY in the build container (needs /root/reference and `make -C oracle ref`);
the outputs (small JSON data files) are committed, this script is the record of
how they were made.  Nothing here is imported by the product.

Fixtures (all paths relative to /root/reference):
  dhtgen_vectors.json   counts -> DHT bytes produced by oracle/_ref/dhtgen_ref, i.e. the
                        reference's own lib/nx_dhtgen.c compiled in place with -D_DHTGEN_TEST
  builtin_dht.json      the 35 canned tables of lib/nx_dht_builtin.c:104-840 (data: dhtlen,
                        bytes, top-literal keys) -- KAT set for the DHT header parser
  crc32_kat.json        (crc_in, buffer, expected) of test/test_crc32.c:38-180
  adler32_kat.json      same for test/test_adler32.c:38-179
  zlib_stream_buf_error.json  the scp zlib stream of test/test_buf_error.c:107-183,217-229
                        with the inflated lengths the test asserts (603, 117)
  alice29_zlib.json     sizes/sha256 of samples/alice29.txt and of zlib level-1 output per
                        64 KiB chunk (the 0.95x ratio gates of BASELINE.md)
"""
import hashlib
import json
import os
import random
import re
import subprocess
import tempfile
import zlib

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DHTGEN = os.path.join(ROOT, "oracle", "_ref", "dhtgen_ref")


def strip_comments(s):
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def run_dhtgen(ll, d, flag):
    """ll: 286 counts, d: 30 counts, flag: '', '-f' or '-g'. Returns (hex, nbits)."""
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "lz.txt")
        with open(p, "w") as f:
            for i, c in enumerate(ll):
                if c:
                    f.write("%d : %d\n" % (i, c))
            for i, c in enumerate(d):
                if c:
                    f.write("%d : %d\n" % (i, c))
        cmd = [DHTGEN] + ([flag] if flag else []) + [p, "unused"]
        subprocess.run(cmd, cwd=td, check=True, stdout=subprocess.DEVNULL)
        raw = open(os.path.join(td, "dht.bin"), "rb").read()
    nbits = (raw[14] << 8) | raw[15]
    return raw[16:].hex(), nbits


def dhtgen_vectors():
    rnd = random.Random(20261002)
    vecs = []

    def add(name, ll, d, flag):
        # the test main requires ascending symbols and restarts at the first distance symbol;
        # a distance list whose first symbol is not smaller than the last LL symbol would be
        # mis-parsed, LL always ends >= 256 (EOB forced) so any d index < 30 is fine.
        ll = list(ll)
        if ll[256] == 0:
            ll[256] = 1
        hexs, nbits = run_dhtgen(ll, d, flag)
        vecs.append({"name": name, "flag": flag, "ll": ll, "d": list(d), "dht": hexs, "dhtlen": nbits})

    # text-like counts from alice29 literals + a few lengths/distances
    data = open(os.path.join(REF, "samples", "alice29.txt"), "rb").read()
    ll = [0] * 286
    for b in data[:65536]:
        ll[b] += 1
    for s in range(257, 286):
        ll[s] = max(0, 4000 // (s - 250))
    d = [max(1, 3000 // (i + 1)) for i in range(30)]
    add("alice_like", ll, d, "")
    add("alice_like_f", ll, d, "-f")
    add("alice_like_g", ll, d, "-g")
    # literals only, no distances at all (single-distance special case nx_dhtgen.c:985-1014)
    ll2 = [0] * 286
    for b in data[:4096]:
        ll2[b] += 1
    add("literals_only", ll2, [0] * 30, "")
    add("one_distance_sym0", ll2[:257] + [5] + [0] * 28, [7] + [0] * 29, "")
    # fibonacci-like skew forces depth > 15 -> length_limit retries (nx_dhtgen.c:576-595)
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    ll3 = [0] * 286
    for i in range(40):
        ll3[i * 7] = fib[i]
    add("fibonacci_skew", ll3, [fib[i % 25] for i in range(30)], "")
    add("fibonacci_skew_f", ll3, [fib[i % 25] for i in range(30)], "-f")
    # saturated 24-bit counters (UM 5.1.1) -> divisor > 1 on the first pass
    add("saturated", [0xFFFFFF if i % 3 == 0 else i for i in range(286)], [0xFFFFFF] * 30, "")
    # flat and random tables
    add("flat_all_ones", [1] * 286, [1] * 30, "")
    for k in range(12):
        ll4 = [0] * 286
        d4 = [0] * 30
        nsym = rnd.choice([3, 10, 60, 200, 286])
        for s in rnd.sample(range(286), nsym):
            ll4[s] = int(rnd.paretovariate(0.7))
        for s in rnd.sample(range(30), rnd.choice([2, 5, 30])):
            d4[s] = int(rnd.paretovariate(0.9))
        add("random_%d" % k, ll4, d4, rnd.choice(["", "-f", "-g"]))
    # long zero runs in the length vector exercise the 17/18 repeat states incl. the 138 limit
    ll5 = [0] * 286
    ll5[0] = 9; ll5[1] = 3; ll5[141] = 5; ll5[256] = 1; ll5[285] = 2
    add("zero_runs", ll5, [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4], "")
    # equal lengths exercise the 16 repeat state incl. the count==6 limit
    add("equal_lengths", [16] * 256 + [1] + [0] * 29, [4] * 16 + [0] * 14, "")
    return vecs


def builtin_tables():
    src = strip_comments(open(os.path.join(REF, "lib", "nx_dht_builtin.c")).read())
    body = src[src.index("builtin1[DHT_NUM_BUILTIN]"):]
    body = body[body.index("{") + 1:]
    out = []
    depth = 0
    cur = ""
    for ch in body:
        if ch == "{":
            depth += 1
        if depth >= 1:
            cur += ch
        if ch == "}":
            depth -= 1
            if depth == 0 and cur:
                # one dht_entry_t initialiser: scalars, {bytes}, {litlen}, {dist}
                groups = re.findall(r"\{([^{}]*)\}", cur[1:-1])
                scalars = re.sub(r"\{[^{}]*\}", "", cur[1:-1])
                nums = [int(x, 0) for x in re.findall(r"-?(?:0x[0-9a-fA-F]+|\d+)", scalars)]
                dhtlen = nums[4]
                by = [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", groups[0])]
                lit = [int(x) for x in re.findall(r"-?\d+", groups[1])]
                nbytes = (dhtlen + 7) // 8
                out.append({"dhtlen": dhtlen, "dht": bytes(by[:nbytes]).hex(), "litlen": lit})
                cur = ""
            if depth < 0:
                break
    assert len(out) == 35, len(out)
    return out


def kat(fname):
    src = strip_comments(open(os.path.join(REF, "test", fname)).read())
    rows = re.findall(r"\{\s*__LINE__\s*,\s*(0x[0-9a-fA-F]+|\d+)\s*,\s*(?:\(Byte \*\)\s*\"((?:[^\"\\\\]|\\\\.)*)\"|(0x0|NULL|0))\s*,"
                      r"\s*(\d+)\s*,\s*(0x[0-9a-fA-F]+|\d+)\s*\}", src)
    out = []
    for init, s, null, n, exp in rows:
        if null:
            out.append({"init": int(init, 0), "buf": None, "len": int(n), "expect": int(exp, 0)})
            continue
        raw = s.encode("latin1").decode("unicode_escape").encode("latin1") + b"\x00"   # C string
        assert len(raw) >= int(n), (s, n)
        out.append({"init": int(init, 0), "buf": raw[:int(n)].hex(), "len": int(n), "expect": int(exp, 0)})
    return out


def buf_error_stream():
    src = strip_comments(open(os.path.join(REF, "test", "test_buf_error.c")).read())
    arrays = re.findall(r"(?:unsigned\s+char|char|uint8_t|Bytef|Byte)\s+(\w+)\s*\[[^\]]*\]\s*=\s*\{([^}]*)\}", src)
    out = {}
    for name, body in arrays:
        by = bytes(int(x, 0) & 0xff for x in re.findall(r"0x[0-9a-fA-F]+|\d+", body))
        out[name] = by.hex()
    return out


def alice():
    data = open(os.path.join(REF, "samples", "alice29.txt"), "rb").read()
    chunks = [data[i:i + 65536] for i in range(0, len(data), 65536)]
    res = {"size": len(data), "sha256": hashlib.sha256(data).hexdigest(), "chunks": []}
    for c in chunks:
        row = {"len": len(c), "sha256": hashlib.sha256(c).hexdigest()}
        for name, strat in (("default", zlib.Z_DEFAULT_STRATEGY), ("fixed", zlib.Z_FIXED)):
            co = zlib.compressobj(1, zlib.DEFLATED, -15, 8, strat)
            row["zlib1_" + name] = len(co.compress(c) + co.flush())
        res["chunks"].append(row)
    return res


def main():
    def dump(name, obj):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(obj, f, separators=(",", ":"))
            f.write("\n")
        print(name, os.path.getsize(os.path.join(HERE, name)))

    dump("dhtgen_vectors.json", dhtgen_vectors())
    dump("builtin_dht.json", builtin_tables())
    dump("crc32_kat.json", kat("test_crc32.c"))
    dump("adler32_kat.json", kat("test_adler32.c"))
    dump("zlib_stream_buf_error.json", buf_error_stream())
    dump("alice29_zlib.json", alice())


def abi_symbols():
    """names, version nodes and kinds of the ELF symbols of the reference's library (test/libnxz.abi)"""
    import re
    abi = open(os.path.join(REF, "test", "libnxz.abi")).read()
    syms = []
    for m in re.finditer(r"<elf-symbol name='([^']*)'([^>]*)>", abi):
        name, rest = m.group(1), m.group(2)
        v = re.search(r"version='([^']*)'", rest)
        t = re.search(r"type='([^']*)'", rest).group(1)
        syms.append({"name": name, "version": v.group(1) if v else "", "type": "object" if t.startswith("object") else "func"})
    json.dump({"source": "/root/reference/test/libnxz.abi (elf-symbol entries: name, version node, kind)", "soname": "libnxz.so.0",
               "symbols": syms}, open(os.path.join(HERE, "libnxz_abi_symbols.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
    abi_symbols()
