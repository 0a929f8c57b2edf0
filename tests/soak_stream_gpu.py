"""Soak run (not collected by pytest): seeded random long deflate streams -- pieces of the corpus, of
generated blocks, of random bytes and of nested deflate output, compressed piece after piece with
random levels, strategies, memLevels and flush points into ONE raw stream of 2..12 MiB -- through
nxz_inflate_stream; output and CRC-32 against zlib.  Streams the engine declines (-ENOTSUP) are
counted, everything else must be exact.  Then the same stream through nx_inflate in steps of random
sizes (parts of the stream that begin and end anywhere: nxz_inflate_stream_part), which must be exact
whatever the engine declines.   python tests/soak_stream_gpu.py [cases]"""
import importlib, os, random, sys, zlib
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import corpus
from datagen import make_block
import zstream as Z
ZL = Z.load("gpu")
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
_, blocks, _ = corpus.load(65536)
raw = b"".join(b for _, _, b in blocks)
kinds = ["zeros", "random", "text33", "alice", "lz", "periodic", "binary", "sparse"]
strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED]
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = declined = 0
first = int(os.environ.get('SOAK_FIRST', '0'))
for case in range(first, ncases):
    rnd = random.Random(7919 * (case + 1))
    target = rnd.randrange(2 << 20, 12 << 20)
    plain, comp = [], b""
    co = zlib.compressobj(rnd.randrange(1, 10), zlib.DEFLATED, -15, rnd.randrange(5, 10), rnd.choice(strategies))
    size = 0
    while size < target:
        how = rnd.random()
        n = rnd.randrange(1000, 900000)
        if how < 0.55:
            o = rnd.randrange(0, len(raw) - n)
            d = raw[o:o + n]
        elif how < 0.8:
            d = b"".join(make_block(rnd.choice(kinds), min(65536, n - o), seed=case * 1000 + o) for o in range(0, n, 65536))
        elif how < 0.9:
            d = rnd.randbytes(n)
        else:
            d = zlib.compress(raw[:n * 3], 6)[:n]                      # deflate output as data: block headers that are none
        plain.append(d); size += len(d)
        comp += co.compress(d)
        f = rnd.random()
        if f < 0.15: comp += co.flush(zlib.Z_SYNC_FLUSH)
        elif f < 0.25: comp += co.flush(zlib.Z_FULL_FLUSH)
        elif f < 0.3:                                                   # another compressor continues the stream
            comp += co.flush(zlib.Z_FULL_FLUSH)
            co = zlib.compressobj(rnd.randrange(1, 10), zlib.DEFLATED, -15, rnd.randrange(5, 10), rnd.choice(strategies))
    comp += co.flush()
    plain = b"".join(plain)
    src = torch.from_numpy(np.frombuffer(comp, np.uint8).copy()).to(eng.dev)
    dst = torch.zeros(len(plain) + 4096, dtype=torch.uint8, device=eng.dev)
    rc, info = eng.inflate_stream(src, len(comp), dst)
    torch.cuda.synchronize()
    if rc == -95:
        declined += 1
        continue
    ok = rc == 0 and info["out_len"] == len(plain) and info["crc"] == zlib.crc32(plain) and dst[:len(plain)].cpu().numpy().tobytes() == plain
    if not ok:
        bad += 1
        print("case %d: rc %d, %s, expected %d bytes" % (case, rc, info, len(plain)), flush=True)
    # ... and in steps through the zlib-style call
    step_in = rnd.choice([rnd.randrange(20000, 200000), rnd.randrange(200000, 3 << 20), len(comp)])
    step_out = rnd.choice([rnd.randrange(30000, 300000), rnd.randrange(300000, 4 << 20), len(plain) + 64])
    got, zrc, total_in, _ = Z.inflate_all(ZL, comp, wbits=-15, cap=len(plain) + 64, step_in=step_in, step_out=step_out)
    if zrc != Z.Z_STREAM_END or got != plain or total_in != len(comp):
        bad += 1
        print("case %d in steps of %d / %d: rc %d, %d bytes of %d, %d of %d consumed" % (case, step_in, step_out, zrc, len(got), len(plain), total_in, len(comp)), flush=True)
    # ... and damaged, inside a gzip wrapper (every third case): a data error or a trailer that does not match, never
    # wrong bytes reported as good, never a crash
    if case % 3 == 0:
        import struct
        gz = bytearray(b"\x1f\x8b\x08\0\0\0\0\0\0\x03" + comp + struct.pack("<II", zlib.crc32(plain), len(plain) & 0xffffffff))
        at = rnd.randrange(10, len(gz) - 8)
        for k in range(at, min(at + rnd.choice([1, 3, 40, 2000]), len(gz) - 8)):
            gz[k] ^= rnd.randrange(1, 256)
        got, zrc, _, _ = Z.inflate_all(ZL, bytes(gz), wbits=31, cap=len(plain) * 2 + (1 << 20), step_in=step_in, step_out=step_out)
        if not (zrc == Z.Z_DATA_ERROR or zrc == Z.Z_BUF_ERROR or (zrc == Z.Z_STREAM_END and got == plain)):
            bad += 1
            print("case %d damaged at %d: rc %d, %d bytes" % (case, at, zrc, len(got)), flush=True)
    if case % 10 == 9:
        print("%d cases done: %d declined, %d bad" % (case + 1, declined, bad), flush=True)
print("SOAK OK" if not bad else "SOAK FAILED", "(%d cases, %d declined)" % (ncases, declined))
sys.exit(1 if bad else 0)
