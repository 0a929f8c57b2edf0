"""Soak run (not collected by pytest): seeded random zlib streams (all levels and strategies, flushed
pieces, sizes 0..300 KB, with history and cut-off tails) through the inflate kernels of the HIP
engine (a stream per workgroup, a stream per lane with and without the fixed-code-only kernel in front, a stream per wave with the window in LDS and in the target, every stream cut into pieces on the device), every result compared with the oracle (output, stop state, checksums).
python tests/soak_inflate_gpu.py [seeds]"""
import importlib, os, random, sys, zlib
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as O
from datagen import make_block
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
kinds = ["zeros", "random", "text33", "alice", "lz", "periodic", "binary", "sparse"]
strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bad = total = 0
for seed in range(1, nseeds + 1):
    rnd = random.Random(seed * 104729)
    cases = []
    for i in range(700):
        big = rnd.random() < 0.05
        n = rnd.randrange(0, 300000) if big else rnd.choice([0, 1, 2, 255, 256, 257, 16383, 16384, 16385, 65535, 65536]) if rnd.random() < 0.3 else rnd.randrange(0, 70000)
        d = b"".join(make_block(rnd.choice(kinds), min(65536, n - o), seed=seed * 100000 + i * 8 + k) for k, o in enumerate(range(0, n, 65536)))
        co = zlib.compressobj(rnd.randrange(0, 10), zlib.DEFLATED, -15, rnd.randrange(1, 10), rnd.choice(strategies))
        c, pos = b"", 0
        while pos < n and rnd.random() < 0.5:
            k = rnd.randrange(1, n - pos + 1)
            c += co.compress(d[pos:pos + k]) + co.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_PARTIAL_FLUSH]))
            pos += k
        c += co.compress(d[pos:]) + co.flush()
        how = rnd.random()
        if how < 0.15 and len(c) > 1: c = c[:rnd.randrange(1, len(c))]            # cut off
        elif how < 0.25: c += rnd.randbytes(rnd.randrange(1, 40))                    # trailing bytes
        elif how < 0.32 and c:                                                       # damaged
            b = bytearray(c); b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8); c = bytes(b)
        cap = rnd.choice([len(d), len(d), len(d) + 100, max(len(d) - rnd.randrange(1, 50), 0), 16]) if rnd.random() < 0.3 else len(d) + 16
        cases.append((c, cap))
    cstride = (max(len(c) for c, _ in cases) + 31) & ~15
    ostride = (max(cap for _, cap in cases) + 64 + 15) & ~15
    host = np.zeros((len(cases), cstride), np.uint8)
    for i, (c, _) in enumerate(cases): host[i, :len(c)] = np.frombuffer(c, np.uint8)
    src = torch.from_numpy(host).to(eng.dev)
    exp = [O.inflate(c, cap) for c, cap in cases]
    for kernel, env in (("a stream per workgroup (what it does not take: handed back to a wavefront each)", {"NXZ_INFLATE_WG": "1"}),
                        ("... pieces of 250 bits, a wavefront a piece from the first round on", {"NXZ_INFLATE_WG": "1", "NXZ_WG_PMIN": "250", "NXZ_WG_COOP": "1024"}),
                        ("lanes", {"NXZ_INFLATE_LANES_MIN": "1", "NXZ_LANES_FIXED": "0"}),
                        ("lanes, the fixed-code-only kernel first", {"NXZ_INFLATE_LANES_MIN": "1", "NXZ_LANES_FIXED": "2"}),
                        ("waves, window in LDS", {"NXZ_INFLATE_LANES_MIN": "1000000000", "NXZ_INFLATE_LDS_MAX": "1000000000"}),
                        ("waves, target as window", {"NXZ_INFLATE_LANES_MIN": "1000000000", "NXZ_INFLATE_LDS_MAX": "0"}),
                        ("every stream cut into pieces on the device", {"NXZ_INFLATE_LANES_MIN": "1000000000", "NXZ_INFLATE_CUT": "1"}),
                        ("... three pieces a round, two rounds", {"NXZ_INFLATE_LANES_MIN": "1000000000", "NXZ_INFLATE_CUT": "1", "NXZ_INFLATE_CUT_PIECES": "3", "NXZ_INFLATE_CUT_ROUNDS": "2"})):
        os.environ["NXZ_INFLATE_CUT"] = "0"
        for k in ("NXZ_INFLATE_WG", "NXZ_WG_PMIN", "NXZ_WG_COOP", "NXZ_INFLATE_LANES_MIN", "NXZ_LANES_FIXED", "NXZ_INFLATE_LDS_MAX"): os.environ.pop(k, None)
        os.environ.pop("NXZ_INFLATE_CUT_PIECES", None); os.environ.pop("NXZ_INFLATE_CUT_ROUNDS", None)
        os.environ.update(env)
        dst = torch.full((len(cases), ostride), 0xAA, dtype=torch.uint8, device=eng.dev)
        jobs = eng.jobs_strided(src, cstride, np.array([len(c) for c, _ in cases], np.uint32), dst, ostride, np.array([cap for _, cap in cases], np.uint32),
                                in_crc=seed, in_adler=seed + 1)
        r = eng.results_to_host(eng.decompress(jobs, len(cases)))
        out = dst.cpu().numpy()
        for i, ((c, cap), (e, st)) in enumerate(zip(cases, exp)):
            total += 1
            ok = (out[i, cap:] == 0xAA).all()
            if st.err:
                ok = ok and r["cc"][i] == st.err
            else:
                subc = st.out_subc
                if st.final_eob and subc > 0xfff8: subc -= 8 * ((subc - 0xfff8 + 7) // 8)
                ok = ok and r["cc"][i] in (0, 3) and r["tpbc"][i] == st.tpbc and out[i, :st.tpbc].tobytes() == e
                ok = ok and (r["sfbt"][i] & 0xf) == st.out_sfbt and r["subc"][i] == subc and bool(r["sfbt"][i] & 0x100) == bool(st.final_eob)
                ok = ok and r["crc"][i] == zlib.crc32(e, seed) and r["adler"][i] == zlib.adler32(e, seed + 1)
            if not ok:
                bad += 1
                if bad < 10: print("MISMATCH seed %d case %d kernel %s: cc %d (oracle err %d) tpbc %d (oracle %d)" % (seed, i, kernel, r["cc"][i], st.err, r["tpbc"][i], st.tpbc))
    print("seed %d done: %d results checked so far, %d mismatches" % (seed, total, bad), flush=True)
print("SOAK OK" if bad == 0 else "SOAK FAILED: %d" % bad)
sys.exit(1 if bad else 0)
