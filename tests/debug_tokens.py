"""Debug aid: decode fixed-Huffman deflate blocks into LZ77 tokens and diff GPU vs oracle."""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # repo root (this file lives in tests/: it uses the oracle)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

LEN_BASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEN_EXTRA = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DIST_BASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DIST_EXTRA = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]


def tokens_fixed(comp):
    bits = np.unpackbits(np.frombuffer(comp, np.uint8), bitorder="little")
    pos = 3
    def get(n):
        nonlocal pos
        v = 0
        for i in range(n):
            v |= int(bits[pos + i]) << i
        pos += n
        return v
    def getcode(n):   # huffman codes are MSB first
        nonlocal pos
        v = 0
        for i in range(n):
            v = (v << 1) | int(bits[pos + i])
        pos += n
        return v
    toks = []
    p = 0
    while True:
        c = getcode(7)
        if c <= 0b0010111:
            sym = 256 + c
        else:
            c = (c << 1) | get(1)
            if 0b00110000 <= c <= 0b10111111:
                sym = c - 0b00110000
            elif 0b11000000 <= c <= 0b11000111:
                sym = 280 + c - 0b11000000
            else:
                c = (c << 1) | get(1)
                sym = 144 + c - 0b110010000
        if sym < 256:
            toks.append((p, "L", sym)); p += 1
        elif sym == 256:
            break
        else:
            ls = sym - 257
            ln = LEN_BASE[ls] + get(LEN_EXTRA[ls])
            ds = getcode(5)
            d = DIST_BASE[ds] + get(DIST_EXTRA[ds])
            toks.append((p, "M", ln, d)); p += ln
    return toks


def main():
    import torch
    import oracle_lib as O
    from datagen import make_block
    pkg = importlib.import_module("power-gzip_amd")
    kind, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    b = make_block(kind, n, seed)
    eng = pkg.Engine(0)
    dbg = torch.zeros(4 * 65536, dtype=torch.int32, device=eng.dev)
    import ctypes as C
    eng.L.nxz_debug_set.argtypes = [C.c_void_p]
    assert eng.L.nxz_debug_set(dbg.data_ptr()) == 0
    src = torch.from_numpy(np.frombuffer(b + bytes(16), np.uint8).copy()).to(eng.dev)
    dst = torch.zeros(2 * n + 4096, dtype=torch.uint8, device=eng.dev)
    jobs = eng.jobs_strided(src, 0, np.array([n], np.uint32), dst, 0, (2 * n + 4096) & ~3)
    r = eng.results_to_host(eng.compress(pkg.FC_COMPRESS_FHT, jobs, 1)[0])
    got = dst[:r["tpbc"][0]].cpu().numpy().tobytes()
    L = O.lib()
    om = (C.c_uint16 * 65536)(); od = (C.c_uint16 * 65536)(); ox = (C.c_uint32 * 4096)()
    C.c_void_p.in_dll(L, "nxo_dbg_mlen").value = C.addressof(om)
    C.c_void_p.in_dll(L, "nxo_dbg_mdist").value = C.addressof(od)
    C.c_void_p.in_dll(L, "nxo_dbg_x").value = C.addressof(ox)
    exp, bits = O.deflate_fixed(b)
    d = dbg.cpu().numpy().view(np.uint32)
    nbad = 0
    for p in range(n):
        tile, i = divmod(p, 16384)
        v = int(d[tile * 65536 + i]); gc, gm = v & 0xffff, (v >> 16) & 0xff
        ol, odist = om[p], od[p]
        exp_m = 0 if ol < 4 else ol - 3
        exp_c = odist if ol >= 4 else None
        ok = gm == exp_m and (exp_c is None or gc == exp_c)
        if not ok:
            nbad += 1
            if nbad < 8:
                print("pos", p, "gpu mlen", gm, "cand", gc, "| oracle len", ol, "dist-1", odist)
    print("position mismatches:", nbad)
    nx = 0
    for sgm in range((n + 15) // 16):
        tile, i = divmod(sgm, 1024)
        v = int(d[tile * 65536 + 16384 + i])
        if (v & 0xffff) + tile * 16384 != ox[sgm]:
            nx += 1
            if nx < 8:
                print("seg", sgm, "gpu X", (v & 0xffff) + tile * 16384, "entry", (v >> 16) & 0x7fff, "entered", v >> 31, "| oracle X", ox[sgm])
    print("segment exit mismatches:", nx)
    print("gpu", len(got), "oracle", len(exp), "cc", r["cc"][0])
    tg, te = tokens_fixed(got), tokens_fixed(exp)
    print("tokens gpu", len(tg), "oracle", len(te))
    for i, (a, c) in enumerate(zip(tg, te)):
        if a != c:
            print("first diff at token", i)
            for k in range(max(0, i - 3), min(len(tg), len(te), i + 6)):
                print("  gpu", tg[k], " | oracle", te[k])
            break
    else:
        print("token streams equal (prefix)")


if __name__ == "__main__":
    main()
