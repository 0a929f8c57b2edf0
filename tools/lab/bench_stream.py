#!/usr/bin/env python3
"""One long zlib-made deflate stream through nxz_inflate_stream (block-boundary speculation):
corpus data (tests/corpus.py) repeated to `mib` MiB, compressed by zlib at `level` as ONE raw stream,
inflated on the device, compared bit for bit, CRC-32 against zlib's.  usage: bench_stream.py [mib] [level]"""
import importlib, json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import corpus
pkg = importlib.import_module("power-gzip_amd")


def make_stream(mib, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    name, blocks, _ = corpus.load(65536)
    only = [c for c in os.environ.get("CLASSES", "").split(",") if c]            # CLASSES=text,xml: those classes only; CLASSES=-image,-packed: all but those
    drop = [c[1:] for c in only if c.startswith("-")]
    keep = [c for c in only if not c.startswith("-")]
    raw = b"".join(b for c, _, b in blocks if c not in drop and (not keep or c in keep))
    data = (raw * (mib * (1 << 20) // len(raw) + 1))[:mib << 20]
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    t0 = time.perf_counter()
    comp = c.compress(data) + c.flush()
    return data, comp, time.perf_counter() - t0, name


def run(mib=64, level=6, reps=3):
    data, comp, tz, name = make_stream(mib, level)
    eng = pkg.Engine(0)
    src = torch.from_numpy(np.frombuffer(comp, np.uint8).copy()).to(eng.dev)
    dst = torch.zeros(len(data) + 4096, dtype=torch.uint8, device=eng.dev)
    rc, info = eng.inflate_stream(src, len(comp), dst)
    if rc != 0:
        print(json.dumps({"rc": rc, **info}))
        return 1
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        rc, info = eng.inflate_stream(src, len(comp), dst)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ok = rc == 0 and info["out_len"] == len(data) and info["crc"] == zlib.crc32(data) and info["adler"] == zlib.adler32(data) \
        and dst[:len(data)].cpu().numpy().tobytes() == data and info["end_bit"] == len(comp) * 8 - (0 if info["end_bit"] % 8 == 0 else 8 - info["end_bit"] % 8) - 0
    t0 = time.perf_counter(); zlib.decompress(comp, -15); tcpu = time.perf_counter() - t0
    print(json.dumps({"corpus": name, "MiB": mib, "zlib_level": level, "compressed_MiB": round(len(comp) / 2 ** 20, 2),
                      "GiB_s_out": round(len(data) / best / 2 ** 30, 3), "ms": round(best * 1e3, 2), "bit_exact": bool(ok),
                      "pieces": info["pieces"], "rounds": info["rounds"], "zlib_inflate_1thread_GiB_s": round(len(data) / tcpu / 2 ** 30, 3)}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(run(int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 6))
