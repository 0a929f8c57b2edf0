"""Throughput behind the reference API (host buffers, PCIe and host copies included):
  one-shot nx_compress2 / nx_uncompress of a large buffer, and T threads x 64 KiB nx_compress2 calls.
usage: python tools/bench_api.py [MiB] [threads]"""
import ctypes as C
import json
import os
import sys
import threading
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import zstream as Z  # noqa: E402
import corpus  # noqa: E402


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    L = Z.load("gpu")
    base = b"".join(b for _, _, b in corpus.load()[1])
    data = (base * ((mib << 20) // len(base) + 1))[:mib << 20]
    out = {"MiB": mib}
    cap = C.c_ulong(L.nx_compressBound(len(data)))
    dst = C.create_string_buffer(cap.value)
    best = 1e9
    for it in range(4):
        cap.value = len(dst)
        t = time.perf_counter()
        rc = L.nx_compress2(dst, C.byref(cap), data, len(data), 1)
        dt = time.perf_counter() - t
        assert rc == 0, rc
        if it:
            best = min(best, dt)
    comp = dst.raw[:cap.value]
    assert zlib.decompress(comp) == data
    out["compress2_one_shot"] = {"GiB_s": round(len(data) / best / 2**30, 2), "ms": round(best * 1e3, 2), "ratio": round(len(data) / len(comp), 3)}
    t = time.perf_counter()
    z1 = zlib.compress(data, 1)
    out["zlib1_one_thread_GiB_s"] = round(len(data) / (time.perf_counter() - t) / 2**30, 3)
    out["zlib1_ratio"] = round(len(data) / len(z1), 3)
    # nx_uncompress of a zlib -6 stream (parallel path)
    z6 = zlib.compress(data, 6)
    back = C.create_string_buffer(len(data))
    best = 1e9
    for it in range(4):
        n = C.c_ulong(len(data))
        t = time.perf_counter()
        rc = L.nx_uncompress(back, C.byref(n), z6, len(z6))
        dt = time.perf_counter() - t
        assert rc == 0 and n.value == len(data), (rc, n.value)
        if it:
            best = min(best, dt)
    assert back.raw == data
    out["uncompress_one_shot_zlib6"] = {"GiB_s": round(len(data) / best / 2**30, 2), "ms": round(best * 1e3, 2)}
    t = time.perf_counter()
    zlib.decompress(z6)
    out["zlib_inflate_one_thread_GiB_s"] = round(len(data) / (time.perf_counter() - t) / 2**30, 3)

    # T threads, each compressing 64 KiB buffers one call after the other (compdecomp_th style)
    per = 512
    blocks = [data[i * 65536:(i + 1) * 65536] for i in range(min(per, len(data) // 65536))]

    def worker(res, k):
        cap = C.c_ulong()
        dst = C.create_string_buffer(L.nx_compressBound(65536))
        tot = 0
        for b in blocks:
            cap.value = len(dst)
            assert L.nx_compress2(dst, C.byref(cap), b, len(b), 1) == 0
            tot += cap.value
        res[k] = tot

    for T in (1, nthreads):
        res = [0] * T
        th = [threading.Thread(target=worker, args=(res, k)) for k in range(T)]
        t = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t
        out["threads_%d_x_64KiB" % T] = {"GiB_s": round(T * len(blocks) * 65536 / dt / 2**30, 3), "us_per_call": round(dt / len(blocks) * 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
