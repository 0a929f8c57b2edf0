"""The reference's own harness shape (samples/compdecomp_th.c:196-222,414-429 swept by samples/run-series.sh:19-41:
threads x buffer sizes, one zlib-style call per buffer) on this engine and on system zlib, same box, same file.
usage: api_sweep.py [file]   ->  a table: GiB/s of uncompressed bytes, compress / decompress, engine | zlib"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "alice29.txt")
sizes = [int(x) for x in os.environ.get("SWEEP_KIB", "4,16,64,256,1024,4096").split(",")]
threads = [int(x) for x in os.environ.get("SWEEP_THREADS", "1,16,64").split(",")]
print("%8s %4s | %10s %10s | %10s %10s | %9s %9s" % ("KiB", "T", "nx comp", "nx decomp", "zlib comp", "zlib dec", "comp x", "decomp x"))
for kib in sizes:
    for T in threads:
        per = max(32, min(1024, (192 << 10) // kib // max(1, T // 8), (8 << 20) // (kib * T)))   # (long enough that a thread's first calls, which make its buffer set, do not decide; 8 GiB of buffers at most)
        row = {}
        for name, exe in (("nx", "compdecomp_th"), ("zlib", "compdecomp_th_zlib")):
            p = subprocess.run([os.path.join(ROOT, "power-gzip_amd", exe), f, str(T), str(kib), str(per)], capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            row[name] = json.loads(line[-1]) if line else None
        if not row["nx"] or not row["zlib"]:
            print("%8d %4d | failed: %s" % (kib, T, (p.stderr or "")[-200:]))
            continue
        a, b = row["nx"], row["zlib"]
        print("%8d %4d | %10.3f %10.3f | %10.3f %10.3f | %9.2f %9.2f" % (kib, T, a["compress_GiB_s"], a["decompress_GiB_s"], b["compress_GiB_s"], b["decompress_GiB_s"],
              a["compress_GiB_s"] / b["compress_GiB_s"], a["decompress_GiB_s"] / b["decompress_GiB_s"]), flush=True)
