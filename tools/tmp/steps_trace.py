import sys, os, zlib
sys.path.insert(0, "tests")
import corpus, zstream as Z
_, blocks, _ = corpus.load(65536)
raw = b"".join(b for _, _, b in blocks)
plain = (raw * 8)[:24 << 20]
co = zlib.compressobj(6, zlib.DEFLATED, 31)
gz = co.compress(plain) + co.flush()
L = Z.load("gpu")
step = int(sys.argv[1])
got, rc, total_in, _ = Z.inflate_all(L, gz, wbits=31, cap=len(plain) + 64, step_in=step, step_out=step)
print("=========== second run", file=sys.stderr, flush=True)
got, rc, total_in, _ = Z.inflate_all(L, gz, wbits=31, cap=len(plain) + 64, step_in=step, step_out=step)
assert got == plain
