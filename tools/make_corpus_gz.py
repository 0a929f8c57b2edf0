"""<MiB> <out.gz>: a zlib -6 .gz of the first MiB of the test corpus (repeated as needed), for tools/inflate_steps.c"""
import sys, zlib
sys.path.insert(0, "tests")
import corpus
_, blocks, _ = corpus.load(65536)
raw = b"".join(b for _, _, b in blocks)
mib = int(sys.argv[1])
plain = (raw * (1 + (mib << 20) // len(raw)))[:mib << 20]
co = zlib.compressobj(6, zlib.DEFLATED, 31)
open(sys.argv[2], "wb").write(co.compress(plain) + co.flush())
