"""BASELINE configs[0] and the LD_PRELOAD drop-in: the unprefixed zlib API of libnxz_preload.so.
CPU: software mode (NX_GZIP_TYPE_SELECTOR=1) and auto mode without a GPU both end in system zlib
(the reference's sw_zlib.c path), so the bytes must equal zlib's.  GPU: engine mode."""
import os
import subprocess
import sys
import zlib

import pytest

from datagen import make_block, ALICE_LIKE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRELOAD = os.path.join(ROOT, "power-gzip_amd", "libnxz_preload.so")

ONE_SHOT = r'''
import ctypes as C, sys, zlib
L = C.CDLL(sys.argv[1])
L.compressBound.restype = C.c_ulong; L.compressBound.argtypes = [C.c_ulong]
L.compress.argtypes = [C.c_char_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong]
L.uncompress.argtypes = [C.c_char_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong]
data = open(sys.argv[2], "rb").read()
n = L.compressBound(len(data)); dst = C.create_string_buffer(n); dl = C.c_ulong(n)
assert L.compress(dst, C.byref(dl), data, len(data)) == 0
comp = dst.raw[:dl.value]
back = C.create_string_buffer(len(data)); bl = C.c_ulong(len(data))
assert L.uncompress(back, C.byref(bl), comp, len(comp)) == 0 and back.raw[:bl.value] == data
assert zlib.decompress(comp) == data
sys.stdout.write(comp.hex())
'''


def run_one_shot(tmp_path, data, selector):
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    env = dict(os.environ, NX_GZIP_TYPE_SELECTOR=str(selector))
    out = subprocess.run([sys.executable, "-c", ONE_SHOT, PRELOAD, str(f)], env=env, check=True, capture_output=True, text=True)
    return bytes.fromhex(out.stdout)


def test_config1_one_shot_software_path_equals_zlib(tmp_path):
    # BASELINE configs[0] / SURVEY 8(d) C1: compress()/uncompress() one-shot on the reference's
    # samples/alice29.txt (tests/golden/alice29.txt, sha256 pinned) through the CPU fallback: the
    # output is identical to system zlib compress2(level -1) because it IS zlib
    import hashlib
    data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "alice29.txt"), "rb").read()
    assert len(data) == 152089
    assert hashlib.sha256(data).hexdigest() == "7467306ee0feed4971260f3c87421154a05be571d944e9cb021a5713700c38f0"
    assert run_one_shot(tmp_path, data, 1) == zlib.compress(data, -1)
    like = ALICE_LIKE(152089)
    assert run_one_shot(tmp_path, like, 1) == zlib.compress(like, -1)


def test_auto_mode_without_engine_falls_back_to_zlib(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: auto mode would use the engine")
    data = make_block("lz", 50000, 3)
    assert run_one_shot(tmp_path, data, 0) == zlib.compress(data, -1)


PY_ZLIB = r'''
import zlib, sys
data = open(sys.argv[1], "rb").read()
c = zlib.compress(data, 6)
assert zlib.decompress(c) == data
co = zlib.compressobj(6, zlib.DEFLATED, 31)
g = co.compress(data[:70000]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(data[70000:]) + co.flush()
do = zlib.decompressobj(47)
assert do.decompress(g) == data and do.eof
assert zlib.crc32(data) == int(sys.argv[2]) and zlib.adler32(data) == int(sys.argv[3])
sys.stdout.write("%d %d" % (len(c), len(g)))
'''


def _ld_preload_python(tmp_path, selector):
    data = ALICE_LIKE(200000, seed=5)
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    env = dict(os.environ, NX_GZIP_TYPE_SELECTOR=str(selector), LD_PRELOAD=PRELOAD)
    out = subprocess.run([sys.executable, "-c", PY_ZLIB, str(f), str(zlib.crc32(data)), str(zlib.adler32(data))],
                         env=env, check=True, capture_output=True, text=True)
    return [int(x) for x in out.stdout.split()], data


def test_ld_preload_software_mode(tmp_path):
    # README.md:15-18 of the reference: LD_PRELOAD=libnxz.so application
    (n1, n2), data = _ld_preload_python(tmp_path, 1)
    assert n1 == len(zlib.compress(data, 6))


@pytest.mark.gpu
def test_ld_preload_engine_mode_python_zlib(tmp_path):
    # an unmodified application (CPython's zlib module) running on the MI355X engine
    (n1, n2), data = _ld_preload_python(tmp_path, 2)
    assert n1 < len(data) / 1.8 and n2 < len(data) / 1.8


@pytest.mark.gpu
def test_one_shot_engine_mode(tmp_path):
    data = ALICE_LIKE(152089)
    comp = run_one_shot(tmp_path, data, 2)
    assert comp[:2] == b"\x78\x01" and zlib.decompress(comp) == data      # FLEVEL 0 header of the engine layer (Q3)
    assert len(comp) < 0.55 * len(data)


@pytest.mark.gpu
def test_mixed_mode_engine_deflate_software_inflate_with_statistics(tmp_path):
    # NX_GZIP_TYPE_SELECTOR=3: the engine compresses, zlib decompresses (lib/nx_zlib.c:1197-1200);
    # NX_GZIP_TRACE=8 prints where each call went when the process ends (lib/nx_zlib.c:876-955)
    import re
    data = ALICE_LIKE(200000, seed=5)
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    log = tmp_path / "nx.log"
    env = dict(os.environ, NX_GZIP_TYPE_SELECTOR="3", NX_GZIP_TRACE="8", NX_GZIP_LOGFILE=str(log), LD_PRELOAD=PRELOAD)
    subprocess.run([sys.executable, "-c", PY_ZLIB, str(f), str(zlib.crc32(data)), str(zlib.adler32(data))],
                   env=env, check=True, capture_output=True, text=True)
    text = log.read_text()
    count = lambda k: int(re.search(r"^%s: (\d+)$" % re.escape(k), text, re.M).group(1))
    assert count("\tdeflate(nx)") >= 1 and count("\tdeflate(sw)") == 0
    assert count("\tinflate(sw)") >= 1 and count("\tinflate(nx)") == 0
    # a fixed-Huffman override reaches the engine layer: the stream starts with a type-01 block
    env = dict(os.environ, NX_GZIP_TYPE_SELECTOR="2", NX_GZIP_STRATEGY="0", LD_PRELOAD=PRELOAD)
    out = subprocess.run([sys.executable, "-c", "import sys,zlib;d=open(sys.argv[1],'rb').read();"
                          "c=zlib.compress(d,6);assert zlib.decompress(c)==d;sys.stdout.write(c[:3].hex())", str(f)],
                         env=env, check=True, capture_output=True, text=True).stdout
    assert (bytes.fromhex(out)[2] >> 1) & 3 == 1


FORK_PROG = r'''
import ctypes as C, os, sys, zlib
sys.path.insert(0, sys.argv[2])
import zstream as Z
d = open(sys.argv[1], "rb").read()
c1 = zlib.compress(d, 6)                      # parent: the engine (opens the HIP runtime)
assert zlib.decompress(c1) == d
r, w = os.pipe()
pid = os.fork()
if pid == 0:
    rc = 1
    try:
        c2 = zlib.compress(d, 6)              # child: must not touch the parent's HIP state
        ok = zlib.decompress(c2) == d and zlib.decompress(c1) == d
        L = C.CDLL(sys.argv[3])               # the nx_* layer itself refuses cleanly instead of hanging
        zs = Z.ZStream()
        L.nx_deflateInit_.argtypes = [C.POINTER(Z.ZStream), C.c_int, C.c_char_p, C.c_int]
        ok = ok and L.nx_deflateInit_(C.byref(zs), 6, b"1.2.11", C.sizeof(Z.ZStream)) == Z.Z_STREAM_ERROR
        os.write(w, len(c2).to_bytes(8, "little") + c2)
        rc = 0 if ok else 2
    finally:
        os._exit(rc)
os.close(w)
buf = b""
while True:
    b = os.read(r, 1 << 20)
    if not b: break
    buf += b
_, st = os.waitpid(pid, 0)
assert os.WIFEXITED(st) and os.WEXITSTATUS(st) == 0, st
c3 = zlib.compress(d, 6)                      # parent: the engine still works
assert c3 == c1
n = int.from_bytes(buf[:8], "little")
sys.stdout.write(buf[8:8 + n].hex() + " " + c1.hex())
'''


@pytest.mark.gpu
def test_forked_child_falls_back_to_software(tmp_path):
    # the HIP runtime does not survive fork(): the child's streams go to zlib (the reference re-opens
    # its device in the child, lib/nx_zlib.c:529-551; test/test_pid_reuse.c forks with a window open)
    data = ALICE_LIKE(150000, seed=9)
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    env = dict(os.environ, NX_GZIP_TYPE_SELECTOR="0", LD_PRELOAD=PRELOAD)
    out = subprocess.run([sys.executable, "-c", FORK_PROG, str(f), os.path.join(ROOT, "tests"),
                          os.path.join(ROOT, "power-gzip_amd", "libnxz_amd.so")],
                         env=env, check=True, capture_output=True, text=True, timeout=300).stdout
    child, parent = (bytes.fromhex(x) for x in out.split())
    assert child == zlib.compress(data, 6)                   # the child's bytes are software zlib's
    assert parent != child and zlib.decompress(parent) == data


AUTO_SLOW = r'''
import ctypes as C, sys, zlib
L = C.CDLL(sys.argv[1])
L.compressBound.restype = C.c_ulong; L.compressBound.argtypes = [C.c_ulong]
L.compress.argtypes = [C.c_char_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong]
data = open(sys.argv[2], "rb").read()
outs = []
for k in range(6):
    n = L.compressBound(len(data)); dst = C.create_string_buffer(n); dl = C.c_ulong(n)
    assert L.compress(dst, C.byref(dl), data, len(data)) == 0
    assert zlib.decompress(dst.raw[:dl.value]) == data
    outs.append(dst.raw[:dl.value] == zlib.compress(data, -1))
sys.stdout.write(" ".join("sw" if o else "nx" for o in outs))
'''


@pytest.mark.gpu
def test_auto_mode_leaves_a_slow_engine_and_comes_back(tmp_path):
    """AUTO (lib/nx_zlib.h:376-422, lib/nx_deflate.c:714): a stream opened while the engine's average job
    delay is above compress_delay is served by software zlib; streams served in software let the average fade
    (decrease_delay), so the engine is tried again.  With a threshold no real job can meet: the first call
    goes to the engine (no measurement yet), the following ones to zlib."""
    data = make_block("alice", 300000, 11)
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    cfg = tmp_path / "nx.conf"
    cfg.write_text("delay_threshold = 3\n")                   # 3 ticks of 512 MHz: nothing is that fast
    env = dict(os.environ, NX_GZIP_TYPE_SELECTOR="0", NX_GZIP_CONFIG=str(cfg))
    out = subprocess.run([sys.executable, "-c", AUTO_SLOW, PRELOAD, str(f)], env=env, check=True, capture_output=True, text=True).stdout.split()
    assert out[0] == "nx" and "sw" in out[1:], out
    # the default thresholds (0.2 s / 33 ms per job) are never reached by this engine: everything stays on it
    env = dict(os.environ, NX_GZIP_TYPE_SELECTOR="0")
    out = subprocess.run([sys.executable, "-c", AUTO_SLOW, PRELOAD, str(f)], env=env, check=True, capture_output=True, text=True).stdout.split()
    assert out == ["nx"] * 6, out
