"""The reference's interoperability matrix (/root/reference oct/generate-test.sh:11-58, oct/tests.mk:54-70,
generated inputs oct/Makefile.am:36-55) on the GPU box: an UNMODIFIED zlib client (power-gzip_amd/minigz, built
from tools/minigz.c against system zlib: the two clients of the matrix in one -- gzip files through the gz*
calls, zlib streams through deflate()/inflate()) runs with and without LD_PRELOAD=libnxz_preload.so:

    compress     engine compresses  | system zlib decompresses
    decompress   system zlib compresses at the level | engine decompresses
    compdecomp   engine compresses  | engine decompresses

levels 1..9 x {gzip, deflate} x {corpus files, empty, random 4 KiB / 13 MiB, sparse 10 MiB, zero 4 KiB /
13 MiB}; the sha256 of what comes out must be the source's.  (All nine levels on the short inputs, 1 / 6 / 9 on
the long ones: every pipeline is two or three processes that open the device.)"""
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINIGZ = os.path.join(ROOT, "power-gzip_amd", "minigz")
PRELOAD = os.path.join(ROOT, "power-gzip_amd", "libnxz_preload.so")


def _inputs():
    import corpus
    rnd = __import__("random").Random(20261003)
    files = {
        "empty": b"",
        "random4k": rnd.randbytes(4096),
        "zero4k": bytes(4096),
        "alice29": open(os.path.join(ROOT, "tests", "golden", "alice29.txt"), "rb").read(),
        "random13M": rnd.randbytes(13 << 20),
        "zero13M": bytes(13 << 20),
        "sparse10M": bytes(10 << 20),
    }
    # two more files of the real-data corpus, whole (an ELF and XML where the fallback corpus is in use)
    _, blocks, _ = corpus.load(65536)
    byname = {}
    for cls, name, b in blocks:
        byname.setdefault((cls, name), []).append(b)
    picked = 0
    for (cls, name), bl in byname.items():
        if name != "alice29.txt" and cls in ("elf", "xml", "exe", "database") and picked < 2:
            files["corpus-" + name] = b"".join(bl)
            picked += 1
    return files


SHORT = ("empty", "random4k", "zero4k", "alice29")


def _run(cmd, data, preload):
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    if preload:
        env["LD_PRELOAD"] = PRELOAD
        env["NX_GZIP_TYPE_SELECTOR"] = "2"          # the engine, always: no AUTO-mode escape to software zlib
    p = subprocess.run([MINIGZ] + cmd, input=data, env=env, capture_output=True)
    assert p.returncode == 0, (cmd, preload, p.stderr[-300:])
    return p.stdout


def _combo(args):
    name, data, level, typ, want = args
    z = ["-z"] if typ == "deflate" else []
    lv = ["-%d" % level]
    failures = []
    nx_comp = _run(z + lv, data, True)
    if hashlib.sha256(_run(z + ["-d"], nx_comp, False)).hexdigest() != want:
        failures.append("%s.%d.compress.%s" % (name, level, typ))
    if hashlib.sha256(_run(z + ["-d"], nx_comp, True)).hexdigest() != want:
        failures.append("%s.%d.compdecomp.%s" % (name, level, typ))
    sw_comp = _run(z + lv, data, False)
    if hashlib.sha256(_run(z + ["-d"], sw_comp, True)).hexdigest() != want:
        failures.append("%s.%d.decompress.%s" % (name, level, typ))
    # the engine's output is a real compression of compressible data (not a stored copy)
    if name in ("zero13M", "sparse10M", "alice29") and len(nx_comp) > len(data) // 2:
        failures.append("%s.%d.%s: %d bytes out of %d" % (name, level, typ, len(nx_comp), len(data)))
    return failures


def test_oct_matrix_under_ld_preload():
    assert os.path.exists(MINIGZ) and os.path.exists(PRELOAD)
    files = _inputs()
    combos = []
    for name, data in files.items():
        want = hashlib.sha256(data).hexdigest()
        for level in (range(1, 10) if name in SHORT else (1, 6, 9)):
            for typ in ("gzip", "deflate"):
                combos.append((name, data, level, typ, want))
    with ThreadPoolExecutor(max_workers=4) as ex:
        failures = [f for fl in ex.map(_combo, combos) for f in fl]
    assert not failures, failures
    assert len(combos) >= 4 * 9 * 2 + 3 * 3 * 2


def test_the_client_really_lands_in_the_engine(tmp_path):
    """the matrix would pass with a preload that did nothing: the library's call statistics (NX_GZIP_TRACE=8, the
    lines of the reference's print_stats, lib/nx_zlib.c:876-955) count the client's deflate / inflate calls as
    served by the engine ("(nx)"), none by software zlib"""
    import re
    data = open(os.path.join(ROOT, "tests", "golden", "alice29.txt"), "rb").read() * 8
    for cmd, key in ((["-z", "-6"], "deflate"), (["-z", "-d"], "inflate")):
        log = tmp_path / (key + ".log")
        env = dict(os.environ, LD_PRELOAD=PRELOAD, NX_GZIP_TYPE_SELECTOR="2", NX_GZIP_TRACE="8", NX_GZIP_LOGFILE=str(log))
        inp = data if key == "deflate" else __import__("zlib").compress(data, 6)
        p = subprocess.run([MINIGZ] + cmd, input=inp, env=env, capture_output=True)
        assert p.returncode == 0, p.stderr[-300:]
        text = log.read_text(errors="replace")
        nx = re.search(r"%s\(nx\): (\d+)" % key, text)
        sw = re.search(r"%s\(sw\): (\d+)" % key, text)
        assert nx and int(nx.group(1)) > 0, text[-600:]
        assert sw and int(sw.group(1)) == 0, text[-600:]


def test_auto_mode_routes_by_the_measured_break_even(tmp_path):
    """AUTO mode (the default, NX_GZIP_TYPE_SELECTOR=0): a stream whose first call brings ALL its input (Z_FINISH) and
    less of it than the break-even (nxz_config auto_comp_min 128 KiB / auto_dec_min 1 MiB, measured against zlib on the
    same box: profiles/r03_api_sweep.txt) is reopened in software zlib.  A client that streams through fixed buffers
    without Z_FINISH stays on the engine however small its first call, as in the reference (lib/nx_zlib.h:389-419: "the
    first call may not have enough input"; round 3 switched such clients to software for good -- advisor finding).
    Either way the bytes round-trip."""
    import re
    import zlib
    alice = open(os.path.join(ROOT, "tests", "golden", "alice29.txt"), "rb").read()

    def stats(cmd, inp, key):
        log = tmp_path / ("auto_%s_%d.log" % (key, len(inp)))
        env = dict(os.environ, LD_PRELOAD=PRELOAD, NX_GZIP_TYPE_SELECTOR="0", NX_GZIP_TRACE="8", NX_GZIP_LOGFILE=str(log))
        p = subprocess.run([MINIGZ] + cmd, input=inp, env=env, capture_output=True)
        assert p.returncode == 0, p.stderr[-300:]
        text = log.read_text(errors="replace")
        return p.stdout, int(re.search(r"%s\(nx\): (\d+)" % key, text).group(1)), int(re.search(r"%s\(sw\): (\d+)" % key, text).group(1))

    small, big = alice[:40000], alice * 8                     # one call of 40 000 bytes with Z_FINISH; 256 KiB chunks
    out, nx, sw = stats(["-z", "-6"], small, "deflate")
    assert nx == 0 and sw > 0 and zlib.decompress(out) == small
    out, nx, sw = stats(["-z", "-6"], big, "deflate")
    assert nx > 0 and sw == 0 and zlib.decompress(out) == big
    # minigz's inflate loop never says Z_FINISH: the engine keeps the stream, small or large
    out, nx, sw = stats(["-z", "-d"], zlib.compress(small, 6), "inflate")
    assert nx > 0 and sw == 0 and out == small
    rnd = __import__("random").Random(7).randbytes(3 << 20)    # 3 MiB that do not compress
    big_z = zlib.compress(rnd, 1)
    out, nx, sw = stats(["-z", "-d"], big_z, "inflate")         # 256 KiB per call
    assert nx > 0 and sw == 0 and out == rnd
    # inflate(Z_FINISH) with a small stream at hand is software zlib's
    code = (
        "import ctypes as C, sys, zlib\n"
        "L = C.CDLL(None)\n"
        "class Z(C.Structure):\n"
        "    _fields_ = [('next_in', C.c_void_p), ('avail_in', C.c_uint), ('total_in', C.c_ulong), ('next_out', C.c_void_p), ('avail_out', C.c_uint),\n"
        "                ('total_out', C.c_ulong), ('msg', C.c_char_p), ('state', C.c_void_p), ('zalloc', C.c_void_p), ('zfree', C.c_void_p),\n"
        "                ('opaque', C.c_void_p), ('data_type', C.c_int), ('adler', C.c_ulong), ('reserved', C.c_ulong)]\n"
        "data = open(sys.argv[1], 'rb').read()\n"
        "z = Z(); src = C.create_string_buffer(data, len(data)); dst = C.create_string_buffer(1 << 20)\n"
        "assert L.inflateInit_(C.byref(z), b'1.2.11', C.sizeof(Z)) == 0\n"
        "z.next_in = C.cast(src, C.c_void_p); z.avail_in = len(data); z.next_out = C.cast(dst, C.c_void_p); z.avail_out = 1 << 20\n"
        "assert L.inflate(C.byref(z), 4) == 1\n"
        "sys.stdout.buffer.write(dst.raw[:z.total_out]); L.inflateEnd(C.byref(z))\n")
    f = tmp_path / "small.z"
    f.write_bytes(zlib.compress(small, 6))
    log = tmp_path / "finish.log"
    env = dict(os.environ, LD_PRELOAD=PRELOAD, NX_GZIP_TYPE_SELECTOR="0", NX_GZIP_TRACE="8", NX_GZIP_LOGFILE=str(log))
    p = subprocess.run([__import__("sys").executable, "-c", code, str(f)], env=env, capture_output=True)
    assert p.returncode == 0 and p.stdout == small, p.stderr[-300:]
    text = log.read_text(errors="replace")
    assert int(re.search(r"inflate\(nx\): (\d+)", text).group(1)) == 0 and int(re.search(r"inflate\(sw\): (\d+)", text).group(1)) > 0, text[-500:]
    # the one-shot call with the whole stream at hand goes to the engine
    code = (
        "import ctypes as C, sys, zlib\n"
        "L = C.CDLL(None)\n"
        "data = open(sys.argv[1], 'rb').read()\n"
        "n = C.c_ulong(4 << 20); dst = C.create_string_buffer(4 << 20)\n"
        "L.uncompress.argtypes = [C.c_char_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong]\n"
        "assert L.uncompress(dst, C.byref(n), data, len(data)) == 0\n"
        "sys.stdout.buffer.write(dst.raw[:n.value])\n")
    f = tmp_path / "big.z"
    f.write_bytes(big_z)
    log = tmp_path / "oneshot.log"
    env = dict(os.environ, LD_PRELOAD=PRELOAD, NX_GZIP_TYPE_SELECTOR="0", NX_GZIP_TRACE="8", NX_GZIP_LOGFILE=str(log))
    p = subprocess.run([__import__("sys").executable, "-c", code, str(f)], env=env, capture_output=True)
    assert p.returncode == 0 and p.stdout == rnd, p.stderr[-300:]
    text = log.read_text(errors="replace")
    assert int(re.search(r"inflate\(nx\): (\d+)", text).group(1)) > 0 or "uncompress: 1" in text, text[-500:]


AUTO_SWITCH = r'''
import ctypes as C, sys, zlib
L = C.CDLL(None)
class Z(C.Structure):
    _fields_ = [('next_in', C.c_void_p), ('avail_in', C.c_uint), ('total_in', C.c_ulong), ('next_out', C.c_void_p), ('avail_out', C.c_uint),
                ('total_out', C.c_ulong), ('msg', C.c_char_p), ('state', C.c_void_p), ('zalloc', C.c_void_p), ('zfree', C.c_void_p),
                ('opaque', C.c_void_p), ('data_type', C.c_int), ('adler', C.c_ulong), ('reserved', C.c_ulong)]
V, SZ, FINISH = b'1.2.11', C.sizeof(Z), 4
def feed(z, data, cap=1 << 20):
    src = C.create_string_buffer(data, len(data)); dst = C.create_string_buffer(cap)
    z.next_in = C.cast(src, C.c_void_p); z.avail_in = len(data); z.next_out = C.cast(dst, C.c_void_p); z.avail_out = cap
    return src, dst
a, b = b'first member ' * 3000, b'second, small ' * 40
# 1. two gzip members through inflate() + inflateReset(): the second member's first bytes wait inside the engine stream
#    (taken with the first member's part of next_in); the stream must not be reopened in software and lose them
def gz(d):
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    return co.compress(d) + co.flush()
two = gz(a) + gz(b)
z = Z(); assert L.inflateInit2_(C.byref(z), 47, V, SZ) == 0
src, dst = feed(z, two)
assert L.inflate(C.byref(z), 0) == 1 and dst.raw[:z.total_out] == a, "member 1"
used = z.total_in
assert L.inflateReset(C.byref(z)) == 0
rest = two[used:]
src, dst = feed(z, rest)
rc = L.inflate(C.byref(z), FINISH)
assert rc == 1 and dst.raw[:z.total_out] == b, "member 2: rc %d, %d bytes" % (rc, z.total_out)
L.inflateEnd(C.byref(z))
# 2. inflateInit2(15) then inflateReset2(-15): a stream reopened in software must be raw, not zlib-wrapped
co = zlib.compressobj(6, zlib.DEFLATED, -15); raw = co.compress(b) + co.flush()
z = Z(); assert L.inflateInit2_(C.byref(z), 15, V, SZ) == 0
assert L.inflateReset2(C.byref(z), -15) == 0
src, dst = feed(z, raw)
rc = L.inflate(C.byref(z), FINISH)
assert rc == 1 and dst.raw[:z.total_out] == b, "reset2: rc %d" % rc
L.inflateEnd(C.byref(z))
# 3. a copy of a gzip deflate stream keeps its wrapper and level when it is reopened in software
z = Z(); assert L.deflateInit2_(C.byref(z), 9, 8, 31, 9, 0, V, SZ) == 0
z2 = Z(); assert L.deflateCopy(C.byref(z2), C.byref(z)) == 0
src, dst = feed(z2, b)
assert L.deflate(C.byref(z2), FINISH) == 1
out = dst.raw[:z2.total_out]
assert out[:2] == b'\x1f\x8b' and zlib.decompress(out, 31) == b, "copy: %r" % out[:4]
L.deflateEnd(C.byref(z2)); L.deflateEnd(C.byref(z))
print("ok")
'''


def test_auto_mode_switch_keeps_what_the_stream_was_made_with(tmp_path):
    """advisor findings of round 3 on AUTO mode's switch to software zlib: bytes of the next gzip member held over
    inflateReset, windowBits changed by inflateReset2, and the parameters of a copied stream"""
    env = dict(os.environ, LD_PRELOAD=PRELOAD, NX_GZIP_TYPE_SELECTOR="0")
    p = subprocess.run([__import__("sys").executable, "-c", AUTO_SWITCH], env=env, capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stdout + p.stderr[-800:]
