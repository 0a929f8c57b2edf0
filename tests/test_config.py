"""Run-time configuration, table-cache keys and call statistics of the host layer
(include/nxz_config.h; reference: lib/nx_zlib.c:849-869, 1065-1347, 876-955, lib/nx_dht.c:169-237,
lib/nx_deflate.c:648-652, sample file test/nx-zlib.conf).  CPU only: the host sources run over the
CPU engine model, the preload library in software mode."""
import ctypes as C
import os
import re
import subprocess
import sys
import zlib

import numpy as np

import zstream as Z

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRELOAD = os.path.join(ROOT, "power-gzip_amd", "libnxz_preload.so")
KNOBS = ["NX_GZIP_CONFIG", "NX_GZIP_TYPE_SELECTOR", "NX_GZIP_COMP_MODE", "NX_GZIP_DEC_MODE", "NX_GZIP_STRATEGY",
         "NX_GZIP_DHT_CONFIG", "NX_GZIP_TRACE", "NX_GZIP_VERBOSE", "NX_GZIP_LOGFILE", "NX_GZIP_DEV_NUM",
         "NX_GZIP_DEF_BUF_SIZE", "NX_GZIP_AUTO_COMP_MIN", "NX_GZIP_AUTO_DEC_MIN"]


class Config(C.Structure):
    _fields_ = [("verbose", C.c_int), ("trace", C.c_int), ("dht", C.c_int), ("strategy_override", C.c_int),
                ("dev_num", C.c_int), ("mode_deflate", C.c_int), ("mode_inflate", C.c_int),
                ("def_buf_size", C.c_uint32), ("cache_threshold", C.c_uint32),
                ("auto_comp_min", C.c_uint64), ("auto_dec_min", C.c_uint64),
                ("compress_delay", C.c_uint64), ("decompress_delay", C.c_uint64),
                ("logfile", C.c_char * 256), ("cfgfile", C.c_char * 256), ("cfgfile_loaded", C.c_int)]


def lib():
    L = Z.load("model")
    L.nxz_config.restype = C.POINTER(Config)
    L.nxz_str_to_num.restype = C.c_uint64
    L.nxz_str_to_num.argtypes = [C.c_char_p]
    L.nxz_dht_top_keys.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int * 3)]
    return L


def reload(L, tmp_path, env=None, file_text=None):
    for k in KNOBS:
        os.environ.pop(k, None)
    cfg = tmp_path / "nx-zlib.conf"
    if file_text is not None:
        cfg.write_text(file_text)
        os.environ["NX_GZIP_CONFIG"] = str(cfg)
    else:
        os.environ["NX_GZIP_CONFIG"] = str(tmp_path / "absent.conf")
    for k, v in (env or {}).items():
        os.environ[k] = v
    L.nxz_config_reload()
    c = L.nxz_config().contents
    for k in KNOBS:
        os.environ.pop(k, None)
    return c


def test_str_to_num_suffixes():
    L = lib()
    assert L.nxz_str_to_num(b"0") == 0
    assert L.nxz_str_to_num(b"4096") == 4096
    assert L.nxz_str_to_num(b"0x10") == 16
    assert L.nxz_str_to_num(b"64KiB") == 65536
    assert L.nxz_str_to_num(b"1MiB") == 1 << 20
    assert L.nxz_str_to_num(b"2GiB") == 2 << 30
    assert L.nxz_str_to_num(b"8MB") == 2 ** 64 - 1          # unknown suffix (lib/nx_zlib.c:863-866)


def test_defaults(tmp_path):
    L = lib()
    try:
        c = reload(L, tmp_path)
        assert (c.verbose, c.trace, c.dht, c.strategy_override, c.dev_num) == (0, 0, 0, 1, -1)
        assert (c.mode_deflate, c.mode_inflate) == (0, 0)
        assert c.def_buf_size == 1 << 20 and c.cache_threshold == 8192
        assert c.logfile == b"/tmp/nx.log" and c.cfgfile_loaded == 0
        # AUTO mode's break-even sizes (measured, profiles/r03_api_sweep.txt); environment and file keys
        assert (c.auto_comp_min, c.auto_dec_min) == (128 << 10, 1 << 20)
        c = reload(L, tmp_path, env={"NX_GZIP_AUTO_COMP_MIN": "1MiB"}, file_text="auto_dec_min = 4096\n")
        assert (c.auto_comp_min, c.auto_dec_min) == (1 << 20, 4096)
    finally:
        reload(L, tmp_path)


def test_file_keys_and_environment_precedence(tmp_path):
    L = lib()
    text = """# sample in the shape of the reference's test/nx-zlib.conf
logfile = %s
verbose = 2
  trace   =  8     # statistics
dht_config = 1
strategy = 0
def_buf_size = 16MiB
nx_selector = 3
comp_mode = 1
dev_num = 1
not a key line
cache_threshold = 1024
strategy = 1
""" % (tmp_path / "a.log")
    try:
        c = reload(L, tmp_path, file_text=text)
        assert c.cfgfile_loaded == 1
        assert c.logfile == str(tmp_path / "a.log").encode()
        assert (c.verbose, c.trace, c.dht, c.dev_num) == (2, 8, 1, 1)
        assert c.strategy_override == 1                       # a repeated key keeps its last value
        assert c.def_buf_size == 8 << 20                      # clamped to 8 MiB
        assert (c.mode_deflate, c.mode_inflate) == (2, 1)     # selector 3: engine deflate, zlib inflate; comp_mode ignored
        assert c.cache_threshold == 1024
        # the environment wins over the file
        c = reload(L, tmp_path, env={"NX_GZIP_STRATEGY": "0", "NX_GZIP_TYPE_SELECTOR": "1", "NX_GZIP_TRACE": "0x1",
                                     "NX_GZIP_DEF_BUF_SIZE": "4KiB", "NX_GZIP_LOGFILE": str(tmp_path / "b.log")},
                   file_text=text)
        assert c.strategy_override == 0 and (c.mode_deflate, c.mode_inflate) == (1, 1) and c.trace == 1
        assert c.def_buf_size == 65536 and c.logfile == str(tmp_path / "b.log").encode()
        # without a selector the per-direction modes apply; values above 2 mean auto
        c = reload(L, tmp_path, env={"NX_GZIP_COMP_MODE": "2", "NX_GZIP_DEC_MODE": "7"})
        assert (c.mode_deflate, c.mode_inflate) == (2, 0)
        # an invalid strategy value falls back to 0 (lib/nx_zlib.c:1267-1270)
        c = reload(L, tmp_path, env={"NX_GZIP_STRATEGY": "5"})
        assert c.strategy_override == 0
    finally:
        reload(L, tmp_path)


def first_block_type(raw):
    return (raw[0] >> 1) & 3


def deflate_raw(L, data, strategy):
    zs = Z.ZStream()
    assert L.nx_deflateInit2_(C.byref(zs), 6, Z.Z_DEFLATED, -15, 8, strategy, b"1.2.11", C.sizeof(Z.ZStream)) == Z.Z_OK
    src = C.create_string_buffer(data, len(data))
    dst = C.create_string_buffer(2 * len(data) + 4096)
    zs.next_in, zs.avail_in = C.addressof(src), len(data)
    zs.next_out, zs.avail_out = C.addressof(dst), len(dst)
    assert L.nx_deflate(C.byref(zs), Z.Z_FINISH) == Z.Z_STREAM_END
    out = dst.raw[:zs.total_out]
    assert L.nx_deflateEnd(C.byref(zs)) == Z.Z_OK
    assert zlib.decompress(out, -15) == data
    return out


def test_strategy_override_forces_fixed_huffman(tmp_path):
    L = lib()
    data = (b"the quick brown fox jumps over the lazy dog. " * 400)[:16000]
    try:
        reload(L, tmp_path)
        assert first_block_type(deflate_raw(L, data, Z.Z_DEFAULT_STRATEGY)) == 2     # dynamic by default
        assert first_block_type(deflate_raw(L, data, Z.Z_FIXED)) == 1
        reload(L, tmp_path, env={"NX_GZIP_STRATEGY": "0"})
        assert first_block_type(deflate_raw(L, data, Z.Z_DEFAULT_STRATEGY)) == 1     # lib/nx_deflate.c:648-652
    finally:
        reload(L, tmp_path)


def ref_top_keys(ll, scan):
    """lib/nx_dht.c:169-237 restated: one scan, strictly-greater updates, the third place is not
    shifted when a new maximum arrives."""
    cnt, key = [0, 0, 0], [-1, -1, -1]
    for i in range(scan):
        c = int(ll[i])
        if c > cnt[0]:
            cnt[1], key[1] = cnt[0], key[0]
            cnt[0], key[0] = c, i
        elif c > cnt[1]:
            cnt[2], key[2] = cnt[1], key[1]
            cnt[1], key[1] = c, i
        elif c > cnt[2]:
            cnt[2], key[2] = c, i
    return key


def test_table_cache_keys_follow_the_reference_scan():
    L = lib()
    rng = np.random.default_rng(5)
    cases = [np.zeros(286, np.uint32), np.arange(286, dtype=np.uint32), np.arange(286, 0, -1).astype(np.uint32),
             np.full(286, 7, np.uint32)]
    for _ in range(200):
        a = rng.integers(0, rng.integers(2, 5000), 286).astype(np.uint32)
        if rng.random() < 0.3:
            a[257 + rng.integers(0, 29)] = 100000                 # a length symbol dominates
        cases.append(a)
    for a in cases:
        for both in (0, 1):
            got = (C.c_int * 3)()
            L.nxz_dht_top_keys(a.ctypes.data, both, C.byref(got))
            assert list(got) == ref_top_keys(a, 286 if both else 256)
    # rising counts: each new maximum overwrites the second place, the third stays empty
    got = (C.c_int * 3)()
    rising = np.arange(286, dtype=np.uint32)
    L.nxz_dht_top_keys(rising.ctypes.data, 0, C.byref(got))
    assert list(got) == [255, 254, -1]


STATS_PROG = r'''
import ctypes as C, sys, zlib
sys.path.insert(0, sys.argv[2])
import zstream as Z
L = C.CDLL(sys.argv[1])
L.compressBound.restype = C.c_ulong; L.compressBound.argtypes = [C.c_ulong]
L.compress.argtypes = [C.c_char_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong]
L.uncompress.argtypes = [C.c_char_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong]
L.deflateBound.restype = C.c_ulong; L.deflateBound.argtypes = [C.c_void_p, C.c_ulong]
data = bytes(range(256)) * 80                      # 20480 bytes -> avail_in slot 5 (24 KiB line)
n = L.compressBound(len(data)); dst = C.create_string_buffer(n); dl = C.c_ulong(n)
assert L.compress(dst, C.byref(dl), data, len(data)) == 0
back = C.create_string_buffer(len(data)); bl = C.c_ulong(len(data))
assert L.uncompress(back, C.byref(bl), dst.raw[:dl.value], dl.value) == 0
zs = Z.ZStream()
L.deflateInit_.argtypes = [C.POINTER(Z.ZStream), C.c_int, C.c_char_p, C.c_int]
assert L.deflateInit_(C.byref(zs), 6, b"1.2.11", C.sizeof(Z.ZStream)) == 0
L.deflateBound(C.byref(zs), 1000)
src = C.create_string_buffer(data, len(data)); out = C.create_string_buffer(65536)
zs.next_in, zs.avail_in, zs.next_out, zs.avail_out = C.addressof(src), len(data), C.addressof(out), 65536
L.deflate.argtypes = [C.POINTER(Z.ZStream), C.c_int]; L.deflateEnd.argtypes = [C.POINTER(Z.ZStream)]
assert L.deflate(C.byref(zs), 4) == 1
comp = out.raw[:zs.total_out]
assert L.deflateEnd(C.byref(zs)) == 0
zi = Z.ZStream()
L.inflateInit_.argtypes = [C.POINTER(Z.ZStream), C.c_char_p, C.c_int]
L.inflate.argtypes = [C.POINTER(Z.ZStream), C.c_int]; L.inflateEnd.argtypes = [C.POINTER(Z.ZStream)]
assert L.inflateInit_(C.byref(zi), b"1.2.11", C.sizeof(Z.ZStream)) == 0
csrc = C.create_string_buffer(comp, len(comp)); o2 = C.create_string_buffer(len(data))
zi.next_in, zi.avail_in, zi.next_out, zi.avail_out = C.addressof(csrc), len(comp), C.addressof(o2), len(data)
assert L.inflate(C.byref(zi), 4) == 1 and o2.raw == data
assert L.inflateEnd(C.byref(zi)) == 0
'''


def test_statistics_are_gathered_and_printed_at_exit(tmp_path):
    log = tmp_path / "nx.log"
    env = dict(os.environ, NX_GZIP_TYPE_SELECTOR="1", NX_GZIP_TRACE="8", NX_GZIP_LOGFILE=str(log),
               NX_GZIP_CONFIG=str(tmp_path / "absent.conf"))
    subprocess.run([sys.executable, "-c", STATS_PROG, PRELOAD, os.path.join(ROOT, "tests")], env=env, check=True)
    text = log.read_text()
    want = {"deflateInit": 1, "deflate": 1, "\tdeflate(sw)": 1, "\tdeflate(nx)": 0, "deflateBound": 2, "deflateEnd": 1,
            "compress": 1, "inflateInit": 1, "\tinflate(sw)": 1, "\tinflate(nx)": 0, "inflateEnd": 1, "uncompress": 1}
    # deflateBound: 2 = compressBound() + deflateBound(), both through nx_deflateBound (lib/nx_deflate.c:1920)
    for k, v in want.items():
        m = re.search(r"^%s: (\d+)$" % re.escape(k), text, re.M)
        assert m and int(m.group(1)) == v, (k, text)
    assert int(re.search(r"^inflate: (\d+)$", text, re.M).group(1)) == 1
    assert re.search(r"^  deflate_avail_in   24 KiB: 1$", text, re.M), text
    assert re.search(r"^  deflate_avail_out   68 KiB: 1$", text, re.M), text
    assert "deflate data length: 20 KiB" in text
    # nothing is gathered, and no log file appears, without the trace bit
    log2 = tmp_path / "nx2.log"
    env2 = dict(env, NX_GZIP_TRACE="0", NX_GZIP_LOGFILE=str(log2))
    subprocess.run([sys.executable, "-c", STATS_PROG, PRELOAD, os.path.join(ROOT, "tests")], env=env2, check=True)
    assert not log2.exists()


def test_config_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "nxz_config.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(nxz_\w+)\s*\(", hdr))
    assert {"nxz_config", "nxz_config_reload", "nxz_str_to_num", "nxz_stats_get", "nxz_stats_print", "nxz_log"} <= names
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "power-gzip_amd", "libnxz_amd.so")],
                         check=True, capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    for n in sorted(names):
        assert n in exported, "missing export: " + n


def test_delay_thresholds_and_the_average_job_delay(tmp_path, monkeypatch):
    """AUTO mode's "the device is slow" input (lib/nx_zlib.c:1121-1122,1306-1318,1487-1511; lib/nx_zlib.h:443-449):
    thresholds from the config file, the exponential moving average of the job delay, its fading."""
    L = lib()
    L.nxz_avg_delay.restype = C.c_uint64
    L.nxz_set_avg_delay.argtypes = [C.c_uint64]
    L.nxz_device_stats.argtypes = [C.c_uint64, C.c_uint64]
    monkeypatch.delenv("NX_GZIP_CONFIG", raising=False)
    L.nxz_config_reload()
    c = L.nxz_config().contents
    assert (c.compress_delay, c.decompress_delay) == (100000000, 17000000)
    cfg = tmp_path / "nx.conf"
    cfg.write_text("delay_threshold = 5000\n")
    monkeypatch.setenv("NX_GZIP_CONFIG", str(cfg))
    L.nxz_config_reload()
    c = L.nxz_config().contents
    assert (c.compress_delay, c.decompress_delay) == (5000, 5000)
    cfg.write_text("compress_delay = 7\ndecompress_delay = 9\n")
    L.nxz_config_reload()
    c = L.nxz_config().contents
    assert (c.compress_delay, c.decompress_delay) == (7, 9)
    monkeypatch.delenv("NX_GZIP_CONFIG")
    L.nxz_config_reload()
    L.nxz_set_avg_delay(0)
    L.nxz_device_stats(1000, 1000 + 51200)                   # 100 us: the first sample is the average
    assert L.nxz_avg_delay() == 51200
    L.nxz_device_stats(0, 512000)                            # 1 ms: (last + 4 * avg) / 5
    assert L.nxz_avg_delay() == (512000 + 4 * 51200) // 5
    a = L.nxz_avg_delay()
    L.nxz_device_stats(0, 10)                                # too short to be real: ignored
    L.nxz_device_stats(0, 600000000)                         # longer than a second: the process slept
    assert L.nxz_avg_delay() == a
    L.nxz_decrease_delay()
    assert L.nxz_avg_delay() == a - a // 4
    L.nxz_set_avg_delay(0)
