"""ctypes access to the nx_* stream API (include/nxz_zlib.h) of either build of the host layer:
  model: oracle/libnxz_amd_model.so  (host sources + CPU engine model, CPU tests)
  gpu:   power-gzip_amd/libnxz_amd.so (host sources + HIP engine, -m gpu tests)"""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Z_NO_FLUSH, Z_PARTIAL_FLUSH, Z_SYNC_FLUSH, Z_FULL_FLUSH, Z_FINISH, Z_BLOCK = 0, 1, 2, 3, 4, 5
Z_OK, Z_STREAM_END, Z_NEED_DICT = 0, 1, 2
Z_STREAM_ERROR, Z_DATA_ERROR, Z_MEM_ERROR, Z_BUF_ERROR, Z_VERSION_ERROR = -2, -3, -4, -5, -6
Z_FIXED, Z_DEFAULT_STRATEGY, Z_DEFLATED = 4, 0, 8


class ZStream(C.Structure):
    _fields_ = [("next_in", C.c_void_p), ("avail_in", C.c_uint), ("total_in", C.c_ulong),
                ("next_out", C.c_void_p), ("avail_out", C.c_uint), ("total_out", C.c_ulong),
                ("msg", C.c_char_p), ("state", C.c_void_p), ("zalloc", C.c_void_p), ("zfree", C.c_void_p),
                ("opaque", C.c_void_p), ("data_type", C.c_int), ("adler", C.c_ulong), ("reserved", C.c_ulong)]


_libs = {}


def load(kind):
    if kind not in _libs:
        if kind == "gpu":
            import torch  # noqa: F401  (one HIP runtime, see power-gzip_amd/engine.py)
            C.CDLL(os.path.join(ROOT, "power-gzip_amd", "libnxz_engine.so"), mode=C.RTLD_GLOBAL)
            p = os.path.join(ROOT, "power-gzip_amd", "libnxz_amd.so")
        else:
            p = os.path.join(ROOT, "oracle", "libnxz_amd_model.so")
        L = C.CDLL(p)
        L.nx_deflateInit2_.argtypes = [C.POINTER(ZStream), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.nx_deflate.argtypes = [C.POINTER(ZStream), C.c_int]
        L.nx_deflateEnd.argtypes = [C.POINTER(ZStream)]
        L.nx_deflateReset.argtypes = [C.POINTER(ZStream)]
        L.nx_deflateBound.restype = C.c_ulong
        L.nx_deflateBound.argtypes = [C.POINTER(ZStream), C.c_ulong]
        L.nx_deflateSetDictionary.argtypes = [C.POINTER(ZStream), C.c_char_p, C.c_uint]
        L.nx_inflateInit2_.argtypes = [C.POINTER(ZStream), C.c_int, C.c_char_p, C.c_int]
        L.nx_inflate.argtypes = [C.POINTER(ZStream), C.c_int]
        L.nx_inflateEnd.argtypes = [C.POINTER(ZStream)]
        L.nx_inflateReset.argtypes = [C.POINTER(ZStream)]
        L.nx_inflateSyncPoint.argtypes = [C.POINTER(ZStream)]
        L.nx_inflateSetDictionary.argtypes = [C.POINTER(ZStream), C.c_char_p, C.c_uint]
        L.nx_compress2.argtypes = [C.c_char_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong, C.c_int]
        L.nx_compress.argtypes = [C.c_char_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong]
        L.nx_uncompress.argtypes = [C.c_char_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong]
        L.nx_compressBound.restype = C.c_ulong
        L.nx_compressBound.argtypes = [C.c_ulong]
        for f in ("nx_crc32", "nx_adler32"):
            getattr(L, f).restype = C.c_ulong
            getattr(L, f).argtypes = [C.c_ulong, C.c_char_p, C.c_size_t]
        for f in ("nx_crc32_combine", "nx_adler32_combine"):
            getattr(L, f).restype = C.c_ulong
            getattr(L, f).argtypes = [C.c_ulong, C.c_ulong, C.c_long]
        L.nx_inflateCopy.argtypes = [C.POINTER(ZStream), C.POINTER(ZStream)]
        L.nx_inflateResetKeep.argtypes = [C.POINTER(ZStream)]
        L.nx_gzopen.restype = C.c_void_p
        L.nx_gzopen.argtypes = [C.c_char_p, C.c_char_p]
        L.nx_gzdopen.restype = C.c_void_p
        L.nx_gzdopen.argtypes = [C.c_int, C.c_char_p]
        L.nx_gzwrite.argtypes = [C.c_void_p, C.c_char_p, C.c_uint]
        L.nx_gzread.argtypes = [C.c_void_p, C.c_char_p, C.c_uint]
        L.nx_gzclose.argtypes = [C.c_void_p]
        for f in ("nx_crc32_combine64", "nx_adler32_combine64"):
            getattr(L, f).restype = C.c_ulong
            getattr(L, f).argtypes = [C.c_ulong, C.c_ulong, C.c_long]
        _libs[kind] = L
    return _libs[kind]


VERSION = b"1.2.11"


def deflate_all(L, data, level=-1, wbits=15, strategy=Z_DEFAULT_STRATEGY, step_in=None, step_out=None, flush=Z_NO_FLUSH,
                dictionary=None):
    """Feed `data` in step_in pieces / collect in step_out pieces (like _test_nx_deflate in the reference's
    test/test_utils.c:232-307: middle calls use `flush`, then a Z_FINISH loop).  Returns (bytes, rc list)."""
    st = ZStream()
    rc = L.nx_deflateInit2_(C.byref(st), level, Z_DEFLATED, wbits, 8, strategy, VERSION, C.sizeof(ZStream))
    assert rc == Z_OK, rc
    if dictionary is not None:
        assert L.nx_deflateSetDictionary(C.byref(st), dictionary, len(dictionary)) == Z_OK
    bound = L.nx_deflateBound(C.byref(st), len(data)) + 64
    src = C.create_string_buffer(data, len(data) or 1)
    dst = C.create_string_buffer(bound)
    st.next_in = C.addressof(src)
    st.next_out = C.addressof(dst)
    step_in = step_in or max(len(data), 1)
    step_out = step_out or bound
    rcs = []
    fed = 0
    while fed < len(data):
        k = min(step_in, len(data) - fed)
        st.avail_in = k
        fed += k
        while st.avail_in:
            if st.avail_out == 0:
                st.avail_out = min(step_out, bound - st.total_out)
            rc = L.nx_deflate(C.byref(st), flush)
            rcs.append(rc)
            assert rc == Z_OK, (rc, fed)
    for _ in range(1000000):
        if st.avail_out == 0:
            st.avail_out = min(step_out, bound - st.total_out)
        rc = L.nx_deflate(C.byref(st), Z_FINISH)
        rcs.append(rc)
        if rc == Z_STREAM_END:
            break
        assert rc == Z_OK, rc
    assert st.total_in == len(data)
    assert st.total_out <= L.nx_deflateBound(C.byref(st), len(data))
    out = dst.raw[:st.total_out]
    adler = st.adler
    assert L.nx_deflateEnd(C.byref(st)) == Z_OK
    return out, rcs, adler


def inflate_all(L, comp, wbits=47, step_in=None, step_out=None, flush=Z_NO_FLUSH, cap=None, dictionary=None):
    st = ZStream()
    assert L.nx_inflateInit2_(C.byref(st), wbits, VERSION, C.sizeof(ZStream)) == Z_OK
    if dictionary is not None and wbits < 0:
        assert L.nx_inflateSetDictionary(C.byref(st), dictionary, len(dictionary)) == Z_OK
    cap = cap or (1 << 20)
    src = C.create_string_buffer(comp, len(comp) or 1)
    dst = C.create_string_buffer(cap)
    st.next_in = C.addressof(src)
    st.next_out = C.addressof(dst)
    step_in = step_in or max(len(comp), 1)
    step_out = step_out or cap
    fed = 0
    rc = Z_OK
    for _ in range(10000000):
        if st.avail_in == 0 and fed < len(comp):
            k = min(step_in, len(comp) - fed)
            st.avail_in = k
            fed += k
        if st.avail_out == 0:
            st.avail_out = min(step_out, cap - st.total_out)
        rc = L.nx_inflate(C.byref(st), flush)
        if rc == Z_NEED_DICT and dictionary is not None:
            assert L.nx_inflateSetDictionary(C.byref(st), dictionary, len(dictionary)) == Z_OK
            continue
        if rc in (Z_STREAM_END,) or rc < 0 and not (rc == Z_BUF_ERROR and (fed < len(comp) or st.avail_out == 0) and st.total_out < cap):
            break
    out = dst.raw[:st.total_out]
    res = (out, rc, st.total_in, st.adler)
    L.nx_inflateEnd(C.byref(st))
    return res
