"""ctypes view of oracle/libnxz_oracle.so (the CPU checker).  Test infrastructure only."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None

TOK_MATCH = 0x80000000


class Codes(C.Structure):
    _fields_ = [("ll_len", C.c_uint8 * 288), ("d_len", C.c_uint8 * 32),
                ("ll_code", C.c_uint16 * 288), ("d_code", C.c_uint16 * 32)]


class InflateState(C.Structure):
    _fields_ = [("subc", C.c_uint32), ("sfbt", C.c_uint32), ("rembytecnt", C.c_uint32),
                ("dht", C.c_void_p), ("dhtlen", C.c_int),
                ("out_sfbt", C.c_uint32), ("out_subc", C.c_uint32), ("out_rembytecnt", C.c_uint32),
                ("out_dht", C.c_uint8 * 288), ("out_dhtlen", C.c_int),
                ("spbc", C.c_size_t), ("tpbc", C.c_size_t), ("final_eob", C.c_int), ("err", C.c_int)]


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(ROOT, "oracle", "libnxz_oracle.so"))
        L.nxo_lz77.restype = C.c_size_t
        L.nxo_lz77.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint32)]
        L.nxo_encode_fixed.restype = C.c_uint64
        L.nxo_encode_fixed.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_char_p, C.c_size_t]
        L.nxo_encode_dynamic.restype = C.c_uint64
        L.nxo_encode_dynamic.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
        L.nxo_count.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.nxo_dhtgen.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_char_p,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.nxo_dht_parse.argtypes = [C.c_char_p, C.c_int, C.POINTER(Codes)]
        L.nxo_inflate.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(InflateState)]
        L.nxo_crc32.restype = C.c_uint32
        L.nxo_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.nxo_adler32.restype = C.c_uint32
        L.nxo_adler32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.nxo_crc32_combine.restype = C.c_uint32
        L.nxo_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.nxo_adler32_combine.restype = C.c_uint32
        L.nxo_adler32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.nxo_fill_zero_lzcounts.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32]
        L.nxo_run_job.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def lz77(data: bytes, hist: int = 0):
    n = len(data) - hist
    tok = (C.c_uint32 * max(n, 1))()
    nt = lib().nxo_lz77(data, hist, n, tok)
    return tok, nt


def deflate_fixed(data: bytes, hist: int = 0):
    """returns (bytes, nbits)"""
    tok, nt = lz77(data, hist)
    cap = 2 * len(data) + 1024
    out = C.create_string_buffer(cap)
    bits = lib().nxo_encode_fixed(tok, nt, out, cap)
    assert bits < (1 << 62)
    return out.raw[:(bits + 7) // 8], bits


def counts(tok, nt):
    ll = (C.c_uint32 * 286)()
    d = (C.c_uint32 * 30)()
    lib().nxo_count(tok, nt, ll, d)
    return ll, d


def dhtgen(ll, d, num_ll=286, num_d=30):
    """ll/d are ctypes arrays (modified in place like the reference does). returns (bytes, dhtlen)"""
    buf = C.create_string_buffer(512)
    nb = C.c_int()
    vb = C.c_int()
    lib().nxo_dhtgen(ll, num_ll, d, num_d, buf, C.byref(nb), C.byref(vb))
    dhtlen = nb.value * 8 - ((8 - vb.value) if vb.value else 0)
    return buf.raw[:nb.value], dhtlen


def deflate_dynamic(data: bytes, dht: bytes, dhtlen: int, hist: int = 0):
    tok, nt = lz77(data, hist)
    cap = 2 * len(data) + 2048
    out = C.create_string_buffer(cap)
    bits = lib().nxo_encode_dynamic(tok, nt, dht, dhtlen, out, cap)
    if bits >= (1 << 62):
        return None, bits
    return out.raw[:(bits + 7) // 8], bits


def inflate(src: bytes, cap: int, hist: bytes = b"", **resume):
    st = InflateState()
    keep = None
    for k, v in resume.items():
        if k == "dht":
            keep = C.create_string_buffer(v, len(v))
            st.dht = C.cast(keep, C.c_void_p)
        else:
            setattr(st, k, v)
    buf = C.create_string_buffer(len(hist) + cap + 16)
    buf[:len(hist)] = hist
    s = C.create_string_buffer(src, len(src))
    dst = C.addressof(buf) + len(hist)
    lib().nxo_inflate(C.addressof(s), len(src), dst, cap, len(hist), C.byref(st))
    return buf.raw[len(hist):len(hist) + st.tpbc], st
