"""Build and read engine jobs (CRB + CPB, include/nxz_engine.h) from Python, for tests.

Mirrors what the reference's job runtime does around nx_submit_job
(/root/reference lib/nx_zlib.c:285-341,469-501): fill the DDEs, function code, CPB inputs,
clear the CSB; after completion read CC / TPBC / the CPB output fields.
"""
import ctypes as C
import struct

CRB_CPB_SIZE = 2048
OFF_CSB = 240
OFF_CPB = 256
O_IN_ADLER, O_IN_CRC, O_IN_W2, O_IN_W3, O_IN_DHT = 0, 4, 8, 12, 16
O_OUT_ADLER, O_OUT_CRC, O_OUT_W2, O_OUT_W3, O_OUT_U, O_OUT_SPBC_DECOMP, O_OUT_SPBC_COUNT = 384, 388, 392, 396, 400, 688, 1664


class Job:
    def __init__(self):
        self._raw = C.create_string_buffer(CRB_CPB_SIZE * 2)
        base = C.addressof(self._raw)
        self.addr = (base + CRB_CPB_SIZE - 1) & ~(CRB_CPB_SIZE - 1)
        self.buf = (C.c_uint8 * CRB_CPB_SIZE).from_address(self.addr)
        self._keep = []

    def _put(self, off, fmt, *v):
        struct.pack_into(fmt, self.buf, off, *v)

    def _get(self, off, fmt):
        return struct.unpack_from(fmt, self.buf, off)

    def _dde(self, off, bufs):
        """bufs: list of ctypes buffers (or (buffer, length)); one -> direct DDE, several -> indirect."""
        items = [(b, len(b)) if not isinstance(b, tuple) else b for b in bufs]
        if len(items) == 1:
            b, n = items[0]
            self._put(off, ">IIQ", 0, n, C.addressof(b))
        else:
            lst = C.create_string_buffer(16 * len(items) + 16)
            la = (C.addressof(lst) + 15) & ~15
            total = 0
            for i, (b, n) in enumerate(items):
                struct.pack_into(">IIQ", (C.c_uint8 * 16).from_address(la + 16 * i), 0, 0, n, C.addressof(b))
                total += n
            self._keep.append(lst)
            self._put(off, ">IIQ", len(items) << 8, total, la)
        self._keep.extend(b for b, _ in items)

    def setup(self, fc, src_bufs, dst_bufs, histlen_qw=0, in_crc=0, in_adler=1, dht=None, dhtlen=0,
              subc=0, sfbt=0, rembytecnt=0):
        C.memset(self.addr, 0, CRB_CPB_SIZE)
        self._put(0, ">I", fc)
        self._dde(16, src_bufs)
        self._dde(32, dst_bufs)
        self._put(OFF_CPB + O_IN_ADLER, ">I", in_adler)
        self._put(OFF_CPB + O_IN_CRC, "<I", in_crc)
        self._put(OFF_CPB + O_IN_W2, ">I", (histlen_qw << 20) | (subc & 7))
        w3 = (sfbt & 15) << 16
        if dht is not None:
            w3 |= dhtlen & 0xfff
            C.memmove(self.addr + OFF_CPB + O_IN_DHT, dht, min(len(dht), 288))
        else:
            w3 |= rembytecnt & 0xffff
        self._put(OFF_CPB + O_IN_W3, ">I", w3)

    # ---- results ----
    @property
    def valid(self):
        return self._get(OFF_CSB, ">I")[0] >> 31

    @property
    def cc(self):
        return (self._get(OFF_CSB, ">I")[0] >> 8) & 0xff

    @property
    def ce3(self):
        return (self._get(OFF_CSB, ">I")[0] >> 5) & 7

    @property
    def tpbc(self):
        return self._get(OFF_CSB + 4, ">I")[0]

    @property
    def out_adler(self):
        return self._get(OFF_CPB + O_OUT_ADLER, ">I")[0]

    @property
    def out_crc(self):
        return self._get(OFF_CPB + O_OUT_CRC, "<I")[0]

    @property
    def out_tebc(self):
        return (self._get(OFF_CPB + O_OUT_W2, ">I")[0] >> 16) & 7

    @property
    def out_subc(self):
        return self._get(OFF_CPB + O_OUT_W2, ">I")[0] & 0xffff

    @property
    def out_sfbt(self):
        return (self._get(OFF_CPB + O_OUT_W3, ">I")[0] >> 16) & 15

    @property
    def out_rembytecnt(self):
        return self._get(OFF_CPB + O_OUT_W3, ">I")[0] & 0xffff

    @property
    def out_dhtlen(self):
        return self._get(OFF_CPB + O_OUT_W3, ">I")[0] & 0xfff

    @property
    def out_dht(self):
        return bytes(self.buf[OFF_CPB + O_OUT_U:OFF_CPB + O_OUT_U + 288])

    @property
    def out_spbc(self):
        return self._get(OFF_CPB + O_OUT_U, ">I")[0]

    @property
    def out_spbc_count(self):
        return self._get(OFF_CPB + O_OUT_SPBC_COUNT, ">I")[0]

    @property
    def out_spbc_decomp(self):
        return self._get(OFF_CPB + O_OUT_SPBC_DECOMP, ">I")[0]

    @property
    def lzcounts(self):
        return list(self._get(OFF_CPB + O_OUT_U, ">316I"))


class DevHandle(C.Structure):
    """prefix of the reference's struct nx_dev_t (lib/nx_zlib.h:178-194)"""
    _fields_ = [("lib_private", C.c_int * 8), ("paste_addr", C.c_void_p), ("fd", C.c_int), ("function", C.c_int),
                ("tail", C.c_uint8 * 128)]
