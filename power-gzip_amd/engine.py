"""ctypes binding of libnxz_engine.so (include/nxz_engine.h).

Fails loudly when the shared library is missing or no gfx950 device is present: there is
no CPU fallback on the product path.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

FC_COMPRESS_FHT = 0x00
FC_COMPRESS_DHT = 0x02
FC_COMPRESS_FHT_COUNT = 0x04
FC_COMPRESS_DHT_COUNT = 0x06
FC_COMPRESS_RESUME_FHT = 0x08
FC_COMPRESS_RESUME_DHT_COUNT = 0x0E
FC_COMPRESS_DHTGEN = 0x22           # additive: the engine builds the job's own table (include/nxz_engine.h)
FC_COMPRESS_DHTGEN_COUNT = 0x26
FC_COMPRESS_RESUME_DHTGEN = 0x2A
FC_DECOMPRESS = 0x10
FC_DECOMPRESS_RESUME = 0x14
FC_WRAP = 0x1E

# nxz_batch_job_t / nxz_batch_result_t / nxz_batch_dht_t
JOB_DTYPE = np.dtype([("src", "<u8"), ("dst", "<u8"), ("src_len", "<u4"), ("hist_len", "<u4"),
                      ("dst_cap", "<u4"), ("in_crc", "<u4"), ("in_adler", "<u4"), ("dht_index", "<u4"),
                      ("resume", "<u4"), ("reserved", "<u4")])
RESULT_DTYPE = np.dtype([("cc", "<u4"), ("tpbc", "<u4"), ("tebc", "<u4"), ("spbc", "<u4"),
                         ("crc", "<u4"), ("adler", "<u4"), ("subc", "<u4"), ("sfbt", "<u4")])
DHT_DTYPE = np.dtype([("dhtlen", "<u4"), ("dht", "u1", (292,))])
assert JOB_DTYPE.itemsize == 48 and RESULT_DTYPE.itemsize == 32 and DHT_DTYPE.itemsize == 296


class StreamResume(C.Structure):
    """nxz_stream_resume_t (include/nxz_engine.h): where a deflate stream stands between two calls"""
    _fields_ = [("sfbt", C.c_uint32), ("rem", C.c_uint32), ("dhtlen", C.c_uint32), ("final", C.c_uint32), ("dht", C.c_uint8 * 288)]


class EngineError(RuntimeError):
    pass


def lib_path():
    # NXZ_ENGINE_LIB: another build of the engine in this directory (A/B runs of kernel variants, tools/)
    return os.path.join(HERE, os.environ.get("NXZ_ENGINE_LIB", "libnxz_engine.so"))


_lib = None


def load_library():
    """dlopen the in-tree engine.  Raises EngineError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        # torch first: it ships its own libamdhip64.so.7; loading it before the engine makes both
        # share ONE HIP runtime (same soname), which is what lets them share streams and pointers
        import torch  # noqa: F401
        p = lib_path()
        if not os.path.exists(p):
            raise EngineError("%s not built: run `make -C power-gzip_amd/csrc` (or __graft_entry__.build())" % p)
        L = C.CDLL(p)
        L.nxz_ctx_create.restype = C.c_void_p
        L.nxz_ctx_create.argtypes = [C.c_int]
        L.nxz_ctx_destroy.argtypes = [C.c_void_p]
        L.nxz_last_error.restype = C.c_char_p
        L.nxz_engine_version.restype = C.c_char_p
        L.nxz_compress_bound.restype = C.c_size_t
        L.nxz_compress_bound.argtypes = [C.c_size_t]
        L.nxz_batch_compress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        L.nxz_batch_dhtgen.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.nxz_batch_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.nxz_batch_wrap.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.nxz_inflate_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p]
        L.nxz_inflate_stream_part.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64,
                                              C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                              C.POINTER(StreamResume), C.POINTER(C.c_uint32), C.c_void_p]
        L.nxz_deflate_host_bound.restype = C.c_size_t
        L.nxz_deflate_host_bound.argtypes = [C.c_size_t]
        L.nxz_deflate_host.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.nxz_ctx_sync.argtypes = [C.c_void_p, C.c_void_p]
        L.nxz_copy_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.nxz_ctx_wg_reasons.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.nxz_ctx_wg_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.nx_function_begin.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.nx_function_end.argtypes = [C.c_void_p]
        L.nxu_run_job.argtypes = [C.c_void_p, C.c_void_p]
        L.nx_wait_ticks.restype = C.c_uint64
        L.nx_wait_ticks.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
        L.__crc32_vpmsum.restype = C.c_uint
        L.__crc32_vpmsum.argtypes = [C.c_uint, C.c_char_p, C.c_ulong]
        _lib = L
    return _lib


class Engine:
    """One engine context on a HIP device; batch calls take torch CUDA tensors."""

    def __init__(self, device=0):
        import torch
        self.torch = torch
        self.L = load_library()
        if not torch.cuda.is_available():
            raise EngineError("no GPU visible: the DEFLATE engine has no CPU fallback")
        self.device = device
        self.ctx = self.L.nxz_ctx_create(device)
        if not self.ctx:
            raise EngineError("nxz_ctx_create failed: %s" % self.L.nxz_last_error().decode())
        self.dev = torch.device("cuda", device)

    def close(self):
        if self.ctx:
            self.L.nxz_ctx_destroy(self.ctx)
            self.ctx = None

    # ---- helpers -------------------------------------------------------
    def stream_handle(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def to_device(self, arr: np.ndarray):
        t = self.torch.from_numpy(arr.view(np.uint8).reshape(-1).copy())
        return t.to(self.dev)

    def jobs_strided(self, src, src_stride, src_lens, dst, dst_stride, dst_cap, hist_len=0, dht_index=None,
                     in_crc=0, in_adler=1, resume=0):
        """Job array for n buffers laid out at fixed strides inside two device tensors."""
        n = len(src_lens)
        j = np.zeros(n, JOB_DTYPE)
        idx = np.arange(n, dtype=np.uint64)
        j["src"] = np.uint64(src.data_ptr()) + idx * np.uint64(src_stride)
        j["dst"] = np.uint64(dst.data_ptr()) + idx * np.uint64(dst_stride)
        j["src_len"] = src_lens
        j["hist_len"] = hist_len
        j["dst_cap"] = dst_cap
        j["in_crc"] = in_crc
        j["in_adler"] = in_adler
        j["resume"] = resume
        if dht_index is not None:
            j["dht_index"] = dht_index
        return self.to_device(j)

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError("%s failed (%d): %s" % (what, rc, self.L.nxz_last_error().decode()))

    # ---- batched entry points (asynchronous on torch's current stream) --
    def compress(self, fc, jobs, n, results=None, dht=None, ntables=0, counts=None):
        t = self.torch
        if results is None:
            results = t.empty(n * RESULT_DTYPE.itemsize, dtype=t.uint8, device=self.dev)
        if (fc & 0x4) and counts is None:
            counts = t.empty(n * 316, dtype=t.int32, device=self.dev)
        rc = self.L.nxz_batch_compress(self.ctx, fc, jobs.data_ptr(), n,
                                       dht.data_ptr() if dht is not None else None, ntables,
                                       results.data_ptr(), counts.data_ptr() if counts is not None else None,
                                       self.stream_handle())
        self._check(rc, "nxz_batch_compress")
        return results, counts

    def dhtgen(self, counts, n, tables=None):
        """device dhtgen: counts = int32/uint32 tensor [n * 316] -> uint8 tensor of n nxz_batch_dht_t"""
        t = self.torch
        if tables is None:
            tables = t.zeros(n * DHT_DTYPE.itemsize, dtype=t.uint8, device=self.dev)
        self._check(self.L.nxz_batch_dhtgen(self.ctx, counts.data_ptr(), n, tables.data_ptr(), self.stream_handle()),
                    "nxz_batch_dhtgen")
        return tables

    def decompress(self, jobs, n, results=None, dht_io=None):
        t = self.torch
        if results is None:
            results = t.empty(n * RESULT_DTYPE.itemsize, dtype=t.uint8, device=self.dev)
        rc = self.L.nxz_batch_decompress(self.ctx, jobs.data_ptr(), n, results.data_ptr(),
                                         dht_io.data_ptr() if dht_io is not None else None, self.stream_handle())
        self._check(rc, "nxz_batch_decompress")
        return results

    def copy_device(self, dst, src):
        """dst <- src (uint8 device tensors of equal size, a multiple of 16 bytes) by the engine's 16-bytes-a-lane copy kernel"""
        self._check(self.L.nxz_copy_device(self.ctx, dst.data_ptr(), src.data_ptr(), src.numel(), self.stream_handle()), "nxz_copy_device")

    def wg_reasons(self):
        """of the last decompress batch that went a stream per workgroup: {"handed_back": n, reason: count}"""
        out = (C.c_uint32 * 16)()
        rc = self.L.nxz_ctx_wg_reasons(self.ctx, self.stream_handle(), out)
        if rc:
            return None
        names = ["", "job", "header", "stored", "dht", "tables", "rounds", "no_eob", "token", "space", "dist"]
        d = {names[i]: out[i] for i in range(1, 11) if out[i]}
        d["handed_back"] = out[15]
        return d

    def wg_prof(self):
        """NXZ_WG_PROF=1: one lane's cycles by phase, per stream, of the last batch that went a stream per workgroup"""
        out = (C.c_uint64 * 32)()
        if self.L.nxz_ctx_wg_prof(self.ctx, self.stream_handle(), out):
            return None
        names = ["load", "header", "dht", "tables", "first", "rounds", "write", "list", "match", "out"]
        ns = max(1, out[11])
        d = {names[i]: out[i] / ns for i in range(10)}
        d["total"] = sum(out[i] for i in range(10)) / ns
        d.update(nrounds=out[10] / max(1, out[12]), streams=out[11], blocks=out[12] / ns, pieces=out[13] / max(1, out[12]),
                 jump_rounds=out[14] / ns, jump_cycles=out[16] / ns, redone=out[17] / max(1, out[12]), redone_in_2=out[15] / max(1, out[12]), per_round=[round(out[18 + i] / ns) for i in range(6)], cnt_r3_6=[round(out[24 + i] / max(1, out[12]), 1) for i in range(4)])
        return d

    def inflate_stream(self, src, src_len, dst, first_bit=0, hist=None):
        """one long raw-deflate stream (uint8 device tensor) -> dst (uint8 device tensor), in parallel by
        block-boundary speculation.  Returns (rc, dict): rc 0 / -errno as nxz_inflate_stream."""
        out_len, end_bit = C.c_uint64(), C.c_uint64()
        crc, adler, pieces, rounds = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        rc = self.L.nxz_inflate_stream(self.ctx, src.data_ptr(), src_len, first_bit,
                                       hist.data_ptr() if hist is not None else None, hist.numel() if hist is not None else 0,
                                       dst.data_ptr(), dst.numel(), C.byref(out_len), C.byref(crc), C.byref(adler),
                                       C.byref(end_bit), C.byref(pieces), C.byref(rounds), self.stream_handle())
        return rc, {"out_len": out_len.value, "crc": crc.value, "adler": adler.value, "end_bit": end_bit.value,
                    "pieces": pieces.value, "rounds": rounds.value}

    def inflate_stream_part(self, src, src_len, dst, state=None, first_bit=0, hist=None):
        """a PART of a raw-deflate stream (uint8 device tensor) -> dst, nxz_inflate_stream_part.  `state`: None at a
        block header / the stream's start, else the StreamResume a previous call returned.  Returns (rc, dict);
        dict["state"] is where the stream stands at dict["end_bit"] (state.final: the final block ended there)."""
        st = state if state is not None else StreamResume()
        out_len, end_bit = C.c_uint64(), C.c_uint64()
        crc, adler, pieces = C.c_uint32(), C.c_uint32(), C.c_uint32()
        rc = self.L.nxz_inflate_stream_part(self.ctx, src.data_ptr(), src_len, first_bit,
                                            hist.data_ptr() if hist is not None else None, hist.numel() if hist is not None else 0,
                                            dst.data_ptr(), dst.numel(), C.byref(out_len), C.byref(crc), C.byref(adler),
                                            C.byref(end_bit), C.byref(st), C.byref(pieces), self.stream_handle())
        return rc, {"out_len": out_len.value, "crc": crc.value, "adler": adler.value, "end_bit": end_bit.value,
                    "pieces": pieces.value, "state": st}

    def deflate_host(self, data, fc=FC_COMPRESS_DHTGEN, final=True, cap=None):
        """a HOST buffer (bytes) -> one raw deflate stream (bytes), nxz_deflate_host.  Returns (rc, stream, crc, adler)."""
        bound = self.L.nxz_deflate_host_bound(len(data)) if cap is None else cap
        dst = C.create_string_buffer(max(bound, 1))
        n, crc, adler = C.c_size_t(), C.c_uint32(), C.c_uint32()
        rc = self.L.nxz_deflate_host(self.ctx, fc, data, len(data), 1 if final else 0, dst, bound, C.byref(n), C.byref(crc), C.byref(adler))
        return rc, dst.raw[:n.value] if rc == 0 else b"", crc.value, adler.value

    def wrap(self, jobs, n, results=None):
        t = self.torch
        if results is None:
            results = t.empty(n * RESULT_DTYPE.itemsize, dtype=t.uint8, device=self.dev)
        self._check(self.L.nxz_batch_wrap(self.ctx, jobs.data_ptr(), n, results.data_ptr(), self.stream_handle()),
                    "nxz_batch_wrap")
        return results

    def results_to_host(self, results):
        self.torch.cuda.synchronize(self.dev)
        return results.cpu().numpy().view(RESULT_DTYPE)
