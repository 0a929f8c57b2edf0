"""power-gzip_amd: MI355X-native DEFLATE engine behind the libnxz engine boundary.

The product is csrc/ (HIP kernels + the C ABI of include/nxz_engine.h, built in-tree as
libnxz_engine.so).  This Python package is a thin ctypes view of that C ABI for tests and
bench.py; torch is used only for device memory and streams.
"""
from .engine import (Engine, EngineError, JOB_DTYPE, RESULT_DTYPE, DHT_DTYPE, lib_path,
                     FC_COMPRESS_FHT, FC_COMPRESS_FHT_COUNT, FC_COMPRESS_DHT, FC_COMPRESS_DHT_COUNT,
                     FC_COMPRESS_RESUME_FHT, FC_COMPRESS_RESUME_DHT_COUNT, FC_COMPRESS_DHTGEN,
                     FC_COMPRESS_DHTGEN_COUNT, FC_COMPRESS_RESUME_DHTGEN, FC_DECOMPRESS,
                     FC_DECOMPRESS_RESUME, FC_WRAP)
