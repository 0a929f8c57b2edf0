/*
 * nxz_zlib.h -- the nx_* stream API of libnxz (libnxz.h:53-192 in the reference tree) on top
 * of the MI355X DEFLATE engine.  Same names, argument meaning and return codes as the
 * reference's lib/nx_deflate.c, lib/nx_inflate.c, lib/nx_compress.c, lib/nx_uncompr.c,
 * lib/nx_crc.c and lib/nx_adler32.c; `strm` is a real zlib z_stream.
 */
#ifndef NXZ_ZLIB_H
#define NXZ_ZLIB_H
#include <zlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* deflate (lib/nx_deflate.c:544-2233) */
int nx_deflateInit_(z_streamp strm, int level, const char *version, int stream_size);
int nx_deflateInit2_(z_streamp strm, int level, int method, int windowBits, int memLevel,
		     int strategy, const char *version, int stream_size);
#define nx_deflateInit(strm, level) nx_deflateInit_((strm), (level), ZLIB_VERSION, (int)sizeof(z_stream))
#define nx_deflateInit2(strm, level, method, windowBits, memLevel, strategy) \
	nx_deflateInit2_((strm), (level), (method), (windowBits), (memLevel), (strategy), ZLIB_VERSION, (int)sizeof(z_stream))
int nx_deflate(z_streamp strm, int flush);
int nx_deflateEnd(z_streamp strm);
int nx_deflateReset(z_streamp strm);
int nx_deflateResetKeep(z_streamp strm);
unsigned long nx_deflateBound(z_streamp strm, unsigned long sourceLen);
int nx_deflateSetHeader(z_streamp strm, gz_headerp head);
int nx_deflateSetDictionary(z_streamp strm, const unsigned char *dictionary, unsigned int dictLength);
int nx_deflateCopy(z_streamp dest, z_streamp source);

/* inflate (lib/nx_inflate.c:134-1980) */
int nx_inflateInit_(z_streamp strm, const char *version, int stream_size);
int nx_inflateInit2_(z_streamp strm, int windowBits, const char *version, int stream_size);
#define nx_inflateInit(strm) nx_inflateInit_((strm), ZLIB_VERSION, (int)sizeof(z_stream))
#define nx_inflateInit2(strm, windowBits) nx_inflateInit2_((strm), (windowBits), ZLIB_VERSION, (int)sizeof(z_stream))
int nx_inflate(z_streamp strm, int flush);
int nx_inflateEnd(z_streamp strm);
int nx_inflateReset(z_streamp strm);
int nx_inflateReset2(z_streamp strm, int windowBits);
int nx_inflateResetKeep(z_streamp strm);
int nx_inflateCopy(z_streamp dest, z_streamp source);         /* lib/nx_inflate.c:1876-1942 */
int nx_inflateSetDictionary(z_streamp strm, const unsigned char *dictionary, unsigned int dictLength);
int nx_inflateSyncPoint(z_streamp strm);
int nx_inflateGetHeader(z_streamp strm, gz_headerp head);

/* one-shot (lib/nx_compress.c:26-75, lib/nx_uncompr.c:32-88) */
int nx_compress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen);
int nx_compress2(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen, int level);
uLong nx_compressBound(uLong sourceLen);
int nx_uncompress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen);
int nx_uncompress2(Bytef *dest, uLongf *destLen, const Bytef *source, uLong *sourceLen);

/* checksums (lib/nx_crc.c:215-446, lib/nx_adler32.c:81-177) */
unsigned long nx_crc32(unsigned long crc, const unsigned char *buf, size_t len);
unsigned long nx_adler32(unsigned long adler, const unsigned char *buf, size_t len);
unsigned long nx_adler32_z(unsigned long adler, const unsigned char *buf, size_t len);      /* lib/nx_adler32.c:150 */
unsigned long nx_crc32_combine(unsigned long crc1, unsigned long crc2, off_t len2);
unsigned long nx_adler32_combine(unsigned long adler1, unsigned long adler2, off_t len2);
unsigned long nx_crc32_combine64(unsigned long crc1, unsigned long crc2, off_t len2);
unsigned long nx_adler32_combine64(unsigned long adler1, unsigned long adler2, off_t len2);

/* gz files (lib/nx_gzlib.c:68-329) */
void *nx_gzopen(const char *path, const char *mode);
void *nx_gzdopen(int fd, const char *mode);
int nx_gzwrite(void *file, const void *buf, unsigned len);
int nx_gzread(void *file, void *buf, unsigned len);
int nx_gzclose(void *file);

/* Dynamic-Huffman table builder, bit for bit what dhtgen() of lib/nx_dhtgen.c:945-1034 produces
 * (pinned by tests/golden/dhtgen_vectors.json), and a batched form for the device-resident path
 * (additive): counts = n x 316 words as the COUNT function codes write them, tables = n x
 * nxz_batch_dht_t (include/nxz_engine.h); zero counts are raised to 1 first (lib/nx_dhtgen.c:235). */
int nxz_dhtgen(unsigned int *lhist, int num_lhist, unsigned int *dhist, int num_dhist,
	       unsigned char *dht, int *dht_num_bytes, int *dht_num_valid_bits);
struct nxz_batch_dht;
int nxz_dhtgen_batch(const unsigned int *counts, size_t n, struct nxz_batch_dht *tables, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
