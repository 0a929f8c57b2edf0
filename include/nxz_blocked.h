/* nxz_blocked.h -- blocked gzip files on the batched engine (libnxz_amd.so).
 *
 * the on-disk step either side of the hot path.  The reference's file layer
 * (lib/nx_gzlib.c, samples/nx_gzip.c, samples/zpipe.c) writes one gzip member through one
 * stream; here a buffer is cut into blocks, the blocks are compressed as ONE batch on the GPU and
 * the device writes the file image itself: a run of gzip members (RFC 1952), one per block, each
 * with the 6-byte "BC" extra subfield holding the member's size (the BGZF layout of bgzip /
 * htslib).  Any gzip reader decodes the result (multi-member files are standard); a reader that
 * knows the subfield hops from member to member and inflates them in parallel, which is what
 * nxz_blocked_inflate does on the GPU.  Host buffers in, host buffers out.
 */
#ifndef NXZ_BLOCKED_H
#define NXZ_BLOCKED_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NXZ_BLOCKED_BLOCK 65280u             /* source bytes per member (0xff00, as bgzip) */

typedef int (*nxz_sink_fn)(void *user, const void *buf, size_t len);    /* 0 = ok, else the call fails with -EIO */

typedef struct nxz_blocked_opts {
	int      device;        /* -1: NX_GZIP_DEV_NUM / current device */
	int      fixed;         /* 1: fixed Huffman (FC 0x00); 0: dynamic Huffman, one table per `group` blocks */
	uint32_t block_size;    /* 0 = NXZ_BLOCKED_BLOCK; multiple of 16, <= 65280 */
	uint32_t chunk_blocks;  /* blocks per batch (0 = 4096 for deflate, 8192 for inflate); two batches are in flight */
	uint32_t group;         /* dynamic: blocks per table (0 = 64); the table comes from the group's first block */
	uint32_t reserved[3];
} nxz_blocked_opts_t;

/* Compress len bytes at src (host) into gzip members, handed to `sink` in order.  An empty
 * input gives no member (write the end marker).  Returns 0 or a negative errno; *out_len (may be
 * NULL) = bytes handed to the sink. */
int nxz_blocked_deflate(const void *src, size_t len, const nxz_blocked_opts_t *opts,
			nxz_sink_fn sink, void *user, uint64_t *out_len);

/* The 28-byte empty member BGZF files end with. */
int nxz_blocked_end_marker(nxz_sink_fn sink, void *user);

/* Walks the members of a blocked gzip image without inflating: returns 0 and the number of
 * members, the uncompressed size (sum of ISIZE) and the length of the prefix made of whole
 * members with the subfield (`consumed` <= len: a partial member or foreign data follows when it
 * is smaller).  Any of the out pointers may be NULL. */
int nxz_blocked_scan(const void *src, size_t len, uint64_t *members, uint64_t *usize, size_t *consumed);

/* Inflate the whole members at src (host): every member's payload is one job of a GPU batch;
 * CRC32 and ISIZE of each member are checked.  Returns 0, -EILSEQ when the data is damaged or a
 * check fails, 1 when src does not start with a blocked member (use the stream API instead), or a
 * negative errno.  *consumed (may be NULL) = bytes of src used (whole members only). */
int nxz_blocked_inflate(const void *src, size_t len, const nxz_blocked_opts_t *opts,
			nxz_sink_fn sink, void *user, uint64_t *out_len, size_t *consumed);

#ifdef __cplusplus
}
#endif
#endif
