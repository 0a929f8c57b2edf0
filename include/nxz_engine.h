/*
 * nxz_engine.h -- C ABI of the MI355X DEFLATE engine (libnxz_engine.so).
 *
 * This is the drop-in boundary of the hot path.  In libnxz/power-gzip the
 * per-byte work (LZ77, Huffman encode/decode, CRC32/Adler32) is done by the
 * POWER NX accelerator; the library talks to it through exactly six symbols
 * (everything in the reference's lib/ links with only these undefined:
 * SURVEY.md 8(b)).  This library provides those six symbols on top of HIP
 * kernels for gfx950, plus an additive batched interface for device-resident
 * buffers.
 *
 * Reference interfaces replaced (paths relative to the libnxz tree):
 *   nx_function_begin   lib/gzip_vas.c:144   (decl lib/nx_zlib.h:626)
 *   nx_function_end     lib/gzip_vas.c:166   (decl lib/nx_zlib.h:627)
 *   nxu_run_job         lib/gzip_vas.c:281   (decl lib/nx_zlib.h:629), called
 *                       only from nx_submit_job lib/nx_zlib.c:493
 *   nx_wait_ticks       lib/gzip_vas.c:203
 *   tb_freq             lib/gzip_vas.c:92    (read by nx_get_freq inc_nx/nxu.h:81-88)
 *   __crc32_vpmsum      lib/crc32_power.c    (called from lib/crc32_ppc.c:55)
 * Wire format of a job (CRB + CPB + CSB + DDE), function codes and completion
 * codes: inc_nx/nxu.h:155-202, 286-616, 803-857.  The structures below are
 * declared from the byte layout (all multi-byte fields BIG-ENDIAN); their
 * offsets are checked against the reference header in tests/test_abi.py.
 *
 * No torch / HIP types appear in any signature: plain pointers and sizes.
 */
#ifndef NXZ_ENGINE_H
#define NXZ_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 * Job wire format (nx_gzip_crb_cpb_t compatible, 2048 bytes, 2048-aligned)
 * ---------------------------------------------------------------------- */

/* Data descriptor element, 16 bytes (inc_nx/nxu.h:155-170).
 *   count == 0 : direct   -- addr = buffer, bytes = length
 *   count  > 0 : indirect -- addr = array of `count` direct DDEs, bytes =
 *                total; at most `bytes` bytes are processed even when the
 *                list is longer (lib/nx_deflate.c:810-813). */
typedef struct nxz_dde {
	uint32_t count_be;   /* dde_count = (be32toh(count_be) >> 8) & 0xff */
	uint32_t bytes_be;   /* ddebc */
	uint64_t addr_be;    /* ddead: host virtual address */
} __attribute__((aligned(16))) nxz_dde_t;

/* Coprocessor status block, 16 bytes (inc_nx/nxu.h:172-202).  With
 * w = be32toh(flags_be): V = w >> 31, CC = (w >> 8) & 0xff, CE = w & 0xff
 * (the three CE flags live in the top 3 bits of that byte). */
typedef struct nxz_csb {
	uint32_t flags_be;
	uint32_t tpbc_be;    /* target processed byte count */
	uint64_t fsaddr_be;
} __attribute__((aligned(16))) nxz_csb_t;

/* Request block, 256 bytes incl. the CSB at +240 (inc_nx/nxu.h:552-609). */
typedef struct nxz_crb {
	uint32_t fc_be;              /* +0   function code = be32toh(fc_be) & 0xff */
	uint32_t reserved1;          /* +4 */
	uint64_t csb_address_be;     /* +8 */
	nxz_dde_t source;            /* +16 */
	nxz_dde_t target;            /* +32 */
	uint8_t  ccb[16];            /* +48 */
	uint8_t  reserved64[176];    /* +64 */
	volatile nxz_csb_t csb;      /* +240 */
} __attribute__((aligned(128))) nxz_crb_t;

#define NXZ_LLSZ       286
#define NXZ_DSZ        30
#define NXZ_DHT_MAXSZ  288     /* bytes of DHT bit string in the CPB */

/* Parameter block, 1680 bytes (inc_nx/nxu.h:286-542). */
typedef struct nxz_cpb {
	/* ---- input region ---- */
	uint32_t in_adler_be;        /* +0   big-endian Adler-32 to continue from */
	uint32_t in_crc_le;          /* +4   CRC-32 to continue from; memory order == gzip trailer order
				      *       (little-endian), see lib/nx_deflate.c:436-449 */
	uint32_t in_w2_be;           /* +8   w=be32toh: in_histlen = w >> 20 (16-byte units); in_subc = w & 7 */
	uint32_t in_w3_be;           /* +12  w=be32toh: in_sfbt = (w >> 16) & 15; in_rembytecnt = w & 0xffff;
				      *       in_dhtlen = w & 0xfff (bits) */
	uint8_t  in_dht[NXZ_DHT_MAXSZ];  /* +16 */
	uint8_t  reserved_in[80];    /* +304 */
	/* ---- output region ---- */
	uint32_t out_adler_be;       /* +384 */
	uint32_t out_crc_le;         /* +388 memory order == gzip trailer order (little-endian) */
	uint32_t out_w2_be;          /* +392 w=be32toh: out_tebc = (w >> 16) & 7; out_subc = w & 0xffff */
	uint32_t out_w3_be;          /* +396 w=be32toh: out_sfbt = (w >> 16) & 15; out_rembytecnt = w & 0xffff;
				      *       out_dhtlen = w & 0xfff */
	union {                      /* +400 */
		uint32_t out_spbc_be;            /* compress w/o counts, wrap */
		uint32_t out_lzcount_be[NXZ_LLSZ + NXZ_DSZ];
		struct {
			uint8_t  out_dht[NXZ_DHT_MAXSZ];
			uint32_t out_spbc_decomp_be;     /* +688 */
		} d;
		uint8_t  qw25[79 * 16];
	} u;
	uint32_t out_spbc_with_count_be; /* +1664 */
	uint8_t  pad[12];
} __attribute__((aligned(128))) nxz_cpb_t;

typedef struct nxz_crb_cpb {
	nxz_crb_t crb;
	nxz_cpb_t cpb;
} __attribute__((aligned(2048))) nxz_crb_cpb_t;

/* Function codes (inc_nx/nxu.h:803-816).  Bit 0x08 = resume (history prefix
 * allowed), bit 0x04 = also return LZ symbol counts, bit 0x02 = dynamic table. */
enum {
	NXZ_FC_COMPRESS_FHT               = 0x00,
	NXZ_FC_COMPRESS_DHT               = 0x02,
	NXZ_FC_COMPRESS_FHT_COUNT         = 0x04,
	NXZ_FC_COMPRESS_DHT_COUNT         = 0x06,
	NXZ_FC_COMPRESS_RESUME_FHT        = 0x08,
	NXZ_FC_COMPRESS_RESUME_DHT        = 0x0a,
	NXZ_FC_COMPRESS_RESUME_FHT_COUNT  = 0x0c,
	NXZ_FC_COMPRESS_RESUME_DHT_COUNT  = 0x0e,
	NXZ_FC_DECOMPRESS                 = 0x10,
	NXZ_FC_DECOMPRESS_SINGLE_BLK      = 0x12,
	NXZ_FC_DECOMPRESS_RESUME          = 0x14,
	NXZ_FC_DECOMPRESS_RESUME_SINGLE_BLK = 0x16,
	NXZ_FC_WRAP                       = 0x1e,
	/* Additive (not in the reference): bit 0x20 on a DHT compress code = the engine generates the
	 * table itself, from the LZ symbol counts of this very job, exactly as the reference's dhtgen()
	 * would (lib/nx_dhtgen.c:945-1034; EOB counted once, unused symbols get no code), and encodes
	 * the job with it.  in_dht / dht[] are ignored. */
	NXZ_FC_COMPRESS_DHTGEN              = 0x22,
	NXZ_FC_COMPRESS_DHTGEN_COUNT        = 0x26,
	NXZ_FC_COMPRESS_RESUME_DHTGEN       = 0x2a,
	NXZ_FC_COMPRESS_RESUME_DHTGEN_COUNT = 0x2e
};

/* Completion codes written to CSB.CC (inc_nx/nxu.h:823-857). */
enum {
	NXZ_CC_OK            = 0,
	NXZ_CC_DATA_LENGTH   = 3,    /* with CE partial bit: normal "source ran out / trailer follows" */
	NXZ_CC_INVALID_OP    = 8,
	NXZ_CC_TARGET_SPACE  = 13,
	NXZ_CC_INVALID_CRB   = 21,
	NXZ_CC_TPBC_GT_SPBC  = 64,
	NXZ_CC_MISSING_CODE  = 66,
	NXZ_CC_INVALID_DIST  = 67,
	NXZ_CC_INVALID_DHT   = 68,
	NXZ_CC_NO_HW         = 254
};
/* CE bits as stored in the 3 most significant bits of the CE byte (inc_nx/nxu.h:762-781) */
#define NXZ_CE_PARTIAL     0x4
#define NXZ_CE_TERMINATE   0x2
#define NXZ_CE_TPBC_VALID  0x1

/* ------------------------------------------------------------------------
 * The six symbols of the reference's device transport
 * ---------------------------------------------------------------------- */

/* Device handle.  Layout-compatible prefix of the reference's struct
 * nx_dev_t (lib/nx_zlib.h:178-194): only paste_addr, fd and function are
 * touched by the transport (lib/gzip_vas.c:94-185); the library allocates
 * the struct and owns every other field. */
typedef struct nxz_dev {
	int   lib_private[8];  /* lock .. creator_pid: owned by the calling library */
	void *paste_addr;      /* +32 engine context (opaque) */
	int   fd;              /* +40 HIP device ordinal + 1 */
	int   function;        /* +44 */
} nxz_dev_t;

#define NXZ_FUNC_COMP_GZIP 2   /* lib/nx_zlib.h: NX_FUNC_COMP_GZIP */

/* Open the engine on HIP device `pri` (-1 = current / $NXZ_DEVICE).
 * 0 on success, -1 with errno set (ENODEV when no gfx950 device or the HIP
 * code object failed to load: there is NO CPU fallback). */
int nx_function_begin(int function, int pri, void *handle);
int nx_function_end(void *handle);

/* Run one job synchronously.  Source/target are HOST virtual addresses in
 * the DDEs; the engine stages them through pinned buffers, runs the kernels
 * on its stream, writes target, the CPB output region and the CSB (V=1).
 * Returns 0 when the job retired (result in csb.cc), -EAGAIN on timeout. */
int nxu_run_job(nxz_crb_cpb_t *job, void *handle);

/* Back-off helper of the retry ladders: sleeps/spins for `ticks` timebase
 * ticks and returns the accumulated wait. */
uint64_t nx_wait_ticks(uint64_t ticks, uint64_t accumulated_ticks, int do_sleep);

/* Timebase frequency in Hz (512 MHz, the POWER timebase the library's delay
 * thresholds are written for). */
extern uint64_t tb_freq;

/* CRC-32 used by the library's exported crc32() (lib/crc32_ppc.c:55 passes
 * the pre-inverted crc; this returns the raw register like the vpmsum code). */
unsigned int __crc32_vpmsum(unsigned int crc, const unsigned char *p, unsigned long len);

/* ------------------------------------------------------------------------
 * Additive batched interface (device-resident buffers).  Names are outside
 * the nx_* / zlib namespaces of lib/Versions.
 * ---------------------------------------------------------------------- */

typedef struct nxz_ctx nxz_ctx_t;

/* One job of a batch.  All pointers are DEVICE pointers; src/dst must be
 * 16-byte aligned.  For compress jobs `hist_len` bytes at src are history
 * (multiple of 16, <= 32768; lib/nx_deflate.c:853-855). */
typedef struct nxz_batch_job {
	const uint8_t *src;       /* [history][source] */
	uint8_t       *dst;
	uint32_t       src_len;   /* bytes at src including history */
	uint32_t       hist_len;
	uint32_t       dst_cap;
	uint32_t       in_crc;    /* running checksums to continue from */
	uint32_t       in_adler;
	uint32_t       dht_index; /* DHT jobs: which table of the batch's dht array */
	uint32_t       resume;    /* decompress resume state: in_rembytecnt | in_sfbt << 16 | in_subc << 20
				   * (0 = start at a block header on a byte boundary, FC 0x10) */
	uint32_t       reserved;  /* flags, additive: NXZ_JOB_SUSPEND_WHEN_FULL */
} nxz_batch_job_t;

/* Decompress jobs (additive; the reference's engine has no such thing and its library runs a job
 * that overflowed again with a quarter of the source, lib/nx_inflate.c:1399-1424): a full target is
 * not an error (CC 13) but a place to suspend, like the end of the source -- CC 3 with the resume state
 * in front of the token that did not fit, spbc = the source bytes used so far, subc = the unused
 * bits of the last of them.  nxu_run_job takes the flag from bit 0 of crb.reserved1 (big-endian 1).
 * Honoured by the stream-per-wave kernels (nxu_run_job, batches below NXZ_INFLATE_LANES_MIN streams);
 * the stream-per-lane kernel reports CC 13 as ever. */
#define NXZ_JOB_SUSPEND_WHEN_FULL 1u

/* Per-job result, written by the device (device memory, 32 bytes). */
typedef struct nxz_batch_result {
	uint32_t cc;          /* completion code (NXZ_CC_*) */
	uint32_t tpbc;        /* bytes written to dst (incl. the partial last byte) */
	uint32_t tebc;        /* compress: valid bits in the last byte, 0 == 8;
			       * decompress: out_rembytecnt when suspended inside a stored block */
	uint32_t spbc;        /* source bytes processed incl. history */
	uint32_t crc;         /* crc32 continued from in_crc */
	uint32_t adler;
	uint32_t subc;        /* decompress: unprocessed source bits */
	uint32_t sfbt;        /* decompress: bits 0..3 source final block type (inc_nx/nxu.h:466-511),
			       * bit 8 = final EOB seen, bits 16..27 = out_dhtlen when suspended
			       * inside a dynamic block (table bits are in the job's dht_io slot) */
} nxz_batch_result_t;

/* DHT table slot for batched dynamic-Huffman jobs (device memory). */
typedef struct nxz_batch_dht {
	uint32_t dhtlen;                 /* bits */
	uint8_t  dht[NXZ_DHT_MAXSZ + 4]; /* RFC1951 3.2.7 bit string, HLIT first */
} nxz_batch_dht_t;

/* Create / destroy an engine context on a HIP device.  In the batch calls
 * `stream` is a hipStream_t passed as void* (NULL = the HIP default stream). */
nxz_ctx_t *nxz_ctx_create(int device);
void       nxz_ctx_destroy(nxz_ctx_t *ctx);
const char *nxz_last_error(void);

/* Batched compress: jobs[n], results[n] (and dht[], counts[]) are DEVICE
 * arrays.  fc is one of the NXZ_FC_COMPRESS_* codes and applies to every job.
 * dht[ntables] (DHT function codes only; not the DHTGEN codes, where the engine makes a table
 * per job) are the tables jobs[].dht_index refers to.  counts (may be NULL unless fc has the COUNT bit): n x 316 uint32
 * (host byte order), LL then D, EOB counted once.
 * Asynchronous on `stream`; returns 0 or a negative errno. */
int nxz_batch_compress(nxz_ctx_t *ctx, int fc,
		       const nxz_batch_job_t *jobs, size_t n,
		       const nxz_batch_dht_t *dht, size_t ntables,
		       nxz_batch_result_t *results, uint32_t *counts,
		       void *stream);

/* The reference's dhtgen() (lib/nx_dhtgen.c:945-1034) on the device: counts[n][316] (286
 * literal/length + 30 distance counts, host byte order, as the COUNT function codes return them;
 * they are taken as they are -- raise zero counts first if the table is to serve other data,
 * lib/nx_dhtgen.c:235) -> tables[n].  Bit for bit what nxz_dhtgen() / the reference produce.
 * Asynchronous on `stream`. */
int nxz_batch_dhtgen(nxz_ctx_t *ctx, const uint32_t *counts, size_t n, nxz_batch_dht_t *tables, void *stream);

/* Batched decompress of raw-deflate streams (FC 0x10, or 0x14 when
 * jobs[].resume / hist_len are set): each job inflates until final EOB, end
 * of source or full target (CC 13); result.sfbt/subc report where it
 * stopped.  dht_io (NULL or n slots): in = table to resume inside a dynamic
 * block, out = table in force when the job suspended inside one. */
int nxz_batch_decompress(nxz_ctx_t *ctx,
			 const nxz_batch_job_t *jobs, size_t n,
			 nxz_batch_result_t *results,
			 nxz_batch_dht_t *dht_io, void *stream);

/* ONE long raw-deflate stream, decoded in parallel by block-boundary speculation (additive; the
 * reference inflates a stream job after job, lib/nx_inflate.c:1060-1762).  src (DEVICE, src_len
 * bytes) holds the stream from bit first_bit on and must reach its final block; hist (DEVICE or
 * NULL): up to 32 KiB that precede the output (dictionary / earlier output); dst (DEVICE).
 * Synchronous.  Returns 0: *out_len bytes at dst, *crc / *adler of exactly those bytes (combine
 * them with yours), *end_bit = first bit behind the final block, *pieces / *rounds for the curious.
 * -ENOTSUP: the stream does not lend itself to it (shorter than 12 KiB, hardly any dynamic blocks, no
 * final block inside src, ...): use nxz_batch_decompress / nxu_run_job's resume loop.  -E2BIG:
 * dst_cap too small (*out_len = bytes needed).  -EILSEQ: not a deflate stream. */
int nxz_inflate_stream(nxz_ctx_t *ctx, const uint8_t *src, uint64_t src_len, uint64_t first_bit,
		       const uint8_t *hist, uint32_t hist_len, uint8_t *dst, uint64_t dst_cap,
		       uint64_t *out_len, uint32_t *crc, uint32_t *adler, uint64_t *end_bit,
		       uint32_t *pieces, uint32_t *rounds, void *stream);

/* The same for a PART of a stream -- what a caller of inflate() holds at one time (additive).  *state
 * in: where the stream stands at first_bit -- all zero at a block header (or the stream's start), else
 * the fields a suspended decompress job reported (out_sfbt with bit 3 set, out_rembytecnt, out_dhtlen /
 * out_dht; first_bit = 8 - in_subc of the partly used first byte); out: the same for *end_bit, which
 * is the end of src unless state->final (the final block ended at *end_bit) or the output of the
 * pieces further on did not fit dst_cap (then *end_bit is a block header before the end of src, state
 * all zero).  -E2BIG only when not even the first piece fits.  Everything else as above. */
typedef struct nxz_stream_resume {
	uint32_t sfbt;                 /* 0, or 0x8 | BFINAL inside a stored block, 0xa | fixed, 0xc | dynamic, 0xe | in a header */
	uint32_t rem;                  /* stored: bytes of the block still to come */
	uint32_t dhtlen;               /* dynamic: bits of the table in dht */
	uint32_t final;                /* out: the final block ended at *end_bit */
	uint8_t  dht[NXZ_DHT_MAXSZ];
} nxz_stream_resume_t;
int nxz_inflate_stream_part(nxz_ctx_t *ctx, const uint8_t *src, uint64_t src_len, uint64_t first_bit,
			    const uint8_t *hist, uint32_t hist_len, uint8_t *dst, uint64_t dst_cap,
			    uint64_t *out_len, uint32_t *crc, uint32_t *adler, uint64_t *end_bit,
			    nxz_stream_resume_t *state, uint32_t *pieces, void *stream);

/* A long HOST buffer -> ONE raw deflate stream in a HOST buffer (additive; what nx_deflate makes of
 * it job after job, lib/nx_deflate.c:1440-1719, for the levels that carry no history from job to job,
 * :654-680).  The source is cut into 64 KiB blocks, compressed side by side (fc = NXZ_FC_COMPRESS_FHT,
 * or NXZ_FC_COMPRESS_DHTGEN: an exact dynamic table per block made on the device), and laid back to
 * back on the device the way the reference strings jobs together: a block that ends inside a byte is
 * followed by an empty stored block (append_sync_flush, :220-243), a block that did not shrink is
 * stored (:1274-1282); with `final` the last block carries BFINAL, otherwise the run ends on a byte
 * boundary and more blocks may follow.  Groups of 256 blocks alternate between two HIP streams, so
 * the host-to-device copy of one group, the kernels of another and the copy back overlap.
 * dst_cap >= nxz_deflate_host_bound(src_len).  *crc / *adler: checksums of the source from 0 / 1
 * (combine with yours).  Synchronous; returns 0 or a negative errno. */
size_t nxz_deflate_host_bound(size_t src_len);
int nxz_deflate_host(nxz_ctx_t *ctx, int fc, const uint8_t *src, size_t src_len, int final,
		     uint8_t *dst, size_t dst_cap, size_t *out_len, uint32_t *crc, uint32_t *adler);
/* The same for the levels that carry history from job to job (5..9: lib/nx_deflate.c:654-680 sets
 * max_history_len to 4..32 KiB; :845-862 puts that much of the earlier input in front of a job's
 * source): every block's window is the hist_max bytes of the INPUT in front of it -- known up
 * front, so the blocks are still compressed side by side --, the first block's the tail of `prev`
 * (prev_len bytes the caller kept of earlier calls; may be NULL).  Blocks are 64 KiB - hist_max long
 * (window + block <= 64 KiB; hist_max is rounded down to a multiple of 16, 32 KiB at most).
 * dst_cap >= nxz_deflate_host_bound_hist(src_len, hist_max). */
size_t nxz_deflate_host_bound_hist(size_t src_len, uint32_t hist_max);
int nxz_deflate_host_hist(nxz_ctx_t *ctx, int fc, const uint8_t *src, size_t src_len, int final, uint32_t hist_max,
			  const uint8_t *prev, size_t prev_len, uint8_t *dst, size_t dst_cap, size_t *out_len,
			  uint32_t *crc, uint32_t *adler);

/* Device memory the engine keeps between calls (workspaces of nxz_inflate_stream / _part: grown on demand,
 * a workspace above NXZ_PINFLATE_KEEP_MB -- default 8192 -- is given back when its call ends): nxz_trim()
 * gives back what no call is using right now and returns the bytes freed.  For processes that share the
 * device with other users (torch, another library). */
size_t nxz_trim(void);

/* The device a context is made on when the caller names none (nx_function_begin with pri = -1, i.e.
 * NX_GZIP_DEV_NUM unset: lib/nx_zlib.c:568-576 "nx_id -1 means open any", :1281-1287): NXZ_DEVICE if set;
 * else the calling thread's device -- the process' first thread gets the current HIP device, every further
 * thread the next visible device in turn (NXZ_DEVICE_POLICY=current: always the current device).  The
 * policy itself, a pure function: requested ordinal (-1 = any), visible devices, current device, index of the
 * calling thread in order of first use, spread on / off -> device, or -1 for "no such device". */
int nxz_pick_device(int requested, int ndev, int current, unsigned thread_index, int spread);

/* Batched wrap (FC 0x1e): copy + crc32 + adler32 from the initial values. */
int nxz_batch_wrap(nxz_ctx_t *ctx, const nxz_batch_job_t *jobs, size_t n,
		   nxz_batch_result_t *results, void *stream);

/* Gzip members (RFC 1952) from the results of a compress batch, written back to back into
 * `packed` (device): member i = 18-byte header carrying the "BC" extra subfield with the
 * member's own size - 1 (the BGZF layout: readers can find every member without inflating),
 * job i's output -- or, when the job failed or did not shrink (CC != 0 or tpbc >= length + 5), a
 * stored block of its source, the per-job fallback of lib/nx_deflate.c:1292-1400 --, CRC32 and
 * ISIZE.  jobs[].dst must be 16-byte aligned as the compress batch requires; source blocks of
 * at most 65 280 bytes keep every member within BGZF's 64 KiB.  offsets (device, n + 1
 * uint64): start of each member in `packed`; offsets[n] = total bytes.  `packed` needs
 * n * 26 + sum(max(tpbc, length + 5)) bytes at most.  Asynchronous on `stream`. */
int nxz_batch_pack_gzip(nxz_ctx_t *ctx, const nxz_batch_job_t *jobs, const nxz_batch_result_t *results,
			size_t n, uint64_t *offsets, uint8_t *packed, void *stream);

/* Device memory, pinned host memory, streams and asynchronous copies, for callers that hold
 * host buffers and do not link the HIP runtime themselves.  A stream made here is passed as
 * the `stream` argument of the batch calls; nxz_stream_destroy also releases the per-stream
 * scratch the batch calls keep. */
void *nxz_dev_malloc(nxz_ctx_t *ctx, size_t bytes);
void  nxz_dev_free(nxz_ctx_t *ctx, void *p);
void *nxz_pinned_malloc(nxz_ctx_t *ctx, size_t bytes);
void  nxz_pinned_free(nxz_ctx_t *ctx, void *p);
void *nxz_stream_create(nxz_ctx_t *ctx);
void  nxz_stream_destroy(nxz_ctx_t *ctx, void *stream);
int   nxz_copy_to_device(nxz_ctx_t *ctx, void *dst_dev, const void *src_host, size_t bytes, void *stream);
int   nxz_copy_to_host(nxz_ctx_t *ctx, void *dst_host, const void *src_dev, size_t bytes, void *stream);
/* Measurement aid: a device-to-device copy by a 16-bytes-a-lane kernel (what the roofline's measured HBM peak is taken with);
 * bytes a multiple of 16, both pointers 16-byte aligned; asynchronous on `stream`. */
int   nxz_copy_device(nxz_ctx_t *ctx, void *dst_dev, const void *src_dev, size_t bytes, void *stream);

/* Measurement aid (bench.py's roofline): with timing on, every compress batch records events
 * around its kernels; nxz_ctx_stage_ms waits for them and returns the milliseconds spent in the
 * LZ77, table generator and entropy kernels since the last call, and how many launches of each. */
void nxz_ctx_stage_timing(nxz_ctx_t *ctx, int on);
int  nxz_ctx_stage_ms(nxz_ctx_t *ctx, double ms[3], unsigned *launches);
/* Measurement aid: of the last nxz_batch_decompress of n streams on `stream` that went a stream per lane, how many
 * streams the fixed-code-only kernel handed back to the general one (waits for the stream; -ENOENT: no such batch). */
int  nxz_ctx_lanes_handed_back(nxz_ctx_t *ctx, void *stream, size_t n, uint32_t *count);
/* Measurement aid: of the last nxz_batch_decompress on `stream` that went a stream per workgroup (nxz_inflate_wg.hip), how many
 * streams that kernel handed back to the stream-per-wavefront kernel (out16[15]) and why (out16[1..10]: the job's fields, a block
 * header, a stored block, a dynamic table, its sub-tables, too many rounds, no end-of-block, a bad token, no room, a bad distance);
 * waits for the stream; -ENOENT: no such batch. */
int  nxz_ctx_wg_reasons(nxz_ctx_t *ctx, void *stream, uint32_t *out16);
/* ... and, for a batch run with NXZ_WG_PROF=1, the cycles of one lane by phase and the counts (12 words: load, block headers, tables, first
 * pass, later rounds, writing pass, matches, out; rounds, streams, coded blocks, pieces) */
int  nxz_ctx_wg_prof(nxz_ctx_t *ctx, void *stream, unsigned long long *out12);

/* Block until everything queued on `stream` by this context has finished. */
int nxz_ctx_sync(nxz_ctx_t *ctx, void *stream);

/* Worst-case compressed size the engine needs as dst_cap for `src_len`
 * source bytes (fixed-Huffman literals are 9 bits, + header/EOB + 16-byte
 * store granularity). */
size_t nxz_compress_bound(size_t src_len);

/* 0 in a process that was forked after a context was created (the HIP runtime does not survive
 * fork(); such a child gets ENODEV / CC 254 from every entry point and should use software zlib,
 * which is what libnxz_preload.so does), else 1. */
int nxz_engine_usable(void);

/* Library version string. */
const char *nxz_engine_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NXZ_ENGINE_H */
