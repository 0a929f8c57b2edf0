// nxz_device.h -- structures shared by the host side of the engine and its HIP kernels.
#ifndef NXZ_DEVICE_H
#define NXZ_DEVICE_H
#include <stdint.h>
#include <stddef.h>
#include "../../include/nxz_engine.h"

// A DHT bit string (RFC1951 3.2.7, HLIT first, as carried in the CPB: inc_nx/nxu.h:390-393)
// parsed once on the device into what the encoder needs.
typedef struct nxz_dht_prepared {
	uint32_t dhtlen;      // bits
	uint32_t status;      // 0 ok, else NXZ_CC_INVALID_DHT
	uint32_t dhtw[74];    // the bit string, zero padded
	uint32_t ll[288];     // bit-reversed code | length << 16 (length 0 = symbol absent)
	uint32_t d[32];
} nxz_dht_prepared_t;

// What the LZ77 kernel hands to the entropy stage, per job (device scratch, NXZ_TOK_STRIDE bytes apart):
// two bitmaps over the positions of the block (bit p of the first: a literal token starts at
// position p; of the second: a match token starts there) and the match tokens in parse order,
// one dword each: length - 3 | (distance - 1) << 8.
#define NXZ_TOK_LITBITS    0u
#define NXZ_TOK_MATCHBITS  8192u
#define NXZ_TOK_RECORDS    16384u
#define NXZ_TOK_STRIDE     106496u     /* 16 KiB of bitmaps + 65536 / 3 records at most */
#define NXZ_TOK_MAXREC     ((NXZ_TOK_STRIDE - NXZ_TOK_RECORDS) / 4u)

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
// Pointers into device memory are used through address space 1: generic ("flat") accesses make
// the compiler drain the LDS queue completely at every wait that follows them.
#define NXZ_GLOBAL_AS __attribute__((address_space(1)))
extern "C" {
#define NXZ_LZ77_MAX_GRID 512          /* workgroups of one LZ77 launch (one per CU) */
#define NXZ_LZ77_FUSED_FHT 2            /* nxz_launch_lz77 count: the kernel writes the finished fixed-Huffman block itself (no entropy launch) */
#define NXZ_LZ77_FUSED_GEN 3            /* ... the finished dynamic-Huffman block, with the table made of its own counts (`tokens`: nxz_lz77_gen_scratch_bytes() of scratch; counts may be NULL) */
size_t nxz_lz77_cand2_bytes(void);     /* scratch of a launch: the second bucket entries in transit */
size_t nxz_lz77_gen_scratch_bytes(void);
int nxz_launch_lz77(int count, const nxz_batch_job_t *jobs, size_t n, uint8_t *tokens, uint16_t *cand2, nxz_batch_result_t *results,
		    uint32_t *counts, uint32_t *job_counter, hipStream_t stream);
int nxz_launch_encode(int dht, int table_per_job, const nxz_batch_job_t *jobs, size_t n, const uint8_t *tokens,
		      const nxz_dht_prepared_t *tables, nxz_batch_result_t *results, hipStream_t stream);
int nxz_launch_dhtgen(const uint32_t *counts, size_t n, nxz_dht_prepared_t *prepared,
		      nxz_batch_dht_t *tables, hipStream_t stream);   /* device dhtgen: counts[n][316] -> tables (either output may be NULL) */
int nxz_launch_dht_prepare(const nxz_batch_dht_t *dht, size_t n, nxz_dht_prepared_t *out, hipStream_t stream);
int nxz_launch_wrap(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, hipStream_t stream);          /* round 1's kernel (kept for comparison: NXZ_WRAP_OLD=1) */
int nxz_launch_wrap_sliced(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, hipStream_t stream);   /* nxz_inflate_lanes.hip: the checksum kernel's pass, storing as it goes */
int nxz_launch_inflate(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results,
		       nxz_batch_dht_t *dht_io, int window_in_lds, const uint32_t *order, hipStream_t stream);   /* order: NULL, or nxz_launch_order_by_length's */
/* the jobs' indices by falling source length, in `workspace` (nxz_order_workspace(n) bytes); NULL when it cannot be made */
size_t nxz_order_workspace(size_t n);
const uint32_t *nxz_launch_order_by_length(const nxz_batch_job_t *jobs, size_t n, uint8_t *workspace, hipStream_t stream);
/* token boundaries inside dynamic blocks (nxz_inflate.hip token_sync_kernel; offsets in bits from src) */
typedef struct nxz_sync_req {
	const uint8_t *src;      /* at or in front of the block's header, 4-byte aligned if the stream is */
	uint32_t srclen;         /* bytes at src */
	uint32_t header_bit;     /* the block's header */
	uint32_t guess_bit;      /* where to look: the boundary found lies behind this bit */
	uint32_t limit_bit;      /* ... and no token looked at reaches this bit (the next block's header) */
} nxz_sync_req_t;
typedef struct nxz_sync_res { uint32_t bit; uint32_t lanes; } nxz_sync_res_t;   /* bit 0xffffffff: none found; lanes bit 31: the block's BFINAL */
size_t nxz_built_tables_bytes(void);
#define NXZ_BLOCKFIND_MORE 3u      /* nxz_launch_find_blocks: `first` holds nseg first headers, then this many further ones per segment (~0: none) */
int nxz_launch_token_sync_more(const nxz_sync_req_t *breqs, uint32_t nb, nxz_batch_dht_t *tables, void *built, uint32_t first,
			       const nxz_sync_req_t *reqs, uint32_t n, nxz_sync_res_t *res, hipStream_t stream);
int nxz_launch_token_sync(const nxz_sync_req_t *breqs, uint32_t nb, nxz_batch_dht_t *tables, void *built,
			  const nxz_sync_req_t *reqs, uint32_t n, nxz_sync_res_t *res, hipStream_t stream);
/* runs of stored blocks (nxz_blockfind.hip stored_walk_kernel) */
typedef struct nxz_walk_req { const uint8_t *src; uint64_t src_len; uint64_t bit; uint32_t rem, bfinal; } nxz_walk_req_t;
typedef struct nxz_walk_res { uint64_t bit; uint32_t flags, reserved; } nxz_walk_res_t;
int nxz_launch_stored_walk(const nxz_walk_req_t *reqs, uint32_t n, nxz_walk_res_t *res, hipStream_t stream);
int nxz_launch_inflate_w16(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io, const void *built,
			   int few_and_even, hipStream_t stream);
int nxz_launch_sample_btype(const nxz_batch_job_t *jobs, size_t n, uint32_t *out, hipStream_t stream);   /* out: pinned host word */
int nxz_launch_cksum(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, hipStream_t stream);
int nxz_launch_pack_stream(const nxz_batch_job_t *jobs, const nxz_batch_result_t *results, size_t n, uint32_t final_index,
			   uint64_t *offsets, uint8_t *packed, hipStream_t stream);
/* several callers' blocks in one batch, each caller's packed as its own stream: blocks [b0, b0 + n) are member's, block `fin`
 * (an index into the batch, 0xffffffff: none) carries BFINAL, its n + 1 offsets stand at offsets[off0 ..], its stream goes to `packed` */
typedef struct nxz_pack_member { uint32_t b0, n, fin, off0; uint8_t *packed; } nxz_pack_member_t;
int nxz_launch_pack_member_streams(const nxz_batch_job_t *jobs, const nxz_batch_result_t *results, size_t nblocks, const nxz_pack_member_t *members,
				   size_t nmembers, const uint16_t *member_of, uint64_t *offsets, hipStream_t stream);
int nxz_launch_copy_items(const void *items, uint32_t n, hipStream_t stream);
int nxz_launch_copy16(const void *src, void *dst, size_t bytes, hipStream_t stream);   /* a plain device-to-device copy, 16 bytes a lane: the roofline's measured peak */   /* items: { src, dst, uint64 bytes } */
size_t nxz_inflate_lanes_workspace(size_t n);
/* nxz_inflate_cut.hip: a batch too small to fill the device a stream per wavefront -- every stream cut inside its first block */
unsigned nxz_inflate_cut_pieces(size_t n);                                   /* pieces per stream for a batch of n (below 2: not worth it) */
size_t nxz_inflate_cut_workspace(size_t n, unsigned pieces, size_t arena);   /* device bytes: control arrays + `arena` bytes for the pieces' elements */
int nxz_launch_copy_out(const nxz_batch_job_t *jobs, const nxz_batch_result_t *results, uint8_t *const *targets, size_t n, hipStream_t stream);   /* results[i].tpbc bytes of jobs[i].dst -> targets[i] */
int nxz_launch_inflate_cut(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io,
			   unsigned pieces, uint8_t *workspace, size_t arena, hipStream_t stream);
int nxz_launch_pack_members(const nxz_batch_job_t *jobs, const nxz_batch_result_t *results, size_t n,
			    uint64_t *offsets, uint8_t *packed, hipStream_t stream);
int nxz_launch_inflate_lanes(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results,
			     nxz_batch_dht_t *dht_io, uint8_t *workspace, int init_fixed, hipStream_t stream);
/* nxz_inflate_wg.hip: a stream per workgroup, source, output and tables in LDS; what it cannot do goes a stream per wavefront behind it */
size_t nxz_inflate_wg_workspace(size_t n);
int nxz_launch_inflate_wg(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io,
			  uint8_t *wg_ws, const uint32_t *order, uint8_t *const *targets, hipStream_t stream);   /* targets (may be NULL): the outputs there too, in the checksum pass */
int nxz_launch_cksum_copy(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, uint8_t *const *targets, hipStream_t stream);
int nxz_inflate_wg_reasons(const uint8_t *wg_ws, uint32_t *out16);
int nxz_inflate_wg_prof(const uint8_t *wg_ws, unsigned long long *out12);
}
#endif
#endif
