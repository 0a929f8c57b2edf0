/*
 * nxz_oracle.h -- CPU restatement of the DEFLATE engine path (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * include, link or execute anything under oracle/.  The product (the HIP
 * engine in power-gzip_amd/csrc) never routes through this code.
 *
 * What is restated, and from where (all paths relative to /root/reference):
 *   - engine ops COMPRESS_{FHT,DHT}[_COUNT], WRAP, DECOMPRESS[_RESUME]:
 *     there is NO implementation of these in the reference (they are POWER NX
 *     silicon; contract = inc_nx/nxu.h:286-616 + the consumer code
 *     lib/nx_deflate.c:969-1078,1209-1412 and lib/nx_inflate.c:1290-1609).
 *     => PARITY UNPINNED for LZ77 token choice / compressed bytes.  The
 *     restatement here defines the deterministic algorithm the HIP kernels
 *     must match bit-for-bit, and is itself pinned by RFC1951 validity
 *     (round trip through system zlib) only.
 *   - dhtgen: lib/nx_dhtgen.c:235-1034 (PINNED: golden vectors generated from
 *     the reference's own file compiled in place, see oracle/Makefile `ref`).
 *   - DHT header parse -> canonical codes: lib/nx_dht_decomp.c:255-653 and
 *     RFC1951 3.2.2/3.2.7 (PINNED by the 35 builtin tables of
 *     lib/nx_dht_builtin.c parsing to exactly in_dhtlen bits).
 *   - crc32/adler32(+combine): lib/nx_crc.c:215-434, lib/nx_adler32.c:81-177
 *     (PINNED by the KATs in test/test_crc32.c:38-180, test/test_adler32.c:38-179).
 */
#ifndef NXZ_ORACLE_H
#define NXZ_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- LZ77 token stream ------------------------------------------------- */
/* literal: the byte value.  match: NXO_TOK_MATCH | (dist-1)<<8 | (len-3). */
#define NXO_TOK_MATCH 0x80000000u
#define NXO_SUBBLOCK  65536u   /* bytes per independently matched sub-block  */
#define NXO_WINDOW    32768u   /* RFC1951 max distance                        */

/* LZ77 over buf[hist .. hist+n): buf[0..hist) is history that may be
 * referenced but is not emitted.  Returns the number of tokens written to
 * tok (capacity >= n).  n may exceed NXO_SUBBLOCK: the source is cut into
 * sub-blocks, each seeing at most NXO_WINDOW bytes before it. */
size_t nxo_lz77(const uint8_t *buf, size_t hist, size_t n, uint32_t *tok);

/* LL(286)+D(30) symbol counts of a token stream (EOB counted once). */
void nxo_count(const uint32_t *tok, size_t ntok, uint32_t ll[286], uint32_t d[30]);

/* ---- Huffman ----------------------------------------------------------- */
typedef struct {
	uint8_t  ll_len[288];   /* code length per lit/len symbol, 0 = absent */
	uint8_t  d_len[32];
	uint16_t ll_code[288];  /* canonical code, ALREADY bit-reversed for LSB-first output */
	uint16_t d_code[32];
} nxo_codes_t;

void nxo_codes_fixed(nxo_codes_t *c);                       /* RFC1951 3.2.6 */
/* canonical codes from lengths (RFC1951 3.2.2); returns 0, or -1 if over-subscribed */
int  nxo_codes_from_lengths(nxo_codes_t *c);

/* Parse a DHT bitstring (HLIT.. as in RFC1951 3.2.7, WITHOUT the 3 block
 * header bits) of dhtlen bits.  Returns the number of bits consumed
 * (== dhtlen on a well-formed table) or <0 on error. */
int  nxo_dht_parse(const uint8_t *dht, int dhtlen, nxo_codes_t *c);

/* Restatement of dhtgen() (lib/nx_dhtgen.c:945): counts are modified in
 * place exactly as the reference does (length_limit).  Returns 0. */
int  nxo_dhtgen(uint32_t *lhist, int num_lhist, uint32_t *dhist, int num_dhist,
		uint8_t *dht, int *dht_num_bytes, int *dht_num_valid_bits);
void nxo_fill_zero_lzcounts(uint32_t *ll, uint32_t *d, uint32_t val);

/* ---- block encoders ---------------------------------------------------- */
/* Encode one deflate block starting at bit 0 of out[0] with BFINAL=1:
 *   fixed:   3 header bits (1, 01) + symbols + EOB
 *   dynamic: 3 header bits (1, 10) + dht bits verbatim + symbols + EOB
 * Returns total number of BITS written, or (uint64_t)-1 if a needed symbol
 * has no code (engine CC=66), (uint64_t)-2 if out_cap is too small. */
uint64_t nxo_encode_fixed(const uint32_t *tok, size_t ntok, uint8_t *out, size_t out_cap);
uint64_t nxo_encode_dynamic(const uint32_t *tok, size_t ntok,
			    const uint8_t *dht, int dhtlen,
			    uint8_t *out, size_t out_cap);

/* ---- inflate ----------------------------------------------------------- */
typedef struct {
	/* in */
	uint32_t subc;        /* on resume: unprocessed bits in first source byte (0 == 8) */
	uint32_t sfbt;        /* on resume: 0 = at a block header; else nxu.h SFBT code   */
	uint32_t rembytecnt;  /* on resume inside a stored block                           */
	const uint8_t *dht;   /* on resume inside a dynamic block                          */
	int      dhtlen;
	/* out */
	uint32_t out_sfbt, out_subc, out_rembytecnt;
	uint8_t  out_dht[288]; int out_dhtlen;
	size_t   spbc;        /* source bytes touched (incl. partially consumed last byte) */
	size_t   tpbc;        /* target bytes produced                                      */
	int      final_eob;   /* saw EOB of a BFINAL block                                  */
	int      err;         /* 0 ok, 13 target full, 66/67/68 data errors                 */
} nxo_inflate_state_t;

/* Inflate raw deflate data src[0..srclen) into dst, with hist bytes of
 * history located immediately before dst (dst[-hist..-1]) readable.
 * Stops at final EOB, end of source, or full target. */
int nxo_inflate(const uint8_t *src, size_t srclen, uint8_t *dst, size_t dstcap,
		size_t hist, nxo_inflate_state_t *st);

/* ---- checksums --------------------------------------------------------- */
uint32_t nxo_crc32(uint32_t crc, const uint8_t *p, size_t n);
uint32_t nxo_adler32(uint32_t adler, const uint8_t *p, size_t n);
uint32_t nxo_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);
uint32_t nxo_adler32_combine(uint32_t a1, uint32_t a2, uint64_t len2);

#ifdef __cplusplus
}
#endif
#endif
