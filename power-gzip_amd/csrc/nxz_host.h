// nxz_host.h -- internal declarations shared by the host-side sources of libnxz_amd.so
#ifndef NXZ_HOST_H
#define NXZ_HOST_H
#include <stdint.h>
#include <stddef.h>
#include "../../include/nxz_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

struct nxz_dht_state;
typedef struct nxz_dht_state nxz_dht_state;

void nxz_fill_zero_lzcounts(uint32_t *ll, uint32_t *d, uint32_t val);
int  nxz_dhtgen_batch(const uint32_t *counts, size_t n, nxz_batch_dht_t *tables, int nthreads);
int  nxz_dhtgen(uint32_t *lhist, int num_lhist, uint32_t *dhist, int num_dhist,
		uint8_t *dht, int *dht_num_bytes, int *dht_num_valid_bits);
void nxz_dht_top_keys(const uint32_t *ll, int lit_and_len, int key[3]);
int  nxz_dht_builtin_count(void);                                   /* 35: lib/nx_dht_builtin.c */
int  nxz_dht_builtin_get(int i, uint8_t *dht_out, uint32_t *dhtlen_out, int key[3]);
nxz_dht_state *nxz_dht_begin(void);
void nxz_dht_end(nxz_dht_state *s);
nxz_dht_state *nxz_dht_copy(const nxz_dht_state *s);
void nxz_dht_lookup(nxz_dht_state *s, const uint32_t *counts, long source_bytes,
		    uint8_t *dht_out, uint32_t *dhtlen_out);


/* a finished inflate stream: the bytes it took from next_in that lie behind its trailer (nxz_stream.cpp) */
struct z_stream_s;
size_t nxz_inflate_unget_size(struct z_stream_s *strm);
void   nxz_inflate_take_unget(struct z_stream_s *strm, unsigned char *dst);

/* AUTO mode's switchable streams (nxz_zlib_api.cpp): the parameters an engine stream was opened with, and whether
 * nothing has happened to it yet (no input taken, no dictionary, no header) -- then it can still become a
 * software zlib stream with the same parameters.  Return 0 if strm is not such a stream. */
int nxz_deflate_pristine(struct z_stream_s *strm, int *level, int *wbits, int *strategy, int *memlevel);
int nxz_inflate_pristine(struct z_stream_s *strm, int *wbits);

#ifdef __cplusplus
}
#endif
#endif
