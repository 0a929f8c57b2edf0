/*
 * nxz_bench.c -- CPU baseline harness (TEST/BENCH INFRASTRUCTURE ONLY, see nxz_oracle.h).
 * Times the oracle's fixed-Huffman deflate ("port") and, beside it, system zlib level 1
 * Z_FIXED -- the library the reference's software path dlopens (lib/sw_zlib.c:283-324) --
 * on the same blocks with T pthreads, shaped like the reference's throughput harness
 * (samples/compdecomp_th.c:134-228: blocks striped across threads, barrier, timed loop).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>
#include "nxz_oracle.h"

typedef struct {
	const uint8_t *blocks; size_t nblocks, blocklen; int tid, nthreads, mode;
	pthread_barrier_t *bar; uint64_t outbytes; int bad;
} arg_t;

static void *worker(void *p)
{
	arg_t *a = p;
	uint32_t *tok = malloc((a->blocklen + 1) * sizeof(uint32_t));
	size_t cap = a->blocklen * 2 + 1024, i;
	uint8_t *out = malloc(cap);
	z_stream zs;
	memset(&zs, 0, sizeof(zs));
	if (a->mode == 1) deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_FIXED);
	if (a->mode == 2) inflateInit2(&zs, -15);
	pthread_barrier_wait(a->bar);
	for (i = a->tid; i < a->nblocks; i += a->nthreads) {
		const uint8_t *b = a->blocks + i * a->blocklen;
		if (a->mode == 0) {
			size_t nt = nxo_lz77(b, 0, a->blocklen, tok);
			uint64_t bits = nxo_encode_fixed(tok, nt, out, cap);
			a->outbytes += (bits + 7) / 8;
		} else if (a->mode == 1) {
			deflateReset(&zs);
			zs.next_in = (Bytef *)b; zs.avail_in = a->blocklen; zs.next_out = out; zs.avail_out = cap;
			if (deflate(&zs, Z_FINISH) != Z_STREAM_END) a->bad++;
			a->outbytes += zs.total_out;
		}
	}
	pthread_barrier_wait(a->bar);
	if (a->mode == 1) deflateEnd(&zs);
	if (a->mode == 2) inflateEnd(&zs);
	free(tok); free(out);
	return NULL;
}

/* mode 0 = oracle fixed-Huffman deflate, 1 = zlib level 1 Z_FIXED.  Returns seconds. */
double nxo_bench_deflate(const uint8_t *blocks, size_t nblocks, size_t blocklen, int nthreads, int mode,
			 uint64_t *outbytes)
{
	pthread_t th[256];
	arg_t a[256];
	pthread_barrier_t bar;
	struct timespec t0, t1;
	int i;
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 256) nthreads = 256;
	pthread_barrier_init(&bar, NULL, nthreads + 1);
	for (i = 0; i < nthreads; i++) {
		a[i] = (arg_t){ blocks, nblocks, blocklen, i, nthreads, mode, &bar, 0, 0 };
		pthread_create(&th[i], NULL, worker, &a[i]);
	}
	pthread_barrier_wait(&bar);
	clock_gettime(CLOCK_MONOTONIC, &t0);
	pthread_barrier_wait(&bar);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	*outbytes = 0;
	for (i = 0; i < nthreads; i++) { pthread_join(th[i], NULL); *outbytes += a[i].outbytes; }
	pthread_barrier_destroy(&bar);
	return (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
}
