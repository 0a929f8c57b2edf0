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

/* Buffers of any length: item i is buf[off[i] .. off[i] + len[i]).
 * mode 0 = oracle fixed-Huffman deflate, 1 = zlib level 1 Z_FIXED, 3 = zlib level 1 default strategy,
 * 4 = oracle deflate with an exact dynamic table per buffer (nxo_dhtgen), 2 = zlib inflate of raw
 * deflate streams (outbytes = bytes produced).  Returns seconds. */
typedef struct {
	const uint8_t *buf; const uint64_t *off; const uint32_t *len; size_t n; int tid, nthreads, mode;
	pthread_barrier_t *bar; uint64_t outbytes; int bad;
} varg_t;

static void *vworker(void *p)
{
	varg_t *a = p;
	const size_t cap = 65536 * 2 + 4096;
	uint32_t *tok = malloc((65536 + 1) * sizeof(uint32_t));
	uint8_t *out = malloc(cap);
	z_stream zs;
	size_t i;
	memset(&zs, 0, sizeof(zs));
	if (a->mode == 1) deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_FIXED);
	if (a->mode == 3) deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
	if (a->mode == 2) inflateInit2(&zs, -15);
	pthread_barrier_wait(a->bar);
	for (i = a->tid; i < a->n; i += a->nthreads) {
		const uint8_t *b = a->buf + a->off[i];
		const uint32_t n = a->len[i];
		if (a->mode == 0 || a->mode == 4) {
			size_t nt = nxo_lz77(b, 0, n, tok);
			uint64_t bits;
			if (a->mode == 0) bits = nxo_encode_fixed(tok, nt, out, cap);
			else {
				uint32_t ll[286], d[30]; uint8_t dht[320]; int nb, vb;
				nxo_count(tok, nt, ll, d);
				nxo_dhtgen(ll, 286, d, 30, dht, &nb, &vb);
				bits = nxo_encode_dynamic(tok, nt, dht, nb * 8 - (vb ? 8 - vb : 0), out, cap);
			}
			a->outbytes += (bits + 7) / 8;
		} else if (a->mode == 2) {
			inflateReset(&zs);
			zs.next_in = (Bytef *)b; zs.avail_in = n; zs.next_out = out; zs.avail_out = cap;
			if (inflate(&zs, Z_FINISH) != Z_STREAM_END) a->bad++;
			a->outbytes += zs.total_out;
		} else {
			deflateReset(&zs);
			zs.next_in = (Bytef *)b; zs.avail_in = n; zs.next_out = out; zs.avail_out = cap;
			if (deflate(&zs, Z_FINISH) != Z_STREAM_END) a->bad++;
			a->outbytes += zs.total_out;
		}
	}
	pthread_barrier_wait(a->bar);
	if (a->mode == 1 || a->mode == 3) deflateEnd(&zs);
	if (a->mode == 2) inflateEnd(&zs);
	free(tok); free(out);
	return NULL;
}

double nxo_bench_run(const uint8_t *buf, const uint64_t *off, const uint32_t *len, size_t n, int nthreads, int mode,
		     uint64_t *outbytes, int *bad)
{
	pthread_t th[256];
	varg_t a[256];
	pthread_barrier_t bar;
	struct timespec t0, t1;
	int i;
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 256) nthreads = 256;
	pthread_barrier_init(&bar, NULL, nthreads + 1);
	for (i = 0; i < nthreads; i++) {
		a[i] = (varg_t){ buf, off, len, n, i, nthreads, mode, &bar, 0, 0 };
		pthread_create(&th[i], NULL, vworker, &a[i]);
	}
	pthread_barrier_wait(&bar);
	clock_gettime(CLOCK_MONOTONIC, &t0);
	pthread_barrier_wait(&bar);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	*outbytes = 0;
	if (bad) *bad = 0;
	for (i = 0; i < nthreads; i++) { pthread_join(th[i], NULL); *outbytes += a[i].outbytes; if (bad) *bad += a[i].bad; }
	pthread_barrier_destroy(&bar);
	return (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
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
