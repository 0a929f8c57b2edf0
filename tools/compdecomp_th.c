/* T threads, each compressing and decompressing its own buffers through the zlib-style API of
 * libnxz_amd.so, one call per buffer -- the shape of the reference's samples/compdecomp_th.c (a file,
 * a thread count; every thread runs compress + decompress rounds and the aggregate rate is printed).
 *   usage: compdecomp_th <file> <threads> [buffer KiB = 64] [buffers per thread = 2048]
 * build: make -C power-gzip_amd/csrc ../../tools/compdecomp_th   (or see tools/Makefile line in README) */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef USE_ZLIB       /* the same harness on system zlib: the CPU beside it (-DUSE_ZLIB -lz) */
#include <zlib.h>
#define nx_compressBound compressBound
#define nx_compress2 compress2
#define nx_uncompress uncompress
#else
#include "../include/nxz_zlib.h"
#endif

static uint8_t *g_data;
static size_t g_len, g_buf, g_per;
static pthread_barrier_t g_bar;

struct Arg { int id; size_t in_bytes, out_bytes; double t_comp, t_decomp; int bad; };

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

static void *worker(void *p)
{
	struct Arg *a = (struct Arg *)p;
	const size_t nbuf = g_len / g_buf;
	unsigned long cap = nx_compressBound(g_buf);
	uint8_t *comp = malloc(cap * g_per), *back = malloc(g_buf);
	unsigned long *clen = malloc(sizeof(unsigned long) * g_per);
	{	/* every thread's first runs, one each way, before the clock starts -- as the reference's harness has them
		   (samples/compdecomp_th.c:161-185: a compress and an uncompress, a barrier after each, "TIMING RUNS start here"
		   at :190): a thread's first call makes its streams and buffers */
		const uint8_t *src = g_data + ((a->id * 7) % nbuf) * g_buf;
		unsigned long n = g_buf;
		clen[0] = cap;
		if (nx_compress2(comp, &clen[0], src, g_buf, 1) != 0) a->bad++;
		pthread_barrier_wait(&g_bar);
		if (nx_uncompress(back, &n, comp, clen[0]) != 0 || n != g_buf) a->bad++;
	}
	pthread_barrier_wait(&g_bar);
	double t0 = now();
	for (size_t i = 0; i < g_per; i++) {
		const uint8_t *src = g_data + ((a->id * 7 + i) % nbuf) * g_buf;
		clen[i] = cap;
		if (nx_compress2(comp + i * cap, &clen[i], src, g_buf, 1) != 0) a->bad++;
		a->in_bytes += g_buf; a->out_bytes += clen[i];
	}
	double t1 = now();
	pthread_barrier_wait(&g_bar);
	double t2 = now();
	for (size_t i = 0; i < g_per; i++) {
		const uint8_t *src = g_data + ((a->id * 7 + i) % nbuf) * g_buf;
		unsigned long n = g_buf;
		if (nx_uncompress(back, &n, comp + i * cap, clen[i]) != 0 || n != g_buf || memcmp(back, src, g_buf)) a->bad++;
	}
	double t3 = now();
	a->t_comp = t1 - t0; a->t_decomp = t3 - t2;
	free(comp); free(back); free(clen);
	return NULL;
}

int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: %s <file> <threads> [buffer KiB] [buffers per thread]\n", argv[0]); return 2; }
	int T = atoi(argv[2]);
	g_buf = (argc > 3 ? (size_t)atoi(argv[3]) : 64) << 10;
	g_per = argc > 4 ? (size_t)atoi(argv[4]) : 2048;
	FILE *f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	fseek(f, 0, SEEK_END); long flen = ftell(f); fseek(f, 0, SEEK_SET);
	size_t want = (size_t)flen < 64 * g_buf ? 64 * g_buf : (size_t)flen;
	g_data = malloc(want);
	if (fread(g_data, 1, flen, f) != (size_t)flen) return 2;
	fclose(f);
	for (size_t i = flen; i < want; i++) g_data[i] = g_data[i - flen];        /* a short file is repeated */
	g_len = want;
	{       /* warm up: opens the engine, builds its staging */
		unsigned long cap = nx_compressBound(g_buf); uint8_t *c = malloc(cap);
		if (nx_compress2(c, &cap, g_data, g_buf, 1) != 0) { fprintf(stderr, "engine not usable\n"); return 1; }
		free(c);
	}
	pthread_t *th = malloc(sizeof(pthread_t) * T);
	struct Arg *args = calloc(T, sizeof(struct Arg));
	pthread_barrier_init(&g_bar, NULL, T);
	for (int i = 0; i < T; i++) { args[i].id = i; pthread_create(&th[i], NULL, worker, &args[i]); }
	size_t in = 0, out = 0; double tc = 0, td = 0; int bad = 0;
	for (int i = 0; i < T; i++) {
		pthread_join(th[i], NULL);
		in += args[i].in_bytes; out += args[i].out_bytes; bad += args[i].bad;
		if (args[i].t_comp > tc) tc = args[i].t_comp;
		if (args[i].t_decomp > td) td = args[i].t_decomp;
	}
	printf("{\"threads\": %d, \"buffer_KiB\": %zu, \"buffers_per_thread\": %zu, \"compress_GiB_s\": %.3f, \"compress_us_per_call\": %.1f, "
	       "\"decompress_GiB_s\": %.3f, \"decompress_us_per_call\": %.1f, \"ratio\": %.3f, \"bad\": %d}\n",
	       T, g_buf >> 10, g_per, in / tc / 1073741824.0, tc / g_per * 1e6, in / td / 1073741824.0, td / g_per * 1e6, (double)in / out, bad);
	return bad != 0;
}
