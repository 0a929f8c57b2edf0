/* inflate() in steps over a .gz file (avail_in / avail_out of 64 KiB - 1 MiB and more): the
 * same loop once through nx_inflate (libnxz_amd.so, the GPU engine) and once through system zlib's
 * inflate on this thread; output sizes and CRC-32 must agree.
 *   usage: inflate_steps <file.gz> [step KiB ...]        (default steps: 64 256 1024 4096 16384)
 * build: make -C power-gzip_amd/csrc   (-> power-gzip_amd/inflate_steps) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../include/nxz_zlib.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

typedef int (*init_fn)(z_streamp, int, const char *, int);
typedef int (*run_fn)(z_streamp, int);
typedef int (*end_fn)(z_streamp);

static int zl_init(z_streamp s, int w, const char *v, int n) { return inflateInit2_(s, w, v, n); }

/* one pass; returns seconds, -1 on error */
static double pass(init_fn init, run_fn run, end_fn end, const uint8_t *gz, size_t n, size_t step, uint8_t *out, size_t *total, uint32_t *crc)
{
	z_stream s;
	memset(&s, 0, sizeof(s));
	if (init(&s, 31, ZLIB_VERSION, (int)sizeof(s)) != Z_OK) return -1;
	size_t fed = 0;
	uint32_t c = 0;
	*total = 0;
	const double t0 = now();
	int rc = Z_OK;
	while (rc != Z_STREAM_END) {
		if (s.avail_in == 0 && fed < n) { const size_t k = n - fed < step ? n - fed : step; s.next_in = (Bytef *)gz + fed; s.avail_in = (uInt)k; fed += k; }
		s.next_out = out; s.avail_out = (uInt)step;
		rc = run(&s, Z_NO_FLUSH);
		if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) { fprintf(stderr, "inflate: %d\n", rc); end(&s); return -1; }
		const size_t got = step - s.avail_out;
		if (rc == Z_BUF_ERROR && got == 0 && s.avail_in == 0 && fed == n) { fprintf(stderr, "truncated\n"); end(&s); return -1; }
		*total += got;
		(void)c;
	}
	const double dt = now() - t0;
	*crc = (uint32_t)s.adler;               /* gzip wrapper: the running CRC-32 */
	end(&s);
	return dt;
}

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: %s <file.gz> [step KiB ...]\n", argv[0]); return 2; }
	FILE *f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
	uint8_t *gz = malloc((size_t)n);
	if (fread(gz, 1, (size_t)n, f) != (size_t)n) { perror("read"); return 2; }
	fclose(f);
	static const size_t dflt[] = { 64, 256, 1024, 4096, 16384 };
	const int ns = argc > 2 ? argc - 2 : 5;
	int bad = 0;
	for (int k = 0; k < ns; k++) {
		const size_t step = (argc > 2 ? (size_t)atoi(argv[2 + k]) : dflt[k]) << 10;
		uint8_t *out = malloc(step);
		size_t tn = 0, tz = 0; uint32_t cn = 0, cz = 0;
		double best = 1e9;
		for (int rep = 0; rep < 3; rep++) {
			const double d = pass(nx_inflateInit2_, nx_inflate, nx_inflateEnd, gz, (size_t)n, step, out, &tn, &cn);
			if (d < 0) { bad++; break; }
			if (d < best) best = d;
		}
		const double dz = pass(zl_init, inflate, inflateEnd, gz, (size_t)n, step, out, &tz, &cz);
		if (tn != tz || cn != cz) { fprintf(stderr, "MISMATCH at step %zu: %zu / %zu bytes, crc %08x / %08x\n", step, tn, tz, cn, cz); bad++; }
		printf("  steps of %6zu KiB in and out: nx_inflate %8.1f ms = %6.3f GiB/s    zlib, this thread %8.1f ms = %6.3f GiB/s   (%zu bytes)\n",
		       step >> 10, best * 1e3, tn / best / 1073741824.0, dz * 1e3, tz / dz / 1073741824.0, tn);
		fflush(stdout);
		free(out);
	}
	return bad ? 1 : 0;
}
