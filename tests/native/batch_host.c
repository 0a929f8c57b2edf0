/* Plain C host of the batch interface (include/nxz_engine.h), host buffers through the engine's own
 * memory helpers: compress n blocks, inflate the outputs, compare.  Built and run by tests/test_blocked.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nxz_engine.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAILED: %s (line %d): %s\n", #x, __LINE__, nxz_last_error()); return 1; } } while (0)

int main(void)
{
	const size_t n = 300, B = 65536;
	nxz_ctx_t *ctx = nxz_ctx_create(-1);
	CHECK(ctx);
	void *st = nxz_stream_create(ctx);
	CHECK(st);
	const size_t slot = (nxz_compress_bound(B) + 15) & ~(size_t)15;
	uint8_t *host = nxz_pinned_malloc(ctx, n * B), *back = nxz_pinned_malloc(ctx, n * B);
	uint8_t *d_in = nxz_dev_malloc(ctx, n * B), *d_out = nxz_dev_malloc(ctx, n * slot), *d_back = nxz_dev_malloc(ctx, n * B);
	nxz_batch_job_t *jobs = nxz_pinned_malloc(ctx, n * sizeof *jobs), *d_jobs = nxz_dev_malloc(ctx, n * sizeof *jobs);
	nxz_batch_result_t *res = nxz_pinned_malloc(ctx, n * sizeof *res), *res2 = nxz_pinned_malloc(ctx, n * sizeof *res);
	nxz_batch_result_t *d_res = nxz_dev_malloc(ctx, n * sizeof *res);
	CHECK(host && back && d_in && d_out && d_back && jobs && d_jobs && res && res2 && d_res);
	/* text-like blocks with repeats; every 7th block is noise (does not shrink: CC 64) */
	uint32_t s = 12345;
	for (size_t i = 0; i < n * B; i++) {
		s = s * 1664525u + 1013904223u;
		const size_t blk = i / B;
		host[i] = blk % 7 == 3 ? (uint8_t)(s >> 24) : (i % B) > 4000 && (s >> 28) ? host[i - 4000] : (uint8_t)("etaoin shrdlu"[(s >> 20) % 13]);
	}
	for (size_t i = 0; i < n; i++) {
		memset(&jobs[i], 0, sizeof jobs[i]);
		jobs[i].src = d_in + i * B; jobs[i].dst = d_out + i * slot;
		jobs[i].src_len = (uint32_t)B; jobs[i].dst_cap = (uint32_t)slot; jobs[i].in_adler = 1;
	}
	CHECK(nxz_copy_to_device(ctx, d_in, host, n * B, st) == 0);
	CHECK(nxz_copy_to_device(ctx, d_jobs, jobs, n * sizeof *jobs, st) == 0);
	CHECK(nxz_batch_compress(ctx, NXZ_FC_COMPRESS_FHT, d_jobs, n, NULL, 0, d_res, NULL, st) == 0);
	CHECK(nxz_copy_to_host(ctx, res, d_res, n * sizeof *res, st) == 0);
	CHECK(nxz_ctx_sync(ctx, st) == 0);
	size_t total = 0, stored = 0;
	for (size_t i = 0; i < n; i++) {
		CHECK(res[i].cc == 0 || res[i].cc == 64);
		if (res[i].cc == 64) stored++;
		total += res[i].tpbc;
		/* the inflate job of block i: its compressed bytes -> d_back */
		jobs[i].src = d_out + i * slot; jobs[i].src_len = res[i].tpbc;
		jobs[i].dst = d_back + i * B; jobs[i].dst_cap = (uint32_t)B;
	}
	CHECK(stored >= n / 7 - 1 && total < n * B * 3 / 4);
	CHECK(nxz_copy_to_device(ctx, d_jobs, jobs, n * sizeof *jobs, st) == 0);
	CHECK(nxz_batch_decompress(ctx, d_jobs, n, d_res, NULL, st) == 0);
	CHECK(nxz_copy_to_host(ctx, res2, d_res, n * sizeof *res, st) == 0);
	CHECK(nxz_copy_to_host(ctx, back, d_back, n * B, st) == 0);
	CHECK(nxz_ctx_sync(ctx, st) == 0);
	for (size_t i = 0; i < n; i++) CHECK(res2[i].cc == 0 && res2[i].tpbc == B && res2[i].crc == res[i].crc && res2[i].adler == res[i].adler);
	CHECK(memcmp(back, host, n * B) == 0);
	nxz_stream_destroy(ctx, st);
	nxz_dev_free(ctx, d_in); nxz_dev_free(ctx, d_out); nxz_dev_free(ctx, d_back); nxz_dev_free(ctx, d_jobs); nxz_dev_free(ctx, d_res);
	nxz_pinned_free(ctx, host); nxz_pinned_free(ctx, back); nxz_pinned_free(ctx, jobs); nxz_pinned_free(ctx, res); nxz_pinned_free(ctx, res2);
	nxz_ctx_destroy(ctx);
	printf("ok: %zu blocks, %zu stored, ratio %.2f\n", n, stored, (double)(n * B) / (double)total);
	return 0;
}
