// slot_pool_host.cpp -- TEST ONLY.  nx_inflate's large-call path (power-gzip_amd/csrc/nxz_stream.cpp parallel_inflate)
// keeps a few sets of device buffers per DEVICE.  This program stands in for the device side (plain memory with live
// counters, a stream inflated by system zlib) under oracle/libnxz_amd_model.so and lets two callers whose contexts sit on
// DIFFERENT devices take turns: round 3's single pool handed the same set from one to the other and dropped its buffers
// and its stream unfreed at every hand-over (advisor finding, round 3).  Prints "ok <live buffers> <streams>".
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <zlib.h>
#include "../../include/nxz_engine.h"
#include "../../include/nxz_zlib.h"

struct FakeCtx { int dev; };
static FakeCtx g_fake[2] = { {0}, {1} };
static thread_local int t_dev = 0;
static std::atomic<long> g_live[2], g_allocs[2], g_streams[2], g_parts;

extern "C" {
int nx_function_begin(int function, int pri, void *handle)
{
	nxz_dev_t *h = (nxz_dev_t *)handle;
	(void)pri;
	if (function != NXZ_FUNC_COMP_GZIP || !h) return -1;
	h->function = function; h->paste_addr = &g_fake[t_dev]; h->fd = 1;
	return 0;
}
int nxz_ctx_device(nxz_ctx_t *c) { return ((FakeCtx *)c)->dev; }
void *nxz_dev_malloc(nxz_ctx_t *c, size_t n) { int d = nxz_ctx_device(c); g_live[d]++; g_allocs[d]++; return malloc(n ? n : 16); }
void nxz_dev_free(nxz_ctx_t *c, void *p) { if (p) { g_live[nxz_ctx_device(c)]--; free(p); } }
int nxz_copy_to_device(nxz_ctx_t *, void *d, const void *s, size_t n, void *) { memcpy(d, s, n); return 0; }
int nxz_copy_to_host(nxz_ctx_t *, void *d, const void *s, size_t n, void *) { memcpy(d, s, n); return 0; }
int nxz_ctx_sync(nxz_ctx_t *, void *) { return 0; }
void *nxz_stream_create(nxz_ctx_t *c) { return (void *)(intptr_t)(100 * (nxz_ctx_device(c) + 1) + ++g_streams[nxz_ctx_device(c)]); }
int nxz_engine_usable(void) { return 1; }
// one whole raw deflate stream from bit 0 (all this test hands over)
int nxz_inflate_stream_part(nxz_ctx_t *, const uint8_t *src, uint64_t src_len, uint64_t first_bit, const uint8_t *, uint32_t hist_len,
			    uint8_t *dst, uint64_t dst_cap, uint64_t *out_len, uint32_t *crc, uint32_t *adler, uint64_t *end_bit,
			    nxz_stream_resume_t *st, uint32_t *pieces, void *)
{
	if (first_bit || hist_len) return -1;
	z_stream z; memset(&z, 0, sizeof(z));
	if (inflateInit2(&z, -15) != Z_OK) return -1;
	z.next_in = (Bytef *)src; z.avail_in = (uInt)src_len; z.next_out = dst; z.avail_out = (uInt)dst_cap;
	const int rc = inflate(&z, Z_FINISH);
	const uint64_t used = z.total_in, made = z.total_out;
	inflateEnd(&z);
	if (rc != Z_STREAM_END) return -1;
	*out_len = made; *end_bit = used * 8;
	*crc = (uint32_t)crc32(0, dst, (uInt)made); *adler = (uint32_t)adler32(1, dst, (uInt)made);
	st->final = 1;
	if (pieces) *pieces = 1;
	g_parts++;
	return 0;
}
}

int main()
{
	std::vector<uint8_t> src(300000), comp(400000), back(300000);
	uint32_t x = 12345;                                      // (random letters: the stream must be longer than the path's 12 KiB minimum)
	for (size_t i = 0; i < src.size(); i++) { x = x * 1664525u + 1013904223u; src[i] = (uint8_t)('a' + (x >> 24) % 26); }
	uLongf clen = comp.size();
	if (compress2(comp.data(), &clen, src.data(), src.size(), 6) != Z_OK) return 2;
	for (int round = 0; round < 40; round++) {
		t_dev = round & 1;                                   // the caller's context sits on device 0, 1, 0, 1, ...
		z_stream z; memset(&z, 0, sizeof(z));
		if (nx_inflateInit2_(&z, 15, "1.2.11", (int)sizeof(z_stream)) != Z_OK) return 3;
		z.next_in = comp.data(); z.avail_in = (uInt)clen; z.next_out = back.data(); z.avail_out = (uInt)back.size();
		const int rc = nx_inflate(&z, Z_FINISH);
		if (rc != Z_STREAM_END || z.total_out != src.size() || memcmp(back.data(), src.data(), src.size())) { printf("round %d: rc %d\n", round, rc); return 4; }
		nx_inflateEnd(&z);
	}
	if (g_parts < 40) { printf("the large-call path was not taken (%ld of 40)\n", (long)g_parts); return 5; }
	// one set per device was enough for one caller at a time: source, target (history only with a dictionary or a resumed stream)
	const long live = g_live[0] + g_live[1], streams = g_streams[0] + g_streams[1];
	if (live > 6 || streams > 2 || g_allocs[0] > 4 || g_allocs[1] > 4) { printf("leak: %ld live buffers, %ld streams, %ld + %ld allocations\n", live, streams, (long)g_allocs[0], (long)g_allocs[1]); return 6; }
	printf("ok %ld %ld\n", live, streams);
	return 0;
}
