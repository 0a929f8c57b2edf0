// nxz_misc.hip -- small kernels of the engine: DHT preparation and WRAP (copy + checksums).
//
// dht_prepare: parses the caller's dynamic-Huffman header (the bit string the
//   reference keeps in cpb.in_dht, inc_nx/nxu.h:390-393; format RFC1951 3.2.7)
//   into canonical codes.  One lane per table: a table is <= 2283 bits and is
//   shared by many jobs, so this is off the critical path.  Same decisions as
//   oracle/nxz_huff.c nxo_dht_parse (what lib/nx_dht_decomp.c:255-653 models).
// wrap: GZIP_FC_WRAP (lib/nx_deflate.c:1774, lib/nx_zlib.c:1398-1443): copy
//   source to target and return CRC-32/Adler-32 from the initial values.
#include <hip/hip_runtime.h>
#include "nxz_device.h"

namespace nxz {

struct BitR {
	const uint8_t *p; uint32_t nbits, pos;
	__device__ int get(int n) {
		if (pos + n > nbits) return -1;
		int v = 0;
		for (int i = 0; i < n; i++, pos++) v |= ((p[pos >> 3] >> (pos & 7)) & 1) << i;
		return v;
	}
};

__device__ bool canon(const uint8_t *len, uint16_t *code, int n)
{
	uint32_t cnt[16], next[16], c = 0, kraft = 0;
	for (int b = 0; b < 16; b++) cnt[b] = 0;
	for (int i = 0; i < n; i++) cnt[len[i]]++;
	cnt[0] = 0;
	for (int b = 1; b <= 15; b++) {
		c = (c + cnt[b - 1]) << 1;
		next[b] = c;
		kraft += cnt[b] << (15 - b);
	}
	for (int i = 0; i < n; i++) {
		uint32_t l = len[i];
		code[i] = l ? (uint16_t)(__builtin_bitreverse32(next[l]++) >> (32 - l)) : 0;
	}
	return kraft <= (1u << 15);
}

__global__ void dht_prepare_kernel(const nxz_batch_dht_t *__restrict__ in, size_t n, nxz_dht_prepared_t *__restrict__ out)
{
	size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= n) return;
	const nxz_batch_dht_t *t = &in[idx];
	nxz_dht_prepared_t *o = &out[idx];
	const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
	uint8_t cl_len[19], lens[288 + 32];
	uint16_t cl_code[19], codes[288];
	uint32_t dhtlen = t->dhtlen;
	bool ok = dhtlen <= NXZ_DHT_MAXSZ * 8;
	BitR b{t->dht, ok ? dhtlen : 0, 0};
	int hlit = 0, hdist = 0, hclen = 0;

	o->dhtlen = dhtlen;
	for (int i = 0; i < 74; i++) {
		uint32_t w = 0;
		for (int k = 0; k < 4; k++) {
			uint32_t byte = (uint32_t)i * 4 + k;
			uint32_t v = (ok && byte * 8 < dhtlen) ? t->dht[byte] : 0;
			if (ok && byte * 8 + 8 > dhtlen && byte * 8 < dhtlen) v &= (1u << (dhtlen - byte * 8)) - 1;
			w |= v << (8 * k);
		}
		o->dhtw[i] = w;
	}
	for (int i = 0; i < 19; i++) cl_len[i] = 0;
	for (int i = 0; i < 320; i++) lens[i] = 0;
	if (ok) {
		hlit = b.get(5); hdist = b.get(5); hclen = b.get(4);
		ok = hlit >= 0 && hdist >= 0 && hclen >= 0;
		hlit += 257; hdist += 1; hclen += 4;
		ok = ok && hlit <= 286 && hdist <= 30;
	}
	for (int i = 0; ok && i < hclen; i++) {
		int v = b.get(3);
		if (v < 0) ok = false; else cl_len[order[i]] = (uint8_t)v;
	}
	ok = ok && canon(cl_len, cl_code, 19);
	int cnt = 0, prev = 0;
	while (ok && cnt < hlit + hdist) {
		int sym = -1, code = 0;
		for (int len = 1; len <= 7 && sym < 0 && ok; len++) {
			int bit = b.get(1);
			if (bit < 0) { ok = false; break; }
			code |= bit << (len - 1);
			for (int i = 0; i < 19; i++)
				if (cl_len[i] == len && cl_code[i] == code) { sym = i; break; }
		}
		if (!ok || sym < 0) { ok = false; break; }
		if (sym < 16) { lens[cnt++] = (uint8_t)sym; prev = sym; }
		else {
			int rep, val = 0;
			if (sym == 16) { if (cnt == 0) { ok = false; break; } rep = b.get(2); if (rep < 0) { ok = false; break; } rep += 3; val = prev; }
			else if (sym == 17) { rep = b.get(3); if (rep < 0) { ok = false; break; } rep += 3; }
			else { rep = b.get(7); if (rep < 0) { ok = false; break; } rep += 11; }
			if (cnt + rep > hlit + hdist) { ok = false; break; }
			while (rep--) lens[cnt++] = (uint8_t)val;
			if (sym != 16) prev = 0;
		}
	}
	ok = ok && b.pos == dhtlen;
	uint8_t ll_len[288], d_len[32];
	for (int i = 0; i < 288; i++) ll_len[i] = (ok && i < hlit) ? lens[i] : 0;
	for (int i = 0; i < 32; i++) d_len[i] = (ok && i < hdist) ? lens[hlit + i] : 0;
	ok = canon(ll_len, codes, 288) && ok;
	for (int i = 0; i < 288; i++) o->ll[i] = codes[i] | ((uint32_t)ll_len[i] << 16);
	ok = canon(d_len, codes, 32) && ok;
	for (int i = 0; i < 32; i++) o->d[i] = codes[i] | ((uint32_t)d_len[i] << 16);
	o->status = ok ? 0 : NXZ_CC_INVALID_DHT;
}

// ---- wrap: one 256-thread workgroup per job ----
__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b)
{
	uint32_t r = 0;
	for (int i = 0; i < 32; i++) {
		r ^= (b & 0x80000000u) ? a : 0;
		a = (a >> 1) ^ ((a & 1) ? 0xedb88320u : 0);
		b <<= 1;
	}
	return r;
}

__global__ __launch_bounds__(256) void wrap_kernel(const nxz_batch_job_t *__restrict__ jobs, nxz_batch_result_t *__restrict__ results)
{
	__shared__ uint32_t tab[256];
	__shared__ uint32_t red[3][256];
	const int t = threadIdx.x;
	const nxz_batch_job_t job = jobs[blockIdx.x];
	const uint32_t n = job.src_len;
	{
		uint32_t c = t;
		for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1) ? 0xedb88320u : 0);
		tab[t] = c;
	}
	__syncthreads();
	if (n > job.dst_cap) {
		if (t == 0) { nxz_batch_result_t r = {NXZ_CC_TARGET_SPACE, 0, 0, 0, 0, 0, 0, 0}; results[blockIdx.x] = r; }
		return;
	}
	// contiguous slice per thread, slice length a multiple of 16 so that copies stay vectorised
	uint32_t per = ((n + 255) / 256 + 15) & ~15u;
	uint32_t lo = (uint32_t)t * per, hi = lo + per < n ? lo + per : n;
	uint32_t crc = 0, s1 = 0, s2 = 0;
	if (lo < n) {
		const uint8_t *s = job.src; uint8_t *d = job.dst;
		uint32_t i = lo;
		for (; i + 16 <= hi; i += 16) {
			uint4 v = *(const uint4 *)(s + i);
			*(uint4 *)(d + i) = v;
			uint32_t w[4] = {v.x, v.y, v.z, v.w};
			for (int k = 0; k < 16; k++) {
				uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
				uint32_t bx = b ^ ((i + k) < 4 ? 0xffu : 0);            // initial value 0 -> ~0 folded into bytes 0..3
				crc = tab[(crc ^ bx) & 0xff] ^ (crc >> 8);
				s1 += b; s2 += s1;
			}
			if ((i & 0xfff) == 0xff0) { s1 %= 65521u; s2 %= 65521u; }
		}
		for (; i < hi; i++) {
			uint32_t b = s[i]; d[i] = (uint8_t)b;
			uint32_t bx = b ^ (i < 4 ? 0xffu : 0);
			crc = tab[(crc ^ bx) & 0xff] ^ (crc >> 8);
			s1 += b; s2 += s1;
		}
		s1 %= 65521u; s2 %= 65521u;
	}
	red[0][t] = crc; red[1][t] = s1; red[2][t] = s2;
	__syncthreads();
	if (t == 0) {
		// sequential combine of 256 slices (tiny): crc = crc*x^(8*len) ^ next; adler combine
		uint32_t c = 0, a1 = 1, a2 = 0;
		uint32_t xp = 0x80000000u;   // x^(8*per), by square-and-multiply on x^8
		{ uint32_t sq = 0x00800000u, e = per; while (e) { if (e & 1) xp = gf_mul(xp, sq); sq = gf_mul(sq, sq); e >>= 1; } }
		for (uint32_t k = 0; k < 256; k++) {
			uint32_t klo = k * per; if (klo >= n) break;
			uint32_t len = klo + per < n ? per : n - klo;
			uint32_t m = xp;
			if (len != per) { m = 0x80000000u; uint32_t sq = 0x00800000u, e = len; while (e) { if (e & 1) m = gf_mul(m, sq); sq = gf_mul(sq, sq); e >>= 1; } }
			c = gf_mul(c, m) ^ red[0][k];
			// adler: s1' = s1 + S1k ; s2' = s2 + len*s1 + S2k   (S* computed from a zero start)
			a2 = (uint32_t)((a2 + (uint64_t)len * a1 + red[2][k]) % 65521u);
			a1 = (a1 + red[1][k]) % 65521u;
		}
		if (n < 4) { c = 0xffffffffu; for (uint32_t i = 0; i < n; i++) c = tab[(c ^ job.src[i]) & 0xff] ^ (c >> 8); }
		nxz_batch_result_t r;
		r.cc = 0; r.tpbc = n; r.tebc = 0; r.spbc = n; r.crc = c ^ 0xffffffffu; r.adler = (a2 << 16) | a1; r.subc = 0; r.sfbt = 0;
		results[blockIdx.x] = r;
	}
}

} // namespace nxz

extern "C" int nxz_launch_dht_prepare(const nxz_batch_dht_t *dht, size_t n, nxz_dht_prepared_t *out, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxz::dht_prepare_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, dht, n, out);
	return (int)hipGetLastError();
}

extern "C" int nxz_launch_wrap(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxz::wrap_kernel, dim3((unsigned)n), dim3(256), 0, stream, jobs, results);
	return (int)hipGetLastError();
}
