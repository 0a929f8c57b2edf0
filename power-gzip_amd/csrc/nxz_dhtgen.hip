// nxz_dhtgen.hip -- dynamic-Huffman table generation on the device (gfx950, wave64).
//
// The reference builds the table of a dynamic-Huffman block on the host: dhtgen()
// (/root/reference lib/nx_dhtgen.c:945-1034) turns the 286 + 30 LZ symbol counts that a
// COMPRESS_*_COUNT job returned into code lengths and the RFC 1951 3.2.7 header, and the next
// job is encoded with it.  Here the same function runs on the GPU, one WAVEFRONT per table,
// so that a batch gets an exact table per block without a host round trip:
//
//   length_limit :295      counts scaled so that their sum is <= limit (2^14, x3/4 per retry)
//   ordering :323-348      rank sort of the used symbols by (count, symbol): every lane counts
//                          the keys below its own (keys are broadcast 16 bytes at a time)
//   two-queue merge :418   one lane; leaf preferred on ties (:484); a node remembers nothing
//                          but its weight, the children remember their parent
//   depth :360-394         pointer jumping over the parent links: 5 rounds cover depth < 32,
//                          deeper trees are retried anyway (limit schedule :576-595)
//   canonical codes        RFC 1951 3.2.2; a symbol's code = first code of its length + the
//                          number of lower symbols of that length (ballots)
//   header :610-915        fixed code-length code (:628-648); the run-length coder is position
//                          parallel: from the start and the end of its run every position of the
//                          length array knows which symbol -- if any -- the reference's state
//                          machine (:758-910) emits there
//
// The result must equal nxz_dhtgen() (csrc/nxz_dht.cpp) / the reference bit for bit:
// tests/test_gpu_dhtgen.py against tests/golden/dhtgen_vectors.json and the host generator;
// tests/dht_model.py restates this file's formulation for the CPU suite.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nxz_device.h"

namespace nxzd {

#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

constexpr uint32_t NLL = 286, ND = 30, NTOT = 316;
constexpr uint32_t INF16 = 0xFFFFu;

// the reference's fixed code-length code (lib/nx_dhtgen.c:628-648): lengths, canonical codes
// (bit-reversed for LSB-first output) and the 71 constant header bits HLIT=29 HDIST=29 HCLEN=15
// followed by the 19 code-length-code lengths in RFC order
struct ClTab { uint8_t len[19]; uint16_t code[19]; uint32_t hdr[3]; };
constexpr ClTab make_cl()
{
	ClTab t{};
	const uint8_t len[19] = { 5, 7, 6, 5, 5, 4, 4, 3, 3, 3, 3, 4, 5, 5, 4, 7, 6, 5, 6 };
	const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
	uint32_t cnt[8] = {}, next[8] = {}, c = 0;
	for (int i = 0; i < 19; i++) { t.len[i] = len[i]; cnt[len[i]]++; }
	for (int b = 1; b <= 7; b++) { c = (c + cnt[b - 1]) << 1; next[b] = c; }
	for (int i = 0; i < 19; i++) {
		uint32_t v = next[len[i]]++, r = 0;
		for (int k = 0; k < len[i]; k++) r |= ((v >> k) & 1u) << (len[i] - 1 - k);
		t.code[i] = (uint16_t)r;
	}
	int n = 0;
	const uint32_t fld[3][2] = { { 286 - 257, 5 }, { 30 - 1, 5 }, { 19 - 4, 4 } };
	for (int i = 0; i < 3 + 19; i++) {
		const uint32_t v = i < 3 ? fld[i][0] : len[order[i - 3]], nb = i < 3 ? fld[i][1] : 3;
		const uint64_t sh = (uint64_t)v << (n & 31);
		t.hdr[n >> 5] |= (uint32_t)sh;
		if (sh >> 32) t.hdr[(n >> 5) + 1] |= (uint32_t)(sh >> 32);
		n += (int)nb;
	}
	return t;
}
__device__ const ClTab CL = make_cl();
constexpr uint32_t HDR_CONST_BITS = 14 + 19 * 3;        // 71

// LDS image of one wavefront (bytes)
constexpr uint32_t O_CNT  = 0;                 // u32[320]  sort keys (count << 9 | symbol; unused symbols all ones)
constexpr uint32_t O_L    = O_CNT + 320 * 4;   // u16[288+8] sorted leaf weights (INF padded)
constexpr uint32_t O_S    = O_L + 296 * 2;     // u16[288]  sorted leaf symbols
constexpr uint32_t O_N    = O_S + 288 * 2;     // u16[288]  node weights, creation order
constexpr uint32_t O_LP   = O_N + 288 * 2;     // u16[288]  parent node of sorted leaf x
constexpr uint32_t O_DP   = O_LP + 288 * 2;    // u32[288]  per node: depth below the root << 16 | ancestor
constexpr uint32_t O_LEN  = O_DP + 288 * 4;    // u8[320]   code lengths LL(286) D(30), sentinel padded
constexpr uint32_t O_HDR  = O_LEN + 320;       // u32[76]   header bit string
constexpr uint32_t WAVE_LDS = O_HDR + 76 * 4;  // 5480
static_assert(O_L % 16 == 0 && O_DP % 4 == 0 && O_HDR % 4 == 0, "alignment");

__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
	for (int o = 32; o > 0; o >>= 1) { const uint32_t u = __shfl_xor(v, o, 64); v = u > v ? u : v; }
	return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}

// Code lengths of one alphabet (nsym <= 286 counts at hist, modified in place as the reference
// does) into lens[0..nsym).  ROWS = ceil(nsym / 64).
template <int ROWS>
__device__ void code_lengths(uint8_t *w, const uint32_t NXZ_GLOBAL_AS *hist, uint32_t nsym, uint8_t *lens, int lane)
{
	uint32_t *cnt = (uint32_t *)(w + O_CNT);
	uint16_t *L = (uint16_t *)(w + O_L), *S = (uint16_t *)(w + O_S), *N = (uint16_t *)(w + O_N), *LP = (uint16_t *)(w + O_LP);
	uint32_t *DP = (uint32_t *)(w + O_DP);
	uint32_t c[ROWS];
#pragma unroll
	for (int r = 0; r < ROWS; r++) {
		const uint32_t i = lane + 64 * r;
		c[r] = i < nsym ? hist[i] : 0;
		if (i < nsym) lens[i] = 0;
	}
	uint32_t limit = 1u << 14;
	for (;;) {
		// ---- length_limit: sum <= limit ----
		uint32_t s32 = 0;
#pragma unroll
		for (int r = 0; r < ROWS; r++) s32 += c[r] >> 8;           // counts are 24-bit saturated (UM 5.1.1): two 32-bit sums
		uint32_t slo = 0;
#pragma unroll
		for (int r = 0; r < ROWS; r++) slo += c[r] & 0xff;
		const uint64_t s = ((uint64_t)wave_sum(s32) << 8) + wave_sum(slo);
		const uint32_t div = (uint32_t)((s + limit - 1) / limit);
		if (div > 1) {
#pragma unroll
			for (int r = 0; r < ROWS; r++) c[r] = (c[r] + div - 1) / div;
		}
		limit = limit * 3 / 4;
		// ---- rank sort by (count, symbol); unused symbols sort last and are not ranked ----
		uint32_t key[ROWS], rank[ROWS], n = 0;
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			const uint32_t i = lane + 64 * r;
			key[r] = c[r] ? (c[r] << 9) | i : 0xFFFFFFFFu;
			cnt[i] = key[r];
			rank[r] = 0;
			n += (uint32_t)__popcll(__ballot(c[r] != 0));
		}
		WSYNC();
		const uint32_t nquad = (nsym + 3) >> 2;
		for (uint32_t q = 0; q < nquad; q++) {
			const uint4 k4 = ((const uint4 *)cnt)[q];                // same address in every lane: broadcast
#pragma unroll
			for (int r = 0; r < ROWS; r++)
				rank[r] += (uint32_t)(k4.x < key[r]) + (uint32_t)(k4.y < key[r]) + (uint32_t)(k4.z < key[r]) + (uint32_t)(k4.w < key[r]);
		}
		WSYNC();
#pragma unroll
		for (int r = 0; r < ROWS; r++)
			if (c[r]) { L[rank[r]] = (uint16_t)c[r]; S[rank[r]] = (uint16_t)(lane + 64 * r); }
		if (lane < 8) L[n + lane] = (uint16_t)INF16;
		WSYNC();
		if (n == 0) return;
		if (n == 1) {
			// the reference is undefined here; like the host generator: one bit
			if (lane == 0) lens[S[0]] = 1;
			WSYNC();
			return;
		}
		// ---- two-queue merge (one lane) ----
		if (lane == 0) {
			uint32_t li = 0, ni = 0, nn = 0;
			uint32_t lf = L[0], nf = INF16;
			const uint32_t steps = n - 1;
			for (uint32_t k = 0; k < steps; k++) {
				uint32_t wsum = 0;
#pragma unroll
				for (int h = 0; h < 2; h++) {
					if (lf <= nf) {                               // leaf preferred on ties; INF16 never wins against a real item
						wsum += lf; LP[li] = (uint16_t)k; li++;
						lf = L[li];                               // INF padded
					} else {
						wsum += nf; DP[ni] = (1u << 16) | k; ni++;
						nf = ni < nn ? N[ni] : INF16;
					}
				}
				N[nn] = (uint16_t)wsum;
				if (ni == nn) nf = wsum;                          // the new node is the only one waiting
				nn++;
			}
			DP[n - 2] = n - 2;                                     // the root: depth 0, its own ancestor
		}
		WSYNC();
		// ---- depths by pointer jumping ----
		const uint32_t nnodes = n - 1, root = n - 2;
		uint32_t dp[ROWS];
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			const uint32_t k = lane + 64 * r;
			dp[r] = k < nnodes ? DP[k] : root;
		}
		for (int round = 0; round < 5; round++) {
			uint32_t up[ROWS];
#pragma unroll
			for (int r = 0; r < ROWS; r++) up[r] = DP[dp[r] & 0xffff];
			WSYNC();
#pragma unroll
			for (int r = 0; r < ROWS; r++) {
				const uint32_t k = lane + 64 * r;
				dp[r] = ((dp[r] & 0xffff0000u) + (up[r] & 0xffff0000u)) | (up[r] & 0xffff);
				if (k < nnodes) DP[k] = dp[r];
			}
			WSYNC();
		}
		bool bad = false;
		uint32_t maxd = 0;
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			const uint32_t k = lane + 64 * r;
			bad |= k < nnodes && (dp[r] & 0xffff) != root;
		}
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			const uint32_t x = lane + 64 * r;
			if (x < n) {
				const uint32_t dl = (DP[LP[x]] >> 16) + 1;
				maxd = dl > maxd ? dl : maxd;
				lens[S[x]] = (uint8_t)(dl > 255 ? 255 : dl);
			}
		}
		maxd = wave_max(maxd);
		WSYNC();
		if (!__ballot(bad) && maxd <= 15) return;
		// too deep: next limit (the scaled counts are scaled again, as the reference does)
	}
}

// canonical codes of lens[0..nsym) -> out[i] = bit-reversed code | length << 16 (i < nout)
template <int ROWS>
__device__ void canon_codes(const uint8_t *lens, uint32_t nsym, uint32_t NXZ_GLOBAL_AS *out, uint32_t nout, int lane)
{
	uint32_t len[ROWS], code[ROWS];
#pragma unroll
	for (int r = 0; r < ROWS; r++) {
		const uint32_t i = lane + 64 * r;
		len[r] = i < nsym ? lens[i] : 0;
		code[r] = 0;
	}
	uint32_t next = 0, prevcnt = 0;
	const unsigned long long below = (1ull << lane) - 1;
	for (uint32_t b = 1; b <= 15; b++) {
		next = (next + prevcnt) << 1;
		uint32_t before = 0;
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			const unsigned long long m = __ballot(len[r] == b);
			if (len[r] == b) code[r] = __builtin_bitreverse32(next + before + (uint32_t)__popcll(m & below)) >> (32 - b);
			before += (uint32_t)__popcll(m);
		}
		prevcnt = before;
	}
#pragma unroll
	for (int r = 0; r < ROWS; r++) {
		const uint32_t i = lane + 64 * r;
		if (i < nout) out[i] = code[r] | (len[r] << 16);
	}
}

// One wavefront per table: counts[316] (LL then D; EOB forced to 1) -> prepared table and/or
// the caller-visible bit string.
__global__ __launch_bounds__(64) void dhtgen_kernel(const uint32_t *__restrict__ counts_, uint32_t n,
						    nxz_dht_prepared_t *__restrict__ prepared_, nxz_batch_dht_t *__restrict__ tables_)
{
	__shared__ __attribute__((aligned(16))) uint8_t w[WAVE_LDS];
	const int lane = threadIdx.x;
	const uint32_t bid = blockIdx.x;
	if (bid >= n) return;
	const uint32_t NXZ_GLOBAL_AS *hist = (const uint32_t NXZ_GLOBAL_AS *)counts_ + (size_t)bid * NTOT;
	uint8_t *lens = w + O_LEN;
	uint32_t *hdr = (uint32_t *)(w + O_HDR);

	// the reference's callers count EOB once (lib/nx_dht.c:189-195); made sure of here in LDS-free form:
	// hist[256] is rewritten by the kernels that count, a caller's array is taken as it is
	code_lengths<5>(w, hist, NLL, lens, lane);
	code_lengths<1>(w, hist + NLL, ND, lens + NLL, lane);
	if (lane < 4) lens[NTOT + lane] = (uint8_t)(0xF0 + lane);      // sentinels: never equal to a length or to each other
	for (int i = lane; i < 76; i += 64) hdr[i] = i < 3 ? CL.hdr[i] : 0;
	WSYNC();

	// ---- run-length coder, position i = 5 * lane + r ----
	uint32_t v[5], f[5], st[5], en[5];
	const uint32_t i0 = 5 * lane;
	const uint32_t vprev = i0 ? lens[i0 - 1] : 0x100;
	uint32_t cmax = 0, cmin = 0xffff;
#pragma unroll
	for (int r = 0; r < 5; r++) {
		v[r] = lens[i0 + r];
		f[r] = v[r] != (r ? v[r - 1] : vprev);
		if (f[r]) { cmax = i0 + r; if (cmin == 0xffff) cmin = i0 + r; }
	}
	// last run start at or before my chunk's first position, from the lanes below
	uint32_t pm = cmax;
	for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(pm, o, 64); if (lane >= o && u > pm) pm = u; }
	uint32_t before = __shfl_up(pm, 1, 64);
	if (lane == 0) before = 0;
	// first run start behind my chunk, from the lanes above
	uint32_t sm = cmin;
	for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_down(sm, o, 64); if (lane + o < 64 && u < sm) sm = u; }
	uint32_t after = __shfl_down(sm, 1, 64);
	if (lane == 63) after = 320;
#pragma unroll
	for (int r = 0; r < 5; r++) st[r] = f[r] ? i0 + r : (r ? st[r - 1] : before);
#pragma unroll
	for (int r = 4; r >= 0; r--) en[r] = r == 4 ? after : (f[r + 1] ? i0 + r + 1 : en[r + 1]);

	uint32_t val[5], nb[5], tot = 0;
#pragma unroll
	for (int r = 0; r < 5; r++) {
		const uint32_t i = i0 + r, k = i - st[r], R = en[r] - st[r];
		uint32_t sym = 0xff, xv = 0, xb = 0;
		if (i < NTOT) {
			if (v[r]) {
				if (k == 0) sym = v[r];
				else {
					const uint32_t kk = k - 1, g0 = kk - kk % 6;
					const uint32_t g = R - 1 - g0 < 6 ? R - 1 - g0 : 6;
					if (g >= 3) { if (kk == g0) { sym = 16; xv = g - 3; xb = 2; } }
					else sym = v[r];
				}
			} else {
				const uint32_t g0 = k - k % 138;
				const uint32_t g = R - g0 < 138 ? R - g0 : 138;
				if (g >= 11) { if (k == g0) { sym = 18; xv = g - 11; xb = 7; } }
				else if (g >= 3) { if (k == g0) { sym = 17; xv = g - 3; xb = 3; } }
				else sym = 0;
			}
		}
		if (sym != 0xff) {
			const uint32_t cl = CL.len[sym];
			val[r] = CL.code[sym] | (xv << cl);
			nb[r] = cl + xb;
		} else { val[r] = 0; nb[r] = 0; }
		tot += nb[r];
	}
	uint32_t incl = tot;
	for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
	uint32_t pos = HDR_CONST_BITS + incl - tot;
#pragma unroll
	for (int r = 0; r < 5; r++) {
		if (nb[r]) {
			const uint64_t sh = (uint64_t)val[r] << (pos & 31);
			atomicOr(&hdr[pos >> 5], (uint32_t)sh);
			if (sh >> 32) atomicOr(&hdr[(pos >> 5) + 1], (uint32_t)(sh >> 32));
		}
		pos += nb[r];
	}
	const uint32_t dhtlen = HDR_CONST_BITS + __shfl(incl, 63, 64);
	WSYNC();

	if (prepared_) {
		nxz_dht_prepared_t NXZ_GLOBAL_AS *p = (nxz_dht_prepared_t NXZ_GLOBAL_AS *)prepared_ + bid;
		if (lane == 0) { p->dhtlen = dhtlen; p->status = 0; }
		for (int i = lane; i < 74; i += 64) p->dhtw[i] = hdr[i];
		canon_codes<5>(lens, NLL, (uint32_t NXZ_GLOBAL_AS *)p->ll, 288, lane);
		canon_codes<1>(lens + NLL, ND, (uint32_t NXZ_GLOBAL_AS *)p->d, 32, lane);
	}
	if (tables_) {
		nxz_batch_dht_t NXZ_GLOBAL_AS *tb = (nxz_batch_dht_t NXZ_GLOBAL_AS *)tables_ + bid;
		if (lane == 0) tb->dhtlen = dhtlen;
		// 292 bytes at offset 4: written as dwords
		uint32_t NXZ_GLOBAL_AS *o = (uint32_t NXZ_GLOBAL_AS *)tb->dht;
		for (int i = lane; i < 73; i += 64) o[i] = hdr[i];
	}
}

} // namespace nxzd

extern "C" int nxz_launch_dhtgen(const uint32_t *counts, size_t n, nxz_dht_prepared_t *prepared,
				 nxz_batch_dht_t *tables, hipStream_t stream)
{
	if (n == 0) return 0;
	hipLaunchKernelGGL(nxzd::dhtgen_kernel, dim3((unsigned)n), dim3(64), 0, stream, counts, (uint32_t)n, prepared, tables);
	return (int)hipGetLastError();
}
