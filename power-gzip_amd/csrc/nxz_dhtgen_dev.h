// nxz_dhtgen_dev.h -- the device side of the dynamic-Huffman table generator (nxz_dhtgen.hip) as functions of ONE wavefront,
// for the kernel of that file and for the LZ77 kernel's fused form (nxz_lz77.hip: a wavefront builds the table of the block
// just parsed while the others encode the block before it).  `Bar`: called at the loops' turning points -- nothing in the
// kernel of nxz_dhtgen.hip; in the fused form the wavefront's share of the workgroup barriers the encoding waves go through
// (all sixteen waves must pass a barrier, whatever they are busy with).
#ifndef NXZ_DHTGEN_DEV_H
#define NXZ_DHTGEN_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nxz_device.h"

namespace nxzd {

#define NXZD_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

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
template <int ROWS, typename H, typename Bar>
__device__ __forceinline__ void code_lengths(uint8_t *w, H hist, uint32_t nsym, uint8_t *lens, int lane, Bar &bar)
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
		NXZD_WSYNC();
		const uint32_t nquad = (nsym + 3) >> 2;
		for (uint32_t q = 0; q < nquad; q++) {
			if ((q & 15) == 15) bar();
			const uint4 k4 = ((const uint4 *)cnt)[q];                // same address in every lane: broadcast
#pragma unroll
			for (int r = 0; r < ROWS; r++)
				rank[r] += (uint32_t)(k4.x < key[r]) + (uint32_t)(k4.y < key[r]) + (uint32_t)(k4.z < key[r]) + (uint32_t)(k4.w < key[r]);
		}
		NXZD_WSYNC();
#pragma unroll
		for (int r = 0; r < ROWS; r++)
			if (c[r]) { L[rank[r]] = (uint16_t)c[r]; S[rank[r]] = (uint16_t)(lane + 64 * r); }
		if (lane < 8) L[n + lane] = (uint16_t)INF16;
		NXZD_WSYNC();
		if (n == 0) return;
		if (n == 1) {
			// the reference is undefined here; like the host generator: one bit
			if (lane == 0) lens[S[0]] = 1;
			NXZD_WSYNC();
			return;
		}
		// ---- two-queue merge: one pick after the other, each depending on the last ----
		// (one lane in the kernel of nxz_dhtgen.hip; in the fused form of nxz_lz77.hip the whole wavefront runs the loop in
		// step, every lane the same picks: its share of the workgroup's barriers falls inside the loop, and a barrier is the
		// whole wavefront's.  Measured and not kept: the queues' heads out of registers instead of LDS -- v_readlane from a
		// 64-entry window a queue -- made the kernel of nxz_dhtgen.hip 45 % slower and the fused form no faster.)
		if (Bar::none ? lane == 0 : true) {
			uint32_t li = 0, ni = 0, nn = 0;
			uint32_t lf = L[0], nf = INF16;
			const uint32_t steps = n - 1;
			for (uint32_t k = 0; k < steps; k++) {
				if (!Bar::none && (k & 15) == 15) bar();
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
		NXZD_WSYNC();
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
			NXZD_WSYNC();
#pragma unroll
			for (int r = 0; r < ROWS; r++) {
				const uint32_t k = lane + 64 * r;
				dp[r] = ((dp[r] & 0xffff0000u) + (up[r] & 0xffff0000u)) | (up[r] & 0xffff);
				if (k < nnodes) DP[k] = dp[r];
			}
			NXZD_WSYNC();
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
		NXZD_WSYNC();
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

// One wavefront, one table: counts hist[316] (LL then D; EOB counted by whoever counted) -> the prepared table and / or the
// caller-visible bit string (either may be NULL).  w: WAVE_LDS bytes of LDS of this wavefront's own.
// (H: where the counts are -- device memory, or the LDS histogram of the kernel that counted)
template <typename H, typename Bar>
__device__ __forceinline__ void dhtgen_wave(uint8_t *w, H hist, nxz_dht_prepared_t *prepared_, nxz_batch_dht_t *tables_, int lane, Bar &bar)
{
	uint8_t *lens = w + O_LEN;
	uint32_t *hdr = (uint32_t *)(w + O_HDR);

	// the reference's callers count EOB once (lib/nx_dht.c:189-195); made sure of here in LDS-free form:
	// hist[256] is rewritten by the kernels that count, a caller's array is taken as it is
	code_lengths<5>(w, hist, NLL, lens, lane, bar);
	code_lengths<1>(w, hist + NLL, ND, lens + NLL, lane, bar);
	if (lane < 4) lens[NTOT + lane] = (uint8_t)(0xF0 + lane);      // sentinels: never equal to a length or to each other
	for (int i = lane; i < 76; i += 64) hdr[i] = i < 3 ? CL.hdr[i] : 0;
	NXZD_WSYNC();

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
	NXZD_WSYNC();

	bar();
	if (prepared_) {
		nxz_dht_prepared_t NXZ_GLOBAL_AS *p = (nxz_dht_prepared_t NXZ_GLOBAL_AS *)prepared_;
		if (lane == 0) { p->dhtlen = dhtlen; p->status = 0; }
		for (int i = lane; i < 74; i += 64) p->dhtw[i] = hdr[i];
		canon_codes<5>(lens, NLL, (uint32_t NXZ_GLOBAL_AS *)p->ll, 288, lane);
		canon_codes<1>(lens + NLL, ND, (uint32_t NXZ_GLOBAL_AS *)p->d, 32, lane);
	}
	if (tables_) {
		nxz_batch_dht_t NXZ_GLOBAL_AS *tb = (nxz_batch_dht_t NXZ_GLOBAL_AS *)tables_;
		if (lane == 0) tb->dhtlen = dhtlen;
		// 292 bytes at offset 4: written as dwords
		uint32_t NXZ_GLOBAL_AS *o = (uint32_t NXZ_GLOBAL_AS *)tb->dht;
		for (int i = lane; i < 73; i += 64) o[i] = hdr[i];
	}
}

} // namespace nxzd
#endif
