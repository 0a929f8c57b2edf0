// nxz_encode.hip -- entropy stage of the DEFLATE compression engine (gfx950, wave64).
//
// Second half of the COMPRESS function codes (issued at /root/reference lib/nx_deflate.c:1808,1841;
// contract inc_nx/nxu.h:286-616, consumer code lib/nx_deflate.c:969-1078): the LZ77 kernel
// (nxz_lz77.hip) leaves, per job, two position bitmaps (where a literal token starts, where a
// match token starts) and the match tokens' (length, distance) records in parse order; this
// kernel turns them into one deflate block -- fixed code, a caller's dynamic table, or the
// table the device built from this very job's symbol counts (nxz_dhtgen.hip).
//
// Position parallel: a 256-thread workgroup per job walks the block in rounds of 4096
// positions, 16 consecutive positions per lane.  A lane knows its tokens from the two bitmaps,
// reads its literals from the source and its matches from the record array (rank of a match =
// number of match bits in front of it; the ranks of all 2048 bitmap words are made once per job),
// sums the code lengths, gets a bit offset from the workgroup prefix sum and ORs its codes into
// an LDS window that leaves as coalesced dwords.  Nothing here is serial over the block, and
// with ~15 KiB of LDS eight workgroups share a CU, so the latencies of one hide behind the others.
// Bit for bit the encoder of oracle/nxz_lz77.c (put_tokens / nxo_encode_fixed / nxo_encode_dynamic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nxz_device.h"

namespace nxze {

constexpr int NT = 256;
constexpr uint32_t RPOS = 4096;                 // positions per round
constexpr uint32_t LANE_BITS_MAX = 288;         // six 48-bit match tokens in 16 positions
constexpr uint32_t HDR_WORDS = 74;              // 3 + 2283 bits of a dynamic header at most
constexpr uint32_t WWORDS = NT * LANE_BITS_MAX / 32 + HDR_WORDS + 2;   // 2380

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

struct Tok {
	uint32_t lit16, tok16;
	v4u bytes;
	uint32_t rec[6];
};

__device__ __forceinline__ uint32_t byte_at(const v4u &q, uint32_t k)
{
	const uint32_t a = (k & 8) ? q.z : q.x, b = (k & 8) ? q.w : q.y;
	const uint32_t w = (k & 4) ? b : a;
	return (w >> (8 * (k & 3))) & 0xff;
}

// length / distance symbol of a match record (len - 3 | (dist - 1) << 8), RFC 1951 3.2.5
struct MatchSym { uint32_t ls, le, lx, ds, de, dx; };
__device__ __forceinline__ MatchSym match_sym(uint32_t rec)
{
	MatchSym m;
	const uint32_t l3 = rec & 0xff, d = (rec >> 8) & 0x7fff;
	uint32_t le = l3 < 8 ? 0 : (29 - (uint32_t)__builtin_clz(l3 | 8));
	m.ls = l3 == 255 ? 28 : (le << 2) + (l3 >> le);
	if (l3 == 255) le = 0;
	m.le = le;
	m.lx = l3 & ((1u << le) - 1);
	const uint32_t de = d < 4 ? 0 : (30 - (uint32_t)__builtin_clz(d | 4));
	m.ds = d < 4 ? d : 2 * de + 2 + ((d >> de) & 1);
	m.de = de;
	m.dx = d & ((1u << de) - 1);
	return m;
}

template <bool DHT>
__global__ __launch_bounds__(NT) void encode_kernel(const nxz_batch_job_t *__restrict__ jobs_, const uint8_t *__restrict__ tokens_,
						     const nxz_dht_prepared_t *__restrict__ tables_, int table_per_job,
						     nxz_batch_result_t *__restrict__ results_, uint32_t njobs)
{
	__shared__ uint32_t lltab[288];
	__shared__ uint32_t dtab[32];
	__shared__ __attribute__((aligned(16))) uint32_t win[WWORDS + 2];
	__shared__ uint16_t rankpre[2048];
	__shared__ uint32_t wsum[8];
	__shared__ uint32_t errflag;
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const uint32_t bid = blockIdx.x;
	if (bid >= njobs) return;
	const nxz_batch_job_t job = jobs_[bid];
	const uint32_t total = job.src_len;
	const uint32_t h = job.hist_len < total ? job.hist_len : total;
	const uint32_t n = total - h;
	const uint8_t NXZ_GLOBAL_AS *src = (const uint8_t NXZ_GLOBAL_AS *)job.src + h;
	const uint8_t NXZ_GLOBAL_AS *tk = (const uint8_t NXZ_GLOBAL_AS *)tokens_ + (size_t)bid * NXZ_TOK_STRIDE;
	const uint16_t NXZ_GLOBAL_AS *litb = (const uint16_t NXZ_GLOBAL_AS *)(tk + NXZ_TOK_LITBITS);
	const uint16_t NXZ_GLOBAL_AS *tokb = (const uint16_t NXZ_GLOBAL_AS *)(tk + NXZ_TOK_MATCHBITS);
	const uint32_t NXZ_GLOBAL_AS *recs = (const uint32_t NXZ_GLOBAL_AS *)(tk + NXZ_TOK_RECORDS);
	uint32_t NXZ_GLOBAL_AS *dstw = (uint32_t NXZ_GLOBAL_AS *)job.dst;
	const uint32_t cap_words = job.dst_cap >> 2;
	const nxz_dht_prepared_t NXZ_GLOBAL_AS *tb = DHT ? (const nxz_dht_prepared_t NXZ_GLOBAL_AS *)tables_ + (table_per_job ? bid : job.dht_index) : nullptr;

	// ---- tables, window, block header ----
	for (uint32_t i = t; i < WWORDS + 2; i += NT) win[i] = 0;
	if (t == 0) errflag = 0;
	if (DHT) {
		for (int i = t; i < 288; i += NT) lltab[i] = tb->ll[i];
		if (t < 32) dtab[t] = tb->d[t];
	} else {
		for (int i = t; i < 288; i += NT) {
			// RFC1951 3.2.6 fixed code; entry = bit-reversed code | len << 16
			uint32_t len, code;
			if (i < 144) { len = 8; code = 0x30 + i; }
			else if (i < 256) { len = 9; code = 0x190 + (i - 144); }
			else if (i < 280) { len = 7; code = i - 256; }
			else { len = 8; code = 0xC0 + (i - 280); }
			lltab[i] = (__builtin_bitreverse32(code) >> (32 - len)) | (len << 16);
		}
		if (t < 32) dtab[t] = (__builtin_bitreverse32((uint32_t)t) >> 27) | (5u << 16);
	}
	__syncthreads();
	uint32_t base_bits;                                          // bits already in the window (uniform)
	if (DHT) {
		const uint32_t hb = tb->dhtlen + 3;                      // BFINAL = 1 as emitted (the host rewrites it, lib/nx_deflate.c:158), BTYPE = 10
		const uint32_t nw = (hb + 31) >> 5;
		for (uint32_t i = t; i < nw && i < HDR_WORDS; i += NT) {
			const uint32_t cur = i < 74 ? tb->dhtw[i] : 0, prev = i ? tb->dhtw[i - 1] : 0;
			uint32_t w = (cur << 3) | (i ? prev >> 29 : 5u);
			if (i == (hb >> 5)) w &= (1u << (hb & 31)) - 1;       // the last, partial dword
			win[i] = w;
		}
		base_bits = hb;
	} else {
		if (t == 0) win[0] = 3u;                                 // BFINAL = 1, BTYPE = 01
		base_bits = 3;
	}
	// ---- rank of the first match of every 32 positions (exclusive prefix sum of the match bitmap) ----
	{
		const uint32_t nwords = (n + 31) >> 5;
		const v4u NXZ_GLOBAL_AS *tw = (const v4u NXZ_GLOBAL_AS *)tokb;
		v4u a = { 0, 0, 0, 0 }, b = a;
		if (8u * t < nwords) { a = tw[2 * t]; b = tw[2 * t + 1]; }
		const uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
		uint32_t c[8], s = 0;
#pragma unroll
		for (int k = 0; k < 8; k++) { c[k] = s; s += (uint32_t)__popc(w[k]); }
		uint32_t incl = s;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
		if (lane == 63) wsum[wave] = incl;
		__syncthreads();
		uint32_t off = incl - s;
		for (int k = 0; k < wave; k++) off += wsum[k];
#pragma unroll
		for (int k = 0; k < 8; k++) rankpre[8 * t + k] = (uint16_t)(off + c[k]);
	}
	__syncthreads();

	uint32_t wordbase = 0;
	bool missing = false;
	for (uint32_t r0 = 0; r0 < n; r0 += RPOS) {
		const uint32_t p0 = r0 + 16 * t;
		Tok k{};
		uint32_t nbits = 0;
		if (p0 < n) {
			k.lit16 = litb[p0 >> 4];
			k.tok16 = tokb[p0 >> 4];
			if (p0 + 16 <= n) k.bytes = *(const v4u NXZ_GLOBAL_AS *)(src + p0);
			else {
				// ragged end: nothing is read past the source
				uint64_t lo = 0, hi = 0;
				for (uint32_t i = 0; i < 8; i++) {
					if (p0 + i < n) lo |= (uint64_t)src[p0 + i] << (8 * i);
					if (p0 + 8 + i < n) hi |= (uint64_t)src[p0 + 8 + i] << (8 * i);
				}
				k.bytes = (v4u){ (uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32) };
			}
			uint32_t rank = rankpre[p0 >> 5];
			if (p0 & 16) rank += (uint32_t)__popc(tokb[(p0 >> 4) - 1]);
			const uint32_t nm = (uint32_t)__popc(k.tok16);
#pragma unroll
			for (int j = 0; j < 6; j++) k.rec[j] = (uint32_t)j < nm ? recs[rank + j] : 0;
			// pass 1: code lengths
			uint32_t m = k.lit16;
			while (m) {
				const uint32_t kk = (uint32_t)__builtin_ctz(m);
				m &= m - 1;
				const uint32_t e = lltab[byte_at(k.bytes, kk)];
				missing |= DHT && (e >> 16) == 0;
				nbits += e >> 16;
			}
#pragma unroll
			for (int j = 0; j < 6; j++) {
				if ((uint32_t)j < nm) {
					const MatchSym s = match_sym(k.rec[j]);
					const uint32_t lt = lltab[257 + s.ls], dt = dtab[s.ds];
					missing |= DHT && ((lt >> 16) == 0 || (dt >> 16) == 0);
					nbits += (lt >> 16) + s.le + (dt >> 16) + s.de;
				}
			}
		}
		uint32_t incl = nbits;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
		if (lane == 63) wsum[wave] = incl;
		__syncthreads();
		uint32_t bitpos = base_bits + incl - nbits, roundbits = 0;
#pragma unroll
		for (int w = 0; w < NT / 64; w++) { const uint32_t s = wsum[w]; if (w < wave) bitpos += s; roundbits += s; }
		// pass 2: codes into the window, tokens in position order
		if (nbits) {
			uint64_t acc = 0;
			uint32_t accn = bitpos & 31, accw = bitpos >> 5;
			bool first = true;
			auto put = [&](uint32_t v, uint32_t nb) {
				acc |= (uint64_t)v << accn;
				accn += nb;
				if (accn >= 32) {
					if (first) atomicOr(&win[accw], (uint32_t)acc); else win[accw] = (uint32_t)acc;
					first = false;
					acc >>= 32; accn -= 32; accw++;
				}
			};
			uint32_t m = k.lit16 | k.tok16, j = 0;
			while (m) {
				const uint32_t kk = (uint32_t)__builtin_ctz(m);
				m &= m - 1;
				if ((k.tok16 >> kk) & 1) {
					const uint32_t rec = j == 0 ? k.rec[0] : j == 1 ? k.rec[1] : j == 2 ? k.rec[2] : j == 3 ? k.rec[3] : j == 4 ? k.rec[4] : k.rec[5];
					j++;
					const MatchSym s = match_sym(rec);
					const uint32_t lt = lltab[257 + s.ls], dt = dtab[s.ds];
					put((lt & 0xffff) | (s.lx << (lt >> 16)), (lt >> 16) + s.le);         // <= 20 bits
					put((dt & 0xffff) | (s.dx << (dt >> 16)), (dt >> 16) + s.de);         // <= 28 bits
				} else {
					const uint32_t e = lltab[byte_at(k.bytes, kk)];
					put(e & 0xffff, e >> 16);
				}
			}
			if (accn) atomicOr(&win[accw], (uint32_t)acc);
		}
		__syncthreads();
		const uint32_t tot = base_bits + roundbits, nfull = tot >> 5;
		for (uint32_t i = t; i < nfull; i += NT)
			if (wordbase + i < cap_words) dstw[wordbase + i] = win[i];
		const uint32_t keep = win[nfull];
		__syncthreads();
		for (uint32_t i = t; i <= nfull; i += NT) win[i] = i == 0 ? keep : 0;
		wordbase += nfull;
		base_bits = tot & 31;
		__syncthreads();
	}
	if (missing) errflag = 1;
	__syncthreads();
	// a job without any round (n == 0) still has its header in the window
	if (base_bits >= 32) {
		const uint32_t nfull = base_bits >> 5;
		for (uint32_t i = t; i < nfull; i += NT)
			if (wordbase + i < cap_words) dstw[wordbase + i] = win[i];
		const uint32_t keep = win[nfull];
		__syncthreads();
		if (t == 0) win[0] = keep;
		wordbase += nfull;
		base_bits &= 31;
		__syncthreads();
	}
	// ---- EOB + tail ----
	if (t == 0) {
		const uint32_t lt = lltab[256];
		uint32_t cc = 0;
		if (wordbase > cap_words) cc = NXZ_CC_TARGET_SPACE;
		if (DHT && (errflag || (lt >> 16) == 0)) cc = NXZ_CC_MISSING_CODE;
		if (DHT && tb->status) cc = NXZ_CC_INVALID_DHT;
		const uint64_t acc = (uint64_t)win[0] | ((uint64_t)(lt & 0xffff) << base_bits);
		const uint32_t bits = base_bits + (lt >> 16);
		const uint64_t totbits = (uint64_t)wordbase * 32 + bits;
		const uint32_t tpbc = (uint32_t)((totbits + 7) >> 3);
		if (tpbc > job.dst_cap) cc = cc ? cc : NXZ_CC_TARGET_SPACE;
		if (cc != NXZ_CC_TARGET_SPACE) {
			uint8_t NXZ_GLOBAL_AS *o = (uint8_t NXZ_GLOBAL_AS *)job.dst + (size_t)wordbase * 4;
			for (uint32_t b = 0; b < (bits + 7) / 8; b++) o[b] = (uint8_t)(acc >> (8 * b));
		}
		if (cc == 0 && tpbc > total) cc = NXZ_CC_TPBC_GT_SPBC;
		nxz_batch_result_t *r = results_ + bid;
		r->cc = cc;
		r->tpbc = cc == NXZ_CC_TARGET_SPACE ? 0 : tpbc;
		r->tebc = (uint32_t)(totbits & 7);
		r->sfbt = 0;                                            // (the LZ77 kernel left its match count there)
	}
}

} // namespace nxze

extern "C" int nxz_launch_encode(int dht, int table_per_job, const nxz_batch_job_t *jobs, size_t n, const uint8_t *tokens,
				 const nxz_dht_prepared_t *tables, nxz_batch_result_t *results, hipStream_t stream)
{
	if (n == 0) return 0;
	if (dht) hipLaunchKernelGGL(nxze::encode_kernel<true>, dim3((unsigned)n), dim3(nxze::NT), 0, stream, jobs, tokens, tables, table_per_job, results, (uint32_t)n);
	else hipLaunchKernelGGL(nxze::encode_kernel<false>, dim3((unsigned)n), dim3(nxze::NT), 0, stream, jobs, tokens, tables, 0, results, (uint32_t)n);
	return (int)hipGetLastError();
}
