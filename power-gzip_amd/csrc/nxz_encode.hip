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
#include <stdlib.h>
#include "nxz_device.h"

namespace nxze {

constexpr int NT = 256;
constexpr uint32_t RPOS = 2048;                 // positions per round: 8 per lane, two quads of 4
constexpr uint32_t LANE_BITS_MAX = 144;         // three 48-bit match tokens in 8 positions
constexpr uint32_t HDR_WORDS = 74;              // 3 + 2283 bits of a dynamic header at most
constexpr uint32_t WWORDS = NT * LANE_BITS_MAX / 32 + HDR_WORDS + 2;   // 1228
constexpr uint32_t RECMAX = RPOS / 3 + 6;       // match tokens that can start in one round

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

// The tokens that start in four consecutive positions, as one bit string (<= 96 bits: a match at
// the first position and one at the fourth).  Straight-line code, the same for every lane.
struct Quad { uint32_t a0, a1, a2, nb; };

// CHECK: a symbol without a code can occur (a caller's table; the table the device made of this very block's counts has
// a code for every symbol the block uses -- nxz_dhtgen.hip as lib/nx_dhtgen.c:252-270 -- and the fixed code for all)
template <bool CHECK>
__device__ __forceinline__ Quad encode_quad(const uint32_t *lltab, const uint32_t *dtab, const uint32_t *rec, uint32_t &ri,
					    uint32_t b, uint32_t lit4, uint32_t tok4, bool &missing)
{
	// literals: four look-ups whether or not a literal starts there (a covered position adds no bits)
	const uint32_t e0 = lltab[b & 0xff], e1 = lltab[(b >> 8) & 0xff], e2 = lltab[(b >> 16) & 0xff], e3 = lltab[b >> 24];
	uint64_t v[4] = { e0 & 0xffff, e1 & 0xffff, e2 & 0xffff, e3 & 0xffff };
	uint32_t nbk[4] = { (lit4 & 1) ? e0 >> 16 : 0, (lit4 & 2) ? e1 >> 16 : 0, (lit4 & 4) ? e2 >> 16 : 0, (lit4 & 8) ? e3 >> 16 : 0 };
	if (CHECK) missing |= ((lit4 & 1) && !(e0 >> 16)) || ((lit4 & 2) && !(e1 >> 16)) || ((lit4 & 4) && !(e2 >> 16)) || ((lit4 & 8) && !(e3 >> 16));
	// matches: at most two start in four positions (they are at least three bytes long), the second
	// one only at the fourth position behind one at the first
	if (__ballot(tok4 != 0)) {
		auto one = [&](uint32_t r, uint64_t &mv, uint32_t &mn) {
			const uint32_t l3 = r & 0xff, d = (r >> 8) & 0x7fff;
			uint32_t le = l3 < 8 ? 0 : (29 - (uint32_t)__builtin_clz(l3 | 8));
			const uint32_t ls = l3 == 255 ? 28 : (le << 2) + (l3 >> le);
			if (l3 == 255) le = 0;
			const uint32_t de = d < 4 ? 0 : (30 - (uint32_t)__builtin_clz(d | 4));
			const uint32_t ds = d < 4 ? d : 2 * de + 2 + ((d >> de) & 1);
			const uint32_t lt = lltab[257 + ls], dt = dtab[ds];
			const uint32_t ll = lt >> 16, dl = dt >> 16;
			if (CHECK) missing |= ll == 0 || dl == 0;
			const uint32_t lo = (lt & 0xffff) | ((l3 & ((1u << le) - 1)) << ll);          // <= 20 bits
			const uint32_t hi = (dt & 0xffff) | ((d & ((1u << de) - 1)) << dl);           // <= 28 bits
			mv = (uint64_t)lo | ((uint64_t)hi << (ll + le));
			mn = ll + le + dl + de;
		};
		const uint32_t k1 = (uint32_t)__builtin_ctz(tok4 | 16);                  // 4 = none
		uint64_t mv; uint32_t mn;
		const bool miss0 = missing;
		one(rec[ri], mv, mn);
		if (!tok4) missing = miss0;
		ri += tok4 ? 1 : 0;
#pragma unroll
		for (int k = 0; k < 4; k++) if (k1 == (uint32_t)k) { v[k] = mv; nbk[k] = mn; }
		if (__ballot(tok4 == 9)) {
			const bool miss1 = missing;
			one(rec[ri], mv, mn);
			if (tok4 != 9) missing = miss1;
			else { v[3] = mv; nbk[3] = mn; ri++; }
		}
	}
	// one string: token k at the sum of the lengths before it
	uint64_t lo = nbk[0] ? v[0] : 0, hi = 0;
	uint32_t off = nbk[0];
#pragma unroll
	for (int k = 1; k < 4; k++) {
		const uint64_t x = nbk[k] ? v[k] : 0;
		// off <= 48 + 45 here, x < 2^48
		if (off < 64) { lo |= x << off; hi |= off ? x >> (64 - off) : 0; }
		else hi |= x << (off - 64);
		off += nbk[k];
	}
	Quad q;
	q.a0 = (uint32_t)lo; q.a1 = (uint32_t)(lo >> 32); q.a2 = (uint32_t)hi; q.nb = off;
	return q;
}

// ORs a quad's bits into the window at bit position bitpos
__device__ __forceinline__ void emit_quad(uint32_t *w, const Quad &q, uint32_t bitpos)
{
	if (!q.nb) return;
	const uint32_t sh = bitpos & 31, wi = bitpos >> 5, endw = (sh + q.nb + 31) >> 5;   // dwords touched: 1..4
	const uint64_t s0 = (uint64_t)q.a0 << sh, s1 = (uint64_t)q.a1 << sh, s2 = (uint64_t)q.a2 << sh;
	atomicOr(&w[wi], (uint32_t)s0);
	if (endw > 1) atomicOr(&w[wi + 1], (uint32_t)(s0 >> 32) | (uint32_t)s1);
	if (endw > 2) atomicOr(&w[wi + 2], (uint32_t)(s1 >> 32) | (uint32_t)s2);
	if (endw > 3) atomicOr(&w[wi + 3], (uint32_t)(s2 >> 32));
}

template <bool DHT, bool CHECK = DHT>
__global__ __launch_bounds__(NT) void encode_kernel(const nxz_batch_job_t *__restrict__ jobs_, const uint8_t *__restrict__ tokens_,
						     const nxz_dht_prepared_t *__restrict__ tables_, int table_per_job,
						     nxz_batch_result_t *__restrict__ results_, uint32_t njobs)
{
	__shared__ uint32_t lltab[288];
	__shared__ uint32_t dtab[32];
	__shared__ __attribute__((aligned(16))) uint32_t win[2][WWORDS + 2];   // two windows: one fills while the other leaves
	__shared__ uint32_t recbuf[2][RECMAX + 2];                             // the match records of this round and of the next
	__shared__ uint16_t rankpre[2048 + 2];                                 // matches in front of every 32 positions; [2048] = all
	__shared__ uint32_t wsum[2][NT / 64];
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
	const uint32_t NXZ_GLOBAL_AS *litb = (const uint32_t NXZ_GLOBAL_AS *)(tk + NXZ_TOK_LITBITS);
	const uint32_t NXZ_GLOBAL_AS *tokb = (const uint32_t NXZ_GLOBAL_AS *)(tk + NXZ_TOK_MATCHBITS);
	const uint32_t NXZ_GLOBAL_AS *recs = (const uint32_t NXZ_GLOBAL_AS *)(tk + NXZ_TOK_RECORDS);
	uint32_t NXZ_GLOBAL_AS *dstw = (uint32_t NXZ_GLOBAL_AS *)job.dst;
	const uint32_t cap_words = job.dst_cap >> 2;
	const nxz_dht_prepared_t NXZ_GLOBAL_AS *tb = DHT ? (const nxz_dht_prepared_t NXZ_GLOBAL_AS *)tables_ + (table_per_job ? bid : job.dht_index) : nullptr;
	const uint32_t nwords = (n + 31) >> 5;

	// What a lane needs of a round is asked for a round ahead, so that the loads' latency hides
	// behind the work on the round before: its 8 source bytes, the two bitmap words that hold its
	// 8 positions, and its share of the round's match records (for the LDS copy).
	struct Fetch { v2u bytes; uint32_t litw, tokw, rec[3]; };
	auto fetch = [&](uint32_t r0, Fetch &f) {
		const uint32_t p0 = r0 + 8 * t;
		f.bytes = (v2u){ 0, 0 }; f.litw = 0; f.tokw = 0;
		if (p0 < n) {
			f.litw = litb[p0 >> 5];
			f.tokw = tokb[p0 >> 5];
			if (p0 + 8 <= n) f.bytes = *(const v2u NXZ_GLOBAL_AS *)(src + p0);
			else {
				// ragged end: nothing is read past the source
				uint64_t v = 0;
				for (uint32_t i = 0; i < 8; i++) if (p0 + i < n) v |= (uint64_t)src[p0 + i] << (8 * i);
				f.bytes = (v2u){ (uint32_t)v, (uint32_t)(v >> 32) };
			}
		}
	};
	auto fetch_recs = [&](uint32_t r0, Fetch &f) {                // needs rankpre
		const uint32_t ra = rankpre[r0 >> 5], rb = rankpre[(r0 + RPOS) >> 5 < 2048 ? (r0 + RPOS) >> 5 : 2048];
#pragma unroll
		for (int j = 0; j < 3; j++) f.rec[j] = ra + t + 256 * j < rb ? recs[ra + t + 256 * j] : 0;
	};
	Fetch nx;
	fetch(0, nx);                                                // round 0's bytes travel while the tables are set up

	// ---- tables, windows, ranks ----
	for (uint32_t i = t; i < 2 * (WWORDS + 2); i += NT) (&win[0][0])[i] = 0;
	if (t == 0) errflag = 0;
	{
		const v4u NXZ_GLOBAL_AS *tw = (const v4u NXZ_GLOBAL_AS *)tokb;
		v4u a = { 0, 0, 0, 0 }, b = a;
		if (8u * t < nwords) { a = tw[2 * t]; b = tw[2 * t + 1]; }
		const uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
		uint32_t c[8], s = 0;
#pragma unroll
		for (int k = 0; k < 8; k++) {
			c[k] = s;
			s += 8u * t + k < nwords ? (uint32_t)__popc(w[k]) : 0;  // (what lies behind the block's last word is not the kernel's)
		}
		uint32_t incl = s;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
		if (lane == 63) wsum[0][wave] = incl;
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
		uint32_t off = incl - s;
		for (int k = 0; k < wave; k++) off += wsum[0][k];
#pragma unroll
		for (int k = 0; k < 8; k++) rankpre[8 * t + k] = (uint16_t)(off + c[k]);
		if (t == NT - 1) rankpre[2048] = (uint16_t)(off + s);
	}
	uint32_t base_bits;                                          // bits already in the current window (uniform)
	if (DHT) {
		const uint32_t hb = tb->dhtlen + 3;                      // BFINAL = 1 as emitted (the host rewrites it, lib/nx_deflate.c:158), BTYPE = 10
		const uint32_t nw = (hb + 31) >> 5;
		for (uint32_t i = t; i < nw && i < HDR_WORDS; i += NT) {
			const uint32_t cur = i < 74 ? tb->dhtw[i] : 0, prev = i ? tb->dhtw[i - 1] : 0;
			uint32_t w = (cur << 3) | (i ? prev >> 29 : 5u);
			if (i == (hb >> 5)) w &= (1u << (hb & 31)) - 1;       // the last, partial dword
			win[0][i] = w;
		}
		base_bits = hb;
	} else {
		if (t == 0) win[0][0] = 3u;                              // BFINAL = 1, BTYPE = 01
		base_bits = 3;
	}
	__syncthreads();
	// round 0's records: no round in front to hide behind
	{
		fetch_recs(0, nx);
#pragma unroll
		for (int j = 0; j < 3; j++) if (t + 256 * j < RECMAX) recbuf[0][t + 256 * j] = nx.rec[j];
	}
	__syncthreads();

	uint32_t wordbase = 0, par = 0;
	bool missing = false;
	for (uint32_t r0 = 0; r0 < n; r0 += RPOS, par ^= 1) {
		const Fetch k = nx;
		const bool more = r0 + RPOS < n;
		if (more) { fetch(r0 + RPOS, nx); fetch_recs(r0 + RPOS, nx); }
		uint32_t *w_ = win[par];
		const uint32_t p0 = r0 + 8 * t;
		const uint32_t sh8 = p0 & 24;
		const uint32_t lit8 = (k.litw >> sh8) & 0xff, tok8 = (k.tokw >> sh8) & 0xff;
		Quad q0{0, 0, 0, 0}, q1{0, 0, 0, 0};
		if (__ballot((lit8 | tok8) != 0)) {
			// my first record: matches of the round in front of my positions
			uint32_t ri = (uint32_t)rankpre[p0 >> 5 < 2048 ? p0 >> 5 : 2048] + (uint32_t)__popc(k.tokw & ((1u << (p0 & 31)) - 1)) - (uint32_t)rankpre[r0 >> 5];
			if (p0 >= n) ri = 0;
			q0 = encode_quad<CHECK>(lltab, dtab, recbuf[par], ri, k.bytes.x, lit8 & 15, tok8 & 15, missing);
			q1 = encode_quad<CHECK>(lltab, dtab, recbuf[par], ri, k.bytes.y, lit8 >> 4, tok8 >> 4, missing);
		}
		const uint32_t nbits = q0.nb + q1.nb;
		uint32_t incl = nbits;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
		if (lane == 63) wsum[par][wave] = incl;
		__syncthreads();
		uint32_t bitpos = base_bits + incl - nbits, roundbits = 0;
#pragma unroll
		for (int w = 0; w < NT / 64; w++) { const uint32_t s = wsum[par][w]; if (w < wave) bitpos += s; roundbits += s; }
		emit_quad(w_, q0, bitpos);
		emit_quad(w_, q1, bitpos + q0.nb);
		// the next round's records (asked for at the top of this round) into their LDS copy
		if (more) {
#pragma unroll
			for (int j = 0; j < 3; j++) if (t + 256 * j < RECMAX) recbuf[par ^ 1][t + 256 * j] = nx.rec[j];
		}
		__syncthreads();
		// this window leaves (whole dwords) and is cleared; its last, partial dword opens the other
		// window, which the next round fills (atomicOr: that round's lanes may be there already)
		const uint32_t tot = base_bits + roundbits, nfull = tot >> 5;
		for (uint32_t i = t; i < nfull; i += NT) {
			if (wordbase + i < cap_words) dstw[wordbase + i] = w_[i];
			w_[i] = 0;
		}
		if (t == 0) {
			const uint32_t keep = w_[nfull];
			if (keep) atomicOr(&win[par ^ 1][0], keep);
			w_[nfull] = 0;
		}
		wordbase += nfull;
		base_bits = tot & 31;
	}
	if (missing) errflag = 1;
	__syncthreads();
	uint32_t *w_ = win[par];
	// a job without any round (n == 0) still has its header in the window
	if (base_bits >= 32) {
		const uint32_t nfull = base_bits >> 5;
		for (uint32_t i = t; i < nfull; i += NT)
			if (wordbase + i < cap_words) dstw[wordbase + i] = w_[i];
		const uint32_t keep = w_[nfull];
		__syncthreads();
		if (t == 0) w_[0] = keep;
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
		const uint64_t acc = (uint64_t)w_[0] | ((uint64_t)(lt & 0xffff) << base_bits);
		const uint32_t bits = base_bits + (lt >> 16);
		const uint64_t totbits = (uint64_t)wordbase * 32 + bits;
		const uint32_t tpbc = (uint32_t)((totbits + 7) >> 3);
		if (tpbc > job.dst_cap) cc = cc ? cc : NXZ_CC_TARGET_SPACE;
		if (cc != NXZ_CC_TARGET_SPACE && (uint64_t)wordbase * 4 + (bits + 7) / 8 <= job.dst_cap) {
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
	// NXZ_ENCODE_CHECK=1: the tables the device made go through the kernel form that looks for symbols without a code as well
	// (cc 11 instead of a zero-length code in the block if the LZ77 kernel's counts and its tokens ever disagreed) -- the same
	// bytes, a few per cent slower; tests/test_gpu_parity.py runs the corpus blocks this way.  (Read at every call: the tests switch it.)
	const char *chk = getenv("NXZ_ENCODE_CHECK");
	if (dht && table_per_job && !(chk && atoi(chk) != 0)) hipLaunchKernelGGL((nxze::encode_kernel<true, false>), dim3((unsigned)n), dim3(nxze::NT), 0, stream, jobs, tokens, tables, table_per_job, results, (uint32_t)n);
	else if (dht) hipLaunchKernelGGL(nxze::encode_kernel<true>, dim3((unsigned)n), dim3(nxze::NT), 0, stream, jobs, tokens, tables, table_per_job, results, (uint32_t)n);
	else hipLaunchKernelGGL(nxze::encode_kernel<false>, dim3((unsigned)n), dim3(nxze::NT), 0, stream, jobs, tokens, tables, 0, results, (uint32_t)n);
	return (int)hipGetLastError();
}
