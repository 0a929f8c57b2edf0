// nxz_inflate_lanes.hip -- batched DEFLATE decompression, one stream per LANE.
//
// Same engine function as nxz_inflate.hip (GZIP_FC_DECOMPRESS / _RESUME semantics, identical
// results incl. the suspend fields; CPU restatement: oracle/nxz_inflate.c), organised for
// throughput over many independent streams: a deflate stream is serial, so instead of
// spending a wavefront on one stream, every lane decodes its own.  64 streams advance per
// wave instruction; divergence (literal / match / header) costs issue slots but keeps all
// lanes on useful work.  The wide parts are done by the whole wave:
//   - dynamic-table construction: a lane that reaches a type-2 block decodes the code lengths
//     into the wave's LDS scratch, then all 64 lanes fill that lane's lookup tables
//   - CRC-32 / Adler-32 of the outputs: a separate cooperative kernel afterwards
// Per-lane lookup tables live in a workspace in HBM/L2 (3.3 KiB per resident lane); the fixed
// Huffman tables are shared.  The wave-per-stream kernel stays for single host jobs.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdlib.h>
#include <stdint.h>
#include "nxz_device.h"
#include "nxz_lane_io.h"

namespace nxzl {

#ifndef NXZ_LANES_LB
#define NXZ_LANES_LB 10
#endif
#ifndef NXZ_LANES_DB
#define NXZ_LANES_DB 8
#endif
constexpr int LB = NXZ_LANES_LB, DB = NXZ_LANES_DB;  // fast-table index bits
constexpr uint32_t WS_LIT = 0, WS_DIST = 2048, WS_LCNT = 2560, WS_LSYM = 2592, WS_DCNT = 3168, WS_DSYM = 3200;
static_assert((2u << LB) <= WS_DIST - WS_LIT && (2u << DB) <= WS_LCNT - WS_DIST, "the fast tables (16-bit entries) must fit their slices of the lane's table set");
// (decode_long's packed counts: 9 bits a length for the literal/length code, 5 for the distance code -- every parser of this
// engine, like zlib, turns HDIST > 30 away, so no length has more than 30 distance codes)
constexpr uint32_t WS_LPK = 3264, WS_DPK = 3280;     // 16 bytes each: what the walk for codes longer than the fast table needs, packed (decode_long)
constexpr uint32_t WS_LENS = 320;                     // code lengths of the block being set up (last part of a slot)
constexpr uint32_t WS_BYTES = 3328 + WS_LENS;

struct Tab {                                         // one table set (workspace view)
	const uint16_t *lit, *dist, *lcnt, *lsym, *dcnt, *dsym;
};

__device__ __forceinline__ Tab tab_at(const uint8_t *ws)
{
	return Tab{ (const uint16_t *)(ws + WS_LIT), (const uint16_t *)(ws + WS_DIST), (const uint16_t *)(ws + WS_LCNT),
		    (const uint16_t *)(ws + WS_LSYM), (const uint16_t *)(ws + WS_DCNT), (const uint16_t *)(ws + WS_DSYM) };
}

// A code no longer than the fast table's index: one load.  A longer one (LONG_CODE) takes the walk through the counts
// below, a dozen loads one after the other -- and what one lane does, its wavefront does: with 64 streams side by side
// some lane meets a rare symbol at nearly every token, so the literal loop looks codes up here only and leaves the
// walk to ONE place behind it (zlib -6 streams of the corpus: 54 -> 71 GiB/s with 10 / 8 index bits instead of 9 / 7,
// -> with the walk in one place).
constexpr int LONG_CODE = -100;
template <int FB>
__device__ __forceinline__ int decode_fast(const uint16_t *fast, uint32_t bits, uint32_t &nb)
{
	const uint32_t e = fast[bits & ((1u << FB) - 1)];
	nb = e >> 12;
	return e ? (int)(e & 0xfff) : LONG_CODE;
}
// A code longer than the fast table's FB bits: the canonical walk (count, first code and index per length) from where
// FB bits leave it -- its state there does not depend on the bits -- with the counts of the lengths FB + 1 .. 15 (CB bits
// each), the first code and the index at length FB + 1 packed in 16 bytes of the table set (build_tables): one load, a few
// steps of arithmetic, one load of the symbol, where the walk through the counts in memory was a load a step.
template <int FB, int CB>
__device__ __forceinline__ int decode_long(const uint8_t *pk, const uint16_t *symt, uint32_t bits, uint32_t &nb)
{
	const uint4 p = *(const uint4 *)pk;
	uint64_t cnts = (uint64_t)p.x | (uint64_t)p.y << 32;
	int first = (int)(p.z & 0xffff), index = (int)(p.z >> 16);
	int code = (int)((__builtin_bitreverse32(bits) >> (32 - FB)) << 1);
	bits >>= FB;
	for (int len = FB + 1; len <= 15; len++) {
		code |= (int)(bits & 1); bits >>= 1;
		const int c = (int)(cnts & ((1u << CB) - 1)); cnts >>= CB;
		if (code - c < first) { nb = (uint32_t)len; return symt[index + (code - first)]; }
		index += c; first += c; first <<= 1; code <<= 1;
	}
	nb = 16;
	return -2;
}
template <int FB, int CB>
__device__ __forceinline__ int decode(const uint16_t *fast, const uint8_t *pk, const uint16_t *symt, uint32_t bits, uint32_t &nb)
{
	uint32_t e = fast[bits & ((1u << FB) - 1)];
	if (e) { nb = e >> 12; return (int)(e & 0xfff); }
	return decode_long<FB, CB>(pk, symt, bits, nb);
}

// The fixed code (RFC1951 3.2.6) needs no table: the symbol follows from the first 9 bits by
// arithmetic, which takes the lookup (and its latency) out of every token of a type-1 block.
#ifndef NXZ_LANES_WAIT_TRIPS
#define NXZ_LANES_WAIT_TRIPS 64
#define NXZ_LANES_WAIT_LANES 8
#endif
#ifndef NXZ_LANES_LITS
#define NXZ_LANES_LITS 6                // literals a lane may decode in one trip round its token loop
#endif
__device__ __forceinline__ int decode_fixed_ll(uint32_t bits, uint32_t &nb)
{
	const uint32_t r9 = __builtin_bitreverse32(bits) >> 23;       // the first 9 bits, first bit most significant
	const uint32_t r8 = r9 >> 1, r7 = r9 >> 2;
	const bool a = r7 < 24, b8 = r8 < 0xC0, c = r8 < 0xC8;         // 7-bit lengths 256..279 | literals 0..143 | 280..287 | literals 144..255
	nb = a ? 7 : c ? 8 : 9;
	return (int)(a ? 256 + r7 : b8 ? r8 - 0x30 : c ? 280 + (r8 - 0xC0) : 144 + (r9 - 0x190));
}
__device__ __forceinline__ int decode_fixed_d(uint32_t bits, uint32_t &nb)
{
	const uint32_t d = __builtin_bitreverse32(bits) >> 27;        // 5 bits; 30 and 31 are no codes: same answer as the
	nb = d < 30 ? 5 : 16;                                         // table walk gives for them (decode() above)
	return d < 30 ? (int)d : -2;
}

__device__ __forceinline__ void len_params(uint32_t s, uint32_t &base, uint32_t &extra)
{
	extra = s < 8 || s == 28 ? 0 : (s - 4) >> 2;
	base = s < 8 ? 3 + s : s == 28 ? 258 : ((4 + (s & 3)) << extra) + 3;
}
__device__ __forceinline__ void dist_params(uint32_t d, uint32_t &base, uint32_t &extra)
{
	extra = d < 4 ? 0 : (d - 2) >> 1;
	base = d < 4 ? d + 1 : ((2 + (d & 1)) << extra) + 1;
}

// Wave-cooperative: build the table set at `ws` from the code lengths in LDS (lens[0..hlit) lit/len,
// lens[hlit..hlit+hdist) distances).  All 64 lanes call this with the same arguments.
__device__ void build_tables(uint8_t *ws, const uint8_t *lens, int hlit, int hdist, int lane)
{
	uint16_t *lit = (uint16_t *)(ws + WS_LIT), *dist = (uint16_t *)(ws + WS_DIST);
	uint16_t *lcnt = (uint16_t *)(ws + WS_LCNT), *lsym = (uint16_t *)(ws + WS_LSYM);
	uint16_t *dcnt = (uint16_t *)(ws + WS_DCNT), *dsym = (uint16_t *)(ws + WS_DSYM);
	for (int i = lane; i < (1 << LB); i += 64) lit[i] = 0;
	for (int i = lane; i < (1 << DB); i += 64) dist[i] = 0;
	// counts per length: lane l (< 16) counts length l
	for (int pass = 0; pass < 2; pass++) {
		const uint8_t *L = pass ? lens + hlit : lens;
		int n = pass ? hdist : hlit;
		uint16_t *cnt = pass ? dcnt : lcnt, *symt = pass ? dsym : lsym, *fast = pass ? dist : lit;
		const int FB = pass ? DB : LB;
		uint32_t mycnt = 0;
		if (lane >= 1 && lane < 16) for (int i = 0; i < n; i++) mycnt += (L[i] == lane);
		// offsets / first codes: prefix over lanes 1..15
		uint32_t offs = 0, next = 0;
		{
			uint32_t c = 0, o = 0;
			uint64_t packed = 0;                                 // counts of the lengths FB + 1 .. 15, CB bits each (decode_long)
			uint32_t at = 0, fi = 0;
			const int CB = pass ? 5 : 9;
			for (int b = 1; b < 16; b++) {
				uint32_t cb = __shfl(mycnt, b, 64);
				if (b == lane) { offs = o; next = c; }
				if (b == FB + 1) fi = (c & 0xffff) | o << 16;      // first code and index at length FB + 1
				if (b > FB) { packed |= (uint64_t)cb << at; at += CB; }
				o += cb;
				c = (c + cb) << 1;
			}
			if (lane == 0) *(uint4 *)(ws + (pass ? WS_DPK : WS_LPK)) = make_uint4((uint32_t)packed, (uint32_t)(packed >> 32), fi, 0);
		}
		if (lane < 16) cnt[lane] = (uint16_t)(lane ? mycnt : 0);
		// sorted symbol list + fast entries: lane per symbol
		for (int i = lane; i < ((n + 63) & ~63); i += 64) {
			uint32_t l = i < n ? L[i] : 0;
			uint32_t rank = 0;
			if (l) for (int k = 0; k < i; k++) rank += (L[k] == l);
			uint32_t o = __shfl(offs, (int)l, 64), nx = __shfl(next, (int)l, 64);
			if (l) {
				symt[o + rank] = (uint16_t)i;
				if (l <= (uint32_t)FB) {
					uint32_t code = nx + rank;
					uint32_t rev = __builtin_bitreverse32(code) >> (32 - l);
					for (uint32_t idx = rev; idx < (1u << FB); idx += (1u << l)) fast[idx] = (uint16_t)(i | (l << 12));
				}
			}
		}
	}
	__threadfence_block();
}

struct LaneState {
	uint32_t cc, out, o_sfbt, o_subc, o_rem, dhtbits;
	bool final_eob;
};

// results.tebc carries out_rembytecnt; results.sfbt bit 8 = final EOB, bits 16.. = dhtlen (see nxz_engine.h)
#ifndef NXZ_LANES_WPE
#define NXZ_LANES_WPE 4      /* wavefronts per SIMD the register budget is cut for (128 VGPRs at 4) */
#endif
__global__ __launch_bounds__(64, NXZ_LANES_WPE) void inflate_lanes_kernel(const nxz_batch_job_t *__restrict__ jobs, size_t n,
							   nxz_batch_result_t *__restrict__ results,
							   nxz_batch_dht_t *__restrict__ dht_io,
							   uint8_t *__restrict__ workspace, const uint8_t *__restrict__ fixed_ws,
							   const uint32_t *__restrict__ order, uint32_t per_wave, const uint32_t *__restrict__ only_if, uint32_t skip)
{
	// only_if (may be NULL): launched behind the fixed-code kernel below, this one does the streams that one handed back and
	// no others: the word counts them, their indices follow it (a batch of 262 144 fixed-code streams with one dynamic block
	// among them took twice its time up to round 4, when the word was a flag and this kernel did the whole batch again)
	// (the first `skip` of them are the stream-per-wave kernel's: a lane takes tens of milliseconds for a stream however few there are)
	if (only_if) {
		const uint32_t c = *(const volatile uint32_t *)only_if;
		if (c <= skip) return;
		n = c - skip;
		order = only_if + 64 + skip;
	}
	__shared__ __attribute__((aligned(16))) uint8_t lens_s[320];
	const int lane = threadIdx.x;
	// slot 0 = fixed tables; then 65 slots per wave: one per lane and a spare that lanes with identical
	// dynamic tables share (blocks compressed with one table, as this engine and the reference's table
	// reuse produce them: one construction instead of 64, and lookups that hit the same cache lines)
	uint8_t *myws = workspace + ((size_t)blockIdx.x * 65 + lane + 1) * WS_BYTES;
	uint8_t *shared_ws = workspace + ((size_t)blockIdx.x * 65 + 64 + 1) * WS_BYTES;

	// per_wave: streams a wavefront takes at a time, 64 or 32.  The kernel lives on the number of wavefronts a CU holds --
	// a lane waits for its stream's bytes, and only other wavefronts fill the gap --, so a batch that would leave the
	// device half empty at 64 a wavefront (fewer than 131072 streams) runs 32 a wavefront on twice as many.
	for (size_t g = blockIdx.x; g * per_wave < n; g += gridDim.x) {
		// order (may be NULL): the jobs by falling source length, so that the 64 streams of a wavefront are much of a
		// size and the long ones go first (a wavefront takes as long as its longest stream: a batch of mixed kinds --
		// zeros, text, copies by turns, BASELINE configs[4] -- ran at half the rate of its kinds one by one)
		const bool active = (uint32_t)lane < per_wave && g * per_wave + lane < n;
		const size_t jid = active && order ? order[g * per_wave + lane] : g * per_wave + lane;
		nxz_batch_job_t job;
		if (active) job = jobs[jid];
		else { job.src = nullptr; job.dst = nullptr; job.src_len = 0; job.hist_len = 0; job.dst_cap = 0; job.resume = 0; job.in_crc = 0; job.in_adler = 1; }
		const uint32_t hist = job.hist_len < job.src_len ? job.hist_len : job.src_len;
		const uint8_t *hsrc = job.src;                           // history bytes [0, hist)
		uint8_t *dst = job.dst;
		const uint32_t cap = job.dst_cap;
		BitRd b;
		b.src = job.src + hist; b.srclen = job.src_len - hist; b.bb = 0; b.bc = 0; b.pos = 0;
		const uint32_t in_subc = (job.resume >> 20) & 7, in_sfbt = (job.resume >> 16) & 15, in_rem = job.resume & 0xffff;
		if (b.srclen && in_subc) b.pos = 8 - in_subc;

		OutWr w{ dst, 0, 0, 0, ((uintptr_t)dst & 3) == 0 };
		uint32_t cc = 0, o_sfbt = 0, o_subc = 0, o_rem = 0, dhtbits = 0;
		uint32_t bfinal = 0, btype = 0, rem = 0;
		int state = active ? 0 : 3;                              // 0 header, 1 stored, 2 coded, 3 done, 4 = need table build
		bool final_eob = false;
		Tab T = tab_at(fixed_ws);
		uint64_t tstart = 0;                                     // where the dynamic table bits start
		bool table_from_slot = false;
		bool using_shared = false;                               // T is the wave's spare slot

		if (active && (in_sfbt & 8)) {
			uint32_t kind = (in_sfbt >> 1) & 7;
			bfinal = in_sfbt & 1;
			if (kind == 4) { state = 1; btype = 0; rem = in_rem; }
			else if (kind == 5) { state = 2; btype = 1; }
			else if (kind == 6) { state = 4; btype = 2; table_from_slot = true; if (!dht_io) { cc = NXZ_CC_INVALID_DHT; state = 3; } }
		}

		// The wave loops until every lane is done.  Lanes in state 4 (need a dynamic table) are
		// served one at a time by the whole wave.
		for (;;) {
			// ---------- dynamic tables ----------
			// Every lane that needs a table parses its own block header at the same time (HLIT /
			// HDIST / HCLEN + code lengths, into its slice of the workspace); only the construction
			// of the lookup tables is done by the whole wave, one owner after the other.
			int rc = 0, hlit = 0, hdist = 0;
			if (state == 4) {
				uint8_t *mylens = myws + WS_BYTES - WS_LENS;
				BitRd r = b;
				if (table_from_slot) {
					const nxz_batch_dht_t *t = &dht_io[jid];
					r.src = t->dht; r.srclen = (t->dhtlen + 7) / 8; r.bb = 0; r.bc = 0; r.pos = 0;
				}
				const uint64_t tbits_avail = table_from_slot ? dht_io[jid].dhtlen : r.total();
				auto have = [&](uint32_t k) { return r.pos + k <= tbits_avail; };
				tstart = r.pos;
				if (!have(14)) rc = 1;
				else {
					uint32_t v = r.take(14);
					hlit = (int)(v & 31) + 257; hdist = (int)((v >> 5) & 31) + 1;
					int hclen = (int)((v >> 10) & 15) + 4;
					if (hlit > 286 || hdist > 30) rc = -1;
					// code-length code lengths, 3 bits each, packed 19 x 3 bits
					uint64_t clp = 0;
					for (int i = 0; i < hclen && !rc; i++) {
						if (!have(3)) { rc = 1; break; }
						static const uint8_t ord[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
						clp |= (uint64_t)r.take(3) << (3 * ord[i]);
					}
					if (!rc) {
						// canonical code of the <= 7-bit code-length alphabet: counts (8 x 5 bits) and sorted symbols (19 x 5 bits)
						uint64_t cnts = 0; uint32_t kraft = 0;
						for (int i = 0; i < 19; i++) { uint32_t l = (uint32_t)(clp >> (3 * i)) & 7; if (l) { cnts += 1ull << (5 * l); kraft += 128u >> l; } }
						if (kraft > 128) rc = -2;
						uint64_t s_lo = 0, s_hi = 0; int ns = 0;               // sorted symbol list, 5 bits each
						for (uint32_t l = 1; l <= 7; l++) for (int i = 0; i < 19; i++) if (((uint32_t)(clp >> (3 * i)) & 7) == l) {
							if (ns < 12) s_lo |= (uint64_t)i << (5 * ns); else s_hi |= (uint64_t)i << (5 * (ns - 12));
							ns++;
						}
						int nlen = 0, prev = 0;
						uint32_t k1 = 0, k2 = 0, eob = 0;                      // Kraft sums (x 32768) and the end-of-block length
						auto emit = [&](uint32_t val, int rep) {
							for (int k = 0; k < rep; k++) mylens[nlen + k] = (uint8_t)val;
							if (val) {
								const int nl = hlit - nlen < 0 ? 0 : hlit - nlen < rep ? hlit - nlen : rep;   // how many are literal/length codes
								k1 += (uint32_t)nl * (32768u >> val);
								k2 += (uint32_t)(rep - nl) * (32768u >> val);
							}
							if (nlen <= 256 && 256 < nlen + rep) eob = val;
							nlen += rep;
						};
						while (!rc && nlen < hlit + hdist) {
							if (!have(1)) { rc = 1; break; }
							r.fill();
							uint32_t bits = (uint32_t)r.bb;
							int code = 0, first = 0, index = 0, sym = -1, len;
							for (len = 1; len <= 7; len++) {
								code |= (int)(bits & 1); bits >>= 1;
								int c = (int)((cnts >> (5 * len)) & 31);
								if (code - c < first) {
									int k = index + (code - first);
									sym = (int)((k < 12 ? s_lo >> (5 * k) : s_hi >> (5 * (k - 12))) & 31);
									break;
								}
								index += c; first += c; first <<= 1; code <<= 1;
							}
							if (sym < 0) { rc = have(7) ? -3 : 1; break; }
							if (!have((uint32_t)len)) { rc = 1; break; }
							r.drop((uint32_t)len);
							if (sym < 16) { emit((uint32_t)sym, 1); prev = sym; }
							else {
								int eb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
								if (!have((uint32_t)eb)) { rc = 1; break; }
								int rep = (int)r.take((uint32_t)eb) + (sym == 18 ? 11 : 3);
								int val = 0;
								if (sym == 16) { if (nlen == 0) { rc = -4; break; } val = prev; }
								if (nlen + rep > hlit + hdist) { rc = -5; break; }
								emit((uint32_t)val, rep);
								if (sym != 16) prev = 0;
							}
						}
						if (!rc) {
							if (eob == 0) rc = -6;
							if (k1 > 32768u || k2 > 32768u) rc = -7;
						}
					}
				}
				if (!rc) {
					dhtbits = (uint32_t)(r.pos - tstart);
					if (table_from_slot) { if (dhtbits != dht_io[jid].dhtlen) rc = -8; }
					else {
						// keep the table bits for a possible suspend inside this block
						if (dht_io) {
							nxz_batch_dht_t *t = &dht_io[jid];
							BitRd c = b; c.sync();
							for (uint32_t i = 0; i * 8 < dhtbits; i++) {
								uint32_t k = dhtbits - i * 8 < 8 ? dhtbits - i * 8 : 8;
								c.fill();
								t->dht[i] = (uint8_t)((uint32_t)c.bb & ((1u << k) - 1));
								c.drop(k);
							}
							t->dhtlen = dhtbits;
						}
						b = r; b.sync();
					}
				}
			}
			unsigned long long need = __ballot(state == 4);
			while (need) {
				const int owner = __builtin_ctzll(need);
				need &= need - 1;
				const int orc = __shfl(rc, owner, 64), ohlit = __shfl(hlit, owner, 64), ohdist = __shfl(hdist, owner, 64);
				uint8_t *ows = workspace + ((size_t)blockIdx.x * 65 + owner + 1) * WS_BYTES;
				uint8_t *target = ows;
				unsigned long long sharers = 0;
				if (orc == 0) {
					const uint8_t *olens = ows + WS_BYTES - WS_LENS;
					__syncthreads();
					for (int i = lane; i < ohlit + ohdist; i += 64) lens_s[i] = olens[i];
					__syncthreads();
					// waiting lanes whose code lengths equal the owner's take the same table: built once, into
					// the spare slot, when no lane is still decoding with what that slot held before
					bool same = state == 4 && lane != owner && ((need >> lane) & 1) && rc == 0 && hlit == ohlit && hdist == ohdist;
					if (same) {
						const uint32_t nl = (uint32_t)(ohlit + ohdist);
						const uint8_t *mine = myws + WS_BYTES - WS_LENS;
						const uint32_t *a = (const uint32_t *)mine, *q = (const uint32_t *)lens_s;
						for (uint32_t i = 0; same && i < nl / 4; i++) same = a[i] == q[i];
						for (uint32_t i = nl & ~3u; same && i < nl; i++) same = mine[i] == lens_s[i];
					}
					sharers = __ballot(same);
					const bool spare_busy = __ballot(using_shared && state == 2) != 0;
					if (sharers && !spare_busy) target = shared_ws; else sharers = 0;
					build_tables(target, lens_s, ohlit, ohdist, lane);
				}
				__syncthreads();
				if (lane == owner) {
					if (orc == 0) { T = tab_at(target); using_shared = target == shared_ws; state = 2; }
					else if (orc == 1 && !table_from_slot) {       // ran out of source inside the header
						uint64_t hdr = tstart - 3;
						o_sfbt = 0xe | bfinal; o_subc = (uint32_t)(b.total() - hdr); state = 3;
					} else { cc = NXZ_CC_INVALID_DHT; state = 3; }
					table_from_slot = false;
				} else if ((sharers >> lane) & 1) {
					T = tab_at(target); using_shared = true; state = 2;
					table_from_slot = false;
				}
				need &= ~sharers;
			}
			if (!__any(state != 3)) break;

			// ---------- per-lane decode: run until this lane needs a table or is done ----------
			// (a lane that meets a block header waits for the next turn of the table set-up above, and it shall not wait long: the
			// others break off after NXZ_LANES_WAIT_TRIPS more trips, or at once when NXZ_LANES_WAIT_LANES lanes wait -- round 4: they
			// went on for up to 4096 trips, and streams of two or three blocks spent more time waiting than decoding)
			uint32_t waited = 0;
			for (int steps = 0; steps < 4096; steps++) {
				const unsigned long long waiting = __ballot(state == 4);
				if (!__ballot(state != 3 && state != 4)) break;
				if (waiting && (++waited > NXZ_LANES_WAIT_TRIPS || __popcll(waiting) >= NXZ_LANES_WAIT_LANES)) break;
				if (state != 3 && state != 4) do {
				w.commit();                                         // every lane, here: see OutWr
				if (state == 0) {
					b.sync();
					uint64_t hdr = b.pos;
					if (!b.have(3)) { o_sfbt = 0xe; o_subc = (uint32_t)(b.total() - hdr); state = 3; break; }
					uint32_t v = b.take(3);
					bfinal = v & 1; btype = v >> 1;
					if (btype == 0) {
						b.pos = (b.pos + 7) & ~7ull; b.sync();
						if (!b.have(32)) { o_sfbt = 0xe | bfinal; o_subc = (uint32_t)(b.total() - hdr); state = 3; break; }
						uint32_t lo = b.take(16), hi = b.take(16);
						if ((lo ^ hi) != 0xffff) { cc = NXZ_CC_INVALID_DHT; state = 3; break; }
						rem = lo; state = 1;
					} else if (btype == 1) { T = tab_at(fixed_ws); using_shared = false; state = 2; }
					else if (btype == 2) { state = 4; }
					else { cc = NXZ_CC_INVALID_DHT; state = 3; }
				} else if (state == 1) {
					uint32_t sp = (uint32_t)(b.pos >> 3);
					uint32_t srcleft = b.srclen - sp;
					uint32_t k = rem < srcleft ? rem : srcleft;
					if (k > cap - w.out) { cc = NXZ_CC_TARGET_SPACE; state = 3; break; }
					w.flush();
					w.copy_in(b.src + sp, k);
					rem -= k; b.pos = (uint64_t)(sp + k) * 8; b.sync();
					if (rem) { o_sfbt = 0x8 | bfinal; o_subc = 0; o_rem = rem; state = 3; break; }
					if (bfinal) { final_eob = true; state = 3; break; }
					state = 0;
				} else {
					const uint32_t sfbt = (btype == 1 ? 0xa : 0xc) | bfinal;
					uint32_t nb;
					b.refill();                                     // every lane, here: see BitRd
					int sym = btype == 1 ? decode_fixed_ll((uint32_t)b.bb, nb) : decode_fast<LB>(T.lit, (uint32_t)b.bb, nb);
					// up to NXZ_LANES_LITS literals before the wavefront's lanes meet again (see the fixed-code kernel below)
					for (int nl = 1; nl < NXZ_LANES_LITS && sym >= 0 && sym < 256 && b.have(nb) && w.out < cap; nl++) {
						b.drop(nb);
						w.lit((uint32_t)sym);
						b.need(15);
						sym = btype == 1 ? decode_fixed_ll((uint32_t)b.bb, nb) : decode_fast<LB>(T.lit, (uint32_t)b.bb, nb);
					}
					if (sym == LONG_CODE) sym = decode_long<LB, 9>((const uint8_t *)T.lit + (WS_LPK - WS_LIT), T.lsym, (uint32_t)b.bb, nb);     // (the one place: see decode_fast)
					const uint64_t sym_start = b.pos;
					if (sym < 0 || !b.have(nb)) {
						if (!b.have(sym < 0 ? 15 : nb)) { o_sfbt = sfbt; o_subc = (uint32_t)(b.total() - sym_start); }
						else cc = NXZ_CC_MISSING_CODE;
						state = 3; break;
					}
					b.drop(nb);
					if (sym < 256) {
						if (w.out >= cap) { cc = NXZ_CC_TARGET_SPACE; state = 3; break; }
						w.lit((uint32_t)sym);
					} else if (sym == 256) {
						if (bfinal) { final_eob = true; state = 3; break; }
						state = 0;
					} else {
						sym -= 257;
						if (sym >= 29) { cc = NXZ_CC_MISSING_CODE; state = 3; break; }
						uint32_t lbase, eb;
						len_params((uint32_t)sym, lbase, eb);
						b.need(5);
						if (!b.have(eb)) { o_sfbt = sfbt; o_subc = (uint32_t)(b.total() - sym_start); state = 3; break; }
						uint32_t len = lbase + ((uint32_t)b.bb & ((1u << eb) - 1));
						b.drop(eb);
						b.need(15);
						int ds = btype == 1 ? decode_fixed_d((uint32_t)b.bb, nb) : decode<DB, 5>(T.dist, (const uint8_t *)T.lit + (WS_DPK - WS_LIT), T.dsym, (uint32_t)b.bb, nb);
						if (ds < 0 || !b.have(nb)) {
							if (!b.have(ds < 0 ? 15 : nb)) { o_sfbt = sfbt; o_subc = (uint32_t)(b.total() - sym_start); }
							else cc = NXZ_CC_INVALID_DIST;
							state = 3; break;
						}
						if (ds >= 30) { cc = NXZ_CC_INVALID_DIST; state = 3; break; }
						b.drop(nb);
						uint32_t dbase;
						dist_params((uint32_t)ds, dbase, eb);
						b.need(13);
						if (!b.have(eb)) { o_sfbt = sfbt; o_subc = (uint32_t)(b.total() - sym_start); state = 3; break; }
						uint32_t dist = dbase + ((uint32_t)b.bb & ((1u << eb) - 1));
						b.drop(eb);
						if (dist > w.out + hist || dist > 32768) { cc = NXZ_CC_INVALID_DIST; state = 3; break; }
						if (len > cap - w.out) { cc = NXZ_CC_TARGET_SPACE; state = 3; break; }
						if (len <= 8 && dist >= 16 && dist <= w.out) { w.copy_short(len, dist); continue; }
						w.flush();
						if (dist > w.out) {
							// (part of) the source is in the history buffer
							for (uint32_t i = 0; i < len; i++) {
								int64_t sidx = (int64_t)w.out + i - dist;
								dst[w.out + i] = sidx < 0 ? hsrc[hist + sidx] : dst[sidx];
							}
							w.out += len;
						} else w.copy(len, dist);
					}
				}
				} while (0);
			}
		}
		w.flush();
		if (active) {
			if (final_eob) { o_sfbt = 0; o_subc = (uint32_t)(b.total() - b.pos); }
			nxz_batch_result_t r;
			uint32_t spbc = job.src_len, subc = o_subc;
			if (final_eob && subc > 0xfff8) { uint32_t drop = (subc - 0xfff8 + 7) / 8; spbc -= drop; subc -= drop * 8; }
			if (cc == 0 && !(final_eob && subc < 8)) cc = NXZ_CC_DATA_LENGTH;
			r.cc = cc; r.tpbc = (cc == 0 || cc == NXZ_CC_DATA_LENGTH) ? w.out : 0;
			r.tebc = o_rem; r.spbc = spbc; r.crc = 0; r.adler = 0;
			r.subc = subc; r.sfbt = o_sfbt | (final_eob ? 0x100u : 0) | (((o_sfbt & 0xe) == 0xc) ? (dhtbits << 16) : 0);
			results[jid] = r;
		}
		__syncthreads();
	}
}

// fixed-Huffman table set (RFC1951 3.2.6) in workspace format
// Streams of fixed-Huffman and stored blocks only -- what this engine's fixed-code deflate makes, and zlib's Z_FIXED: the
// same lane-per-stream decode without anything a dynamic table needs (no header parse, no table builds by the whole
// wave, no table pointers and no workspace): a kernel of its own, which the compiler gives fewer registers and the CU
// more wavefronts.  A stream that turns out to hold a dynamic block (or resumes inside one) sets *bail, and the kernel
// above, queued behind this one, then does the whole batch again.
#ifndef NXZ_LANES_FIXED_WPE
#define NXZ_LANES_FIXED_WPE 6
#endif
__global__ __launch_bounds__(64, NXZ_LANES_FIXED_WPE) void inflate_lanes_fixed_kernel(const nxz_batch_job_t *__restrict__ jobs, size_t n,
							   nxz_batch_result_t *__restrict__ results,
							   const uint32_t *__restrict__ order, uint32_t per_wave, uint32_t *__restrict__ bail)
{
	const int lane = threadIdx.x;
	for (size_t g = blockIdx.x; g * per_wave < n; g += gridDim.x) {
		const bool active = (uint32_t)lane < per_wave && g * per_wave + lane < n;
		const size_t jid = active && order ? order[g * per_wave + lane] : g * per_wave + lane;
		nxz_batch_job_t job;
		if (active) job = jobs[jid];
		else { job.src = nullptr; job.dst = nullptr; job.src_len = 0; job.hist_len = 0; job.dst_cap = 0; job.resume = 0; job.in_crc = 0; job.in_adler = 1; }
		const uint32_t hist = job.hist_len < job.src_len ? job.hist_len : job.src_len;
		const uint8_t *hsrc = job.src;                           // history bytes [0, hist)
		uint8_t *dst = job.dst;
		const uint32_t cap = job.dst_cap;
		BitRd b;
		b.src = job.src + hist; b.srclen = job.src_len - hist; b.bb = 0; b.bc = 0; b.pos = 0;
		const uint32_t in_subc = (job.resume >> 20) & 7, in_sfbt = (job.resume >> 16) & 15, in_rem = job.resume & 0xffff;
		if (b.srclen && in_subc) b.pos = 8 - in_subc;
		OutWr w{ dst, 0, 0, 0, ((uintptr_t)dst & 3) == 0 };
		uint32_t cc = 0, o_sfbt = 0, o_subc = 0, o_rem = 0;
		uint32_t bfinal = 0, rem = 0;
		int state = active ? 0 : 3;                              // 0 header, 1 stored, 2 fixed code, 3 done
		bool final_eob = false, dyn = false;
		if (active && (in_sfbt & 8)) {
			const uint32_t kind = (in_sfbt >> 1) & 7;
			bfinal = in_sfbt & 1;
			if (kind == 4) { state = 1; rem = in_rem; }
			else if (kind == 5) state = 2;
			else if (kind == 6) { dyn = true; state = 3; }
		}
		while (__any(state != 3)) {
			for (int steps = 0; steps < 4096 && state != 3; steps++) {
				w.commit();                                         // every lane, here: see OutWr
				if (state == 0) {
					b.sync();
					const uint64_t hdr = b.pos;
					if (!b.have(3)) { o_sfbt = 0xe; o_subc = (uint32_t)(b.total() - hdr); state = 3; break; }
					const uint32_t v = b.take(3);
					bfinal = v & 1;
					const uint32_t btype = v >> 1;
					if (btype == 0) {
						b.pos = (b.pos + 7) & ~7ull; b.sync();
						if (!b.have(32)) { o_sfbt = 0xe | bfinal; o_subc = (uint32_t)(b.total() - hdr); state = 3; break; }
						const uint32_t lo = b.take(16), hi = b.take(16);
						if ((lo ^ hi) != 0xffff) { cc = NXZ_CC_INVALID_DHT; state = 3; break; }
						rem = lo; state = 1;
					} else if (btype == 1) state = 2;
					else if (btype == 2) { dyn = true; state = 3; }
					else { cc = NXZ_CC_INVALID_DHT; state = 3; }
				} else if (state == 1) {
					const uint32_t sp = (uint32_t)(b.pos >> 3);
					const uint32_t srcleft = b.srclen - sp;
					const uint32_t k = rem < srcleft ? rem : srcleft;
					if (k > cap - w.out) { cc = NXZ_CC_TARGET_SPACE; state = 3; break; }
					w.flush();
					w.copy_in(b.src + sp, k);
					rem -= k; b.pos = (uint64_t)(sp + k) * 8; b.sync();
					if (rem) { o_sfbt = 0x8 | bfinal; o_subc = 0; o_rem = rem; state = 3; break; }
					if (bfinal) { final_eob = true; state = 3; break; }
					state = 0;
				} else {
					const uint32_t sfbt = 0xa | bfinal;
					uint32_t nb;
					b.refill();                                     // every lane, here: see BitRd
					// up to NXZ_LANES_LITS literals before the wavefront's lanes meet again: a trip round the token loop costs the
					// wavefront the same whatever its lanes do in it, and a lane with nothing but literals (text) had 64 Ki trips
					// to make while the lanes with matches waited for it (c5: 123.6 ms -> 77 ms with 3, 71 with 6)
					int sym = decode_fixed_ll((uint32_t)b.bb, nb);
					for (int nl = 1; nl < NXZ_LANES_LITS && sym >= 0 && sym < 256 && b.have(nb) && w.out < cap; nl++) {
						b.drop(nb);
						w.lit((uint32_t)sym);
						b.need(9);
						sym = decode_fixed_ll((uint32_t)b.bb, nb);
					}
					const uint64_t sym_start = b.pos;
					if (sym < 0 || !b.have(nb)) {
						if (!b.have(sym < 0 ? 15 : nb)) { o_sfbt = sfbt; o_subc = (uint32_t)(b.total() - sym_start); }
						else cc = NXZ_CC_MISSING_CODE;
						state = 3; break;
					}
					b.drop(nb);
					if (sym < 256) {
						if (w.out >= cap) { cc = NXZ_CC_TARGET_SPACE; state = 3; break; }
						w.lit((uint32_t)sym);
					} else if (sym == 256) {
						if (bfinal) { final_eob = true; state = 3; break; }
						state = 0;
					} else {
						sym -= 257;
						if (sym >= 29) { cc = NXZ_CC_MISSING_CODE; state = 3; break; }
						uint32_t lbase, eb;
						len_params((uint32_t)sym, lbase, eb);
						b.need(5);
						if (!b.have(eb)) { o_sfbt = sfbt; o_subc = (uint32_t)(b.total() - sym_start); state = 3; break; }
						const uint32_t len = lbase + ((uint32_t)b.bb & ((1u << eb) - 1));
						b.drop(eb);
						b.need(5);
						const int ds = decode_fixed_d((uint32_t)b.bb, nb);
						if (ds < 0 || !b.have(nb)) {
							if (!b.have(ds < 0 ? 15 : nb)) { o_sfbt = sfbt; o_subc = (uint32_t)(b.total() - sym_start); }
							else cc = NXZ_CC_INVALID_DIST;
							state = 3; break;
						}
						if (ds >= 30) { cc = NXZ_CC_INVALID_DIST; state = 3; break; }
						b.drop(nb);
						uint32_t dbase;
						dist_params((uint32_t)ds, dbase, eb);
						b.need(13);
						if (!b.have(eb)) { o_sfbt = sfbt; o_subc = (uint32_t)(b.total() - sym_start); state = 3; break; }
						const uint32_t dist = dbase + ((uint32_t)b.bb & ((1u << eb) - 1));
						b.drop(eb);
						if (dist > w.out + hist || dist > 32768) { cc = NXZ_CC_INVALID_DIST; state = 3; break; }
						if (len > cap - w.out) { cc = NXZ_CC_TARGET_SPACE; state = 3; break; }
						if (len <= 8 && dist >= 16 && dist <= w.out) { w.copy_short(len, dist); continue; }
						w.flush();
						if (dist > w.out) {
							// (part of) the source is in the history buffer
							for (uint32_t i = 0; i < len; i++) {
								const int64_t sidx = (int64_t)w.out + i - dist;
								dst[w.out + i] = sidx < 0 ? hsrc[hist + sidx] : dst[sidx];
							}
							w.out += len;
						} else w.copy(len, dist);
					}
				}
			}
		}
		w.flush();
		{
			// streams with a dynamic block in them: handed to the other kernel, which does them from their first byte
			const unsigned long long dm = __ballot(dyn);
			if (dm) {
				uint32_t at = 0;
				if (lane == 0) at = atomicAdd(bail, (uint32_t)__popcll(dm));
				at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
				if (dyn) bail[64 + at + (uint32_t)__popcll(dm & ((1ull << lane) - 1))] = (uint32_t)jid;
			}
		}
		if (active && !dyn) {
			if (final_eob) { o_sfbt = 0; o_subc = (uint32_t)(b.total() - b.pos); }
			nxz_batch_result_t r;
			uint32_t spbc = job.src_len, subc = o_subc;
			if (final_eob && subc > 0xfff8) { const uint32_t drop = (subc - 0xfff8 + 7) / 8; spbc -= drop; subc -= drop * 8; }
			if (cc == 0 && !(final_eob && subc < 8)) cc = NXZ_CC_DATA_LENGTH;
			r.cc = cc; r.tpbc = (cc == 0 || cc == NXZ_CC_DATA_LENGTH) ? w.out : 0;
			r.tebc = o_rem; r.spbc = spbc; r.crc = 0; r.adler = 0;
			r.subc = subc; r.sfbt = o_sfbt | (final_eob ? 0x100u : 0);
			results[jid] = r;
		}
	}
}

__global__ void fixed_tables_kernel(uint8_t *ws)
{
	__shared__ uint8_t lens[320];
	const int lane = threadIdx.x;
	for (int i = lane; i < 288; i += 64) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
	if (lane < 30) lens[288 + lane] = 5;
	__syncthreads();
	build_tables(ws, lens, 288, 30, lane);
}

// ---- checksums of the outputs: one 256-thread workgroup per job (continues from in_crc / in_adler) ----
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
__device__ __forceinline__ uint32_t xpow8(uint32_t n)
{
	uint32_t r = 0x80000000u, sq = 0x00800000u;
	while (n) { if (n & 1) r = gf_mul(r, sq); sq = gf_mul(sq, sq); n >>= 1; }
	return r;
}

// x^(512 k) mod P for k = 0..1023 (compile time): the weight of a 64-byte slice that is followed by k more slices
constexpr uint32_t cgf_mul(uint32_t a, uint32_t b)
{
	uint32_t r = 0;
	for (int i = 0; i < 32; i++) {
		if (b & 0x80000000u) r ^= a;
		a = (a >> 1) ^ ((a & 1) ? 0xedb88320u : 0);
		b <<= 1;
	}
	return r;
}
struct PowTab { uint32_t v[1024]; };
constexpr PowTab make_pow()
{
	PowTab p{};
	uint32_t m = 0x00800000u;                 // x^8
	for (int k = 0; k < 6; k++) m = cgf_mul(m, m);   // x^512
	p.v[0] = 0x80000000u;
	for (int i = 1; i < 1024; i++) p.v[i] = cgf_mul(p.v[i - 1], m);
	return p;
}
__device__ const PowTab CRC_POW = make_pow();
constexpr uint32_t cxpow8(uint32_t n)            // x^(8n) mod P, compile time
{
	uint32_t r = 0x80000000u, sq = 0x00800000u;
	while (n) { if (n & 1) r = cgf_mul(r, sq); sq = cgf_mul(sq, sq); n >>= 1; }
	return r;
}

// CRC-32 and Adler-32 of every job's output (results[].tpbc bytes at job.dst), continued from
// job.in_crc / job.in_adler.  One workgroup per job; the output is cut into 64 KiB chunks and
// those into 64-byte slices: a thread feeds its slices through a slice-by-4 table (16-byte
// loads, v_dot4 for the Adler sums), weighs a slice's raw CRC by x^(8 * bytes that follow it in
// the chunk), and the XOR of all of them is the chunk's raw CRC (same scheme as the deflate
// kernel of round 1).  Outputs that are not 16-byte aligned take the bytewise path.
// WRAP (function code 0x1e: copy + checksums from the initial values, /root/reference lib/nx_deflate.c:1774, lib/nx_zlib.c:1398-1443):
// the same pass over job.src that also stores what it reads to job.dst and writes the whole result record.  (Round 3's
// wrap kernel walked 256 bytes a thread through a one-byte table and let thread 0 combine 256 slices with a
// bit-serial multiply each: 0.8 TB/s.)
// MODE 2 (round 6): checksums AND the output on its way to the caller's target (targets[i], pinned host memory: the rounds of
// nxu_run_job) -- one pass and one launch instead of checksums, then a copy kernel; the result record keeps what the inflate
// kernel wrote, a job that failed is not copied.
template <int MODE>
__global__ __launch_bounds__(256) void cksum_kernel(const nxz_batch_job_t *__restrict__ jobs, nxz_batch_result_t *__restrict__ results, uint8_t *const *__restrict__ targets)
{
	constexpr bool WRAP = MODE == 1;
	__shared__ uint32_t T[1024];            // T[k*256 + i] = i advanced by k+1 zero bytes
	__shared__ uint32_t red[3][256];
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	nxz_batch_job_t job = jobs[blockIdx.x];
	const uint32_t n = WRAP ? job.src_len : results[blockIdx.x].tpbc;
	if (WRAP) {
		if (n > job.dst_cap) {
			if (t == 0) { nxz_batch_result_t r = {NXZ_CC_TARGET_SPACE, 0, 0, 0, 0, 0, 0, 0}; results[blockIdx.x] = r; }
			return;
		}
		job.in_crc = 0; job.in_adler = 1;       // (WRAP ignores what the job brings: the caller combines, lib/nx_deflate.c:1565-1578)
	}
	uint8_t *const wdst = MODE == 2 ? targets[blockIdx.x] : job.dst;
	bool copy = WRAP;                                       // store what is read
	if (MODE == 2) { const uint32_t cc = results[blockIdx.x].cc; copy = cc == 0 || cc == NXZ_CC_DATA_LENGTH; }
	const uint32_t rd_cap = WRAP ? n : job.dst_cap;          // bytes that may be read at p
	auto finish = [&](uint32_t crc, uint32_t adler) {
		if (WRAP) { nxz_batch_result_t r; r.cc = 0; r.tpbc = n; r.tebc = 0; r.spbc = n; r.crc = crc; r.adler = adler; r.subc = 0; r.sfbt = 0; results[blockIdx.x] = r; }
		else { results[blockIdx.x].crc = crc; results[blockIdx.x].adler = adler; }
	};
	for (int k = 0; k < 4; k++) {
		uint32_t c = t;
		for (int i = 0; i < 8 * (k + 1); i++) c = (c >> 1) ^ ((c & 1) ? 0xedb88320u : 0);
		T[k * 256 + t] = c;
	}
	__syncthreads();
	const uint8_t *p = WRAP ? job.src : job.dst;
	if (((uintptr_t)p & 15) || (copy && ((uintptr_t)wdst & 15))) {
		// bytewise: 256 contiguous pieces
		uint32_t per = ((n + 255) / 256 + 15) & ~15u;
		uint32_t lo = (uint32_t)t * per, hi = lo + per < n ? lo + per : n;
		uint32_t crc = 0, s1 = 0, s2 = 0;
		for (uint32_t i = lo; i < hi; i++) {
			uint32_t byte = p[i];
			if (copy) wdst[i] = (uint8_t)byte;
			crc = T[(crc ^ byte) & 0xff] ^ (crc >> 8);
			s1 += byte; s2 += s1;
			if ((i & 0xfff) == 0xfff) { s1 %= 65521u; s2 %= 65521u; }
		}
		red[0][t] = crc; red[1][t] = s1 % 65521u; red[2][t] = s2 % 65521u;
		__syncthreads();
		if (t == 0) {
			uint32_t c = job.in_crc ^ 0xffffffffu, a1 = job.in_adler & 0xffff, a2 = job.in_adler >> 16;
			uint32_t xp = xpow8(per);
			for (uint32_t k = 0; k < 256; k++) {
				uint32_t klo = k * per; if (klo >= n) break;
				uint32_t len = klo + per < n ? per : n - klo;
				c = gf_mul(c, len == per ? xp : xpow8(len)) ^ red[0][k];
				a2 = (uint32_t)((a2 + (uint64_t)len * a1 + red[2][k]) % 65521u);
				a1 = (a1 + red[1][k]) % 65521u;
			}
			finish(c ^ 0xffffffffu, (a2 << 16) | a1);
		}
		return;
	}
	uint32_t c_run = job.in_crc ^ 0xffffffffu, a1_run = job.in_adler & 0xffff, a2_run = job.in_adler >> 16;   // thread 0
	for (uint32_t base = 0; base < n; base += 65536) {
		const uint32_t len = n - base < 65536 ? n - base : 65536;
		const uint32_t K1 = (len - 1) >> 6, r = len - K1 * 64;        // last slice and its bytes (1..64)
		uint32_t crc_w = 0, tailcrc = 0, s1 = 0, s2 = 0;
		for (uint32_t sl = t; sl <= K1; sl += 256) {
			const uint4 *sp = (const uint4 *)(p + base + (size_t)sl * 64);
			const uint32_t nb = sl < K1 ? 64 : r;                     // data bytes in this slice
			uint4 q[4];
#pragma unroll
			for (int k = 0; k < 4; k++) {
				q[k] = make_uint4(0, 0, 0, 0);
				if (16u * k < nb) {
					if ((size_t)base + (size_t)sl * 64 + 16 * k + 16 <= rd_cap) {
						q[k] = sp[k];
						if (copy) ((uint4 *)(wdst + base + (size_t)sl * 64))[k] = q[k];
					} else {                                        // never read past the caller's buffer
						const uint8_t *bp = (const uint8_t *)&sp[k];
						uint32_t ww[4] = { 0, 0, 0, 0 };
#pragma unroll
						for (int b = 0; b < 16; b++)
							if (16u * k + b < nb) {
								ww[b >> 2] |= (uint32_t)bp[b] << (8 * (b & 3));
								if (copy) wdst[base + (size_t)sl * 64 + 16 * k + b] = bp[b];
							}
						q[k] = make_uint4(ww[0], ww[1], ww[2], ww[3]);
					}
				}
			}
			uint32_t *w = (uint32_t *)q;
			if (nb < 64) {
#pragma unroll
				for (int k = 0; k < 16; k++) {
					if (4u * k >= nb) w[k] = 0;
					else if (4u * k + 4 > nb) w[k] &= (1u << (8 * (nb & 3))) - 1;
				}
			}
			uint32_t S = 0, Wt = 0;
#pragma unroll
			for (int k = 0; k < 16; k++) {
				S = __builtin_amdgcn_udot4(w[k], 0x01010101u, S, false);
				Wt = __builtin_amdgcn_udot4(w[k], 0x03020100u + 0x04040404u * k, Wt, false);
			}
			s1 += S;
			s2 = (s2 + S * (len - sl * 64) - Wt) % 65521u;
			uint32_t crc = 0;
			const uint32_t nd = nb >> 2;                               // full dwords
#pragma unroll
			for (int k = 0; k < 16; k++) {
				const uint32_t c = crc ^ w[k];
				const uint32_t nc = T[768 + (c & 0xff)] ^ T[512 + ((c >> 8) & 0xff)] ^ T[256 + ((c >> 16) & 0xff)] ^ T[c >> 24];
				crc = (uint32_t)k < nd ? nc : crc;
			}
			if (nb & 3) {
				uint32_t v = 0;                                     // w[nd] without a dynamic index (that would put w[] in scratch)
#pragma unroll
				for (int k = 0; k < 16; k++) v = (uint32_t)k == nd ? w[k] : v;
				for (uint32_t k = 0; k < (nb & 3); k++) crc = T[(crc ^ (v >> (8 * k))) & 0xff] ^ (crc >> 8);
			}
			if (sl == K1) tailcrc = crc;
			else crc_w ^= gf_mul(crc, CRC_POW.v[K1 - 1 - sl]);
		}
		for (int o = 32; o > 0; o >>= 1) {
			crc_w ^= __shfl_down(crc_w, o, 64);
			tailcrc ^= __shfl_down(tailcrc, o, 64);
			s1 += __shfl_down(s1, o, 64);
			s2 += __shfl_down(s2, o, 64);
		}
		__syncthreads();
		if (lane == 0) { red[0][wave] = crc_w; red[0][4 + wave] = tailcrc; red[1][wave] = s1; red[2][wave] = s2 % 65521u; }
		__syncthreads();
		if (t == 0) {
			const uint32_t full = red[0][0] ^ red[0][1] ^ red[0][2] ^ red[0][3];
			const uint32_t tail = red[0][4] ^ red[0][5] ^ red[0][6] ^ red[0][7];
			// (whole chunks, whole slices: the two powers are constants -- computing them is 2000 instructions of one thread)
			constexpr uint32_t X64 = cxpow8(64), X64K = cxpow8(65536);
			const uint32_t chunk = gf_mul(full, r == 64 ? X64 : xpow8(r)) ^ tail;          // raw CRC of the chunk
			c_run = gf_mul(c_run, len == 65536 ? X64K : xpow8(len)) ^ chunk;
			const uint32_t b1 = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) % 65521u;
			const uint32_t b2 = (red[2][0] + red[2][1] + red[2][2] + red[2][3]) % 65521u;
			a2_run = (uint32_t)((a2_run + (uint64_t)len * a1_run + b2) % 65521u);
			a1_run = (a1_run + b1) % 65521u;
		}
	}
	if (t == 0) finish(c_run ^ 0xffffffffu, (a2_run << 16) | a1_run);
}

} // namespace nxzl

// the WRAP function code for a batch (nxz_engine.cpp nxz_batch_wrap)
extern "C" int nxz_launch_wrap_sliced(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxzl::cksum_kernel<1>, dim3((unsigned)n), dim3(256), 0, stream, jobs, results, (uint8_t *const *)nullptr);
	return (int)hipGetLastError();
}

#define NXZ_LANES_MAX_GRID (1024u * NXZ_LANES_WPE)
// resident wavefronts (each works its way through groups of 64 streams): NXZ_LANES_GRID overrides, for measurements
static unsigned lanes_max_grid(void)
{
	static const unsigned v = [] { const char *e = getenv("NXZ_LANES_GRID"); unsigned g = e ? (unsigned)atoi(e) : NXZ_LANES_MAX_GRID; return g < 1 ? 1u : g > NXZ_LANES_MAX_GRID ? NXZ_LANES_MAX_GRID : g; }();
	return v;
}

// workspace bytes a batch of n streams needs (grows with n up to the largest grid)
static unsigned lanes_per_wave(size_t n)
{
	static const int env = getenv("NXZ_LANES_PER_WAVE") ? atoi(getenv("NXZ_LANES_PER_WAVE")) : 0;   // measurements: 32 or 64
	if (env == 8 || env == 16 || env == 32 || env == 64) return (unsigned)env;
	return n <= 64 * (size_t)(NXZ_LANES_MAX_GRID / 2) ? 32u : 64u;     /* (own fixed-code streams: 54 against 50 GiB/s at 98304, 67 against 62 at 131072; 58 against 82 at 196608) */
}
static size_t lanes_tables_bytes(size_t n)
{
	size_t groups = (n + lanes_per_wave(n) - 1) / lanes_per_wave(n);
	size_t grid = groups < NXZ_LANES_MAX_GRID ? groups : NXZ_LANES_MAX_GRID;
	return ((grid * 65 + 1) * nxzl::WS_BYTES + 255) & ~(size_t)255;
}
static size_t lanes_sort_temp_bytes(size_t n)
{
	size_t t = 0;
	(void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, t, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n);
	return (t + 255) & ~(size_t)255;
}
// the tables' slots, then what ordering the jobs by length needs: keys and indices in and out, the sort's own room; then the
// word by which the fixed-code kernel asks for the general one
static size_t lanes_bail_offset(size_t n)
{
	return lanes_tables_bytes(n) + 4 * ((n * sizeof(uint32_t) + 255) & ~(size_t)255) + lanes_sort_temp_bytes(n);
}
extern "C" size_t nxz_inflate_lanes_workspace(size_t n)
{
	return lanes_bail_offset(n) + 256 + ((n * sizeof(uint32_t) + 255) & ~(size_t)255);    // (the count of streams handed back, then their indices)
}
// (diagnostic / tests: how many streams the fixed-code kernel of the last batch on this workspace handed back; the caller has waited for the stream)
extern "C" int nxz_inflate_lanes_handed_back(const uint8_t *workspace, size_t n, uint32_t *count)
{
	return (int)hipMemcpy(count, workspace + lanes_bail_offset(n), sizeof(uint32_t), hipMemcpyDeviceToHost);
}

extern "C" int nxz_launch_inflate_order_only(const nxz_batch_job_t *jobs, size_t nslots, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io,
					     const uint32_t *order, hipStream_t stream);
namespace nxzl {
__global__ void order_keys_kernel(const nxz_batch_job_t *__restrict__ jobs, uint32_t n, uint32_t *__restrict__ keys, uint32_t *__restrict__ idx)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { keys[i] = jobs[i].src_len >> 11; idx[i] = i; }          // (2 KiB classes: streams of one class stay in the caller's order, neighbours in memory)
}
}

// the jobs by falling source length, for a launch that ends with its slowest stream (this file's lane kernels, and the
// stream-per-wave kernel's batches of a few rounds of wavefronts): keys, indices, the sort's own room in `workspace`
extern "C" size_t nxz_order_workspace(size_t n)
{
	return 4 * ((n * sizeof(uint32_t) + 255) & ~(size_t)255) + lanes_sort_temp_bytes(n);
}
extern "C" const uint32_t *nxz_launch_order_by_length(const nxz_batch_job_t *jobs, size_t n, uint8_t *workspace, hipStream_t stream)
{
	if (n < 128 || n >= (1u << 31) || !workspace) return nullptr;
	const size_t arr = (n * sizeof(uint32_t) + 255) & ~(size_t)255;
	uint32_t *k_in = (uint32_t *)workspace, *k_out = (uint32_t *)(workspace + arr), *v_in = (uint32_t *)(workspace + 2 * arr), *v_out = (uint32_t *)(workspace + 3 * arr);
	size_t tb = lanes_sort_temp_bytes(n);
	hipLaunchKernelGGL(nxzl::order_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, jobs, (uint32_t)n, k_in, v_in);
	if (hipcub::DeviceRadixSort::SortPairsDescending(workspace + 4 * arr, tb, k_in, k_out, v_in, v_out, (int)n, 0, 21, stream) != hipSuccess) return nullptr;
	return v_out;
}

// checksums of the outputs, and the outputs to targets[i] in the same pass (the rounds of nxu_run_job: device buffers -> pinned host)
extern "C" int nxz_launch_cksum_copy(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, uint8_t *const *targets, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxzl::cksum_kernel<2>, dim3((unsigned)n), dim3(256), 0, stream, jobs, results, targets);
	return (int)hipGetLastError();
}

extern "C" int nxz_launch_cksum(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxzl::cksum_kernel<0>, dim3((unsigned)n), dim3(256), 0, stream, jobs, results, (uint8_t *const *)nullptr);
	return (int)hipGetLastError();
}

extern "C" int nxz_launch_inflate_lanes(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results,
					nxz_batch_dht_t *dht_io, uint8_t *workspace, int init_fixed, hipStream_t stream)
{
	if (!n) return 0;
	const unsigned pw = lanes_per_wave(n);
	size_t groups = (n + pw - 1) / pw;
	unsigned grid = (unsigned)(groups < lanes_max_grid() ? groups : lanes_max_grid());
	if (init_fixed & 1) hipLaunchKernelGGL(nxzl::fixed_tables_kernel, dim3(1), dim3(64), 0, stream, workspace);
	// the jobs by falling source length, always (NXZ_LANES_ORDER=0: never).  Earlier in round 4 only when the sampled lengths
	// differed eightfold (init_fixed bit 1, no longer looked at): ordering the bench's synthetic blocks, all of a size, cost
	// 5 % -- neighbours in memory no longer neighbours in a wavefront.  With the kernels' memory instructions where all lanes
	// pass together it pays everywhere: synthetic blocks 104.6 -> 112.7 GiB/s at 65 536 streams, 183.9 -> 187.7 at 2^18, the
	// corpus' fixed-code streams 68.8 -> 88.2 and 145.7 -> 163.7 (a wavefront takes as long as its longest stream).
	static const int order_env = getenv("NXZ_LANES_ORDER") ? atoi(getenv("NXZ_LANES_ORDER")) : -1;
	const bool ordered = order_env != 0;
	const uint32_t *order = ordered ? nxz_launch_order_by_length(jobs, n, workspace + lanes_tables_bytes(n), stream) : nullptr;
	// init_fixed bit 2: the sampled streams all begin with fixed-code or stored blocks: the kernel that does only those first, and
	// this one behind it for the batch that turns out to hold a dynamic block somewhere (NXZ_LANES_FIXED=0: never)
	// (NXZ_LANES_FIXED=2: always first, whatever the sample said -- the tests' way to put every stream through it)
	const char *fe = getenv("NXZ_LANES_FIXED");
	const int fixed_env = fe ? atoi(fe) : 1;
	const bool fixed_first = fixed_env != 0;
	if (fixed_env == 2) init_fixed |= 4;
	uint32_t *bail = nullptr;
	// streams the fixed-code kernel hands back: up to HANDBACK_WAVES of them a wavefront each (nxz_inflate.hip: 0.6-3 ms a stream,
	// thousands side by side, where a lane takes 40 ms), what is beyond that by the general lane kernel
	constexpr uint32_t HANDBACK_WAVES = 16384;
	const uint32_t hb_slots = (uint32_t)(n < HANDBACK_WAVES ? n : HANDBACK_WAVES);
	if (fixed_first && (init_fixed & 4)) {
		bail = (uint32_t *)(workspace + lanes_bail_offset(n));
		(void)hipMemsetAsync(bail, 0, sizeof(uint32_t), stream);
		(void)hipMemsetAsync(bail + 64, 0xff, (size_t)hb_slots * sizeof(uint32_t), stream);
		// (no table slots to hold: its grid is bounded by the wavefronts the CUs hold; NXZ_LANES_FIXED_GRID overrides, for measurements)
		static const unsigned fgmax = [] { const char *e = getenv("NXZ_LANES_FIXED_GRID"); return e && atoi(e) > 0 ? (unsigned)atoi(e) : 1024u * NXZ_LANES_FIXED_WPE; }();
		const unsigned fgrid = (unsigned)(groups < fgmax ? groups : fgmax);
		hipLaunchKernelGGL(nxzl::inflate_lanes_fixed_kernel, dim3(fgrid), dim3(64), 0, stream, jobs, n, results, order, pw, bail);
	}
	if (bail) {
		int rc = nxz_launch_inflate_order_only(jobs, hb_slots, results, dht_io, bail + 64, stream);
		if (rc) return rc;
	}
	hipLaunchKernelGGL(nxzl::inflate_lanes_kernel, dim3(grid), dim3(64), 0, stream, jobs, n, results, dht_io, workspace, workspace, order, pw, bail, bail ? hb_slots : 0u);
	hipLaunchKernelGGL(nxzl::cksum_kernel<0>, dim3((unsigned)n), dim3(256), 0, stream, jobs, results, (uint8_t *const *)nullptr);
	return (int)hipGetLastError();
}
