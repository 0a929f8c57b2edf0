// nxz_inflate_wg.hip -- batched DEFLATE decompression, one stream per WORKGROUP, the whole stream in LDS.
//
// Same engine function as nxz_inflate.hip / nxz_inflate_lanes.hip (GZIP_FC_DECOMPRESS, issued at /root/reference
// lib/nx_inflate.c:909-912; outputs inc_nx/nxu.h:403-541; CPU restatement: oracle/nxz_inflate.c) for the streams a batch is
// made of nearly always: a fresh stream (no resume state, no history) of at most 64 KiB of output and 64 KiB of source that
// runs to its final end-of-block.  Anything else -- an error of any kind, a stream that ends early, a target that is too
// small, resume fields -- is HANDED BACK (a list of job indices, as inflate_lanes_fixed_kernel does it) to the kernels that
// know every case; this one never reports an error itself.
//
// Why a workgroup per stream: a CU of gfx950 has 160 KiB of LDS, which holds a 64 KiB block's source AND its output AND its
// decode tables.  So the source is read from HBM once, coalesced; the output is written once, coalesced; every table look-up,
// every match copy and every bit of the stream in between is an LDS access -- the stream-per-lane kernels issue one
// 64-cache-line memory instruction per token and wait three quarters of their time for them, the stream-per-wavefront kernel
// spends thirty wave-instructions on a token.  Here 1024 lanes decode ONE Huffman-coded block side by side:
//   1. the block's header is read by one wavefront, the decode tables (10 / 9 root bits + sub-tables, 32-bit entries that
//      carry base and extra-bit count) are built by all;
//   2. the rest of the source is cut into pieces, a lane each.  Lane 0 starts at the block's first token, the others at a
//      guess; every lane decodes (lengths only) to the first token that starts in the next piece.  Huffman-coded data
//      synchronises itself, so most lanes END on a true token boundary although they began on a false one: in the next
//      round every lane begins where its neighbour ended, and the rounds go on until no lane's start moves (usually 2-4);
//      the lane that met the end-of-block code ends the block;
//   3. a prefix sum over the pieces' output counts gives every lane its place in the output; a last pass writes the
//      literals and parks every match as a 3-byte record in the first bytes of the room it will fill (two bitmaps: where
//      matches start, which bytes are not there yet);
//   4. when all blocks are decoded, every lane resolves the matches that start in its 64 bytes of the output, in order, each
//      as soon as its source bytes are there (the bitmap says so; the lowest unresolved match can always go);
//   5. the output leaves LDS 16 bytes a lane.  (CRC-32 / Adler-32: nxzl::cksum_kernel behind this one, as for the lane kernels.)
//
// Written against a small subset of the device language (barriers, ballots, shuffles, LDS atomics) so that
// tests/native/hip_cpu_shim.h can run a workgroup on the CPU, an OS thread per lane: tests/test_inflate_wg_sim.py.
#ifndef NXZ_CPU_SIM
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "nxz_device.h"
#define NXZ_SPIN_HINT() __builtin_amdgcn_s_sleep(1)
#define NXZ_WG_GLOBAL NXZ_GLOBAL_AS
#else
#include <stdint.h>
#include "../../include/nxz_engine.h"
#define NXZ_SPIN_HINT() sched_yield()
#define NXZ_WG_GLOBAL
#endif

namespace nxzw {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));     // 16 bytes a lane, in whatever address space

constexpr int NT = 1024, NW = NT / 64;
constexpr uint32_t OUT_MAX = 65536;
constexpr uint32_t SRC_MAX = 65600;                   // bytes of source in LDS (incl. up to 15 bytes in front: the load is 16-byte aligned)
constexpr uint32_t SRC_WORDS = SRC_MAX / 4 + 8;
constexpr int RL = 10, RD = 9;                        // root bits of the literal/length and distance tables
constexpr uint32_t LSUB = 384, DSUB = 256;            // sub-table entries (zlib's ENOUGH for 286 symbols, root 10: 1332 - 1024)
constexpr uint32_t NSUBMAX = (1u << RL) + (1u << RD);

// table entry: bits 0..4 code bits (root entries: of the whole code; a link: unused), 5..7 kind, 8..11 extra bits (a link: the
// sub-table's index bits), 16..31 value (literal, base length, base distance; a link: index of the sub-table in the array)
enum { K_INVALID = 0, K_LIT = 1, K_LEN = 2, K_EOB = 3, K_LINK = 4, K_DIST = 5 };
__device__ __forceinline__ constexpr uint32_t mk(uint32_t nb, uint32_t kind, uint32_t xb, uint32_t val) { return nb | kind << 5 | xb << 8 | val << 16; }
__device__ __forceinline__ uint32_t e_kind(uint32_t e) { return (e >> 5) & 7; }
__device__ __forceinline__ uint32_t e_xb(uint32_t e) { return (e >> 8) & 15; }

// piece flags (pend[] = end bit | flag << 24)
enum { F_OK = 0, F_EOB = 1, F_ERR = 2, F_RUNOUT = 3 };
// why a stream was handed back (dbg[reason]++)
enum { R_JOB = 1, R_HEADER = 2, R_STORED = 3, R_DHT = 4, R_TABLES = 5, R_ROUNDS = 6, R_NOEOB = 7, R_TOKEN = 8, R_SPACE = 9, R_DIST = 10 };

struct __attribute__((aligned(16))) Lds {
	uint32_t out[OUT_MAX / 4];
	uint32_t src[SRC_WORDS];
	uint32_t mstart[OUT_MAX / 32];      // bit p: a match starts at output byte p (its record stands there)
	uint32_t unres[OUT_MAX / 32];       // bit p: output byte p is part of a match that is not copied yet
	uint32_t lit[(1 << RL) + LSUB];
	uint32_t dist[(1 << RD) + DSUB];
	uint32_t pend[NSUBMAX];             // table build: sub-table bits per root index; the rounds: every piece's end | flag << 24
	uint16_t lcount[16], dcount[16];
	uint16_t lsym[288], dsym[32];
	uint8_t lens[320];
	uint32_t wsum[NW];
	// wave-uniform scalars, written by one thread in front of a barrier
	uint32_t jid, bail, pos, outn, bfinal, btype, st_len, hlit, hdist, firstbad, total;
};
static_assert(NT == (1 << RL) && NT >= (1 << RD), "a lane per root entry");
static_assert(sizeof(Lds) <= 163840, "the workgroup's LDS image must fit the CU's 160 KiB");

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

// 32 bits of the source at bit p (LDS; the array is zero behind the stream)
__device__ __forceinline__ uint32_t peek32(const Lds &L, uint32_t p)
{
	const uint32_t w = p >> 5;
	return __builtin_amdgcn_alignbit(L.src[w + 1], L.src[w], p & 31);
}

// ---- canonical code of n symbols (one wavefront; symbol i = row * 64 + lane): counts per length, the symbols in
// (length, symbol) order, every symbol's code (RFC 1951 3.2.2) -- by ballots, as nxzi::build ----
template <int ROWS>
__device__ __forceinline__ void canon(const uint8_t *len, int n, uint16_t *count, uint16_t *symtab, uint32_t (&l)[ROWS], uint32_t (&code)[ROWS], int lane)
{
	const uint64_t below = (1ull << lane) - 1;
#pragma unroll
	for (int r = 0; r < ROWS; r++) { const int i = r * 64 + lane; l[r] = i < n ? len[i] : 0; code[r] = 0; }
	uint32_t c = 0, prevcnt = 0, offs = 0;
	for (uint32_t b = 1; b <= 15; b++) {
		c = (c + prevcnt) << 1;
		uint32_t run = 0;
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			const uint64_t m = __ballot(l[r] == b);
			const uint32_t k = run + (uint32_t)__popcll(m & below);
			if (l[r] == b) { code[r] = c + k; symtab[offs + k] = (uint16_t)(r * 64 + lane); }
			run += (uint32_t)__popcll(m);
		}
		if (lane == 0) count[b] = (uint16_t)run;
		prevcnt = run; offs += run;
	}
	if (lane == 0) count[0] = 0;
}

// the root entry of index e: the code that e's low bits begin with, if it is no longer than RB bits (canonical walk)
template <int RB>
__device__ __forceinline__ bool root_walk(uint32_t e, const uint16_t *count, const uint16_t *symtab, uint32_t &sym, uint32_t &len)
{
	int code = 0, first = 0, index = 0;
	for (int b = 1; b <= RB; b++) {
		code |= (int)(e & 1); e >>= 1;
		const int c = count[b];
		if (code - c < first) { sym = symtab[index + (code - first)]; len = (uint32_t)b; return true; }
		index += c; first += c; first <<= 1; code <<= 1;
	}
	return false;
}
__device__ __forceinline__ uint32_t lit_entry(uint32_t sym, uint32_t len)
{
	if (sym < 256) return mk(len, K_LIT, 0, sym);
	if (sym == 256) return mk(len, K_EOB, 0, 0);
	if (sym >= 286) return 0;
	uint32_t base, xb;
	len_params(sym - 257, base, xb);
	return mk(len, K_LEN, xb, base);
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t len)
{
	if (sym >= 30) return 0;
	uint32_t base, xb;
	dist_params(sym, base, xb);
	return mk(len, K_DIST, xb, base);
}

// exclusive prefix sum over the workgroup (all threads call it); *total = the sum
__device__ __forceinline__ uint32_t block_scan(Lds &L, uint32_t v, uint32_t *total)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint32_t inc = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = __shfl_up(inc, (unsigned)d, 64);
		if (lane >= d) inc += o;
	}
	__syncthreads();                               // (wsum may still be read from the last scan)
	if (lane == 63) L.wsum[wave] = inc;
	__syncthreads();
	uint32_t base = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < NW; w++) { const uint32_t s = L.wsum[w]; if (w < wave) base += s; tot += s; }
	*total = tot;
	return base + inc - v;
}

// ---- one piece: tokens that start in [st, lim).  WRITE: literals to their place, matches parked as records. ----
template <bool WRITE>
__device__ __forceinline__ void decode_piece(Lds &L, uint32_t st, uint32_t lim, uint32_t T, uint32_t obase, uint32_t &en, uint32_t &no, uint32_t &fl)
{
	uint8_t *ob = (uint8_t *)L.out;
	uint32_t p = st, n = 0;
	fl = F_OK;
	while (p < lim) {
		const uint32_t w = p >> 5, sh = p & 31;
		const uint32_t a = L.src[w], b = L.src[w + 1], c = L.src[w + 2];
		const uint32_t lo = __builtin_amdgcn_alignbit(b, a, sh), hi = __builtin_amdgcn_alignbit(c, b, sh);
		uint32_t e = L.lit[lo & ((1u << RL) - 1)];
		if (e_kind(e) == K_LINK) e = L.lit[(e >> 16) + ((lo >> RL) & ((1u << e_xb(e)) - 1))];
		const uint32_t nb = e & 31, k = e_kind(e);
		if (k == K_LIT) {
			if (WRITE) ob[obase + n] = (uint8_t)(e >> 16);
			n++; p += nb;
			continue;
		}
		if (k == K_LEN) {
			const uint32_t x = e_xb(e);
			const uint32_t len = (e >> 16) + ((lo >> nb) & ((1u << x) - 1));
			const uint32_t q = nb + x;                                        // <= 20
			const uint32_t db = __builtin_amdgcn_alignbit(hi, lo, q);
			uint32_t d = L.dist[db & ((1u << RD) - 1)];
			if (e_kind(d) == K_LINK) d = L.dist[(d >> 16) + ((db >> RD) & ((1u << e_xb(d)) - 1))];
			if (e_kind(d) != K_DIST) { fl = F_ERR; break; }
			const uint32_t dl = d & 31, dx = e_xb(d);
			const uint32_t dist = (d >> 16) + ((db >> dl) & ((1u << dx) - 1));  // dl + dx <= 28
			p += q + dl + dx;
			if (WRITE) {
				const uint32_t at = obase + n;
				if (dist > at) { fl = F_ERR; break; }
				ob[at] = (uint8_t)(len - 3); ob[at + 1] = (uint8_t)(dist - 1); ob[at + 2] = (uint8_t)((dist - 1) >> 8);
				atomicOr(&L.mstart[at >> 5], 1u << (at & 31));
				const uint32_t last = at + len - 1, wa = at >> 5, wb = last >> 5;
				const uint32_t ma = ~0u << (at & 31), mb = ~0u >> (31 - (last & 31));
				if (wa == wb) atomicOr(&L.unres[wa], ma & mb);
				else {
					atomicOr(&L.unres[wa], ma);
					for (uint32_t i = wa + 1; i < wb; i++) atomicOr(&L.unres[i], ~0u);
					atomicOr(&L.unres[wb], mb);
				}
			}
			n += len;
			continue;
		}
		if (k == K_EOB) { p += nb; fl = F_EOB; break; }
		fl = F_ERR;
		break;
	}
	if (p > T) fl = F_RUNOUT;            // the last token reaches beyond the source
	en = p; no = n;
}

// is no byte of [a, e) part of a match that is still to be copied?  (a < e)
__device__ __forceinline__ bool range_there(const Lds &L, uint32_t a, uint32_t e)
{
	const uint32_t last = e - 1, wa = a >> 5, wb = last >> 5;
	const uint32_t ma = ~0u << (a & 31), mb = ~0u >> (31 - (last & 31));
	const volatile uint32_t *u = L.unres;
	if (wa == wb) return (u[wa] & ma & mb) == 0;
	uint32_t acc = (u[wa] & ma) | (u[wb] & mb);
	for (uint32_t i = wa + 1; i < wb; i++) acc |= u[i];
	return acc == 0;
}

// out[m, m + len) = out[m - dist, ...), bytes written in front serve as source (dist < len)
__device__ __forceinline__ void copy_match(Lds &L, uint32_t m, uint32_t len, uint32_t dist)
{
	uint8_t *ob = (uint8_t *)L.out;
	uint32_t *ow = L.out;
	uint32_t k = 0;
	const uint32_t head = (4 - (m & 3)) & 3;
	if (dist >= 4) {
		for (; k < head && k < len; k++) ob[m + k] = ob[m + k - dist];
		for (; k + 4 <= len; k += 4) {
			const uint32_t s = m + k - dist, si = s >> 2;
			const uint32_t lo = ow[si], hi = ow[si + 1];
			ow[(m + k) >> 2] = __builtin_amdgcn_alignbyte(hi, lo, s & 3);
		}
		for (; k < len; k++) ob[m + k] = ob[m + k - dist];
		return;
	}
	// period 1, 2 or 3: eight bytes of the pattern, read where byte k of the match is byte k % dist of the period
	const uint32_t p0 = ob[m - dist], p1 = dist > 1 ? ob[m - dist + 1] : p0, p2 = dist > 2 ? ob[m - dist + 2] : dist == 2 ? p0 : p0;
	uint64_t pat;
	if (dist == 3) {
		const uint64_t t = (uint64_t)p0 | (uint64_t)p1 << 8 | (uint64_t)p2 << 16;
		pat = t | t << 24 | t << 48;
	} else {
		const uint64_t t = (uint64_t)p0 | (uint64_t)p1 << 8;
		pat = t * 0x0001000100010001ull;
	}
	uint32_t r = 0;                                           // k % dist
	for (; k < head && k < len; k++) { ob[m + k] = (uint8_t)(pat >> (8 * r)); r = r + 1 == dist ? 0 : r + 1; }
	for (; k + 4 <= len; k += 4) {
		ow[(m + k) >> 2] = (uint32_t)(pat >> (8 * r));
		if (dist == 3) r = r == 2 ? 0 : r + 1;                  // (r + 4) % 3
	}
	for (; k < len; k++) { ob[m + k] = (uint8_t)(pat >> (8 * r)); r = r + 1 == dist ? 0 : r + 1; }
}

// ---- the header of a dynamic block, by wavefront 0 (the algorithm of nxzi::read_dht; the stream's bits come from LDS):
// code lengths into L.lens, L.hlit / L.hdist, L.pos behind the header.  false: not a header this kernel takes on. ----
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ bool read_header(Lds &L, uint32_t start, uint32_t T, int lane)
{
	if (start + 14 > T) return false;
	const uint32_t d0 = start >> 5;
	auto word_at = [&](uint32_t idx) -> uint32_t { return idx < SRC_WORDS ? L.src[idx] : 0; };
	const uint32_t R0 = word_at(d0 + lane), R1 = word_at(d0 + 64 + lane);
	auto word = [&](uint32_t i) -> uint32_t {
		const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)R0, (int)(i & 63)), hi = (uint32_t)__builtin_amdgcn_readlane((int)R1, (int)(i & 63));
		return i < 64 ? lo : hi;
	};
	const uint32_t v = peek32(L, start);
	const int hlit = (int)(v & 31) + 257, hdist = (int)((v >> 5) & 31) + 1, hclen = (int)((v >> 10) & 15) + 4;
	uint32_t pos = start + 14;
	if (hlit > 286 || hdist > 30) return false;
	if (pos + 3 * (uint32_t)hclen > T) return false;
	// the code-length code: lane i < hclen reads the i-th 3-bit length, which belongs to symbol order[i]
	uint32_t myl = 0;
	{
		const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
		uint32_t mine = 0, mysym = 0;
		if (lane < hclen) { mine = peek32(L, pos + 3 * (uint32_t)lane) & 7; mysym = order[lane]; }
		for (int i = 0; i < 19; i++) {
			const uint32_t sy = (uint32_t)__builtin_amdgcn_readlane((int)mysym, i), ln = (uint32_t)__builtin_amdgcn_readlane((int)mine, i);
			if (i < hclen && (uint32_t)lane == sy) myl = ln;
		}
	}
	pos += 3 * (uint32_t)hclen;
	// canonical codes by ranks (lane = symbol), then the look-up: entries `lane` and `lane + 64` of the 7-bit table
	uint32_t tlo = 0xff, thi = 0xff;
	{
		uint32_t c = 0, prevcnt = 0, kraft = 0, mycode = 0;
		for (uint32_t bl = 1; bl <= 7; bl++) {
			c = (c + prevcnt) << 1;
			const uint64_t m = __ballot(myl == bl);
			if (myl == bl) mycode = c + (uint32_t)__popcll(m & ((1ull << lane) - 1));
			prevcnt = (uint32_t)__popcll(m);
			kraft += prevcnt << (7 - bl);
		}
		if (kraft > 128) return false;
		const uint32_t myrev = myl ? __builtin_bitreverse32(mycode) >> (32 - myl) : 0;
		for (int sy = 0; sy < 19; sy++) {
			const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)myl, sy);
			const uint32_t rev = (uint32_t)__builtin_amdgcn_readlane((int)myrev, sy), mask = (1u << l) - 1;
			if (!l) continue;
			if (((uint32_t)lane & mask) == rev) tlo = (uint32_t)sy | (l << 5);
			if ((((uint32_t)lane + 64) & mask) == rev) thi = (uint32_t)sy | (l << 5);
		}
	}
	int n = 0, prev = 0;
	const int total = hlit + hdist;
	uint64_t win = 0;
	uint32_t wbits = 0;
	while (n < total) {
		if (pos + 1 > T) return false;
		if (wbits < 14) {
			const uint32_t o = uni(pos - d0 * 32), i = o >> 5, sh = o & 31;
			if (i + 1 >= 128) return false;                            // (a header is 2283 bits at most)
			win = (((uint64_t)word(i + 1) << 32) | word(i)) >> sh;
			wbits = 64 - sh;
		}
		const uint32_t bits = (uint32_t)win;
		const uint32_t k = bits & 127;
		const uint32_t e = k < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)tlo, (int)k) : (uint32_t)__builtin_amdgcn_readlane((int)thi, (int)(k - 64));
		if (e == 0xff) return false;
		const int sym = (int)(e & 31), len = (int)(e >> 5);
		if (pos + (uint32_t)len > T) return false;
		pos += (uint32_t)len;
		win >>= len; wbits -= (uint32_t)len;
		if (sym < 16) { if (lane == 0) L.lens[n] = (uint8_t)sym; n++; prev = sym; }
		else {
			const int eb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
			if (pos + (uint32_t)eb > T) return false;
			const int rep = (int)((bits >> len) & ((1u << eb) - 1)) + (sym == 18 ? 11 : 3);
			pos += (uint32_t)eb;
			win >>= eb; wbits -= (uint32_t)eb;
			int val = 0;
			if (sym == 16) { if (n == 0) return false; val = prev; }
			if (n + rep > total) return false;
			for (int q = lane; q < rep; q += 64) L.lens[n + q] = (uint8_t)val;
			n += rep;
			if (sym != 16) prev = 0;
		}
	}
	if (lane == 0) { L.hlit = (uint32_t)hlit; L.hdist = (uint32_t)hdist; L.pos = pos; }
	return true;
}

// PROF (NXZ_WG_PROF=1, measurements): thread 0's clock at the ends of the phases, summed over the launch's streams in prof[]:
// 0 load, 1 block headers, 2 tables, 3 the first pass, 4 the later rounds, 5 prefix sum + the writing pass, 6 matches, 7 out;
// 8 rounds, 9 streams, 10 coded blocks, 11 pieces
enum { P_LOAD, P_HEADER, P_TABLES, P_FIRST, P_ROUNDS, P_WRITE, P_MATCH, P_OUT, P_NROUNDS, P_STREAMS, P_BLOCKS, P_PIECES, P_N };
template <bool PROF>
__global__ __launch_bounds__(NT) void inflate_wg_kernel(const nxz_batch_job_t *__restrict__ jobs, uint32_t n, nxz_batch_result_t *__restrict__ results,
							 const uint32_t *__restrict__ order, uint32_t *__restrict__ ctr, uint32_t *__restrict__ bail,
							 uint32_t pmin_bits, uint32_t *__restrict__ dbg, unsigned long long *__restrict__ prof)
{
	__shared__ Lds L;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint8_t *ob = (uint8_t *)L.out;
	const uint8_t *sb = (const uint8_t *)L.src;
	unsigned long long pacc[P_N] = { 0 }, tprev = 0;
#define WGPROF(idx) do { if (PROF && tid == 0) { const unsigned long long now_ = (unsigned long long)clock64(); pacc[idx] += now_ - tprev; tprev = now_; } } while (0)
#define WGCOUNT(idx, v) do { if (PROF && tid == 0) pacc[idx] += (unsigned long long)(v); } while (0)
	if (PROF && tid == 0) tprev = (unsigned long long)clock64();

	for (;;) {
		// ---- next stream ----
		__syncthreads();
		if (tid == 0) {
			const uint32_t k = atomicAdd(ctr, 1u);
			L.jid = k < n ? (order ? order[k] : k) : 0xffffffffu;
			L.bail = 0; L.outn = 0;
		}
		__syncthreads();
		const uint32_t jid = L.jid;
		if (jid == 0xffffffffu) break;
		const nxz_batch_job_t job = jobs[jid];
		const uint32_t off = (uint32_t)((uintptr_t)job.src & 15);
		const uint32_t nbytes = off + job.src_len;                     // bytes of the LDS image that belong to the stream's granules
		const uint32_t T = nbytes * 8;
		const bool takes = job.resume == 0 && job.hist_len == 0 && job.src_len > 0 && nbytes <= SRC_MAX && ((uintptr_t)job.dst & 15) == 0 &&
				   (job.reserved & NXZ_JOB_SUSPEND_WHEN_FULL) == 0;
		if (!takes) {
			if (tid == 0) { const uint32_t at = atomicAdd(bail, 1u); bail[64 + at] = jid; if (dbg) atomicAdd(&dbg[R_JOB], 1u); }
			continue;
		}
		const uint32_t cap = job.dst_cap < OUT_MAX ? job.dst_cap : OUT_MAX;
		// ---- the source into LDS (16 bytes a lane, zeros behind it), the bitmaps cleared ----
		{
			const NXZ_WG_GLOBAL v4u *g = (const NXZ_WG_GLOBAL v4u *)(job.src - off);
			const uint32_t full = nbytes >> 4, chunks = (nbytes + 15) >> 4;
			v4u *ls = (v4u *)L.src;
			for (uint32_t i = tid; i < chunks + 2 && i < SRC_WORDS / 4; i += NT) {
				v4u v = { 0, 0, 0, 0 };
				if (i < chunks) {
					v = g[i];
					if (i >= full) {                                       // the stream's last bytes: what follows them in the granule counts as zero
						const uint32_t keep = nbytes & 15;
						uint32_t wv[4] = { v.x, v.y, v.z, v.w };
						for (uint32_t q = 0; q < 4; q++) {
							if (4 * q >= keep) wv[q] = 0;
							else if (4 * q + 4 > keep) wv[q] &= (1u << (8 * (keep & 3))) - 1;
						}
						v.x = wv[0]; v.y = wv[1]; v.z = wv[2]; v.w = wv[3];
					}
				}
				ls[i] = v;
			}
			v4u *b0 = (v4u *)L.mstart, *b1 = (v4u *)L.unres;
			const v4u z = { 0, 0, 0, 0 };
			for (uint32_t i = tid; i < OUT_MAX / 32 / 4; i += NT) { b0[i] = z; b1[i] = z; }
			if (tid == 0) L.pos = off * 8;
		}
		WGPROF(P_LOAD);
		WGCOUNT(P_STREAMS, 1);

		// ---- block after block ----
		bool done = false;
		for (;;) {
			__syncthreads();
			if (tid == 0) {
				uint32_t p = L.pos;
				if (p + 3 > T) L.bail = R_HEADER;
				else {
					const uint32_t v = peek32(L, p);
					L.bfinal = v & 1; L.btype = (v >> 1) & 3;
					p += 3;
					if (L.btype == 0) {
						p = (p + 7) & ~7u;
						if (p + 32 > T) L.bail = R_STORED;
						else {
							const uint32_t w = peek32(L, p), len = w & 0xffff;
							p += 32;
							if (((w >> 16) ^ len) != 0xffff || p + 8 * len > T || len > cap - L.outn) L.bail = R_STORED;
							L.st_len = len;
						}
					} else if (L.btype == 3) L.bail = R_HEADER;
					L.pos = p;
				}
			}
			__syncthreads();
			if (L.bail) break;
			const uint32_t btype = L.btype, bfinal = L.bfinal;
			if (btype == 0) {
				const uint32_t len = L.st_len, from = L.pos >> 3, to = L.outn;
				for (uint32_t i = tid; i < len; i += NT) ob[to + i] = sb[from + i];
				__syncthreads();
				if (tid == 0) { L.pos += 8 * len; L.outn += len; }
				if (bfinal) { done = true; break; }
				continue;
			}
			// ---- code lengths ----
			if (btype == 2) {
				if (wave == 0) {
					const bool ok = read_header(L, L.pos, T, lane);
					if (!ok && lane == 0) L.bail = R_DHT;
				}
			} else {
				for (int i = tid; i < 320; i += NT) L.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5);
				if (tid == 0) { L.hlit = 288; L.hdist = 30; }
			}
			__syncthreads();
			if (L.bail) break;
			WGPROF(P_HEADER);
			WGCOUNT(P_BLOCKS, 1);
			const int hlit = (int)L.hlit, hdist = (int)L.hdist;
			// (the checks of nxzi::read_dht: an end-of-block code, no code over-subscribed)
			if (btype == 2 && wave == 0) {
				uint32_t k1 = 0, k2 = 0;
				for (int i = lane; i < hlit; i += 64) if (L.lens[i]) k1 += 1u << (15 - L.lens[i]);
				if (lane < hdist && L.lens[hlit + lane]) k2 = 1u << (15 - L.lens[hlit + lane]);
				for (int o = 32; o > 0; o >>= 1) { k1 += __shfl(k1, lane ^ o, 64); k2 += __shfl(k2, lane ^ o, 64); }
				if (lane == 0 && (L.lens[256] == 0 || k1 > (1u << 15) || k2 > (1u << 15))) L.bail = R_DHT;
			}
			// ---- tables ----
			uint32_t cl[5], ccode[5];                                 // wave 0: literal/length symbols, wave 1: distance symbols (row 0)
			for (uint32_t i = tid; i < NSUBMAX; i += NT) L.pend[i] = 0;
			for (uint32_t i = tid; i < LSUB; i += NT) L.lit[(1u << RL) + i] = 0;
			for (uint32_t i = tid; i < DSUB; i += NT) L.dist[(1u << RD) + i] = 0;
			if (wave == 0) canon<5>(L.lens, hlit, L.lcount, L.lsym, cl, ccode, lane);
			else if (wave == 1) { uint32_t l1[1], c1[1]; canon<1>(L.lens + hlit, hdist, L.dcount, L.dsym, l1, c1, lane); cl[0] = l1[0]; ccode[0] = c1[0]; }
			__syncthreads();
			if (L.bail) break;
			{
				uint32_t sym, len;
				L.lit[tid] = root_walk<RL>((uint32_t)tid, L.lcount, L.lsym, sym, len) ? lit_entry(sym, len) : 0;
				if (tid < (1 << RD)) L.dist[tid] = root_walk<RD>((uint32_t)tid, L.dcount, L.dsym, sym, len) ? dist_entry(sym, len) : 0;
			}
			// codes longer than the root: how many index bits the sub-table behind their root index needs
			if (wave == 0) {
#pragma unroll
				for (int r = 0; r < 5; r++) if (cl[r] > (uint32_t)RL) {
					const uint32_t lsb = __builtin_bitreverse32(ccode[r]) >> (32 - cl[r]);
					atomicMax(&L.pend[lsb & ((1u << RL) - 1)], cl[r] - RL);
				}
			} else if (wave == 1 && cl[0] > (uint32_t)RD) {
				const uint32_t lsb = __builtin_bitreverse32(ccode[0]) >> (32 - cl[0]);
				atomicMax(&L.pend[(1u << RL) + (lsb & ((1u << RD) - 1))], cl[0] - RD);
			}
			__syncthreads();
			{
				const uint32_t sl = L.pend[tid], sd = tid < (1 << RD) ? L.pend[(1u << RL) + tid] : 0;
				uint32_t totl, totd;
				const uint32_t ol = block_scan(L, sl ? 1u << sl : 0, &totl);
				const uint32_t od = block_scan(L, sd ? 1u << sd : 0, &totd);
				if (sl) L.lit[tid] = mk(0, K_LINK, sl, (1u << RL) + ol);
				if (sd) L.dist[tid] = mk(0, K_LINK, sd, (1u << RD) + od);
				if (tid == 0 && (totl > LSUB || totd > DSUB)) L.bail = R_TABLES;
			}
			__syncthreads();
			if (L.bail) break;
			if (wave == 0) {
#pragma unroll
				for (int r = 0; r < 5; r++) if (cl[r] > (uint32_t)RL) {
					const uint32_t lsb = __builtin_bitreverse32(ccode[r]) >> (32 - cl[r]);
					const uint32_t link = L.lit[lsb & ((1u << RL) - 1)], rem = cl[r] - RL;
					const uint32_t ent = lit_entry((uint32_t)(r * 64 + lane), cl[r]);
					for (uint32_t k = lsb >> RL; k < (1u << e_xb(link)); k += 1u << rem) L.lit[(link >> 16) + k] = ent;
				}
			} else if (wave == 1 && cl[0] > (uint32_t)RD) {
				const uint32_t lsb = __builtin_bitreverse32(ccode[0]) >> (32 - cl[0]);
				const uint32_t link = L.dist[lsb & ((1u << RD) - 1)], rem = cl[0] - RD;
				const uint32_t ent = dist_entry((uint32_t)lane, cl[0]);
				for (uint32_t k = lsb >> RD; k < (1u << e_xb(link)); k += 1u << rem) L.dist[(link >> 16) + k] = ent;
			}
			__syncthreads();
			WGPROF(P_TABLES);

			// ---- the pieces: rounds until every lane starts where its neighbour ended ----
			const uint32_t cur = L.pos, R = T - cur;
			uint32_t np0 = R / (pmin_bits ? pmin_bits : 512u);
			np0 = np0 < 1 ? 1 : np0 > (uint32_t)NT ? (uint32_t)NT : np0;
			const uint32_t pdw = (((R + np0 - 1) / np0 + 31) >> 5) | 1;          // dwords a piece, odd: neighbours begin in different LDS banks
			const uint32_t P = pdw * 32;
			const uint32_t NP = (R + P - 1) / P ? (R + P - 1) / P : 1;
			const bool active = (uint32_t)tid < NP;
			const uint32_t g = cur + (uint32_t)tid * P;
			const uint32_t lim = (uint32_t)tid + 1 == NP ? T : g + P;
			uint32_t st = g, en = 0, no = 0, fl = F_ERR;
			if (active) { decode_piece<false>(L, st, lim, T, 0, en, no, fl); L.pend[tid] = en | fl << 24; }
			if (tid == 0) L.firstbad = NP;
			WGPROF(P_FIRST);
			WGCOUNT(P_PIECES, NP);
			uint32_t rounds = 0;
			for (;;) {
				__syncthreads();
				bool redo = false;
				if (active && tid > 0) {
					const uint32_t pe = L.pend[tid - 1];
					if ((pe >> 24) == F_OK && (pe & 0xffffff) != st) { st = pe & 0xffffff; redo = true; }
				}
				if (!__syncthreads_or(redo)) break;
				if (++rounds > 64) { if (tid == 0) L.bail = R_ROUNDS; break; }
				if (redo) { decode_piece<false>(L, st, lim, T, 0, en, no, fl); L.pend[tid] = en | fl << 24; }
			}
			__syncthreads();
			WGPROF(P_ROUNDS);
			WGCOUNT(P_NROUNDS, rounds);
			if (L.bail) break;
			if (active && fl != F_OK) atomicMin(&L.firstbad, (uint32_t)tid);
			__syncthreads();
			const uint32_t B = L.firstbad;
			if (B >= NP) { if (tid == 0) L.bail = R_NOEOB; __syncthreads(); break; }
			{
				const uint32_t fb = L.pend[B] >> 24;
				uint32_t tot;
				const uint32_t obase = block_scan(L, (uint32_t)tid <= B ? no : 0, &tot);
				const uint32_t outn = L.outn;
				if (fb != F_EOB || tot > cap - outn) { if (tid == 0) L.bail = fb != F_EOB ? R_TOKEN : R_SPACE; __syncthreads(); break; }
				if ((uint32_t)tid <= B) {
					uint32_t en2, no2, fl2;
					decode_piece<true>(L, st, lim, T, outn + obase, en2, no2, fl2);
					if (fl2 != fl || no2 != no) L.bail = R_DIST;                 // (a distance beyond the output so far)
				}
				__syncthreads();
				if (L.bail) break;
				if (tid == 0) { L.outn = outn + tot; L.pos = L.pend[B] & 0xffffff; }
				WGPROF(P_WRITE);
			}
			if (bfinal) { done = true; break; }
		}
		__syncthreads();
		if (!done || L.bail) {
			if (tid == 0) {
				const uint32_t at = atomicAdd(bail, 1u);
				bail[64 + at] = jid;
				if (dbg) atomicAdd(&dbg[L.bail < 16 ? L.bail : 0], 1u);
			}
			continue;
		}
		// ---- matches: every lane those that start in its 64 bytes, each as soon as its source is there ----
		const uint32_t outn = L.outn;
		{
			uint32_t mw0 = L.mstart[2 * tid], mw1 = L.mstart[2 * tid + 1];
			uint32_t m = 0, len = 0, dist = 0;
			bool have = false;
			while (mw0 | mw1 | (uint32_t)have) {
				if (!have) {
					const uint32_t h = mw0 ? 0 : 1, bits = mw0 ? mw0 : mw1, bit = (uint32_t)__builtin_ctz(bits);
					m = (2 * (uint32_t)tid + h) * 32 + bit;
					if (h) mw1 &= mw1 - 1; else mw0 &= mw0 - 1;
					len = (uint32_t)ob[m] + 3; dist = ((uint32_t)ob[m + 1] | (uint32_t)ob[m + 2] << 8) + 1;
					have = true;
				}
				const uint32_t a = m - dist, e = a + len < m ? a + len : m;
				if (range_there(L, a, e)) {
					__threadfence_block();
					copy_match(L, m, len, dist);
					__threadfence_block();
					const uint32_t last = m + len - 1, wa = m >> 5, wb = last >> 5;
					const uint32_t ma = ~0u << (m & 31), mb = ~0u >> (31 - (last & 31));
					if (wa == wb) atomicAnd(&L.unres[wa], ~(ma & mb));
					else {
						atomicAnd(&L.unres[wa], ~ma);
						for (uint32_t i = wa + 1; i < wb; i++) atomicAnd(&L.unres[i], 0u);
						atomicAnd(&L.unres[wb], ~mb);
					}
					have = false;
				} else NXZ_SPIN_HINT();
			}
		}
		__syncthreads();
		WGPROF(P_MATCH);
		// ---- out ----
		{
			NXZ_WG_GLOBAL v4u *gd = (NXZ_WG_GLOBAL v4u *)job.dst;
			const v4u *lo = (const v4u *)L.out;
			const uint32_t full = outn >> 4;
			for (uint32_t i = tid; i < full; i += NT) gd[i] = lo[i];
			if ((uint32_t)tid < (outn & 15)) ((NXZ_WG_GLOBAL uint8_t *)job.dst)[full * 16 + tid] = ob[full * 16 + tid];
			if (tid == 0) {
				// (the record of nxzl::inflate_lanes_kernel for a stream that ran to its final end-of-block)
				nxz_batch_result_t r;
				uint32_t spbc = job.src_len, subc = T - L.pos;
				if (subc > 0xfff8) { const uint32_t drop = (subc - 0xfff8 + 7) / 8; spbc -= drop; subc -= drop * 8; }
				r.cc = subc < 8 ? 0 : NXZ_CC_DATA_LENGTH;
				r.tpbc = outn; r.tebc = 0; r.spbc = spbc; r.crc = 0; r.adler = 0; r.subc = subc; r.sfbt = 0x100u;
				results[jid] = r;
			}
		}
		WGPROF(P_OUT);
	}
	if (PROF && tid == 0) for (int i = 0; i < P_N; i++) atomicAdd(&prof[i], pacc[i]);
#undef WGPROF
#undef WGCOUNT
}

} // namespace nxzw

#ifndef NXZ_CPU_SIM
extern "C" int nxz_launch_inflate_order_only(const nxz_batch_job_t *jobs, size_t nslots, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io,
					     const uint32_t *order, hipStream_t stream);

// control words of a launch: the job counter, the reasons, then the hand-back list (a count, the indices from word 64 on)
extern "C" size_t nxz_inflate_wg_workspace(size_t n)
{
	return 256 + 256 + 256 + ((n * sizeof(uint32_t) + 255) & ~(size_t)255);
}

// All n streams a workgroup each; the streams the kernel hands back are decoded behind it by the kernel that knows every
// case, a wavefront each (nxzi::inflate_kernel through the list: a slot that holds no job ends at once).  order (may be
// NULL): the jobs by falling source length.  Checksums by nxzl::cksum_kernel.
extern "C" int nxz_launch_inflate_wg(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io,
				     uint8_t *wg_ws, const uint32_t *order, hipStream_t stream)
{
	if (!n) return 0;
	if (n >= (1u << 31)) return (int)hipErrorInvalidValue;
	uint32_t *ctr = (uint32_t *)wg_ws, *dbg = (uint32_t *)(wg_ws + 256), *bail = (uint32_t *)(wg_ws + 512);
	unsigned long long *prof = (unsigned long long *)(wg_ws + 320);
	(void)hipMemsetAsync(wg_ws, 0, 512 + 256, stream);
	(void)hipMemsetAsync(bail + 64, 0xff, n * sizeof(uint32_t), stream);
	static const unsigned cus = [] { int dev = 0, v = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return (unsigned)(v > 0 ? v : 256); }();
	const char *pe = getenv("NXZ_WG_PMIN");                            // (read at every call: the tests switch it)
	const unsigned pmin = pe && atoi(pe) >= 64 ? (unsigned)atoi(pe) : 512u;
	const unsigned grid = (unsigned)(n < cus ? n : cus);
	const char *pr = getenv("NXZ_WG_PROF");
	if (pr && atoi(pr)) hipLaunchKernelGGL(nxzw::inflate_wg_kernel<true>, dim3(grid), dim3(nxzw::NT), 0, stream, jobs, (uint32_t)n, results, order, ctr, bail, pmin, dbg, prof);
	else hipLaunchKernelGGL(nxzw::inflate_wg_kernel<false>, dim3(grid), dim3(nxzw::NT), 0, stream, jobs, (uint32_t)n, results, order, ctr, bail, pmin, dbg, prof);
	int rc = (int)hipGetLastError();
	if (rc) return rc;
	rc = nxz_launch_inflate_order_only(jobs, n, results, dht_io, bail + 64, stream);
	if (rc) return rc;
	return nxz_launch_cksum(jobs, n, results, stream);
}

// (diagnostic / tests: the reasons of the last launch on this workspace, 16 words; [0] unused, [15] = streams handed back; the caller has waited for the stream)
extern "C" int nxz_inflate_wg_reasons(const uint8_t *wg_ws, uint32_t *out16)
{
	int rc = (int)hipMemcpy(out16, wg_ws + 256, 15 * sizeof(uint32_t), hipMemcpyDeviceToHost);
	if (!rc) rc = (int)hipMemcpy(out16 + 15, wg_ws + 512, sizeof(uint32_t), hipMemcpyDeviceToHost);
	return rc;
}
// (NXZ_WG_PROF=1: thread 0's cycles by phase and the counts, 12 words -- nxzw::P_*)
extern "C" int nxz_inflate_wg_prof(const uint8_t *wg_ws, unsigned long long *out12)
{
	return (int)hipMemcpy(out12, wg_ws + 320, nxzw::P_N * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#endif
