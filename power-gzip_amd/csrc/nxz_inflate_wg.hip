// nxz_inflate_wg.hip -- batched DEFLATE decompression, one stream per WORKGROUP, the stream in LDS.
//
// Same engine function as nxz_inflate.hip / nxz_inflate_lanes.hip (GZIP_FC_DECOMPRESS, issued at /root/reference
// lib/nx_inflate.c:909-912; outputs inc_nx/nxu.h:403-541; CPU restatement: oracle/nxz_inflate.c) for the streams a batch is
// made of nearly always: a fresh stream (no resume state, no history) that runs to its final end-of-block, of any length.
// Anything else -- an error of any kind, a stream that ends early, a target that is too small, resume fields -- is HANDED
// BACK (a list of job indices, as inflate_lanes_fixed_kernel does it) to the kernels that know every case; this one never
// reports an error itself.
//
// Why a workgroup per stream: a CU of gfx950 has 160 KiB of LDS, which holds a 64 KiB block's source AND its output AND its
// decode tables.  So the source is read from HBM once, coalesced; the output is written once, coalesced; every table look-up,
// every match copy and every bit of the stream in between is an LDS access -- the stream-per-lane kernels issue one
// 64-cache-line memory instruction per token and wait three quarters of their time for them, the stream-per-wavefront kernel
// spends thirty wave-instructions on a token.  Here 1024 lanes decode ONE Huffman-coded block side by side:
//   1. the block's header is read by one wavefront, the decode tables (10 / 9 root bits + sub-tables, 32-bit entries that
//      carry base and extra-bit count) are built by all;
//   2. the span to decode is cut into pieces, a lane each.  Lane 0 starts at the first token, the others at a guess; every
//      lane decodes (lengths only) to the first token that starts in the next piece.  Huffman-coded data synchronises itself,
//      so most lanes END on a true token boundary although they began on a false one: in the next round every piece whose
//      neighbour ended elsewhere is walked again from there, until no start moves (6-8 rounds; never more than pieces);
//   3. a prefix sum over the pieces' output counts gives every lane its place in the output; a last pass writes the
//      literals and parks every match as a 3-byte record in the first bytes of the room it will fill (a bitmap: where
//      matches start);
//   4. when LDS is full or the stream is over, the matches are resolved by pointer jumping (resolve_matches) and
//   5. the output leaves LDS 16 bytes a lane.  (CRC-32 / Adler-32: nxzl::cksum_kernel behind this one, as for the lane kernels.)
// A stream longer than LDS goes in spans: the source through a 64 KiB window, the output flushed with its last 32 KiB kept
// as the window of distances (flush_out), a piece committed whole or not at all.
//
// Written against a small subset of the device language (barriers, ballots, shuffles, LDS atomics) so that
// tests/native/hip_cpu_shim.h can run a workgroup on the CPU, an OS thread per lane: tests/test_inflate_wg_sim.py.
#ifndef NXZ_CPU_SIM
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "nxz_device.h"
#define NXZ_WG_GLOBAL NXZ_GLOBAL_AS
#else
#include <stdint.h>
#include "../../include/nxz_engine.h"
#define NXZ_WG_GLOBAL
#endif

namespace nxzw {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));     // 16 bytes a lane, in whatever address space

constexpr int NT = 1024, NW = NT / 64;
constexpr uint32_t OUT_MAX = 65536;
constexpr uint32_t SRC_MAX = 65600;                   // bytes of source in LDS (incl. up to 15 bytes in front: the load is 16-byte aligned)
constexpr uint32_t SRC_WORDS = SRC_MAX / 4 + 8;
constexpr int RL = 10, RD = 9;                        // root bits of the literal/length and distance tables
constexpr uint32_t LSUB = 352, DSUB = 256;            // sub-table entries (zlib's ENOUGH for 286 symbols, root 10: 1332 - 1024)
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
// PROF (NXZ_WG_PROF=1, measurements): thread 0's clock at the ends of the phases, summed over the launch's streams in prof[]:
// load, block headers, dynamic headers read, tables, the first pass, the later rounds, prefix sum + the writing pass, the list of
// matches, the matches, out; then counts: rounds, streams, coded blocks, pieces
enum { P_LOAD, P_HEADER, P_DHT, P_TABLES, P_FIRST, P_ROUNDS, P_WRITE, P_LIST, P_MATCH, P_OUT, P_NROUNDS, P_STREAMS, P_BLOCKS, P_PIECES, P_MTRIPS, P_MTRIPMAX, P_MWAITS, P_MATCHES, P_R1, P_R2, P_R3, P_R4, P_R5, P_R6, P_C3, P_C4, P_C5, P_C6, P_N };

struct __attribute__((aligned(16))) Lds {
	uint32_t pad[4];                    // (the match copies read up to four dwords in front of the output without looking)
	uint32_t out[OUT_MAX / 4];
	uint32_t src[SRC_WORDS];            // the stream; when all blocks are decoded: the positions of the matches, 16 bits each
	uint32_t mstart[OUT_MAX / 32];      // bit p: a match starts at output byte p (its record stands there)
	uint32_t lit[(1 << RL) + LSUB];
	uint32_t dist[(1 << RD) + DSUB];
	uint32_t pend[NSUBMAX];             // table build: sub-table bits per root index; the rounds: every piece's end | flag << 24
	uint16_t lcount[16], dcount[16];
	uint16_t crow[6][16];               // the table build: symbols per code length in every row of 64 symbols (five literal/length rows, the distance row)
	uint16_t lsym[288], dsym[32];
	uint8_t lens[320];
	uint32_t wsum[NW];
	uint32_t nout[NT];                  // the rounds: the bytes every piece makes
	uint16_t list[NT];                  // the rounds: the pieces to decode again (piece | start - first bit << 10)
	uint32_t nredo[2];
	// wave-uniform scalars, written by one thread in front of a barrier
	uint32_t jid, bail, pos, outn, bfinal, btype, st_len, hlit, hdist, firstbad;
	// the stream in parts: the window of the source (its first byte's offset, its bits, does it reach the stream's end, is it there),
	// the output (bytes of it in front of LDS position 0, the first LDS position memory does not have yet, the lowest a distance may reach)
	uint32_t wb, wbits, wend, wvalid, aoff, fl0, lowest, inblock, act, cut, stall, done;
	// the header's code lengths by all wavefronts (read_lengths): every segment's first bit and what stands in front of it; the walk's state
	uint32_t segin[NW], hstate, hn, hprev, hein, hend, hpos;
	uint32_t fixed_ok;                  // the tables in lit / dist are the fixed code's (they stand from stream to stream)
	uint32_t span_m, span_len, span_dist;   // the match that reaches from the first half of the output into the second
	uint32_t prof[P_N], tprev[2], tripmax;
};
static_assert(NT == (1 << RL) && NT >= (1 << RD), "a lane per root entry");
static_assert(sizeof(Lds) <= 163840, "the workgroup's LDS image must fit the CU's 160 KiB");
static_assert(32768 * 2 <= SRC_WORDS * 4, "the pointers of half the output must fit the room of the source");

// The workgroup's LDS image: at namespace scope, so that the phases below can be functions of their own (each with its own
// register allocation: as one inlined body the kernel spilled in its loops) and still address LDS directly.
__shared__ Lds L;

#define NXZ_WG_PHASE __device__ __attribute__((noinline))

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
__device__ __forceinline__ uint32_t peek32(uint32_t p)
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

// inclusive prefix sum over the wavefront: on the device by DPP moves (four shifts within the rows of 16, two broadcasts across
// them: six vector instructions), not by six LDS-crossbar shuffles of a hundred cycles each
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, int lane)
{
#ifndef NXZ_CPU_SIM
	(void)lane;
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
	return v;
#else
	for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, (unsigned)d, 64); if (lane >= d) v += o; }
	return v;
#endif
}

// exclusive prefix sum over the workgroup (all threads call it); *total = the sum
__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t *total)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t inc = wave_scan_incl(v, lane);
	__syncthreads();                               // (wsum may still be read from the last scan)
	if (lane == 63) L.wsum[wave] = inc;
	__syncthreads();
	uint32_t base = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < NW; w++) { const uint32_t s = L.wsum[w]; if (w < wave) base += s; tot += s; }
	*total = tot;
	return base + inc - v;
}

// ---- one piece: the tokens that start in [st, lim).  Returns end bit | flag << 24 | the bytes they make << 32.
// WRITE: literals to their place in the output (from obase on), matches parked as records.
// A trip round the loop: 64 bits of the source at p (three dwords), the literal/length look-up; behind a LITERAL the next code lies
// in the same 32 bits, and if that is a literal too the trip takes both (literal-heavy data -- binaries, images -- makes half
// as many trips: with all sixteen wavefronts at work a trip costs 1200 cycles whatever it decodes -- 75 instructions a wavefront). ----
template <bool WRITE>
__device__ __forceinline__ unsigned long long decode_piece(uint32_t st, uint32_t lim, uint32_t T, uint32_t obase)
{
	uint8_t *ob = (uint8_t *)L.out;
	uint32_t p = st, n = 0, fl = F_OK;
	while (p < lim) {
		const uint32_t w = p >> 5, sh = p & 31;
		const uint32_t a = L.src[w], b = L.src[w + 1], c = L.src[w + 2];
		const uint32_t lo = __builtin_amdgcn_alignbit(b, a, sh), hi = __builtin_amdgcn_alignbit(c, b, sh);
		uint32_t e = L.lit[lo & ((1u << RL) - 1)];
		uint32_t nb = e & 31, kind = e_kind(e), x = e_xb(e), q = nb + x;           // (a literal: x = 0)
		// The look-up behind a literal and the distance look-up behind a length are BOTH issued, whatever the code turns out to
		// be: their latencies overlap, a wavefront whose lanes meet both kinds walks both paths anyway, and the choice is made by
		// masks -- one straight run of instructions for a literal (or two) and for a match whose codes lie in the root tables.
		const uint32_t e2 = L.lit[(lo >> nb) & ((1u << RL) - 1)];
		uint32_t db = __builtin_amdgcn_alignbit(hi, lo, q);
		uint32_t d = L.dist[db & ((1u << RD) - 1)];
		const uint32_t ml = 0u - (uint32_t)(kind == K_LIT);                       // all ones: a literal
		const uint32_t okm = (uint32_t)(kind == K_LEN) & (uint32_t)(e_kind(d) == K_DIST);
		if (__builtin_expect(!(ml | okm), 0)) {
			// the rest: a code of a sub-table (either kind), the end of the block, no code at all
			if (kind == K_LINK) {
				e = L.lit[(e >> 16) + ((lo >> RL) & ((1u << x) - 1))];
				nb = e & 31; kind = e_kind(e); x = e_xb(e); q = nb + x;
				if (kind == K_LIT) {
					if (WRITE) ob[obase + n] = (uint8_t)(e >> 16);
					n++; p += nb;
					continue;
				}
				db = __builtin_amdgcn_alignbit(hi, lo, q);
				d = L.dist[db & ((1u << RD) - 1)];
			}
			if (kind != K_LEN) { if (kind == K_EOB) { p += nb; fl = F_EOB; } else fl = F_ERR; break; }
			if (e_kind(d) == K_LINK) d = L.dist[(d >> 16) + ((db >> RD) & ((1u << e_xb(d)) - 1))];
			if (e_kind(d) != K_DIST) { fl = F_ERR; break; }
		}
		const uint32_t m2 = ml & (0u - ((uint32_t)(e_kind(e2) == K_LIT) & (uint32_t)(p + nb < lim)));   // ... and a literal behind it that starts in the piece
		const uint32_t mlen = (e >> 16) + ((lo >> nb) & ((1u << x) - 1));
		const uint32_t dl = d & 31, dx = e_xb(d);
		if (WRITE) {
			const uint32_t at = obase + n;
			if (ml) {
				ob[at] = (uint8_t)(e >> 16);
				if (m2) ob[at + 1] = (uint8_t)(e2 >> 16);
			} else {
				const uint32_t dist = (d >> 16) + ((db >> dl) & ((1u << dx) - 1));     // dl + dx <= 28
				if (dist > at - L.lowest) { fl = F_ERR; break; }
				ob[at] = (uint8_t)(mlen - 3); ob[at + 1] = (uint8_t)(dist - 1); ob[at + 2] = (uint8_t)((dist - 1) >> 8);
				atomicOr(&L.mstart[at >> 5], 1u << (at & 31));
			}
		}
		p += ((nb + (e2 & 31 & m2)) & ml) | ((q + dl + dx) & ~ml);
		n += ((1u + (m2 & 1)) & ml) | (mlen & ~ml);
	}
	if (p > T) fl = F_RUNOUT;            // the last token reaches beyond the source
	return (unsigned long long)(p | fl << 24) | (unsigned long long)n << 32;
}
NXZ_WG_PHASE unsigned long long piece_count(uint32_t st, uint32_t lim, uint32_t T) { return decode_piece<false>(st, lim, T, 0); }
NXZ_WG_PHASE unsigned long long piece_write(uint32_t st, uint32_t lim, uint32_t T, uint32_t obase) { return decode_piece<true>(st, lim, T, obase); }

// ---- the counting walk of ONE piece by a whole wavefront.  What the later rounds cost is the walk of the few pieces at the
// heads of the chains that have not fallen in step, token after token, each a trip of 500 cycles, while a thousand lanes wait.
// Here the 64 lanes decode the tokens that WOULD start at the next 64 bits, one bit each, side by side (one trip), and the walk
// then only follows the lengths from the true position on -- a register read a token, four to five tokens a trip.  Same result
// as piece_count (the tokens that start in [st, lim), their bytes, the flag), in every lane. ----
NXZ_WG_PHASE unsigned long long piece_count_wave(uint32_t st, uint32_t lim, uint32_t T, int lane)
{
	uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)st), n = 0, fl = F_OK;
	lim = (uint32_t)__builtin_amdgcn_readfirstlane((int)lim);
	while (p0 < lim) {
		const uint32_t p = p0 + (uint32_t)lane;
		const uint32_t w = p >> 5, sh = p & 31;
		const uint32_t a = L.src[w], b = L.src[w + 1], c = L.src[w + 2];
		const uint32_t lo = __builtin_amdgcn_alignbit(b, a, sh), hi = __builtin_amdgcn_alignbit(c, b, sh);
		uint32_t e = L.lit[lo & ((1u << RL) - 1)];
		if (e_kind(e) == K_LINK) e = L.lit[(e >> 16) + ((lo >> RL) & ((1u << e_xb(e)) - 1))];
		const uint32_t nb = e & 31, kind = e_kind(e), x = e_xb(e), q = nb + x;
		// what this bit would be the start of: bits to the next token | bytes it makes << 8; the end of the block and "no token"
		// stand as a step of 128 bits that makes nothing (the walk below ends on it by itself) | the code's bits << 17 | 1 or 2 << 24
		uint32_t tok = nb | 1u << 8;
		if (kind == K_LEN) {
			const uint32_t db = __builtin_amdgcn_alignbit(hi, lo, q);
			uint32_t d = L.dist[db & ((1u << RD) - 1)];
			if (e_kind(d) == K_LINK) d = L.dist[(d >> 16) + ((db >> RD) & ((1u << e_xb(d)) - 1))];
			const uint32_t mlen = (e >> 16) + ((lo >> nb) & ((1u << x) - 1));
			tok = e_kind(d) == K_DIST ? (q + (d & 31) + e_xb(d)) | mlen << 8 : 128u | 2u << 24;
		} else if (kind != K_LIT) tok = kind == K_EOB ? 128u | nb << 17 | 1u << 24 : 128u | 2u << 24;
		// the walk: a register read a token, one way out of the loop
		const uint32_t olim = lim - p0 < 64 ? lim - p0 : 64;
		uint32_t o = 0, t;
		do {
			t = (uint32_t)__builtin_amdgcn_readlane((int)tok, (int)o);
			n += (t >> 8) & 0x1ff; o += t & 0xff;
		} while (o < olim);
		if (t >> 24) {
			p0 += o - 128;
			if ((t >> 24) == 1) { p0 += (t >> 17) & 127; fl = F_EOB; } else fl = F_ERR;
			break;
		}
		p0 += o;
	}
	if (p0 > T) fl = F_RUNOUT;
	return (unsigned long long)(p0 | fl << 24) | (unsigned long long)n << 32;
}

// ---- the matches, when all blocks are decoded: every byte of a match is the byte `distance` in front of it, which may itself
// be a byte of a match ... down to a literal.  The first forms of this phase copied match after match, each as soon as a bitmap
// showed its source bytes there: whatever the lanes' order, 130-260 matches deep in a 64 KiB block hang one behind the other,
// and every link of that chain cost a trip of the polling loop, 1500 cycles -- 200 000 to 400 000 cycles a stream with most of
// the CU waiting.  Now the chains are followed by POINTER JUMPING, a byte a pointer: P[x] = where byte x comes from (itself for
// a literal), then P[x] = P[P[x]] for all x side by side until nothing moves -- a chain of depth d is flat after log2 d rounds
// (a byte is 30-250 copies away from its literal; three links a round: 3-4 rounds), whatever hangs on whatever.  Then one gather.  16 bits a pointer:
// the room the source no longer needs holds them for 32 KiB of output, so the output goes in two halves; pointers of the
// second half into the first are ends of their chains (those bytes are final by then). ----
constexpr uint32_t HALF = 32768;
NXZ_WG_PHASE void resolve_matches(uint32_t from, uint32_t outn, int prof)
{
	const int tid = threadIdx.x;
	uint8_t *ob = (uint8_t *)L.out;
	uint16_t *P = (uint16_t *)L.src;
#define WGPROF2(idx) do { if (prof && tid == 0) { const unsigned long long now_ = (unsigned long long)clock64(), then_ = (unsigned long long)L.tprev[0] | (unsigned long long)L.tprev[1] << 32; L.prof[idx] += (uint32_t)(now_ - then_); L.tprev[0] = (uint32_t)now_; L.tprev[1] = (uint32_t)(now_ >> 32); } } while (0)
	for (uint32_t base = from; base < outn; base += HALF) {
		// A lane owns 32 bytes of the half -- one word of the bitmap of match starts, 32 pointers (16 registers), eight dwords
		// of the output -- and makes ITS pointers: the match that reaches into its bytes from in front (the last start within 258
		// bytes, if it is long enough; from the other half: the one match saved when that half was done), then the matches that
		// start in its word.  Every lane the same 32 steps, whatever the matches' lengths.
		const uint32_t i0 = 32 * (uint32_t)tid, x0 = base + i0;
		uint32_t *p2 = (uint32_t *)(P + i0);                                 // pointers 2 j, 2 j + 1 of this lane: p2[j]
		{
			const uint32_t starts = L.mstart[x0 >> 5];
			uint32_t own[9];                                                    // this lane's bytes and four behind them: the records of the matches that start here
#pragma unroll
			for (uint32_t j = 0; j < 9; j++) own[j] = L.out[(x0 >> 2) + j];
			uint32_t mend = 0, mdist = 1, mbeg = 0, mlen = 0, r = 0, even = 0; // the match the walk stands in: its end, distance, start, length; r = (x - mbeg) % mdist
			{
				// the last start in front of x0, within 258 bytes and within this half
				const uint32_t lowest = x0 >= base + 288 ? (x0 - 258) >> 5 : base >> 5;
				bool found = false;
				for (uint32_t w = x0 >> 5; w-- > lowest;) {
					const uint32_t bits = L.mstart[w];
					if (bits) {
						const uint32_t m = w * 32 + 31 - (uint32_t)__builtin_clz(bits), len = (uint32_t)ob[m] + 3;
						if (m + len > x0) { mbeg = m; mlen = len; mend = m + len; mdist = ((uint32_t)ob[m + 1] | (uint32_t)ob[m + 2] << 8) + 1; }
						found = true;
						break;
					}
				}
				if (!found && base && L.span_len && L.span_m + L.span_len > x0 && x0 < base + 288) { mbeg = L.span_m; mlen = L.span_len; mend = mbeg + mlen; mdist = L.span_dist; }
				if (mend > x0 && mdist < mlen) r = (x0 - mbeg) % mdist;
			}
#pragma unroll
			for (uint32_t j = 0; j < 32; j++) {
				const uint32_t x = x0 + j;
				if ((starts >> j) & 1) {
					const uint32_t rec = __builtin_amdgcn_alignbyte(own[(j >> 2) + 1], own[j >> 2], j & 3);
					mbeg = x; mlen = (rec & 0xff) + 3; mend = x + mlen; mdist = ((rec >> 8) & 0xffff) + 1; r = 0;
				}
				// a literal is its own source; a byte of a match comes from `distance` in front -- of a match that overlaps itself
				// (a period) from the period in front of the match, not from its own bytes (else a run of one value is a chain as long
				// as the run)
				uint32_t src = x;
				if (x < mend) src = mdist < mlen ? mbeg - mdist + r : x - mdist;
				r = r + 1 == mdist ? 0 : r + 1;
				if (j & 1) p2[j >> 1] = even | src << 16; else even = src;
			}
			// the match that reaches from this half into the next: its record will be gone by then
			if (base == 0 && outn > HALF && tid == NT - 1) { L.span_m = mbeg; L.span_dist = mdist; L.span_len = mend > HALF ? mlen : 0; }
		}
		__syncthreads();
		WGPROF2(P_LIST);
		// The jumping and the gather go by ANOTHER split of the half than the build's: a lane takes the four bytes 4 g .. 4 g + 3
		// of the groups g = tid, tid + 1024, ... (eight of them) -- the wavefront's lanes stand on 256 consecutive bytes, whose
		// sources are mostly consecutive too (the bytes of a match), so the wavefront's look-ups fall into neighbouring LDS banks
		// instead of 64 random ones, and its own pointers and output dwords are side by side.
		struct alignas(8) U2 { uint32_t x, y; };
		U2 *PG = (U2 *)P;
		uint32_t pp[16];                                                     // pp[2 j], pp[2 j + 1]: the pointers of group tid + NT j
#pragma unroll
		for (uint32_t j = 0; j < 8; j++) { const U2 v = PG[(uint32_t)tid + NT * j]; pp[2 * j] = v.x; pp[2 * j + 1] = v.y; }
		uint32_t rounds = 0, open = 0;
#pragma unroll
		for (uint32_t j = 0; j < 16; j++) {                                  // (open: pointers that are not their own source)
			const uint32_t xg = base + 4 * ((uint32_t)tid + NT * (j >> 1)) + 2 * (j & 1);
			if ((pp[j] & 0xffff) != xg) open |= 1u << (2 * j);
			if ((pp[j] >> 16) != xg + 1) open |= 2u << (2 * j);
		}
		for (;;) {
			bool moved = false;
#pragma unroll
			for (uint32_t j = 0; j < 16; j++) {
				if (!((open >> (2 * j)) & 3)) continue;
				uint32_t a = pp[j] & 0xffff, b = pp[j] >> 16;
				// three links a round (the rounds' barriers and the sixteen blocks of this loop cost more than the look-ups once few
				// pointers are still open)
				uint32_t na = a >= base ? P[a - base] : a, nb = b >= base ? P[b - base] : b;   // (a pointer into the first half: the end of its chain)
				if (na == a) open &= ~(1u << (2 * j)); else {
					uint32_t n2 = na >= base ? P[na - base] : na;
					if (n2 == na) open &= ~(1u << (2 * j)); else { const uint32_t n3 = n2 >= base ? P[n2 - base] : n2; if (n3 == n2) open &= ~(1u << (2 * j)); n2 = n3; }
					na = n2;
				}
				if (nb == b) open &= ~(2u << (2 * j)); else {
					uint32_t n2 = nb >= base ? P[nb - base] : nb;
					if (n2 == nb) open &= ~(2u << (2 * j)); else { const uint32_t n3 = n2 >= base ? P[n2 - base] : n2; if (n3 == n2) open &= ~(2u << (2 * j)); n2 = n3; }
					nb = n2;
				}
				if (na != a || nb != b) { pp[j] = na | nb << 16; ((uint32_t *)P)[2 * ((uint32_t)tid + NT * (j >> 1)) + (j & 1)] = pp[j]; moved = true; }
			}
			rounds++;
			if (!__syncthreads_or(moved)) break;
		}
		WGPROF2(P_MWAITS);
		// every byte from the end of its chain
		{
#pragma unroll
			for (uint32_t j = 0; j < 8; j++) {
				const uint32_t pa = pp[2 * j], pb = pp[2 * j + 1];
				L.out[(base >> 2) + (uint32_t)tid + NT * j] = (uint32_t)ob[pa & 0xffff] | (uint32_t)ob[pa >> 16] << 8 | (uint32_t)ob[pb & 0xffff] << 16 | (uint32_t)ob[pb >> 16] << 24;
			}
		}
		if (prof && tid == 0) L.prof[P_MTRIPS] += rounds;
		__syncthreads();
	}
#undef WGPROF2
}

// ---- the header of a dynamic block, by wavefront 0 (the algorithm of nxzi::read_dht; the stream's bits come from LDS):
// code lengths into L.lens, L.hlit / L.hdist, L.pos behind the header.  false: not a header this kernel takes on. ----
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
NXZ_WG_PHASE bool read_header(uint32_t start, uint32_t T, int lane, bool codes_only)
{
	start = uni(start); T = uni(T);                                   // (read from LDS: the same in every lane, and now the compiler knows)
	if (start + 14 > T) return false;
	auto word_at = [&](uint32_t idx) -> uint32_t { return idx < SRC_WORDS ? L.src[idx] : 0; };
	const uint32_t v = uni(peek32(start));
	const int hlit = (int)(v & 31) + 257, hdist = (int)((v >> 5) & 31) + 1, hclen = (int)((v >> 10) & 15) + 4;
	uint32_t pos = start + 14;
	if (hlit > 286 || hdist > 30) return false;
	if (pos + 3 * (uint32_t)hclen > T) return false;
	// the code-length code: lane i < hclen reads the i-th 3-bit length, which belongs to symbol order[i]
	uint32_t myl = 0;
	{
		const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
		uint32_t mine = 0, mysym = 0;
		if (lane < hclen) { mine = peek32(pos + 3 * (uint32_t)lane) & 7; mysym = order[lane]; }
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
	// ---- the code lengths themselves: up to 316 symbols of 2 to 14 bits, one after the other -- as a loop of the whole wavefront
	// over them (nxzi::read_dht) 650 cycles a symbol, 100 000 a header, with fifteen wavefronts waiting.  So the lanes walk the
	// sequence side by side, 32 bits of it each, as the pieces of a block are walked below: lane 0 begins at the first symbol,
	// the others where their chunk begins; in the next round every lane begins where its neighbour ended, until no start
	// moves (a code of at most 7 bits falls in step within a chunk or two).  Then the counts' prefix sum says where every
	// lane's lengths go, and a last walk writes them (runs of zeros are nothing to write: the array is cleared first). ----
	const int total = hlit + hdist;
	uint8_t *cl7 = (uint8_t *)L.pend;                                   // the 7-bit look-up as bytes: symbol | length << 5, 0xff: no such code
	cl7[lane] = (uint8_t)tlo; cl7[lane + 64] = (uint8_t)thi;
	for (int i = lane; i < 320; i += 64) L.lens[i] = 0;
	__threadfence_block();
	(void)__ballot(1);                                                  // (the wavefront's lanes are in step here -- the CPU shim's are not by themselves)
	if (codes_only) {                                                   // (the lengths themselves: read_lengths, all wavefronts)
		if (lane == 0) { L.hlit = (uint32_t)hlit; L.hdist = (uint32_t)hdist; L.hpos = pos; }
		return true;
	}
	int n0 = 0;
	uint32_t inh = 16;                                                  // the length in front of this window (16: none yet)
	for (;;) {
		const uint32_t cs = pos + 32 * (uint32_t)lane, ce = cs + 32;
		uint64_t W;
		{
			const uint32_t w = cs >> 5, sh = cs & 31;
			const uint32_t a = word_at(w), b = word_at(w + 1), c = word_at(w + 2);
			W = (uint64_t)__builtin_amdgcn_alignbit(b, a, sh) | (uint64_t)__builtin_amdgcn_alignbit(c, b, sh) << 32;
		}
		// one walk: the symbols that start in [s, ce); WRITE: their lengths to L.lens from n on, at most `room` of them
		uint32_t st = cs, en = cs, cnt = 0, pv = 16;
		bool bad = false;
		auto walk = [&]() __attribute__((always_inline)) {
			uint32_t s = st, c = 0, v = 16;
			bool bd = false;
			while (s < ce) {
				const uint32_t o = s - cs;
				const uint32_t e = cl7[(uint32_t)(W >> o) & 127];
				if (e == 0xff) { bd = true; break; }
				const uint32_t len = e >> 5, sym = e & 31;
				const uint32_t eb = sym < 16 ? 0 : sym == 16 ? 2 : sym == 17 ? 3 : 7;
				const uint32_t rep = sym < 16 ? 1 : ((uint32_t)(W >> (o + len)) & ((1u << eb) - 1)) + (sym == 18 ? 11 : 3);
				if (sym < 16) v = sym; else if (sym != 16) v = 0;
				c += rep; s += len + eb;
			}
			en = s; cnt = c; pv = v; bad = bd;
		};
		walk();
		for (;;) {
			const uint32_t pen = __shfl_up(en, 1, 64), pbad = __shfl_up((uint32_t)bad, 1, 64);
			const bool redo = lane > 0 && !pbad && pen != st;
			if (!__ballot(redo)) break;
			if (redo) { st = pen; walk(); }
		}
		const uint64_t badm = __ballot(bad);
		const uint32_t fb = badm ? (uint32_t)__builtin_ctzll(badm) : 64;
		uint32_t pc = cnt;                                                // inclusive prefix sum of the counts
		uint32_t lv = pv;                                                 // ... and the last length that is known, up to and including this lane
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t oc = __shfl_up(pc, (unsigned)d, 64), ov = __shfl_up(lv, (unsigned)d, 64);
			if (lane >= d) { pc += oc; if (lv == 16) lv = ov; }
		}
		const uint64_t reachm = __ballot(n0 + (int)pc >= total);
		const uint32_t lstar = reachm ? (uint32_t)__builtin_ctzll(reachm) : 64;      // the lane in which the last length lies
		if (fb < 64 && fb <= lstar) return false;                           // a code that is none, in front of the end
		const uint32_t lastl = lstar < 64 ? lstar : 63;
		if (__ballot((uint32_t)lane <= lastl && en > T)) return false;      // the source ends inside the header
		uint32_t inherited = __shfl_up(lv, 1, 64);
		if (lane == 0 || inherited == 16) inherited = lane == 0 ? inh : (inherited == 16 ? inh : inherited);
		// the writing walk
		uint32_t endpos = 0;
		bool wrong = false;
		if ((uint32_t)lane <= lastl) {
			uint32_t s = st, v = inherited;
			int n = n0 + (int)(pc - cnt);
			while (s < ce && n < total) {
				const uint32_t o = s - cs;
				const uint32_t e = cl7[(uint32_t)(W >> o) & 127];
				const uint32_t len = e >> 5, sym = e & 31;
				const uint32_t eb = sym < 16 ? 0 : sym == 16 ? 2 : sym == 17 ? 3 : 7;
				const int rep = sym < 16 ? 1 : (int)(((uint32_t)(W >> (o + len)) & ((1u << eb) - 1)) + (sym == 18 ? 11 : 3));
				if (n + rep > total) { wrong = true; break; }
				if (sym < 16) { L.lens[n] = (uint8_t)sym; v = sym; }
				else if (sym == 16) {
					if (v == 16) { wrong = true; break; }                     // "repeat the last length" with no length in front of it
					for (int q = 0; q < rep; q++) L.lens[n + q] = (uint8_t)v;
				} else v = 0;
				n += rep; s += len + eb;
			}
			endpos = s;
		}
		if (__ballot(wrong)) return false;
		if (lstar < 64) { pos = (uint32_t)__builtin_amdgcn_readlane((int)endpos, (int)lstar); break; }
		// the sequence goes on behind this window
		n0 += (int)(uint32_t)__builtin_amdgcn_readlane((int)pc, 63);
		pos = (uint32_t)__builtin_amdgcn_readlane((int)en, 63);
		const uint32_t l63 = (uint32_t)__builtin_amdgcn_readlane((int)lv, 63);
		if (l63 != 16) inh = l63;
		if (pos > T) return false;
	}
	__threadfence_block();
	(void)__ballot(1);
	if (lane == 0) { L.hlit = (uint32_t)hlit; L.hdist = (uint32_t)hdist; L.pos = pos; }
	return true;
}

// ---- the code lengths of a dynamic block's header by ALL wavefronts (read_header with codes_only has left the look-up of the
// code-length code, the counts and the position): up to 316 symbols of 2 to 14 bits, one after the other -- one wavefront
// alone took 22 000 cycles for them with fifteen waiting.  A window of 1024 bits, a lane a bit:
//   1. every lane looks up the symbol that WOULD start at its bit;
//   2. a wavefront's 64 bits are a segment; a symbol of the segment in front reaches at most 13 bits into it, so fourteen
//      lanes walk the segment from its fourteen possible first bits: where the walk leaves the segment, how many lengths it
//      makes, the last length it defines;
//   3. one lane strings the segments together (sixteen look-ups): every segment's true first bit, the count and the last
//      length in front of it; the segment in which the count reaches the header's total is the last;
//   4. every wavefront follows its segment from the true first bit (scalar: a register read a symbol), a prefix sum over the
//      lanes on the way gives every symbol its place, and those lanes write.
// L.bail = R_DHT: not a header this kernel takes on.  L.pos: behind the header. ----
NXZ_WG_PHASE void read_lengths(uint32_t T)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint8_t *cl7 = (const uint8_t *)L.pend;
	const uint32_t total = L.hlit + L.hdist;
	uint32_t *tk = L.nout, *seg = (uint32_t *)L.list;
	if (tid == 0) { L.hn = 0; L.hprev = 16; L.hein = 0; L.hend = 0xffffffffu; }
	__syncthreads();
	for (;;) {
		const uint32_t pos = L.hpos;
		// 1. bits to the next symbol | lengths it stands for << 8 | the symbol << 16; no such code: a step out of every segment, flagged
		const uint32_t v = peek32(pos + (uint32_t)tid);
		const uint32_t e = cl7[v & 127];
		const uint32_t len = e >> 5, sym = e & 31;
		const uint32_t eb = sym < 16 ? 0 : sym == 16 ? 2 : sym == 17 ? 3 : 7;
		const uint32_t rep = sym < 16 ? 1 : ((v >> len) & ((1u << eb) - 1)) + (sym == 18 ? 11 : 3);
		const bool valid = e != 0xff;
		const uint32_t tok = valid ? (len + eb) | rep << 8 | sym << 16 : 128u | 1u << 24;
		tk[tid] = tok;
		__syncthreads();
		// 2. exit (4 bits) | met no-such-code << 4 | last length defined (16: none) << 8 | lengths << 16
		if (lane < 14) {
			uint32_t o = (uint32_t)lane, cnt = 0, lastdef = 16, stop = 0;
			while (o < 64) {
				const uint32_t t = tk[64 * wave + o];
				if (t >> 24) { stop = 1; break; }
				const uint32_t sy = (t >> 16) & 31;
				cnt += (t >> 8) & 0xff;
				if (sy != 16) lastdef = sy < 16 ? sy : 0;
				o += t & 0xff;
			}
			seg[16 * wave + lane] = ((o - 64) & 15) | stop << 4 | lastdef << 8 | cnt << 16;
		}
		__syncthreads();
		// 3. first bit | lengths in front << 4 | last length in front << 20 | 1 << 31 for every segment the header reaches into
		if (wave == 0) {
			// (wavefront 0, the segments' entries in its registers -- lane e holds what the walk from bit e found in every segment -- so
			// that the chain from segment to segment reads registers, not LDS: sixteen dependent LDS look-ups by one lane were 6000 cycles)
			uint32_t col[NW];
#pragma unroll
			for (uint32_t w = 0; w < NW; w++) col[w] = lane < 16 ? seg[16 * w + (uint32_t)lane] : 0;
			uint32_t en = uni(L.hein), n = uni(L.hn), prev = uni(L.hprev), state = 0, mine = 0;
#pragma unroll
			for (uint32_t w = 0; w < NW; w++) {
				if (state == 0) {
					const uint32_t ent = (uint32_t)__builtin_amdgcn_readlane((int)col[w], (int)en);
					if ((uint32_t)lane == w) mine = en | n << 4 | prev << 20 | 1u << 31;
					n += ent >> 16;
					if (n >= total) state = 1;                                // (what the segment holds behind the header's end: step 4 looks)
					else if (ent & 16) state = 2;                              // no such code in front of the header's end
					else { if (((ent >> 8) & 31) != 16) prev = (ent >> 8) & 31; en = ent & 15; }
				}
			}
			if (lane < NW) L.segin[lane] = mine;
			if (lane == 0) { L.hstate = state; L.hn = n; L.hprev = prev; L.hein = en; L.hpos = pos + 64 * NW; }
		}
		__syncthreads();
		if (L.hstate == 2) { if (tid == 0) L.bail = R_DHT; __syncthreads(); return; }
		// 4.
		const uint32_t sin = L.segin[wave];
		if (sin >> 31) {
			const uint32_t en = uni(sin & 15), n_in = uni((sin >> 4) & 0xffff), prev_in = uni((sin >> 20) & 31);
			unsigned long long path = 0;
			uint32_t o = en;
			do {
				const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tok, (int)o);
				path |= 1ull << o;
				o += t & 0xff;
			} while (o < 64);
			const bool on = (path >> lane) & 1;
			const uint32_t inc = wave_scan_incl(on && valid ? rep : 0, lane);   // inclusive prefix sum of the lengths on the way
			const uint32_t at = n_in + inc - (on && valid ? rep : 0);
			// the last length defined in front of this lane's symbol
			const unsigned long long defm = __ballot(on && valid && sym != 16) & ((1ull << lane) - 1);
			const uint32_t defval = sym < 16 ? sym : 0;
			const uint32_t from = defm ? 63u - (uint32_t)__builtin_clzll(defm) : 0u;
			const uint32_t got = __shfl(defval, (int)from, 64);
			const uint32_t pv = defm ? got : prev_in;
			if (on && at < total) {
				bool wrong = !valid || at + rep > total || (sym == 16 && pv == 16);   // no such code; more lengths than announced; "repeat" with nothing in front
				if (!wrong) {
					if (sym < 16) L.lens[at] = (uint8_t)sym;
					else if (sym == 16) for (uint32_t q = 0; q < rep; q++) L.lens[at + q] = (uint8_t)pv;
					if (at + rep == total) L.hend = pos + (uint32_t)tid + (tok & 0xff);
				} else L.bail = R_DHT;
			}
		}
		__syncthreads();
		if (L.bail) return;
		if (L.hstate == 1) break;
	}
	if (tid == 0) { if (L.hend > T) L.bail = R_DHT; else L.pos = L.hend; }
	__syncthreads();
}

// ---- the decode tables of a block from L.lens (all lanes; R_TABLES in L.bail: the sub-tables do not fit) ----
NXZ_WG_PHASE void build_tables(int hlit, int hdist)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	for (uint32_t i = tid; i < NSUBMAX; i += NT) L.pend[i] = 0;
	for (uint32_t i = tid; i < LSUB; i += NT) L.lit[(1u << RL) + i] = 0;
	for (uint32_t i = tid; i < DSUB; i += NT) L.dist[(1u << RD) + i] = 0;
	// The canonical codes (RFC 1951 3.2.2), a wavefront a row of 64 symbols: wavefronts 0-4 the literal/length symbols, wavefront 5
	// the distance symbols (one wavefront for all five rows took 7000 cycles with the others waiting).  Per length: the row's
	// count by a ballot and the lane's rank among the row's symbols of its length; then, the rows' counts side by side in LDS,
	// every wavefront works out for ITS symbols where the length's codes and places begin and how many symbols of the rows in front
	// share them.
	uint32_t myl = 0, mycode = 0, rk = 0;                       // this lane's symbol: its code length, its code; its rank in its row
	const uint32_t mysym = wave < 5 ? (uint32_t)(wave * 64 + lane) : (uint32_t)lane;
	if (wave < 6) {
		const int n = wave < 5 ? hlit : hdist;
		myl = (int)mysym < n ? L.lens[(wave < 5 ? 0 : hlit) + (int)mysym] : 0;
		const uint64_t below = (1ull << lane) - 1;
		for (uint32_t bl = 1; bl <= 15; bl++) {
			const uint64_t m = __ballot(myl == bl);
			if (lane == 0) L.crow[wave][bl] = (uint16_t)__popcll(m);
			if (myl == bl) rk = (uint32_t)__popcll(m & below);
		}
		if (lane == 0) L.crow[wave][0] = 0;
	}
	__syncthreads();
	if (wave < 6) {
		// lane b < 16: the symbols of length b in all rows of this alphabet, and in the rows in front of this one
		const int r0 = wave < 5 ? 0 : 5, r1 = wave < 5 ? 5 : 6;
		uint32_t tb = 0, bel = 0;
		for (int r = r0; r < r1; r++) { const uint32_t v = L.crow[r][lane & 15]; tb += v; if (r < wave) bel += v; }
		if (lane >= 16) tb = 0;
		const uint32_t offs = wave_scan_incl(tb, lane) - tb;       // where length b's symbols begin in the sorted list
		uint32_t cb = 0;                                           // the first code of length b: c_b = (c_(b-1) + n_(b-1)) << 1
		for (int k = 1; k <= 14; k++) {
			const uint32_t tk = (uint32_t)__builtin_amdgcn_readlane((int)tb, k);
			if (lane > k && lane < 16) cb += tk << (lane - k);
		}
		const uint32_t c_l = __shfl(cb, (int)myl, 64), o_l = __shfl(offs, (int)myl, 64), b_l = __shfl(bel, (int)myl, 64);
		if (myl) {
			mycode = c_l + b_l + rk;
			(wave < 5 ? L.lsym : L.dsym)[o_l + b_l + rk] = (uint16_t)mysym;
		}
		if ((wave == 0 || wave == 5) && lane < 16) (wave == 0 ? L.lcount : L.dcount)[lane] = (uint16_t)tb;
	}
	__syncthreads();
	{
		uint32_t sym, len;
		L.lit[tid] = root_walk<RL>((uint32_t)tid, L.lcount, L.lsym, sym, len) ? lit_entry(sym, len) : 0;
		if (tid < (1 << RD)) L.dist[tid] = root_walk<RD>((uint32_t)tid, L.dcount, L.dsym, sym, len) ? dist_entry(sym, len) : 0;
	}
	// codes longer than the root: how many index bits the sub-table behind their root index needs
	if (wave < 5 && myl > (uint32_t)RL) {
		const uint32_t lsb = __builtin_bitreverse32(mycode) >> (32 - myl);
		atomicMax(&L.pend[lsb & ((1u << RL) - 1)], myl - RL);
	} else if (wave == 5 && myl > (uint32_t)RD) {
		const uint32_t lsb = __builtin_bitreverse32(mycode) >> (32 - myl);
		atomicMax(&L.pend[(1u << RL) + (lsb & ((1u << RD) - 1))], myl - RD);
	}
	__syncthreads();
	{
		const uint32_t sl = L.pend[tid], sd = tid < (1 << RD) ? L.pend[(1u << RL) + tid] : 0;
		uint32_t tot2;                                              // (both sums in one scan: 16 bits each are room enough)
		const uint32_t o2 = block_scan((sl ? 1u << sl : 0) | (sd ? 1u << sd : 0) << 16, &tot2);
		const uint32_t ol = o2 & 0xffff, od = o2 >> 16, totl = tot2 & 0xffff, totd = tot2 >> 16;
		if (sl) L.lit[tid] = mk(0, K_LINK, sl, (1u << RL) + ol);
		if (sd) L.dist[tid] = mk(0, K_LINK, sd, (1u << RD) + od);
		if (tid == 0 && (totl > LSUB || totd > DSUB)) L.bail = R_TABLES;
	}
	__syncthreads();
	if (L.bail) return;
	if (wave < 5 && myl > (uint32_t)RL) {
		const uint32_t lsb = __builtin_bitreverse32(mycode) >> (32 - myl);
		const uint32_t link = L.lit[lsb & ((1u << RL) - 1)], rem = myl - RL;
		const uint32_t ent = lit_entry(mysym, myl);
		for (uint32_t k = lsb >> RL; k < (1u << e_xb(link)); k += 1u << rem) L.lit[(link >> 16) + k] = ent;
	} else if (wave == 5 && myl > (uint32_t)RD) {
		const uint32_t lsb = __builtin_bitreverse32(mycode) >> (32 - myl);
		const uint32_t link = L.dist[lsb & ((1u << RD) - 1)], rem = myl - RD;
		const uint32_t ent = dist_entry(mysym, myl);
		for (uint32_t k = lsb >> RD; k < (1u << e_xb(link)); k += 1u << rem) L.dist[(link >> 16) + k] = ent;
	}
	__syncthreads();
}

// what a span leaves for the stream's loop to do next: the block is over; the block goes on (behind the span, or behind the
// window); the output in LDS must go to memory first
enum { A_NEXT = 0, A_MORE = 1, A_FLUSH = 2 };

// ---- a span of a Huffman-coded block whose tables stand: at most spanbits of the window from L.pos on (T: the window's bits) --
// the pieces in rounds, the prefix sum, the writing pass for as many pieces as the room in LDS takes (a piece is all there or not
// at all: a match never lies across a flush).  Leaves L.outn and L.pos behind the last piece written and L.act, or a reason in
// L.bail.  capleft: what the job's target still takes. ----
NXZ_WG_PHASE void decode_span(uint32_t T, uint32_t spanbits, uint32_t capleft, uint32_t pmin_bits, uint32_t max_rounds, int prof)
{
	const int tid = threadIdx.x;
#define WGPROF(idx) do { if (prof && tid == 0) { const unsigned long long now_ = (unsigned long long)clock64(), then_ = (unsigned long long)L.tprev[0] | (unsigned long long)L.tprev[1] << 32; L.prof[idx] += (uint32_t)(now_ - then_); L.tprev[0] = (uint32_t)now_; L.tprev[1] = (uint32_t)(now_ >> 32); } } while (0)
	const uint32_t coop_max = max_rounds >> 16;                             // (so many pieces to decode again or fewer: a wavefront each)
	max_rounds &= 0xffff;
	const uint32_t cur = L.pos, R = T - cur < spanbits ? T - cur : spanbits, E = cur + R;
	const uint32_t outn = L.outn, room = OUT_MAX - outn;
	const bool wend = L.wend != 0;
	uint32_t np0 = (R + pmin_bits - 1) / pmin_bits;                      // (pieces of pmin_bits at most unless the lanes run out: a piece of 2-bit tokens, 258 bytes each, must fit the 32 KiB a flush frees)
	np0 = np0 < 1 ? 1 : np0 > (uint32_t)NT ? (uint32_t)NT : np0;
	const uint32_t pdw = (((R + np0 - 1) / np0 + 31) >> 5) | 1;          // dwords a piece, odd: neighbours begin in different LDS banks
	const uint32_t P = pdw * 32;
	const uint32_t NP = (R + P - 1) / P ? (R + P - 1) / P : 1;
	const bool active = (uint32_t)tid < NP;
	const uint32_t g = cur + (uint32_t)tid * P;
	const uint32_t lim = (uint32_t)tid + 1 == NP ? E : g + P;
	uint32_t st = g, no = 0, pe = F_ERR << 24;
	if (active) { const unsigned long long r = piece_count(st, lim, T); pe = (uint32_t)r; L.pend[tid] = pe; L.nout[tid] = (uint32_t)(r >> 32); }
	if (tid == 0) { L.firstbad = NP; L.cut = NT; L.nredo[0] = 0; L.nredo[1] = 0; }
	WGPROF(P_FIRST);
	if (prof && tid == 0) L.prof[P_PIECES] += NP;
	// The rounds: a piece whose neighbour in front ended elsewhere than the piece began is decoded again from there.  Who must go
	// again is scattered over the wavefronts and fewer every round: the pieces to do go on a list and the first lanes take one
	// each; sixteen or fewer: a wavefront each (piece_count_wave).  What the later rounds cost is the longest chain of pieces that
	// have not fallen in step, token after token -- a round of a few pieces takes as long as a round of all of them (the slowest
	// lane's twenty trips), a wavefront a piece 40 % of that.
	uint32_t rounds = 0;
	for (;;) {
		__syncthreads();
		const uint32_t par = rounds & 1;
		bool redo = false;
		if (active && tid > 0) {
			const uint32_t prev = L.pend[tid - 1];
			if ((prev >> 24) == F_OK && (prev & 0xffffff) != st) { st = prev & 0xffffff; redo = true; }
		}
		{
			const unsigned long long m = __ballot(redo);
			if (m) {
				uint32_t base = 0;
				if ((tid & 63) == 0) base = atomicAdd(&L.nredo[par], (uint32_t)__popcll(m));
				base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
				// (a piece's new start lies within a token of its first bit: 6 bits beside the piece's number)
				if (redo) L.list[base + (uint32_t)__popcll(m & ((1ull << (tid & 63)) - 1))] = (uint16_t)((uint32_t)tid | (st - g) << 10);
			}
		}
		__syncthreads();
		const uint32_t cnt = L.nredo[par];
		if (!cnt) break;
		// (a round settles one more piece from the front at least, so there are never more rounds than pieces; the limit is a knob
		// of the tests.  Packed data under nearly flat codes needs hundreds of rounds -- still cheaper than handing the stream back:
		// own exact-table streams of the corpus, 4096 of them, 34 -> 94 GiB/s when the limit went from 256 to "none")
		if (++rounds > max_rounds) { if (tid == 0) L.bail = R_ROUNDS; break; }
		if (tid == 0) { L.nredo[par ^ 1] = 0; if (prof) { L.prof[P_MATCHES] += cnt; if (rounds <= 2) L.prof[P_MTRIPMAX] += cnt; if (rounds >= 3 && rounds <= 6) L.prof[P_C3 + rounds - 3] += cnt;
			const unsigned long long now_ = (unsigned long long)clock64(), then_ = (unsigned long long)L.tprev[0] | (unsigned long long)L.tprev[1] << 32; if (rounds >= 2 && rounds <= 7) L.prof[P_R1 + rounds - 2] += (uint32_t)(now_ - then_); L.tprev[0] = (uint32_t)now_; L.tprev[1] = (uint32_t)(now_ >> 32); } }
		if (cnt <= coop_max) {
			// few pieces left -- the heads of the chains: a wavefront each
			for (uint32_t k = (uint32_t)tid >> 6; k < cnt; k += NW) {
				const uint32_t ent = L.list[k], j = ent & 1023;
				const uint32_t gj = cur + j * P, limj = j + 1 == NP ? E : gj + P;
				const unsigned long long r = piece_count_wave(gj + (ent >> 10), limj, T, tid & 63);
				if ((tid & 63) == 0) { L.pend[j] = (uint32_t)r; L.nout[j] = (uint32_t)(r >> 32); }
			}
		} else if ((uint32_t)tid < cnt) {
			const uint32_t ent = L.list[tid], j = ent & 1023;
			const uint32_t gj = cur + j * P, limj = j + 1 == NP ? E : gj + P;
			const unsigned long long r = piece_count(gj + (ent >> 10), limj, T);
			L.pend[j] = (uint32_t)r; L.nout[j] = (uint32_t)(r >> 32);
		}
	}
	__syncthreads();
	if (active) { pe = L.pend[tid]; no = L.nout[tid]; }
	__syncthreads();
	WGPROF(P_ROUNDS);
	if (prof && tid == 0) L.prof[P_NROUNDS] += rounds;
	if (L.bail) return;
	if (active && (pe >> 24) != F_OK) atomicMin(&L.firstbad, (uint32_t)tid);
	__syncthreads();
	const uint32_t B = L.firstbad;
	const uint32_t fb = B < NP ? L.pend[B] >> 24 : (uint32_t)F_OK;
	// the pieces in front of B are whole; B's is if it met the end of the block.  A token that reaches behind the window is
	// read again when the window has moved -- unless the window ends where the stream ends.
	uint32_t why = 0;
	if (fb == F_ERR || (fb == F_RUNOUT && wend)) why = R_TOKEN;
	else if (B >= NP && E == T && wend) why = R_NOEOB;
	const uint32_t nc = fb == F_EOB ? B + 1 : B < NP ? B : NP;
	uint32_t tot;
	const uint32_t obase = block_scan((uint32_t)tid < nc ? no : 0, &tot);
	if (!why && tot > capleft) why = R_SPACE;
	if (why) { if (tid == 0) L.bail = why; __syncthreads(); return; }
	uint32_t C = nc;
	if (tot > room) {
		if ((uint32_t)tid < nc && obase + no > room) atomicMin(&L.cut, (uint32_t)tid);
		__syncthreads();
		C = L.cut;
	}
	if ((uint32_t)tid < C) {
		const unsigned long long r = piece_write(st, lim, T, outn + obase);
		if ((uint32_t)r != pe || (uint32_t)(r >> 32) != no) L.bail = R_DIST;                 // (a distance beyond the output so far)
		if ((uint32_t)tid + 1 == C) { L.outn = outn + obase + no; L.pos = pe & 0xffffff; }
	}
	if (tid == 0) {
		const uint32_t act = C < nc ? (uint32_t)A_FLUSH : fb == F_EOB ? (uint32_t)A_NEXT : (uint32_t)A_MORE;
		L.act = act;
		if (act == A_NEXT) { L.inblock = 0; if (L.bfinal) L.done = 1; }
		// a span that got nowhere: once is the window's end (it moves now), twice is a stream this kernel does not take
		if (C == 0 && act != A_FLUSH) { if (++L.stall >= 2) L.bail = R_TOKEN; } else L.stall = 0;
	}
	WGPROF(P_WRITE);
	__syncthreads();
}

// ---- the window: 64 KiB of the source from the 16-byte granule of byte `from` on, zeros behind the stream's end ----
NXZ_WG_PHASE void load_window(const NXZ_WG_GLOBAL uint8_t *gsrc, uint32_t nbytes, uint32_t from)
{
	const int tid = threadIdx.x;
	const uint32_t wbn = from & ~15u;
	const uint32_t avail = nbytes - wbn, wl = avail < SRC_MAX ? avail : SRC_MAX;
	const NXZ_WG_GLOBAL v4u *gq = (const NXZ_WG_GLOBAL v4u *)(gsrc + wbn);
	const uint32_t full = wl >> 4, chunks = (wl + 15) >> 4;
	v4u *ls = (v4u *)L.src;
	for (uint32_t i = tid; i < chunks + 2 && i < SRC_WORDS / 4; i += NT) {
		v4u v = { 0, 0, 0, 0 };
		if (i < chunks) {
			v = gq[i];
			if (i >= full) {                                       // the stream's last bytes: what follows them in the granule counts as zero
				const uint32_t keep = wl & 15;
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
	__syncthreads();
	if (tid == 0) { L.pos = L.wb * 8 + L.pos - wbn * 8; L.wb = wbn; L.wbits = wl * 8; L.wend = wbn + wl == nbytes; L.wvalid = 1; }
	__syncthreads();
}

// ---- what LDS holds of the output and memory does not yet (from L.fl0 on): the matches followed to their bytes, out 16 bytes
// a lane.  Not the stream's end: the last 32 KiB stay, as the first half of the output (the window of distances), the rest of
// the stream follows them in the second half; the source's window is gone (the pointers stood in its room). ----
NXZ_WG_PHASE void flush_out(NXZ_WG_GLOBAL uint8_t *dst, bool final, int prof)
{
	const int tid = threadIdx.x;
	uint8_t *ob = (uint8_t *)L.out;
	const uint32_t fl0 = L.fl0, outn = L.outn;
	if (prof && tid == 0) L.tripmax = 0;
	resolve_matches(fl0, outn, prof);
	__syncthreads();
	WGPROF(P_MATCH);
	{
		const uint32_t g0 = L.aoff + fl0, m = outn - fl0;
		uint32_t h = (16 - (g0 & 15)) & 15;
		h = h < m ? h : m;
		if ((uint32_t)tid < h) dst[g0 + tid] = ob[fl0 + tid];
		const uint32_t mid = (m - h) >> 4, l0 = fl0 + h;
		NXZ_WG_GLOBAL v4u *gd = (NXZ_WG_GLOBAL v4u *)(dst + g0 + h);
		if ((l0 & 15) == 0) {
			const v4u *lo = (const v4u *)(ob + l0);
			for (uint32_t i = tid; i < mid; i += NT) gd[i] = lo[i];
		} else {
			const uint32_t sh = l0 & 3;
			for (uint32_t i = tid; i < mid; i += NT) {
				const uint32_t w = (l0 >> 2) + 4 * i;
				const uint32_t a = L.out[w], b = L.out[w + 1], c = L.out[w + 2], d = L.out[w + 3], e = L.out[w + 4 < OUT_MAX / 4 ? w + 4 : w + 3];
				v4u v;
				v.x = __builtin_amdgcn_alignbyte(b, a, sh); v.y = __builtin_amdgcn_alignbyte(c, b, sh);
				v.z = __builtin_amdgcn_alignbyte(d, c, sh); v.w = __builtin_amdgcn_alignbyte(e, d, sh);
				gd[i] = v;
			}
		}
		const uint32_t tail = (m - h) & 15;
		if ((uint32_t)tid < tail) dst[g0 + h + 16 * mid + tid] = ob[l0 + 16 * mid + tid];
	}
	WGPROF(P_OUT);
	if (final) return;
	__syncthreads();
	{
		const uint32_t shift = outn - HALF, sh = shift & 3, w0 = (32 * (uint32_t)tid + shift) >> 2;
		uint32_t v[9];
#pragma unroll
		for (uint32_t j = 0; j < 9; j++) v[j] = L.out[w0 + j < OUT_MAX / 4 ? w0 + j : OUT_MAX / 4 - 1];
		__syncthreads();
#pragma unroll
		for (uint32_t j = 0; j < 8; j++) L.out[8 * (uint32_t)tid + j] = __builtin_amdgcn_alignbyte(v[j + 1], v[j], sh);
		v4u *b0 = (v4u *)L.mstart;
		const v4u z = { 0, 0, 0, 0 };
		for (uint32_t i = tid; i < OUT_MAX / 32 / 4; i += NT) b0[i] = z;
		if (tid == 0) { L.aoff += shift; L.outn = HALF; L.fl0 = HALF; L.span_len = 0; L.wvalid = 0; L.act = A_MORE; }
	}
	__syncthreads();
}
#undef WGPROF

template <bool PROF>
__global__ __launch_bounds__(NT) void inflate_wg_kernel(const nxz_batch_job_t *__restrict__ jobs, uint32_t n, nxz_batch_result_t *__restrict__ results,
							 const uint32_t *__restrict__ order, uint32_t *__restrict__ ctr, uint32_t *__restrict__ bail,
							 uint32_t pmin_bits, uint32_t coop, uint32_t *__restrict__ dbg, unsigned long long *__restrict__ prof)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint8_t *ob = (uint8_t *)L.out;
	const uint8_t *sb = (const uint8_t *)L.src;
#define WGPROF(idx) do { if (PROF && tid == 0) { const unsigned long long now_ = (unsigned long long)clock64(), then_ = (unsigned long long)L.tprev[0] | (unsigned long long)L.tprev[1] << 32; L.prof[idx] += (uint32_t)(now_ - then_); L.tprev[0] = (uint32_t)now_; L.tprev[1] = (uint32_t)(now_ >> 32); } } while (0)
	if (tid == 0) L.fixed_ok = 0;
	if (PROF && tid == 0) { for (int i = 0; i < P_N; i++) L.prof[i] = 0; const unsigned long long now_ = (unsigned long long)clock64(); L.tprev[0] = (uint32_t)now_; L.tprev[1] = (uint32_t)(now_ >> 32); }

	for (;;) {
		// ---- next stream ----
		__syncthreads();
		if (tid == 0) {
			const uint32_t k = atomicAdd(ctr, 1u);
			L.jid = k < n ? (order ? order[k] : k) : 0xffffffffu;
			L.bail = 0; L.outn = 0; L.fl0 = 0; L.aoff = 0; L.lowest = 0; L.inblock = 0; L.act = A_NEXT; L.stall = 0;
			L.wvalid = 0; L.wb = 0; L.span_len = 0; L.done = 0; L.st_len = 0;
		}
		__syncthreads();
		const uint32_t jid = L.jid;
		if (jid == 0xffffffffu) break;
		const nxz_batch_job_t job = jobs[jid];
		const uint32_t off = (uint32_t)((uintptr_t)job.src & 15);
		const uint32_t nbytes = off + job.src_len;                     // the stream's bytes from its first 16-byte granule on
		const bool takes = job.resume == 0 && job.hist_len == 0 && job.src_len > 0 && job.src_len < (1u << 28) && ((uintptr_t)job.dst & 15) == 0;
		// (NXZ_JOB_SUSPEND_WHEN_FULL needs no look: an output that does not fit is handed back like everything this kernel does not do)
		if (!takes) {
			if (tid == 0) { const uint32_t at = atomicAdd(bail, 1u); bail[64 + at] = jid; if (dbg) atomicAdd(&dbg[R_JOB], 1u); }
			continue;
		}
		// a stream that may not fit LDS in one piece goes in spans sized by what it made of its source so far
		const bool longmode = nbytes > SRC_MAX || job.dst_cap > OUT_MAX + 40960;
		const NXZ_WG_GLOBAL uint8_t *gsrc = (const NXZ_WG_GLOBAL uint8_t *)(job.src - off);
		{
			v4u *b0 = (v4u *)L.mstart;
			const v4u z = { 0, 0, 0, 0 };
			for (uint32_t i = tid; i < OUT_MAX / 32 / 4; i += NT) b0[i] = z;
			if (tid == 0) L.pos = off * 8;
		}
		if (PROF && tid == 0) L.prof[P_STREAMS]++;

		// ---- the stream: window, block header, span after span, a flush when LDS is full ----
		for (;;) {
			__syncthreads();
			if (L.bail || L.done) break;
			const uint32_t outn = L.outn, made = L.aoff + outn;
			if (L.act == A_FLUSH || (OUT_MAX - outn < 4096 && outn > L.fl0)) {
				// (a flush frees 32 KiB at least: a piece that makes more than that is not this kernel's)
				if (outn <= HALF || outn == L.fl0) { __syncthreads(); if (tid == 0) L.bail = R_SPACE; continue; }
				flush_out((NXZ_WG_GLOBAL uint8_t *)job.dst, false, PROF);
				continue;
			}
			uint32_t spanbits = longmode ? 24576u * 8 : 0xffffffffu;
			if (made >= 4096 && (longmode || L.fl0)) {
				const uint32_t used = L.wb * 8 + L.pos - off * 8;
				const unsigned long long sbits = (unsigned long long)(OUT_MAX - outn) * used / made;
				spanbits = sbits > (1u << 28) ? 1u << 28 : (uint32_t)sbits + (uint32_t)(sbits >> 4);
				if (spanbits < 16384) spanbits = 16384;
			}
			{
				const uint32_t want = (spanbits < 49152u * 8 ? spanbits : 49152u * 8) + 4096 * 8;
				if (!L.wvalid || (!L.wend && (L.wbits < L.pos || L.wbits - L.pos < want))) {
					load_window(gsrc, nbytes, L.wb + (L.pos >> 3));
					WGPROF(P_LOAD);
				}
			}
			const uint32_t T = L.wbits;
			if (!L.inblock) {
				__syncthreads();
				if (tid == 0) {
					uint32_t p = L.pos;
					if (p + 3 > T) L.bail = R_HEADER;
					else {
						const uint32_t v = peek32(p);
						L.bfinal = v & 1; L.btype = (v >> 1) & 3;
						p += 3;
						if (L.btype == 0) {
							p = (p + 7) & ~7u;
							if (p + 32 > T) L.bail = R_STORED;
							else {
								const uint32_t w = peek32(p), len = w & 0xffff;
								p += 32;
								if (((w >> 16) ^ len) != 0xffff || (p >> 3) + len > nbytes - L.wb) L.bail = R_STORED;
								L.st_len = len;
							}
						} else if (L.btype == 3) L.bail = R_HEADER;
						L.pos = p;
						L.inblock = L.btype == 0 ? 2 : 1;
					}
				}
				__syncthreads();
				if (L.bail) continue;
				const uint32_t btype = L.btype;
				WGPROF(P_HEADER);
				// (the tables of the fixed code stand from the last block that used them -- the workgroup's last stream, as a rule: a batch of
				// fixed-code streams builds them once a workgroup, not once a stream)
				if (btype != 0 && !(btype == 1 && L.fixed_ok)) {
					// ---- code lengths ----
					if (btype == 2) {
						if (wave == 0) {
							const bool ok = read_header(L.pos, T, lane, true);
							if (!ok && lane == 0) L.bail = R_DHT;
						}
						__syncthreads();
						if (!L.bail) read_lengths(T);
					} else {
						for (int i = tid; i < 320; i += NT) L.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5);
						if (tid == 0) { L.hlit = 288; L.hdist = 30; }
					}
					__syncthreads();
					if (L.bail) continue;
					const int hlit = (int)L.hlit, hdist = (int)L.hdist;
					// (the checks of nxzi::read_dht: an end-of-block code, no code over-subscribed)
					if (btype == 2 && wave == 0) {
						uint32_t k1 = 0, k2 = 0;
						for (int i = lane; i < hlit; i += 64) if (L.lens[i]) k1 += 1u << (15 - L.lens[i]);
						if (lane < hdist && L.lens[hlit + lane]) k2 = 1u << (15 - L.lens[hlit + lane]);
						k1 = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(k1, lane), 63);
						k2 = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(k2, lane), 63);
						if (lane == 0 && (L.lens[256] == 0 || k1 > (1u << 15) || k2 > (1u << 15))) L.bail = R_DHT;
					}
					__syncthreads();
					if (L.bail) continue;
					if (PROF) { WGPROF(P_DHT); if (tid == 0) L.prof[P_BLOCKS]++; }
					if (tid == 0) L.fixed_ok = 0;
					build_tables(hlit, hdist);
					if (L.bail) continue;
					if (tid == 0) L.fixed_ok = btype == 1;
					WGPROF(P_TABLES);
				}
			}
			if (L.inblock == 2) {
				// ---- a stored block, or as much of it as LDS has room for ----
				const uint32_t rem = L.st_len, pos = L.pos, room = OUT_MAX - outn;
				const uint32_t k = rem < room ? rem : room;
				if (rem > job.dst_cap - made) { __syncthreads(); if (tid == 0) L.bail = R_SPACE; continue; }
				if (k == 0 && rem) { __syncthreads(); if (tid == 0) L.act = A_FLUSH; continue; }
				const uint32_t from = pos >> 3;
				const bool in_window = L.wvalid && pos <= T && from + k <= (T >> 3);
				if (in_window) for (uint32_t i = tid; i < k; i += NT) ob[outn + i] = sb[from + i];
				else {
					const NXZ_WG_GLOBAL uint8_t *gs = gsrc + L.wb + from;
					for (uint32_t i = tid; i < k; i += NT) ob[outn + i] = gs[i];
				}
				__syncthreads();
				if (tid == 0) {
					L.pos = pos + 8 * k; L.outn = outn + k; L.st_len = rem - k;
					if (!in_window) L.wvalid = 0;
					if (rem == k) { L.inblock = 0; if (L.bfinal) L.done = 1; }
				}
				WGPROF(P_HEADER);
				continue;
			}
			decode_span(T, spanbits, job.dst_cap - made, pmin_bits & 0xffff, pmin_bits >> 16 | coop << 16, PROF);
		}
		__syncthreads();
		if (L.bail) {
			if (tid == 0) {
				const uint32_t at = atomicAdd(bail, 1u);
				bail[64 + at] = jid;
				if (dbg) atomicAdd(&dbg[L.bail < 16 ? L.bail : 0], 1u);
			}
			continue;
		}
		flush_out((NXZ_WG_GLOBAL uint8_t *)job.dst, true, PROF);
		if (tid == 0) {
			// (the record of nxzl::inflate_lanes_kernel for a stream that ran to its final end-of-block)
			nxz_batch_result_t r;
			uint32_t spbc = job.src_len, subc = (nbytes - L.wb) * 8 - L.pos;
			if (subc > 0xfff8) { const uint32_t drop = (subc - 0xfff8 + 7) / 8; spbc -= drop; subc -= drop * 8; }
			r.cc = subc < 8 ? 0 : NXZ_CC_DATA_LENGTH;
			r.tpbc = L.aoff + L.outn; r.tebc = 0; r.spbc = spbc; r.crc = 0; r.adler = 0; r.subc = subc; r.sfbt = 0x100u;
			results[jid] = r;
		}
	}
	if (PROF && tid == 0) for (int i = 0; i < P_N; i++) atomicAdd(&prof[i], (unsigned long long)L.prof[i]);
#undef WGPROF
}

} // namespace nxzw

#ifndef NXZ_CPU_SIM
extern "C" int nxz_launch_inflate_order_only(const nxz_batch_job_t *jobs, size_t nslots, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io,
					     const uint32_t *order, hipStream_t stream);

// control words of a launch: the job counter, the reasons, then the hand-back list (a count, the indices from word 64 on)
extern "C" size_t nxz_inflate_wg_workspace(size_t n)
{
	return 1024 + 256 + ((n * sizeof(uint32_t) + 255) & ~(size_t)255);
}

// All n streams a workgroup each; the streams the kernel hands back are decoded behind it by the kernel that knows every
// case, a wavefront each (nxzi::inflate_kernel through the list: a slot that holds no job ends at once).  order (may be
// NULL): the jobs by falling source length.  Checksums by nxzl::cksum_kernel -- with targets (may be NULL), the outputs go
// there in the same pass (the rounds of nxu_run_job: device buffers to pinned host memory).
extern "C" int nxz_launch_inflate_wg(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io,
				     uint8_t *wg_ws, const uint32_t *order, uint8_t *const *targets, hipStream_t stream)
{
	if (!n) return 0;
	if (n >= (1u << 31)) return (int)hipErrorInvalidValue;
	uint32_t *ctr = (uint32_t *)wg_ws, *dbg = (uint32_t *)(wg_ws + 256), *bail = (uint32_t *)(wg_ws + 1024);
	unsigned long long *prof = (unsigned long long *)(wg_ws + 320);
	(void)hipMemsetAsync(wg_ws, 0, 1024 + 256, stream);
	(void)hipMemsetAsync(bail + 64, 0xff, n * sizeof(uint32_t), stream);
	static const unsigned cus = [] { int dev = 0, v = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return (unsigned)(v > 0 ? v : 256); }();
	const char *pe = getenv("NXZ_WG_PMIN");                            // (read at every call: the tests switch it)
	const unsigned pmin = pe && atoi(pe) >= 64 ? (unsigned)atoi(pe) : 128u;
	const char *mre = getenv("NXZ_WG_ROUNDS");
	const unsigned maxr = mre && atoi(mre) >= 2 && atoi(mre) <= 1024 ? (unsigned)atoi(mre) : 1024u;
	const char *nre = getenv("NXZ_WG_COOP");                           // so many pieces left in a round or fewer: a wavefront walks each
	const unsigned coop = nre && atoi(nre) >= 0 && atoi(nre) <= 1024 ? (unsigned)atoi(nre) : 16u;
	const unsigned grid = (unsigned)(n < cus ? n : cus);
	const char *pr = getenv("NXZ_WG_PROF");
	if (pr && atoi(pr)) hipLaunchKernelGGL(nxzw::inflate_wg_kernel<true>, dim3(grid), dim3(nxzw::NT), 0, stream, jobs, (uint32_t)n, results, order, ctr, bail, pmin | maxr << 16, coop, dbg, prof);
	else hipLaunchKernelGGL(nxzw::inflate_wg_kernel<false>, dim3(grid), dim3(nxzw::NT), 0, stream, jobs, (uint32_t)n, results, order, ctr, bail, pmin | maxr << 16, coop, dbg, prof);
	int rc = (int)hipGetLastError();
	if (rc) return rc;
	rc = nxz_launch_inflate_order_only(jobs, n, results, dht_io, bail + 64, stream);
	if (rc) return rc;
	return targets ? nxz_launch_cksum_copy(jobs, n, results, targets, stream) : nxz_launch_cksum(jobs, n, results, stream);
}

// (diagnostic / tests: the reasons of the last launch on this workspace, 16 words; [0] unused, [15] = streams handed back; the caller has waited for the stream)
extern "C" int nxz_inflate_wg_reasons(const uint8_t *wg_ws, uint32_t *out16)
{
	int rc = (int)hipMemcpy(out16, wg_ws + 256, 15 * sizeof(uint32_t), hipMemcpyDeviceToHost);
	if (!rc) rc = (int)hipMemcpy(out16 + 15, wg_ws + 1024, sizeof(uint32_t), hipMemcpyDeviceToHost);
	return rc;
}
// (NXZ_WG_PROF=1: thread 0's cycles by phase and the counts, 12 words -- nxzw::P_*)
extern "C" int nxz_inflate_wg_prof(const uint8_t *wg_ws, unsigned long long *out12)
{
	return (int)hipMemcpy(out12, wg_ws + 320, nxzw::P_N * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#endif
