// nxz_blockfind.hip -- block-boundary speculation for ONE long deflate stream (gfx950, wave64).
//
// A deflate stream is serial: the reference's inflate loop (/root/reference lib/nx_inflate.c:1060-1762)
// feeds the accelerator job after job, each resuming where the last one stopped, and relies on
// an engine that is fast on a single stream (7 GB/s on POWER9, samples/simpleapi/README:27-30).
// A GPU gets there by decoding many pieces of the stream at once, and for that it must know
// where deflate blocks start inside it.  find_blocks_kernel scans every bit position of every
// segment (1 - 8 KiB) of the compressed stream for a plausible header of a dynamic-Huffman
// block (RFC 1951 3.2.7) and reports the first one per segment:
//   BTYPE = 10 (BFINAL either way: a stream's last block is as long as any), HLIT <= 29, HDIST <= 29,
//   the code-length code complete (Kraft sum exactly 1),
//   the HLIT + 257 + HDIST + 1 code lengths decode without a bad repeat or an overrun,
//   end-of-block has a code, the literal/length code is complete, the distance code is
//   complete, empty or a single code of one bit (what zlib's inflate_table accepts).
// Nothing but a real header passes all of that in practice (a wrong guess would be caught later:
// the piece in front of it then does not end at a block header, and the stream's CRC-32 is
// checked at the end).  Stored and fixed blocks are not searched for (a stored header is only
// LEN == ~NLEN, one position in 65536 passes by chance): a stretch of such blocks is decoded as
// part of the piece in front of it.
//
// The other kernels here move data between the rounds of the speculative decode (nxz_engine.cpp,
// nxz_inflate_stream): copy_items_kernel (a list of byte ranges, workgroup per item).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "nxz_device.h"

namespace nxzb {

constexpr int NT = 256;
constexpr uint32_t SEG = 8192;                  // bytes of the stream per workgroup
constexpr uint32_t LOOK = 320;                  // a header is at most 17 + 19 * 3 + 316 * 7 + ... bits: < 300 bytes
constexpr uint32_t MAXCAND = 1024;
constexpr uint32_t LEFT_MAX = 126;                // positions per segment handed to check_headers_kernel
constexpr uint32_t QCHUNK = 8192;               // positions per pass of the first test (its survivors queue up in LDS)

__device__ __forceinline__ uint32_t peek(const uint8_t *s, uint32_t bit, uint32_t n)   // n <= 25
{
	const uint32_t by = bit >> 3;
	const uint32_t w = (uint32_t)s[by] | (uint32_t)s[by + 1] << 8 | (uint32_t)s[by + 2] << 16 | (uint32_t)s[by + 3] << 24;
	return (w >> (bit & 7)) & ((1u << n) - 1);
}

// the full check of a candidate (one lane); `limit` = first bit that is not part of the stream.
// Everything the lane keeps per candidate is packed into registers (the code-length code's lengths, the
// counts per length, the symbols in canonical order: 3, 5 and 5 bits an entry) -- arrays indexed at run
// time would live in scratch memory, a trip to the caches per symbol -- and a code-length symbol costs
// one look at the source (14 bits: the longest code and the longest repeat count).
// (max_syms: give up -- answer "may be one" -- after that many code-length symbols without a contradiction: the
// first sieve, a lane per survivor; what it lets through goes to header_ok_wave)
__device__ bool header_ok(const uint8_t *s, uint32_t bit, uint32_t limit, uint32_t max_syms)
{
	const uint32_t *s32 = (const uint32_t *)s;                     // (the segment starts on a 16-byte boundary of LDS)
	if (bit + 17 > limit) return false;
	uint32_t v = peek(s, bit, 17);
	const uint32_t hlit = ((v >> 3) & 31) + 257, hdist = ((v >> 8) & 31) + 1, hclen = ((v >> 13) & 15) + 4;
	uint32_t pos = bit + 17;
	if (pos + 3 * hclen > limit) return false;
	// code-length code: lengths packed 3 bits per symbol, counts per length 5 bits per length
	uint64_t cll = 0;
	uint64_t cnt = 0;
	uint32_t lastl = 0;
	{
		const uint64_t raw = (uint64_t)peek(s, pos, 24) | ((uint64_t)peek(s, pos + 24, 24) << 24) | ((uint64_t)peek(s, pos + 48, 9) << 48);   // 19 x 3 bits
		constexpr uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
#pragma unroll
		for (uint32_t i = 0; i < 19; i++) {
			const uint32_t l = i < hclen ? (uint32_t)(raw >> (3 * i)) & 7 : 0;
			cll |= (uint64_t)l << (3 * order[i]);
			cnt += 1ull << (5 * l);
			if (i + 1 == hclen) lastl = l;
		}
		pos += 3 * hclen;
	}
	// canonical order: the symbols of length 1 in ascending order, then those of length 2, ... (5 bits each, 19 at most);
	// and per length l, eight bits each: lim -- the first 7-bit value (code bits first, left-justified) that is no code of
	// length <= l; bas -- 128 + (place of the first symbol of length l in that order) - (first code of length l)
	uint64_t st_lo = 0, st_hi = 0;                                 // entries 0..11, 12..18
	uint64_t lim = 0, bas = 0;
	{
		uint64_t offs = 0;                                     // first entry of each length, 5 bits per length
		uint32_t o = 0, code = 0;
#pragma unroll
		for (uint32_t l = 1; l < 8; l++) {
			const uint32_t c = (uint32_t)(cnt >> (5 * l)) & 31;
			offs |= (uint64_t)o << (5 * l);
			lim |= (uint64_t)((code + c) << (7 - l)) << (8 * l);
			bas |= (uint64_t)((128 + o - code) & 0xff) << (8 * l);
			o += c; code = (code + c) << 1;
		}
#pragma unroll
		for (uint32_t sy = 0; sy < 19; sy++) {
			const uint32_t l = (uint32_t)(cll >> (3 * sy)) & 7;
			if (l) {
				const uint32_t at = (uint32_t)(offs >> (5 * l)) & 31;
				if (at < 12) st_lo |= (uint64_t)sy << (5 * at); else st_hi |= (uint64_t)sy << (5 * (at - 12));
				offs += 1ull << (5 * l);
			}
		}
	}
	// every encoder trims the three counts to the last length that is not zero (zlib: build_bl_tree /
	// send_all_trees; so do libdeflate, miniz, 7-zip, zopfli and this engine): a header whose last
	// code-length-code length, last literal/length length or last distance length is zero although
	// the count could have been smaller is taken for chance.  (A stream of an encoder that does not
	// trim offers fewer starts here and is decoded in longer pieces, or job after job.)
	// The exception is the POWER NX engine's table generator (and this engine's, which makes the same tables:
	// /root/reference lib/nx_dhtgen.c:628-648): it always sends all 286 + 30 lengths with all 19
	// code-length-code lengths, used or not -- that signature is taken as it comes.
	const bool nx_made = hlit == 286 && hdist == 30 && hclen == 19;
	if (hclen > 4 && lastl == 0 && !nx_made) return false;
	uint32_t n = 0, prev = 0, kraft_ll = 0, kraft_d = 0, nd = 0, maxd = 0, eob = 0, last_ll = 0, last_d = 0;
	const uint32_t total = hlit + hdist;
	uint32_t nsym = 0;
	while (n < total) {
		if (nsym++ >= max_syms) return true;
		// one code-length symbol: its code (7 bits at most) and what follows it (7 at most).  The code's length is one
		// more than the number of lengths whose limit the 7 bits reach (the code-length code is complete -- the Kraft
		// sum was checked -- so there is always one); no branches, two dwords of the source.  (Round 3's first form
		// tried the seven lengths in turn, ~280 instructions with their branches: 1 us a symbol for a lane on its own,
		// 100-300 us for a real header -- and the search took as long as its real headers.)
		if (pos >= limit) return false;
		const uint32_t w = __builtin_amdgcn_alignbit(s32[(pos >> 5) + 1], s32[pos >> 5], pos & 31);
		const uint32_t c7 = __builtin_bitreverse32(w) >> 25;
		const uint32_t llo = (uint32_t)lim, lhi = (uint32_t)(lim >> 32);
		const uint32_t len = 1 + (c7 >= ((llo >> 8) & 0xff)) + (c7 >= ((llo >> 16) & 0xff)) + (c7 >= (llo >> 24)) +
				     (c7 >= (lhi & 0xff)) + (c7 >= ((lhi >> 8) & 0xff)) + (c7 >= ((lhi >> 16) & 0xff));
		const uint32_t at = (((uint32_t)(bas >> (8 * len)) & 0xff) + (c7 >> (7 - len)) - 128) & 31;
		const uint32_t sym = (uint32_t)((at < 12 ? st_lo : st_hi) >> (5 * (at < 12 ? at : at - 12))) & 31;
		if (pos + len > limit) return false;
		pos += len;
		const uint32_t x = w >> len;
		uint32_t rep = 1, val = sym;
		if (sym == 16) {
			if (n == 0 || pos + 2 > limit) return false;
			rep = 3 + (x & 3); pos += 2; val = prev;
		} else if (sym == 17) {
			if (pos + 3 > limit) return false;
			rep = 3 + (x & 7); pos += 3; val = 0;
		} else if (sym == 18) {
			if (pos + 7 > limit) return false;
			rep = 11 + (x & 127); pos += 7; val = 0;
		}
		if (n + rep > total) return false;
		if (sym < 16) prev = sym; else if (sym != 16) prev = 0;
		if (val) {
			// (a run of one length: what of it is literal/length codes, what distance codes)
			const uint32_t e = n + rep;
			const uint32_t nl = n < hlit ? (e < hlit ? e : hlit) - n : 0, ndist = rep - nl;
			kraft_ll += nl << (15 - val);
			kraft_d += ndist << (15 - val);
			nd += ndist;
			if (ndist && val > maxd) maxd = val;
			if (n <= 256 && e > 256) eob = 1;
			if (nl && n + nl == hlit) last_ll = 1;
			if (ndist && e == total) last_d = 1;
		}
		n += rep;
		// (lengths read off chance bits oversubscribe a code within a few dozen symbols: no need to go on)
		if (kraft_ll > (1u << 15) || kraft_d > (1u << 15)) return false;
	}
	if (!eob || kraft_ll != (1u << 15)) return false;
	if (!nx_made && ((hlit > 257 && !last_ll) || (hdist > 1 && !last_d))) return false;
	if (!(kraft_d == (1u << 15) || nd == 0 || (nd == 1 && maxd == 1))) return false;
	return true;
}

// The same check by a whole wavefront (all lanes call it with the same candidate; the answer is the same in all):
// a lane on its own issues an instruction every five cycles or so and needs ~150 of them per code-length symbol,
// 100 us for a real header -- which is what the search cost when every survivor had a lane of its own and the
// wavefront waited for its slowest lane.  Here the header's bits (at most 2283 + 17) sit in two registers per
// lane, a symbol's bits are fetched with v_readlane, the 7-bit code-length code is looked up in two more
// registers (as the inflate kernel reads tables: nxz_inflate.hip read_dht), and the bookkeeping is scalar.
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ bool header_ok_wave(const uint32_t *s32, uint32_t ndw, uint32_t bit, uint32_t limit, int lane)
{
	bit = uni32(bit);
	if (bit + 17 > limit) return false;
	const uint32_t d0 = bit >> 5;
	const uint32_t R0 = d0 + lane < ndw ? s32[d0 + lane] : 0, R1 = d0 + 64 + lane < ndw ? s32[d0 + 64 + lane] : 0;
	// up to 25 bits at bit p (p >= bit), wave-uniform
	auto peek = [&](uint32_t p) __attribute__((always_inline)) -> uint32_t {
		const uint32_t o = uni32(p - d0 * 32), i = o >> 5, sh = o & 31;
		const uint32_t lo = i < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)R0, (int)i) : (uint32_t)__builtin_amdgcn_readlane((int)R1, (int)(i & 63));
		const uint32_t j = i + 1;
		const uint32_t hi = j < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)R0, (int)j) : (uint32_t)__builtin_amdgcn_readlane((int)R1, (int)(j & 63));
		return (uint32_t)(((((uint64_t)hi << 32) | lo) >> sh));
	};
	const uint32_t v = peek(bit);
	const uint32_t hlit = ((v >> 3) & 31) + 257, hdist = ((v >> 8) & 31) + 1, hclen = ((v >> 13) & 15) + 4;
	uint32_t pos = bit + 17;
	if (pos + 3 * hclen > limit) return false;
	// the code-length code: lane = symbol; its 3-bit length is the inv[symbol]-th that is sent
	uint32_t myl = 0;
	{
		// position of symbol sy in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15, five bits each
		const uint64_t inv_lo = 3ull | 17ull << 5 | 15ull << 10 | 13ull << 15 | 11ull << 20 | 9ull << 25 | 7ull << 30 | 5ull << 35 | 4ull << 40 | 6ull << 45 | 8ull << 50 | 10ull << 55;
		const uint64_t inv_hi = 12ull | 14ull << 5 | 16ull << 10 | 18ull << 15 | 0ull << 20 | 1ull << 25 | 2ull << 30;
		const uint32_t sy = (uint32_t)lane;
		const uint32_t at = sy < 12 ? (uint32_t)(inv_lo >> (5 * sy)) & 31 : sy < 19 ? (uint32_t)(inv_hi >> (5 * (sy - 12))) & 31 : 31;
		if (at < hclen) {
			const uint32_t o = pos + 3 * at;
			const uint64_t w = (uint64_t)s32[o >> 5] | ((uint64_t)s32[(o >> 5) + 1] << 32);
			myl = (uint32_t)(w >> (o & 31)) & 7;
		}
	}
	const bool nx_made = hlit == 286 && hdist == 30 && hclen == 19;
	// (the last length that is sent belongs to symbol order[hclen - 1]: the lane whose `at` is hclen - 1)
	{
		const uint64_t inv_lo = 3ull | 17ull << 5 | 15ull << 10 | 13ull << 15 | 11ull << 20 | 9ull << 25 | 7ull << 30 | 5ull << 35 | 4ull << 40 | 6ull << 45 | 8ull << 50 | 10ull << 55;
		const uint64_t inv_hi = 12ull | 14ull << 5 | 16ull << 10 | 18ull << 15 | 0ull << 20 | 1ull << 25 | 2ull << 30;
		const uint32_t sy = (uint32_t)lane;
		const uint32_t at = sy < 12 ? (uint32_t)(inv_lo >> (5 * sy)) & 31 : sy < 19 ? (uint32_t)(inv_hi >> (5 * (sy - 12))) & 31 : 31;
		const uint64_t last_nonzero = __ballot(at == hclen - 1 && myl != 0);
		if (hclen > 4 && !last_nonzero && !nx_made) return false;
	}
	pos += 3 * hclen;
	// canonical codes by ranks (lane = symbol), then the look-up: entries `lane` and `lane + 64` of the 7-bit table,
	// symbol | length << 5, 0xff = no code
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
		if (kraft != 128) return false;
		const uint32_t myrev = myl ? __builtin_bitreverse32(mycode) >> (32 - myl) : 0;
		for (int sy = 0; sy < 19; sy++) {
			const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)myl, sy);
			if (!l) continue;
			const uint32_t rev = (uint32_t)__builtin_amdgcn_readlane((int)myrev, sy), mask = (1u << l) - 1;
			if (((uint32_t)lane & mask) == rev) tlo = (uint32_t)sy | (l << 5);
			if ((((uint32_t)lane + 64) & mask) == rev) thi = (uint32_t)sy | (l << 5);
		}
	}
	uint32_t n = 0, prev = 0, kraft_ll = 0, kraft_d = 0, nd = 0, maxd = 0, eob = 0, last_ll = 0, last_d = 0;
	const uint32_t total = hlit + hdist;
	while (n < total) {
		if (pos >= limit) return false;
		const uint32_t w = peek(pos);
		const uint32_t k = w & 127;
		const uint32_t e = k < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)tlo, (int)k) : (uint32_t)__builtin_amdgcn_readlane((int)thi, (int)(k - 64));
		if (e == 0xff) return false;
		const uint32_t sym = e & 31, len = e >> 5;
		if (pos + len > limit) return false;
		pos += len;
		const uint32_t x = w >> len;
		uint32_t rep = 1, val = sym;
		if (sym == 16) {
			if (n == 0 || pos + 2 > limit) return false;
			rep = 3 + (x & 3); pos += 2; val = prev;
		} else if (sym == 17) {
			if (pos + 3 > limit) return false;
			rep = 3 + (x & 7); pos += 3; val = 0;
		} else if (sym == 18) {
			if (pos + 7 > limit) return false;
			rep = 11 + (x & 127); pos += 7; val = 0;
		}
		if (n + rep > total) return false;
		if (sym < 16) prev = sym; else if (sym != 16) prev = 0;
		if (val) {
			const uint32_t en = n + rep;
			const uint32_t nl = n < hlit ? (en < hlit ? en : hlit) - n : 0, ndist = rep - nl;
			kraft_ll += nl << (15 - val);
			kraft_d += ndist << (15 - val);
			nd += ndist;
			if (ndist && val > maxd) maxd = val;
			if (n <= 256 && en > 256) eob = 1;
			if (nl && n + nl == hlit) last_ll = 1;
			if (ndist && en == total) last_d = 1;
		}
		n += rep;
		if (kraft_ll > (1u << 15) || kraft_d > (1u << 15)) return false;
	}
	if (!eob || kraft_ll != (1u << 15)) return false;
	if (!nx_made && ((hlit > 257 && !last_ll) || (hdist > 1 && !last_d))) return false;
	if (!(kraft_d == (1u << 15) || nd == 0 || (nd == 1 && maxd == 1))) return false;
	return true;
}

// a segment of the stream (and the LOOK bytes behind it a header may reach into) into LDS, zeros behind the stream's end
template <int THREADS>
__device__ __forceinline__ void load_segment(uint8_t *s, const uint8_t *__restrict__ src, uint64_t base, uint32_t have, uint32_t seg_bytes, int t)
{
	if ((((uintptr_t)src + base) & 15) == 0) {
		// (16 bytes a lane: a byte a lane is 34 trips to device memory, one after the other, for an 8 KiB segment)
		const uint32_t whole = have >> 4;
		for (uint32_t i = t; i < (seg_bytes + LOOK + 16) / 16; i += THREADS) {
			uint4 v = make_uint4(0, 0, 0, 0);
			if (i < whole) v = ((const uint4 *)(src + base))[i];
			((uint4 *)s)[i] = v;
		}
		__syncthreads();
		for (uint32_t i = whole * 16 + t; i < have; i += THREADS) s[i] = src[base + i];
	} else
		for (uint32_t i = t; i < seg_bytes + LOOK + 16; i += THREADS) s[i] = i < have ? src[base + i] : 0;
}

// first[seg] = bit position (in the whole stream) of the first plausible dynamic block header that
// starts inside segment seg, or 0xffffffffffffffff
// (seg: bytes of the stream per workgroup, <= SEG -- a short stream is cut finer, so that the search takes less
// long and blocks of a KiB or two are not hidden behind the first of their segment)
__global__ __launch_bounds__(NT) void find_blocks_kernel(const uint8_t *__restrict__ src, uint64_t srclen, uint64_t first_bit,
							   uint64_t *__restrict__ first, uint32_t nseg, uint32_t seg_bytes, uint32_t diag,
							   uint16_t *__restrict__ left)
{
	__shared__ __attribute__((aligned(16))) uint8_t s[SEG + LOOK + 16];
	__shared__ uint32_t cand[MAXCAND];
	__shared__ uint16_t queue[QCHUNK];
	__shared__ uint32_t ncand, best, nq;
	__shared__ uint8_t klut[512];                   // Kraft sum, in 1/128, of three code lengths of 3 bits each
	__shared__ __attribute__((aligned(8))) uint32_t hmask[16][2];   // the bits of the first HCLEN (4..19) lengths
	const int t = threadIdx.x;
	const uint32_t seg = blockIdx.x;
	if (seg >= nseg) return;
	for (uint32_t i = t; i < 512; i += NT) {
		uint32_t k = 0;
		for (uint32_t f = 0; f < 3; f++) { const uint32_t l = (i >> (3 * f)) & 7; k += l ? 128u >> l : 0; }
		klut[i] = (uint8_t)k;
	}
	if (t < 16) {
		const uint64_t m = (1ull << (3 * (t + 4))) - 1;
		hmask[t][0] = (uint32_t)m; hmask[t][1] = (uint32_t)(m >> 32);
	}
	const uint64_t base = (uint64_t)seg * seg_bytes;
	const uint32_t have = (uint32_t)(srclen - base < seg_bytes + LOOK ? srclen - base : seg_bytes + LOOK);
	load_segment<NT>(s, src, base, have, seg_bytes, t);
	if (t == 0) { ncand = 0; best = 0xffffffffu; }
	__syncthreads();
	const uint32_t limit = have * 8;
	const uint32_t nbits = (have < seg_bytes ? have : seg_bytes) * 8;
	// phase 1: the cheap part of the test at every bit position of the segment, in two steps per chunk
	// of 8192 positions so that the lanes stay busy: (a) every lane looks at the 13 bits that decide for
	// three positions in four (BTYPE 10, HLIT <= 29, HDIST <= 29) and the survivors -- 22 % --
	// are packed into a queue; (b) the queue, a lane per entry, gets the Kraft sum of the code-length
	// code (up to 19 three-bit lengths from four dwords).  Consecutive lanes test consecutive
	// positions in (a), so the dwords they read are the same LDS words for 32 lanes: broadcast reads.
	const uint32_t *s32 = (const uint32_t *)s;
	if (diag == 3) { if (t == 0) first[seg] = ~0ull - s[have - 1]; return; }      // (timing of the load alone: tools)
	for (uint32_t c0 = 0; c0 < nbits; c0 += QCHUNK) {
		if (t == 0) nq = 0;
		__syncthreads();
		const uint32_t c1 = c0 + QCHUNK < nbits ? c0 + QCHUNK : nbits;
		// (a) 32 positions per lane at once, bitwise: position p passes when bits p+1, p+2 are 0, 1 and
		// neither bits p+4..p+7 (HLIT 30, 31) nor bits p+9..p+12 (HDIST 30, 31) are all ones
		for (uint32_t w0 = (c0 >> 5) + t; w0 * 32 < c1; w0 += NT) {
			const uint64_t w = (uint64_t)s32[w0] | ((uint64_t)s32[w0 + 1] << 32);
			const uint64_t hl = (w >> 4) & (w >> 5) & (w >> 6) & (w >> 7), hd = (w >> 9) & (w >> 10) & (w >> 11) & (w >> 12);
			uint32_t m = (uint32_t)(~(w >> 1) & (w >> 2) & ~hl & ~hd);
			const uint32_t p0 = w0 * 32;
			if (p0 + 32 > c1) m &= (1u << (c1 - p0)) - 1;                       // positions of this chunk only
			if (base * 8 + p0 < first_bit) m &= first_bit - base * 8 - p0 >= 32 ? 0u : ~0u << (uint32_t)(first_bit - base * 8 - p0);
			if (p0 + 32 + 17 > limit) { for (uint32_t k = 0; k < 32; k++) if (p0 + k + 17 > limit) m &= ~(1u << k); }
			if (m) {
				uint32_t at = atomicAdd(&nq, (uint32_t)__popc(m));
				while (m) { queue[at++] = (uint16_t)(p0 + (uint32_t)__builtin_ctz(m)); m &= m - 1; }
			}
		}
		__syncthreads();
		const uint32_t n1 = diag == 2 ? 0 : nq;                                    // (diag 2: without the Kraft sums)
		for (uint32_t k = t; k < n1; k += NT) {
			const uint32_t p = queue[k];
			const uint32_t wi = p >> 5, sh = p & 31;
			const uint32_t d0 = s32[wi], d1 = s32[wi + 1], d2 = s32[wi + 2], d3 = s32[wi + 3];
			const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh), mid = __builtin_amdgcn_alignbit(d2, d1, sh), hi = __builtin_amdgcn_alignbit(d3, d2, sh);
			const uint32_t hclen = ((lo >> 13) & 15) + 4;
			if (p + 17 + 3 * hclen > limit) continue;
			// the 3-bit lengths in the order they are sent, those beyond HCLEN masked off; their Kraft sum three
			// lengths a look-up (19 compare-shift-select-adds cost three times as much, and this loop is what the
			// search spends its time in)
			const uint2 hm = ((const uint2 *)hmask)[hclen - 4];
			const uint32_t xlo = __builtin_amdgcn_alignbit(mid, lo, 17) & hm.x, xhi = __builtin_amdgcn_alignbit(hi, mid, 17) & hm.y;
			const uint32_t kraft = (uint32_t)klut[xlo & 511] + klut[(xlo >> 9) & 511] + klut[(xlo >> 18) & 511] +
					       klut[__builtin_amdgcn_alignbit(xhi, xlo, 27) & 511] + klut[(xhi >> 4) & 511] + klut[(xhi >> 13) & 511] + klut[(xhi >> 22) & 7];
			if (kraft != 128) continue;
			const uint32_t kc = atomicAdd(&ncand, 1u);
			if (kc < MAXCAND) cand[kc] = p;
		}
	}
	__syncthreads();
	if (diag == 1 || diag == 2) { if (t == 0) first[seg] = ~0ull - ncand; return; }          // (timing of phase 1 alone: tools)
	// phase 2: the whole header, a lane per survivor.  Measured on 256 MiB of the corpus at zlib -6 (8 KiB segments): 55
	// survivors a segment, and chance bits do NOT contradict themselves soon -- 42 of the 55 are still alive after 4
	// code-length symbols, 35 after 12, 20 after 40, 9 after 80 (their code-length codes give the long lengths and the
	// zero runs the short codes, so the Kraft sums fill up as slowly as a real header's) -- while a lane needs ~150
	// instructions a symbol.  With the survivors spread over the workgroup's four wavefronts every wavefront had such
	// a lane and ran its 100+ symbols at 1/64 of its width: 1.13 of the search's 1.42 ms.  So: the first 8 symbols by
	// all lanes, then what is left (38 of 55) packed into the lanes of ONE wavefront for the whole header.
	const uint32_t nc0 = ncand < MAXCAND ? ncand : MAXCAND;
	if (diag != 5) {
		const uint32_t s1 = diag >= 200 ? diag - 200 : 8;
		__syncthreads();
		if (t == 0) nq = 0;
		__syncthreads();
		for (uint32_t k = t; k < nc0; k += NT) {
			const uint32_t p = cand[k];
			if (header_ok(s, p, limit, s1)) queue[atomicAdd(&nq, 1u)] = (uint16_t)p;
		}
		__syncthreads();
		const uint32_t n2 = nq;
		if (left) {
			// what is left goes to check_headers_kernel: LEFT_MAX positions per segment, their number in front.
			// (This workgroup holds 30 KiB of LDS for the search; with five of them on a CU, each down to a few lanes
			// walking a few hundred symbols, the CU did next to nothing for 0.8 of the search's 1.1 ms.  The second kernel
			// is a wavefront and 8.5 KiB a segment: eighteen of them on a CU.)
			uint16_t *o = left + (size_t)seg * (LEFT_MAX + 2);
			const uint32_t n3 = n2 < LEFT_MAX ? n2 : LEFT_MAX;
			if (t == 0) { o[0] = (uint16_t)n3; o[1] = (uint16_t)(n2 > LEFT_MAX); }
			for (uint32_t k = t; k < n3; k += NT) o[2 + k] = queue[k];
			if (n2 <= LEFT_MAX) return;
			// (more than the list holds -- data made to look like headers: this workgroup does them all, as before)
		}
		for (uint32_t k = t; k < n2; k += NT) {
			const uint32_t p = queue[k];
			if (p < best && header_ok(s, p, limit, 0xffffffffu)) atomicMin(&best, p);
		}
		__syncthreads();
		if (t == 0) first[seg] = best == 0xffffffffu ? ~0ull : base * 8 + best;
		return;
	}
	// (diag 5, tools: the form short segments had up to round 3 -- 40 symbols by the lane, then a wavefront per survivor)
	__syncthreads();
	if (t == 0) nq = 0;
	__syncthreads();
	for (uint32_t k = t; k < nc0; k += NT) {
		const uint32_t p = cand[k];
		if (header_ok(s, p, limit, diag >= 100 ? diag - 100 : 40)) queue[atomicAdd(&nq, 1u)] = (uint16_t)p;
	}
	__syncthreads();
	if (diag >= 100) { if (t == 0) first[seg] = ~0ull - nq; return; }           // (how many the sieve lets through, and its time: tools)
	// ... then the whole header, a wavefront per survivor of that
	const uint32_t nc = nq;
	for (uint32_t k = (uint32_t)t >> 6; k < nc; k += NT / 64) {
		const uint32_t p = uni32(queue[k]);
		if (p >= uni32(best)) continue;
		if (header_ok_wave(s32, (SEG + LOOK + 16) / 4, p, limit, t & 63) && (t & 63) == 0) atomicMin(&best, p);
	}
	__syncthreads();
	if (t == 0) first[seg] = best == 0xffffffffu ? ~0ull : base * 8 + best;
}

// the whole header for the positions find_blocks_kernel left over, a wavefront per segment (see there)
__global__ __launch_bounds__(64) void check_headers_kernel(const uint8_t *__restrict__ src, uint64_t srclen, uint64_t *__restrict__ first,
							    uint32_t nseg, uint32_t seg_bytes, const uint16_t *__restrict__ left)
{
	__shared__ __attribute__((aligned(16))) uint8_t s[SEG + LOOK + 16];
	__shared__ uint32_t best, npass;
	const int t = threadIdx.x;
	const uint32_t seg = blockIdx.x;
	if (seg >= nseg) return;
	const uint16_t *o = left + (size_t)seg * (LEFT_MAX + 2);
	const uint32_t n = o[0];
	if (o[1] || !n) { if (t < NXZ_BLOCKFIND_MORE) first[nseg + (size_t)seg * NXZ_BLOCKFIND_MORE + t] = ~0ull; }
	if (o[1]) return;                                  // (the first kernel did this segment itself)
	if (!n) { if (t == 0) first[seg] = ~0ull; return; }
	const uint64_t base = (uint64_t)seg * seg_bytes;
	const uint32_t have = (uint32_t)(srclen - base < seg_bytes + LOOK ? srclen - base : seg_bytes + LOOK);
	// (a header reaches LOOK bytes at most: what lies behind the last candidate's is not wanted)
	load_segment<64>(s, src, base, have, seg_bytes, t);
	if (t == 0) best = 0xffffffffu;
	__syncthreads();
	const uint32_t limit = have * 8;
	// (round 5: not the first header of the segment alone but up to NXZ_BLOCKFIND_MORE further ones, behind the nseg firsts in
	// `first`: packed data -- deflate streams carried inside stored blocks -- puts headers that are none in front of the
	// block's that follows the stored run, in the same segment; that block's start went unseen, and the block was decoded
	// in a later round, as one piece)
	// (which ones: the NXZ_BLOCKFIND_MORE LAST passers, found by as many rounds of a maximum below the one before -- the first form kept
	// the first sixteen passers in the order they arrived in, so that with more than sixteen the pieces, the rounds and the timings of
	// nxz_inflate_stream differed from run to run, and a late true header could be lost to early false ones: advisor finding of round 5)
	uint64_t mine = 0;                                   // this lane's candidates that passed, by their index k / 64 (a segment leaves LEFT_MAX at most)
	for (uint32_t k = t, i = 0; k < n; k += 64, i++) {
		const uint32_t p = o[2 + k];
		if (header_ok(s, p, limit, 0xffffffffu)) { atomicMin(&best, p); if (i < 64) mine |= 1ull << i; }
	}
	__syncthreads();
	uint64_t *more = first + nseg + (size_t)seg * NXZ_BLOCKFIND_MORE;
	uint32_t below = 0xffffffffu;
	for (uint32_t j = 0; j < NXZ_BLOCKFIND_MORE; j++) {
		if (t == 0) npass = 0;                               // (npass: this round's maximum + 1, 0: none)
		__syncthreads();
		for (uint64_t m = mine; m; m &= m - 1) {
			const uint32_t p = o[2 + t + 64 * (uint32_t)__builtin_ctzll(m)];
			if (p != best && p < below) atomicMax(&npass, p + 1);
		}
		__syncthreads();
		const uint32_t got = npass;
		if (t == 0) more[j] = got ? base * 8 + (got - 1) : ~0ull;
		below = got ? got - 1 : 0;
		__syncthreads();
	}
	if (t == 0) first[seg] = best == 0xffffffffu ? ~0ull : base * 8 + best;
}

// A run of stored blocks is followed header by header (a thread per request; LEN tells where the next header
// is): from a byte position inside a stored block with `rem` bytes of it to come, to the header of the
// first block that is not a stored one -- or that is not whole inside the source, or whose LEN / NLEN do
// not match, or to the end of a final stored block (flag 1).  Stored data may look like anything, block
// headers included: a piece that stopped inside a stored block is continued behind the run in one go.
__global__ __launch_bounds__(64) void stored_walk_kernel(const nxz_walk_req_t *__restrict__ reqs, uint32_t n, nxz_walk_res_t *__restrict__ res)
{
	const uint32_t i = blockIdx.x * 64 + threadIdx.x;
	if (i >= n) return;
	const nxz_walk_req_t rq = reqs[i];
	const uint8_t *s = rq.src;
	const uint64_t total = rq.src_len * 8;
	uint64_t pos = rq.bit + (uint64_t)rq.rem * 8;
	uint32_t fin = rq.bfinal, flags = 0;
	for (uint32_t k = 0; k < (1u << 20); k++) {
		if (fin) { flags = 1; break; }
		if (pos + 3 > total) break;
		const uint32_t by = (uint32_t)(pos & 7);
		uint32_t h = s[pos >> 3] >> by;
		if (by > 5) h |= (uint32_t)s[(pos >> 3) + 1] << (8 - by);
		if (((h >> 1) & 3) != 0) break;                                  // not a stored block
		const uint64_t p2 = (pos + 3 + 7) & ~7ull;
		if (p2 + 32 > total) break;
		const uint8_t *q = s + (p2 >> 3);
		const uint32_t len = q[0] | (uint32_t)q[1] << 8, nlen = q[2] | (uint32_t)q[3] << 8;
		if ((len ^ nlen) != 0xffff) break;
		if (p2 + 32 + (uint64_t)len * 8 > total) break;                  // (the block reaches beyond the source: the piece suspends in it)
		fin = h & 1;
		pos = p2 + 32 + (uint64_t)len * 8;
	}
	nxz_walk_res_t out;
	out.bit = pos; out.flags = flags; out.reserved = 0;
	res[i] = out;
}

// items[i] = { src, dst, bytes }: dst <- src (any alignment; a workgroup per item)
struct CopyItem { const uint8_t *src; uint8_t *dst; uint64_t bytes; };

__global__ __launch_bounds__(NT) void copy_items_kernel(const CopyItem *__restrict__ items, uint32_t n)
{
	const uint32_t i = blockIdx.x;
	if (i >= n) return;
	const CopyItem it = items[i];
	const uint8_t NXZ_GLOBAL_AS *sp = (const uint8_t NXZ_GLOBAL_AS *)it.src;
	uint8_t NXZ_GLOBAL_AS *dp = (uint8_t NXZ_GLOBAL_AS *)it.dst;
	uint64_t nbytes = it.bytes;
	const int t = threadIdx.x;
	if (!nbytes) return;
	if ((uintptr_t)it.src < 16) {
		// fills: 0 zeros; 1, 2, 3 the three probe windows of the speculative decode (a window byte at
		// index k -- k counted from the item's first byte -- is k's low byte, its high byte, or 255 - low byte)
		const uint32_t mode = (uint32_t)(uintptr_t)it.src;
		for (uint64_t k = t; k < nbytes; k += NT)
			dp[k] = mode == 0 ? 0 : mode == 1 ? (uint8_t)k : mode == 2 ? (uint8_t)(k >> 8) : (uint8_t)(255 - (k & 255));
		return;
	}
	if ((((uintptr_t)it.src ^ (uintptr_t)it.dst) & 15) == 0 && nbytes >= 64) {
		// same alignment: bytes up to a 16-byte boundary, then 16 bytes per lane
		const uint32_t head = (uint32_t)((16 - ((uintptr_t)it.dst & 15)) & 15);
		if ((uint32_t)t < head) dp[t] = sp[t];
		typedef uint32_t v4u __attribute__((ext_vector_type(4)));
		const v4u NXZ_GLOBAL_AS *s4 = (const v4u NXZ_GLOBAL_AS *)(sp + head);
		v4u NXZ_GLOBAL_AS *d4 = (v4u NXZ_GLOBAL_AS *)(dp + head);
		const uint64_t nv = (nbytes - head) >> 4;
		for (uint64_t k = t; k < nv; k += NT) d4[k] = s4[k];
		const uint64_t done = head + (nv << 4);
		if (done + t < nbytes) dp[done + t] = sp[done + t];
	} else {
		for (uint64_t k = t; k < nbytes; k += NT) dp[k] = sp[k];
	}
}

// flags[i] = 1 when the two byte ranges of item i differ
__global__ __launch_bounds__(NT) void differ_items_kernel(const CopyItem *__restrict__ items, uint32_t n, uint32_t *__restrict__ flags)
{
	const uint32_t i = blockIdx.x;
	if (i >= n) return;
	const CopyItem it = items[i];
	const uint8_t NXZ_GLOBAL_AS *a = (const uint8_t NXZ_GLOBAL_AS *)it.src;
	const uint8_t NXZ_GLOBAL_AS *b = (const uint8_t NXZ_GLOBAL_AS *)it.dst;
	bool d = false;
	for (uint64_t k = threadIdx.x; k < it.bytes; k += NT) d |= a[k] != b[k];
	if (__syncthreads_or(d) && threadIdx.x == 0) flags[i] = 1;
}

// ---- resolving what the pieces copied out of their unknown histories ----
// A piece is decoded with 16-bit elements (nxz_inflate.hip, W16): an element is the byte itself or
// 0x8000 | the index of the byte of the 32 KiB in front of the piece that it is a copy of.
struct Piece { const uint16_t *o; uint64_t len, place; };

// The window behind a piece as a function of the window in front of it -- the piece's tail map: entry k is the
// byte itself (0..255) or 0x8000 | index into the window in front.  It is the last 32 KiB of the piece's
// elements; in front of a shorter piece, the old window moved up.  Nobody stores it: the kernels below read
// eight entries at a time straight from the piece (round 2 wrote all maps out first: 64 KiB a piece of device
// memory, and a third of the chain's time).
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v4u_any __attribute__((ext_vector_type(4), aligned(2)));
__device__ __forceinline__ v4u tail_entries(const Piece &p, uint32_t kq)            // entries kq .. kq + 7 (kq a multiple of 8)
{
	const uint32_t L = p.len >= 32768 ? 32768u : (uint32_t)p.len, keep = 32768 - L;
	const uint16_t NXZ_GLOBAL_AS *o = (const uint16_t NXZ_GLOBAL_AS *)p.o + (p.len - L);         // the last L elements
	if (kq >= keep) return *(const NXZ_GLOBAL_AS v4u_any *)(o + (kq - keep));
	uint32_t e[8];
#pragma unroll
	for (uint32_t j = 0; j < 8; j++) e[j] = kq + j < keep ? 0x8000u | (kq + j + L) : (uint32_t)o[kq + j - keep];
	return (v4u){ e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16 };
}

// A workgroup walks `per` pieces in order (workgroup g: pieces g * per ...) and writes, for each, the
// 32 KiB window BEHIND it (= the history of the next one): windows[i * 32768 ..].  The window in front
// of workgroup g's first piece is win0 for g = 0, else front[(g - 1) * 32768 ..].  A thread owns 32
// consecutive entries; the next piece's map is on its way while this one is applied.
// Used three ways (nxz_launch_window_chain): one workgroup over all pieces when they are few; else
// over the composed maps of the groups (windows behind the groups; PIECES false: the maps are an array),
// then a workgroup per group.
template <bool PIECES>
__global__ __launch_bounds__(1024) void window_chain_kernel(const uint16_t *__restrict__ maps, const Piece *__restrict__ pieces, uint32_t n, uint32_t per,
							     const uint8_t *__restrict__ win0, const uint8_t *__restrict__ front, uint8_t *__restrict__ windows)
{
	__shared__ __attribute__((aligned(16))) uint8_t w[2][32768];
	const int t = threadIdx.x;
	const uint32_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
	if (lo >= hi) return;
	const uint8_t *wf = blockIdx.x ? front + (size_t)(blockIdx.x - 1) * 32768 : win0;
	for (uint32_t k = t; k < 32768 / 16; k += 1024) ((uint4 *)w[0])[k] = ((const uint4 *)wf)[k];
	__syncthreads();
	uint32_t cur = 0;
	const v4u NXZ_GLOBAL_AS *mp = (const v4u NXZ_GLOBAL_AS *)maps;
	auto fetch = [&](uint32_t i, int q) __attribute__((always_inline)) -> v4u {
		if (PIECES) return tail_entries(pieces[i], 32 * t + 8 * q);
		return mp[(size_t)i * 4096 + t * 4 + q];
	};
	v4u nx[4];
#pragma unroll
	for (int q = 0; q < 4; q++) nx[q] = fetch(lo, q);
	for (uint32_t i = lo; i < hi; i++) {
		v4u m[4];
#pragma unroll
		for (int q = 0; q < 4; q++) m[q] = nx[q];
		if (i + 1 < hi) {
#pragma unroll
			for (int q = 0; q < 4; q++) nx[q] = fetch(i + 1, q);
		}
		const uint8_t *wi = w[cur];
		uint32_t out[8];
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const uint32_t d[4] = { m[q].x, m[q].y, m[q].z, m[q].w };
#pragma unroll
			for (int e = 0; e < 4; e++) {
				const uint32_t lo16 = d[e] & 0xffff, hi16 = d[e] >> 16;
				const uint32_t b0 = (lo16 & 0x8000) ? wi[lo16 & 0x7fff] : lo16, b1 = (hi16 & 0x8000) ? wi[hi16 & 0x7fff] : hi16;
				const uint32_t idx = q * 8 + e * 2;              // entry pair within my 32
				out[idx >> 2] = (idx & 2) ? out[idx >> 2] | b0 << 16 | b1 << 24 : b0 | b1 << 8;
			}
		}
		uint4 *wo = (uint4 *)(w[cur ^ 1] + 32 * t);
		wo[0] = make_uint4(out[0], out[1], out[2], out[3]);
		wo[1] = make_uint4(out[4], out[5], out[6], out[7]);
		v4u NXZ_GLOBAL_AS *g = (v4u NXZ_GLOBAL_AS *)(windows + (size_t)i * 32768 + 32 * t);
		g[0] = (v4u){ out[0], out[1], out[2], out[3] };
		g[1] = (v4u){ out[4], out[5], out[6], out[7] };
		__syncthreads();
		cur ^= 1;
	}
}

// The maps of `per` consecutive pieces composed into one: what the window behind the group is, as a
// function of the window in front of it (entries as in a tail map).  A workgroup per group.
// (PIECES false: the maps are an array -- groups of groups, for the third level of a long chain)
template <bool PIECES>
__global__ __launch_bounds__(1024) void compose_maps_kernel(const Piece *__restrict__ pieces, const uint16_t *__restrict__ maps, uint32_t n, uint32_t per,
							     uint16_t *__restrict__ gmaps)
{
	__shared__ __attribute__((aligned(16))) uint16_t c[2][32768];
	const int t = threadIdx.x;
	const v4u NXZ_GLOBAL_AS *mp = (const v4u NXZ_GLOBAL_AS *)maps;
	auto fetch = [&](uint32_t i, int q) __attribute__((always_inline)) -> v4u {
		if (PIECES) return tail_entries(pieces[i], 32 * t + 8 * q);
		return mp[(size_t)i * 4096 + t * 4 + q];
	};
	const uint32_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
	if (lo >= hi) return;
	for (uint32_t k = t; k < 32768; k += 1024) c[0][k] = (uint16_t)(0x8000u | k);
	__syncthreads();
	uint32_t cur = 0;
	v4u nx[4];
#pragma unroll
	for (int q = 0; q < 4; q++) nx[q] = fetch(lo, q);
	for (uint32_t i = lo; i < hi; i++) {
		const uint16_t *ci = c[cur];
		v4u mq[4];
#pragma unroll
		for (int q = 0; q < 4; q++) mq[q] = nx[q];
		if (i + 1 < hi) {
#pragma unroll
			for (int q = 0; q < 4; q++) nx[q] = fetch(i + 1, q);
		}
		uint32_t out[16];
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const v4u m = mq[q];
			const uint32_t d[4] = { m.x, m.y, m.z, m.w };
#pragma unroll
			for (int e = 0; e < 4; e++) {
				const uint32_t lo16 = d[e] & 0xffff, hi16 = d[e] >> 16;
				const uint32_t v0 = (lo16 & 0x8000) ? ci[lo16 & 0x7fff] : lo16, v1 = (hi16 & 0x8000) ? ci[hi16 & 0x7fff] : hi16;
				out[q * 4 + e] = v0 | v1 << 16;
			}
		}
		uint4 *co = (uint4 *)(c[cur ^ 1] + 32 * t);
#pragma unroll
		for (int q = 0; q < 4; q++) co[q] = make_uint4(out[q * 4], out[q * 4 + 1], out[q * 4 + 2], out[q * 4 + 3]);
		__syncthreads();
		cur ^= 1;
	}
	v4u NXZ_GLOBAL_AS *g = (v4u NXZ_GLOBAL_AS *)(gmaps + (size_t)blockIdx.x * 32768);
	const uint4 *cf = (const uint4 *)(c[cur] + 32 * t);
#pragma unroll
	for (int q = 0; q < 4; q++) { const uint4 v = cf[q]; g[t * 4 + q] = (v4u){ v.x, v.y, v.z, v.w }; }
}

// every piece to its place: final[place + o] = the byte, or the byte of its history it is a copy of.
// A piece is taken in chunks of 32 Ki elements by up to blocks_per_piece workgroups (chunk c by workgroup
// c mod blocks_per_piece; a workgroup without a chunk ends at once): each loads the piece's window once, so a
// piece of the usual 10-40 KiB costs one or two window loads (a fixed four parts per piece cost four, as much
// traffic as the elements themselves), and the odd piece of half a megabyte still has four workgroups.
__global__ __launch_bounds__(256) void resolve_kernel(const Piece *__restrict__ pieces, uint32_t n, const uint8_t *__restrict__ win0,
						       const uint8_t *__restrict__ windows, uint8_t *__restrict__ dst, uint32_t blocks_per_piece)
{
	__shared__ __attribute__((aligned(16))) uint8_t w[32768];
	const uint32_t i = blockIdx.x / blocks_per_piece, part = blockIdx.x % blocks_per_piece;
	if (i >= n) return;
	const Piece p = pieces[i];
	constexpr uint64_t CH = 32768;
	const uint64_t nch = (p.len + CH - 1) / CH;
	if (part >= nch) return;
	const uint8_t *win = i ? windows + (size_t)(i - 1) * 32768 : win0;
	for (uint32_t k = threadIdx.x; k < 32768 / 16; k += 256) ((uint4 *)w)[k] = ((const uint4 *)win)[k];
	__syncthreads();
	uint8_t *out = dst + p.place;
	for (uint64_t c = part; c < nch; c += blocks_per_piece) {
		const uint64_t lo = c * CH, hi = lo + CH < p.len ? lo + CH : p.len;
		// 4 elements per thread and trip where the piece's place allows whole dwords
		const uint64_t head = ((4 - ((uintptr_t)(out + lo) & 3)) & 3);
		const uint64_t a0 = lo + head < hi ? lo + head : hi;
		for (uint64_t o = lo + threadIdx.x; o < a0; o += 256) { const uint32_t v = p.o[o]; out[o] = (uint8_t)((v & 0x8000) ? w[v & 0x7fff] : v); }
		// (eight elements a lane and trip -- one 16-byte load at any alignment, one 8-byte store on a dword boundary --,
		// two trips in flight: with four elements and one trip a workgroup spent its time waiting for its loads)
		typedef uint32_t v2u_st __attribute__((ext_vector_type(2), aligned(4)));
		const uint64_t no = (hi - a0) >> 3;
		const uint16_t NXZ_GLOBAL_AS *eo = (const uint16_t NXZ_GLOBAL_AS *)p.o;
		auto eight = [&](uint64_t o, const v4u e4) __attribute__((always_inline)) {
			const uint32_t d[4] = { e4.x, e4.y, e4.z, e4.w };
			uint32_t r[2] = { 0, 0 };
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t v0 = d[k] & 0xffff, v1 = d[k] >> 16;
				const uint32_t b0 = (v0 & 0x8000) ? (uint32_t)w[v0 & 0x7fff] : (v0 & 0xff), b1 = (v1 & 0x8000) ? (uint32_t)w[v1 & 0x7fff] : (v1 & 0xff);
				r[k >> 1] |= (b0 | b1 << 8) << (16 * (k & 1));
			}
			*(NXZ_GLOBAL_AS v2u_st *)(out + o) = (v2u_st){ r[0], r[1] };
		};
		uint64_t q = threadIdx.x;
		for (; q + 256 < no; q += 512) {
			const uint64_t o0 = a0 + q * 8, o1 = o0 + 256 * 8;
			const v4u ea = *(const NXZ_GLOBAL_AS v4u_any *)(eo + o0), ebb = *(const NXZ_GLOBAL_AS v4u_any *)(eo + o1);
			eight(o0, ea);
			eight(o1, ebb);
		}
		if (q < no) { const uint64_t o0 = a0 + q * 8; eight(o0, *(const NXZ_GLOBAL_AS v4u_any *)(eo + o0)); }
		const uint64_t nq = no * 2;                               // (what follows counts in fours, as before)
		for (uint64_t o = a0 + nq * 4 + threadIdx.x; o < hi; o += 256) { const uint32_t v = p.o[o]; out[o] = (uint8_t)((v & 0x8000) ? w[v & 0x7fff] : v); }
	}
}

} // namespace nxzb

extern "C" uint32_t nxz_blockfind_segment(uint64_t srclen)
{
	static const uint32_t env = getenv("NXZ_BLOCKFIND_SEG") ? (uint32_t)atoi(getenv("NXZ_BLOCKFIND_SEG")) : 0;
	if (env >= 512 && env <= nxzb::SEG && !(env & (env - 1))) return env;
	// (a workgroup per segment; what a workgroup takes longest over is the real header in its segment, if there is one,
	// so short segments buy little: 3.7 MiB of stream: 0.36 ms at 1 KiB, 0.29 at 2 KiB, 0.24 at 4 KiB; 13 MiB: 0.48 at
	// 4 KiB, 0.39 at 8 KiB; 53 MiB: 1.56 at 4 KiB, 1.10 at 8 KiB.  Only a part of a stream of a MiB or so is cut finer)
	return srclen <= (1u << 20) ? 1024 : srclen <= (2u << 20) ? 2048 : srclen <= (8u << 20) ? 4096 : nxzb::SEG;
}

// scratch: nxz_blockfind_scratch(nseg) bytes of DEVICE memory, or NULL (the search as one kernel)
extern "C" size_t nxz_blockfind_scratch(uint32_t nseg) { return (size_t)nseg * (nxzb::LEFT_MAX + 2) * sizeof(uint16_t); }
extern "C" int nxz_launch_find_blocks(const uint8_t *src, uint64_t srclen, uint64_t first_bit, uint64_t *first, uint32_t nseg, void *scratch, hipStream_t stream)
{
	if (!nseg) return 0;
	static const uint32_t diag = getenv("NXZ_BLOCKFIND_DIAG") ? (uint32_t)atoi(getenv("NXZ_BLOCKFIND_DIAG")) : 0;
	static const bool two = !(getenv("NXZ_BLOCKFIND_TWO") && atoi(getenv("NXZ_BLOCKFIND_TWO")) == 0);
	const uint32_t seg = nxz_blockfind_segment(srclen);
	uint16_t *left = two && !diag ? (uint16_t *)scratch : nullptr;
	hipLaunchKernelGGL(nxzb::find_blocks_kernel, dim3(nseg), dim3(nxzb::NT), 0, stream, src, srclen, first_bit, first, nseg, seg, diag, left);
	if (left) hipLaunchKernelGGL(nxzb::check_headers_kernel, dim3(nseg), dim3(64), 0, stream, src, srclen, first, nseg, seg, (const uint16_t *)left);
	else if (hipMemsetAsync(first + nseg, 0xff, (size_t)nseg * NXZ_BLOCKFIND_MORE * sizeof(uint64_t), stream) != hipSuccess) return -1;
	return (int)hipGetLastError();
}

extern "C" int nxz_launch_stored_walk(const nxz_walk_req_t *reqs, uint32_t n, nxz_walk_res_t *res, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxzb::stored_walk_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, reqs, n, res);
	return (int)hipGetLastError();
}

extern "C" int nxz_launch_copy_items(const void *items, uint32_t n, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxzb::copy_items_kernel, dim3(n), dim3(nxzb::NT), 0, stream, (const nxzb::CopyItem *)items, n);
	return (int)hipGetLastError();
}

// group maps / group windows: room for n / nxz_window_chain_group(0) + 1 maps (64 KiB each) and windows (32 KiB each).
// Pieces per group (n = 0: the smallest group there is): composing a group's maps takes ~7 us a piece, a walk over
// piece or group maps in memory 1.7 us a step.
extern "C" uint32_t nxz_window_chain_group(uint32_t n)
{
	static const uint32_t env = getenv("NXZ_CHAIN_GROUP") ? (uint32_t)atoi(getenv("NXZ_CHAIN_GROUP")) : 0;
	if (n && (env == 8 || env == 16 || env == 32 || env == 64)) return env;
	// (with the third level the walk over the groups no longer grows with their number, and short groups mean short
	// composing: 8 pieces a group at every length -- 749 pieces: 1.30 -> 1.21 ms, 1405: 2.02 -> 1.94, 4926 and 6821: the same
	// as with 16 or 32; up to the third level it was 8 / 16 / 32 by length, least at 0.48 sqrt(n))
	return 8;
}
extern "C" int nxz_launch_window_chain(const void *pieces, uint32_t n, const uint8_t *win0, uint8_t *windows,
				       uint16_t *gmaps, uint8_t *gwin, hipStream_t stream)
{
	if (!n) return 0;
	const uint32_t per = nxz_window_chain_group(n), ng = (n + per - 1) / per;
	const nxzb::Piece *pc = (const nxzb::Piece *)pieces;
	if (ng <= 2 || !gmaps || !gwin) {
		hipLaunchKernelGGL(nxzb::window_chain_kernel<true>, dim3(1), dim3(1024), 0, stream, (const uint16_t *)nullptr, pc, n, n, win0, (const uint8_t *)nullptr, windows);
		return (int)hipGetLastError();
	}
	// the groups' composed maps (all at once), the windows behind the groups, then every group's pieces from the
	// window in front of the group (all groups at once).  The windows behind the groups: one walk over the groups
	// when they are few (1.7 us a step); else the same again one level up -- groups of 16 groups composed, one
	// walk over those, then every group of groups side by side (214 groups: 0.37 ms -> under 0.1)
	hipLaunchKernelGGL(nxzb::compose_maps_kernel<true>, dim3(ng), dim3(1024), 0, stream, pc, (const uint16_t *)nullptr, n, per, gmaps);
	static const uint32_t two_max = getenv("NXZ_CHAIN_TWO_LEVELS_MAX") ? (uint32_t)atoi(getenv("NXZ_CHAIN_TWO_LEVELS_MAX")) : 64;
	if (ng <= two_max)
		hipLaunchKernelGGL(nxzb::window_chain_kernel<false>, dim3(1), dim3(1024), 0, stream, (const uint16_t *)gmaps, pc, ng, ng, win0, (const uint8_t *)nullptr, gwin);
	else {
		const uint32_t per2 = 16, ng2 = (ng + per2 - 1) / per2;
		uint16_t *gmaps2 = gmaps + (size_t)ng * 32768;                  // (room: see nxz_pinflate.cpp, n / 8 + 1 maps and windows)
		uint8_t *gwin2 = gwin + (size_t)ng * 32768;
		hipLaunchKernelGGL(nxzb::compose_maps_kernel<false>, dim3(ng2), dim3(1024), 0, stream, pc, (const uint16_t *)gmaps, ng, per2, gmaps2);
		hipLaunchKernelGGL(nxzb::window_chain_kernel<false>, dim3(1), dim3(1024), 0, stream, (const uint16_t *)gmaps2, pc, ng2, ng2, win0, (const uint8_t *)nullptr, gwin2);
		hipLaunchKernelGGL(nxzb::window_chain_kernel<false>, dim3(ng2), dim3(1024), 0, stream, (const uint16_t *)gmaps, pc, ng, per2, win0, (const uint8_t *)gwin2, gwin);
	}
	hipLaunchKernelGGL(nxzb::window_chain_kernel<true>, dim3(ng), dim3(1024), 0, stream, (const uint16_t *)nullptr, pc, n, per, win0, (const uint8_t *)gwin, windows);
	return (int)hipGetLastError();
}

extern "C" int nxz_launch_resolve(const void *pieces, uint32_t n, const uint8_t *win0, const uint8_t *windows, uint8_t *dst, hipStream_t stream)
{
	if (!n) return 0;
	const uint32_t bpp = 4;
	hipLaunchKernelGGL(nxzb::resolve_kernel, dim3(n * bpp), dim3(256), 0, stream, (const nxzb::Piece *)pieces, n, win0, windows, dst, bpp);
	return (int)hipGetLastError();
}

extern "C" int nxz_launch_differ_items(const void *items, uint32_t n, uint32_t *flags, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxzb::differ_items_kernel, dim3(n), dim3(nxzb::NT), 0, stream, (const nxzb::CopyItem *)items, n, flags);
	return (int)hipGetLastError();
}
