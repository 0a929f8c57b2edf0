// nxz_inflate.hip -- DEFLATE decompression engine for MI355X (gfx950, wave64).
//
// Replaces the POWER NX accelerator's GZIP_FC_DECOMPRESS / _RESUME function
// codes (issued at /root/reference lib/nx_inflate.c:909-912; resume inputs
// inc_nx/nxu.h:296-393, outputs :403-541, consumer lib/nx_inflate.c:1308-1609).
// Same decisions as the CPU restatement oracle/nxz_inflate.c.
//
// One wavefront per stream (workgroup = 64 lanes, 39.5 KiB LDS -> 4 streams per
// CU, one per SIMD, 1024 per chip): a deflate stream is serial by construction, so the
// symbol loop is wave-uniform and the lanes are used where there is width:
//   - coalesced 16 B/lane staging of the compressed input into LDS
//   - decode-table construction (lane per symbol)
//   - match copies (lane per byte, pattern-replicated for dist < len)
//   - 16 KiB coalesced flushes of the LDS output window + CRC-32/Adler-32
//     (lane per 256-byte slice, GF(2) tree combine)
// The 32 KiB circular output window in LDS is also the history, so match
// sources never touch HBM.
//
// Inside a coded block the common case runs through a multi-token step: the next 96 bits of the
// source are kept wave-uniform (taken with v_readlane from two 256-byte blocks of the source that
// the lanes hold in registers, so the bit reader needs no LDS), every lane looks up BOTH decode
// tables at its own bit offset (one LDS round trip for all 64 offsets), and a uniform walk then
// follows the chain of real token starts through those answers with v_readlane: 5-7 tokens per
// round trip instead of one token per 2-6.  Anything unusual (codes longer than the fast tables,
// end of block, errors, the last bytes of the source, a full target) is left to the one-token
// path below it, which keeps the suspend/error semantics exactly as before.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include "nxz_device.h"
#include "nxz_inflate_tables.h"

#ifndef NXZ_INFLATE_CHAIN_BY_LANE
#define NXZ_INFLATE_CHAIN_BY_LANE 1       /* 0: the chain of token starts link by link in scalar arithmetic, as up to round 5 (for comparisons) */
#endif
#ifndef NXZ_SYNC_EXTRA
#define NXZ_SYNC_EXTRA 2048               /* token_sync_kernel: bits behind the stretch every lane walks alone in which the lanes may still fall in step */
#endif
#ifndef NXZ_INFLATE_LIT_STEP
#define NXZ_INFLATE_LIT_STEP 1            /* 0: no short step for stretches of literals (for comparisons) */
#endif
namespace nxzi {

// Diagnostic only (tools/bench_inflate_kinds.py): cycle sums of lane 0 of every stream.
__device__ unsigned long long *prof_buf = nullptr;
#ifdef NXZ_INFLATE_PROF        /* build with -DNXZ_INFLATE_PROF to use tools/bench_inflate_kinds.py's cycle sums */
#define IPROF(idx) do { if (prof) { unsigned long long now_ = clock64(); pacc[idx] += now_ - tprev; tprev = now_; } } while (0)
#define ICOUNT(idx, v) do { if (prof) pacc[idx] += (unsigned long long)(v); } while (0)
#else
#define IPROF(idx) do { } while (0)
#define ICOUNT(idx, v) do { } while (0)
#endif

// a value that is the same in all lanes but sits in a vector register: tell the compiler
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

constexpr uint32_t WIN = 32768, WMASK = WIN - 1;
constexpr uint32_t FLUSH = 16384;
constexpr uint32_t STAGE = 512;              // staged compressed bytes (one-token path and headers only)
// (LBITS, DBITS, Huff, HuffD: nxz_inflate_tables.h)

// GW: the window is the target itself (and the history in front of the source) in global memory,
// for batches: 7.6 KiB of LDS per stream instead of 39.5, so 16 streams per CU instead of 4.
// (W16 without GW: a window of 16-bit elements in LDS, 64 KiB -- two streams per CU, for the few pieces of a
// short stretch of a stream, where one wavefront's latency is all that counts)
template <bool GW, bool W16 = false>
struct SmemT {
	typedef typename std::conditional<W16, uint16_t, uint8_t>::type elem_t;
	elem_t win[GW ? 16 : WIN];
	uint32_t stage[STAGE / 4 + 4];
	Huff hl;
	HuffD hd;
	uint8_t lens[320];
	uint8_t cl[32];
};

static_assert(sizeof(Huff) % 16 == 0 && sizeof(HuffD) % 16 == 0 && sizeof(Built) % 16 == 0, "tables are copied 16 bytes a lane");

template <typename Smem>
__device__ __forceinline__ void load_built(Smem &sm, const Built *bt, int lane)
{
	const uint4 *src = (const uint4 *)bt;
	uint4 *dl = (uint4 *)&sm.hl, *dd = (uint4 *)&sm.hd;
	for (uint32_t i = lane; i < sizeof(Huff) / 16; i += 64) dl[i] = src[i];
	for (uint32_t i = lane; i < sizeof(HuffD) / 16; i += 64) dd[i] = src[sizeof(Huff) / 16 + i];
	__syncthreads();
}

__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b)
{
	uint32_t r = 0;
#pragma unroll 8
	for (int i = 0; i < 32; i++) {
		r ^= (b & 0x80000000u) ? a : 0;
		a = (a >> 1) ^ ((a & 1) ? 0xedb88320u : 0);
		b <<= 1;
	}
	return r;
}

__device__ __forceinline__ uint32_t gf_xpow8(uint32_t n)   // x^(8n)
{
	uint32_t r = 0x80000000u, sq = 0x00800000u;
	while (n) { if (n & 1) r = gf_mul(r, sq); sq = gf_mul(sq, sq); n >>= 1; }
	return r;
}

// Build the decode tables for `n` symbols with code lengths len[] (wave cooperative).
// All of it by the whole wave: symbol i = row * 64 + lane; counts per length, first codes and the rank
// of a symbol among those of its length come out of ballots (RFC 1951 3.2.2), every symbol then writes
// its slots of the fast table and its place in the (length, symbol) order of the slow path.
template <int FB, typename H>
__device__ __forceinline__ void build(H &h, const uint8_t *len, int n, int lane)
{
	constexpr int ROWS = 5;                                          // n <= 288 + 32
	const uint64_t below = (1ull << lane) - 1;
	uint32_t l[ROWS], code[ROWS], place[ROWS];
#pragma unroll
	for (int r = 0; r < ROWS; r++) { const int i = r * 64 + lane; l[r] = i < n ? len[i] : 0; code[r] = 0; place[r] = 0; }
	for (int i = lane; i < (1 << FB); i += 64) h.fast[i] = 0;
	uint32_t c = 0, prevcnt = 0, offs = 0;
	for (uint32_t b = 1; b <= 15; b++) {
		c = (c + prevcnt) << 1;
		uint32_t run = 0;
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			const uint64_t m = __ballot(l[r] == b);
			const uint32_t k = run + (uint32_t)__popcll(m & below);
			if (l[r] == b) { code[r] = c + k; place[r] = offs + k; }
			run += (uint32_t)__popcll(m);
		}
		if (lane == 0) h.count[b] = (uint16_t)run;
		prevcnt = run; offs += run;
	}
	if (lane == 0) h.count[0] = 0;
	__syncthreads();
#pragma unroll
	for (int r = 0; r < ROWS; r++) {
		const uint32_t i = r * 64 + lane;
		if (!l[r]) continue;
		h.sym[place[r]] = (uint16_t)i;
		if (l[r] > (uint32_t)FB) continue;
		const uint32_t rev = __builtin_bitreverse32(code[r]) >> (32 - l[r]);
		for (uint32_t idx = rev; idx < (1u << FB); idx += (1u << l[r])) h.fast[idx] = (uint16_t)(i | (l[r] << 12));
	}
	__syncthreads();
}

struct Bits {
	const NXZ_GLOBAL_AS uint8_t *src;   // device memory, used through address space 1 (a generic access makes the compiler drain the LDS queue at every later wait)
	uint32_t srclen;
	uint64_t total_bits;      // 8*srclen
	uint64_t pos;             // next unread bit
	uint32_t stage_base;      // byte offset of stage[0] in src (multiple of 16), 0xffffffff = none
	uint32_t *stage;
	int lane;

	__device__ __forceinline__ void restage(uint32_t byte)
	{
		// all lanes: load STAGE bytes starting at byte & ~15
		uint32_t base = byte & ~15u;
		__syncthreads();
		for (uint32_t i = lane; i < STAGE / 16; i += 64) {
			uint32_t off = base + i * 16;
			uint4 v = make_uint4(0, 0, 0, 0);
			if (off + 16 <= srclen) v = *(const uint4 *)(src + off);
			else if (off < srclen) {
				uint32_t w[4] = {0, 0, 0, 0};
				for (uint32_t k = 0; off + k < srclen; k++) w[k >> 2] |= (uint32_t)src[off + k] << (8 * (k & 3));
				v = make_uint4(w[0], w[1], w[2], w[3]);
			}
			((uint4 *)stage)[i] = v;
		}
		stage_base = base;
		__syncthreads();
	}
	// make sure [byte, byte+span) is staged (wave-uniform call)
	__device__ __forceinline__ void ensure(uint32_t byte, uint32_t span)
	{
		if (stage_base == 0xffffffffu || byte < stage_base || byte + span > stage_base + STAGE) restage(byte);
	}
	// peek up to 32 bits at the current position (bits past the end read as 0)
	__device__ __forceinline__ uint32_t peek()
	{
		ensure((uint32_t)(pos >> 3), 8);
		return raw_peek();
	}
	__device__ __forceinline__ uint32_t raw_peek() const
	{
		uint32_t byte = (uint32_t)(pos >> 3);
		uint32_t o = byte - stage_base;
		uint32_t a = stage[o >> 2], b = stage[(o >> 2) + 1], c = stage[(o >> 2) + 2];
		uint32_t lo = __builtin_amdgcn_alignbyte(b, a, o & 3);
		uint32_t hi = __builtin_amdgcn_alignbyte(c, b, o & 3);
		uint32_t sh = (uint32_t)pos & 7;
		return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
	}
	__device__ __forceinline__ bool have(uint32_t n) const { return pos + n <= total_bits; }

	// ---- register bit buffer for the symbol loop: `bb` holds bits [pos, pos+bc) ----
	uint64_t bb = 0; uint32_t bc = 0;
	__device__ __forceinline__ void bb_sync() { bb = 0; bc = 0; }            // after pos was changed by hand
	__device__ __forceinline__ void bb_fill()                                 // make bc >= 32
	{
		if (bc >= 32) return;
		uint64_t p2 = pos + bc;                                               // first bit not in bb
		uint32_t byte = (uint32_t)(p2 >> 3);
		ensure(byte, 12);
		uint32_t o = byte - stage_base;
		uint32_t a = stage[o >> 2], b = stage[(o >> 2) + 1];
		uint32_t v = __builtin_amdgcn_alignbyte(b, a, o & 3);
		uint32_t sh = (uint32_t)p2 & 7;                                       // bits of that byte already consumed/held
		// take the (32 - sh) fresh bits of v
		bb |= (uint64_t)(v >> sh) << bc;
		bc += 32 - sh;
	}
	__device__ __forceinline__ void bb_drop(uint32_t n) { bb >>= n; bc -= n; pos += n; }
};

// RFC1951 3.2.5 length / distance code parameters, computed
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

template <int FB, typename H>
__device__ __forceinline__ int decode_sym(const H &h, uint32_t bits, uint32_t &nbits)
{
	uint32_t e = h.fast[bits & ((1u << FB) - 1)];
	if (e) { nbits = e >> 12; return (int)(e & 0xfff); }
	// slow canonical walk (codes longer than FB bits)
	int code = 0, first = 0, index = 0;
	for (int len = 1; len <= 15; len++) {
		code |= (int)(bits & 1); bits >>= 1;
		int count = h.count[len];
		if (code - count < first) { nbits = len; return h.sym[index + (code - first)]; }
		index += count; first += count; first <<= 1; code <<= 1;
	}
	nbits = 16;
	return -2;
}

// Parse a dynamic block header at b.pos (after the 3 header bits).  Returns
// 0 ok (lens filled, b.pos advanced, *tbits = table bits), 1 out of source, <0 invalid.
// The header (at most 2283 bits) is taken into two registers per lane (lane k: dwords k and 64 + k
// from the dword the header starts in; a copy in sm.stage for the one step where lanes read at
// different places), so the serial part -- one code-length symbol after the other -- reads its bits
// with v_readlane and looks the 7-bit code-length code up in two more registers: no LDS round trip
// per symbol.  The code-length code's canonical codes come from ballots (lane = symbol).
template <typename Smem>
__device__ __forceinline__ int read_dht(Bits &b, Smem &sm, int &hlit, int &hdist, uint32_t &tbits)
{
	const int lane = b.lane;
	const uint64_t start = b.pos;
	if (!b.have(14)) return 1;
	const uint32_t d0 = (uint32_t)(start >> 5);                       // first dword of the header
	auto dword = [&](uint32_t idx) __attribute__((always_inline)) -> uint32_t {
		const uint64_t byte = (uint64_t)idx * 4;
		uint32_t w = 0;
		if (byte + 4 <= b.srclen && ((uintptr_t)b.src & 3) == 0) w = ((const NXZ_GLOBAL_AS uint32_t *)b.src)[idx];
		else for (uint32_t k = 0; k < 4 && byte + k < b.srclen; k++) w |= (uint32_t)b.src[byte + k] << (8 * k);
		return w;
	};
	const uint32_t R0 = dword(d0 + lane), R1 = dword(d0 + 64 + lane);
	__syncthreads();
	sm.stage[lane] = R0; sm.stage[64 + lane] = R1;
	b.stage_base = 0xffffffffu;                                       // (the stage no longer holds what Bits put there)
	__syncthreads();
	// up to 25 bits at bit p of the source (p >= start), wave-uniform
	auto peek = [&](uint64_t p) __attribute__((always_inline)) -> uint32_t {
		const uint32_t o = uni((uint32_t)(p - (uint64_t)d0 * 32)), i = o >> 5, sh = o & 31;
		const uint32_t lo = i < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)R0, (int)i) : (uint32_t)__builtin_amdgcn_readlane((int)R1, (int)(i - 64));
		const uint32_t j = i + 1;
		const uint32_t hi = j < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)R0, (int)j) : (uint32_t)__builtin_amdgcn_readlane((int)R1, (int)(j & 63));
		return (uint32_t)(((((uint64_t)hi << 32) | lo) >> sh));
	};
	uint32_t v = peek(start);
	hlit = (int)(v & 31) + 257; hdist = (int)((v >> 5) & 31) + 1;
	const int hclen = (int)((v >> 10) & 15) + 4;
	uint64_t pos = start + 14;
	if (hlit > 286 || hdist > 30) return -1;
	if (pos + 3 * (uint32_t)hclen > b.total_bits) return 1;
	// the code-length code: lane i < hclen reads the i-th 3-bit length, which belongs to symbol order[i]
	uint32_t myl = 0;
	{
		const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
		if (lane < 32) sm.cl[lane] = 0;
		__syncthreads();
		if (lane < hclen) {
			const uint32_t o = (uint32_t)(pos - (uint64_t)d0 * 32) + 3 * (uint32_t)lane;
			const uint64_t w = (uint64_t)sm.stage[o >> 5] | ((uint64_t)sm.stage[(o >> 5) + 1] << 32);
			sm.cl[order[lane]] = (uint8_t)((w >> (o & 31)) & 7);
		}
		__syncthreads();
		myl = lane < 19 ? sm.cl[lane] : 0;
	}
	pos += 3 * (uint32_t)hclen;
	// canonical codes by ranks (lane = symbol), then the look-up: entries `lane` and `lane + 64` of the
	// 7-bit table, symbol | length << 5, 0xff = no code
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
		if (kraft > 128) return -2;
		const uint32_t myrev = myl ? __builtin_bitreverse32(mycode) >> (32 - myl) : 0;
		for (int sy = 0; sy < 19; sy++) {
			const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)myl, sy);
			if (!l) continue;
			const uint32_t rev = (uint32_t)__builtin_amdgcn_readlane((int)myrev, sy), mask = (1u << l) - 1;
			if (((uint32_t)lane & mask) == rev) tlo = (uint32_t)sy | (l << 5);
			if ((((uint32_t)lane + 64) & mask) == rev) thi = (uint32_t)sy | (l << 5);
		}
	}
	int n = 0, prev = 0;
	const int total = hlit + hdist;
	// (the bits at pos in a scalar window, refilled from the lanes' registers every few symbols: a symbol takes 14
	// bits at most -- two trips through v_readlane per symbol would otherwise be the better part of the loop)
	uint64_t win = 0;
	uint32_t wbits = 0;
	auto word = [&](uint32_t i) __attribute__((always_inline)) -> uint32_t {
		const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)R0, (int)(i & 63)), hi = (uint32_t)__builtin_amdgcn_readlane((int)R1, (int)(i & 63));
		return i < 64 ? lo : hi;
	};
	while (n < total) {
		if (pos + 1 > b.total_bits) return 1;
		if (wbits < 14) {
			const uint32_t o = uni((uint32_t)(pos - (uint64_t)d0 * 32)), i = o >> 5, sh = o & 31;
			win = (((uint64_t)word(i + 1) << 32) | word(i)) >> sh;
			wbits = 64 - sh;
		}
		const uint32_t bits = (uint32_t)win;
		const uint32_t k = bits & 127;
		const uint32_t e = k < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)tlo, (int)k) : (uint32_t)__builtin_amdgcn_readlane((int)thi, (int)(k - 64));
		if (e == 0xff) return pos + 7 <= b.total_bits ? -3 : 1;
		const int sym = (int)(e & 31), len = (int)(e >> 5);
		if (pos + (uint32_t)len > b.total_bits) return 1;
		pos += (uint32_t)len;
		win >>= len; wbits -= (uint32_t)len;
		if (sym < 16) { if (lane == 0) sm.lens[n] = (uint8_t)sym; n++; prev = sym; }
		else {
			const int eb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
			if (pos + (uint32_t)eb > b.total_bits) return 1;
			const int rep = (int)((bits >> len) & ((1u << eb) - 1)) + (sym == 18 ? 11 : 3);
			pos += (uint32_t)eb;
			win >>= eb; wbits -= (uint32_t)eb;
			int val = 0;
			if (sym == 16) { if (n == 0) return -4; val = prev; }
			if (n + rep > total) return -5;
			for (int q = lane; q < rep; q += 64) sm.lens[n + q] = (uint8_t)val;
			n += rep;
			if (sym != 16) prev = 0;
		}
	}
	b.pos = pos;
	tbits = (uint32_t)(pos - start);
	__syncthreads();
	if (uni(sm.lens[256]) == 0) return -6;
	// over-subscription check: lane per symbol
	uint32_t k1 = 0, k2 = 0;
	for (int i = lane; i < hlit; i += 64) if (sm.lens[i]) k1 += 1u << (15 - sm.lens[i]);
	if (lane < hdist && sm.lens[hlit + lane]) k2 = 1u << (15 - sm.lens[hlit + lane]);
	for (int o = 32; o > 0; o >>= 1) { k1 += __shfl_xor(k1, o, 64); k2 += __shfl_xor(k2, o, 64); }
	if (uni(k1) > (1u << 15) || uni(k2) > (1u << 15)) return -7;
	return 0;
}

// job.resume: rembytecnt | sfbt << 16 | subc << 20.  results: tebc field carries out_rembytecnt.
// W16 (with GW; nxz_inflate_stream's pieces): the history in front of the output is not known yet.
// The target holds 16-bit elements (job.dst_cap counts elements): a byte, or 0x8000 | k for "byte k
// of the 32 KiB that precede this output", which is what a match that reaches back before the output
// copies; matches copy elements, so such references travel.  No history bytes are read, no checksums.
template <bool GW, bool W16>
__device__ __forceinline__ void inflate_body(const nxz_batch_job_t *__restrict__ jobs,
					     nxz_batch_result_t *__restrict__ results,
					     nxz_batch_dht_t *__restrict__ dht_io, const Built *__restrict__ built, const uint32_t *__restrict__ order)
{
	__shared__ __attribute__((aligned(16))) SmemT<GW, W16> sm;
	typedef typename SmemT<GW, W16>::elem_t elem_t;
	const int lane = threadIdx.x;
	// order (may be NULL): the jobs by falling source length -- a launch of a few rounds of wavefronts ends with its slowest
	// stream, which had better be among the first to start (nxz_launch_inflate)
	const uint32_t jid = order ? order[blockIdx.x] : blockIdx.x;
	if (jid == 0xffffffffu) return;                      // (a slot of the order that holds no job: nxz_inflate_cut.hip)
	const nxz_batch_job_t job = jobs[jid];
	// (pieces of a stream come longest first, and the launch ends with its slowest piece: the sixteenth of them in front
	// gets the instruction issue of its SIMD before the others, the next quarter before the rest)
	if (W16) {
		if (blockIdx.x * 16 < gridDim.x) __builtin_amdgcn_s_setprio(3);
		else if (blockIdx.x * 4 < gridDim.x) __builtin_amdgcn_s_setprio(2);
		else if (blockIdx.x * 2 < gridDim.x) __builtin_amdgcn_s_setprio(1);
	}
	// (a piece of a stream -- W16 -- has no history in front of its source: hist_len is the bit it starts at, counted
	// from src, which nxz_pinflate.cpp keeps 16-byte aligned inside the caller's stream)
	const uint32_t hist_bytes = W16 ? 0 : job.hist_len < job.src_len ? job.hist_len : job.src_len;
	const uint32_t hist = W16 ? WIN : hist_bytes;      // how far back a match may reach before the output
	uint32_t srclen = job.src_len - hist_bytes;          // (a piece of a stream may run on behind it: `ext` below)
	const NXZ_GLOBAL_AS uint8_t *src = (const NXZ_GLOBAL_AS uint8_t *)job.src + hist_bytes;
	NXZ_GLOBAL_AS uint8_t *dst = (NXZ_GLOBAL_AS uint8_t *)job.dst;
	const uint32_t cap = job.dst_cap;
	// window access: position p counts output bytes, negative positions (as uint32) are history
	const NXZ_GLOBAL_AS uint8_t *hist_end = (const NXZ_GLOBAL_AS uint8_t *)job.src + hist_bytes;
	NXZ_GLOBAL_AS uint16_t *dst16 = (NXZ_GLOBAL_AS uint16_t *)job.dst;
	auto wr = [&](uint32_t p, uint32_t v) __attribute__((always_inline)) {
		if (W16 && GW) dst16[p] = (uint16_t)v;
		else if (GW) dst[p] = (uint8_t)v;
		else sm.win[p & WMASK] = (elem_t)v;
	};
	uint32_t out = 0, flushed = 0;             // bytes produced / bytes already written to dst
	// read position p of the window for a match that writes at `at` (p is 1..32768 bytes behind `at`;
	// what lies in front of position 0 is the history)
	auto rd = [&](uint32_t p, uint32_t at) __attribute__((always_inline)) -> uint32_t {
		if (W16 && GW) { const uint32_t back = at - p; return back > at ? 0x8000u | (WIN - (back - at)) : dst16[p]; }
		if (GW) { const uint32_t back = at - p; return back > at ? hist_end[-(ptrdiff_t)(back - at)] : dst[p]; }
		return sm.win[p & WMASK];
	};
	// A match of len bytes from dist back, written at `at`: lane per byte (the source pattern repeats
	// with period dist when dist < len).  Every source byte lies in front of `at`, so all the loads
	// (five at most: len <= 258) are issued before the first store: one round trip to the window per
	// match instead of one per 64 bytes.  dist and len are wave-uniform: the branches are scalar.
	auto copy_match = [&](uint32_t at, uint32_t dist, uint32_t len) __attribute__((always_inline)) {
		at = __builtin_amdgcn_readfirstlane(at); dist = __builtin_amdgcn_readfirstlane(dist); len = __builtin_amdgcn_readfirstlane(len);
		// i % dist without a division: one multiplication by the rounded-up reciprocal is exact for these
		// small numbers (checked for every dist < 259, i < 320, with the reciprocal off by 2 ulp either way)
		const uint32_t m = dist < len ? (uint32_t)(__builtin_amdgcn_rcpf((float)dist) * 1048576.0f) + 1 : 0;
		if (len <= 64) {
			uint32_t i = lane;
			if (dist < len) i -= ((i * m) >> 20) * dist;
			const uint32_t v = (uint32_t)lane < len ? rd(at - dist + i, at) : 0;
			if ((uint32_t)lane < len) wr(at + lane, v);
			return;
		}
		// device memory, source entirely inside the output and not overlapping the copy (the long matches of
		// highly repetitive data): eight bytes per lane and load -- four 16-bit elements, or eight bytes --
		// instead of one element; device memory takes these loads and stores at any alignment
		if (GW && dist >= len && dist <= at) {
			typedef uint32_t v2u_any __attribute__((ext_vector_type(2), aligned(1)));
			constexpr uint32_t PER = W16 ? 4 : 8;                      // elements per lane and trip
			const NXZ_GLOBAL_AS uint8_t *sp = W16 ? (const NXZ_GLOBAL_AS uint8_t *)(dst16 + (at - dist)) : dst + (at - dist);
			NXZ_GLOBAL_AS uint8_t *dp = W16 ? (NXZ_GLOBAL_AS uint8_t *)(dst16 + at) : dst + at;
			const uint32_t full = len / PER;                          // whole groups; <= 64 for 16-bit elements, <= 32 for bytes
			v2u_any g0 = { 0, 0 }, g1 = { 0, 0 };
			if ((uint32_t)lane < full) g0 = *(const NXZ_GLOBAL_AS v2u_any *)(sp + 8 * lane);
			if (W16 && (uint32_t)lane + 64 < full) g1 = *(const NXZ_GLOBAL_AS v2u_any *)(sp + 8 * (lane + 64));
			const uint32_t rest = len - full * PER, e = full * PER + lane;          // the last 0..7 elements, one per lane
			uint32_t tail = 0;
			if ((uint32_t)lane < rest) tail = W16 ? dst16[at - dist + e] : dst[at - dist + e];
			if ((uint32_t)lane < full) *(NXZ_GLOBAL_AS v2u_any *)(dp + 8 * lane) = g0;
			if (W16 && (uint32_t)lane + 64 < full) *(NXZ_GLOBAL_AS v2u_any *)(dp + 8 * (lane + 64)) = g1;
			if ((uint32_t)lane < rest) { if (W16) dst16[at + e] = (uint16_t)tail; else dst[at + e] = (uint8_t)tail; }
			return;
		}
		uint32_t v[5];
#pragma unroll
		for (int k = 0; k < 5; k++) {
			const uint32_t i = lane + 64 * k;
			const uint32_t r = dist < len ? i - ((i * m) >> 20) * dist : i;
			v[k] = (k < 2 || len > 64u * k) && i < len ? rd(at - dist + r, at) : 0;
		}
#pragma unroll
		for (int k = 0; k < 5; k++) {
			const uint32_t i = lane + 64 * k;
			if ((k < 2 || len > 64u * k) && i < len) wr(at + i, v[k]);
		}
	};
#ifdef NXZ_INFLATE_PROF
	unsigned long long *prof = prof_buf;
	unsigned long long tprev = prof ? clock64() : 0;
	unsigned long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

	// history (the last <= 32 KiB before the output) goes into the window just below position 0
	{
		uint32_t h = hist_bytes > WIN ? WIN : hist_bytes;
		const NXZ_GLOBAL_AS uint8_t *hp = (const NXZ_GLOBAL_AS uint8_t *)job.src + (hist_bytes - h);
		if (!GW && !W16) for (uint32_t i = lane; i < h; i += 64) sm.win[(0u - h + i) & WMASK] = hp[i];
		// (16-bit elements: what lies in front of the output is unknown -- byte k of those 32 KiB is the element 0x8000 | k)
		if (!GW && W16) for (uint32_t i = lane; i < WIN; i += 64) sm.win[i] = (elem_t)(0x8000u | i);
	}
	__syncthreads();

	Bits b;
	b.src = src; b.srclen = srclen; b.total_bits = (uint64_t)srclen * 8; b.pos = 0;
	b.stage_base = 0xffffffffu; b.stage = sm.stage; b.lane = lane;
	uint32_t in_subc = (job.resume >> 20) & 7, in_sfbt = (job.resume >> 16) & 15, in_rem = job.resume & 0xffff;
	if (srclen && in_subc) b.pos = 8 - in_subc;
	if (W16 && job.hist_len < b.total_bits) b.pos = job.hist_len;
	// (pieces of a stream, which carry no checksums: in_adler, when not 0, is the bit of the source where the next
	// piece begins -- no token that starts there or reaches beyond it is decoded, so a piece that is in step
	// with the next one suspends exactly there; the number of block headers read comes back in its place)
	if (W16 && job.in_adler && job.in_adler < b.total_bits) b.total_bits = job.in_adler;
	uint32_t nheaders = 0;
	const unsigned long long t_begin = W16 ? wall_clock64() : 0;      // (pieces report how long they took, 10 ns units, where a job reports its CRC)

	uint32_t crc_state = job.in_crc ^ 0xffffffffu;
	uint32_t ad1 = job.in_adler & 0xffff, ad2 = job.in_adler >> 16;
	int state = 0;                             // 0 header, 1 stored, 2 coded
	uint32_t bfinal = 0, btype = 0, rem = 0;
	uint32_t cc = 0, o_sfbt = 0, o_subc = 0, o_rem = 0, dhtbits = 0;
	bool final_eob = false, have_dht = false;
	// job.reserved bit 0 (NXZ_JOB_SUSPEND_WHEN_FULL, additive): a full target is no error (CC 13) but a place
	// to suspend, like the end of the source: the job reports the state in front of the token that did
	// not fit, and how much of the source it has used, and is resumed with the rest
	const bool stop_full = (job.reserved & 1) != 0;
	uint64_t stop_bits = ~0ull;                    // source position (bits) of such a suspension
	// A piece of a stream (W16) that reads the stream in place and whose range ends inside a STORED block -- the start
	// it was cut at lies in stored data that looks like a header -- runs on: dht_index says how many bytes of the stream
	// lie behind its range.  It copies the block out, follows the stored blocks behind it while they fit its room and
	// stands at the first header that is no stored block's, or whose block does not fit (nxz_pinflate.cpp finds or makes
	// the piece that starts there; up to round 3's end such a piece stopped where its range ended, and a second round of
	// launches decoded the rest of the run: 0.2-0.35 ms a stream).
	uint32_t ext_avail = (W16 && GW) ? job.dht_index : 0, ext_total = 0;       // (the bit the next piece starts at -- in_adler -- does not hold a stored block back: a cut inside one is none)
	bool ext = false;

	// flush window bytes [flushed, upto) to dst and fold them into the checksums
	auto flush = [&](uint32_t upto) {
		IPROF(0);
		if (GW) { flushed = upto; return; }              // the bytes are in place
		// (checksums: nxzl::cksum_kernel over the finished output, for both forms of the window -- a lane
		// walking its 256 bytes through a table costs the wave 80 K cycles per 16 KiB, the kernel 1 K)
		while (flushed < upto) {
			uint32_t n = upto - flushed < FLUSH ? upto - flushed : FLUSH;
			// coalesced copy out: 16 bytes per lane from where `flushed` is a multiple of that (it is a multiple of
			// FLUSH unless a stored stretch went straight to the target, below), single elements in front and behind
			constexpr uint32_t EPV = 16 / sizeof(elem_t);             // elements per 16 bytes
			NXZ_GLOBAL_AS elem_t *de = (NXZ_GLOBAL_AS elem_t *)job.dst;
			uint32_t head = (EPV - (flushed & (EPV - 1))) & (EPV - 1);
			if (head > n) head = n;
			if ((uint32_t)lane < head) de[flushed + lane] = sm.win[(flushed + lane) & WMASK];
			for (uint32_t i = head + lane * EPV; i < n; i += 64 * EPV) {
				if (i + EPV <= n) {
					uint4 v = *(const uint4 *)&sm.win[(flushed + i) & WMASK];
					*(uint4 *)(de + flushed + i) = v;
				} else {
					for (uint32_t k = i; k < n; k++) de[flushed + k] = sm.win[(flushed + k) & WMASK];
				}
			}
			flushed += n;
			__syncthreads();
		}
		IPROF(1);
	};

	// two 256-byte blocks of the source in registers (lane k: dword k), for the multi-token step
	const bool fast_ok = ((uintptr_t)src & 3) == 0;
	bool lit_mode = false;                                    // the last multi-token step met literals only: see there
	constexpr bool lit_step_on = NXZ_INFLATE_LIT_STEP != 0;
	// The chain of real token starts through the lanes' answers (tl: the bits of the token that would start at a lane's bit,
	// 0: none for this step): every lane names the lane its token ends at -- itself when there is no token, or when the
	// token reaches past the lanes -- and the walk is one v_readlane a link, the lane it reads being what the last one read
	// out; a lane that names itself is where the chain ends (a link more does no harm: groups of four without a test).
	// starts: the lanes on the chain that hold a token; off: the bits they use, all told.
	// (Up to round 5 a link was ten scalar instructions -- the length read out, tested, added up, the start's bit shifted
	// into place -- and the scalar unit, one for the CU's four SIMDs, was what a full batch waited for: 74 % busy.)
	auto walk_chain = [&](const uint32_t tl, uint64_t &starts, uint32_t &off) __attribute__((always_inline)) {
#if NXZ_INFLATE_CHAIN_BY_LANE
		const uint32_t to = (uint32_t)lane + tl;
		const uint32_t nxt = (tl && to < 64) ? to : (uint32_t)lane;
		const uint64_t valid = __ballot(tl != 0);
		uint32_t at = 0;
		uint64_t seen = 0;
		for (;;) {
			uint32_t before = 0;
#pragma unroll
			for (int u = 0; u < 4; u++) {
				seen |= 1ull << at;
				before = at;
				at = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)at);
			}
			if (at == before) break;
		}
		starts = seen & valid;
		off = 0;
		if (starts) {
			const uint32_t last = 63 - (uint32_t)__builtin_clzll(starts);
			off = last + (uint32_t)__builtin_amdgcn_readlane((int)tl, (int)last);
		}
#else
		off = 0; starts = 0;
		for (;;) {
			uint32_t t = 0;
#pragma unroll
			for (int u = 0; u < 4; u++) {
				t = (uint32_t)__builtin_amdgcn_readlane((int)tl, (int)(off & 63));
				t = off < 64 ? t : 0;
				starts |= (uint64_t)((t + 63) >> 6) << (off & 63);         // t is 0..63: 1 for a token, in scalar arithmetic
				off += t;
			}
			if (!t || off > 63) break;
		}
#endif
	};
	uint32_t W0 = 0, W1 = 0, wbase = 0x80000000u;          // wbase: dword index of W0's lane 0 (a multiple of 64; none yet)
	auto load_block = [&](uint32_t blk) -> uint32_t {
		const uint32_t idx = blk * 64 + lane;
		const uint64_t byte = (uint64_t)idx * 4;
		uint32_t w = 0;
		if (byte + 4 <= srclen) w = ((const uint32_t *)src)[idx];
		else for (uint32_t k = 0; byte + k < srclen; k++) w |= (uint32_t)src[byte + k] << (8 * k);
		return w;
	};

	// resume state
	if (in_sfbt & 8) {
		uint32_t kind = (in_sfbt >> 1) & 7;
		bfinal = in_sfbt & 1;
		if (kind == 4) { state = 1; btype = 0; rem = in_rem; }
		else if (kind == 5) { state = 2; btype = 1; }
		else if (kind == 6) {
			state = 2; btype = 2;
			// re-parse the table handed back by the caller -- or, a piece of a stream that begins at a cut inside a block
			// whose tables were built for the whole block (in_crc, which such pieces have no use for: which): load them
			const nxz_batch_dht_t *t = &dht_io[jid];
			if (W16 && built && job.in_crc) {
				load_built(sm, &built[job.in_crc - 1], lane);
				have_dht = true; dhtbits = t->dhtlen;
				b.stage_base = 0xffffffffu;
				goto tables_ready;
			}
			Bits tb;
			tb.src = (const NXZ_GLOBAL_AS uint8_t *)t->dht; tb.srclen = (t->dhtlen + 7) / 8; tb.total_bits = t->dhtlen; tb.pos = 0;
			tb.stage_base = 0xffffffffu; tb.stage = sm.stage; tb.lane = lane;
			int hlit, hdist; uint32_t tbits;
			int rc = read_dht(tb, sm, hlit, hdist, tbits);
			if (rc != 0 || tbits != t->dhtlen) { cc = NXZ_CC_INVALID_DHT; goto done; }
			build<LBITS>(sm.hl, sm.lens, hlit, lane);
			build<DBITS>(sm.hd, sm.lens + hlit, hdist, lane);
			have_dht = true; dhtbits = t->dhtlen;
			b.stage_base = 0xffffffffu;
		}
	}
tables_ready:
	if (state == 2 && btype == 1) {
		for (int i = lane; i < 288; i += 64) sm.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
		if (lane < 30) sm.lens[288 + lane] = 5;
		__syncthreads();
		build<LBITS>(sm.hl, sm.lens, 288, lane);
		build<DBITS>(sm.hd, sm.lens + 288, 30, lane);
	}

	for (;;) {
		if (state != 2) b.bb_sync();
		if (state == 0) {
			uint64_t hdr = b.pos;
			if (!b.have(3)) { o_sfbt = 0xe; o_subc = (uint32_t)(b.total_bits - hdr); break; }
			uint32_t v = b.peek();
			if (W16 && ext && ((v >> 1) & 3) != 0) { o_sfbt = 0xe; stop_bits = hdr; break; }      // (ran on through stored blocks: this one is none)
			bfinal = v & 1; btype = (v >> 1) & 3;
			b.pos += 3;
			nheaders++;
			if (btype == 0) {
				b.pos = (b.pos + 7) & ~7ull;
				if (!b.have(32)) {
					if (W16 && ext) { o_sfbt = 0xe; stop_bits = hdr; break; }
					o_sfbt = 0xe | bfinal; o_subc = (uint32_t)(b.total_bits - hdr); break;
				}
				uint32_t w = b.peek();
				b.pos += 32;
				if (((w ^ (w >> 16)) & 0xffff) != 0xffff) {
					if (W16 && ext) { o_sfbt = 0xe; stop_bits = hdr; break; }                       // (whoever starts at this header says what is wrong with it)
					cc = NXZ_CC_INVALID_DHT; break;
				}
				// (running on: a block that does not fit the room, or does not lie whole inside the stream, is left to the piece that starts here)
				if (W16 && ext && ((w & 0xffff) > cap - out || (uint64_t)(w & 0xffff) * 8 > b.total_bits - b.pos)) { o_sfbt = 0xe; stop_bits = hdr; break; }
				rem = w & 0xffff;
				state = 1;
			} else if (btype == 1) {
				for (int i = lane; i < 288; i += 64) sm.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
				if (lane < 30) sm.lens[288 + lane] = 5;
				__syncthreads();
				build<LBITS>(sm.hl, sm.lens, 288, lane);
				build<DBITS>(sm.hd, sm.lens + 288, 30, lane);
				state = 2;
			} else if (btype == 2) {
				int hlit, hdist; uint32_t tbits;
				uint64_t tstart = b.pos;
				int rc = read_dht(b, sm, hlit, hdist, tbits);
				if (rc == 1) { o_sfbt = 0xe | bfinal; o_subc = (uint32_t)(b.total_bits - hdr); break; }
				if (rc < 0) { cc = NXZ_CC_INVALID_DHT; break; }
				build<LBITS>(sm.hl, sm.lens, hlit, lane);
				build<DBITS>(sm.hd, sm.lens + hlit, hdist, lane);
				// keep the table bits for a possible suspend inside this block
				if (dht_io) {
					nxz_batch_dht_t *t = &dht_io[jid];
					uint64_t save = b.pos;
					b.ensure((uint32_t)(tstart >> 3), 320);
					for (uint32_t i = lane; i < (tbits + 31) / 32; i += 64) {
						b.pos = tstart + (uint64_t)i * 32;
						uint32_t w = b.raw_peek();
						if ((i + 1) * 32 > tbits && (tbits & 31)) w &= (1u << (tbits & 31)) - 1;
						((uint32_t *)t->dht)[i] = w;       // dht[] is 4-byte aligned inside the struct
					}
					b.pos = save;
					if (lane == 0) t->dhtlen = tbits;
				}
				have_dht = true; dhtbits = tbits;
				state = 2;
			} else { cc = NXZ_CC_INVALID_DHT; break; }
		} else if (state == 1) {
			// stored bytes: byte aligned; copy through the window
			uint32_t srcleft = (uint32_t)((b.total_bits - b.pos) >> 3);
			if (W16 && ext_avail && rem > srcleft && rem <= cap - out && rem - srcleft <= ext_avail) {
				srclen += ext_avail; b.srclen = srclen; b.total_bits = (uint64_t)srclen * 8;
				ext_total = ext_avail; ext_avail = 0; ext = true;
				srcleft = (uint32_t)((b.total_bits - b.pos) >> 3);
			}
			uint32_t n = rem < srcleft ? rem : srcleft;
			bool full_stop = false;
			if (n > cap - out) {
				if (!stop_full) { cc = NXZ_CC_TARGET_SPACE; break; }
				n = cap - out; full_stop = true;                   // what fits, then suspend inside the stored block
			}
			uint32_t sp = (uint32_t)(b.pos >> 3);
			if (GW) {
				// the target is the window: four bytes per lane and trip (device memory takes them at any alignment)
				typedef uint32_t u32_any __attribute__((aligned(1)));
				typedef uint32_t v2u_any __attribute__((ext_vector_type(2), aligned(1)));
				const uint32_t quads = n >> 2;
				// (eight loads in flight a lane: one at a time a lone wavefront copied a stored block at 50 MB/s -- 2 ms for the 110 KB
				// run of stored blocks that is ONE piece of a stream)
				for (uint32_t i0 = 0; i0 < quads; i0 += 512) {
					uint32_t w[8];
#pragma unroll
					for (int u = 0; u < 8; u++) { const uint32_t i = i0 + 64 * u + lane; w[u] = i < quads ? *(const NXZ_GLOBAL_AS u32_any *)(src + sp + 4 * i) : 0; }
#pragma unroll
					for (int u = 0; u < 8; u++) {
						const uint32_t i = i0 + 64 * u + lane;
						if (i >= quads) continue;
						if (W16) *(NXZ_GLOBAL_AS v2u_any *)(dst16 + out + 4 * i) = (v2u_any){ (w[u] & 0xff) | ((w[u] & 0xff00) << 8), ((w[u] >> 16) & 0xff) | ((w[u] >> 24) << 16) };
						else *(NXZ_GLOBAL_AS u32_any *)(dst + out + 4 * i) = w[u];
					}
				}
				for (uint32_t i = quads * 4 + lane; i < n; i += 64) wr(out + i, src[sp + i]);
				out += n; sp += n; rem -= n; n = 0;
				flushed = out;
				__syncthreads();
			} else if (n >= 1024) {
				// window in LDS: a long stored stretch goes straight to the target the same way (what the window holds
				// is written out first), and only its last 32 KiB into the window
				__syncthreads();
				flush(out);
				typedef uint32_t u32_any __attribute__((aligned(1)));
				typedef uint32_t v2u_any __attribute__((ext_vector_type(2), aligned(1)));
				const uint32_t quads = n >> 2;
				// (eight loads in flight a lane: one at a time a lone wavefront copied a stored block at 50 MB/s -- 2 ms for the 110 KB
				// run of stored blocks that is ONE piece of a stream)
				for (uint32_t i0 = 0; i0 < quads; i0 += 512) {
					uint32_t w[8];
#pragma unroll
					for (int u = 0; u < 8; u++) { const uint32_t i = i0 + 64 * u + lane; w[u] = i < quads ? *(const NXZ_GLOBAL_AS u32_any *)(src + sp + 4 * i) : 0; }
#pragma unroll
					for (int u = 0; u < 8; u++) {
						const uint32_t i = i0 + 64 * u + lane;
						if (i >= quads) continue;
						if (W16) *(NXZ_GLOBAL_AS v2u_any *)(dst16 + out + 4 * i) = (v2u_any){ (w[u] & 0xff) | ((w[u] & 0xff00) << 8), ((w[u] >> 16) & 0xff) | ((w[u] >> 24) << 16) };
						else *(NXZ_GLOBAL_AS u32_any *)(dst + out + 4 * i) = w[u];
					}
				}
				for (uint32_t i = quads * 4 + lane; i < n; i += 64) { if (W16) dst16[out + i] = src[sp + i]; else dst[out + i] = src[sp + i]; }
				const uint32_t keep = n < WIN ? n : WIN;
				for (uint32_t i = n - keep + lane; i < n; i += 64) sm.win[(out + i) & WMASK] = (elem_t)src[sp + i];
				out += n; sp += n; rem -= n; n = 0;
				flushed = out;
				__syncthreads();
			}
			while (n) {
				uint32_t room = FLUSH - (out - flushed);
				uint32_t k = n < room ? n : room;
				for (uint32_t i = lane; i < k; i += 64) wr(out + i, src[sp + i]);
				out += k; sp += k; n -= k; rem -= k;
				__syncthreads();
				if (out - flushed == FLUSH) flush(out);
			}
			b.pos = (uint64_t)sp * 8;
			if (full_stop) { o_sfbt = 0x8 | bfinal; o_subc = 0; o_rem = rem; stop_bits = b.pos; break; }
			if (rem) { o_sfbt = 0x8 | bfinal; o_subc = (uint32_t)(b.total_bits - b.pos);       // (0, unless a piece was cut at a bit inside a byte)
				 o_rem = rem; break; }
			if (bfinal) { final_eob = true; break; }
			state = 0;
		} else {
			// ---- multi-token step (see the header) ----
			while (fast_ok && b.pos + 256 <= b.total_bits) {
				IPROF(0);
				// (the position and the window's base are the same in all lanes: said so, or the compiler keeps them in
				// vector registers and every word of the five below costs eight vector instructions instead of two)
				const uint32_t q = uni((uint32_t)(b.pos >> 5)), sh = uni((uint32_t)b.pos & 31);
				uint32_t wb = uni(wbase);
				if (q - wb >= 64u) {                                     // (one scalar test a step; the source is below 2^31 dwords)
					if (q - wb < 128u) { wb += 64; W0 = W1; W1 = load_block((wb >> 6) + 1); }
					else { wb = q & ~63u; W0 = load_block(wb >> 6); W1 = load_block((wb >> 6) + 1); }
					wbase = wb;
				}
				wb = uni(wb);
				const uint32_t qi = q - wb;                              // 0..63: dwords qi..qi+4 are in W0/W1
				// (both registers are read and one result is picked: a select where a branch would be)
				auto word = [&](uint32_t i) __attribute__((always_inline)) -> uint32_t {
					const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)W0, (int)(i & 63)), hi = (uint32_t)__builtin_amdgcn_readlane((int)W1, (int)(i & 63));
					return i < 64 ? lo : hi;
				};
				// (five dwords from qi on: in W0 alone fifteen steps in sixteen -- one v_readlane each then, no test and no select)
				uint32_t s0, s1, s2, s3, s4;
				if (qi < 60) {
					s0 = (uint32_t)__builtin_amdgcn_readlane((int)W0, (int)qi); s1 = (uint32_t)__builtin_amdgcn_readlane((int)W0, (int)(qi + 1));
					s2 = (uint32_t)__builtin_amdgcn_readlane((int)W0, (int)(qi + 2)); s3 = (uint32_t)__builtin_amdgcn_readlane((int)W0, (int)(qi + 3));
					s4 = (uint32_t)__builtin_amdgcn_readlane((int)W0, (int)(qi + 4));
				} else { s0 = word(qi); s1 = word(qi + 1); s2 = word(qi + 2); s3 = word(qi + 3); s4 = word(qi + 4); }
				// this lane's 64 bits of the source: [pos + lane, pos + lane + 64)
				const uint32_t bo = sh + (uint32_t)lane, di = bo >> 5, r = bo & 31;        // di = 0..2
				const uint32_t a0 = di == 0 ? s0 : di == 1 ? s1 : s2;
				const uint32_t a1 = di == 0 ? s1 : di == 1 ? s2 : s3;
				const uint32_t w0 = __builtin_amdgcn_alignbit(a1, a0, r);
				// the whole token that would start at this lane's bit: literal, or length + distance
				const uint32_t el = sm.hl.fast[w0 & ((1u << LBITS) - 1)];
				if (lit_mode) {
					// A stretch of literals (image-like and packed data: nine tokens in ten are, and a block of 16384 of them took a
					// lone wavefront 2 ms): the step for literals alone -- no length and distance of the token that might start at
					// a lane's bit, the tokens' places by a count of the starts in front of them -- for as long as the step before
					// met nothing but literals.  A chain that meets something else ends there; the full step takes over.
					const uint32_t lsym = el & 0xfff;
					const uint32_t ltl = (el && lsym < 256) ? el >> 12 : 0;
					uint32_t off;
					uint64_t starts;
					walk_chain(ltl, starts, off);
					lit_mode = off > 63;                                      // (the chain left the lanes: literals all the way)
					const uint32_t cnt = (uint32_t)__builtin_popcountll(starts);
					if (!cnt || cnt > cap - out) { lit_mode = false; continue; }
					if ((starts >> lane) & 1)
						wr(out + __builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0)), lsym);
					out += cnt;
					b.pos += off;
					if (out - flushed >= FLUSH) break;
					continue;
				}
				const uint32_t a2 = di == 0 ? s2 : di == 1 ? s3 : s4;
				const uint32_t w1 = __builtin_amdgcn_alignbit(a2, a1, r);
				const uint32_t nb = el >> 12, sym = el & 0xfff;
				const bool islit = el && sym < 256;
				const bool islen = sym > 256 && sym < 257 + 29;
				uint32_t lbase, eb, dbase, ebd;
				len_params(islen ? sym - 257 : 0, lbase, eb);
				const uint32_t mlen = lbase + (__builtin_amdgcn_alignbit(w1, w0, nb) & ((1u << eb) - 1));
				const uint32_t o2 = nb + eb;                                               // <= 11 + 5
				const uint32_t ed = sm.hd.fast[__builtin_amdgcn_alignbit(w1, w0, o2) & ((1u << DBITS) - 1)];
				const uint32_t ds = ed & 0xfff;
				const bool okd = ed && ds < 30;
				dist_params(okd ? ds : 0, dbase, ebd);
				const uint32_t o3 = o2 + (ed >> 12);                                       // <= 16 + 9
				const uint32_t mdist = dbase + (__builtin_amdgcn_alignbit(w1, w0, o3) & ((1u << ebd) - 1));
				const uint32_t tl = islit ? nb : (islen && okd) ? o3 + ebd : 0;            // bits of the token; 0: not for this step
				const uint32_t ob = islit ? 1 : mlen;                                      // bytes it makes
				IPROF(2);
#ifdef NXZ_INFLATE_PROF
				const uint32_t out0 = out;
#endif
				// the chain of real token starts
				// (four links at a time without a branch: a link that meets the end of the chain -- a token that is
				// not for this step, or the end of the lanes -- stays where it is)
				uint32_t off;
				uint64_t starts;
				walk_chain(tl, starts, off);
				if (!starts) break;
				IPROF(8);
				// where each token writes: prefix sum of the byte counts over the token starts
				bool isstart = (starts >> lane) & 1;
				const uint32_t x = isstart ? ob : 0;
				uint32_t incl = x;
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);   // row_shr:1
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);   // row_shr:2
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);   // row_shr:4
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);   // row_shr:8
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);   // row_bcast:15
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);   // row_bcast:31
				const uint32_t opos = incl - x;
				// a token that does not fit the target, or a match that reaches in front of the history: the step
				// ends before it (the one-token path says what is wrong)
				const uint64_t bad = __ballot(isstart && (incl > cap - out || (!islit && mdist > out + opos + hist)));
				if (bad) {
					const uint32_t first = (uint32_t)__builtin_ctzll(bad);
					starts &= (1ull << first) - 1;
					off = first;
					if (!starts) break;
					isstart = (starts >> lane) & 1;
				}
				const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(63 - __builtin_clzll(starts)));
				IPROF(9);
				const uint64_t lits = __ballot(islit);
				// All literals of the step go out at once, then the matches in order.  In the circular LDS window
				// a literal written ahead of its turn may land on the slot of a byte, nearly 32 KiB back, that a
				// match in front of it still has to read: such a step (rare) is done token by token.
				const bool ordered = !GW && __ballot(isstart && !islit && mdist + total > WIN) != 0;
				if (!ordered && isstart && islit) wr(out + opos, sym);
				uint64_t mm = ordered ? starts : starts & ~lits;
				// Device memory: up to four matches of any length whose sources lie inside the output and in front of
				// the whole step -- all their loads (eight bytes per lane, as in copy_match), then all their
				// stores: one trip to memory for the step instead of one per match.
				// (taken when a long match is among them: short ones alone are cheaper an element per lane, below)
				if (GW && mm && __builtin_popcountll(mm) <= 4 && __ballot(isstart && !islit && ob > 64) &&
				    !__ballot(isstart && !islit && (mdist < opos + ob || mdist > out + opos))) {
					typedef uint32_t v2u_any __attribute__((ext_vector_type(2), aligned(1)));
					constexpr uint32_t PER = W16 ? 4 : 8;
					const uint32_t ES = W16 ? 2 : 1;                          // bytes per element
					NXZ_GLOBAL_AS uint8_t *base = W16 ? (NXZ_GLOBAL_AS uint8_t *)dst16 : dst;
					uint32_t mp[4], mn[4], tl_[4];
					v2u_any g[4];
#pragma unroll
					for (int k = 0; k < 4; k++) {
						mp[k] = 0; mn[k] = 0; tl_[k] = 0; g[k] = (v2u_any){ 0, 0 };
						if (mm) {
							const uint32_t l = (uint32_t)__builtin_ctzll(mm);
							mm &= mm - 1;
							mp[k] = out + (uint32_t)__builtin_amdgcn_readlane((int)opos, (int)l);
							mn[k] = (uint32_t)__builtin_amdgcn_readlane((int)ob, (int)l);
							const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)mdist, (int)l);
							const NXZ_GLOBAL_AS uint8_t *sp = base + (size_t)(mp[k] - d) * ES;
							const uint32_t full = mn[k] / PER, rest = mn[k] - full * PER;
							if ((uint32_t)lane < full) g[k] = *(const NXZ_GLOBAL_AS v2u_any *)(sp + 8 * lane);
							if ((uint32_t)lane < rest) tl_[k] = W16 ? ((const NXZ_GLOBAL_AS uint16_t *)sp)[full * PER + lane] : sp[full * PER + lane];
						}
					}
		#pragma unroll
					for (int k = 0; k < 4; k++) {
						NXZ_GLOBAL_AS uint8_t *dp = base + (size_t)mp[k] * ES;
						const uint32_t full = mn[k] / PER, rest = mn[k] - full * PER;
						if ((uint32_t)lane < full) *(NXZ_GLOBAL_AS v2u_any *)(dp + 8 * lane) = g[k];
						if ((uint32_t)lane < rest) { if (W16) ((NXZ_GLOBAL_AS uint16_t *)dp)[full * PER + lane] = (uint16_t)tl_[k]; else dp[full * PER + lane] = (uint8_t)tl_[k]; }
					}
				}
				// (window in LDS) up to four short matches whose sources lie in front of the whole step: all their
				// loads, then all their stores
				else if (!ordered && mm && __builtin_popcountll(mm) <= 4 && !__ballot(isstart && !islit && (ob > 64 || mdist < opos + ob))) {
					uint32_t mp[4], mn[4], mv[4];
#pragma unroll
					for (int k = 0; k < 4; k++) {
						mp[k] = 0; mn[k] = 0; mv[k] = 0;
						if (mm) {
							const uint32_t l = (uint32_t)__builtin_ctzll(mm);
							mm &= mm - 1;
							mp[k] = out + (uint32_t)__builtin_amdgcn_readlane((int)opos, (int)l);
							mn[k] = (uint32_t)__builtin_amdgcn_readlane((int)ob, (int)l);
							const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)mdist, (int)l);
							if ((uint32_t)lane < mn[k]) mv[k] = rd(mp[k] - d + lane, mp[k]);
						}
					}
#pragma unroll
					for (int k = 0; k < 4; k++)
						if ((uint32_t)lane < mn[k]) wr(mp[k] + lane, mv[k]);
				}
				while (mm) {
					const uint32_t l = (uint32_t)__builtin_ctzll(mm);
					mm &= mm - 1;
					if ((lits >> l) & 1) { if ((uint32_t)lane == l) wr(out + opos, sym); continue; }
					copy_match(out + (uint32_t)__builtin_amdgcn_readlane((int)opos, (int)l), (uint32_t)__builtin_amdgcn_readlane((int)mdist, (int)l),
						   (uint32_t)__builtin_amdgcn_readlane((int)ob, (int)l));
				}
				out += total;
				lit_mode = lit_step_on && !(starts & ~lits);                  // (nothing but literals: the next step may be the short one)
				IPROF(3);
				ICOUNT(4, 1); ICOUNT(5, out - out0); ICOUNT(6, off);
				b.pos += off;
				if (out - flushed >= FLUSH) break;
			}
			if (out - flushed >= FLUSH) { __syncthreads(); flush(flushed + FLUSH); continue; }
			// the step stopped at a token it leaves to the one-token path (or never ran)
			b.bb_sync();
			ICOUNT(7, 1);
			const uint64_t sym_start = b.pos;
			const uint32_t sfbt = (btype == 1 ? 0xa : 0xc) | bfinal;
			uint32_t nb;
			b.bb_fill();
			int sym = decode_sym<LBITS>(sm.hl, (uint32_t)b.bb, nb);
			if (sym < 0 || !b.have(nb)) {
				if (!b.have(sym < 0 ? 15 : nb)) { o_sfbt = sfbt; o_subc = (uint32_t)(b.total_bits - sym_start); break; }
				cc = NXZ_CC_MISSING_CODE; break;
			}
			b.bb_drop(nb);
			if (sym < 256) {
				if (out >= cap) {
					if (stop_full) { o_sfbt = sfbt; stop_bits = sym_start; break; }   // suspend in front of this token
					cc = NXZ_CC_TARGET_SPACE; break;
				}
				if (lane == 0) wr(out, (uint32_t)sym);
				out++;
			} else if (sym == 256) {
				if (bfinal) { final_eob = true; break; }
				state = 0;
				continue;
			} else {
				sym -= 257;
				if (sym >= 29) { cc = NXZ_CC_MISSING_CODE; break; }
				uint32_t lbase, eb;
				len_params((uint32_t)sym, lbase, eb);
				b.bb_fill();
				if (!b.have(eb)) { o_sfbt = sfbt; o_subc = (uint32_t)(b.total_bits - sym_start); break; }
				uint32_t len = lbase + ((uint32_t)b.bb & ((1u << eb) - 1));
				b.bb_drop(eb);
				b.bb_fill();
				int ds = decode_sym<DBITS>(sm.hd, (uint32_t)b.bb, nb);
				if (ds < 0 || !b.have(nb)) {
					if (!b.have(ds < 0 ? 15 : nb)) { o_sfbt = sfbt; o_subc = (uint32_t)(b.total_bits - sym_start); break; }
					cc = NXZ_CC_INVALID_DIST; break;
				}
				if (ds >= 30) { cc = NXZ_CC_INVALID_DIST; break; }
				b.bb_drop(nb);
				uint32_t dbase;
				dist_params((uint32_t)ds, dbase, eb);
				b.bb_fill();
				if (!b.have(eb)) { o_sfbt = sfbt; o_subc = (uint32_t)(b.total_bits - sym_start); break; }
				uint32_t dist = dbase + ((uint32_t)b.bb & ((1u << eb) - 1));
				b.bb_drop(eb);
				if (dist > out + hist || dist > WIN) { cc = NXZ_CC_INVALID_DIST; break; }
				if (len > cap - out) {
					if (stop_full) { o_sfbt = sfbt; stop_bits = sym_start; break; }
					cc = NXZ_CC_TARGET_SPACE; break;
				}
				copy_match(out, dist, len);
				out += len;
			}
			if (out - flushed >= FLUSH) { __syncthreads(); flush(flushed + FLUSH); }
		}
	}
	if (final_eob) { o_sfbt = 0; o_subc = (uint32_t)(b.total_bits - b.pos); }
done:
	__syncthreads();
	if (cc == 0) flush(out);
	IPROF(0);
#ifdef NXZ_INFLATE_PROF
	if (prof && lane == 0) for (int k = 0; k < 12; k++) __hip_atomic_fetch_add(&prof[k], pacc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
	if (lane == 0) {
		nxz_batch_result_t r;
		uint32_t spbc = job.src_len + ext_total, subc = o_subc;
		if (stop_bits != ~0ull) {                      // the source bytes touched, and the bits of the last one that are not used yet
			const uint32_t touched = (uint32_t)((stop_bits + 7) >> 3);
			spbc = hist_bytes + touched; subc = touched * 8 - (uint32_t)stop_bits;
		}
		if (final_eob && subc > 0xfff8) {              // 16-bit SUBC: leave the excess unread
			uint32_t drop = (subc - 0xfff8 + 7) / 8;
			spbc -= drop; subc -= drop * 8;
		}
		if (cc == 0 && !(final_eob && subc < 8)) cc = NXZ_CC_DATA_LENGTH;
		r.cc = cc; r.tpbc = (cc == 0 || cc == NXZ_CC_DATA_LENGTH) ? out : 0;
		r.tebc = o_rem; r.spbc = spbc;
		r.crc = W16 ? (uint32_t)(wall_clock64() - t_begin) : crc_state ^ 0xffffffffu; r.adler = W16 ? nheaders : (ad2 << 16) | ad1;
		r.subc = subc; r.sfbt = o_sfbt | (final_eob ? 0x100u : 0) | (((o_sfbt & 0xe) == 0xc && have_dht) ? (dhtbits << 16) : 0);
		results[jid] = r;
	}
}

// The kernels.  With the target as window (GW) a stream needs 6.6 KiB of LDS and what bounds the number of
// wavefronts per SIMD is registers: five at 96 VGPRs, four at 97 -- and the fifth is worth a tenth of the
// throughput of a full batch (51.8 against 46.7 GiB/s at 65 536 zlib -6 streams).  The compiler is told so; left to
// itself it takes 95 to 104 registers from one edit of this file to the next.  With the window in LDS only
// one or two streams fit a CU anyway.
template <bool GW, bool W16 = false>
__global__ void inflate_kernel(const nxz_batch_job_t *__restrict__ jobs, nxz_batch_result_t *__restrict__ results,
			       nxz_batch_dht_t *__restrict__ dht_io, const Built *__restrict__ built, const uint32_t *__restrict__ order);
template <>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5))) void inflate_kernel<true, false>(const nxz_batch_job_t *__restrict__ jobs,
		nxz_batch_result_t *__restrict__ results, nxz_batch_dht_t *__restrict__ dht_io, const Built *__restrict__ built, const uint32_t *__restrict__ order)
{
	inflate_body<true, false>(jobs, results, dht_io, built, order);
}
template <>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5))) void inflate_kernel<true, true>(const nxz_batch_job_t *__restrict__ jobs,
		nxz_batch_result_t *__restrict__ results, nxz_batch_dht_t *__restrict__ dht_io, const Built *__restrict__ built, const uint32_t *__restrict__ order)
{
	inflate_body<true, true>(jobs, results, dht_io, built, order);
}
template <>
__global__ __launch_bounds__(64) void inflate_kernel<false, false>(const nxz_batch_job_t *__restrict__ jobs,
		nxz_batch_result_t *__restrict__ results, nxz_batch_dht_t *__restrict__ dht_io, const Built *__restrict__ built, const uint32_t *__restrict__ order)
{
	inflate_body<false, false>(jobs, results, dht_io, built, order);
}
template <>
__global__ __launch_bounds__(64) void inflate_kernel<false, true>(const nxz_batch_job_t *__restrict__ jobs,
		nxz_batch_result_t *__restrict__ results, nxz_batch_dht_t *__restrict__ dht_io, const Built *__restrict__ built, const uint32_t *__restrict__ order)
{
	inflate_body<false, true>(jobs, results, dht_io, built, order);
}

// ---- token boundaries inside a dynamic block (nxz_inflate_stream, few pieces) ----
// A block is the unit the pieces of a stream are cut at, and a wavefront takes milliseconds for one: a
// caller that holds a megabyte of the stream has some forty blocks, and waits for the longest.  Inside
// a block the stream can be cut at any token boundary -- the decoder needs the block's tables and
// nothing else -- but where the tokens begin is only known to who has decoded from the block's start.
// Huffman-coded data synchronises itself, though: decoders that start at neighbouring bits fall in
// step with each other within a few dozen tokens, because a wrong start soon decodes a token that
// ends where a right one does.  So: a request names a block header and a bit somewhere in the
// block; the wavefront reads the header (as the inflate kernel does), then its 64 lanes decode token
// LENGTHS from the 64 bits that follow the guess -- one of them is a true token start, no token being
// longer than 48 bits -- each on its own, from a 2 KiB copy of the source in LDS.  Lanes that meet a
// code that does not exist, or end-of-block, drop out; when all that are left stand on the same bit
// that bit is a token boundary, if the guess was inside the block at all.  (That it is, and that the
// boundary is a true one, is not taken on trust: the piece in front must arrive exactly there, in
// this block -- nxz_pinflate.cpp.)
constexpr uint32_t SYNC_DW = 512;                 // dwords of the source a request may walk through
constexpr uint32_t SYNC_RUN = 1024;               // bits every lane decodes before the lanes are compared

// the tables of the blocks that requests or pieces will start in, a wavefront per block.  header_bit 0xffffffff:
// the block's header is not in the source (the caller holds a part of a stream that begins inside the block);
// its table is in the request's slot of `tables`, as a suspended job handed it back.  Else the header's table
// bits go into that slot, for the pieces that start inside the block (jobs that resume in a dynamic block).
__global__ __launch_bounds__(64) void block_tables_kernel(const nxz_sync_req_t *__restrict__ reqs, nxz_batch_dht_t *__restrict__ tables,
							   Built *__restrict__ built)
{
	__shared__ __attribute__((aligned(16))) SmemT<true> sm;
	const int lane = threadIdx.x;
	const nxz_sync_req_t rq = reqs[blockIdx.x];
	Bits b;
	b.src = (const NXZ_GLOBAL_AS uint8_t *)rq.src; b.srclen = rq.srclen; b.total_bits = (uint64_t)rq.srclen * 8; b.pos = rq.header_bit;
	b.stage_base = 0xffffffffu; b.stage = sm.stage; b.lane = lane;
	const bool given = rq.header_bit == 0xffffffffu;
	bool ok = given || b.have(17);
	uint32_t tbits = 0, bfinal = 0;
	uint64_t tstart = 0;
	int hlit = 0, hdist = 0;
	if (given) {
		const nxz_batch_dht_t *t = &tables[blockIdx.x];
		Bits tb;
		tb.src = (const NXZ_GLOBAL_AS uint8_t *)t->dht; tb.srclen = (t->dhtlen + 7) / 8; tb.total_bits = t->dhtlen; tb.pos = 0;
		tb.stage_base = 0xffffffffu; tb.stage = sm.stage; tb.lane = lane;
		ok = read_dht(tb, sm, hlit, hdist, tbits) == 0 && tbits == t->dhtlen;
		tbits = 0;
	} else if (ok) {
		const uint32_t v = b.peek();
		ok = (v & 6) == 4;                            // BTYPE 10
		bfinal = v & 1;
		b.pos += 3;
		tstart = b.pos;
		if (ok) ok = read_dht(b, sm, hlit, hdist, tbits) == 0;
	}
	if (ok) {
		build<LBITS>(sm.hl, sm.lens, hlit, lane);
		build<DBITS>(sm.hd, sm.lens + hlit, hdist, lane);
	}
	if (ok && !given) {
		nxz_batch_dht_t *t = &tables[blockIdx.x];
		b.stage_base = 0xffffffffu;
		b.ensure((uint32_t)(tstart >> 3), 320);
		for (uint32_t i = lane; i < (tbits + 31) / 32; i += 64) {
			b.pos = tstart + (uint64_t)i * 32;
			uint32_t w = b.raw_peek();
			if ((i + 1) * 32 > tbits && (tbits & 31)) w &= (1u << (tbits & 31)) - 1;
			((uint32_t *)t->dht)[i] = w;
		}
		if (lane == 0) t->dhtlen = tbits;
	}
	Built *o = &built[blockIdx.x];
	if (ok) {
		__syncthreads();
		uint4 *dst = (uint4 *)o;
		const uint4 *sl = (const uint4 *)&sm.hl, *sd = (const uint4 *)&sm.hd;
		for (uint32_t i = lane; i < sizeof(Huff) / 16; i += 64) dst[i] = sl[i];
		for (uint32_t i = lane; i < sizeof(HuffD) / 16; i += 64) dst[sizeof(Huff) / 16 + i] = sd[i];
	}
	if (lane == 0) { o->ok = ok ? 1 : 0; o->bfinal = bfinal; o->end_bit = given ? 0 : (uint32_t)(tstart + tbits); o->pad = 0; }
}

// (a request's header_bit names its block's tables: the index into `built`)
__global__ __launch_bounds__(64) void token_sync_kernel(const nxz_sync_req_t *__restrict__ reqs, nxz_sync_res_t *__restrict__ res,
							 const Built *__restrict__ built)
{
	__shared__ __attribute__((aligned(16))) SmemT<true> sm;
	__shared__ uint32_t region[SYNC_DW + 4];
	const int lane = threadIdx.x;
	const nxz_sync_req_t rq = reqs[blockIdx.x];
	nxz_sync_res_t out;
	out.bit = 0xffffffffu; out.lanes = 0;
	const Built *bt = &built[rq.header_bit];
	bool ok = bt->ok != 0;
	const uint64_t tstart = 0;
	const uint32_t tbits = ok ? bt->end_bit : 0;
	struct { const NXZ_GLOBAL_AS uint8_t *src; } b;
	b.src = (const NXZ_GLOBAL_AS uint8_t *)rq.src;
	if (ok) {
		load_built(sm, bt, lane);
		out.lanes = bt->bfinal << 31;                 // (BFINAL goes back in the top bit)
	}
	ok = ok && rq.guess_bit >= tstart + tbits && rq.limit_bit > rq.guess_bit + 64;
	if (ok) {
		// the stretch of the source the lanes walk through
		const uint32_t d0 = rq.guess_bit >> 5;
		const bool aligned = ((uintptr_t)rq.src & 3) == 0;
		for (uint32_t i = lane; i < SYNC_DW + 4; i += 64) {
			const uint64_t byte = (uint64_t)(d0 + i) * 4;
			uint32_t w = 0;
			if (byte + 4 <= rq.srclen && aligned) w = ((const NXZ_GLOBAL_AS uint32_t *)b.src)[d0 + i];
			else for (uint32_t k = 0; k < 4 && byte + k < rq.srclen; k++) w |= (uint32_t)b.src[byte + k] << (8 * k);
			region[i] = w;
		}
		__syncthreads();
		const uint32_t rbase = d0 * 32;
		uint32_t rend = rbase + SYNC_DW * 32;                  // a token may be looked at while it starts in front of this bit
		if (rend > rq.limit_bit) rend = rq.limit_bit;
		// (lanes that have not fallen in step 2048 bits behind the stretch each walked alone will hardly do so: the
		// request is given up -- walking on to the end of the copy would make this wavefront the one the launch waits for)
		if (rend > rq.guess_bit + 64 + SYNC_RUN + NXZ_SYNC_EXTRA) rend = rq.guess_bit + 64 + SYNC_RUN + NXZ_SYNC_EXTRA;
		uint32_t pos = rq.guess_bit + (uint32_t)lane;
		bool alive = true;
		uint32_t made = 0;                                          // bytes the tokens this lane walked over make (an estimate of the data's ratio for the caller)
		auto step = [&]() __attribute__((always_inline)) {
			const uint32_t o = pos - rbase, i = o >> 5, sh = o & 31;
			const uint32_t a0 = region[i], a1 = region[i + 1], a2 = region[i + 2];
			const uint32_t w0 = __builtin_amdgcn_alignbit(a1, a0, sh), w1 = __builtin_amdgcn_alignbit(a2, a1, sh);
			uint32_t nb = 0;
			const int sym = decode_sym<LBITS>(sm.hl, w0, nb);
			if (sym < 0 || sym == 256 || sym >= 257 + 29) { alive = false; return; }
			if (sym < 256) { pos += nb; made++; return; }
			uint32_t lbase, eb, dbase, ebd, nbd = 0;
			len_params((uint32_t)sym - 257, lbase, eb);
			made += lbase + (__builtin_amdgcn_alignbit(w1, w0, nb) & ((1u << eb) - 1));
			const uint32_t o2 = nb + eb;                                      // <= 15 + 5
			const int ds = decode_sym<DBITS>(sm.hd, __builtin_amdgcn_alignbit(w1, w0, o2), nbd);
			if (ds < 0 || ds >= 30) { alive = false; return; }
			dist_params((uint32_t)ds, dbase, ebd);
			pos += o2 + nbd + ebd;
		};
		// every lane on its own for a good stretch ...
		const uint32_t target = rq.guess_bit + 64 + SYNC_RUN;
		while (__ballot(alive && pos < target && pos + 64 <= rend)) {
			if (alive && pos < target && pos + 64 <= rend) step();
		}
		if (pos + 64 > rend && pos < target) alive = false;          // ran out of room
		// ... then all of them up to the foremost, until they stand on one bit (or none is left)
		for (int it = 0; it < 64; it++) {
			uint32_t mx = alive ? pos : 0, mn = alive ? pos : 0xffffffffu;
			for (int o = 32; o > 0; o >>= 1) {
				const uint32_t x = (uint32_t)__shfl_xor((int)mx, o, 64), y = (uint32_t)__shfl_xor((int)mn, o, 64);
				mx = x > mx ? x : mx; mn = y < mn ? y : mn;
			}
			if (mn == 0xffffffffu) break;                                     // no lane left
			if (mn == mx) {
				// (bits 8..23: bytes of output per 256 bits of source on the way here, as the first lane still walking saw it)
				const unsigned long long al = __ballot(alive);
				const uint32_t l0 = (uint32_t)__builtin_ctzll(al);
				const uint32_t m0 = (uint32_t)__shfl((int)made, (int)l0, 64), b0 = mx - (rq.guess_bit + l0);
				uint32_t per256 = b0 ? (uint32_t)(((uint64_t)m0 << 8) / b0) : 0;
				if (per256 > 0xffff) per256 = 0xffff;
				out.bit = mx; out.lanes |= (uint32_t)__popcll(al) | per256 << 8; break;
			}
			while (__ballot(alive && pos < mx)) {
				if (alive && pos < mx) { if (pos + 64 <= rend) step(); else alive = false; }
			}
		}
	}
	if (lane == 0) res[blockIdx.x] = out;
}

} // namespace nxzi

// requests for token boundaries inside dynamic blocks (see token_sync_kernel); tables[n]: the blocks' tables
extern "C" size_t nxz_built_tables_bytes(void) { return sizeof(nxzi::Built); }

// the tables of nb blocks (breqs: src, srclen, header_bit), then n requests for token boundaries in them (reqs:
// header_bit = which block of breqs); tables[nb]: the blocks' table bits; built: nb x nxz_built_tables_bytes() of DEVICE memory
extern "C" int nxz_launch_token_sync(const nxz_sync_req_t *breqs, uint32_t nb, nxz_batch_dht_t *tables, void *built,
				     const nxz_sync_req_t *reqs, uint32_t n, nxz_sync_res_t *res, hipStream_t stream)
{
	if (!n || !nb) return 0;
	hipLaunchKernelGGL(nxzi::block_tables_kernel, dim3(nb), dim3(64), 0, stream, breqs, tables, (nxzi::Built *)built);
	hipLaunchKernelGGL(nxzi::token_sync_kernel, dim3(n), dim3(64), 0, stream, reqs, res, (const nxzi::Built *)built);
	return (int)hipGetLastError();
}

// more of the same later on (nxz_pinflate.cpp, pieces that are decoded again): the tables of nb further blocks go behind the
// `first` that are there already, the requests name any of them
extern "C" int nxz_launch_token_sync_more(const nxz_sync_req_t *breqs, uint32_t nb, nxz_batch_dht_t *tables, void *built, uint32_t first,
					  const nxz_sync_req_t *reqs, uint32_t n, nxz_sync_res_t *res, hipStream_t stream)
{
	if (!n) return 0;
	if (nb) hipLaunchKernelGGL(nxzi::block_tables_kernel, dim3(nb), dim3(64), 0, stream, breqs, tables, (nxzi::Built *)built + first);
	hipLaunchKernelGGL(nxzi::token_sync_kernel, dim3(n), dim3(64), 0, stream, reqs, res, (const nxzi::Built *)built);
	return (int)hipGetLastError();
}

// pieces of many streams (nxz_inflate_cut.hip): the jobs `order` names, slots of 0xffffffff are none
extern "C" int nxz_launch_inflate_w16_order(const nxz_batch_job_t *jobs, size_t nslots, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io, const void *built,
					    const uint32_t *order, hipStream_t stream)
{
	if (!nslots) return 0;
	// (a few pieces, all of them resident at two per CU: the window of elements in LDS, where a match costs a lone wavefront a
	// fraction of the trip to device memory)
	static const unsigned lds_max = getenv("NXZ_INFLATE_W16_LDS_MAX") ? (unsigned)atoi(getenv("NXZ_INFLATE_W16_LDS_MAX")) : 512;
	if (nslots <= lds_max) hipLaunchKernelGGL((nxzi::inflate_kernel<false, true>), dim3((unsigned)nslots), dim3(64), 0, stream, jobs, results, dht_io, (const nxzi::Built *)built, order);
	else hipLaunchKernelGGL((nxzi::inflate_kernel<true, true>), dim3((unsigned)nslots), dim3(64), 0, stream, jobs, results, dht_io, (const nxzi::Built *)built, order);
	return (int)hipGetLastError();
}

// the jobs `order` names (slots of 0xffffffff are none), a wavefront each, the target as window; no checksums (the caller's)
extern "C" int nxz_launch_inflate_order_only(const nxz_batch_job_t *jobs, size_t nslots, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io,
					     const uint32_t *order, hipStream_t stream)
{
	if (!nslots) return 0;
	hipLaunchKernelGGL(nxzi::inflate_kernel<true>, dim3((unsigned)nslots), dim3(64), 0, stream, jobs, results, dht_io, (const nxzi::Built *)nullptr, order);
	return (int)hipGetLastError();
}

extern "C" int nxz_inflate_prof_set(unsigned long long *buf)
{
	return (int)hipMemcpyToSymbol(HIP_SYMBOL(nxzi::prof_buf), &buf, sizeof(buf));
}

// window_in_lds != 0: the 39.5 KiB variant (4 streams per CU, matches never leave LDS: the lower latency
// for one job or one round of jobs); 0: the window is the target itself (16 streams per CU).  Checksums
// by nxzl::cksum_kernel afterwards.
extern "C" int nxz_launch_inflate(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results,
				  nxz_batch_dht_t *dht_io, int window_in_lds, const uint32_t *order, hipStream_t stream)
{
	if (!n) return 0;
	if (window_in_lds) hipLaunchKernelGGL(nxzi::inflate_kernel<false>, dim3((unsigned)n), dim3(64), 0, stream, jobs, results, dht_io, (const nxzi::Built *)nullptr, order);
	else hipLaunchKernelGGL(nxzi::inflate_kernel<true>, dim3((unsigned)n), dim3(64), 0, stream, jobs, results, dht_io, (const nxzi::Built *)nullptr, order);
	int rc = (int)hipGetLastError();
	return rc ? rc : nxz_launch_cksum(jobs, n, results, stream);
}

// nxz_inflate_stream's pieces: 16-bit elements, references into the unknown 32 KiB in front as 0x8000 | index
extern "C" int nxz_launch_inflate_w16(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io, const void *built,
				      int few_and_even, hipStream_t stream)
{
	if (!n) return 0;
	// a few pieces of about one size (all of them resident at two per CU): the window in LDS, where a match costs one
	// wavefront a fraction of the trip to device memory.  (Not for the odd pieces that are decoded again: long
	// stored stretches are among them, which the other form copies four bytes per lane.)
	static const unsigned lds_max = getenv("NXZ_INFLATE_W16_LDS_MAX") ? (unsigned)atoi(getenv("NXZ_INFLATE_W16_LDS_MAX")) : 512;
	if (few_and_even && n <= lds_max) hipLaunchKernelGGL((nxzi::inflate_kernel<false, true>), dim3((unsigned)n), dim3(64), 0, stream, jobs, results, dht_io, (const nxzi::Built *)built, (const uint32_t *)nullptr);
	else hipLaunchKernelGGL((nxzi::inflate_kernel<true, true>), dim3((unsigned)n), dim3(64), 0, stream, jobs, results, dht_io, (const nxzi::Built *)built, (const uint32_t *)nullptr);
	return (int)hipGetLastError();
}
