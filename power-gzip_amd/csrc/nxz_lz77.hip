// nxz_lz77.hip -- LZ77 stage of the DEFLATE compression engine for MI355X (gfx950, wave64).
//
// First half of the POWER NX accelerator's COMPRESS function codes
// (GZIP_FC_COMPRESS_[RESUME_]{FHT,DHT}[_COUNT], issued at
// /root/reference lib/nx_deflate.c:1808,1841; contract inc_nx/nxu.h:286-616 and
// the consumer code lib/nx_deflate.c:969-1078).  One workgroup (1024 threads =
// 16 wavefronts, one per CU: the working set is 158 KiB of LDS) turns one
// sub-block of <= 64 KiB (history window included) into its LZ77 token sequence; the grid is one
// persistent workgroup per CU that draws job after job from a counter.
// The algorithm is the position-parallel LZ77 defined in oracle/nxz_lz77.c;
// this kernel must reproduce that restatement's tokens exactly.  What leaves the kernel
// (nxz_device.h, NXZ_TOK_*): a bitmap of the positions where a literal token starts, a bitmap of
// the positions where a match token starts, the (length, distance) records of the matches in
// parse order, the LZ symbol counts (286 + 30, the COUNT function codes' out_lzcount and the
// input of the table generator nxz_dhtgen.hip) and the checksums.  The entropy stage
// (nxz_encode.hip) makes the deflate block from them.
//
// Phases per sub-block (the block stays in LDS from load to the last tile):
//   load     coalesced 16 B/lane global loads of [window|block] into LDS
//   cksum    CRC-32 (64-byte slices, slice-by-4, GF(2) weights, XOR reduce) and Adler-32 (v_dot4)
//   seed     window positions -> head[] by LDS atomicMax (order free)
//   per 16 KiB tile:
//     hash   four positions per lane: 4-byte hash -> slot offsets, transposed per 512-position
//            piece for the chain wave; "byte equals predecessor" flag bitmap
//     chain  ONE wave walks the tile in 64-position steps: lookup head[], then atomicMax insert
//            (the only serial dependence of the algorithm), four pieces per loop trip, results
//            published half a piece late ...
//     match  ... while all waves take published pieces: verify the candidate, compare 8 bytes
//            (M1, four positions per lane); positions with 8 equal bytes are classified from a
//            queue: member of its successor's chain, or tail that compares on (M2); after a
//            barrier members read their length off the chain end and the distance-1 runs are
//            evaluated from the flag bitmap (M3)
//     parse  lane per 16-byte segment: speculative greedy/lazy walk -> exit X[s];
//            pointer jumping marks the chain of entered segments; entered segments re-walk
//            [entry, X[s]) and mark their tokens in the two bitmaps (register masks, one LDS
//            atomic per lane)
//     out    lane per 16 positions: match records to the job's record array at the rank a
//            workgroup prefix sum of the match counts gives; symbol counts by LDS atomics;
//            the two bitmaps as coalesced dwords
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include "nxz_device.h"
#include "nxz_dhtgen_dev.h"

// Diagnostic only (tools/phase_profile.py, a build with -DNXZ_LZ77_PROF: tools/build_variant.sh prof nxz_lz77.hip -DNXZ_LZ77_PROF):
// per-phase cycle sums of every workgroup's thread 0.  Compiled out of the product: the counters cost a dozen
// vector registers in a kernel that sits at its 128-register cap.
__device__ unsigned long long *nxz_lz77_prof_buf = nullptr;
#define NXZ_GLOBAL NXZ_GLOBAL_AS
#ifdef NXZ_LZ77_PROF
#define WPROF_BEGIN() unsigned long long wp_ = prof ? clock64() : 0
#define WPROF_END(idx) do { if (prof) { const unsigned long long now_ = clock64(); wacc[(idx) - 16] += now_ - wp_; wp_ = now_; } } while (0)
// (phase sums are collected in LDS and leave with one atomic per counter and job: the chain wave
// waits on its own vector-memory counter, an atomic in flight there would be timed as chain)
#define PROF(idx) do { if (prof) { if (t == 0) { unsigned long long now_ = clock64(); profacc[idx] += (uint32_t)(now_ - tprev); tprev = now_; } } } while (0)
#define PCOUNT(idx, n) do { if (prof && lane == 0) atomicAdd(&profacc[idx], (uint32_t)(n)); } while (0)
#ifdef NXZ_LZ77_PROF2
#define PC2(idx, n) do { if (prof && lane == 0) __hip_atomic_fetch_add(&prof[32 + (idx)], (unsigned long long)(n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define PC2(idx, n) do { } while (0)
#endif
#else
#define PC2(idx, n) do { } while (0)
#define PCOUNT(idx, n) do { } while (0)
#define WPROF_BEGIN() do { } while (0)
#define WPROF_END(idx) do { } while (0)
#define PROF(idx) do { } while (0)
#endif

namespace nxzl77 {

constexpr int NT = 1024;                 // threads per workgroup
constexpr int HBITS = 13;
constexpr uint32_t HSIZE = 1u << HBITS;
constexpr uint32_t PTILE = 16384;        // positions per parse tile
constexpr uint32_t PSEG = 16;            // positions per parse segment (one lane)
constexpr uint32_t NSEG = PTILE / PSEG;  // 1024
constexpr uint32_t ETILE = 2048;         // positions per encode step
constexpr uint32_t LAZY_MAX = 32;
constexpr uint32_t MAXMATCH = 258;
constexpr uint32_t WINDOW = 32768;
constexpr uint32_t OPEN2 = 24;           // bytes that the first level of M2 settles (8 from M1 + one 16-byte round trip)
constexpr uint32_t NOHASH = 0xFFFFu;
constexpr uint32_t SECOND_MIN_TOKENS = 3072; // oracle/nxz_lz77.c step 3b: tokens of the first tile below which the later tiles do without second bucket entries
constexpr uint32_t TEXT_HIGH_DIV = 16;       // oracle/nxz_lz77.c step 3c: fewer than 1 byte in 16 of the first tile with its top bit set = text
#ifndef NXZ_GROUP_ROUNDS
#define NXZ_GROUP_ROUNDS 24
#endif
#ifndef NXZ_GROUP_MIN
#define NXZ_GROUP_MIN 2
#endif
#ifndef NXZ_GROUP_GATE
#define NXZ_GROUP_GATE 3
#endif
constexpr uint32_t GROUP_ROUNDS = NXZ_GROUP_ROUNDS;   // distances looked at per batch of 64 long positions (open tails are extended a group at a time)
constexpr uint32_t GROUP_GATE = NXZ_GROUP_GATE;       // ... tails of a batch that must have a neighbour in the queue at their distance before the wave looks for groups at all
constexpr uint32_t GROUP_MIN = NXZ_GROUP_MIN;         // ... tails of one distance that are worth a trip of the whole wave
// device scratch of a workgroup (16-bit units): carries a tile's second bucket entries from the chain
// wave to the match waves (+ room for a piece past the end)
constexpr uint32_t C2_STRIDE = PTILE + 1024;

// ---- LDS carve (bytes) ----
constexpr uint32_t OFF_IN    = 0;                        // 65536 + 32 pad
constexpr uint32_t OFF_HEAD  = 65536 + 32;               // 8192 x u32
constexpr uint32_t OFF_CAND  = OFF_HEAD + HSIZE * 4 + 256; // 16384 x u16 (head[HSIZE + lane]: one dummy slot per lane of the chain wave)
constexpr uint32_t OFF_MLEN  = OFF_CAND + PTILE * 2;     // 16384 x u8
constexpr uint32_t OFF_SBITS = OFF_MLEN + PTILE;         // 512 x u32   (aliased: MARK u8[1024] during chain marking; match-token bitmap of the final walk)
constexpr uint32_t OFF_MBITS = OFF_SBITS + PTILE / 8;    // 512 x u32   (aliased: JUMP u16[1024]; literal-token bitmap of the final walk)
constexpr uint32_t OFF_X     = OFF_MBITS + PTILE / 8;    // 1024 x u16
constexpr uint32_t OFF_ENTRY = OFF_X + NSEG * 2;         // 1024 x u16
constexpr uint32_t OFF_BITS  = OFF_ENTRY + NSEG * 2;     // the CRC tables (4 KiB) before the first tile, match-phase queues later
// The match phase's queues, per wave: LQ_CAP long positions (a batch goes to level 1 of M2 when LQ_MIN are waiting: a unit adds
// up to 64 at a time) and XQ_CAP positions that are still open after OPEN2 bytes (level 2 when XQ_MIN are waiting).  The
// fuller a batch, the fewer trips: with 96 / 32 and 80 / 16 (round 4) level 1 ran 70 % full and level 2 with 16 lanes of 64.
#ifndef NXZ_LQ_CAP
#define NXZ_LQ_CAP 128
#define NXZ_LQ_MIN 64
#endif
#ifndef NXZ_XQ_CAP
#define NXZ_XQ_CAP 120
#define NXZ_XQ_MIN 56
#endif
constexpr uint32_t LQ_CAP = NXZ_LQ_CAP, LQ_MIN = NXZ_LQ_MIN, XQ_CAP = NXZ_XQ_CAP, XQ_MIN = NXZ_XQ_MIN;
static_assert(LQ_MIN + 64 <= LQ_CAP && XQ_MIN + 64 <= XQ_CAP && LQ_CAP % 8 == 0 && XQ_CAP % 8 == 0, "a batch of 64 must fit behind what waits");
constexpr uint32_t BITS_BYTES = 16 * (LQ_CAP + XQ_CAP) * 2 + 2096 - NSEG * 4 < ETILE * 2 + 64 ? ETILE * 2 + 64 : 16 * (LQ_CAP + XQ_CAP) * 2 + 2096 - NSEG * 4;
// during the match phase the window region holds: 16 x LQ_CAP queue entries (long positions), 16 x XQ_CAP
// (positions still open after OPEN2 bytes) and the e-flag bitmap (PTILE + 288 bits)
constexpr uint32_t OFF_LQ    = OFF_X;
constexpr uint32_t OFF_DQ    = OFF_LQ;                   // after the match phase: the segments the distance-1 pass has to look at (1024 x u16)
constexpr uint32_t OFF_XQ    = OFF_LQ + 16 * LQ_CAP * 2;
constexpr uint32_t OFF_EB    = OFF_XQ + 16 * XQ_CAP * 2;     // 2096 bytes; bit EBO + i = flag of tile position i
constexpr uint32_t EBO       = 32;                       // flags of the 32 positions in front of the tile come first
static_assert(OFF_EB % 16 == 0 && OFF_EB + 2096 <= OFF_BITS + BITS_BYTES, "match-phase carve");
constexpr uint32_t OFF_SCAN  = OFF_BITS + BITS_BYTES;    // 64 x u32
constexpr uint32_t OFF_HIST  = OFF_SCAN + 256;           // 316 x u32
constexpr uint32_t OFF_MISC  = OFF_HIST + 316 * 4;       // 16 x u32
constexpr uint32_t OFF_PROF  = OFF_MISC + 64;            // 20 x u32 (diagnostic)
constexpr uint32_t LDS_BYTES = OFF_PROF + 80;
static_assert(LDS_BYTES <= 163840, "LDS budget");
static_assert(OFF_HEAD % 16 == 0 && OFF_BITS % 16 == 0 && OFF_SCAN % 16 == 0, "alignment");

enum { M_NREC = 0, M_TOK0 = 1, M_HIGH = 2, M_DQ = 3, M_PROGRESS = 4, M_TICKET = 5, M_DEFER = 6, M_NEXT = 7, M_KEEP = 8, M_DEFER2 = 9, M_PREV = 12, M_JOBNO = 13 };
static_assert(M_TICKET == M_PROGRESS + 1 && M_DEFER == M_PROGRESS + 2, "cleared together");

__device__ __forceinline__ uint32_t lds_ld32(const uint32_t *inw, uint32_t r)
{
	// unaligned little-endian 32-bit load from the LDS input image
	uint32_t a = inw[r >> 2], b = inw[(r >> 2) + 1];
	return __builtin_amdgcn_alignbyte(b, a, r & 3);
}

__device__ __forceinline__ uint32_t hash4(uint32_t v) { return (v * 0x9E3779B1u) >> (32 - HBITS); }

// number of equal bytes (0..16) of the strings at a and b, compared over 16 bytes with one LDS round trip
__device__ __forceinline__ uint32_t equal16(const uint32_t *inw, uint32_t a, uint32_t b)
{
	const uint32_t aw = a >> 2, bw = b >> 2, as = a & 3, bs = b & 3;
	const uint32_t a0 = inw[aw], a1 = inw[aw + 1], a2 = inw[aw + 2], a3 = inw[aw + 3], a4 = inw[aw + 4];
	const uint32_t b0 = inw[bw], b1 = inw[bw + 1], b2 = inw[bw + 2], b3 = inw[bw + 3], b4 = inw[bw + 4];
	const uint32_t x0 = __builtin_amdgcn_alignbyte(a1, a0, as) ^ __builtin_amdgcn_alignbyte(b1, b0, bs);
	const uint32_t x1 = __builtin_amdgcn_alignbyte(a2, a1, as) ^ __builtin_amdgcn_alignbyte(b2, b1, bs);
	const uint32_t x2 = __builtin_amdgcn_alignbyte(a3, a2, as) ^ __builtin_amdgcn_alignbyte(b3, b2, bs);
	const uint32_t x3 = __builtin_amdgcn_alignbyte(a4, a3, as) ^ __builtin_amdgcn_alignbyte(b4, b3, bs);
	uint32_t n = 16;
	if (x3) n = 12 + ((uint32_t)__builtin_ctz(x3) >> 3);
	if (x2) n = 8 + ((uint32_t)__builtin_ctz(x2) >> 3);
	if (x1) n = 4 + ((uint32_t)__builtin_ctz(x1) >> 3);
	if (x0) n = (uint32_t)__builtin_ctz(x0) >> 3;
	return n;
}

// GF(2)[x] multiply modulo the reflected CRC-32 polynomial
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

// x^(8*64*k) mod P for k = 0..1023 (compile-time): what a 64-byte slice that is followed by k
// more slices has to be multiplied with.
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
// x^(8 r) mod P for r = 0..64
struct Pow8Tab { uint32_t v[65]; };
constexpr Pow8Tab make_pow8()
{
	Pow8Tab p{};
	p.v[0] = 0x80000000u;
	for (int i = 1; i <= 64; i++) p.v[i] = cgf_mul(p.v[i - 1], 0x00800000u);
	return p;
}
__device__ const Pow8Tab CRC_POW8 = make_pow8();

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		uint32_t u = __shfl_up(v, o, 64);
		if (lane >= o) v += u;
	}
	return v;
}

// ---- fixed-Huffman output inside the LZ77 kernel (FUSED: function codes 0x00 / 0x08) ----
// The fixed code needs no table and no second pass: a tile's tokens go from LDS straight into the job's target
// (round 2 wrote them to device scratch -- 56 KB per block -- and a second kernel read them and the source again:
// 12 % of the step and 2.6 x the bytes, for a split that only dynamic tables require).
// The tokens that start in four consecutive positions as one bit string (<= 96 bits: a match at the first position
// and one at the fourth): RFC 1951 3.2.6 codes by arithmetic, bit-reversed; bit for bit oracle/nxz_lz77.c put_tokens.
struct Quad { uint32_t a0, a1, a2, nb; };
__device__ __forceinline__ void fx_match(uint32_t l3, uint32_t d, uint64_t &mv, uint32_t &mn)
{
	uint32_t le = l3 < 8 ? 0 : (29 - (uint32_t)__builtin_clz(l3 | 8));
	const uint32_t ls = l3 == 255 ? 28 : (le << 2) + (l3 >> le);
	if (l3 == 255) le = 0;
	const uint32_t de = d < 4 ? 0 : (30 - (uint32_t)__builtin_clz(d | 4));
	const uint32_t ds = d < 4 ? d : 2 * de + 2 + ((d >> de) & 1);
	// symbol 257 + ls: 7 bits (code ls + 1) below 280, else 8 bits (0xC0 + ls - 23); distance: 5 bits
	const bool l8 = ls >= 23;
	const uint32_t ll = l8 ? 8 : 7;
	const uint32_t lc = __builtin_bitreverse32(l8 ? 0xC0u + ls - 23 : ls + 1) >> (l8 ? 24 : 25);
	const uint32_t dc = __builtin_bitreverse32(ds) >> 27;
	const uint32_t lo = lc | ((l3 & ((1u << le) - 1)) << ll);                // <= 13 bits
	const uint32_t hi = dc | ((d & ((1u << de) - 1)) << 5);                  // <= 18 bits
	mv = (uint64_t)lo | ((uint64_t)hi << (ll + le));
	mn = ll + le + 5 + de;
}
// b: the four bytes; lit4 / tok4: which of the positions start a literal / a match; m4: the stored lengths
// (len - 3, a byte each); c01, c23: the distances - 1 (16 bits each)
__device__ __forceinline__ Quad fx_quad(uint32_t b, uint32_t lit4, uint32_t tok4, uint32_t m4, uint32_t c01, uint32_t c23)
{
	uint64_t v[4];
	uint32_t nbk[4];
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const uint32_t x = (b >> (8 * k)) & 0xff;
		const bool hi = x >= 144;
		v[k] = __builtin_bitreverse32(x + (hi ? 0x100u : 0x30u)) >> (hi ? 23 : 24);
		nbk[k] = (lit4 >> k) & 1 ? (hi ? 9 : 8) : 0;
	}
	// matches: at most two start in four positions (they are at least three bytes long), the second one only at
	// the fourth position behind one at the first
	if (__ballot(tok4 != 0)) {
		const uint32_t k1 = (uint32_t)__builtin_ctz(tok4 | 16);                 // 4 = none
		const uint32_t kk = k1 & 3;
		uint64_t mv; uint32_t mn;
		fx_match((m4 >> (8 * kk)) & 0xff, ((kk & 2 ? c23 : c01) >> (16 * (kk & 1))) & 0xffff, mv, mn);
#pragma unroll
		for (int k = 0; k < 4; k++) if (k1 == (uint32_t)k) { v[k] = mv; nbk[k] = mn; }
		if (__ballot(tok4 == 9)) {
			fx_match(m4 >> 24, c23 >> 16, mv, mn);
			if (tok4 == 9) { v[3] = mv; nbk[3] = mn; }
		}
	}
	// one string: token k at the sum of the lengths before it
	uint64_t lo = nbk[0] ? v[0] : 0, hi = 0;
	uint32_t off = nbk[0];
#pragma unroll
	for (int k = 1; k < 4; k++) {
		const uint64_t x = nbk[k] ? v[k] : 0;
		// off <= 31 + 9 + 9 here, x < 2^31
		if (off < 64) { lo |= x << off; hi |= off ? x >> (64 - off) : 0; }
		else hi |= x << (off - 64);
		off += nbk[k];
	}
	Quad q;
	q.a0 = (uint32_t)lo; q.a1 = (uint32_t)(lo >> 32); q.a2 = (uint32_t)hi; q.nb = off;
	return q;
}
// ORs a quad's bits into the window at bit position bitpos
__device__ __forceinline__ void fx_emit(uint32_t *w, const Quad &q, uint32_t bitpos)
{
	if (!q.nb) return;
	const uint32_t sh = bitpos & 31, wi = bitpos >> 5, endw = (sh + q.nb + 31) >> 5;   // dwords touched: 1..4
	const uint64_t s0 = (uint64_t)q.a0 << sh, s1 = (uint64_t)q.a1 << sh, s2 = (uint64_t)q.a2 << sh;
	atomicOr(&w[wi], (uint32_t)s0);
	if (endw > 1) atomicOr(&w[wi + 1], (uint32_t)(s0 >> 32) | (uint32_t)s1);
	if (endw > 2) atomicOr(&w[wi + 2], (uint32_t)(s1 >> 32) | (uint32_t)s2);
	if (endw > 3) atomicOr(&w[wi + 3], (uint32_t)(s2 >> 32));
}


// ---- the table and the Huffman coding inside the LZ77 kernel (GEN: the additive DHTGEN function codes) ----
// One job = LZ77 + table + encode in one engine pass (/root/reference lib/nx_deflate.c:1841; the table as lib/nx_dhtgen.c:945-1034
// makes it).  The table of a block needs the counts of the whole block, and making it is a chain of some 8000 dependent
// instructions of ONE wavefront -- 40 K cycles with fifteen others waiting, a tenth of the block's time.  So the work is
// pipelined over the jobs a workgroup takes: behind the last tile of job k wavefront 0 makes the table of job k (from the
// histogram in LDS, into the workgroup's table slot in device scratch) WHILE wavefronts 1-15 encode job k - 1 (its tokens from
// the workgroup's other token slot, its literals from the source, its table from the other table slot).  A workgroup's last
// job is encoded behind its loop.  Tokens and tables never leave the L2: two slots of each per workgroup instead of one per job
// of a 65536-job chunk (6.6 GiB), no launch of nxz_dhtgen.hip / nxz_encode.hip, whose code this is (bit for bit the same blocks).
// All sixteen wavefronts must pass every workgroup barrier: wavefront 0 takes the encoders' barriers -- their number follows
// from the block's size -- at the turning points of the table generator's loops, and the rest of them when it is done.
namespace gen {
constexpr uint32_t ENT = 960;                              // encoding threads (wavefronts 1-15)
constexpr uint32_t RPOS = ENT * 8;                         // positions per round: 8 per lane, two quads of 4
constexpr uint32_t HDR_WORDS = 74;
constexpr uint32_t WW = ENT * 144 / 32 + HDR_WORDS + 2;    // window words: three 48-bit match tokens in 8 positions per lane
constexpr uint32_t RECMAX = RPOS / 3 + 6;
// LDS of the post phase (head[], cand[], mlen[] are free behind the last tile)
constexpr uint32_t P_LL = OFF_HEAD, P_D = P_LL + 288 * 4, P_WIN = P_D + 32 * 4, P_REC = P_WIN + 2 * (WW + 2) * 4,
		   P_RANK = P_REC + 2 * (RECMAX + 2) * 4, P_WSUM = P_RANK + 4104, P_DHT = (P_WSUM + 2 * 16 * 4 + 15) & ~15u, P_END = P_DHT + nxzd::WAVE_LDS;
static_assert(P_WIN % 16 == 0 && P_END <= OFF_SBITS, "post-phase carve");
__host__ __device__ constexpr size_t table_off(uint32_t grid) { return (size_t)grid * 2 * NXZ_TOK_STRIDE; }

struct Credit {
	static constexpr bool none = false;
	uint32_t n;
	__device__ __forceinline__ void operator()() { if (n) { __builtin_amdgcn_s_barrier(); n--; } }
};
// barriers the encoders of a block of n bytes pass (encode() below: three while they set up, two a round, one behind the
// rounds, two more for a block without any round)
__device__ __forceinline__ uint32_t encode_barriers(uint32_t n) { const uint32_t R = (n + RPOS - 1) / RPOS; return 3 + 2 * R + 1 + (R ? 0 : 2); }

struct Quad { uint32_t a0, a1, a2, nb; };
// (nxz_encode.hip encode_quad without the missing-code checks: the table has a code for every symbol the block uses)
__device__ __forceinline__ Quad encode_quad(const uint32_t *lltab, const uint32_t *dtab, const uint32_t *rec, uint32_t &ri, uint32_t b, uint32_t lit4, uint32_t tok4)
{
	const uint32_t e0 = lltab[b & 0xff], e1 = lltab[(b >> 8) & 0xff], e2 = lltab[(b >> 16) & 0xff], e3 = lltab[b >> 24];
	uint64_t v[4] = { e0 & 0xffff, e1 & 0xffff, e2 & 0xffff, e3 & 0xffff };
	uint32_t nbk[4] = { (lit4 & 1) ? e0 >> 16 : 0, (lit4 & 2) ? e1 >> 16 : 0, (lit4 & 4) ? e2 >> 16 : 0, (lit4 & 8) ? e3 >> 16 : 0 };
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
			const uint32_t lo = (lt & 0xffff) | ((l3 & ((1u << le) - 1)) << ll);
			const uint32_t hi = (dt & 0xffff) | ((d & ((1u << de) - 1)) << dl);
			mv = (uint64_t)lo | ((uint64_t)hi << (ll + le));
			mn = ll + le + dl + de;
		};
		const uint32_t k1 = (uint32_t)__builtin_ctz(tok4 | 16);
		uint64_t mv; uint32_t mn;
		one(rec[ri], mv, mn);
		ri += tok4 ? 1 : 0;
#pragma unroll
		for (int k = 0; k < 4; k++) if (k1 == (uint32_t)k) { v[k] = mv; nbk[k] = mn; }
		if (__ballot(tok4 == 9)) {
			one(rec[ri], mv, mn);
			if (tok4 == 9) { v[3] = mv; nbk[3] = mn; ri++; }
		}
	}
	uint64_t lo = nbk[0] ? v[0] : 0, hi = 0;
	uint32_t off = nbk[0];
#pragma unroll
	for (int k = 1; k < 4; k++) {
		const uint64_t x = nbk[k] ? v[k] : 0;
		if (off < 64) { lo |= x << off; hi |= off ? x >> (64 - off) : 0; }
		else hi |= x << (off - 64);
		off += nbk[k];
	}
	Quad q;
	q.a0 = (uint32_t)lo; q.a1 = (uint32_t)(lo >> 32); q.a2 = (uint32_t)hi; q.nb = off;
	return q;
}
__device__ __forceinline__ void emit_quad(uint32_t *w, const Quad &q, uint32_t bitpos)
{
	if (!q.nb) return;
	const uint32_t sh = bitpos & 31, wi = bitpos >> 5, endw = (sh + q.nb + 31) >> 5;
	const uint64_t s0 = (uint64_t)q.a0 << sh, s1 = (uint64_t)q.a1 << sh, s2 = (uint64_t)q.a2 << sh;
	atomicOr(&w[wi], (uint32_t)s0);
	if (endw > 1) atomicOr(&w[wi + 1], (uint32_t)(s0 >> 32) | (uint32_t)s1);
	if (endw > 2) atomicOr(&w[wi + 2], (uint32_t)(s1 >> 32) | (uint32_t)s2);
	if (endw > 3) atomicOr(&w[wi + 3], (uint32_t)(s2 >> 32));
}

// wavefronts 1-15 (te = 0 .. 959): the block of job `bid` from its tokens at tk and its table tb_ (nxz_encode.hip's
// encode_kernel<true, false>, 960 threads wide; EVERY __syncthreads here is counted in encode_barriers)
__device__ __forceinline__ void encode(uint8_t *lds, const nxz_batch_job_t *__restrict__ jobs, uint32_t bid, const NXZ_GLOBAL uint8_t *tk,
				       const NXZ_GLOBAL nxz_dht_prepared_t *tb, nxz_batch_result_t *__restrict__ results, const int te)
{
	typedef uint32_t v4u __attribute__((ext_vector_type(4)));
	typedef uint32_t v2u __attribute__((ext_vector_type(2)));
	uint32_t *lltab = (uint32_t *)(lds + P_LL), *dtab = (uint32_t *)(lds + P_D);
	uint32_t *win0 = (uint32_t *)(lds + P_WIN);                      // win[par] = win0 + par * (WW + 2)
	uint32_t *rec0 = (uint32_t *)(lds + P_REC);                      // recbuf[par] = rec0 + par * (RECMAX + 2)
	uint16_t *rankpre = (uint16_t *)(lds + P_RANK);
	uint32_t *wsum = (uint32_t *)(lds + P_WSUM);                     // wsum[par * 16 + wave]
	const int lane = te & 63, wave = te >> 6;
	const nxz_batch_job_t job = jobs[bid];
	const uint32_t total = job.src_len;
	const uint32_t h = job.hist_len < total ? job.hist_len : total;
	const uint32_t n = total - h;
	const NXZ_GLOBAL uint8_t *src = (const NXZ_GLOBAL uint8_t *)job.src + h;
	const NXZ_GLOBAL uint32_t *litb = (const NXZ_GLOBAL uint32_t *)(tk + NXZ_TOK_LITBITS);
	const NXZ_GLOBAL uint32_t *tokb = (const NXZ_GLOBAL uint32_t *)(tk + NXZ_TOK_MATCHBITS);
	const NXZ_GLOBAL uint32_t *recs = (const NXZ_GLOBAL uint32_t *)(tk + NXZ_TOK_RECORDS);
	NXZ_GLOBAL uint32_t *dstw = (NXZ_GLOBAL uint32_t *)job.dst;
	const uint32_t cap_words = job.dst_cap >> 2;
	const uint32_t nwords = (n + 31) >> 5;
	struct Fetch { v2u bytes; uint32_t litw, tokw, rec[3]; };
	auto fetch = [&](uint32_t r0, Fetch &f) {
		const uint32_t p0 = r0 + 8 * (uint32_t)te;
		f.bytes = (v2u){ 0, 0 }; f.litw = 0; f.tokw = 0;
		if (p0 < n) {
			f.litw = litb[p0 >> 5];
			f.tokw = tokb[p0 >> 5];
			if (p0 + 8 <= n) f.bytes = *(const NXZ_GLOBAL v2u *)(src + p0);
			else {
				uint64_t v = 0;
				for (uint32_t i = 0; i < 8; i++) if (p0 + i < n) v |= (uint64_t)src[p0 + i] << (8 * i);
				f.bytes = (v2u){ (uint32_t)v, (uint32_t)(v >> 32) };
			}
		}
	};
	auto fetch_recs = [&](uint32_t r0, Fetch &f) {
		const uint32_t ra = rankpre[r0 >> 5], rb = rankpre[(r0 + RPOS) >> 5 < 2048 ? (r0 + RPOS) >> 5 : 2048];
#pragma unroll
		for (int j = 0; j < 3; j++) f.rec[j] = ra + (uint32_t)te + ENT * j < rb ? recs[ra + (uint32_t)te + ENT * j] : 0;
	};
	Fetch nx;
	fetch(0, nx);
	for (uint32_t i = te; i < 2 * (WW + 2); i += ENT) win0[i] = 0;
	{
		// the matches in front of every 32 positions: the first 256 threads, eight bitmap words each
		uint32_t c[8], s = 0, incl = 0;
		if (te < 256) {
			const NXZ_GLOBAL v4u *tw = (const NXZ_GLOBAL v4u *)tokb;
			v4u a = { 0, 0, 0, 0 }, b = a;
			if (8u * te < nwords) { a = tw[2 * te]; b = tw[2 * te + 1]; }
			const uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
			for (int k = 0; k < 8; k++) { c[k] = s; s += 8u * te + k < nwords ? (uint32_t)__popc(w[k]) : 0; }
			incl = s;
			for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
			if (lane == 63) wsum[wave] = incl;
		}
		for (uint32_t i = te; i < 288; i += ENT) lltab[i] = tb->ll[i];
		if (te < 32) dtab[te] = tb->d[te];
		__syncthreads();                                              // 1
		if (te < 256) {
			uint32_t off = incl - s;
			for (int k = 0; k < wave; k++) off += wsum[k];
#pragma unroll
			for (int k = 0; k < 8; k++) rankpre[8 * te + k] = (uint16_t)(off + c[k]);
			if (te == 255) rankpre[2048] = (uint16_t)(off + s);
		}
	}
	uint32_t base_bits;
	{
		const uint32_t hb = tb->dhtlen + 3;                          // BFINAL = 1 as emitted (the host rewrites it, lib/nx_deflate.c:158), BTYPE = 10
		const uint32_t nw = (hb + 31) >> 5;
		for (uint32_t i = te; i < nw && i < HDR_WORDS; i += ENT) {
			const uint32_t cur = i < 74 ? tb->dhtw[i] : 0, prev = i ? tb->dhtw[i - 1] : 0;
			uint32_t w = (cur << 3) | (i ? prev >> 29 : 5u);
			if (i == (hb >> 5)) w &= (1u << (hb & 31)) - 1;
			win0[i] = w;
		}
		base_bits = hb;
	}
	__syncthreads();                                                  // 2
	{
		fetch_recs(0, nx);
#pragma unroll
		for (int j = 0; j < 3; j++) if ((uint32_t)te + ENT * j < RECMAX) rec0[(uint32_t)te + ENT * j] = nx.rec[j];
	}
	__syncthreads();                                                  // 3
	uint32_t wordbase = 0, par = 0;
	for (uint32_t r0 = 0; r0 < n; r0 += RPOS, par ^= 1) {
		const Fetch k = nx;
		const bool more = r0 + RPOS < n;
		if (more) { fetch(r0 + RPOS, nx); fetch_recs(r0 + RPOS, nx); }
		uint32_t *w_ = win0 + par * (WW + 2);
		const uint32_t *rb_ = rec0 + par * (RECMAX + 2);
		const uint32_t p0 = r0 + 8 * (uint32_t)te;
		const uint32_t sh8 = p0 & 24;
		const uint32_t lit8 = (k.litw >> sh8) & 0xff, tok8 = (k.tokw >> sh8) & 0xff;
		Quad q0{0, 0, 0, 0}, q1{0, 0, 0, 0};
		if (__ballot((lit8 | tok8) != 0)) {
			uint32_t ri = (uint32_t)rankpre[p0 >> 5 < 2048 ? p0 >> 5 : 2048] + (uint32_t)__popc(k.tokw & ((1u << (p0 & 31)) - 1)) - (uint32_t)rankpre[r0 >> 5];
			if (p0 >= n) ri = 0;
			q0 = encode_quad(lltab, dtab, rb_, ri, k.bytes.x, lit8 & 15, tok8 & 15);
			q1 = encode_quad(lltab, dtab, rb_, ri, k.bytes.y, lit8 >> 4, tok8 >> 4);
		}
		const uint32_t nbits = q0.nb + q1.nb;
		uint32_t incl = nbits;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
		if (lane == 63) wsum[par * 16 + wave] = incl;
		__syncthreads();                                              // 2 a round: a
		uint32_t bitpos = base_bits + incl - nbits, roundbits = 0;
#pragma unroll
		for (int w = 0; w < 15; w++) { const uint32_t s = wsum[par * 16 + w]; if (w < wave) bitpos += s; roundbits += s; }
		emit_quad(w_, q0, bitpos);
		emit_quad(w_, q1, bitpos + q0.nb);
		if (more) {
			uint32_t *rn = rec0 + (par ^ 1) * (RECMAX + 2);
#pragma unroll
			for (int j = 0; j < 3; j++) if ((uint32_t)te + ENT * j < RECMAX) rn[(uint32_t)te + ENT * j] = nx.rec[j];
		}
		__syncthreads();                                              // 2 a round: b
		const uint32_t tot = base_bits + roundbits, nfull = tot >> 5;
		for (uint32_t i = te; i < nfull; i += ENT) {
			if (wordbase + i < cap_words) dstw[wordbase + i] = w_[i];
			w_[i] = 0;
		}
		if (te == 0) {
			const uint32_t keep = w_[nfull];
			if (keep) atomicOr(&win0[(par ^ 1) * (WW + 2)], keep);
			w_[nfull] = 0;
		}
		wordbase += nfull;
		base_bits = tot & 31;
	}
	__syncthreads();                                                  // 1 behind the rounds
	uint32_t *w_ = win0 + par * (WW + 2);
	if (n == 0) {
		// a job without any round still has its header in the window
		const uint32_t nfull = base_bits >> 5;
		for (uint32_t i = te; i < nfull; i += ENT)
			if (wordbase + i < cap_words) dstw[wordbase + i] = w_[i];
		const uint32_t keep = w_[nfull];
		__syncthreads();                                              // 2 more for a block without a round: a
		if (te == 0) w_[0] = keep;
		wordbase += nfull;
		base_bits &= 31;
		__syncthreads();                                              // ... b
	}
	if (te == 0) {
		const uint32_t lt = lltab[256];
		uint32_t cc = 0;
		if (wordbase > cap_words) cc = NXZ_CC_TARGET_SPACE;
		if ((lt >> 16) == 0) cc = NXZ_CC_MISSING_CODE;
		if (tb->status) cc = NXZ_CC_INVALID_DHT;
		const uint64_t acc = (uint64_t)w_[0] | ((uint64_t)(lt & 0xffff) << base_bits);
		const uint32_t bits = base_bits + (lt >> 16);
		const uint64_t totbits = (uint64_t)wordbase * 32 + bits;
		const uint32_t tpbc = (uint32_t)((totbits + 7) >> 3);
		if (tpbc > job.dst_cap) cc = cc ? cc : NXZ_CC_TARGET_SPACE;
		if (cc != NXZ_CC_TARGET_SPACE && (uint64_t)wordbase * 4 + (bits + 7) / 8 <= job.dst_cap) {
			NXZ_GLOBAL uint8_t *o = (NXZ_GLOBAL uint8_t *)job.dst + (size_t)wordbase * 4;
			for (uint32_t b = 0; b < (bits + 7) / 8; b++) o[b] = (uint8_t)(acc >> (8 * b));
		}
		if (cc == 0 && tpbc > total) cc = NXZ_CC_TPBC_GT_SPBC;
		nxz_batch_result_t *r = results + bid;
		r->cc = cc;
		r->tpbc = cc == NXZ_CC_TARGET_SPACE ? 0 : tpbc;
		r->tebc = (uint32_t)(totbits & 7);
		r->sfbt = 0;
	}
}

// behind a job's last tile (and once more behind a workgroup's last job): see above
__device__ __forceinline__ void post(uint8_t *lds, const nxz_batch_job_t *__restrict__ jobs, NXZ_GLOBAL uint8_t *scratch, nxz_batch_result_t *__restrict__ results,
				     bool have_cur, uint32_t cur_slot, uint32_t prev_plus1, uint32_t prev_slot, const int t)
{
	const NXZ_GLOBAL uint8_t *tabs = scratch + table_off(gridDim.x);
	const bool have_prev = prev_plus1 != 0;
	uint32_t nprev = 0;
	if (have_prev) {
		const uint32_t sl = jobs[prev_plus1 - 1].src_len, hl = jobs[prev_plus1 - 1].hist_len;
		nprev = sl - (hl < sl ? hl : sl);
	}
	nprev = (uint32_t)__builtin_amdgcn_readfirstlane((int)nprev);
	if (t < 64) {
		Credit cr{ have_prev ? encode_barriers(nprev) : 0u };
		if (have_cur) {
			uint32_t *hist = (uint32_t *)(lds + OFF_HIST);
			if (t == 0) hist[256] = 1;                               // the callers count EOB once (lib/nx_dht.c:189-195)
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_s_setprio(3);                           // (the one wavefront everything may end up waiting for)
			nxzd::dhtgen_wave(lds + P_DHT, (const uint32_t *)hist, (nxz_dht_prepared_t *)(tabs + ((size_t)blockIdx.x * 2 + cur_slot) * sizeof(nxz_dht_prepared_t)),
					  (nxz_batch_dht_t *)nullptr, t, cr);
			__builtin_amdgcn_s_setprio(0);
		}
		while (cr.n) cr();
	} else if (have_prev) {
		encode(lds, jobs, prev_plus1 - 1, scratch + ((size_t)blockIdx.x * 2 + prev_slot) * NXZ_TOK_STRIDE,
		       (const NXZ_GLOBAL nxz_dht_prepared_t *)(tabs + ((size_t)blockIdx.x * 2 + prev_slot) * sizeof(nxz_dht_prepared_t)), results, t - 64);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__syncthreads();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
} // namespace gen

template <bool COUNT, bool FUSED = false, bool GEN = false>
__global__ __launch_bounds__(NT) void lz77_kernel(const nxz_batch_job_t *__restrict__ jobs,
						  uint8_t *__restrict__ tokens, uint16_t *__restrict__ cand2,
						  nxz_batch_result_t *__restrict__ results,
						  uint32_t *__restrict__ counts, uint32_t njobs, uint32_t *__restrict__ next_job)
{
	// (a static array: its address is a compile-time constant, 0; with `extern __shared__` every LDS address of the
	// kernel carried one more vector add -- of a link-time zero --, 117 of them)
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
	uint32_t *inw = (uint32_t *)(lds + OFF_IN);
	uint32_t *head = (uint32_t *)(lds + OFF_HEAD);
	uint16_t *cand = (uint16_t *)(lds + OFF_CAND);
	uint8_t *mlen = lds + OFF_MLEN;
	uint16_t *X = (uint16_t *)(lds + OFF_X);
	uint16_t *entry = (uint16_t *)(lds + OFF_ENTRY);
	uint32_t *sbits = (uint32_t *)(lds + OFF_SBITS);
	uint32_t *mbits = (uint32_t *)(lds + OFF_MBITS);
	uint32_t *tokbits = sbits;                           // match-token bitmap of the final walk (SBITS is free by then)
	uint32_t *litbits = mbits;                           // literal-token bitmap (JUMP is dead by then)
	uint8_t *mark = lds + OFF_SBITS;
	uint16_t *jump = (uint16_t *)(lds + OFF_MBITS);
	uint32_t *bitbuf = (uint32_t *)(lds + OFF_BITS);
	uint32_t *scan = (uint32_t *)(lds + OFF_SCAN);
	uint32_t *hist = (uint32_t *)(lds + OFF_HIST);
	uint32_t *misc = (uint32_t *)(lds + OFF_MISC);
#ifdef NXZ_LZ77_PROF
	uint32_t *profacc = (uint32_t *)(lds + OFF_PROF);
#endif

#ifdef NXZ_LZ77_PROF
	NXZ_GLOBAL unsigned long long *prof = (NXZ_GLOBAL unsigned long long *)nxz_lz77_prof_buf;
#endif
	// The grid is one workgroup per CU (the LDS image allows no more); a workgroup starts with job
	// blockIdx.x and then draws further jobs from a counter (or strides by the grid without one), so
	// nothing waits for a dispatch in between and slow jobs do not pile up in one place.  The draw
	// for the job after this one is issued at once; its latency hides behind the work.
	if constexpr (GEN) {
		if (threadIdx.x == 0) { misc[M_PREV] = 0; misc[M_JOBNO] = 0; }
		__syncthreads();
	}
	for (uint32_t bid = blockIdx.x; bid < njobs;) {
	if (next_job && threadIdx.x == 0) misc[M_NEXT] = gridDim.x + atomicAdd(next_job, 1u);   // parked in LDS, not in a register
	// (the thread index is made opaque per job: what is derived from it would otherwise be hoisted out
	// of this loop and kept -- spilled -- for the whole kernel)
	int tid_ = threadIdx.x;
	asm volatile("" : "+v"(tid_));
	const int t = tid_, lane = t & 63, wave = t >> 6;
#ifdef NXZ_LZ77_PROF
	unsigned long long tprev = prof ? clock64() : 0;
	unsigned long long wacc[4] = { 0, 0, 0, 0 };               // diagnostic: this wave's cycles in sections of the match phase
#endif
	const nxz_batch_job_t job = jobs[bid];
	const uint32_t total = job.src_len;                  // window + block
	const uint32_t h = job.hist_len < total ? job.hist_len : total;
	const uint32_t n = total - h;
	const uint32_t end = total;
	const NXZ_GLOBAL uint8_t *src = (const NXZ_GLOBAL uint8_t *)job.src;
	// CRC weights (constant memory), fetched up front so that their latency hides behind the load phase
	const uint32_t K1 = n ? (end - 1) >> 6 : 0;              // 64-byte slice (LDS aligned) that holds the last byte
	const uint32_t pw_slice = CRC_POW.v[(1022 - t) & 1023];  // x^(512 (1022 - t))
	const uint32_t pw_tail = CRC_POW8.v[n ? end - K1 * 64 : 0]; // x^(8 r), r = bytes of the last slice (an empty block has none)

	// ---------------- load ----------------
	{
		typedef uint32_t v4u __attribute__((ext_vector_type(4)));
		const NXZ_GLOBAL v4u *s4 = (const NXZ_GLOBAL v4u *)src;
		v4u *d4 = (v4u *)inw;
		uint32_t nfull = total >> 4;
		for (uint32_t i = t; i < nfull; i += NT) d4[i] = s4[i];
		// tail bytes + zero pad (so that over-reads past `end` are defined)
		uint32_t base = nfull << 4;
		for (uint32_t i = base + t; i < base + 48 && i < 65536 + 32; i += NT)
			lds[OFF_IN + i] = i < total ? src[i] : 0;
	}
	for (uint32_t i = t; i < HSIZE; i += NT) head[i] = 0;
	if (t < 316) hist[t] = 0;
	if (t < 16 && t != M_NEXT && !(GEN && (t == M_PREV || t == M_JOBNO))) misc[t] = FUSED && t == M_KEEP ? 3u : 0u;       // (FUSED: the block's header bits, BFINAL = 1, BTYPE = 01)
#ifdef NXZ_LZ77_PROF
	if (t < 20) profacc[t] = 0;
#endif
	// slice-by-4 CRC tables live in the (not yet used) bit buffer: T[k][i] = i advanced by k+1 zero bytes
	{
		uint32_t c = t & 255;
		for (int k = 0; k < 8 * ((t >> 8) + 1); k++) c = (c >> 1) ^ ((c & 1) ? 0xedb88320u : 0);
		bitbuf[t] = c;
	}
	__syncthreads();
	PROF(0);

	// ---------------- checksums of the non-history source ----------------
	// The stream is cut into 64-byte slices aligned in LDS; K1 is the slice that holds the last
	// byte (r = 1..64 bytes of it are data).  Thread 1023 owns that last slice, thread t < 1023 the
	// full slice K1 - (1023 - t); leading threads without a slice contribute 0, which does not
	// change a raw (init 0) CRC.  Slices are read as 4 x ds_read_b128 and fed through a slice-by-4
	// table.  A full slice followed by j more full slices and the r tail bytes weighs
	// x^(512 j) * x^(8 r): the first factor is applied per thread, the XOR reduce is linear, so the
	// second is applied once to the reduced value before the tail slice's CRC joins.  in_crc is
	// XORed into the state in front of the first data byte (history length is a multiple of 16).
	uint32_t out_crc, out_adler;
	{
		uint32_t *T = bitbuf;                                   // T[k*256 + i], k = 0..3
		const uint32_t initx = job.in_crc ^ 0xffffffffu;
		const int sidx = (int)t - 1023 + (int)K1;               // my slice
		uint32_t crc = 0, s1 = 0, sj = 0;
		if (n && sidx >= (int)(h >> 6)) {
			const uint4 *sp = (const uint4 *)(lds + OFF_IN + (uint32_t)sidx * 64);
			uint4 q[4] = { sp[0], sp[1], sp[2], sp[3] };
			uint32_t *w = (uint32_t *)q;
			const uint32_t a0 = (uint32_t)sidx * 64;                // LDS address of the slice
			if (a0 < h || a0 + 64 > end) {
				// first / last slice: dwords of the history or past the end are not part of the stream
#pragma unroll
				for (int k = 0; k < 16; k++)
					if (a0 + 4 * k < h || a0 + 4 * k + 4 > end) w[k] = 0;
			}
			// Adler: S = sum of the bytes, Wt = sum of byte * (offset in the slice); a byte at stream
			// offset i weighs (n - i) in the second sum
			uint32_t S = 0, Wt = 0;
#pragma unroll
			for (int k = 0; k < 16; k++) {
				S = __builtin_amdgcn_udot4(w[k], 0x01010101u, S, false);
				Wt = __builtin_amdgcn_udot4(w[k], 0x03020100u + 0x04040404u * k, Wt, false);
			}
			s1 = S;
			sj = S * (n + h - a0) - Wt;
			// CRC: zero dwords in front of the first data dword leave a zero state alone (the history
			// is a multiple of 16 bytes, so the stream starts at dword 0, 4, 8 or 12 of its slice);
			// past the end only thread 1023 has dwords, and they come last
			const uint32_t kfirst = a0 < h ? (h - a0) >> 2 : 0;
			if (a0 <= h && h < a0 + 64) {
				if (h + 4 <= end) {
					if (kfirst == 0) w[0] ^= initx;
					if (kfirst == 4) w[4] ^= initx;
					if (kfirst == 8) w[8] ^= initx;
					if (kfirst == 12) w[12] ^= initx;
				}
			}
			const uint32_t nd = a0 + 64 <= end ? 16 : (end - a0) >> 2;   // full data dwords in my slice
#pragma unroll
			for (int k = 0; k < 16; k++) {
				const uint32_t c = crc ^ w[k];
				const uint32_t nc = T[768 + (c & 0xff)] ^ T[512 + ((c >> 8) & 0xff)] ^ T[256 + ((c >> 16) & 0xff)] ^ T[c >> 24];
				crc = (uint32_t)k < nd ? nc : crc;
			}
			if (t == 1023 && (end & 3)) {
				// the last 1..3 bytes of the stream
				const uint32_t a = end & ~3u, nb = end & 3, i = a - h;
				const uint32_t v = inw[a >> 2] & ((1u << (8 * nb)) - 1);
				uint32_t b0 = v & 0xff, b1 = (v >> 8) & 0xff, b2 = (v >> 16) & 0xff;
				s1 += b0 + b1 + b2;
				sj += (b0 + b1 + b2) * (n - i) - (b1 + 2 * b2);
				if (i == 0) crc ^= initx;                           // stream shorter than 4 bytes
				for (uint32_t k = 0; k < nb; k++) crc = T[(crc ^ (v >> (8 * k))) & 0xff] ^ (crc >> 8);
			}
		}
		uint32_t tailcrc = 0;
		if (t == 1023) { tailcrc = crc; crc = 0; }
		else if (crc) crc = gf_mul(crc, pw_slice);              // (waves without data skip the multiply: small jobs)
		for (int o = 32; o > 0; o >>= 1) crc ^= __shfl_down(crc, o, 64);
		if (lane == 0) scan[wave] = crc;
		if (t == 1023) scan[50] = tailcrc;
		uint32_t a1 = s1, a2 = sj % 65521u;
		for (int o = 32; o > 0; o >>= 1) {
			a1 += __shfl_down(a1, o, 64);
			a2 += __shfl_down(a2, o, 64);
		}
		if (lane == 0) { scan[16 + wave] = a1; scan[32 + wave] = a2 % 65521u; }
		__syncthreads();
		if (wave == 0) {
			uint32_t c = lane < 16 ? scan[lane] : 0;
			uint32_t b1 = lane < 16 ? scan[16 + lane] : 0, b2 = lane < 16 ? scan[32 + lane] : 0;
			for (int o = 8; o > 0; o >>= 1) {
				c ^= __shfl_down(c, o, 64);
				b1 += __shfl_down(b1, o, 64);
				b2 += __shfl_down(b2, o, 64);
			}
			if (lane == 0) {
				c = n ? gf_mul(c, pw_tail) ^ scan[50] : initx;
				uint32_t ia = job.in_adler & 0xffff, ib = job.in_adler >> 16;
				uint32_t s1f = (ia + b1) % 65521u;
				uint32_t s2f = (uint32_t)(((uint64_t)ib + (uint64_t)n * ia + b2) % 65521u);
				scan[48] = c ^ 0xffffffffu;
				scan[49] = (s2f << 16) | s1f;
			}
		}
		__syncthreads();
		out_crc = __builtin_amdgcn_readfirstlane(scan[48]);      // wave-uniform: keep them in scalar registers until the end
		out_adler = __builtin_amdgcn_readfirstlane(scan[49]);
		__syncthreads();
	}
	PROF(1);

	// ---------------- seed head[] with the window ----------------
	// A bucket is one dword of head[]: the newest position with that hash (+ 1; 0 = empty;
	// positions are below 65533) in the lower half, the entry that was the newest before it in the
	// upper half.  Of the window the two largest positions of a bucket are wanted: first the largest
	// (atomic max, order free; collected in cand[], which is free until the first tile), then the
	// largest of the others (in head[] itself, still whole dwords), then both packed.
	if (h) {
		uint32_t *top = (uint32_t *)cand;
		for (uint32_t i = t; i < HSIZE; i += NT) top[i] = 0;
		__syncthreads();
		for (uint32_t r = t; r < h; r += NT)
			if (r + 4 <= end) {
				const uint32_t v = lds_ld32(inw, r);
				const bool deep = r >= 8 && lds_ld32(inw, r - 8) == v && lds_ld32(inw, r - 4) == v && v == __builtin_amdgcn_alignbyte(v, v, 1);
				if (!deep) atomicMax(&top[hash4(v)], r + 1);
			}
		__syncthreads();
		for (uint32_t r = t; r < h; r += NT)
			if (r + 4 <= end) {
				const uint32_t v = lds_ld32(inw, r);
				const bool deep = r >= 8 && lds_ld32(inw, r - 8) == v && lds_ld32(inw, r - 4) == v && v == __builtin_amdgcn_alignbyte(v, v, 1);
				if (!deep && top[hash4(v)] != r + 1) atomicMax(&head[hash4(v)], r + 1);
			}
		__syncthreads();
		for (uint32_t i = t; i < HSIZE; i += NT) head[i] = top[i] | (head[i] << 16);
	}

	// FUSED: where the block's bits go; dwords written so far, bits in the dword that is open (wave-uniform)
	NXZ_GLOBAL uint32_t *dstw = (NXZ_GLOBAL uint32_t *)job.dst;
	const uint32_t cap_words = job.dst_cap >> 2;
	uint32_t wordbase = 0, obits = 3;
	// where this job's tokens go
	// (GEN: the workgroup's two token slots take turns, the job before this one is encoded from the other)
	const uint32_t gslot = GEN ? (uint32_t)__builtin_amdgcn_readfirstlane((int)misc[M_JOBNO]) & 1u : 0u;
	NXZ_GLOBAL uint8_t *tk = (NXZ_GLOBAL uint8_t *)tokens + (GEN ? ((size_t)blockIdx.x * 2 + gslot) : (size_t)bid) * NXZ_TOK_STRIDE;
	NXZ_GLOBAL uint32_t *g_lit = (NXZ_GLOBAL uint32_t *)(tk + NXZ_TOK_LITBITS);
	NXZ_GLOBAL uint32_t *g_tok = (NXZ_GLOBAL uint32_t *)(tk + NXZ_TOK_MATCHBITS);
	NXZ_GLOBAL uint32_t *g_rec = (NXZ_GLOBAL uint32_t *)(tk + NXZ_TOK_RECORDS);
	NXZ_GLOBAL uint16_t *g_c2 = (NXZ_GLOBAL uint16_t *)cand2 + (size_t)blockIdx.x * C2_STRIDE;   // the second bucket entries of the tile in work
	__syncthreads();

	PROF(2);

	// ================= tiles =================
	for (uint32_t tb0 = 0; tb0 < n; tb0 += PTILE) {
		const uint32_t tn = n - tb0 < PTILE ? n - tb0 : PTILE;       // positions in this tile
		const uint32_t nseg = (tn + PSEG - 1) / PSEG;
		// second bucket entries: always in the first tile, later only where that tile's parse says the data is hard
		// (oracle/nxz_lz77.c step 3b); wave-uniform, so the chain and M1 branch around what serves them
		// (step 3c: text -- few bytes of the first tile with their top bit set -- that is not easy either does without them
		// too, and without the lazy step: its literals are cheap, what is lost there is a few per cent of a margin of 7-8)
		const bool hard = __builtin_amdgcn_readfirstlane(misc[M_TOK0]) >= SECOND_MIN_TOKENS;
		const bool text = __builtin_amdgcn_readfirstlane(misc[M_HIGH]) * TEXT_HIGH_DIV < (n < PTILE ? n : PTILE);
		const bool use2 = tb0 == 0 || (hard && !text);
		const uint32_t lazy_max = tb0 != 0 && hard && text ? 0u : LAZY_MAX;
#ifdef NXZ_CHAIN_ALWAYS2
#define CHAIN_USE2 true
#else
#define CHAIN_USE2 use2
#endif

		// ---- hash ----
		// cand[i] = byte offset of the position's head[] slot (dummy slot for positions without a
		// hash and for the padding up to a multiple of 512 positions, so the chain loop is guard free).
		// Four consecutive positions per lane (two aligned dwords give the four 4-byte strings); a
		// wave then transposes its 512-position piece in place -- lane l of the chain wave gets its 8
		// steps (positions l, l+64, ..) as one 16-byte group -- so that the chain wave moves a piece's
		// slot offsets and results with one ds_read_b128 / ds_write_b128 per lane.
		const uint32_t tnpad = (tn + 511) & ~511u;
		const uint32_t npieces = tnpad >> 9;
		if (t < 3) misc[M_PROGRESS + t] = 0;                    // M_PROGRESS, M_TICKET, M_DEFER
		if (t == 3) misc[M_DEFER2] = 0;
		if (t == 4) misc[M_DQ] = 0;
		sbits[t] = 0;                                           // vb and kb (adjacent, 2 x 512 words)
		uint32_t nhigh = 0;                                     // bytes of the first tile with their top bit set (step 3c)
		for (uint32_t piece = wave; piece < npieces; piece += NT / 64) {
#pragma unroll
			for (int it = 0; it < 2; it++) {
				const uint32_t i = (piece << 9) + (it << 8) + 4 * lane, r = h + tb0 + i;   // r is a multiple of 4
				const uint32_t d0 = inw[r >> 2], d1 = inw[(r >> 2) + 1];
				if (tb0 == 0 && i < tn) nhigh += (uint32_t)__popc(d0 & 0x80808080u);     // (behind the data the image holds zeros)
				const uint32_t v1 = __builtin_amdgcn_alignbyte(d1, d0, 1), v2 = __builtin_amdgcn_alignbyte(d1, d0, 2), v3 = __builtin_amdgcn_alignbyte(d1, d0, 3);
				uint32_t o0 = hash4(d0) << 2, o1 = hash4(v1) << 2, o2 = hash4(v2) << 2, o3 = hash4(v3) << 2;
				// positions that take no part in the table get the dummy slot of their chain lane
				const uint32_t dummy = (HSIZE + ((4 * lane) & 63)) * 4;
				if (i + 4 > tn || r + 7 > end) {                    // ragged end of the tile / of the data
					if (i + 0 >= tn || r + 4 > end) o0 = dummy;
					if (i + 1 >= tn || r + 5 > end) o1 = dummy + 4;
					if (i + 2 >= tn || r + 6 > end) o2 = dummy + 8;
					if (i + 3 >= tn || r + 7 > end) o3 = dummy + 12;
				}
				// deep inside a run of one byte value (the 12 bytes r-8 .. r+3 are equal, oracle/nxz_lz77.c
				// deep_in_run): no lookup, no insert.  The 8 bytes r-4 .. r+3 of the first position
				// are a cheap necessary condition for all four.
				if (r >= 8 && d0 == __builtin_amdgcn_alignbyte(d0, d0, 1) && inw[(r >> 2) - 1] == d0) {
					// the bytes r-4 .. r+3 are one value: position r+j is deep if the 4-j bytes in front of them and the j
					// behind them are that value too
					const uint32_t x2 = inw[(r >> 2) - 2] ^ d0, y = d1 ^ d0;
					if (x2 == 0) o0 = dummy;
					if ((x2 & 0xffffff00u) == 0 && (y & 0xffu) == 0) o1 = dummy + 4;
					if ((x2 & 0xffff0000u) == 0 && (y & 0xffffu) == 0) o2 = dummy + 8;
					if ((x2 & 0xff000000u) == 0 && (y & 0xffffffu) == 0) o3 = dummy + 12;
				}
				*(uint2 *)(cand + i) = make_uint2(o0 | o1 << 16, o2 | o3 << 16);
			}
			__builtin_amdgcn_wave_barrier();
			uint32_t o[8];
#pragma unroll
			for (int u = 0; u < 8; u++) o[u] = cand[(piece << 9) + (u << 6) + lane];
			__builtin_amdgcn_wave_barrier();                      // LDS runs a wave's operations in order: reads before the write
			((uint4 *)cand)[(piece << 6) + lane] = make_uint4(o[0] | o[1] << 16, o[2] | o[3] << 16, o[4] | o[5] << 16, o[6] | o[7] << 16);
		}
		if (tb0 == 0) {
			for (int o = 32; o > 0; o >>= 1) nhigh += __shfl_down(nhigh, o, 64);
			if (lane == 0 && nhigh) atomicAdd(&misc[M_HIGH], nhigh);
		}
		// e(x) = "byte x equals byte x-1" flags for the tile and 288 positions beyond: the lengths of the
		// distance-1 candidates are runs of these flags (used by the match phase and M3)
		{
			uint16_t *eb16 = (uint16_t *)(lds + OFF_EB);
			const uint32_t ngroups = (tn + EBO + 288 + 15) / 16 + 2;
			for (uint32_t g = t; g < ngroups; g += NT) {
				const int32_t r0s = (int32_t)(h + tb0 + 16 * g) - (int32_t)EBO;     // 16-byte aligned, may lie in front of the data
				const uint32_t r0 = r0s < 0 ? end : (uint32_t)r0s;
				uint32_t bits = 0;
				if (r0 < end) {
					const uint4 dv = *(const uint4 *)(lds + OFF_IN + r0);
					const uint32_t pv = r0 ? inw[(r0 >> 2) - 1] : 0;
					auto eq4 = [](uint32_t d, uint32_t prev) -> uint32_t {
						const uint32_t xx = d ^ __builtin_amdgcn_alignbyte(d, prev, 3);      // byte k: b[k] ^ b[k-1]
						const uint32_t nzb = (((xx & 0x7f7f7f7fu) + 0x7f7f7f7fu) | xx) & 0x80808080u;
						return ~((((nzb >> 7) * 0x00204081u) >> 21)) & 0xf;
					};
					bits = eq4(dv.x, pv) | eq4(dv.y, dv.x) << 4 | eq4(dv.z, dv.y) << 8 | eq4(dv.w, dv.z) << 12;
					if (r0 == 0) bits &= ~1u;
					if (end - r0 < 16) bits &= (1u << (end - r0)) - 1;
				}
				eb16[g] = (uint16_t)bits;
			}
		}
		__syncthreads();
		PROF(3);

		// ---- chain (wave 0) overlapped with match (every wave) ----
		// The chain is the only serial dependence of the algorithm.  Wave 0 walks the tile in
		// 512-position pieces and publishes how many are done; all waves (wave 0 too, once the chain
		// is finished) draw piece numbers from a ticket counter and run the match stages on a piece
		// as soon as it is published, so the other 15 waves do not idle behind the chain.
		if (wave == 0) {
			// Per 64-position step: look the slot up, insert (max wins).  The LDS executes one wave's
			// operations in order, so lookup(k+1) only has to be ISSUED after insert(k); nothing waits
			// for a result inside a piece.  The slot offsets of the next piece are fetched ahead and
			// the candidate positions of the previous piece (16 bit) are written one piece late, so no LDS round trip is exposed.
			// Publishing needs no wait either: the flag store follows the data stores in LDS order.
			uint8_t *headb = (uint8_t *)head;
			__builtin_amdgcn_s_setprio(3);                    // the chain is the critical path of this phase
			// (the lane number is made opaque per tile: the eight values lane + 64 u + 1 of a piece's steps would
			// otherwise be computed once per kernel and kept -- three of them spilled, and reloaded right here)
			int clane = lane;
			asm volatile("" : "+v"(clane));
			// Per 64-position step: read the bucket (both entries), store my position + 1 over its
			// newest entry -- a plain 16-bit store: of the lanes of one store instruction that hit the
			// same bucket the highest, i.e. the largest position, lands last (tools/micro/
			// lds_write_order.hip) -- so nothing in a step waits for a result; the LDS executes one
			// wave's operations in order, lookup(k + 1) only has to be ISSUED after insert(k).  What the
			// steps of a piece read (the newest entries before each chunk) is stored as the second
			// entry of their buckets once the NEXT piece has done its lookups (oracle/nxz_lz77.c step
			// 3); by then it has long arrived, and so the piece is handed over at that point too: the
			// newest entries into cand[] (transposed, 16 bytes per lane), the second entries -- no room
			// in LDS -- through this workgroup's ring in device scratch (same layout, one store per
			// piece, it stays in L2); a piece is published when the store of the piece behind it has
			// been issued and its own has been acknowledged.
			auto unpack = [](const uint4 pk, uint32_t (&off)[8]) {
				off[0] = pk.x & 0xffff; off[1] = pk.x >> 16; off[2] = pk.y & 0xffff; off[3] = pk.y >> 16;
				off[4] = pk.z & 0xffff; off[5] = pk.z >> 16; off[6] = pk.w & 0xffff; off[7] = pk.w >> 16;
			};
			auto steps = [&](const uint4 pk, uint32_t piece, uint32_t (&old)[8]) {
				uint32_t off[8]; unpack(pk, off);
#pragma unroll
				for (int u = 0; u < 8; u++) {
					uint32_t *slot = (uint32_t *)(headb + off[u]);
					old[u] = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					__builtin_amdgcn_wave_barrier();
					*(uint16_t *)slot = (uint16_t)(h + tb0 + (piece << 9) + 64 * u + clane + 1);
					__builtin_amdgcn_wave_barrier();
				}
			};
			auto note = [&](const uint4 pk, const uint32_t (&old)[8]) {
				if (!CHAIN_USE2) return;
				uint32_t off[8]; unpack(pk, off);
#pragma unroll
				for (int u = 0; u < 8; u++) *(uint16_t *)(headb + off[u] + 2) = (uint16_t)old[u];
				__builtin_amdgcn_wave_barrier();
			};
			auto hand = [&](const uint32_t (&o)[8], uint32_t piece) {
				// position + 1 (0 = empty), 16 bits each; the consumer subtracts the 1
				typedef uint32_t v4u __attribute__((ext_vector_type(4)));
				// (the newest entries in natural order, position i of the piece at cand[i]: eight 16-bit stores here instead of
				// one 16-byte store and a turn by the wave that takes the piece -- the match phase draws HALF pieces, and two
				// waves cannot turn one piece in place)
#pragma unroll
				for (int u = 0; u < 8; u++) cand[(piece << 9) + 64 * u + lane] = (uint16_t)o[u];
				if (CHAIN_USE2) {
					((NXZ_GLOBAL v4u *)g_c2)[(piece << 6) + lane] = (v4u){ __builtin_amdgcn_perm(o[1], o[0], 0x07060302), __builtin_amdgcn_perm(o[3], o[2], 0x07060302),
											    __builtin_amdgcn_perm(o[5], o[4], 0x07060302), __builtin_amdgcn_perm(o[7], o[6], 0x07060302) };
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_s_waitcnt(0x0F71);            // vmcnt(1): all but this piece's store
				}
				__builtin_amdgcn_wave_barrier();
				if (lane == 0) __hip_atomic_store(&misc[M_PROGRESS], piece, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // pieces below `piece`
				__builtin_amdgcn_wave_barrier();
			};
			const uint32_t dm = (HSIZE + lane) * 4 | ((HSIZE + lane) * 4) << 16;
			const uint4 dummy = make_uint4(dm, dm, dm, dm);
			// two pieces per loop trip (two register sets, no moves); a piece past the end works on the dummy slots
			uint4 pkB = dummy;
			uint32_t oA[8], oB[8];
			for (uint32_t base = 0; base < npieces; base += 2) {
				const uint4 *cp = (const uint4 *)cand + (base << 6) + lane;
				const bool two = base + 1 < npieces;
				const uint4 l1 = cp[two ? 64 : 0];                     // (unconditional 16-byte load of an existing piece, then a select)
				const uint4 pkA = cp[0];
				steps(pkA, base, oA);
				if (base) { note(pkB, oB); hand(oB, base - 1); }
				pkB = two ? l1 : dummy;
				steps(pkB, base + 1, oB);
				note(pkA, oA);
				hand(oA, base);
			}
			// the tile's last piece (its notes too: nothing is pending across tiles)
			{
				const uint32_t lastp = ((npieces + 1) & ~1u) - 1;
				note(pkB, oB);
				hand(oB, lastp);
				__builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
				__builtin_amdgcn_wave_barrier();
				if (lane == 0) __hip_atomic_store(&misc[M_PROGRESS], npieces, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
			__builtin_amdgcn_s_setprio(0);
		}
		PROF(4);

		// ---- match ----
		// Natural match lengths obey N(p) = N(p') + (p' - p) for positions p < p' that match at the
		// same distance when p' lies inside p's match, so of the positions inside one long match
		// only the LAST one (the "tail") compares bytes; the others get their length from it
		// afterwards.  The distance-1 candidate needs no compares at all: its length is the run of
		// "byte equals its predecessor" flags.
		//   M1 (four consecutive positions per lane): validate the hash candidate (window, 4 bytes)
		//      and compare bytes 4..7.  Lengths below 8 are final.  Positions with 8 equal bytes
		//      ("long") are collected in a per-wave queue.
		//   M2 (queue, 64 at a time): a long position whose successor -- the next position with a
		//      verified candidate, at most 8 away -- matches at the same distance is a member of
		//      that successor's chain (bitmap kb: members and the gaps inside chains); the others
		//      are tails and are extended, lane-serial up to LCAP bytes, 16 lanes per tail beyond.
		//   M3 (after a barrier): members take their length from the end of their chain;
		//      distance-1 runs are evaluated and win ties (oracle/nxz_lz77.c step 4).
		// first clear bit of a bitmap at or after bit s (below limit)
		auto first_zero = [](const uint32_t *bm, uint32_t s0, uint32_t limit) -> uint32_t {
			// 128 bits per LDS round trip
			uint32_t wi = (s0 >> 5) & ~3u;
			uint4 q = *(const uint4 *)(bm + wi);
			// bits below s0 count as set
			const uint32_t sw = (s0 >> 5) & 3, lowmask = ~(~0u << (s0 & 31));
			uint32_t w0 = ~q.x, w1 = ~q.y, w2 = ~q.z, w3 = ~q.w;
			if (sw > 0) w0 = 0; if (sw > 1) w1 = 0; if (sw > 2) w2 = 0;
			if (sw == 0) w0 &= ~lowmask; else if (sw == 1) w1 &= ~lowmask; else if (sw == 2) w2 &= ~lowmask; else w3 &= ~lowmask;
			for (;;) {
				if (w0 | w1 | w2 | w3) {
					const uint32_t pos = wi * 32 + (w0 ? (uint32_t)__builtin_ctz(w0) : w1 ? 32 + (uint32_t)__builtin_ctz(w1)
								  : w2 ? 64 + (uint32_t)__builtin_ctz(w2) : 96 + (uint32_t)__builtin_ctz(w3));
					return pos < limit ? pos : limit;
				}
				wi += 4;
				if (wi * 32 >= limit) return limit;
				q = *(const uint4 *)(bm + wi);
				w0 = ~q.x; w1 = ~q.y; w2 = ~q.z; w3 = ~q.w;
			}
		};
		uint32_t *kb = mbits;                                 // bitmap: long members (and the gaps inside their chains)
		uint32_t *vb = sbits;                                 // bitmap: position has a verified candidate
		{
			const uint32_t *eb = (const uint32_t *)(lds + OFF_EB);
			// 32 bits of a bitmap from bit b on
			auto bits32 = [](const uint32_t *bm, uint32_t b) -> uint32_t {
				const uint32_t lo = bm[b >> 5], hi = bm[(b >> 5) + 1], sh = b & 31;
				return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
			};
			uint16_t *lq = (uint16_t *)(lds + OFF_LQ) + wave * LQ_CAP;          // long positions: quad number | position bits << 12
			uint16_t *xq = (uint16_t *)(lds + OFF_XQ) + wave * XQ_CAP;          // tails beyond LCAP
			uint32_t xqn = 0;
			uint32_t lqn = 0;
			// M2 in two levels (a long position = one with 8 equal bytes whose direct successor does not go on at its distance):
			//  level 1 (stage2, the queue lq[]): bytes 8..23 in one LDS round trip.  A mismatch among them makes the length
			//     final -- seven in ten end here, and they pay for nothing else (up to round 3 every long position looked
			//     for its successor first: two more round trips and the member bookkeeping for all of them).
			//  level 2 (the queue xq[], 16 or more at a time): positions that are still open after OPEN2 = 24 bytes.  A
			//     successor at the same distance inside those bytes makes one a member of that successor's chain (M3 reads its
			//     length off the chain's end); the last position of a unit is settled after the barrier; the others are TAILS:
			//     a. a GROUP at a time.  All the tails of one unit of 256 positions that have the same distance end at the
			//        same few mismatches: the wave compares the 512 bytes behind the unit's start with those a distance in
			//        front of them ONCE (8 bytes per lane, one ballot), and every tail of the group reads its end off -- the
			//        first lane at or behind its own position that holds a mismatch (its first 24 bytes are known to be
			//        equal, so a lane boundary up to 8 bytes behind its start is where it begins to look).  Data that repeats
			//        at several distances at once (JSON, msgpack: neighbours take turns between two or three distances, a
			//        chain is 4.6 positions long and its tail 113 bytes) costs one trip per distance instead of one per tail
			//        and 16 bytes.  Whether a batch is worth the look is decided by a count: tails of one group stand close
			//        together in the queue, so those whose distance is that of one of the two entries in front of them are
			//        counted (two DPP moves); text and binaries, where every match has its own distance, stop there.
			//     b. what is left: with a dozen or more every lane finishes its own, 32 bytes per LDS round trip (periodic
			//        data, candidates at ever changing distances); else 16 lanes per tail, 64 bytes per step.
			// The lengths are exact whichever way they are found (oracle/nxz_lz77.c step 4).
			auto level2 = [&](uint32_t nl) {
				for (uint32_t base = 0; base < nl; base += 64) {
					const bool act = base + lane < nl;
					const uint32_t i = act ? (uint32_t)xq[base + lane] : 0;
					// round trip 1: my distance and the verified positions i+1 .. i+32, not looking beyond the unit (another wave's)
					const uint32_t dT = cand[i];                            // my distance - 1
					const uint32_t pend = (i | 255) + 1 < tn ? (i | 255) + 1 : tn;
					uint32_t w32 = (uint32_t)((((uint64_t)vb[((i + 1) >> 5) + 1] << 32) | vb[(i + 1) >> 5]) >> ((i + 1) & 31));
					if (pend - i <= 32) w32 &= (1u << (pend - i - 1)) - 1;
					const uint32_t g1 = w32 ? (uint32_t)__builtin_ctz(w32) : 32, sp = w32 ? i + 1 + g1 : i;
					// round trip 2: the successor's distance
					const bool same_d = w32 != 0 && cand[sp] == dT;
					__builtin_amdgcn_wave_barrier();
					bool lg = false;
					if (act) {
						if (same_d && g1 < OPEN2) {
							// the successor lies inside my verified bytes: N = N(successor) + gap
							const uint64_t km = (((uint64_t)2 << g1) - 1) << (i & 31);               // bits i .. sp-1
							atomicOr(&kb[i >> 5], (uint32_t)km);
							if (km >> 32) atomicOr(&kb[(i >> 5) + 1], (uint32_t)(km >> 32));
						} else if (w32 == 0 && i + 1 == pend && pend < tn) {
							// last position of the unit: settled after the barrier
							atomicOr(&misc[(i >> 13) & 1 ? M_DEFER2 : M_DEFER], 1u << ((i >> 8) & 31));
						} else
							lg = true;
					}
					// the next 32 bytes of every tail, each lane its own: that ends the short ones (tables, XML: nearly all), and
					// what is left is worth looking for groups in
					if (__ballot(lg)) {
						const uint32_t r = h + tb0 + i, q = lg ? r - dT - 1 : 0;
						const uint32_t maxlen = end - r < MAXMATCH ? end - r : MAXMATCH;
						if (lg) {
							const uint32_t k1 = equal16(inw, q + OPEN2, r + OPEN2), k2 = equal16(inw, q + OPEN2 + 16, r + OPEN2 + 16);
							const uint32_t k = k1 == 16 ? 16 + k2 : k1;
							if (k < 32 || OPEN2 + k >= maxlen) {
								lg = false;
								mlen[i] = (uint8_t)((OPEN2 + k < maxlen ? OPEN2 + k : maxlen) - 3);
							}
						}
					}
					constexpr uint32_t SEEN = OPEN2 + 32;
					unsigned long long todo = __ballot(lg);                 // tails whose distance has not been looked at
					PCOUNT(15, __popcll(todo));                             // ... tails
					PC2(0, 1);
#ifdef NXZ_GROUP_NEVER
					todo = 0;
#endif
					if (todo) {
						// worth a look at all?  Tails of one group stand close together in the queue (they come in position
						// order): count those whose distance is that of one of the two entries in front of them
						const uint32_t key = lg ? dT | (i >> 8) << 16 : 0x80000000u | (uint32_t)lane;
						const uint32_t k1 = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)key, 0x111, 0xf, 0xf, false);   // row_shr:1
						const uint32_t k2 = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)key, 0x112, 0xf, 0xf, false);   // row_shr:2
						// ... and only those lead a round: a tail on its own never does (round 4 took the leaders in queue order
						// and gave a batch up after two of them had no company)
						todo = __ballot(lg && (key == k1 || key == k2));
						if ((uint32_t)__popcll(todo) < GROUP_GATE) { todo = 0; PC2(13, 1); }
					}
					for (uint32_t round = 0; todo && round < GROUP_ROUNDS; round++) {
						const int ldr = __builtin_ctzll(todo);
						const uint32_t dL = (uint32_t)__builtin_amdgcn_readlane((int)dT, ldr);
						const uint32_t uL = (uint32_t)__builtin_amdgcn_readlane((int)i, ldr) >> 8;
						const bool mine = lg && dT == dL && (i >> 8) == uL;
						const unsigned long long mm = __ballot(mine);
						todo &= ~mm;
						if ((uint32_t)__popcll(mm) < GROUP_MIN) {               // a tail on its own: the rounds below are cheaper,
							PC2(4, 1);
							continue;
						}
						PC2(2, 1); PC2(3, __popcll(mm));
						const uint32_t R = h + tb0 + (uL << 8);                 // 16-byte aligned
						const uint32_t a = R + 8 + 8 * (uint32_t)lane;
						const uint2 rv = *(const uint2 *)(lds + OFF_IN + a);
						const uint32_t qa = a > dL ? a - dL - 1 : 0;            // (lanes in front of every tail of the group may lie in front of the data)
						const uint32_t qw = qa >> 2, qs = qa & 3;
						const uint32_t u0 = inw[qw], u1 = inw[qw + 1], u2 = inw[qw + 2];
						const uint32_t x0 = rv.x ^ __builtin_amdgcn_alignbyte(u1, u0, qs), x1 = rv.y ^ __builtin_amdgcn_alignbyte(u2, u1, qs);
						const uint32_t f = x0 ? (uint32_t)__builtin_ctz(x0) >> 3 : 4 + ((x1 ? (uint32_t)__builtin_ctz(x1) : 32u) >> 3);
						const unsigned long long eqm = __ballot((x0 | x1) == 0);
						const uint32_t o = i & 255, l0 = o <= 8 ? 0 : (o - 1) >> 3;
						const unsigned long long z = mine ? ~eqm >> l0 : 0;
						const uint32_t fl = z ? l0 + (uint32_t)__builtin_ctzll(z) : (uint32_t)lane;
						const uint32_t ff = __shfl(f, fl, 64);
						if (mine) {
							const uint32_t r = h + tb0 + i;
							const uint32_t maxlen = end - r < MAXMATCH ? end - r : MAXMATCH;
							uint32_t N = z ? R + 8 + 8 * fl + ff - r : maxlen;
							if (N > maxlen) N = maxlen;
							mlen[i] = (uint8_t)(N - 3);
							lg = false;
						}
					}
					const unsigned long long ml = __ballot(lg);
					const uint32_t nleft = (uint32_t)__popcll(ml);
					if (nleft >= 12) {
						const uint32_t r = h + tb0 + i, q = lg ? r - dT - 1 : 0;
						const uint32_t maxlen = end - r < MAXMATCH ? end - r : MAXMATCH;
						uint32_t len = SEEN;
						PC2(5, 1); PC2(7, nleft);
						while (__ballot(lg)) {
							PC2(6, 1);
							if (lg) {
								const uint32_t k1 = equal16(inw, q + len, r + len), k2 = equal16(inw, q + len + 16, r + len + 16);
								const uint32_t k = k1 == 16 ? 16 + k2 : k1;
								len += k;
								if (k < 32 || len >= maxlen) {
									lg = false;
									mlen[i] = (uint8_t)((len < maxlen ? len : maxlen) - 3);
								}
							}
						}
					} else if (nleft) {
						// to the front of this batch's part of the queue (every entry of it is in a register by now)
						if (lg) xq[base + __popcll(ml & ((1ull << lane) - 1))] = (uint16_t)i;
						__builtin_amdgcn_wave_barrier();
						const uint32_t g = lane >> 4, li = lane & 15;
						PC2(8, 1); PC2(14, nleft);
						for (uint32_t b4 = 0; b4 < nleft; b4 += 4) {
							const bool act = b4 + g < nleft;
							const uint32_t i = act ? (uint32_t)xq[base + b4 + g] : 0;
							const uint32_t r = h + tb0 + i, q = r - cand[i] - 1;
							const uint32_t maxlen = end - r < MAXMATCH ? end - r : MAXMATCH;
							uint32_t N = 0;
							bool done = !act;
							for (uint32_t off = SEEN;; off += 64) {
								PC2(9, 1);
								const uint32_t o = off + 4 * li;
								uint32_t x = 0;
								if (!done && o < maxlen) x = lds_ld32(inw, q + o) ^ lds_ld32(inw, r + o);
								const bool ev = !done && (x != 0 || o + 4 >= maxlen);   // mismatch or end of the compare
								const unsigned long long mm = __ballot(ev);
								const uint32_t gm = (uint32_t)(mm >> (16 * g)) & 0xffffu;
								uint32_t myN = x ? o + ((uint32_t)__builtin_ctz(x) >> 3) : maxlen;
								if (myN > maxlen) myN = maxlen;
								const uint32_t firstN = __shfl(myN, gm ? (g << 4) + (uint32_t)__builtin_ctz(gm) : (uint32_t)lane, 64);
								if (!done && gm) { N = firstN; done = true; }
								if (!__ballot(!done)) break;
							}
							if (act && li == 0) mlen[i] = (uint8_t)(N - 3);
						}
					}
				}
			};
			// (ONE call site for either level, see the loop below: the kernel sits at its register cap and every copy of
			// this code was paid for in spills)
			auto stage2 = [&](uint32_t nq, bool last) {
				const uint32_t ent = lane < nq ? (uint32_t)lq[lane] : 0;
				__builtin_amdgcn_wave_barrier();
				if (nq > 64 && 64 + lane < nq) lq[lane] = lq[64 + lane];
				// almost always one position per entry; more only when neighbours have different distances
				uint32_t pb = ent >> 12;
				do {
					const uint32_t i = pb ? ((ent & 0xfff) << 2) + (uint32_t)__builtin_ctz(pb) : 0xffffffffu;
					bool open = false;
					if (i != 0xffffffffu) {
						const uint32_t r = h + tb0 + i, q = r - cand[i] - 1;
						const uint32_t maxlen = end - r < MAXMATCH ? end - r : MAXMATCH;
						uint32_t len = 8 + equal16(inw, q + 8, r + 8);
						open = len == OPEN2 && maxlen > OPEN2;
						if (len > maxlen) len = maxlen;
						if (!open) mlen[i] = (uint8_t)(len - 3);
					}
					const unsigned long long ml = __ballot(open);
					if (ml) {
						if (open) xq[xqn + __popcll(ml & ((1ull << lane) - 1))] = (uint16_t)i;
						xqn += (uint32_t)__popcll(ml);
						PCOUNT(14, __popcll(ml));                  // ... still open after OPEN2 bytes
						__builtin_amdgcn_wave_barrier();
					}
					pb &= pb - 1;
					const bool more = __ballot(pb != 0) != 0;
					// (XQ_MIN or more are worth the trip; the phase's last call empties the queue)
					if (xqn >= XQ_MIN || (last && !more && nq <= 64 && xqn)) { level2(xqn); xqn = 0; }
					if (!more) break;
				} while (true);
			};
			// The unit of work a wave draws is 256 positions (four per lane): a tile is 64 of them, so the waves finish
			// within one unit's time of each other (with whole pieces of 512 the last ones decided the phase).  A unit is
			// classified on its own: nothing looks beyond its last position (another wave's), whose successor is settled
			// after the barrier.
			const uint32_t nunits = (tn + 255) >> 8;
			for (;;) {
				uint32_t unit = 0;
				if (lane == 0) unit = atomicAdd(&misc[M_TICKET], 1u);
				unit = __builtin_amdgcn_readfirstlane(unit);
				const bool last = unit >= nunits;                      // nothing left to draw: the queues are emptied
				const uint32_t piece = unit >> 1;
				uint32_t qbits = 0;
				const uint32_t it = unit & 1;
				const uint32_t ib = unit << 8;
				const uint32_t i4 = ib + 4 * lane, r4 = h + tb0 + i4;       // r4 is a multiple of 4
				WPROF_BEGIN();
				if (!last) {
					while (__hip_atomic_load(&misc[M_PROGRESS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= piece)
						__builtin_amdgcn_s_sleep(4);
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					WPROF_END(16);                                         // waiting for the chain
					auto quad = [&](auto fullt, auto use2t) {
						// FULL: far enough from the end of the tile and of the data, no clamps needed
						// USE2: with the second bucket entries (a variant of its own: a branch inside it cost the common case 2-4 %)
						constexpr bool FULL = decltype(fullt)::value, USE2 = decltype(use2t)::value;
						// the second bucket entries of my four positions: device scratch, written by the chain
						// wave (this CU's L1 may hold the previous tile's: read past it)
						// (transposed like cand[] before it is turned back: position w of the piece at 8 (w % 64) + w / 64)
						uint32_t q2[4] = { 0, 0, 0, 0 };
						if constexpr (USE2) {
#pragma unroll
							for (int j = 0; j < 4; j++)
								q2[j] = __hip_atomic_load(g_c2 + (piece << 9) + (4 * (lane & 15) + j) * 8 + 4 * it + (lane >> 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						}
						const uint2 qq = *(const uint2 *)(cand + i4);
						// deep-in-run positions took no part in the table (what the chain wave left
						// in their place is meaningless): flags i-7 .. i+3 all set
						const uint32_t w14 = bits32(eb, EBO + i4 - 7);
						const uint32_t d0 = inw[r4 >> 2], d1 = inw[(r4 >> 2) + 1], d2 = inw[(r4 >> 2) + 2];
						uint32_t mw = 0, c01 = 0, c23 = 0, vbits = 0, kbits = 0;
						bool okA[4], raw8[4], lng[4];
						uint32_t dA[4];
#pragma unroll
						for (int j = 0; j < 4; j++) {
							const uint32_t i = i4 + j, r = r4 + j;
							const bool ok = (FULL || (i < tn && r + 4 <= end)) && ((w14 >> j) & 0x7ff) != 0x7ff;
							const uint32_t maxlen = FULL || end - r >= MAXMATCH ? MAXMATCH : end - r;
							const uint32_t v = j ? __builtin_amdgcn_alignbyte(d1, d0, j) : d0, v4 = j ? __builtin_amdgcn_alignbyte(d2, d1, j) : d1;
							// a bucket entry (position + 1 from the chain, 0 = none): a candidate if inside the
							// window with four equal bytes; returns the equal bytes among the first eight (4..8),
							// 0 if it is none
							auto probe = [&](uint32_t entry, uint32_t &dist1) -> uint32_t {
								const uint32_t qc = (uint16_t)(entry - 1);      // 0xffff = none
								dist1 = r - qc - 1;                             // distance - 1; wraps to something huge if qc >= r
								const bool qok = ok && dist1 < WINDOW;
								const uint32_t q = FULL || qok ? qc : 0;
								const uint32_t qa = q >> 2, u0 = inw[qa], u1 = inw[qa + 1], u2 = inw[qa + 2];
								const uint32_t qv = __builtin_amdgcn_alignbyte(u1, u0, q & 3), qv4 = __builtin_amdgcn_alignbyte(u2, u1, q & 3);
								const uint32_t x = qv4 ^ v4;
								uint32_t n8 = 4 + ((x ? (uint32_t)__builtin_ctz(x) : 32u) >> 3);
								if (!FULL && n8 > maxlen) n8 = maxlen;          // (maxlen >= 4 where ok)
								return qok && qv == v ? n8 : 0;
							};
							uint32_t d1st, d2nd;
							const uint32_t n1st = probe((j & 2 ? qq.y : qq.x) >> (16 * (j & 1)), d1st);
							uint32_t n2nd = 0;
							d2nd = 0;
							if constexpr (USE2) n2nd = probe(q2[j], d2nd);
							// the older entry only if it has more of the first eight bytes (oracle/nxz_lz77.c step 4)
							const bool second = n2nd > n1st;
							const uint32_t lenA = second ? n2nd : n1st;
							okA[j] = lenA != 0;
							dA[j] = second ? d2nd : d1st;
							raw8[j] = lenA == 8;                        // at least 8 bytes
							lng[j] = raw8[j] && (FULL || maxlen > 8);   // ... and possibly more
							mw |= (lenA ? lenA - 3 : 0) << (8 * j);
							const uint32_t cj = okA[j] ? dA[j] : NOHASH;
							if (j < 2) c01 |= cj << (16 * j); else c23 |= cj << (16 * (j - 2));
							vbits |= (uint32_t)okA[j] << j;
						}
						// A long position whose direct successor is one of my four, verified and at the same
						// distance, is settled right here: N = N(successor) + 1, which is the 8 stored above
						// unless the successor has 8 bytes too -- then it is a member of the successor's
						// chain.  Every other long position goes to the queue.
#pragma unroll
						for (int j = 0; j < 3; j++) {
							const bool direct = okA[j + 1] && dA[j] == dA[j + 1];
							kbits |= (uint32_t)(lng[j] && direct && raw8[j + 1]) << j;
							qbits |= (uint32_t)(lng[j] && !direct) << j;
						}
						// ... or whose successor is two or three positions on, still among my four (the
						// positions in between have no candidate): member, the gap belongs to the chain
						{
							const bool g13 = lng[1] && !okA[2] && okA[3] && dA[1] == dA[3];
							const bool g02 = lng[0] && !okA[1] && okA[2] && dA[0] == dA[2];
							const bool g03 = lng[0] && !okA[1] && !okA[2] && okA[3] && dA[0] == dA[3];
							kbits |= (g13 ? 6u : 0u) | (g02 ? 3u : 0u) | (g03 ? 7u : 0u);
							qbits &= ~((g13 ? 2u : 0u) | (g02 || g03 ? 1u : 0u));
						}
						// my last position: its direct successor is the first position of the next lane
						// (unknown at the unit's end: settled after the barrier)
						{
							const uint32_t key0 = okA[0] ? dA[0] | (uint32_t)raw8[0] << 16 : 0xffffffffu;
							uint32_t skey = __shfl_down(key0, 1, 64);
							if (lane == 63) skey = 0xffffffffu;
							const bool direct = skey != 0xffffffffu && (skey & 0xffff) == dA[3];
							kbits |= (uint32_t)(lng[3] && direct && (skey >> 16)) << 3;
							qbits |= (uint32_t)(lng[3] && !direct) << 3;
						}
						*(uint32_t *)(mlen + i4) = mw;
						*(uint2 *)(cand + i4) = make_uint2(c01, c23);
						if (vbits) atomicOr(&vb[i4 >> 5], vbits << (i4 & 31));
						if (kbits) atomicOr(&kb[i4 >> 5], kbits << (i4 & 31));
					};
#ifndef NXZ_ABL_NO_M1
					const bool full = ib + 256 <= tn && h + tb0 + ib + 256 + MAXMATCH + 8 <= end;
					if (use2) { if (full) quad(std::true_type{}, std::true_type{}); else quad(std::false_type{}, std::true_type{}); }
					else if (full) quad(std::true_type{}, std::false_type{});
					else quad(std::false_type{}, std::false_type{});
#endif
				}
				// queue entry: quad number | position bits.  One position per entry as a rule (a lane's
				// second position gets an entry of its own), so that a batch is one pass in stage2
				for (uint32_t rest = qbits, round = 0; round < 2; round++) {
					const uint32_t bits = round ? rest : rest & (0u - rest);      // lowest position first, then what is left
					const unsigned long long m = __ballot(bits != 0);
#ifndef NXZ_ABL_NO_M2
					if (bits) lq[lqn + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)((i4 >> 2) | bits << 12);
					lqn += (uint32_t)__popcll(m);
					PCOUNT(13, __popcll(m));                       // positions queued for M2
#endif
					__builtin_amdgcn_wave_barrier();
					WPROF_END(17);                                 // M1
					while (lqn >= (last ? 1u : LQ_MIN) || (last && xqn)) { stage2(lqn, last); lqn = lqn > 64 ? lqn - 64 : 0; }
					WPROF_END(18);                                 // M2 (classification, tails)
					rest &= rest - 1;
				}
				if (last) break;
			}
		}
		__syncthreads();
		PROF(12);
		// the piece-last long positions: member if the first position of the next piece continues
		// the match, tail otherwise -- extended right here by a whole wave, 256 bytes per step
		for (uint32_t pc = wave; pc < 64; pc += NT / 64) {
			if (!((misc[pc & 32 ? M_DEFER2 : M_DEFER] >> (pc & 31)) & 1)) continue;
			const uint32_t ii = (pc << 8) + 255;
			if (cand[ii + 1] == cand[ii]) {
				if (lane == 0 && mlen[ii + 1] >= 5) kb[ii >> 5] |= 0x80000000u;     // successor has 8 bytes or more: member
				continue;
			}
			const uint32_t r = h + tb0 + ii, q = r - cand[ii] - 1;
			const uint32_t maxlen = end - r < MAXMATCH ? end - r : MAXMATCH;
			uint32_t N = maxlen;
			for (uint32_t ob = 8; ob < maxlen; ob += 256) {
				const uint32_t o = ob + 4 * lane;
				const uint32_t x = o < maxlen ? lds_ld32(inw, q + o) ^ lds_ld32(inw, r + o) : 0;
				const unsigned long long mm = __ballot(x != 0);
				if (mm) {
					const int fl = __builtin_ctzll(mm);
					const uint32_t nn = ob + 4 * fl + ((uint32_t)__builtin_ctz(__shfl(x, fl, 64)) >> 3);
					N = nn < maxlen ? nn : maxlen;
					break;
				}
			}
			if (lane == 0) mlen[ii] = (uint8_t)(N - 3);
		}
		__syncthreads();
		PROF(10);
		// ---- M3: long members, then distance-1 runs ----
		const uint32_t p0 = (uint32_t)t * PSEG;
		{
			// segments in which a run of four "equals its predecessor" flags starts: queued for the distance-1 pass below
			const uint32_t *eb = (const uint32_t *)(lds + OFF_EB);
			uint16_t *dq = (uint16_t *)(lds + OFF_DQ);
			const uint32_t e32 = p0 < tn ? lds_ld32(eb, 2 * t + EBO / 8) : 0;
			const bool runs = (e32 & (e32 >> 1) & (e32 >> 2) & (e32 >> 3) & 0xffff) != 0;
			const unsigned long long rm = __ballot(runs);
			if (rm) {
				uint32_t qb = 0;
				if (lane == 0) qb = atomicAdd(&misc[M_DQ], (uint32_t)__popcll(rm));
				qb = __builtin_amdgcn_readfirstlane(qb);
				if (runs) dq[qb + (uint32_t)__popcll(rm & ((1ull << lane) - 1))] = (uint16_t)t;
			}
		}
		if (p0 < tn) {
			// members of my 16 positions, last to first: N = (end of the chain) - position; the chain
			// ends at the first non-member behind it, whose length is final (M1, M2 or the link pass).
			// Straight-line code over the 16 positions (their stored lengths in four registers, one 16-byte
			// read and one 16-byte write): a loop over the members alone ran as long as the lane with the
			// most members in every wave.
			const uint32_t kb16 = ((const uint16_t *)kb)[t];
			const uint32_t kbits = kb16 & ((const uint16_t *)vb)[t];
			if (kbits) {
				const uint4 mv = *(const uint4 *)(mlen + p0);
				uint32_t mw[4] = { mv.x, mv.y, mv.z, mv.w };
				// the end (position + length) of the chain that runs out of my segment
				uint32_t E = 0;
				if (kb16 >> 15) {
					// more than 258 + 16 positions away is as good as infinitely far
					const uint32_t lim = p0 + 16 + 272 < tn ? p0 + 16 + 272 : tn;
					uint32_t s0 = p0 + 16;
					asm volatile("" : "+v"(s0));                  // not hoisted out of the tile loop (it would be spilled)
					const uint32_t T = first_zero(kb, s0, lim);
					E = T < lim ? T + mlen[T] + 3 : T + MAXMATCH;
				}
				const uint32_t room15 = end - (h + tb0 + p0 + 15);      // bytes from my last position to the end of the data
#pragma unroll
				for (int j = 15; j >= 0; j--) {
					const uint32_t m = (mw[j >> 2] >> (8 * (j & 3))) & 0xff;
					E = (kb16 >> j) & 1 ? E : p0 + j + m + 3;             // a non-member ends the chains in front of it
					const uint32_t room = room15 + (15 - j);
					uint32_t N = E - (p0 + j);
					N = N < room ? N : room;
					N = N < MAXMATCH ? N : MAXMATCH;
					if ((kbits >> j) & 1) mw[j >> 2] = (mw[j >> 2] & ~(0xffu << (8 * (j & 3)))) | ((N - 3) & 0xff) << (8 * (j & 3));
				}
				*(uint4 *)(mlen + p0) = make_uint4(mw[0], mw[1], mw[2], mw[3]);
			}
		}
		__syncthreads();
		PROF(11);
		// chain bookkeeping init: mark[] aliases the vb bitmap, which nobody reads any more
		for (uint32_t s = t; s < NSEG; s += NT) mark[s] = 0;
		// distance 1: the length is the run of e flags that starts at the position; it wins ties.  Only the segments
		// in which a run of four flags starts have anything to do, and where such runs are scattered (the blanks of
		// text and sources, the zero words of binaries) every wavefront held one: they were queued by the lanes that
		// own them (above, beside the members) and are worked through 64 at a time here.
		{
			const uint32_t *eb = (const uint32_t *)(lds + OFF_EB);
			const uint16_t *dq = (const uint16_t *)(lds + OFF_DQ);
			const uint32_t qn = misc[M_DQ];
			for (uint32_t qi = t; qi < qn; qi += NT) {
				const uint32_t sg = dq[qi], q0 = sg * PSEG;
				const uint32_t e32 = lds_ld32(eb, 2 * sg + EBO / 8);     // the segment's 16 flags and the next 16
				const uint32_t e16 = e32 & 0xffff;
				const uint4 mv = *(const uint4 *)(mlen + q0);
				const uint4 c0 = ((const uint4 *)cand)[2 * sg], c1 = ((const uint4 *)cand)[2 * sg + 1];
				const uint32_t mw[4] = { mv.x, mv.y, mv.z, mv.w };
				uint32_t cw[8] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
				uint32_t zb = 16;
				if (e16 >> 15) {
					uint32_t s0 = EBO + q0 + 16;
					asm volatile("" : "+v"(s0));                      // not hoisted out of the tile loop (it would be spilled)
					zb = first_zero(eb, s0, s0 + 272) - EBO - q0;
				}
				const uint32_t r15 = h + tb0 + q0 + 15;
				uint32_t om[4] = { 0, 0, 0, 0 };
#pragma unroll
				for (int j = 15; j >= 0; j--) {
					const uint32_t i = q0 + j, r = r15 - (15 - j);
					const uint32_t room = end > r ? end - r : 0;
					const uint32_t ml = room < MAXMATCH ? room : MAXMATCH;
					const uint32_t mj = (mw[j >> 2] >> (8 * (j & 3))) & 0xff;
					uint32_t cj = (cw[j >> 1] >> (16 * (j & 1))) & 0xffff;
					const uint32_t NA = mj ? mj + 3 : 0;
					const uint32_t zmask = ~e16 & (0xffffu << j) & 0xffffu;
					const uint32_t zpos = zmask ? (uint32_t)__builtin_ctz(zmask) : zb;
					uint32_t NB = zpos - j;
					if (NB > ml) NB = ml;
					uint32_t len = NA;
					if (NB >= 4 && NB >= NA && i < tn) { len = NB; cj = 0; }
					om[j >> 2] |= (len ? len - 3 : 0) << (8 * (j & 3));
					cw[j >> 1] = (cw[j >> 1] & ~(0xffffu << (16 * (j & 1)))) | cj << (16 * (j & 1));
				}
				*(uint4 *)(mlen + q0) = make_uint4(om[0], om[1], om[2], om[3]);
				((uint4 *)cand)[2 * sg] = make_uint4(cw[0], cw[1], cw[2], cw[3]);
				((uint4 *)cand)[2 * sg + 1] = make_uint4(cw[4], cw[5], cw[6], cw[7]);
			}
		}
		__syncthreads();

		PROF(5);
		// ---- parse pass 1: speculative walk of segment s from its own start ----
		// The 16 stored lengths of the segment (+1 for the lazy look-ahead) are fetched with one
		// 16-byte LDS read and the walk runs out of registers.
		uint32_t sm0 = 0, sm1 = 0, sm2 = 0, sm3 = 0;          // final stored lengths (len-3, 0 = none) of my 16 positions
		uint32_t seg_next = 0, seg_nz = 0;
		uint32_t mm1 = 0, lit1 = 0;                            // what the speculative walk visits: match / literal token starts (16 bits)
		uint32_t lastd = 0;                                    // distance - 1 of its last match
		auto seg_m = [&](uint32_t k) -> uint32_t {              // stored length (len-3, 0 = none) of position p0+k, k <= 16
			// two levels of 2-way selects (a 4-way select on a computed index becomes a scratch table)
			const uint32_t a = (k & 8) ? sm2 : sm0, b = (k & 8) ? sm3 : sm1;
			const uint32_t w = (k & 4) ? b : a;
			const uint32_t v = (w >> (8 * (k & 3))) & 0xff;
			return (k & 16) ? seg_next : v;
		};
		// one greedy/lazy step on register data (oracle/nxz_lz77.c walk())
		auto seg_step = [&](uint32_t p, uint32_t limit, bool &is_match) -> uint32_t {
			const uint32_t k = p - p0, cur = seg_m(k);
			const uint32_t full = cur ? cur + 3 : 0;
			const uint32_t len = full < limit - p ? full : limit - p;
			is_match = false;
			if (full >= 4 && len >= 3) {
				if (len < lazy_max && p + 1 < limit) {
					uint32_t m2 = seg_m(k + 1);
					uint32_t l2 = m2 ? m2 + 3 : 0;
					if (l2 > limit - p - 1) l2 = limit - p - 1;
					if (l2 > len) return 1;
				}
				is_match = true;
				return len;
			}
			return 1;
		};
		if ((uint32_t)t < nseg) {
			{
				const uint4 mv = *(const uint4 *)(mlen + p0);
				sm0 = mv.x; sm1 = mv.y; sm2 = mv.z; sm3 = mv.w;
			}
			seg_next = p0 + 16 < tn ? mlen[p0 + 16] : 0;
			// 16-bit mask of positions that have a match (non-zero byte), 4 bits per dword
			auto nz4 = [](uint32_t w) -> uint32_t {
				uint32_t x = (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;
				return (((x >> 7) * 0x00204081u) >> 21) & 0xf;
			};
			seg_nz = nz4(sm0) | nz4(sm1) << 4 | nz4(sm2) << 8 | nz4(sm3) << 12;
			uint32_t p = p0, stop = p + PSEG < tn ? p + PSEG : tn;
			uint32_t cov = 0;                                   // positions of the segment inside a match of this walk (its start too)
			while (p < stop) {
				// branch free: skip the literals in front of the next match, then take one step there
				const uint32_t rest = seg_nz >> (p - p0);
				p += rest ? (uint32_t)__builtin_ctz(rest) : stop - p;
				bool m;
				const uint32_t adv = seg_step(p < stop ? p : p0, tn, m);
				const bool took = p < stop && m;
				mm1 |= took ? 1u << (p - p0) : 0;
				cov |= took ? (adv >= 16 ? 0xffffu : (1u << adv) - 1) << (p - p0) : 0;
				p += p < stop ? adv : 0;
			}
			lit1 = ~cov & (stop - p0 >= 16 ? 0xffffu : (1u << (stop - p0)) - 1);
			// (read now: in pass 2 another segment's walk may put the rest of ITS last match on that position)
			lastd = mm1 ? cand[p0 + 31 - (uint32_t)__builtin_clz(mm1)] : 0;
			if (p > stop && stop == tn) p = tn;
			X[t] = (uint16_t)(p < tn ? p : tn);
			jump[t] = (uint16_t)(p >= tn ? NSEG : p / PSEG);
		}
		if (t == 0) mark[0] = 1;
		__syncthreads();
		PROF(6);
		// ---- chain of entered segments by pointer jumping ----
		// Four hops per round (one barrier per round is what costs): a marked segment marks the
		// segments 1, 2 and 3 jumps ahead, and the jump array is replaced by its fourth power.
		// 4^5 = 1024 segments.
		{
			uint16_t *ja = jump, *jb = entry;                  // entry[] is free until the chain is known
			uint32_t j1 = (uint32_t)t < nseg ? ja[t] : NSEG;   // my own entry stays in a register from round to round
			for (int k = 0; k < 5; k++) {
				if ((uint32_t)t < nseg) {
					const uint32_t j2 = j1 < NSEG ? ja[j1] : NSEG;
					const uint32_t j3 = j2 < NSEG ? ja[j2] : NSEG;
					const uint32_t j4 = j3 < NSEG ? ja[j3] : NSEG;
					if (mark[t]) {
						if (j1 < NSEG) mark[j1] = 1;
						if (j2 < NSEG) mark[j2] = 1;
						if (j3 < NSEG) mark[j3] = 1;
					}
					jb[t] = (uint16_t)j4;
					j1 = j4;
				}
				__syncthreads();
				uint16_t *tmp = ja; ja = jb; jb = tmp;
			}
		}
		// entries: entered segment s hands its exit to the segment that contains it
		bool entered = (uint32_t)t < nseg && mark[t];
		uint32_t myx = (uint32_t)t < nseg ? X[t] : 0;
		if (t == 0) entry[0] = 0;
		if (entered && myx < tn) entry[myx / PSEG] = (uint16_t)myx;
		__syncthreads();
		uint32_t mye = entered ? entry[t] : 0;
		// token bitmaps (alias mark/jump, which are dead now: every thread has its entry) are cleared
		sbits[t] = 0;                                           // tokbits and litbits (adjacent, 2 x 512 words)
		__syncthreads();

		PROF(7);
		// ---- parse pass 2: entered segments walk [entry, X[s]) for real ----
		// The walk only decides: literal and match token starts are collected in two register masks
		// (32 positions from the segment start; beyond that -- a walk that runs on behind a long
		// match of its neighbours -- bit by bit) and a match gets its final, possibly truncated length.
		if (entered) {
			uint32_t p = mye;
			const uint32_t lim = myx;
			uint32_t lm = 0, mm = 0;
			const uint32_t vis1 = lit1 | mm1;
			// the speculative walk's last token, if that is a match: every position behind its start lies inside it
			const uint32_t jm = mm1 ? 31 - (uint32_t)__builtin_clz(mm1) : 0;
			const bool lastm = mm1 != 0 && (lit1 >> jm) <= 1u;     // (bit jm of lit1 is clear)
			while (p < lim) {
				const uint32_t k = p - p0;
				if (k < 16 && ((vis1 >> k) & 1)) {
					// the speculative walk passed here: from now on this IS that walk (oracle/nxz_lz77.c
					// step 5); its tokens end at lim.  Only its last match can have been cut by the end of
					// the tile: that one gets its final length.
					const uint32_t ms = mm1 & (0xffffffffu << k);
					lm |= lit1 & (0xffffffffu << k);
					mm |= ms;
					if (lim == tn && ms) {
						const uint32_t j = 31 - (uint32_t)__builtin_clz(ms), pl = p0 + j;
						if (pl + seg_m(j) + 3 > tn) mlen[pl] = (uint8_t)(tn - pl - 3);
					}
					break;
				}
				if (lastm && k > jm) {
					// stepped over the start of the speculative walk's last match: what is left of that match
					// (same distance, up to lim) is the last token; literals if fewer than 3 bytes are left
					const uint32_t left = lim - p;
					if (left >= 3) {
						if (k < 32) mm |= 1u << k; else atomicOr(&tokbits[p >> 5], 1u << (p & 31));
						mlen[p] = (uint8_t)(left - 3);
						cand[p] = (uint16_t)lastd;
					} else {
						for (uint32_t q = p; q < lim; q++) {
							if (q - p0 < 32) lm |= 1u << (q - p0); else atomicOr(&litbits[q >> 5], 1u << (q & 31));
						}
					}
					break;
				}
				// (here p is inside my segment: in front of the last match's start, or there is none and lim is the segment's end)
				// literals in front of the next stored match of my segment / of the next position the
				// speculative walk visited
				const uint32_t rest = (seg_nz | vis1) >> k;
				uint32_t run = rest ? (uint32_t)__builtin_ctz(rest) : 16 - k;
				if (run > lim - p) run = lim - p;
				if (run) {
					lm |= ((1u << run) - 1) << k;
					p += run;
					continue;
				}
				bool m;
				const uint32_t l = seg_step(p, lim, m);
				lm |= (m ? 0u : 1u) << k;
				mm |= (m ? 1u : 0u) << k;
				if (m) mlen[p] = (uint8_t)(l - 3);
				p += l;
			}
			// 32 positions from p0 = 16 t: bit 16 (t & 1) of word t / 2 onwards
			const uint32_t sh = 16 * ((uint32_t)t & 1), wi = (uint32_t)t >> 1;
			const uint64_t lv = (uint64_t)lm << sh, mv = (uint64_t)mm << sh;
			if ((uint32_t)lv) atomicOr(&litbits[wi], (uint32_t)lv);
			if (lv >> 32) atomicOr(&litbits[wi + 1], (uint32_t)(lv >> 32));
			if ((uint32_t)mv) atomicOr(&tokbits[wi], (uint32_t)mv);
			if (mv >> 32) atomicOr(&tokbits[wi + 1], (uint32_t)(mv >> 32));
		}
		__syncthreads();
		PROF(8);
		if (tb0 == 0 && n > PTILE && t < 512) {
			// tokens of the first tile (oracle/nxz_lz77.c step 3b)
			uint32_t c = (uint32_t)__popc(litbits[t]) + (uint32_t)__popc(tokbits[t]);
			for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
			if (lane == 0) atomicAdd(&misc[M_TOK0], c);
		}

		// ---- out (FUSED): the tile's tokens as fixed-Huffman bits, straight into the target ----
		if constexpr (FUSED) {
			Quad q[4] = { {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0} };
			uint32_t nb = 0;
			if (p0 < tn) {
				const uint32_t tok16 = ((const uint16_t *)tokbits)[t], lit16 = ((const uint16_t *)litbits)[t];
				if (tok16 | lit16) {
					const uint4 bv = *(const uint4 *)(lds + OFF_IN + h + tb0 + p0);      // 16-byte aligned: h, tb0, p0 are
					const uint4 mv = *(const uint4 *)(mlen + p0);
					const uint4 c0 = ((const uint4 *)cand)[2 * t], c1 = ((const uint4 *)cand)[2 * t + 1];
					q[0] = fx_quad(bv.x, lit16 & 15, tok16 & 15, mv.x, c0.x, c0.y);
					q[1] = fx_quad(bv.y, (lit16 >> 4) & 15, (tok16 >> 4) & 15, mv.y, c0.z, c0.w);
					q[2] = fx_quad(bv.z, (lit16 >> 8) & 15, (tok16 >> 8) & 15, mv.z, c1.x, c1.y);
					q[3] = fx_quad(bv.w, lit16 >> 12, tok16 >> 12, mv.w, c1.z, c1.w);
					nb = q[0].nb + q[1].nb + q[2].nb + q[3].nb;
				}
			}
			const uint32_t incl = wave_incl_scan(nb, lane);
			if (lane == 63) scan[wave] = incl;
			__syncthreads();                                       // cand[] and mlen[] have been read: their place is the window now
			uint32_t bitpos = obits + incl - nb, tilebits = 0;
#pragma unroll
			for (int w = 0; w < 16; w++) {
				const uint32_t sc = scan[w];
				if (w < wave) bitpos += sc;
				tilebits += sc;
			}
			uint32_t *win = (uint32_t *)(lds + OFF_CAND);          // <= 16384 x 9 bits + a carried dword: 18.5 KiB of the 48 KiB
			const uint32_t tot = obits + tilebits, nfull = tot >> 5;
			for (uint32_t i = t; i < (nfull + 8) / 4 + 1; i += NT) ((uint4 *)win)[i] = make_uint4(0, 0, 0, 0);
			__syncthreads();
			if (t == 0) atomicOr(&win[0], misc[M_KEEP]);           // the partial dword the tile before left
			fx_emit(win, q[0], bitpos);
			fx_emit(win, q[1], bitpos + q[0].nb);
			fx_emit(win, q[2], bitpos + q[0].nb + q[1].nb);
			fx_emit(win, q[3], bitpos + q[0].nb + q[1].nb + q[2].nb);
			__syncthreads();
			// whole dwords leave; the last, partial one opens the next tile's window
			for (uint32_t i = t; i < nfull; i += NT)
				if (wordbase + i < cap_words) dstw[wordbase + i] = win[i];
			if (t == 0) misc[M_KEEP] = win[nfull];
			wordbase += nfull;
			obits = tot & 31;
			__syncthreads();                                       // the window is cand[] again; scan[] and the bitmaps are reused by the next tile
		} else
		// ---- out: records, counts, bitmaps ----
		{
			const uint32_t tok16 = p0 < tn ? ((const uint16_t *)tokbits)[t] : 0;
			const uint32_t nm = (uint32_t)__popc(tok16);
			uint32_t incl = wave_incl_scan(nm, lane);
			if (lane == 63) scan[wave] = incl;
			const uint32_t recbase = misc[M_NREC];
			__syncthreads();
			uint32_t rank = recbase + incl - nm, tot = 0;
#pragma unroll
			for (int w = 0; w < 16; w++) {
				const uint32_t sc = scan[w];
				if (w < wave) rank += sc;
				tot += sc;
			}
			for (uint32_t mbits_ = tok16; mbits_; mbits_ &= mbits_ - 1) {
				const uint32_t p = p0 + (uint32_t)__builtin_ctz(mbits_);
				const uint32_t l3 = mlen[p], d = cand[p];
				if (rank < NXZ_TOK_MAXREC) g_rec[rank] = l3 | (d << 8);
				rank++;
				if (COUNT) {
					uint32_t le = l3 < 8 ? 0 : (29 - (uint32_t)__builtin_clz(l3 | 8));
					const uint32_t ls = l3 == 255 ? 28 : (le << 2) + (l3 >> le);
					const uint32_t de = d < 4 ? 0 : (30 - (uint32_t)__builtin_clz(d | 4));
					const uint32_t ds = d < 4 ? d : 2 * de + 2 + ((d >> de) & 1);
					atomicAdd(&hist[257 + ls], 1u);
					atomicAdd(&hist[286 + ds], 1u);
				}
			}
			if (COUNT && p0 < tn) {
				const uint32_t lit16 = ((const uint16_t *)litbits)[t];
				if (lit16) {
					const uint4 bv = *(const uint4 *)(lds + OFF_IN + h + tb0 + p0);      // 16-byte aligned: h, tb0, p0 are
					for (uint32_t lb = lit16; lb; lb &= lb - 1) {
						const uint32_t k = (uint32_t)__builtin_ctz(lb);
						const uint32_t a = (k & 8) ? bv.z : bv.x, b = (k & 8) ? bv.w : bv.y;
						const uint32_t w = (k & 4) ? b : a;
						atomicAdd(&hist[(w >> (8 * (k & 3))) & 0xff], 1u);
					}
				}
			}
			// the bitmaps of this tile, coalesced: the first 512 threads the literal bits, the others the match bits
			{
				const uint32_t nwords = (tn + 31) >> 5, i = (uint32_t)t & 511;
				if (i < nwords) {
					if (t < 512) g_lit[(tb0 >> 5) + i] = litbits[i];
					else g_tok[(tb0 >> 5) + i] = tokbits[i];
				}
			}
			__syncthreads();                                       // scan[] and the bitmaps are reused by the next tile
			if (t == 0) misc[M_NREC] = recbase + tot;
		}
		PROF(9);
	}

	// ---------------- result: checksums and counts (the entropy stage adds cc, tpbc, tebc) ----------------
	if (t == 0) {
		nxz_batch_result_t r;
		r.cc = 0; r.tpbc = 0; r.tebc = 0;
		r.spbc = total; r.crc = out_crc; r.adler = out_adler; r.subc = 0;
		r.sfbt = misc[M_NREC];                                  // match tokens (diagnostic; the entropy stage checks it against the record array's size)
		if constexpr (FUSED) {
			// end of block (seven zero bits) behind what the tiles left, the last bytes, the completion code
			// (as the entropy kernel sets them: nxz_encode.hip)
			const uint64_t acc = misc[M_KEEP];
			const uint32_t bits = obits + 7;
			const uint64_t totbits = (uint64_t)wordbase * 32 + bits;
			const uint32_t tpbc = (uint32_t)((totbits + 7) >> 3);
			uint32_t cc = wordbase > cap_words || tpbc > job.dst_cap ? NXZ_CC_TARGET_SPACE : 0;
			if (!cc) {
				NXZ_GLOBAL uint8_t *o = (NXZ_GLOBAL uint8_t *)job.dst + (size_t)wordbase * 4;
				for (uint32_t b = 0; b < (bits + 7) / 8; b++) o[b] = (uint8_t)(acc >> (8 * b));
				if (tpbc > total) cc = NXZ_CC_TPBC_GT_SPBC;
			}
			r.cc = cc; r.tpbc = cc == NXZ_CC_TARGET_SPACE ? 0 : tpbc; r.tebc = (uint32_t)(totbits & 7); r.sfbt = 0;
		}
		results[bid] = r;
	}
#ifdef NXZ_LZ77_PROF
	if (prof) {
		// (one atomic per counter and job: thousands of waves adding to four addresses would stall every
		// wave's next loads behind them)
		if (lane == 0) for (int k = 0; k < 4; k++) atomicAdd(&profacc[16 + k], (uint32_t)wacc[k]);
		__syncthreads();
		if (t < 20) __hip_atomic_fetch_add(&prof[t], (unsigned long long)profacc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
#endif
	if (COUNT && (!GEN || counts)) {
		__syncthreads();
		if (t < 316) counts[(size_t)bid * 316 + t] = (t == 256) ? 1u : hist[t];
	}
	if constexpr (GEN) {
		// the table of this job (wavefront 0) while the job before it is encoded (wavefronts 1-15)
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__syncthreads();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		const uint32_t prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)misc[M_PREV]);
		gen::post(lds, jobs, (NXZ_GLOBAL uint8_t *)tokens, results, true, gslot, prev, gslot ^ 1u, t);
		if (t == 0) { misc[M_PREV] = bid + 1; misc[M_JOBNO] = misc[M_JOBNO] + 1; }
	}
	__syncthreads();                                           // the LDS image is reused by the next job
	const uint32_t drawn = next_job ? misc[M_NEXT] : bid + gridDim.x;
	__syncthreads();                                           // misc[M_NEXT] is rewritten at the top
	bid = drawn;
	}
	if constexpr (GEN) {
		// the workgroup's last job
		const uint32_t prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)misc[M_PREV]);
		const uint32_t slot = ((uint32_t)__builtin_amdgcn_readfirstlane((int)misc[M_JOBNO]) & 1u) ^ 1u;
		if (prev) gen::post(lds, jobs, (NXZ_GLOBAL uint8_t *)tokens, results, false, 0, prev, slot, (int)threadIdx.x);
	}
}

} // namespace nxzl77

extern "C" int nxz_lz77_prof_set(unsigned long long *buf)
{
#ifndef NXZ_LZ77_PROF
	if (buf) return -1;                                        // this build carries no counters
#endif
	return (int)hipMemcpyToSymbol(HIP_SYMBOL(nxz_lz77_prof_buf), &buf, sizeof(buf));
}

// One persistent workgroup per CU; job_counter: one device word per launch in flight (or NULL:
// workgroups stride over the jobs).  tokens: n x NXZ_TOK_STRIDE bytes of device scratch.
// Device scratch a launch needs for the second bucket entries in transit (one tile per workgroup).
// ... and for the fused dynamic-Huffman form: two token slots and two table slots per workgroup
extern "C" size_t nxz_lz77_gen_scratch_bytes(void) { return nxzl77::gen::table_off(NXZ_LZ77_MAX_GRID) + (size_t)NXZ_LZ77_MAX_GRID * 2 * sizeof(nxz_dht_prepared_t); }
extern "C" size_t nxz_lz77_cand2_bytes(void) { return (size_t)NXZ_LZ77_MAX_GRID * nxzl77::C2_STRIDE * sizeof(uint16_t); }

extern "C" int nxz_launch_lz77(int count, const nxz_batch_job_t *jobs, size_t n, uint8_t *tokens, uint16_t *cand2, nxz_batch_result_t *results,
			       uint32_t *counts, uint32_t *job_counter, hipStream_t stream)
{
	using namespace nxzl77;
	if (n == 0) return 0;
	// count: 0 tokens for the entropy kernel, 1 tokens + symbol counts, 2 (NXZ_LZ77_FUSED_FHT) the finished fixed-Huffman block,
	// 3 (NXZ_LZ77_FUSED_GEN) the finished dynamic-Huffman block with the table of its own counts (`tokens`: nxz_lz77_gen_scratch_bytes())
	void (*k)(const nxz_batch_job_t *, uint8_t *, uint16_t *, nxz_batch_result_t *, uint32_t *, uint32_t, uint32_t *);
	k = count == 3 ? lz77_kernel<true, false, true> : count == 2 ? lz77_kernel<false, true> : count ? lz77_kernel<true> : lz77_kernel<false>;
	// per device: the attribute belongs to the loaded code object of a device, and so does the CU count
	static int ncu_of[64];
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
	int ncu = __atomic_load_n(&ncu_of[dev], __ATOMIC_ACQUIRE);
	if (!ncu) {
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
		if (ncu <= 0) ncu = 256;
		if (ncu > NXZ_LZ77_MAX_GRID) ncu = NXZ_LZ77_MAX_GRID;
		__atomic_store_n(&ncu_of[dev], ncu, __ATOMIC_RELEASE);
	}
	const unsigned grid = (unsigned)(n < (size_t)ncu ? n : (size_t)ncu);
	if (n <= grid) job_counter = nullptr;                      // one job per workgroup: nothing to draw
	if (job_counter && hipMemsetAsync(job_counter, 0, sizeof(uint32_t), stream) != hipSuccess) job_counter = nullptr;
	hipLaunchKernelGGL(k, dim3(grid), dim3(NT), 0, stream, jobs, tokens, cand2, results, counts, (uint32_t)n, job_counter);
	return (int)hipGetLastError();
}
