// nxz_inflate_cut.hip -- a batch of deflate streams that is too small to fill the device with a stream per wavefront:
// every stream is cut inside its blocks, at token boundaries, and the pieces are decoded side by side.
//
// The reference's engine takes one job at a time and is fast on ONE stream (/root/reference lib/nx_inflate.c:1143-1762,
// samples/compdecomp_th.c:155-222: T threads, a call each); here a wavefront decodes one stream at 20-100 MB/s, so a
// batch of 4096 streams of 64 KiB -- or the sixteen callers a round of nxu_run_job gathers -- takes as long as its
// slowest stream, 1-8 ms, with nine tenths of the device idle.  The machinery that cuts ONE long stream
// (nxz_pinflate.cpp, driven by the host: a dozen launches, four waits) is used here per stream and WITHOUT the host: a
// fixed sequence of launches on the caller's stream, every decision taken by small kernels in between.  A stream is
// worked through in ROUNDS, each of which covers a stretch of it -- about a block, whose end nobody knows beforehand:
//
//   plan      a thread per stream: where the stream stands (a block header, or inside a dynamic block whose table the last
//             round's last piece handed on), how far this round looks, where in that stretch to look for cuts
//   tables    nxz_inflate.hip block_tables_kernel: that block's decode tables, a wavefront per stream
//   sync      nxz_inflate.hip token_sync_kernel: a token boundary behind every guessed bit (64 lanes fall in step)
//   jobs      a thread per stream: the cuts that were found become pieces -- jobs that resume inside the block with its
//             tables and stop at the next cut -- with room for their 16-bit elements from a bump arena
//   decode    nxz_inflate.hip inflate_kernel<true, true>: the round's pieces of all streams at once
//   check     a thread per stream: every piece must have arrived exactly at the next one's start, in the block it began in;
//             the pieces behind the first that did not are dropped (a guess behind the block's end: the piece in front has
//             read the next header and decoded on, with the right tables) and the stream stands where that one stopped
//
// then, whatever the rounds have left of a stream, as ONE more piece (decode), and
//
//   resolve   a workgroup per stream: the pieces' elements -- a byte, or "byte k of the 32 KiB in front of this piece" --
//             become bytes in the caller's target, piece after piece; the result record a single job would have left
//   plain     nxz_inflate.hip inflate_kernel<true>: whatever was not cut (a first block that is stored or fixed-code, a
//             short stream) or did not work out (an error in a piece, a target that is too small: the plain kernel reports
//             those as the oracle does), a wavefront per stream
//   cksum     nxz_inflate_lanes.hip cksum_kernel over all outputs
//
// Results are those of the stream-per-wave kernel, field for field (tests/test_gpu_parity.py runs every inflate case
// through this route as well).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "nxz_device.h"
#include "nxz_inflate_tables.h"

extern "C" int nxz_launch_inflate_w16_order(const nxz_batch_job_t *jobs, size_t nslots, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io, const void *built,
					    const uint32_t *order, hipStream_t stream);

namespace nxzc {

constexpr uint32_t PMAX = 32;                  // pieces a stream is cut into per round at most
constexpr uint32_t RMAX = 8;                   // rounds at most
constexpr uint32_t PIECE_MIN_BYTES = 512;      // bytes of the stream a piece should have at least
constexpr uint32_t CUT_MIN_SRC = 2048;         // streams shorter than this are left alone
constexpr uint32_t MIN_GAP_BITS = 1024;        // cuts closer together than this are one
constexpr uint32_t ELEM_FLOOR = 4096;          // elements of room a piece gets at least
// bytes of a stream the first round looks at: a zlib block of 16 Ki tokens is 10-25 KiB, this engine's own blocks hold up to 64 KiB
// of data.  Few streams (all their pieces resident at once: what counts is the number of rounds) look further than many.
constexpr uint32_t EXT_FIRST = 24u << 10, EXT_FIRST_FEW = 72u << 10;

enum { ST_PLAIN = 0, ST_ACTIVE = 1, ST_DONE = 2, ST_FAILED = 3, ST_REST = 4 };

struct Ctl {
	uint32_t state;
	uint32_t cur_bit;          // where the stream stands, in bits from its first byte
	uint32_t sfbt, rem;        // ... and how: the resume fields a job suspended there reports (0xe: at a block header)
	uint32_t ext;              // bytes of the stream the next round covers
	uint32_t made;             // elements the pieces so far have made
	uint32_t npieces;          // pieces whose output stands: slots 0 .. npieces - 1 of the stream
	uint32_t np_round;         // pieces of the round in work: slots npieces .. npieces + np_round - 1
	uint32_t want;             // cuts the round in work looks for, + 1
	uint32_t round_bfinal;     // BFINAL of the block the round in work began in
	uint32_t boost;            // a piece outgrew its room: the next ones get 4^boost times as much
	uint32_t fin, fin_cc, fin_subc, fin_spbc;     // the stream's final block ended inside a piece that does not reach the end of the source: the stream's result
	uint32_t why;              // (diagnostic: what made a stream ST_FAILED)
	uint32_t round_hi;         // the bit up to which the round in work looks
	uint32_t pad[3];
};
static_assert(sizeof(Ctl) == 80, "Ctl");

struct Arena { unsigned long long used, size; uint32_t npieces, nrest; };

__device__ __forceinline__ uint32_t hist_of(const nxz_batch_job_t &j) { return j.hist_len < j.src_len ? j.hist_len : j.src_len; }

// ---- plan: what this round looks at ----
__global__ void plan_kernel(const nxz_batch_job_t *__restrict__ jobs, uint32_t n, uint32_t P, uint32_t PT, uint32_t round, nxz_sync_req_t *__restrict__ bq,
			    nxz_sync_req_t *__restrict__ rq, Ctl *__restrict__ ctl, Arena *__restrict__ arena, unsigned long long arena_size,
			    uint32_t *__restrict__ order, uint32_t *__restrict__ order2, const nxz_batch_dht_t *__restrict__ dht_io, nxz_batch_dht_t *__restrict__ tb)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i == 0) {
		if (round == 0) { arena->used = 0; arena->size = arena_size; arena->nrest = 0; }
		arena->npieces = 0;
	}
	if (i >= n) return;
	// (the lists of pieces to decode are filled from the front by whoever has some: empty before anyone does)
	for (uint32_t j = 0; j < P; j++) order[(size_t)i * P + j] = 0xffffffffu;
	if (round == 0) order2[i] = 0xffffffffu;
	const nxz_batch_job_t job = jobs[i];
	const uint32_t hb = hist_of(job), srclen = job.src_len - hb, bits = srclen * 8;
	const uint8_t *S = job.src + hb;
	Ctl c;
	if (round == 0) {
		const uint32_t in_subc = (job.resume >> 20) & 7, in_sfbt = (job.resume >> 16) & 15;
		// at a block header (a fresh stream, or one that was suspended there) or inside a dynamic block whose table the job brings (a
		// part of a stream that a caller of inflate() feeds in steps); its bytes where the piece kernel can load 16 a lane
		const bool inside = (in_sfbt & 0xe) == 0xc && dht_io != nullptr;
		const bool ok = srclen >= CUT_MIN_SRC && srclen < (1u << 28) && ((uintptr_t)S & 15) == 0 && (in_sfbt == 0 || (in_sfbt & 0xe) == 0xe || inside) && job.dst_cap >= 1024;
		c = Ctl();
		c.state = ok ? ST_ACTIVE : ST_PLAIN;
		c.cur_bit = srclen && in_subc ? 8 - in_subc : 0;
		c.sfbt = inside ? in_sfbt : 0xe;
		if (ok && inside) tb[i] = dht_io[i];
		const uint32_t ext0 = n * P <= 2048 ? EXT_FIRST_FEW : EXT_FIRST;
		c.ext = srclen < ext0 ? srclen : ext0;
	} else c = ctl[i];
	c.np_round = 0; c.want = 0;
	nxz_sync_req_t b;
	b.src = S; b.srclen = 0; b.header_bit = 0; b.guess_bit = 0; b.limit_bit = 0;
	uint32_t want = 0, lo = 0, hi = 0;
	if (c.state == ST_ACTIVE) {
		lo = c.cur_bit;
		const uint64_t h64 = (uint64_t)lo + (uint64_t)c.ext * 8;
		hi = h64 < bits ? (uint32_t)h64 : bits;
		if (bits - hi < PIECE_MIN_BYTES * 8) hi = bits;               // (a little more is not worth a round of its own)
		want = (hi - lo) / (PIECE_MIN_BYTES * 8);
		if (want > P) want = P;
		if (want < 1) want = 1;
		const uint32_t kind = c.sfbt & 0xe;
		if (kind == 0xe) { b.srclen = srclen; b.header_bit = lo; }                    // the tables of the block that starts here
		else if (kind == 0xc) { b.srclen = srclen; b.header_bit = 0xffffffffu; }      // ... of the block the stream stands in: from the table in its slot
		else want = 1;                                                                 // inside a stored or fixed-code block: one piece goes on
		c.want = want;
	}
	ctl[i] = c;
	bq[i] = b;
	for (uint32_t k = 1; k < P; k++) {
		nxz_sync_req_t r;
		r.src = S; r.srclen = srclen; r.header_bit = i; r.guess_bit = 0; r.limit_bit = 0;
		if (k < want) { r.guess_bit = lo + (uint32_t)(((uint64_t)(hi - lo) * k) / want); r.limit_bit = bits; }
		rq[(size_t)i * (P - 1) + (k - 1)] = r;
	}
}

// ---- jobs: the cuts that were found become pieces ----
// the pieces of stream i: slots i * PT + k
__global__ void jobs_kernel(const nxz_batch_job_t *__restrict__ jobs, uint32_t n, uint32_t P, uint32_t PT, uint32_t round, const nxz_sync_res_t *__restrict__ rs,
			    const nxz_batch_dht_t *__restrict__ tb, const nxzi::Built *__restrict__ bt, Ctl *__restrict__ ctl,
			    nxz_batch_job_t *__restrict__ pj, nxz_batch_dht_t *__restrict__ pd, uint8_t *__restrict__ elems, Arena *__restrict__ arena,
			    uint32_t *__restrict__ order)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	Ctl c = ctl[i];
	if (c.state != ST_ACTIVE) return;
	const nxzi::Built *B = &bt[i];
	const uint32_t kind = c.sfbt & 0xe;
	const bool coded = (kind == 0xe || kind == 0xc) && B->ok;        // the round's block is a dynamic one, and its tables are made
	if (round == 0 && !coded) { c.state = ST_PLAIN; ctl[i] = c; return; }   // a stream that begins with a stored or fixed-code block: the plain kernel's
	const nxz_batch_job_t job = jobs[i];
	const uint32_t hb = hist_of(job), srclen = job.src_len - hb, bits = srclen * 8;
	const uint8_t *S = job.src + hb;
	const uint32_t lo = c.cur_bit;
	const uint64_t h64 = (uint64_t)lo + (uint64_t)c.ext * 8;
	uint32_t hi = h64 < bits ? (uint32_t)h64 : bits;
	if (bits - hi < PIECE_MIN_BYTES * 8) hi = bits;
	uint32_t cuts[PMAX + 1];
	uint32_t nc = 1;
	cuts[0] = lo;
	const uint32_t first_ok = kind == 0xe ? B->end_bit + 64 : lo + MIN_GAP_BITS;    // (a cut in the header is none)
	for (uint32_t k = 1; coded && k < c.want; k++) {
		const uint32_t b = rs[(size_t)i * (P - 1) + (k - 1)].bit;
		if (b == 0xffffffffu || b < first_ok || b < cuts[nc - 1] + MIN_GAP_BITS || b + MIN_GAP_BITS > hi) continue;
		cuts[nc++] = b;
	}
	if (c.npieces + nc > PT - 1) { c.state = ST_REST; ctl[i] = c; return; }      // (no slots left: the rest in one piece)
	// No cut where several were looked for: data whose tokens are all of one length (the 8-bit literals of what hardly
	// compresses) never lets decoders that start at neighbouring bits fall in step.  Such a stream is decoded in ONE piece,
	// to its end, right away -- rounds of one slow piece each would hold every other stream of the batch up once per round.
	if (coded && nc == 1 && c.want >= 4) hi = bits;
	c.round_hi = hi;
	const uint32_t bfinal = kind == 0xe ? B->bfinal : c.sfbt & 1;
	// room for the pieces' elements: two and a half times a piece's share of what the caller's target says the stream can make, and
	// four times that for every time a piece of this stream has outgrown its room
	uint32_t cap[PMAX];
	unsigned long long tot = 0;
	const uint32_t left = job.dst_cap > c.made ? job.dst_cap - c.made : 0;
	for (uint32_t j = 0; j < nc; j++) {
		const uint32_t end = j + 1 < nc ? cuts[j + 1] : hi;
		const unsigned long long sb = (end - cuts[j]) / 8 + 32;
		unsigned long long e = (sb * job.dst_cap * 5 / 2) / srclen + 2048;
		if (e > sb * 160) e = sb * 160;                              // (a caller that names a target far larger than the stream can fill)
		e <<= 2 * (c.boost < 8 ? c.boost : 8);
		if (e < ELEM_FLOOR) e = ELEM_FLOOR;
		if (e > left) e = left;
		cap[j] = ((uint32_t)e + 127) & ~127u;
		tot += (unsigned long long)cap[j] * 2;
	}
	const unsigned long long off = atomicAdd(&arena->used, tot);
	if (off + tot > arena->size) { c.state = c.npieces ? ST_FAILED : ST_PLAIN; c.why = 1; ctl[i] = c; return; }   // no room: the plain kernel
	const size_t base = (size_t)i * PT + c.npieces;
	unsigned long long at = off;
	for (uint32_t j = 0; j < nc; j++) {
		nxz_batch_job_t p;
		const uint32_t cstart = (cuts[j] >> 3) & ~15u;
		const uint32_t cend = j + 1 < nc ? (cuts[j + 1] + 7) >> 3 : (hi + 7) >> 3;
		p.src = S + cstart; p.src_len = cend - cstart;
		p.hist_len = cuts[j] - cstart * 8;                          // (a piece: the bit it starts at)
		p.dst = elems + at; p.dst_cap = cap[j];
		at += (unsigned long long)cap[j] * 2;
		p.in_adler = j + 1 < nc ? cuts[j + 1] - cstart * 8 : 0;     // (the bit the next piece starts at; the round's last piece runs out of source)
		p.dht_index = 0; p.reserved = 0;
		if (j == 0) {
			p.resume = (c.rem & 0xffff) | (c.sfbt & 0xf) << 16;
			p.in_crc = kind == 0xc && B->ok ? i + 1 : 0;
		} else {
			p.resume = (0xcu | bfinal) << 16;
			p.in_crc = i + 1;                                       // (which block's ready-made tables)
		}
		pj[base + j] = p;
		// (the table of the block a piece starts in: what it hands on if it stops in that block)
		if (j || kind == 0xc) pd[base + j] = tb[i];
	}
	c.np_round = nc; c.round_bfinal = bfinal;
	ctl[i] = c;
	const uint32_t slot = atomicAdd(&arena->npieces, nc);
	for (uint32_t j = 0; j < nc; j++) order[slot + j] = (uint32_t)(base + j);
}

// ---- check: every piece must have arrived at the next one; where the stream stands now ----
__global__ void check_kernel(const nxz_batch_job_t *__restrict__ jobs, uint32_t n, uint32_t PT, Ctl *__restrict__ ctl, nxz_batch_dht_t *__restrict__ tb,
			     const nxz_batch_job_t *__restrict__ pj, const nxz_batch_result_t *__restrict__ pr, const nxz_batch_dht_t *__restrict__ pd)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	Ctl c = ctl[i];
	if (c.state != ST_ACTIVE || !c.np_round) return;
	const nxz_batch_job_t job = jobs[i];
	const uint32_t hb = hist_of(job), srclen = job.src_len - hb, bits = srclen * 8;
	const uint8_t *S = job.src + hb;
	const size_t base = (size_t)i * PT + c.npieces;
	const uint32_t from = c.cur_bit, started_at_header = (c.sfbt & 0xe) == 0xe;
	const uint32_t hi = c.round_hi;                                     // (how far the round looked)
	uint32_t valid = 0;
	bool all = true, at_end = false;
	for (uint32_t j = 0; j < c.np_round; j++) {
		const nxz_batch_result_t r = pr[base + j];
		const nxz_batch_job_t p = pj[base + j];
		const uint32_t cstart = (uint32_t)(p.src - S);
		if (r.cc == NXZ_CC_TARGET_SPACE) {
			// it outgrew its room: what is in front of it stands, the next round begins where it did, with more room
			c.boost++;
			all = false;
			if (j == 0) break;                                        // (the stream stands where it stood)
			// (the stream's table slot holds the round's block's table: the piece in front arrived in step, in that block; this
			// piece's own slot may hold the NEXT block's by now, if it read on into it before it ran out of room)
			c.cur_bit = cstart * 8 + p.hist_len; c.sfbt = 0xc | c.round_bfinal; c.rem = 0;
			break;
		}
		if (r.cc != 0 && r.cc != NXZ_CC_DATA_LENGTH) { c.state = ST_FAILED; c.why = 2 | r.cc << 8; ctl[i] = c; return; }   // (bad data: the plain kernel says what)
		valid = j + 1;
		c.made += r.tpbc;
		const uint32_t stop = p.in_adler ? p.in_adler : p.src_len * 8;  // (bits from the piece's first byte it could use)
		if (r.sfbt & 0x100) {
			// the stream's final block ended inside this piece: what lies behind is not the stream's.  The result a job over the
			// whole source would have left (nxz_inflate.hip: the unused bits count from the end of the source)
			const uint32_t pos = cstart * 8 + stop - r.subc;
			uint32_t subc = bits - pos, spbc = job.src_len;
			if (subc > 0xfff8) { const uint32_t drop = (subc - 0xfff8 + 7) / 8; spbc -= drop; subc -= drop * 8; }
			c.fin = 1; c.fin_cc = subc < 8 ? 0 : NXZ_CC_DATA_LENGTH; c.fin_subc = subc; c.fin_spbc = spbc;
			c.state = ST_DONE;
			break;
		}
		// where the stream stands behind this piece
		c.cur_bit = cstart * 8 + stop - r.subc;
		c.sfbt = (r.sfbt & 0xf) ? (r.sfbt & 0xf) : 0xe; c.rem = r.tebc;
		if ((r.sfbt & 0xe) == 0xc) tb[i] = pd[base + j];
		if (j + 1 == c.np_round) { at_end = hi == bits; break; }          // (the round's last piece: it ran out of source -- the stream's, if the round looked that far)
		// in step with the next piece: suspended inside the block the round began in (it has read that block's header and no
		// other, or none if it began inside the block), right at the next piece's first bit
		const bool arrived = r.cc == NXZ_CC_DATA_LENGTH && (r.sfbt & 0xe) == 0xc && (r.sfbt & 1) == c.round_bfinal && r.subc == 0 &&
				     r.adler == (j == 0 && started_at_header ? 1u : 0u);
		if (!arrived) { all = false; break; }                         // the next cut is none: a guess behind the end of the block
	}
	c.npieces += valid;
	c.np_round = 0;
	if (c.state == ST_ACTIVE) {
		if (at_end || c.cur_bit >= bits) {
			// the source is used up: the last piece's result is the stream's
			c.state = ST_DONE;
		} else if (c.made >= job.dst_cap) { c.state = ST_FAILED; c.why = 3; }      // (a full target: the plain kernel knows what to do about it)
		else {
			const uint32_t used = (c.cur_bit - from) / 8;
			uint64_t e = all ? (uint64_t)c.ext * 3 / 2 : (uint64_t)used * 9 / 8;
			if (e < 4096) e = 4096;
			c.ext = e > srclen ? srclen : (uint32_t)e;
		}
	}
	ctl[i] = c;
}

// ---- rest: what the rounds have left of a stream, as one piece ----
__global__ void rest_kernel(const nxz_batch_job_t *__restrict__ jobs, uint32_t n, uint32_t PT, Ctl *__restrict__ ctl, const nxz_batch_dht_t *__restrict__ tb,
			    nxz_batch_job_t *__restrict__ pj, nxz_batch_dht_t *__restrict__ pd, uint8_t *__restrict__ elems, Arena *__restrict__ arena,
			    uint32_t *__restrict__ order2)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	Ctl c = ctl[i];
	if (c.state != ST_ACTIVE && c.state != ST_REST) return;
	const nxz_batch_job_t job = jobs[i];
	const uint32_t hb = hist_of(job), srclen = job.src_len - hb;
	const uint8_t *S = job.src + hb;
	if (c.npieces == 0 || c.npieces >= PT || c.made >= job.dst_cap) { c.state = c.npieces ? ST_FAILED : ST_PLAIN; c.why = 4; ctl[i] = c; return; }
	nxz_batch_job_t q;
	const uint32_t X = c.cur_bit, qstart = (X >> 3) & ~15u;
	q.src = S + qstart; q.src_len = srclen - qstart; q.hist_len = X - qstart * 8;
	unsigned long long e = job.dst_cap - c.made;
	const unsigned long long by_ratio = ((unsigned long long)q.src_len + 32) * 160 << 2 * (c.boost < 4 ? c.boost : 4);
	if (e > by_ratio) e = by_ratio;
	if (e < ELEM_FLOOR) e = ELEM_FLOOR;
	const uint32_t cap = ((uint32_t)e + 127) & ~127u;
	const unsigned long long off = atomicAdd(&arena->used, (unsigned long long)cap * 2);
	if (off + (unsigned long long)cap * 2 > arena->size) { c.state = ST_FAILED; c.why = 5; ctl[i] = c; return; }
	q.dst = elems + off; q.dst_cap = cap;
	q.in_adler = 0; q.in_crc = 0; q.dht_index = 0; q.reserved = 0;
	q.resume = (c.rem & 0xffff) | (c.sfbt & 0xf) << 16;
	const size_t slot = (size_t)i * PT + c.npieces;
	pj[slot] = q;
	if ((c.sfbt & 0xe) == 0xc) pd[slot] = tb[i];
	c.state = ST_REST;
	ctl[i] = c;
	order2[atomicAdd(&arena->nrest, 1u)] = (uint32_t)slot;
}

// ---- resolve: elements -> bytes, piece after piece; the stream's result ----
__global__ __launch_bounds__(256) void resolve_kernel(const nxz_batch_job_t *__restrict__ jobs, uint32_t PT, Ctl *__restrict__ ctl,
						      const nxz_batch_job_t *__restrict__ pj, const nxz_batch_result_t *__restrict__ pr,
						      const nxz_batch_dht_t *__restrict__ pd, nxz_batch_result_t *__restrict__ results,
						      nxz_batch_dht_t *__restrict__ dht_io, uint32_t *__restrict__ plain)
{
	__shared__ uint32_t s_off[PMAX * RMAX + 3], s_len[PMAX * RMAX + 3];
	__shared__ uint32_t s_np, s_bad;
	const uint32_t i = blockIdx.x, t = threadIdx.x;
	const Ctl c = ctl[i];
	if (c.state != ST_DONE && c.state != ST_REST) { if (t == 0) plain[i] = i; return; }
	const nxz_batch_job_t job = jobs[i];
	const uint32_t hb = hist_of(job);
	const size_t base = (size_t)i * PT;
	if (t == 0) {
		uint32_t np = c.npieces, at = 0, bad = 0;
		if (c.state == ST_REST) {
			const nxz_batch_result_t r = pr[base + np];
			if (r.cc != 0 && r.cc != NXZ_CC_DATA_LENGTH) bad = 1;      // (an error, or a target that is too small: the plain kernel's to report)
			np++;
		}
		for (uint32_t k = 0; k < np; k++) { s_off[k] = at; s_len[k] = pr[base + k].tpbc; at += s_len[k]; if (at > job.dst_cap) bad = 1; }
		s_off[np] = at;
		s_np = np; s_bad = bad || np == 0;
	}
	__syncthreads();
	if (s_bad) { if (t == 0) plain[i] = i; return; }
	const uint32_t np = s_np;
	uint8_t *dst = job.dst;
	const uint8_t *hist_end = job.src + hb;
	uint32_t bad = 0;
	for (uint32_t k = 0; k < np; k++) {
		const uint16_t *el = (const uint16_t *)pj[base + k].dst;
		const uint32_t off = s_off[k], len = s_len[k];
		// eight elements a lane and trip (the pieces' buffers are 256-byte aligned)
		for (uint32_t e0 = t * 8; e0 < len; e0 += 256 * 8) {
			uint32_t v[8];
			if (e0 + 8 <= len) {
				const uint4 q = *(const uint4 *)(el + e0);
				v[0] = q.x & 0xffff; v[1] = q.x >> 16; v[2] = q.y & 0xffff; v[3] = q.y >> 16; v[4] = q.z & 0xffff; v[5] = q.z >> 16; v[6] = q.w & 0xffff; v[7] = q.w >> 16;
			} else {
				for (uint32_t x = 0; x < 8; x++) v[x] = e0 + x < len ? el[e0 + x] : 0;
			}
			for (uint32_t x = 0; x < 8 && e0 + x < len; x++) {
				uint32_t b = v[x];
				if (b & 0x8000u) {
					// byte k of the 32 KiB in front of this piece: in the target, or in the history in front of the source
					const uint32_t back = 32768u - (b & 0x7fffu);           // 1 .. 32768 bytes in front of the piece's first
					if (back <= off) b = dst[off - back];
					else if (back - off <= hb) b = hist_end[-(ptrdiff_t)(back - off)];
					else { bad = 1; b = 0; }                               // (further back than anything there is: a damaged stream)
				}
				dst[off + e0 + x] = (uint8_t)b;
			}
		}
		__threadfence_block();
		__syncthreads();
	}
	if (__syncthreads_or((int)bad)) { if (t == 0) plain[i] = i; return; }
	if (t == 0) {
		const uint32_t last = np - 1;
		const nxz_batch_result_t r = pr[base + last];
		nxz_batch_result_t o;
		o.tpbc = s_off[np]; o.tebc = r.tebc; o.crc = 0; o.adler = 0;
		o.sfbt = r.sfbt;
		if (c.fin) { o.cc = c.fin_cc; o.subc = c.fin_subc; o.spbc = c.fin_spbc; }
		else {
			// the stream's last piece: its source ends where the stream's does, so what it says of the bits it did not use
			// holds for the stream; the bytes it was given count from its own first
			const uint32_t cstart = (uint32_t)(pj[base + last].src - (job.src + hb));
			o.cc = r.cc; o.subc = r.subc; o.spbc = hb + cstart + r.spbc;
		}
		results[i] = o;
		if ((r.sfbt & 0xe) == 0xc && dht_io) dht_io[i] = pd[base + last];
		plain[i] = 0xffffffffu;
	}
}

// the outputs of a batch whose targets were device memory, to where the callers want them (pinned host memory: nxu_run_job's rounds)
__global__ __launch_bounds__(256) void copy_out_kernel(const nxz_batch_job_t *__restrict__ jobs, const nxz_batch_result_t *__restrict__ results,
							uint8_t *const *__restrict__ targets)
{
	const uint32_t i = blockIdx.x, t = threadIdx.x;
	const nxz_batch_result_t r = results[i];
	if (r.cc != 0 && r.cc != NXZ_CC_DATA_LENGTH) return;
	const uint8_t *sp = jobs[i].dst;
	uint8_t *dp = targets[i];
	const uint32_t n = r.tpbc;
	if ((((uintptr_t)sp | (uintptr_t)dp) & 15) == 0) {
		const uint32_t nv = n >> 4;
		for (uint32_t k = t; k < nv; k += 256) ((uint4 *)dp)[k] = ((const uint4 *)sp)[k];
		for (uint32_t k = (nv << 4) + t; k < n; k += 256) dp[k] = sp[k];
	} else for (uint32_t k = t; k < n; k += 256) dp[k] = sp[k];
}

} // namespace nxzc

extern "C" int nxz_launch_copy_out(const nxz_batch_job_t *jobs, const nxz_batch_result_t *results, uint8_t *const *targets, size_t n, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxzc::copy_out_kernel, dim3((unsigned)n), dim3(256), 0, stream, jobs, results, targets);
	return (int)hipGetLastError();
}

static unsigned cut_rounds(size_t n)
{
	const char *e = getenv("NXZ_INFLATE_CUT_ROUNDS");                   // (read at every call: the tests switch it)
	const int v = e ? atoi(e) : n <= 64 ? 3 : 4;                        // (a round that has nothing left to do still costs its five launches)
	return v < 1 ? 1u : v > (int)nxzc::RMAX ? nxzc::RMAX : (unsigned)v;
}

// Device memory a batch of n streams cut into P pieces a round needs (control arrays + the arena of the pieces' elements)
extern "C" size_t nxz_inflate_cut_workspace(size_t n, unsigned P, size_t arena)
{
	auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
	const size_t PT = (size_t)P * nxzc::RMAX + 1, slots = n * PT;
	return up(n * sizeof(nxz_sync_req_t)) + up(n * (P - 1) * sizeof(nxz_sync_req_t)) + up(n * (P - 1) * sizeof(nxz_sync_res_t)) + up(n * sizeof(nxz_batch_dht_t)) +
	       up(n * sizeof(nxzi::Built)) + up(n * sizeof(nxzc::Ctl)) + up(slots * sizeof(nxz_batch_job_t)) + up(slots * sizeof(nxz_batch_result_t)) +
	       up(slots * sizeof(nxz_batch_dht_t)) + up(n * P * 4) + up(n * 4) * 2 + 256 + up(arena);
}

extern "C" unsigned nxz_inflate_cut_pieces(size_t n)
{
	const char *e = getenv("NXZ_INFLATE_CUT_PIECES");                  // (read at every call: the tests switch it)
	const int env = e ? atoi(e) : 0;
	if (env >= 2 && env <= (int)nxzc::PMAX) return (unsigned)env;
	// about 49 152 pieces a launch (the device holds 5120 wavefronts of the piece kernel at a time), 32 a stream at most
	size_t p = 49152 / (n ? n : 1);
	return (unsigned)(p > nxzc::PMAX ? nxzc::PMAX : p);
}

// The whole batch: see the head of this file.  ws: nxz_inflate_cut_workspace(n, P, arena) bytes of device memory.
extern "C" int nxz_launch_inflate_cut(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io,
				      unsigned P, uint8_t *ws, size_t arena, hipStream_t stream)
{
	using namespace nxzc;
	if (!n) return 0;
	if (P < 2 || P > PMAX) return -1;
	auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
	const unsigned rounds = cut_rounds(n);
	const uint32_t PT = P * RMAX + 1;
	const size_t slots = n * PT;
	uint8_t *p = ws;
	auto take = [&](size_t bytes) { uint8_t *r = p; p += up(bytes); return r; };
	nxz_sync_req_t *bq = (nxz_sync_req_t *)take(n * sizeof(nxz_sync_req_t));
	nxz_sync_req_t *rq = (nxz_sync_req_t *)take(n * (P - 1) * sizeof(nxz_sync_req_t));
	nxz_sync_res_t *rs = (nxz_sync_res_t *)take(n * (P - 1) * sizeof(nxz_sync_res_t));
	nxz_batch_dht_t *tb = (nxz_batch_dht_t *)take(n * sizeof(nxz_batch_dht_t));
	nxzi::Built *bt = (nxzi::Built *)take(n * sizeof(nxzi::Built));
	Ctl *ctl = (Ctl *)take(n * sizeof(Ctl));
	nxz_batch_job_t *pj = (nxz_batch_job_t *)take(slots * sizeof(nxz_batch_job_t));
	nxz_batch_result_t *pr = (nxz_batch_result_t *)take(slots * sizeof(nxz_batch_result_t));
	nxz_batch_dht_t *pd = (nxz_batch_dht_t *)take(slots * sizeof(nxz_batch_dht_t));
	uint32_t *order = (uint32_t *)take(n * P * 4);
	uint32_t *order2 = (uint32_t *)take(n * 4);
	uint32_t *plain = (uint32_t *)take(n * 4);
	Arena *ar = (Arena *)take(256);
	uint8_t *elems = p;
	const unsigned nb = (unsigned)((n + 255) / 256);
	int rc;
	for (unsigned r = 0; r < rounds; r++) {
		hipLaunchKernelGGL(plan_kernel, dim3(nb), dim3(256), 0, stream, jobs, (uint32_t)n, P, PT, r, bq, rq, ctl, ar, (unsigned long long)arena, order, order2, dht_io, tb);
		rc = nxz_launch_token_sync(bq, (uint32_t)n, tb, bt, rq, (uint32_t)(n * (P - 1)), rs, stream);
		if (rc) return rc;
		hipLaunchKernelGGL(jobs_kernel, dim3(nb), dim3(256), 0, stream, jobs, (uint32_t)n, P, PT, r, rs, tb, bt, ctl, pj, pd, elems, ar, order);
		rc = nxz_launch_inflate_w16_order(pj, n * P, pr, pd, bt, order, stream);
		if (rc) return rc;
		hipLaunchKernelGGL(check_kernel, dim3(nb), dim3(256), 0, stream, jobs, (uint32_t)n, PT, ctl, tb, pj, pr, pd);
	}
	hipLaunchKernelGGL(rest_kernel, dim3(nb), dim3(256), 0, stream, jobs, (uint32_t)n, PT, ctl, tb, pj, pd, elems, ar, order2);
	rc = nxz_launch_inflate_w16_order(pj, n, pr, pd, bt, order2, stream);
	if (rc) return rc;
	hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)n), dim3(256), 0, stream, jobs, PT, ctl, pj, pr, pd, results, dht_io, plain);
	rc = (int)hipGetLastError();
	if (rc) return rc;
	// whatever was not cut, or did not work out: a wavefront per stream; then the checksums of all outputs
	rc = nxz_launch_inflate(jobs, n, results, dht_io, 0, plain, stream);
	if (getenv("NXZ_INFLATE_CUT_TRACE")) {
		// diagnostic: what became of the streams
		(void)hipStreamSynchronize(stream);
		Ctl *h = (Ctl *)malloc(n * sizeof(Ctl));
		nxz_batch_result_t *hr = (nxz_batch_result_t *)malloc(slots * sizeof(nxz_batch_result_t));
		Arena a;
		(void)hipMemcpy(h, ctl, n * sizeof(Ctl), hipMemcpyDeviceToHost);
		(void)hipMemcpy(hr, pr, slots * sizeof(nxz_batch_result_t), hipMemcpyDeviceToHost);
		(void)hipMemcpy(&a, ar, sizeof(a), hipMemcpyDeviceToHost);
		size_t st[5] = {0, 0, 0, 0, 0}, np = 0, why[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		double us = 0, usmax = 0, rest_us = 0, rest_max = 0;
		for (size_t i = 0; i < n; i++) {
			st[h[i].state < 5 ? h[i].state : 0]++; np += h[i].npieces;
			if (h[i].state == ST_FAILED) why[h[i].why & 7]++;
			for (uint32_t j = 0; j < h[i].npieces; j++) { const double t = hr[i * PT + j].crc * 0.01; us += t; if (t > usmax) usmax = t; }
			if (h[i].state == ST_REST) { const double t = hr[i * PT + h[i].npieces].crc * 0.01; rest_us += t; if (t > rest_max) rest_max = t; }
		}
		fprintf(stderr, "nxz_inflate_cut: %zu streams x %u pieces x %u rounds: plain %zu, done in pieces %zu, + a rest piece %zu, failed %zu (no room %zu, bad piece %zu, full target %zu, rest %zu/%zu); "
				"%zu pieces stand; arena %llu of %llu bytes; piece times: mean %.0f us, max %.0f us; rest pieces: mean %.0f us, max %.0f us\n", n, P, rounds, st[0], st[2], st[4], st[3],
			why[1], why[2], why[3], why[4], why[5], np, a.used, a.size, np ? us / np : 0.0, usmax, st[4] ? rest_us / st[4] : 0.0, rest_max);
		if (getenv("NXZ_INFLATE_CUT_TRACE")[0] == '2')
			for (size_t i = getenv("NXZ_INFLATE_CUT_TRACE_FROM") ? (size_t)atoi(getenv("NXZ_INFLATE_CUT_TRACE_FROM")) : 0, i0 = i; i < n && i < i0 + 8; i++) {
				fprintf(stderr, "  stream %zu: state %u npieces %u made %u cur_bit %u ext %u boost %u why %#x |", i, h[i].state, h[i].npieces, h[i].made, h[i].cur_bit, h[i].ext, h[i].boost, h[i].why);
				for (uint32_t j = 0; j < h[i].npieces + (h[i].state == ST_REST) && j < 40; j++) { const nxz_batch_result_t &r = hr[i * PT + j]; fprintf(stderr, " [cc %u out %u sfbt %#x subc %u hdr %u %.0fus]", r.cc, r.tpbc, r.sfbt & 0xfff, r.subc, r.adler, r.crc * 0.01); }
				fprintf(stderr, "\n");
			}
		free(h); free(hr);
	}
	return rc;
}
