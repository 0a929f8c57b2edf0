// nxz_misc.hip -- small kernels of the engine: DHT preparation and WRAP (copy + checksums).
//
// dht_prepare: parses the caller's dynamic-Huffman header (the bit string the
//   reference keeps in cpb.in_dht, inc_nx/nxu.h:390-393; format RFC1951 3.2.7)
//   into canonical codes.  One lane per table: a table is <= 2283 bits and is
//   shared by many jobs, so this is off the critical path.  Same decisions as
//   oracle/nxz_huff.c nxo_dht_parse (what lib/nx_dht_decomp.c:255-653 models).
// wrap: GZIP_FC_WRAP (lib/nx_deflate.c:1774, lib/nx_zlib.c:1398-1443): copy
//   source to target and return CRC-32/Adler-32 from the initial values.
// member_offsets / pack_members: turn the outputs of a compress batch into a contiguous run of
//   gzip members, one per job (RFC 1952; each carries the 6-byte "BC" extra subfield with its own
//   size, the BGZF layout, so a reader can hop from member to member and inflate them in
//   parallel).  A job that did not shrink is written as a stored block, the fallback the reference
//   applies per job (lib/nx_deflate.c:1292-1400, append_btype00_header :175-201).
#include <hip/hip_runtime.h>
#include "nxz_device.h"

namespace nxz {

// canonical codes of one alphabet by ranks: symbol i = row * 64 + lane; code = first code of its
// length + the number of lower symbols of that length (RFC 1951 3.2.2).  out[i] = bit-reversed
// code | length << 16, 0 for an absent symbol.  Returns false when the lengths oversubscribe.
template <int ROWS>
__device__ bool canon_wave(const uint8_t *len, uint32_t n, uint32_t *__restrict__ out, uint32_t n_out)
{
	const uint32_t lane = threadIdx.x;
	const uint64_t below = (1ull << lane) - 1;
	uint32_t l[ROWS], code[ROWS];
#pragma unroll
	for (int r = 0; r < ROWS; r++) { const uint32_t i = r * 64 + lane; l[r] = i < n ? len[i] : 0; code[r] = 0; }
	uint32_t c = 0, prevcnt = 0, kraft = 0;
	for (uint32_t b = 1; b <= 15; b++) {
		c = (c + prevcnt) << 1;
		uint32_t run = 0;
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			const uint64_t m = __ballot(l[r] == b);
			if (l[r] == b) code[r] = c + run + (uint32_t)__popcll(m & below);
			run += (uint32_t)__popcll(m);
		}
		prevcnt = run;
		kraft += run << (15 - b);
	}
#pragma unroll
	for (int r = 0; r < ROWS; r++) {
		const uint32_t i = r * 64 + lane;
		if (i < n_out) out[i] = l[r] ? ((__builtin_bitreverse32(code[r]) >> (32 - l[r])) | (l[r] << 16)) : 0;
	}
	return kraft <= (1u << 15);
}

// One wavefront per table.  The bit string is taken into LDS in one go (it may sit in pinned host
// memory: nxu_run_job's rounds hand it over in place), the code-length code becomes a 128-entry
// look-up held in two registers per lane, and the run-length coded lengths are decoded by all lanes
// in step (everything in that loop is wave-uniform: the look-up is a v_readlane).
__global__ __launch_bounds__(64) void dht_prepare_kernel(const nxz_batch_dht_t *__restrict__ in, size_t n, nxz_dht_prepared_t *__restrict__ out)
{
	__shared__ uint32_t bits[80];
	__shared__ uint8_t lens[320 + 8];
	__shared__ uint8_t cl_len[32];
	const uint32_t lane = threadIdx.x;
	const size_t idx = blockIdx.x;
	if (idx >= n) return;
	const nxz_batch_dht_t *t = &in[idx];
	nxz_dht_prepared_t *o = &out[idx];
	const uint32_t dhtlen = t->dhtlen;
	bool ok = dhtlen <= NXZ_DHT_MAXSZ * 8;
	if (lane == 0) o->dhtlen = dhtlen;
	for (uint32_t i = lane; i < 80; i += 64) {
		uint32_t w = 0;
		const uint32_t bit0 = i * 32;
		if (ok && i < 73 && bit0 < dhtlen) {
			w = ((const uint32_t *)t->dht)[i];
			if (bit0 + 32 > dhtlen) w &= (1u << (dhtlen - bit0)) - 1;
		}
		bits[i] = w;
		if (i < 74) o->dhtw[i] = w;
	}
	for (uint32_t i = lane; i < 320; i += 64) lens[i] = 0;
	if (lane < 32) cl_len[lane] = 0;
	__syncthreads();
	auto peek = [&](uint32_t pos, uint32_t nb) -> uint32_t {                       // nb <= 25, pos + nb within the padded string
		const uint32_t w = pos >> 5, sh = pos & 31;
		const uint64_t v = (uint64_t)bits[w] | ((uint64_t)bits[w + 1] << 32);
		return (uint32_t)(v >> sh) & ((1u << nb) - 1);
	};
	uint32_t hlit = 0, hdist = 0, hclen = 0;
	if (ok) {
		ok = dhtlen >= 14;
		const uint32_t v = peek(0, 14);
		hlit = (v & 31) + 257; hdist = ((v >> 5) & 31) + 1; hclen = (v >> 10) + 4;
		ok = ok && hlit <= 286 && hdist <= 30 && 14 + 3 * hclen <= dhtlen;
	}
	if (ok && lane < hclen) {
		const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
		cl_len[order[lane]] = (uint8_t)peek(14 + 3 * lane, 3);
	}
	__syncthreads();
	// code-length code: canonical codes (19 symbols, every lane does all of it), then the look-up
	// entry of peek values `lane` and `lane + 64`: symbol | length << 5, 0xff = no code
	uint32_t tlo = 0xff, thi = 0xff;
	{
		uint32_t cnt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, next[8], c = 0, kraft = 0;
		for (int i = 0; i < 19; i++) cnt[cl_len[i] & 7]++;
		cnt[0] = 0;
		for (int b = 1; b <= 7; b++) { c = (c + cnt[b - 1]) << 1; next[b] = c; kraft += cnt[b] << (7 - b); }
		ok = ok && kraft <= 128;
		for (int i = 0; i < 19; i++) {
			const uint32_t l = cl_len[i] & 7;
			if (!l) continue;
			const uint32_t code = __builtin_bitreverse32(next[l]++) >> (32 - l), mask = (1u << l) - 1;
			if ((lane & mask) == code) tlo = (uint32_t)i | (l << 5);
			if (((lane + 64) & mask) == code) thi = (uint32_t)i | (l << 5);
		}
	}
	const uint32_t total = hlit + hdist;
	uint32_t pos = 14 + 3 * hclen, cnt = 0, prev = 0;
	while (ok && cnt < total) {
		const uint32_t v = __builtin_amdgcn_readfirstlane(peek(pos, 14));
		const uint32_t e = (v & 64) ? __builtin_amdgcn_readlane(thi, v & 63) : __builtin_amdgcn_readlane(tlo, v & 63);
		if (e == 0xff) { ok = false; break; }
		const uint32_t sym = e & 31, l = e >> 5;
		pos += l;
		if (pos > dhtlen) { ok = false; break; }
		const uint32_t x = v >> l;                                                 // the extra bits, if any
		if (sym < 16) {
			if (lane == 0) lens[cnt] = (uint8_t)sym;
			cnt++; prev = sym;
			continue;
		}
		uint32_t rep, val = 0, nb;
		if (sym == 16) { if (cnt == 0) { ok = false; break; } nb = 2; rep = (x & 3) + 3; val = prev; }
		else if (sym == 17) { nb = 3; rep = (x & 7) + 3; }
		else { nb = 7; rep = (x & 127) + 11; }
		pos += nb;
		if (pos > dhtlen || cnt + rep > total) { ok = false; break; }
		for (uint32_t k = lane; k < rep; k += 64) lens[cnt + k] = (uint8_t)val;
		cnt += rep;
		if (sym != 16) prev = 0;
	}
	ok = ok && pos == dhtlen;
	__syncthreads();
	// literal/length and distance alphabets: lengths beyond hlit / hdist (and of a bad table) are zero
	if (!ok) for (uint32_t i = lane; i < 320; i += 64) lens[i] = 0;
	__syncthreads();
	const bool okl = canon_wave<5>(lens, ok ? hlit : 0, o->ll, 288);
	const bool okd = canon_wave<1>(lens + hlit, ok ? hdist : 0, o->d, 32);
	if (lane == 0) o->status = ok && okl && okd ? 0 : NXZ_CC_INVALID_DHT;
}

// ---- wrap: one 256-thread workgroup per job ----
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

__global__ __launch_bounds__(256) void wrap_kernel(const nxz_batch_job_t *__restrict__ jobs, nxz_batch_result_t *__restrict__ results)
{
	__shared__ uint32_t tab[256];
	__shared__ uint32_t red[3][256];
	const int t = threadIdx.x;
	const nxz_batch_job_t job = jobs[blockIdx.x];
	const uint32_t n = job.src_len;
	{
		uint32_t c = t;
		for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1) ? 0xedb88320u : 0);
		tab[t] = c;
	}
	__syncthreads();
	if (n > job.dst_cap) {
		if (t == 0) { nxz_batch_result_t r = {NXZ_CC_TARGET_SPACE, 0, 0, 0, 0, 0, 0, 0}; results[blockIdx.x] = r; }
		return;
	}
	// contiguous slice per thread, slice length a multiple of 16 so that copies stay vectorised
	uint32_t per = ((n + 255) / 256 + 15) & ~15u;
	uint32_t lo = (uint32_t)t * per, hi = lo + per < n ? lo + per : n;
	uint32_t crc = 0, s1 = 0, s2 = 0;
	if (lo < n) {
		const uint8_t *s = job.src; uint8_t *d = job.dst;
		uint32_t i = lo;
		for (; i + 16 <= hi; i += 16) {
			uint4 v = *(const uint4 *)(s + i);
			*(uint4 *)(d + i) = v;
			uint32_t w[4] = {v.x, v.y, v.z, v.w};
			for (int k = 0; k < 16; k++) {
				uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
				uint32_t bx = b ^ ((i + k) < 4 ? 0xffu : 0);            // initial value 0 -> ~0 folded into bytes 0..3
				crc = tab[(crc ^ bx) & 0xff] ^ (crc >> 8);
				s1 += b; s2 += s1;
			}
			if ((i & 0xfff) == 0xff0) { s1 %= 65521u; s2 %= 65521u; }
		}
		for (; i < hi; i++) {
			uint32_t b = s[i]; d[i] = (uint8_t)b;
			uint32_t bx = b ^ (i < 4 ? 0xffu : 0);
			crc = tab[(crc ^ bx) & 0xff] ^ (crc >> 8);
			s1 += b; s2 += s1;
		}
		s1 %= 65521u; s2 %= 65521u;
	}
	red[0][t] = crc; red[1][t] = s1; red[2][t] = s2;
	__syncthreads();
	if (t == 0) {
		// sequential combine of 256 slices (tiny): crc = crc*x^(8*len) ^ next; adler combine
		uint32_t c = 0, a1 = 1, a2 = 0;
		uint32_t xp = 0x80000000u;   // x^(8*per), by square-and-multiply on x^8
		{ uint32_t sq = 0x00800000u, e = per; while (e) { if (e & 1) xp = gf_mul(xp, sq); sq = gf_mul(sq, sq); e >>= 1; } }
		for (uint32_t k = 0; k < 256; k++) {
			uint32_t klo = k * per; if (klo >= n) break;
			uint32_t len = klo + per < n ? per : n - klo;
			uint32_t m = xp;
			if (len != per) { m = 0x80000000u; uint32_t sq = 0x00800000u, e = len; while (e) { if (e & 1) m = gf_mul(m, sq); sq = gf_mul(sq, sq); e >>= 1; } }
			c = gf_mul(c, m) ^ red[0][k];
			// adler: s1' = s1 + S1k ; s2' = s2 + len*s1 + S2k   (S* computed from a zero start)
			a2 = (uint32_t)((a2 + (uint64_t)len * a1 + red[2][k]) % 65521u);
			a1 = (a1 + red[1][k]) % 65521u;
		}
		if (n < 4) { c = 0xffffffffu; for (uint32_t i = 0; i < n; i++) c = tab[(c ^ job.src[i]) & 0xff] ^ (c >> 8); }
		nxz_batch_result_t r;
		r.cc = 0; r.tpbc = n; r.tebc = 0; r.spbc = n; r.crc = c ^ 0xffffffffu; r.adler = (a2 << 16) | a1; r.subc = 0; r.sfbt = 0;
		results[blockIdx.x] = r;
	}
}


// ---- gzip members from a compress batch -------------------------------------------------------
#define NXZ_MEMBER_OVERHEAD 26u      /* 18-byte header with the BC subfield + CRC32 + ISIZE */

__device__ inline bool member_stored(const nxz_batch_job_t &job, const nxz_batch_result_t &r)
{
	const uint32_t len = job.src_len - job.hist_len;
	return r.cc != 0 || r.tpbc >= len + 5;
}

// offsets[0..n]: exclusive prefix sum of the member sizes; one workgroup
__global__ __launch_bounds__(1024) void member_offsets_kernel(const nxz_batch_job_t *__restrict__ jobs, const nxz_batch_result_t *__restrict__ results,
							      uint32_t n, uint64_t *__restrict__ offsets)
{
	__shared__ uint64_t part[1024];
	const uint32_t t = threadIdx.x;
	const uint32_t per = (n + 1023) / 1024;
	const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
	uint64_t sum = 0;
	for (uint32_t i = lo; i < hi; i++) {
		const nxz_batch_job_t job = jobs[i];
		const nxz_batch_result_t r = results[i];
		sum += NXZ_MEMBER_OVERHEAD + (member_stored(job, r) ? 5 + (job.src_len - job.hist_len) : r.tpbc);
	}
	part[t] = sum;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) {
		uint64_t v = t >= d ? part[t - d] : 0;
		__syncthreads();
		part[t] += v;
		__syncthreads();
	}
	uint64_t off = part[t] - sum;
	for (uint32_t i = lo; i < hi; i++) {
		const nxz_batch_job_t job = jobs[i];
		const nxz_batch_result_t r = results[i];
		offsets[i] = off;
		off += NXZ_MEMBER_OVERHEAD + (member_stored(job, r) ? 5 + (job.src_len - job.hist_len) : r.tpbc);
	}
	if (t == 1023) offsets[n] = part[1023];
}

// one workgroup per member: header, payload (the job's output, or 5 bytes of stored-block header and
// the source), trailer.  Whole dwords of the packed buffer are stored at once where the member
// covers them; the payload bytes of such a dword come from two aligned dword loads.
__global__ __launch_bounds__(256) void pack_members_kernel(const nxz_batch_job_t *__restrict__ jobs, const nxz_batch_result_t *__restrict__ results,
							   const uint64_t *__restrict__ offsets, uint8_t *__restrict__ packed)
{
	const uint32_t t = threadIdx.x;
	const nxz_batch_job_t job = jobs[blockIdx.x];
	const nxz_batch_result_t r = results[blockIdx.x];
	const uint64_t off = offsets[blockIdx.x];
	const uint32_t len = job.src_len - job.hist_len;
	const bool stored = member_stored(job, r);
	const uint32_t paylen = stored ? 5 + len : r.tpbc;
	const uint32_t size = NXZ_MEMBER_OVERHEAD + paylen;
	const uint32_t d0 = stored ? 23 : 18;                         // first member byte that comes from `data`
	const uint32_t dn = stored ? len : r.tpbc;
	const uint8_t *data = stored ? job.src + job.hist_len : job.dst;
	const uint32_t bsize = size - 1;
	auto byte_at = [&](uint32_t j) -> uint32_t {
		if (j < 18) {
			// ID1 ID2 CM FLG=FEXTRA MTIME(4) XFL OS=unknown XLEN=6 'B' 'C' SLEN=2 BSIZE
			const uint32_t h0 = 0x04088b1fu, h2 = 0x0006ff00u, h3 = 0x00024342u;
			return j < 4 ? (h0 >> (8 * j)) & 0xff : j < 8 ? 0 : j < 12 ? (h2 >> (8 * (j - 8))) & 0xff :
			       j < 16 ? (h3 >> (8 * (j - 12))) & 0xff : (bsize >> (8 * (j - 16))) & 0xff;
		}
		if (j < d0) {                                           // stored block: BFINAL=1 BTYPE=00, LEN, NLEN
			const uint32_t k = j - 18;
			return k == 0 ? 1 : k < 3 ? (len >> (8 * (k - 1))) & 0xff : (~len >> (8 * (k - 3))) & 0xff;
		}
		if (j < d0 + dn) return data[j - d0];
		const uint32_t k = j - d0 - dn;                         // CRC32, ISIZE
		return k < 4 ? (r.crc >> (8 * k)) & 0xff : (len >> (8 * (k - 4))) & 0xff;
	};
	uint8_t *o = packed + off;
	const uint32_t head = (uint32_t)((4 - ((uintptr_t)o & 3)) & 3);      // bytes before the first aligned dword
	const uint32_t nd = size > head ? (size - head) >> 2 : 0;            // whole dwords
	if (t < head && t < size) o[t] = (uint8_t)byte_at(t);
	for (uint32_t k = head + nd * 4 + t; k < size; k += 256) o[k] = (uint8_t)byte_at(k);
	uint32_t *od = (uint32_t *)(o + head);
	for (uint32_t w = t; w < nd; w += 256) {
		const uint32_t j = head + w * 4;
		uint32_t v;
		if (j >= d0 && j + 4 <= d0 + dn) {
			const uintptr_t a = (uintptr_t)data + (j - d0);
			const uint32_t *q = (const uint32_t *)(a & ~(uintptr_t)3);
			const uint32_t bo = (uint32_t)a & 3;
			const uint32_t lo = q[0], hi = bo ? q[1] : 0;
			v = __builtin_amdgcn_alignbyte(hi, lo, bo);
		} else {
			v = byte_at(j) | byte_at(j + 1) << 8 | byte_at(j + 2) << 16 | byte_at(j + 3) << 24;
		}
		od[w] = v;
	}
}

// ---- one deflate stream from a compress batch ---------------------------------------------------
// The blocks of a batch, made independently, laid back to back as ONE raw deflate stream the way
// the reference strings its jobs together: a block that ends inside a byte is followed by an
// empty stored block (append_sync_flush, lib/nx_deflate.c:220-243), so every block starts on a
// byte boundary; a block that failed or did not shrink goes out as stored blocks of its source
// (lib/nx_deflate.c:1274-1282, append_btype00_header :175-201); block `final_index` carries BFINAL.
struct StreamPiece {
	bool stored;
	uint32_t len;       // source bytes
	uint32_t keep;      // compressed: bytes taken from the job's output
	uint32_t size;      // bytes this block occupies in the stream
	uint32_t pad;       // compressed: zero bytes between the output and 00 00 FF FF (the 3 header bits of the empty block may spill)
	bool marker;
};

__device__ inline StreamPiece stream_piece(const nxz_batch_job_t &job, const nxz_batch_result_t &r, bool final)
{
	StreamPiece p;
	p.len = job.src_len - job.hist_len;
	p.stored = r.cc != 0 || r.tpbc > p.len;
	if (p.stored) {
		p.keep = 0; p.pad = 0; p.marker = false;
		p.size = p.len > 65535 ? p.len + 10 : p.len + 5;
	} else {
		p.keep = r.tpbc;
		p.marker = !final && r.tebc != 0;
		p.pad = p.marker && r.tebc + 3 > 8 ? 1 : 0;
		p.size = r.tpbc + (p.marker ? p.pad + 4 : 0);
	}
	return p;
}

__global__ __launch_bounds__(1024) void stream_offsets_kernel(const nxz_batch_job_t *__restrict__ jobs, const nxz_batch_result_t *__restrict__ results,
							      uint32_t n, uint32_t final_index, uint64_t *__restrict__ offsets)
{
	__shared__ uint64_t part[1024];
	const uint32_t t = threadIdx.x;
	const uint32_t per = (n + 1023) / 1024;
	const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
	uint64_t sum = 0;
	for (uint32_t i = lo; i < hi; i++) sum += stream_piece(jobs[i], results[i], i == final_index).size;
	part[t] = sum;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) {
		uint64_t v = t >= d ? part[t - d] : 0;
		__syncthreads();
		part[t] += v;
		__syncthreads();
	}
	uint64_t off = part[t] - sum;
	for (uint32_t i = lo; i < hi; i++) {
		offsets[i] = off;
		off += stream_piece(jobs[i], results[i], i == final_index).size;
	}
	if (t == 1023) offsets[n] = part[1023];
}

// one block of the stream, by one workgroup of 256: `o` is where it goes
__device__ inline void pack_block(const nxz_batch_job_t &job, const nxz_batch_result_t &r, const bool final, uint8_t *__restrict__ o)
{
	const uint32_t t = threadIdx.x;
	const StreamPiece p = stream_piece(job, r, final);
	const uint8_t *data = p.stored ? job.src + job.hist_len : job.dst;
	// stored: [hdr 5][first][hdr 5][rest]; compressed: [keep][pad][00 00 FF FF]
	const uint32_t first = p.len > 65535 ? 65535 : p.len, rest = p.len - first;
	const uint32_t d0 = p.stored ? 5 : 0;                           // first stream byte that is a plain copy of `data`
	const uint32_t dn = p.stored ? first : p.keep;
	const uint32_t lastmask = r.tebc ? (1u << r.tebc) - 1 : 0xff;
	auto byte_at = [&](uint32_t j) -> uint32_t {
		if (p.stored) {
			auto hdr = [&](uint32_t k, uint32_t n, bool last) -> uint32_t {
				return k == 0 ? (last ? 1u : 0u) : k < 3 ? (n >> (8 * (k - 1))) & 0xff : (~n >> (8 * (k - 3))) & 0xff;
			};
			if (j < 5) return hdr(j, first, final && rest == 0);
			if (j < 5 + first) return data[j - 5];
			if (j < 10 + first) return hdr(j - 5 - first, rest, final);
			return data[j - 10];
		}
		if (j < p.keep) {
			uint32_t v = data[j];
			if (j == 0) v = (v & ~1u) | (final ? 1u : 0u);          // set_bfinal (lib/nx_deflate.c:1404-1413)
			if (j == p.keep - 1) v &= lastmask;
			return v;
		}
		const uint32_t k = j - p.keep - p.pad;                      // after the pad byte: LEN = 0, NLEN = ffff
		return j < p.keep + p.pad ? 0 : k < 2 ? 0 : 0xff;
	};
	const uint32_t size = p.size;
	const uint32_t head = (uint32_t)((4 - ((uintptr_t)o & 3)) & 3);
	const uint32_t nd = size > head ? (size - head) >> 2 : 0;
	if (t < head && t < size) o[t] = (uint8_t)byte_at(t);
	for (uint32_t k = head + nd * 4 + t; k < size; k += 256) o[k] = (uint8_t)byte_at(k);
	uint32_t *od = (uint32_t *)(o + head);
	for (uint32_t w = t; w < nd; w += 256) {
		const uint32_t j = head + w * 4;
		uint32_t v;
		// whole dwords strictly inside the copied region (not its first or last byte, which are patched)
		if (j > d0 && j + 4 < d0 + dn) {
			const uintptr_t a = (uintptr_t)data + (j - d0);
			const uint32_t *q = (const uint32_t *)(a & ~(uintptr_t)3);
			const uint32_t bo = (uint32_t)a & 3;
			const uint32_t lo = q[0], hi = bo ? q[1] : 0;
			v = __builtin_amdgcn_alignbyte(hi, lo, bo);
		} else {
			v = byte_at(j) | byte_at(j + 1) << 8 | byte_at(j + 2) << 16 | byte_at(j + 3) << 24;
		}
		od[w] = v;
	}
}


__global__ __launch_bounds__(256) void pack_stream_kernel(const nxz_batch_job_t *__restrict__ jobs, const nxz_batch_result_t *__restrict__ results,
							  uint32_t final_index, const uint64_t *__restrict__ offsets, uint8_t *__restrict__ packed)
{
	pack_block(jobs[blockIdx.x], results[blockIdx.x], blockIdx.x == final_index, packed + offsets[blockIdx.x]);
}

// ---- the same for several callers' streams at once ------------------------------------------------
// nxz_deflate_host calls of a few MiB from many threads go out together (nxz_engine.cpp, merged calls): the batch holds
// the blocks of all of them, member after member; each member's blocks become that member's stream.  A wavefront per
// member lays out its offsets (a member has a few dozen blocks), a workgroup per block packs.
__global__ __launch_bounds__(64) void member_stream_offsets_kernel(const nxz_batch_job_t *__restrict__ jobs, const nxz_batch_result_t *__restrict__ results,
									const nxz_pack_member_t *__restrict__ members, uint64_t *__restrict__ offsets)
{
	const nxz_pack_member_t m = members[blockIdx.x];
	const uint32_t lane = threadIdx.x;
	uint64_t base = 0;
	for (uint32_t i0 = 0; i0 < m.n; i0 += 64) {
		const uint32_t i = i0 + lane;
		const uint32_t size = i < m.n ? stream_piece(jobs[m.b0 + i], results[m.b0 + i], m.b0 + i == m.fin).size : 0;
		uint64_t incl = size;
		for (uint32_t d = 1; d < 64; d <<= 1) {
			const uint64_t v = __shfl_up(incl, d);
			if (lane >= d) incl += v;
		}
		if (i < m.n) offsets[m.off0 + i] = base + incl - size;
		base += __shfl(incl, 63);
	}
	if (lane == 0) offsets[m.off0 + m.n] = base;
}

__global__ __launch_bounds__(256) void member_pack_stream_kernel(const nxz_batch_job_t *__restrict__ jobs, const nxz_batch_result_t *__restrict__ results,
								      const nxz_pack_member_t *__restrict__ members, const uint16_t *__restrict__ member_of,
								      const uint64_t *__restrict__ offsets)
{
	const nxz_pack_member_t m = members[member_of[blockIdx.x]];
	pack_block(jobs[blockIdx.x], results[blockIdx.x], blockIdx.x == m.fin, m.packed + offsets[m.off0 + (blockIdx.x - m.b0)]);
}

} // namespace nxz

extern "C" int nxz_launch_pack_stream(const nxz_batch_job_t *jobs, const nxz_batch_result_t *results, size_t n, uint32_t final_index,
				      uint64_t *offsets, uint8_t *packed, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxz::stream_offsets_kernel, dim3(1), dim3(1024), 0, stream, jobs, results, (uint32_t)n, final_index, offsets);
	hipLaunchKernelGGL(nxz::pack_stream_kernel, dim3((unsigned)n), dim3(256), 0, stream, jobs, results, final_index, offsets, packed);
	return (int)hipGetLastError();
}

extern "C" int nxz_launch_pack_member_streams(const nxz_batch_job_t *jobs, const nxz_batch_result_t *results, size_t nblocks, const nxz_pack_member_t *members,
					       size_t nmembers, const uint16_t *member_of, uint64_t *offsets, hipStream_t stream)
{
	if (!nblocks || !nmembers) return 0;
	hipLaunchKernelGGL(nxz::member_stream_offsets_kernel, dim3((unsigned)nmembers), dim3(64), 0, stream, jobs, results, members, offsets);
	hipLaunchKernelGGL(nxz::member_pack_stream_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream, jobs, results, members, member_of, offsets);
	return (int)hipGetLastError();
}

extern "C" int nxz_launch_dht_prepare(const nxz_batch_dht_t *dht, size_t n, nxz_dht_prepared_t *out, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxz::dht_prepare_kernel, dim3((unsigned)n), dim3(64), 0, stream, dht, n, out);
	return (int)hipGetLastError();
}

namespace nxz {
// How many of 256 streams spread over a batch begin with a block that brings its own table (BTYPE 10): the batched
// inflate picks its kernel by that (nxz_engine.cpp nxz_batch_decompress).  out[0]: the count; out[1], out[2]: the shortest and the longest source among them.
__global__ __launch_bounds__(256) void sample_btype_kernel(const nxz_batch_job_t *__restrict__ jobs, uint32_t n, uint32_t *out)
{
	__shared__ uint32_t cnt, lo, hi;
	if (threadIdx.x == 0) { cnt = 0; lo = 0xffffffffu; hi = 0; }
	__syncthreads();
	const uint32_t i = (uint32_t)(((uint64_t)threadIdx.x * n) >> 8);
	const nxz_batch_job_t j = jobs[i];
	atomicMin(&lo, j.src_len - (j.hist_len < j.src_len ? j.hist_len : j.src_len));
	atomicMax(&hi, j.src_len - (j.hist_len < j.src_len ? j.hist_len : j.src_len));
	// (a job that resumes inside a stream names its block type in the resume word; a fresh stream in its first byte)
	const uint32_t first = j.src_len > j.hist_len ? j.src[j.hist_len] : 0;
	const bool dyn = (j.resume >> 16) & 15 ? ((j.resume >> 17) & 7) == 6 : ((first >> 1) & 3) == 2;
	if (dyn) atomicAdd(&cnt, 1u);
	__syncthreads();
	if (threadIdx.x == 0) { out[0] = cnt; out[1] = lo; out[2] = hi; }      // (and the shortest and the longest of the sampled streams)
}
}
extern "C" int nxz_launch_sample_btype(const nxz_batch_job_t *jobs, size_t n, uint32_t *out, hipStream_t stream)
{
	hipLaunchKernelGGL(nxz::sample_btype_kernel, dim3(1), dim3(256), 0, stream, jobs, (uint32_t)n, out);
	return (int)hipGetLastError();
}

extern "C" int nxz_launch_wrap(const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxz::wrap_kernel, dim3((unsigned)n), dim3(256), 0, stream, jobs, results);
	return (int)hipGetLastError();
}

extern "C" int nxz_launch_pack_members(const nxz_batch_job_t *jobs, const nxz_batch_result_t *results, size_t n,
				       uint64_t *offsets, uint8_t *packed, hipStream_t stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(nxz::member_offsets_kernel, dim3(1), dim3(1024), 0, stream, jobs, results, (uint32_t)n, offsets);
	hipLaunchKernelGGL(nxz::pack_members_kernel, dim3((unsigned)n), dim3(256), 0, stream, jobs, results, offsets, packed);
	return (int)hipGetLastError();
}

// ---- the achievable HBM figure (bench.py's roofline: peak_measured): a device-to-device copy, 16 bytes a lane, a grid of a few
// workgroups a CU that strides through the buffer (MI355X_MICROARCH.md: 6.29 TB/s of read + written bytes this way) ----
namespace nxz {
typedef uint32_t copy_v4 __attribute__((ext_vector_type(4)));
template <bool NT, int U>
__global__ __launch_bounds__(256) void copy16_kernel(const copy_v4 *__restrict__ src, copy_v4 *__restrict__ dst, size_t n16)
{
	const NXZ_GLOBAL_AS copy_v4 *s = (const NXZ_GLOBAL_AS copy_v4 *)src;
	NXZ_GLOBAL_AS copy_v4 *d = (NXZ_GLOBAL_AS copy_v4 *)dst;
	const size_t stride = (size_t)gridDim.x * 256;
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	for (; i + (U - 1) * stride < n16; i += U * stride) {
		copy_v4 v[U];
#pragma unroll
		for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(&s[i + k * stride]) : s[i + k * stride];
#pragma unroll
		for (int k = 0; k < U; k++) { if (NT) __builtin_nontemporal_store(v[k], &d[i + k * stride]); else d[i + k * stride] = v[k]; }
	}
	for (; i < n16; i += stride) d[i] = s[i];
}
}
extern "C" int nxz_launch_copy16(const void *src, void *dst, size_t bytes, hipStream_t stream)
{
	if (((uintptr_t)src | (uintptr_t)dst | bytes) & 15) return (int)hipErrorInvalidValue;
	static const unsigned cus = [] { int dev = 0, v = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return (unsigned)(v > 0 ? v : 256); }();
	// (NXZ_COPY_VARIANT, measurements: grid per CU * 100 + nontemporal * 10 + unroll)
	const char *e = getenv("NXZ_COPY_VARIANT");
	const int var = e ? atoi(e) : 611;      // (six workgroups a CU, nontemporal, one 16-byte load and store in flight a lane: 6.04 TB/s; eight a CU with four in flight 4.6, the runtime's memcpy 4.8-5.5: tools/lab/copy_peak.py)
	const unsigned per_cu = (unsigned)(var / 100 > 0 ? var / 100 : 8);
	const bool nt = (var / 10) % 10 != 0;
	const int u = var % 10;
	const dim3 g(cus * per_cu), b(256);
	const nxz::copy_v4 *sp = (const nxz::copy_v4 *)src; nxz::copy_v4 *dp = (nxz::copy_v4 *)dst; const size_t n16 = bytes / 16;
	if (nt && u >= 4) hipLaunchKernelGGL((nxz::copy16_kernel<true, 4>), g, b, 0, stream, sp, dp, n16);
	else if (nt && u == 2) hipLaunchKernelGGL((nxz::copy16_kernel<true, 2>), g, b, 0, stream, sp, dp, n16);
	else if (nt) hipLaunchKernelGGL((nxz::copy16_kernel<true, 1>), g, b, 0, stream, sp, dp, n16);
	else if (u >= 4) hipLaunchKernelGGL((nxz::copy16_kernel<false, 4>), g, b, 0, stream, sp, dp, n16);
	else if (u == 2) hipLaunchKernelGGL((nxz::copy16_kernel<false, 2>), g, b, 0, stream, sp, dp, n16);
	else hipLaunchKernelGGL((nxz::copy16_kernel<false, 1>), g, b, 0, stream, sp, dp, n16);
	return (int)hipGetLastError();
}
