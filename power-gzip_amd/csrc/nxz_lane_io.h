// nxz_lane_io.h -- the per-lane bit reader and output writer of the lane-per-stream inflate kernel
// (nxz_inflate_lanes.hip).  Plain C++ apart from the NXZ_LANE_* macros, so that the CPU test
// (tests/test_lane_io.py) can run exactly this code under AddressSanitizer.
#ifndef NXZ_LANE_IO_H
#define NXZ_LANE_IO_H
#include <stdint.h>
#ifdef __HIPCC__
#define NXZ_LANE_FN __device__ __forceinline__
#define NXZ_LANE_ALIGNBYTE(hi, lo, sh) __builtin_amdgcn_alignbyte(hi, lo, sh)
#define NXZ_LANE_GLOBAL __attribute__((address_space(1)))      // job buffers are device memory: global_load, not flat_load
#else
#define NXZ_LANE_FN inline
#define NXZ_LANE_GLOBAL
static inline uint32_t nxz_lane_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh)
{
	return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3)));
}
#define NXZ_LANE_ALIGNBYTE(hi, lo, sh) nxz_lane_alignbyte(hi, lo, sh)
#endif

#ifndef NXZ_LANE_REFILL_ABOVE
#define NXZ_LANE_REFILL_ABOVE 56     /* refill() loads unless more bits than this are at hand (56: whenever a whole byte fits) */
#endif

namespace nxzl {

// per-lane bit reader over global memory
struct BitRd {
	const uint8_t *src; uint32_t srclen;
	uint64_t bb; uint32_t bc;        // bb holds bits [pos, pos + bc)
	uint64_t pos;                    // next unread bit
	NXZ_LANE_FN uint64_t total() const { return (uint64_t)srclen * 8; }
	NXZ_LANE_FN bool have(uint32_t n) const { return pos + n <= total(); }
	NXZ_LANE_FN void sync() { bb = 0; bc = 0; }
	// A load by one lane is a load instruction of the whole wavefront, and those (64 lanes, 64 cache lines) are what the lane
	// kernels are bound by -- so a lane that loads takes all the bytes bb has room for, and the kernels call refill() where
	// every lane passes at the same time (the head of a trip round the token loop), whether a lane is short of bits or not.
	// Afterwards: 57..64 bits while 8 bytes of the source remain (one unaligned 8-byte load); the last bytes come 4 at a
	// time from the aligned dwords that hold them, and only when fewer than 32 bits are at hand (>= 32 afterwards, 25 right
	// after a sync() at a bit inside a byte; zero bits past the end) -- a byte outside the source is never read.
	NXZ_LANE_FN void refill()
	{
		if (bc > NXZ_LANE_REFILL_ABOVE) return;
		const uint64_t p2 = pos + bc;
		const uint32_t byte = (uint32_t)(p2 >> 3), sh = (uint32_t)p2 & 7;
		if (byte + 8 <= srclen) {
			uint64_t v;
			__builtin_memcpy(&v, (const NXZ_LANE_GLOBAL uint8_t *)src + byte, 8);
			const uint32_t room = 64 - bc + sh;                     // bits of v that fit, the sh already consumed ones included
			const uint32_t nbits = room >= 64 ? 64 : room & ~7u;    // whole bytes
			if (nbits < 64) v &= ((uint64_t)1 << nbits) - 1;
			bb |= (v >> sh) << bc;
			bc += nbits - sh;
			return;
		}
		if (bc >= 32) return;
		uint32_t v = 0;
		if (byte < srclen) {
			const uintptr_t a = (uintptr_t)src + byte;
			const uint32_t *w = (const uint32_t *)(a & ~(uintptr_t)3);
			const uint32_t bo = (uint32_t)a & 3, left = srclen - byte;         // bytes of the source from here on
			const uint32_t lo = w[0], hi = bo + left > 4 ? w[1] : 0;
			v = NXZ_LANE_ALIGNBYTE(hi, lo, bo);
			if (left < 4) v &= (1u << (8 * left)) - 1;
		}
		bb |= (uint64_t)(v >> sh) << bc;
		bc += 32 - sh;
	}
	NXZ_LANE_FN void fill() { if (bc < 32) refill(); }
	NXZ_LANE_FN void need(uint32_t n) { if (bc < n) refill(); }        // n <= 32 (25 right after a sync())
	NXZ_LANE_FN void drop(uint32_t n) { bb >>= n; bc -= n; pos += n; }
	NXZ_LANE_FN uint32_t take(uint32_t n)      // caller checked have(n), n <= 16
	{
		fill();
		uint32_t v = (uint32_t)bb & ((1u << n) - 1);
		drop(n);
		return v;
	}
};

// per-lane output: literals are collected four at a time once the output position is 4-byte aligned
struct OutWr {
	uint8_t *dst;
	uint32_t out;        // bytes produced (including the pending ones)
	uint64_t wb;         // pending bytes (literals and short matches), they belong to dst[out - wn, out)
	uint32_t wn;         // 0 .. 8
	bool al;             // dst is 4-byte aligned (the copies below use it)
	// Stores, like the loads, are instructions of the whole wavefront whichever lane has something to store, so the pending
	// bytes are written where every lane passes at the same time: commit(), once a trip round the token loop (4 or 8
	// bytes, at whatever address: up to 3 stay pending), and in between only when the 8 are full.
	template <typename V> NXZ_LANE_FN void put(uint32_t at, V v) { __builtin_memcpy((NXZ_LANE_GLOBAL uint8_t *)dst + at, &v, sizeof(V)); }
	NXZ_LANE_FN void lit(uint32_t sym)
	{
		if (wn == 8) { put<uint64_t>(out - 8, wb); wb = 0; wn = 0; }
		wb |= (uint64_t)sym << (8 * wn);
		wn++; out++;
	}
	NXZ_LANE_FN void commit()
	{
		if (wn == 8) { put<uint64_t>(out - 8, wb); wb = 0; wn = 0; }
		else if (wn >= 4) { put<uint32_t>(out - wn, (uint32_t)wb); wb >>= 32; wn -= 4; }
	}
	NXZ_LANE_FN void flush()
	{
		commit();
		uint32_t at = out - wn;
		if (wn & 2) { put<uint16_t>(at, (uint16_t)wb); wb >>= 16; at += 2; }
		if (wn & 1) put<uint8_t>(at, (uint8_t)wb);
		wb = 0; wn = 0;
	}
	// append n <= 8 bytes (low byte first) to the pending ones
	NXZ_LANE_FN void append(uint64_t v, uint32_t n)
	{
		if (n < 8) v &= ((uint64_t)1 << (8 * n)) - 1;
		const uint32_t room = 8 - wn, k = n < room ? n : room;
		if (k) wb |= v << (8 * wn);
		wn += k; out += k;
		if (k < n) {                                                        // the 8 are full: out they go, the rest begins anew
			put<uint64_t>(out - 8, wb);
			wb = k ? v >> (8 * k) : v;
			wn = n - k; out += n - k;
		}
	}
	// a short match (len <= 8) whose source lies clear of the pending bytes (dist >= 16): one 8-byte load, appended -- no
	// flush, no store of its own
	NXZ_LANE_FN void copy_short(uint32_t len, uint32_t dist)
	{
		uint64_t v;
		__builtin_memcpy(&v, (const NXZ_LANE_GLOBAL uint8_t *)dst + out - dist, 8);
		append(v, len);
	}
	// append n bytes of another buffer (stored blocks), pending bytes flushed by the caller: dword
	// stores once the destination is aligned, the source dwords from aligned loads, eight in flight
	NXZ_LANE_FN void copy_in(const uint8_t *s, uint32_t n)
	{
		uint8_t *d = dst + out;
		uint32_t i = 0;
		if (al && n >= 8) {
			for (; (out + i) & 3; i++) d[i] = s[i];
			const uintptr_t sa = (uintptr_t)(s + i);
			const uint32_t *sw = (const uint32_t *)(sa & ~(uintptr_t)3);
			const uint32_t bo = (uint32_t)sa & 3;
			uint32_t k = 0;
			if (bo == 0) {
				for (; i + 32 <= n; i += 32, k += 8) {
					uint32_t w[8];
					for (int j = 0; j < 8; j++) w[j] = sw[k + j];
					for (int j = 0; j < 8; j++) ((uint32_t *)(d + i))[j] = w[j];
				}
				for (; i + 4 <= n; i += 4, k++) *(uint32_t *)(d + i) = sw[k];
			} else {
				// every output dword takes bytes of two source dwords; the second one holds a byte of the
				// source as long as at least 4 bytes are left
				uint32_t lo = sw[0];
				k = 1;
				for (; i + 32 <= n; i += 32, k += 8) {
					uint32_t w[8];
					for (int j = 0; j < 8; j++) w[j] = sw[k + j];
					uint32_t *o = (uint32_t *)(d + i);
					o[0] = NXZ_LANE_ALIGNBYTE(w[0], lo, bo);
					for (int j = 1; j < 8; j++) o[j] = NXZ_LANE_ALIGNBYTE(w[j], w[j - 1], bo);
					lo = w[7];
				}
				for (; i + 4 <= n; i += 4, k++) {
					const uint32_t hi = sw[k];
					*(uint32_t *)(d + i) = NXZ_LANE_ALIGNBYTE(hi, lo, bo);
					lo = hi;
				}
			}
		}
		for (; i < n; i++) d[i] = s[i];
		out += n;
	}
	// copy len bytes from distance dist (1 <= dist <= out), pending bytes flushed by the caller
	NXZ_LANE_FN void copy(uint32_t len, uint32_t dist)
	{
		// A long run of a short period (4 <= dist < 36): what follows the first few periods is also found
		// m periods back, and from 36 bytes back or more the copy keeps eight loads in flight.
		if (al && dist >= 4 && dist < 36 && len >= 96) {
			const uint32_t m = (35 + dist) / dist, first = (m - 1) * dist;
			copy1(first, dist);
			copy1(len - first, m * dist);
			return;
		}
		copy1(len, dist);
	}
	// n bytes (16, 8, 4, 2 or 1) from s to d, any alignment: one load and one store
	template <typename V> static NXZ_LANE_FN void move(uint8_t *d, const uint8_t *s)
	{
		V v;
		__builtin_memcpy(&v, (const NXZ_LANE_GLOBAL uint8_t *)s, sizeof(V));
		__builtin_memcpy((NXZ_LANE_GLOBAL uint8_t *)d, &v, sizeof(V));
	}
	struct V16 { uint64_t a, b; };
	NXZ_LANE_FN void copy1(uint32_t len, uint32_t dist)
	{
		uint8_t *d = dst + out;
		const uint8_t *s = d - dist;
		uint32_t i = 0;
		if (dist >= 16) {
			// 16 bytes a load and a store, whatever the alignment: the lanes of a wavefront copy side by side, the longest
			// match of the 64 sets the number of trips, and every trip is a load and a store instruction of the whole wavefront
			// -- which is what bounds this kernel (nxz_inflate_lanes.hip) -- so the trips are made as few as can be
			for (; i + 16 <= len; i += 16) move<V16>(d + i, s + i);
			if (i < len) {
				if (len >= 16) move<V16>(d + len - 16, s + len - 16);      // the last 16 bytes, some of them a second time
				else {
					const uint32_t r = len - i;
					if (r & 8) { move<uint64_t>(d + i, s + i); i += 8; }
					if (r & 4) { move<uint32_t>(d + i, s + i); i += 4; }
					if (r & 2) { move<uint16_t>(d + i, s + i); i += 2; }
					if (r & 1) d[i] = s[i];
				}
			}
			out += len;
			return;
		}
		if (al && len >= 8 && dist != 3) {
			for (; (out + i) & 3; i++) d[i] = s[i];                             // align the destination (< 4 bytes)
			if (dist >= 4) {
				// source dwords from two aligned loads; the source lies at least 4 bytes behind
				const uintptr_t sa = (uintptr_t)(s + i);
				const uint32_t *sw = (const uint32_t *)(sa & ~(uintptr_t)3);
				const uint32_t bo = (uint32_t)sa & 3;
				uint32_t lo = sw[0];
				uint32_t k = 1;
				// a source at least 36 / 20 bytes behind: eight / four loads in flight per wait instead of one
				if (dist >= 36) {
					for (; i + 32 <= len; i += 32, k += 8) {
						uint32_t w[8];
						for (int j = 0; j < 8; j++) w[j] = sw[k + j];
						uint32_t *o = (uint32_t *)(d + i);
						if (bo) {
							o[0] = NXZ_LANE_ALIGNBYTE(w[0], lo, bo);
							for (int j = 1; j < 8; j++) o[j] = NXZ_LANE_ALIGNBYTE(w[j], w[j - 1], bo);
						} else {
							o[0] = lo;
							for (int j = 1; j < 8; j++) o[j] = w[j - 1];
						}
						lo = w[7];
					}
				}
				if (dist >= 20) {
					for (; i + 16 <= len; i += 16, k += 4) {
						const uint32_t a = sw[k], b = sw[k + 1], c = sw[k + 2], e = sw[k + 3];
						uint32_t *o = (uint32_t *)(d + i);
						if (bo) {
							o[0] = NXZ_LANE_ALIGNBYTE(a, lo, bo); o[1] = NXZ_LANE_ALIGNBYTE(b, a, bo);
							o[2] = NXZ_LANE_ALIGNBYTE(c, b, bo); o[3] = NXZ_LANE_ALIGNBYTE(e, c, bo);
						} else {
							o[0] = lo; o[1] = a; o[2] = b; o[3] = c;
						}
						lo = e;
					}
				}
				for (; i + 4 <= len; i += 4, k++) {
					uint32_t v = lo;
					if (bo) { const uint32_t hi = sw[k]; v = NXZ_LANE_ALIGNBYTE(hi, lo, bo); lo = hi; }
					*(uint32_t *)(d + i) = v;
					if (!bo) lo = sw[k];
				}
			} else {
				// period 1 or 2: one dword pattern
				const uint32_t v = dist == 1 ? s[i] * 0x01010101u : ((uint32_t)s[i] | (uint32_t)s[i + 1] << 8) * 0x00010001u;   // the bytes at distance 1 / 2
				for (; i + 4 <= len; i += 4) *(uint32_t *)(d + i) = v;
			}
		}
		else if (al && len >= 8) {
			// period 3: three dwords that repeat every 12 bytes
			for (; (out + i) & 3; i++) d[i] = s[i];
			const uint32_t p0 = s[i], p1 = s[i + 1], p2 = s[i + 2];
			uint32_t w0 = p0 | p1 << 8 | p2 << 16 | p0 << 24, w1 = p1 | p2 << 8 | p0 << 16 | p1 << 24, w2 = p2 | p0 << 8 | p1 << 16 | p2 << 24;
			for (; i + 4 <= len; i += 4) {
				*(uint32_t *)(d + i) = w0;
				const uint32_t t = w0; w0 = w1; w1 = w2; w2 = t;
			}
		}
		for (; i < len; i++) d[i] = s[i];
		out += len;
	}
};

} // namespace nxzl
#endif
