// CPU fuzz of the lane-per-stream inflate kernel's bit reader and output writer
// (power-gzip_amd/csrc/nxz_lane_io.h) against bytewise references, run under AddressSanitizer by
// tests/test_lane_io.py.  Buffers are sized to the 4-byte words that cover them: the helpers may
// read the word that holds the first / last byte, never a word beyond.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "nxz_lane_io.h"
using namespace nxzl;

// what fill() promises: bb holds the stream's bits [pos, pos + bc), zeros above them and past the end of the source
static bool fill_ok(const BitRd &b, uint64_t pos, uint32_t least)
{
	if (b.pos != pos || b.bc > 64 || b.bc < least) return false;
	for (uint32_t i = 0; i < b.bc; i++) {
		const uint64_t bi = pos + i;
		const uint32_t ref = (bi >> 3) < b.srclen ? (b.src[bi >> 3] >> (bi & 7)) & 1 : 0;
		if (((b.bb >> i) & 1) != ref) return false;
	}
	return b.bc == 64 || (b.bb >> b.bc) == 0;
}

int main()
{
	srand(1);
	for (int it = 0; it < 200000; it++) {
		uint32_t n = rand() % 70, off = rand() % 8;
		std::vector<uint8_t> buf(((off + n + 3) & ~3u) ? ((off + n + 3) & ~3u) : 4);
		for (auto &x : buf) x = (uint8_t)rand();
		BitRd a{ buf.data() + off, n, 0, 0, (uint64_t)(rand() % (n * 8 + 20)) };
		uint64_t pos = a.pos;
		bool inside = (pos & 7) != 0;
		for (int k = 0; k < 12; k++) {
			uint32_t least = (rand() & 1) ? 32 : 1 + rand() % 32;
			if (inside && least > 25) least = 25;
			if (rand() % 4 == 0) a.refill();
			if (least == 32) a.fill(); else a.need(least);
			if (!fill_ok(a, pos, least)) { printf("fill mismatch (case %d step %d, bc %u)\n", it, k, a.bc); return 1; }
			uint32_t dr = rand() % ((rand() & 1) ? 17 : 49); if (dr > a.bc) dr = a.bc;
			a.drop(dr); pos += dr;
			if (a.bc < 32) inside = false;                      // the next fill starts at a byte
			if (rand() % 9 == 0) { a.pos = pos = (uint64_t)(rand() % (n * 8 + 20)); a.sync(); inside = (pos & 7) != 0; }
		}
	}
	for (int it = 0; it < 40000; it++) {
		uint32_t cap = 64 + rand() % 900, off = (rand() % 2) ? 0 : rand() % 4;
		std::vector<uint8_t> m1((off + cap + 3) & ~3u), m2(off + cap);
		uint8_t *ref = m2.data() + off; uint32_t rout = 0;
		OutWr w{ m1.data() + off, 0, 0, 0, (((uintptr_t)(m1.data() + off)) & 3) == 0 };
		for (;;) {
			if (rand() % 3 && w.out) {
				uint32_t dist = 1 + rand() % (w.out < 40 ? w.out : 40);
				if (rand() % 4 == 0) dist = 1 + rand() % w.out;
				uint32_t len = 3 + rand() % ((rand() & 1) ? 8 : 256); if (len > cap - w.out) break;
				if (len <= 8 && dist >= 16) w.copy_short(len, dist);                 // as the kernel chooses
				else { w.flush(); w.copy(len, dist); }
				for (uint32_t i = 0; i < len; i++) ref[rout + i] = ref[rout + i - dist];
				rout += len;
			} else {
				if (w.out >= cap) break;
				uint8_t c = (uint8_t)rand(); w.lit(c); ref[rout++] = c;
			}
			if (rand() % 3 == 0) w.commit();
		}
		w.flush();
		if (w.out != rout || memcmp(m1.data() + off, ref, rout)) { printf("output mismatch, case %d\n", it); return 1; }
	}
	// copy_in: bytes of another buffer (stored blocks), all alignments of both sides
	for (int it = 0; it < 60000; it++) {
		uint32_t n = rand() % 200, soff = rand() % 8, doff = (rand() % 2) ? 0 : rand() % 4, pre = rand() % 9;
		std::vector<uint8_t> sb((soff + n + 3) & ~3u ? (soff + n + 3) & ~3u : 4), m1((doff + pre + n + 3) & ~3u ? (doff + pre + n + 3) & ~3u : 4);
		for (auto &x : sb) x = (uint8_t)rand();
		OutWr w{ m1.data() + doff, 0, 0, 0, (((uintptr_t)(m1.data() + doff)) & 3) == 0 };
		std::vector<uint8_t> ref;
		for (uint32_t k = 0; k < pre; k++) { uint8_t c = (uint8_t)rand(); w.lit(c); ref.push_back(c); }
		w.flush();
		w.copy_in(sb.data() + soff, n);
		ref.insert(ref.end(), sb.begin() + soff, sb.begin() + soff + n);
		if (w.out != ref.size() || memcmp(m1.data() + doff, ref.data(), ref.size())) { printf("copy_in mismatch, case %d\n", it); return 1; }
	}
	printf("ok\n");
	return 0;
}
