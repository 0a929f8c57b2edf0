// inflate_wg_sim.cpp -- TEST INFRASTRUCTURE: the workgroup-per-stream inflate kernel (power-gzip_amd/csrc/nxz_inflate_wg.hip,
// the product source itself) run on the CPU through tests/native/hip_cpu_shim.h, an OS thread per lane, against streams made
// by system zlib.  Checks: every stream the kernel takes comes out byte for byte with the result record of a finished
// stream; every stream it does not take is on the hand-back list (and only those); damaged streams are handed back, never
// "decoded".
//   usage: inflate_wg_sim <file with sample text> [seed] [pmin_bits] [pieces a wavefront walks] [bytes of the long cases]
#include "hip_cpu_shim.h"
#include "../../power-gzip_amd/csrc/nxz_inflate_wg.hip"
#include <zlib.h>
#include <stdio.h>
#include <string>
#include <vector>

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &in, int level, int strategy, int memlevel = 8)
{
	z_stream z;
	memset(&z, 0, sizeof(z));
	if (deflateInit2(&z, level, Z_DEFLATED, -15, memlevel, strategy) != Z_OK) abort();
	std::vector<uint8_t> out(deflateBound(&z, in.size()) + 64);
	z.next_in = (Bytef *)in.data(); z.avail_in = (uInt)in.size();
	z.next_out = out.data(); z.avail_out = (uInt)out.size();
	if (deflate(&z, Z_FINISH) != Z_STREAM_END) abort();
	out.resize(z.total_out);
	deflateEnd(&z);
	return out;
}

struct Case { std::string name; std::vector<uint8_t> plain, stream; bool expect_taken; uint32_t src_off; uint32_t cap; };

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: %s <text file> [seed] [pmin_bits]\n", argv[0]); return 2; }
	std::vector<uint8_t> text;
	{
		FILE *f = fopen(argv[1], "rb");
		if (!f) { perror(argv[1]); return 2; }
		uint8_t buf[65536]; size_t k;
		while ((k = fread(buf, 1, sizeof(buf), f)) > 0) text.insert(text.end(), buf, buf + k);
		fclose(f);
	}
	if (argc > 2) rng_state ^= (uint64_t)strtoull(argv[2], nullptr, 0) * 0x9E3779B97F4A7C15ull;
	const uint32_t pmin = argc > 3 ? (uint32_t)atoi(argv[3]) : 512;
	const uint32_t nres = argc > 4 ? (uint32_t)atoi(argv[4]) : 16;          // (so many pieces left in a round or fewer: a wavefront walks each)
	const size_t longn = argc > 5 ? (size_t)atoi(argv[5]) : 600000;      // bytes of the long cases (the simulation takes a minute per MiB)
	auto slice = [&](size_t at, size_t n) { return std::vector<uint8_t>(text.begin() + at % (text.size() - n), text.begin() + at % (text.size() - n) + n); };
	std::vector<Case> cases;
	auto add = [&](const char *name, std::vector<uint8_t> plain, int level, int strategy, bool taken = true, uint32_t off = 0, uint32_t cap = 0) {
		Case c; c.name = name; c.plain = plain; c.stream = deflate_raw(plain, level, strategy); c.expect_taken = taken; c.src_off = off; c.cap = cap ? cap : 65536;
		cases.push_back(c);
	};
	add("text 64K -6", slice(rnd(), 65536), 6, Z_DEFAULT_STRATEGY);
	add("text 64K -1", slice(rnd(), 65536), 1, Z_DEFAULT_STRATEGY);
	add("text 64K fixed", slice(rnd(), 65536), 6, Z_FIXED);
	add("text 64K huffman only", slice(rnd(), 65536), 6, Z_HUFFMAN_ONLY, true, 5);
	add("text 20000 -9", slice(rnd(), 20000), 9, Z_DEFAULT_STRATEGY, true, 12);
	add("text 300 -6", slice(rnd(), 300), 6, Z_DEFAULT_STRATEGY);
	add("one byte", std::vector<uint8_t>(1, 'x'), 6, Z_DEFAULT_STRATEGY);
	add("empty", std::vector<uint8_t>(), 6, Z_DEFAULT_STRATEGY);
	add("zeros 64K", std::vector<uint8_t>(65536, 0), 6, Z_DEFAULT_STRATEGY);
	{
		std::vector<uint8_t> v(65536);
		for (auto &b : v) b = (uint8_t)rnd();
		add("random 64K stored", v, 6, Z_DEFAULT_STRATEGY);
		add("random 64K huffman", v, 6, Z_HUFFMAN_ONLY, true, 3);
		add("random 4000 -0", std::vector<uint8_t>(v.begin(), v.begin() + 4000), 0, Z_DEFAULT_STRATEGY);
		for (size_t i = 0; i < v.size(); i++) v[i] = (uint8_t)("abc"[i % 3]);
		add("period 3", v, 6, Z_DEFAULT_STRATEGY);
		for (size_t i = 0; i < v.size(); i++) v[i] = (uint8_t)((i % 2) ? rnd() & 3 : 'a' + (rnd() & 1));
		add("low entropy rle", v, 6, Z_RLE);
		// 16-bit tables of slowly varying values (image-like), packed records
		for (size_t i = 0; i < v.size(); i += 2) { const uint32_t x = 1000 + (uint32_t)(i / 7) + (rnd() & 7); v[i] = (uint8_t)x; v[i + 1] = (uint8_t)(x >> 8); }
		add("image-like -6", v, 6, Z_DEFAULT_STRATEGY, true, 8);
		// mixed: text, zeros, random, text again with long-distance repeats
		std::vector<uint8_t> m = slice(rnd(), 20000);
		m.insert(m.end(), 9000, 0);
		for (int i = 0; i < 6000; i++) m.push_back((uint8_t)rnd());
		std::vector<uint8_t> again(m.begin() + 100, m.begin() + 20100);
		m.insert(m.end(), again.begin(), again.end());
		m.resize(65536, 'q');
		add("mixed -6", m, 6, Z_DEFAULT_STRATEGY);
		add("mixed -1 memlevel", m, 1, Z_DEFAULT_STRATEGY, true, 1);
	}
	// what the kernel must hand back
	// streams longer than LDS: in spans, the output flushed 32 KiB and more at a time
	add("text 70000 -6", slice(rnd(), 70000), 6, Z_DEFAULT_STRATEGY);
	{
		std::vector<uint8_t> big;
		while (big.size() < longn) { const std::vector<uint8_t> s = slice(rnd(), 50000); big.insert(big.end(), s.begin(), s.end()); }
		add("text long -6", big, 6, Z_DEFAULT_STRATEGY, true, 7);
		add("text long -1", big, 1, Z_DEFAULT_STRATEGY);
		add("text long fixed", big, 6, Z_FIXED, true, 2);
		add("text 300K huffman only", std::vector<uint8_t>(big.begin(), big.begin() + longn / 2), 6, Z_HUFFMAN_ONLY);
		std::vector<uint8_t> v(longn / 2 > 65536 ? longn / 2 : 65536);
		for (auto &b : v) b = (uint8_t)rnd();
		add("random 300K stored", v, 6, Z_DEFAULT_STRATEGY, true, 9);
		add("random 300K -0", v, 0, Z_DEFAULT_STRATEGY);
		add("random 64K fixed", std::vector<uint8_t>(v.begin(), v.begin() + 65536), 6, Z_FIXED);
		add("zeros 1M", std::vector<uint8_t>(longn * 2, 0), 6, Z_DEFAULT_STRATEGY);
		for (size_t i = 0; i < v.size(); i++) v[i] = (uint8_t)("abcdefghijklmnopqrstuvwxyz0123456"[i % 33]);
		add("period 33, 300K", v, 6, Z_DEFAULT_STRATEGY);
		std::vector<uint8_t> m = big;
		for (size_t i = longn / 6; i < longn / 3; i++) m[i] = (uint8_t)rnd();
		for (size_t i = longn / 2; i < longn / 2 + longn / 8; i++) m[i] = 0;
		add("mixed long -6", m, 6, Z_DEFAULT_STRATEGY, true, 15);
		add("mixed long -9", m, 9, Z_DEFAULT_STRATEGY);
		add("long, target too small", std::vector<uint8_t>(big.begin(), big.begin() + longn / 3), 6, Z_DEFAULT_STRATEGY, false, 0, (uint32_t)(longn / 4));
	}
	add("target too small", slice(rnd(), 30000), 6, Z_DEFAULT_STRATEGY, false, 0, 29999);
	{
		Case c = cases[0]; c.name = "cut short"; c.stream.resize(c.stream.size() / 2); c.expect_taken = false; cases.push_back(c);
		Case d = cases[0]; d.name = "damaged"; d.expect_taken = false;
		for (size_t i = 200; i < d.stream.size(); i += 97) d.stream[i] ^= 0x5a;
		cases.push_back(d);
		Case e = cases[2]; e.name = "trailing bytes"; e.stream.insert(e.stream.end(), 8, 0xee); cases.push_back(e);
	}

	const size_t n = cases.size();
	std::vector<nxz_batch_job_t> jobs(n);
	std::vector<nxz_batch_result_t> res(n);
	std::vector<std::vector<uint8_t>> srcbuf(n), dstbuf(n);
	for (size_t i = 0; i < n; i++) {
		Case &c = cases[i];
		srcbuf[i].assign(c.stream.size() + 64 + 16, 0xa5);
		uint8_t *base = (uint8_t *)(((uintptr_t)srcbuf[i].data() + 15) & ~(uintptr_t)15) + c.src_off;
		memcpy(base, c.stream.data(), c.stream.size());
		dstbuf[i].assign((c.plain.size() > 65536 ? c.plain.size() : 65536) + 5000 + 32, 0xcd);
		uint8_t *dst = (uint8_t *)(((uintptr_t)dstbuf[i].data() + 15) & ~(uintptr_t)15);
		memset(&jobs[i], 0, sizeof(jobs[i]));
		jobs[i].src = base; jobs[i].dst = dst; jobs[i].src_len = (uint32_t)c.stream.size(); jobs[i].dst_cap = c.cap == 65536 ? (uint32_t)(c.plain.size() > 65536 ? c.plain.size() : 65536) : c.cap;
		jobs[i].in_adler = 1;
		memset(&res[i], 0xff, sizeof(res[i]));
	}
	std::vector<uint32_t> bail(64 + n, 0), dbg(16, 0);
	uint32_t ctr = 0;
	hipsim_run_block(0, 1, nxzw::NT, [&] { nxzw::inflate_wg_kernel<false>(jobs.data(), (uint32_t)n, res.data(), nullptr, &ctr, bail.data(), pmin | 200u << 16, nres, dbg.data(), nullptr); });

	int bad = 0;
	std::vector<bool> handed(n, false);
	for (uint32_t k = 0; k < bail[0]; k++) handed[bail[64 + k]] = true;
	for (size_t i = 0; i < n; i++) {
		const Case &c = cases[i];
		if (handed[i] != !c.expect_taken) { printf("FAIL %s: %s\n", c.name.c_str(), handed[i] ? "handed back" : "taken, should have been handed back"); bad++; continue; }
		if (handed[i]) continue;
		const uint8_t *dst = jobs[i].dst;
		const bool trailing = c.name == "trailing bytes";
		if (res[i].tpbc != c.plain.size() || memcmp(dst, c.plain.data(), c.plain.size()) != 0) {
			size_t at = 0;
			while (at < c.plain.size() && at < res[i].tpbc && dst[at] == c.plain[at]) at++;
			printf("FAIL %s: output differs (tpbc %u, expected %zu, first difference at %zu)\n", c.name.c_str(), res[i].tpbc, c.plain.size(), at);
			bad++; continue;
		}
		if (dst[c.plain.size()] != 0xcd && (c.plain.size() & 15) == 0) { printf("FAIL %s: wrote behind the output\n", c.name.c_str()); bad++; }
		const uint32_t want_cc = trailing ? 3u : 0u;
		if (res[i].cc != want_cc || res[i].sfbt != 0x100 || res[i].spbc != jobs[i].src_len || res[i].tebc != 0 || (!trailing && res[i].subc >= 8) || (trailing && (res[i].subc < 64 || res[i].subc >= 72))) {
			printf("FAIL %s: result cc %u sfbt %#x spbc %u subc %u\n", c.name.c_str(), res[i].cc, res[i].sfbt, res[i].spbc, res[i].subc);
			bad++;
		}
	}
	printf("%zu streams, %u handed back, reasons:", n, bail[0]);
	for (int r = 1; r < 12; r++) printf(" %u", dbg[r]);
	printf("\n%s\n", bad ? "FAILED" : "ok");
	return bad ? 1 : 0;
}
