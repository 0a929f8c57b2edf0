// nxz_blocked.cpp -- blocked gzip files on the batched engine (include/nxz_blocked.h).
//
// Host buffers in, host buffers out: pieces of the input go through pinned staging to the device,
// one batch per piece (two in flight), and come back as finished gzip members that the device
// packed back to back (nxz_batch_pack_gzip).  Plain host C++ over the engine's C ABI; the only
// compute done here is the Huffman table generator (nxz_dhtgen_batch), which the reference keeps
// on the host too (lib/nx_dhtgen.c).
#include "../../include/nxz_blocked.h"
#include "../../include/nxz_config.h"
#include "../../include/nxz_engine.h"
#include "nxz_host.h"
#include <errno.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t MEMBER_OVERHEAD = 26;       // 18-byte header with the BC subfield + CRC32 + ISIZE

struct Ctx {
	nxz_ctx_t *c = nullptr;
	explicit Ctx(int device) { c = nxz_ctx_create(device >= 0 ? device : nxz_config()->dev_num); }
	~Ctx() { if (c) nxz_ctx_destroy(c); }
};

inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }
inline uint32_t rd16(const uint8_t *p) { return p[0] | (uint32_t)p[1] << 8; }
inline uint32_t rd32(const uint8_t *p) { return p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

// device + pinned buffers of one batch in flight
struct Set {
	nxz_ctx_t *c = nullptr;
	void *stream = nullptr;
	std::vector<void *> dev, pin;
	template <class T> T *dmalloc(size_t n) { void *p = nxz_dev_malloc(c, n * sizeof(T)); if (p) dev.push_back(p); return (T *)p; }
	template <class T> T *pmalloc(size_t n) { void *p = nxz_pinned_malloc(c, n * sizeof(T)); if (p) pin.push_back(p); return (T *)p; }
	~Set()
	{
		if (stream) nxz_stream_destroy(c, stream);
		for (void *p : dev) nxz_dev_free(c, p);
		for (void *p : pin) nxz_pinned_free(c, p);
	}
};

struct DeflateSet : Set {
	uint8_t *h_in = nullptr, *d_in = nullptr, *d_slots = nullptr, *d_packed = nullptr, *h_packed = nullptr;
	nxz_batch_job_t *h_jobs = nullptr, *d_jobs = nullptr, *h_lead = nullptr, *d_lead = nullptr;
	nxz_batch_result_t *d_res = nullptr;
	uint64_t *d_off = nullptr, *h_off = nullptr;
	uint32_t *d_cnt = nullptr, *h_cnt = nullptr;
	nxz_batch_dht_t *h_tab = nullptr, *d_tab = nullptr;
	size_t n = 0;                     // blocks of the batch in flight (0 = idle)
};

double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
struct Phase { double t = 0; double t0 = 0; void in() { t0 = now_s(); } void out() { t += now_s() - t0; } };

unsigned host_threads()
{
	unsigned n = std::thread::hardware_concurrency();
	return n ? std::min(n, 16u) : 4;
}

} // namespace

extern "C" int nxz_blocked_end_marker(nxz_sink_fn sink, void *user)
{
	static const uint8_t eof[28] = { 0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 'B', 'C', 0x02, 0, 0x1b, 0,
					 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	return sink(user, eof, sizeof(eof)) ? -EIO : 0;
}

extern "C" int nxz_blocked_deflate(const void *src_, size_t len, const nxz_blocked_opts_t *opts,
				   nxz_sink_fn sink, void *user, uint64_t *out_len)
{
	nxz_blocked_opts_t o;
	memset(&o, 0, sizeof(o));
	o.device = -1;
	if (opts) o = *opts;
	const uint32_t B = o.block_size ? o.block_size : NXZ_BLOCKED_BLOCK;
	if (!sink || (B & 15) || B > NXZ_BLOCKED_BLOCK || (len && !src_)) return -EINVAL;
	if (out_len) *out_len = 0;
	if (len == 0) return 0;
	const uint8_t *src = (const uint8_t *)src_;
	const size_t nblocks = (len + B - 1) / B;
	const size_t chunk = std::min<size_t>(o.chunk_blocks ? o.chunk_blocks : 4096, nblocks);
	const uint32_t G = o.group ? o.group : 64;
	const bool dynamic = !o.fixed;
	const size_t slot = up16(nxz_compress_bound(B));
	const size_t ngmax = (chunk + G - 1) / G;

	Phase t_setup, t_stage, t_tables, t_wait, t_sink;
	t_setup.in();
	Ctx ctx(o.device);
	if (!ctx.c) return -ENODEV;
	DeflateSet sets[2];
	const int nsets = nblocks > chunk ? 2 : 1;
	for (int k = 0; k < nsets; k++) {
		DeflateSet &S = sets[k];
		S.c = ctx.c;
		S.stream = nxz_stream_create(ctx.c);
		S.h_in = S.pmalloc<uint8_t>(chunk * B); S.d_in = S.dmalloc<uint8_t>(chunk * B + 16);
		S.d_slots = S.dmalloc<uint8_t>(chunk * slot);
		S.d_packed = S.dmalloc<uint8_t>(chunk * (slot + MEMBER_OVERHEAD) + 16);
		S.h_packed = S.pmalloc<uint8_t>(chunk * (slot + MEMBER_OVERHEAD) + 16);
		S.h_jobs = S.pmalloc<nxz_batch_job_t>(chunk); S.d_jobs = S.dmalloc<nxz_batch_job_t>(chunk);
		S.d_res = S.dmalloc<nxz_batch_result_t>(chunk);
		S.d_off = S.dmalloc<uint64_t>(chunk + 1); S.h_off = S.pmalloc<uint64_t>(chunk + 1);
		bool ok = S.stream && S.h_in && S.d_in && S.d_slots && S.d_packed && S.h_packed && S.h_jobs && S.d_jobs && S.d_res && S.d_off && S.h_off;
		if (dynamic) {
			S.h_lead = S.pmalloc<nxz_batch_job_t>(ngmax); S.d_lead = S.dmalloc<nxz_batch_job_t>(ngmax);
			S.d_cnt = S.dmalloc<uint32_t>(ngmax * 316); S.h_cnt = S.pmalloc<uint32_t>(ngmax * 316);
			S.h_tab = S.pmalloc<nxz_batch_dht_t>(ngmax); S.d_tab = S.dmalloc<nxz_batch_dht_t>(ngmax);
			ok = ok && S.h_lead && S.d_lead && S.d_cnt && S.h_cnt && S.h_tab && S.d_tab;
		}
		if (!ok) return -ENOMEM;
	}
	t_setup.out();

	uint64_t total_out = 0;
	// queue one batch: upload, compress, pack, fetch the offsets
	auto start = [&](DeflateSet &S, size_t first, size_t n) -> int {
		const size_t bytes = std::min(len - first * B, n * (size_t)B);
		t_stage.in();
		memcpy(S.h_in, src + first * B, bytes);
		t_stage.out();
		int rc = nxz_copy_to_device(ctx.c, S.d_in, S.h_in, bytes, S.stream);
		for (size_t i = 0; i < n; i++) {
			nxz_batch_job_t &j = S.h_jobs[i];
			memset(&j, 0, sizeof(j));
			j.src = S.d_in + i * B;
			j.dst = S.d_slots + i * slot;
			j.src_len = (uint32_t)std::min<size_t>(B, bytes - i * B);
			j.dst_cap = (uint32_t)slot;
			j.in_adler = 1;
			j.dht_index = (uint32_t)(i / G);
		}
		if (!rc) rc = nxz_copy_to_device(ctx.c, S.d_jobs, S.h_jobs, n * sizeof(nxz_batch_job_t), S.stream);
		if (!rc && dynamic) {
			// table of a group = what the generator makes of the LZ77 symbol counts of the group's
			// first block (the reference reuses a table over neighbouring data too, lib/nx_dht.c:480-566)
			const size_t ng = (n + G - 1) / G;
			for (size_t g = 0; g < ng; g++) S.h_lead[g] = S.h_jobs[g * G];
			rc = nxz_copy_to_device(ctx.c, S.d_lead, S.h_lead, ng * sizeof(nxz_batch_job_t), S.stream);
			if (!rc) rc = nxz_batch_compress(ctx.c, NXZ_FC_COMPRESS_FHT_COUNT, S.d_lead, ng, nullptr, 0, S.d_res, S.d_cnt, S.stream);
			if (!rc) rc = nxz_copy_to_host(ctx.c, S.h_cnt, S.d_cnt, ng * 316 * sizeof(uint32_t), S.stream);
			t_wait.in();
			if (!rc) rc = nxz_ctx_sync(ctx.c, S.stream);
			t_wait.out();
			t_tables.in();
			if (!rc && nxz_dhtgen_batch(S.h_cnt, ng, S.h_tab, (int)host_threads())) rc = -EIO;
			t_tables.out();
			if (!rc) rc = nxz_copy_to_device(ctx.c, S.d_tab, S.h_tab, ng * sizeof(nxz_batch_dht_t), S.stream);
			if (!rc) rc = nxz_batch_compress(ctx.c, NXZ_FC_COMPRESS_DHT, S.d_jobs, n, S.d_tab, ng, S.d_res, nullptr, S.stream);
		} else if (!rc) {
			rc = nxz_batch_compress(ctx.c, NXZ_FC_COMPRESS_FHT, S.d_jobs, n, nullptr, 0, S.d_res, nullptr, S.stream);
		}
		if (!rc) rc = nxz_batch_pack_gzip(ctx.c, S.d_jobs, S.d_res, n, S.d_off, S.d_packed, S.stream);
		if (!rc) rc = nxz_copy_to_host(ctx.c, S.h_off, S.d_off, (n + 1) * sizeof(uint64_t), S.stream);
		S.n = n;
		return rc;
	};
	// wait for it, fetch the packed members and hand them on
	auto finish = [&](DeflateSet &S) -> int {
		if (!S.n) return 0;
		t_wait.in();
		int rc = nxz_ctx_sync(ctx.c, S.stream);
		const uint64_t bytes = S.h_off[S.n];
		S.n = 0;
		if (!rc) rc = nxz_copy_to_host(ctx.c, S.h_packed, S.d_packed, bytes, S.stream);
		if (!rc) rc = nxz_ctx_sync(ctx.c, S.stream);
		t_wait.out();
		t_sink.in();
		if (!rc && sink(user, S.h_packed, bytes)) rc = -EIO;
		t_sink.out();
		if (!rc) total_out += bytes;
		return rc;
	};

	int rc = 0;
	size_t k = 0;
	for (size_t first = 0; first < nblocks && !rc; first += chunk, k++) {
		rc = start(sets[k & 1 & (nsets - 1)], first, std::min(chunk, nblocks - first));
		if (!rc && k > 0) rc = finish(sets[(k - 1) & 1]);
	}
	if (!rc && k > 0) rc = finish(sets[(k - 1) & 1 & (nsets - 1)]);
	if (rc) for (auto &S : sets) if (S.stream) nxz_ctx_sync(ctx.c, S.stream);
	if (out_len) *out_len = total_out;
	nxz_log(2, "nxz_blocked_deflate: %zu bytes in %zu blocks: setup %.3f s, staging copies %.3f s, tables %.3f s, waiting for the device %.3f s, sink %.3f s\n",
		len, nblocks, t_setup.t, t_stage.t, t_tables.t, t_wait.t, t_sink.t);
	return rc;
}

// ---------------------------------------------------------------------------------------------
namespace {

struct Member { size_t pay; uint32_t paylen, isize, crc; };

// one member with the BC subfield at p: its total size, or 0
size_t member_size(const uint8_t *p, size_t left)
{
	if (left < MEMBER_OVERHEAD || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || p[3] != 4) return 0;     // FLG = FEXTRA only
	const uint32_t xlen = rd16(p + 10);
	if (xlen < 6 || 12 + (size_t)xlen + 8 > left) return 0;
	// walk the subfields for SI1='B' SI2='C' SLEN=2
	for (uint32_t q = 0; q + 4 <= xlen;) {
		const uint8_t *f = p + 12 + q;
		const uint32_t slen = rd16(f + 2);
		if (f[0] == 'B' && f[1] == 'C' && slen == 2 && q + 6 <= xlen) {
			const size_t size = (size_t)rd16(f + 4) + 1;
			return size >= 12 + (size_t)xlen + 8 && size <= left ? size : 0;
		}
		q += 4 + slen;
	}
	return 0;
}

size_t scan(const uint8_t *p, size_t len, std::vector<Member> *out, uint64_t *members, uint64_t *usize)
{
	size_t pos = 0;
	uint64_t n = 0, u = 0;
	while (pos < len) {
		const size_t size = member_size(p + pos, len - pos);
		if (!size) break;
		const uint32_t xlen = rd16(p + pos + 10);
		Member m;
		m.pay = pos + 12 + xlen;
		m.paylen = (uint32_t)(size - 12 - xlen - 8);
		m.crc = rd32(p + pos + size - 8);
		m.isize = rd32(p + pos + size - 4);
		if (out) out->push_back(m);
		n++; u += m.isize;
		pos += size;
	}
	if (members) *members = n;
	if (usize) *usize = u;
	return pos;
}

struct InflateSet : Set {
	uint8_t *h_in = nullptr, *d_in = nullptr, *d_out = nullptr, *h_out = nullptr;
	nxz_batch_job_t *h_jobs = nullptr, *d_jobs = nullptr;
	nxz_batch_result_t *d_res = nullptr, *h_res = nullptr;
	size_t first = 0, n = 0;
};

} // namespace

extern "C" int nxz_blocked_scan(const void *src, size_t len, uint64_t *members, uint64_t *usize, size_t *consumed)
{
	if (len && !src) return -EINVAL;
	size_t c = scan((const uint8_t *)src, len, nullptr, members, usize);
	if (consumed) *consumed = c;
	return 0;
}

extern "C" int nxz_blocked_inflate(const void *src_, size_t len, const nxz_blocked_opts_t *opts,
				   nxz_sink_fn sink, void *user, uint64_t *out_len, size_t *consumed)
{
	nxz_blocked_opts_t o;
	memset(&o, 0, sizeof(o));
	o.device = -1;
	if (opts) o = *opts;
	if (!sink || (len && !src_)) return -EINVAL;
	if (out_len) *out_len = 0;
	if (consumed) *consumed = 0;
	const uint8_t *src = (const uint8_t *)src_;
	std::vector<Member> mem;
	const size_t used = scan(src, len, &mem, nullptr, nullptr);
	if (mem.empty()) return 1;
	if (consumed) *consumed = used;
	// deflate cannot expand by more than 1032 : 1 (258 bytes from a 2-bit match): an ISIZE beyond that
	// is damage, not a reason to allocate gigabytes
	for (const Member &m : mem) if ((uint64_t)m.isize > (uint64_t)m.paylen * 1032 + 64) return -EILSEQ;

	// batches of at most `chunk` members; staging sized for the largest batch (8192 members take
	// the stream-per-wave kernel about 45 ms)
	const size_t chunk = std::min<size_t>(o.chunk_blocks ? o.chunk_blocks : 8192, mem.size());
	size_t max_in = 0, max_out = 0;
	for (size_t f = 0; f < mem.size(); f += chunk) {
		size_t in = 0, out = 0;
		for (size_t i = f; i < std::min(mem.size(), f + chunk); i++) { in += up16(mem[i].paylen) + 16; out += up16(mem[i].isize); }
		max_in = std::max(max_in, in); max_out = std::max(max_out, out);
	}
	Phase t_setup, t_stage, t_wait, t_sink;
	t_setup.in();
	Ctx ctx(o.device);
	if (!ctx.c) return -ENODEV;
	InflateSet sets[2];
	const int nsets = mem.size() > chunk ? 2 : 1;
	for (int k = 0; k < nsets; k++) {
		InflateSet &S = sets[k];
		S.c = ctx.c;
		S.stream = nxz_stream_create(ctx.c);
		S.h_in = S.pmalloc<uint8_t>(max_in + 16); S.d_in = S.dmalloc<uint8_t>(max_in + 16);
		S.d_out = S.dmalloc<uint8_t>(max_out + 16); S.h_out = S.pmalloc<uint8_t>(max_out + 16);
		S.h_jobs = S.pmalloc<nxz_batch_job_t>(chunk); S.d_jobs = S.dmalloc<nxz_batch_job_t>(chunk);
		S.d_res = S.dmalloc<nxz_batch_result_t>(chunk); S.h_res = S.pmalloc<nxz_batch_result_t>(chunk);
		if (!(S.stream && S.h_in && S.d_in && S.d_out && S.h_out && S.h_jobs && S.d_jobs && S.d_res && S.h_res)) return -ENOMEM;
	}
	t_setup.out();

	uint64_t total = 0;
	auto start = [&](InflateSet &S, size_t first, size_t n) -> int {
		// payloads at 16-byte aligned places of the staging buffer, outputs at 16-byte aligned places too
		size_t in = 0, out = 0, jobs = 0;
		t_stage.in();
		for (size_t i = first; i < first + n; i++) {
			const Member &m = mem[i];
			if (m.isize == 0 && m.crc == 0) continue;                      // empty member (the end marker)
			memcpy(S.h_in + in, src + m.pay, m.paylen);
			nxz_batch_job_t &j = S.h_jobs[jobs++];
			memset(&j, 0, sizeof(j));
			j.src = S.d_in + in; j.src_len = m.paylen;
			j.dst = S.d_out + out; j.dst_cap = m.isize;
			j.in_adler = 1;
			j.reserved = (uint32_t)(i - first);
			in += up16(m.paylen) + 16; out += up16(m.isize);
		}
		t_stage.out();
		S.first = first; S.n = jobs;
		if (!jobs) return 0;
		int rc = nxz_copy_to_device(ctx.c, S.d_in, S.h_in, in, S.stream);
		if (!rc) rc = nxz_copy_to_device(ctx.c, S.d_jobs, S.h_jobs, jobs * sizeof(nxz_batch_job_t), S.stream);
		if (!rc) rc = nxz_batch_decompress(ctx.c, S.d_jobs, jobs, S.d_res, nullptr, S.stream);
		if (!rc) rc = nxz_copy_to_host(ctx.c, S.h_res, S.d_res, jobs * sizeof(nxz_batch_result_t), S.stream);
		if (!rc) rc = nxz_copy_to_host(ctx.c, S.h_out, S.d_out, out, S.stream);
		return rc;
	};
	auto finish = [&](InflateSet &S) -> int {
		if (!S.n) return 0;
		t_wait.in();
		int rc = nxz_ctx_sync(ctx.c, S.stream);
		t_wait.out();
		const size_t jobs = S.n;
		S.n = 0;
		if (rc) return rc;
		t_sink.in();
		struct Done { Phase &p; ~Done() { p.out(); } } done{t_sink};
		for (size_t q = 0; q < jobs; q++) {
			const Member &m = mem[S.first + S.h_jobs[q].reserved];
			const nxz_batch_result_t &r = S.h_res[q];
			// the whole payload is one deflate stream that ends with its final block
			if (r.cc != 0 || !(r.sfbt & 0x100) || r.tpbc != m.isize || r.crc != m.crc) return -EILSEQ;
		}
		// runs of outputs that sit back to back (blocks of a multiple of 16 bytes) go out in one piece
		size_t pos = 0, run0 = 0, runlen = 0;
		for (size_t q = 0; q < jobs; q++) {
			const uint32_t isize = mem[S.first + S.h_jobs[q].reserved].isize;
			if (pos != run0 + runlen) { if (runlen && sink(user, S.h_out + run0, runlen)) return -EIO; run0 = pos; runlen = 0; }
			runlen += isize; total += isize;
			pos += up16(isize);
		}
		if (runlen && sink(user, S.h_out + run0, runlen)) return -EIO;
		return 0;
	};

	int rc = 0;
	size_t k = 0;
	for (size_t first = 0; first < mem.size() && !rc; first += chunk, k++) {
		rc = start(sets[k & 1 & (nsets - 1)], first, std::min(chunk, mem.size() - first));
		if (!rc && k > 0) rc = finish(sets[(k - 1) & 1]);
	}
	if (!rc && k > 0) rc = finish(sets[(k - 1) & 1 & (nsets - 1)]);
	if (rc) for (auto &S : sets) if (S.stream) nxz_ctx_sync(ctx.c, S.stream);
	if (out_len) *out_len = total;
	nxz_log(2, "nxz_blocked_inflate: %zu members: setup %.3f s, staging copies %.3f s, waiting for the device %.3f s, checks + sink %.3f s\n",
		mem.size(), t_setup.t, t_stage.t, t_wait.t, t_sink.t);
	return rc;
}
