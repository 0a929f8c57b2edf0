// nxz_stream.cpp -- zlib-style streaming (nx_deflate / nx_inflate), one-shot calls and software
// checksums on top of the engine boundary (nxu_run_job and friends, include/nxz_engine.h).
//
// Host-side counterpart of the reference's stream engines; behaviour follows
//   lib/nx_deflate.c   framing :158-468, init/reset :488-699, driver :1628-1901, bound :1909
//   lib/nx_inflate.c   reset/init :134-260, header state machine :332-747, trailer check
//                      :763-848, job loop and resume handling :1060-1762
//   lib/nx_compress.c :26-75, lib/nx_uncompr.c :32-88, lib/nx_crc.c, lib/nx_adler32.c
// It is a re-design, not a transcription: every job's output is staged in a stream-owned
// buffer and drained to next_out (the reference scatters into next_out + fifo_out), framing bits
// are appended by a small bit writer, and a job is one 64 KiB sub-block (the engine's unit).
// Deliberate deviations from the reference's quirks: a trailer checksum
// mismatch returns Z_DATA_ERROR (Q9: the reference returns Z_STREAM_ERROR).
#include <zlib.h>
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <new>
#include <vector>
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <atomic>
#include "nxz_host.h"
#include "../../include/nxz_config.h"
#include "../../include/nxz_wire.h"
#include "../../include/nxz_zlib.h"

namespace {

constexpr uint64_t MAGIC_DEF = 0x6e787a2d64656621ull, MAGIC_INF = 0x6e787a2d696e6621ull;
constexpr uint32_t JOB_UNIT = 65536;          // engine sub-block incl. history
constexpr uint32_t WINDOW = 32768;
constexpr uint32_t STORED_MAX = 60000;        // lib/nx_deflate.c:125
enum { HDR_RAW = 0, HDR_ZLIB = 1, HDR_GZIP = 2 };

struct JobBuf {
	nxz_crb_cpb_t *job = nullptr;
	nxz_dde_t *ddl = nullptr;            // 8 entries for an indirect source list
	JobBuf()
	{
		void *p = nullptr;
		if (posix_memalign(&p, 2048, sizeof(nxz_crb_cpb_t)) == 0) job = (nxz_crb_cpb_t *)p;
		if (posix_memalign(&p, 16, 8 * sizeof(nxz_dde_t)) == 0) ddl = (nxz_dde_t *)p;
	}
	~JobBuf() { free(job); free(ddl); }
	JobBuf(const JobBuf &) = delete;
};

struct Engine {
	nxz_dev_t dev;
	bool open = false;
	bool begin()
	{
		memset(&dev, 0, sizeof(dev));
		open = nx_function_begin(NXZ_FUNC_COMP_GZIP, nxz_config()->dev_num, &dev) == 0;   // NX_GZIP_DEV_NUM
		// The reference opens its device once per process (nx_hw_init, lib/nx_zlib.c:1015-1090) and
		// every stream shares that handle; here the first stream leaves one handle open for the life
		// of the process, so that the engine context (streams, staging buffers) outlives the
		// init/end pair of every nx_compress2 call instead of being rebuilt each time.
		// (one such handle per device: with NX_GZIP_DEV_NUM = -1 the threads of a process spread over the devices)
		if (open) {
			static std::once_flag once[64];
			static nxz_dev_t keep[64];
			const int d = dev.fd - 1;                                   // (the handle's fd is the device ordinal + 1)
			if (d >= 0 && d < 64)
				std::call_once(once[d], [d] { memset(&keep[d], 0, sizeof(keep[d])); (void)nx_function_begin(NXZ_FUNC_COMP_GZIP, d, &keep[d]); });
		}
		return open;
	}
	void end() { if (open) nx_function_end(&dev); open = false; }
	// nx_submit_job (lib/nx_zlib.c:469-501): clear CSB and the spbc words, run, return CC
	int submit(nxz_crb_cpb_t *j)
	{
		memset((void *)&j->crb.csb, 0, sizeof(j->crb.csb));
		nxz_wr64(&j->crb.csb_address_be, (uint64_t)(uintptr_t)&j->crb.csb & ~15ull);
		j->cpb.u.out_spbc_be = 0; j->cpb.out_spbc_with_count_be = 0; j->cpb.u.d.out_spbc_decomp_be = 0;
		const uint64_t t0 = nxz_ticks();
		if (nxu_run_job(j, &dev)) return -1;
		nxz_device_stats(t0, nxz_ticks());                 // lib/nx_zlib.c:493-499: the job delay feeds AUTO mode
		return (int)nxz_csb_cc(j);
	}
};

uint32_t sw_adler32(uint32_t adler, const uint8_t *p, size_t n)
{
	uint32_t a = adler & 0xffff, b = adler >> 16;
	while (n) {
		size_t k = std::min<size_t>(n, 5552);
		n -= k;
		while (k--) { a += *p++; b += a; }
		a %= 65521; b %= 65521;
	}
	return (b << 16) | a;
}

// header bytes only (a few dozen): bitwise
uint32_t sw_crc32(uint32_t crc, const uint8_t *p, size_t n)
{
	crc = ~crc;
	while (n--) {
		crc ^= *p++;
		for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ ((crc & 1) ? 0xedb88320u : 0);
	}
	return ~crc;
}

uint32_t gf2_mul(uint32_t a, uint32_t b)
{
	uint32_t r = 0;
	for (int i = 0; i < 32; i++, b <<= 1) {
		if (b & 0x80000000u) r ^= a;
		a = (a >> 1) ^ ((a & 1) ? 0xedb88320u : 0);
	}
	return r;
}

// ---------------------------------------------------------------------------
// deflate
// ---------------------------------------------------------------------------
struct Deflate {
	uint64_t magic = MAGIC_DEF;
	z_streamp z = nullptr;
	int wrap = HDR_ZLIB, level = 6, strategy = Z_DEFAULT_STRATEGY;
	int init_wbits = 15, init_level = -1, init_memlevel = 8;   // as the caller gave them (AUTO mode may reopen the stream in software zlib)
	uint32_t max_history = 0;
	enum St { INIT, BUSY, BFINAL, TRAILER } st = INIT;
	std::vector<uint8_t> pend; size_t pend_off = 0;     // complete bytes waiting for next_out
	uint32_t tail_bits = 0; int tail_n = 0;             // bits after the last complete byte
	std::vector<uint8_t> fifo; size_t hist_len = 0, used = 0;   // [history][cached input]
	gz_headerp gzhead = nullptr;
	uint32_t dict_id = 0; size_t dict_len = 0;
	uint32_t crc = 0, adler = 1; bool cksum_set = false;
	nxz_dht_state *dht = nullptr; uint32_t counts[316]; bool have_counts = false; long last_job_bytes = 0;
	Engine eng; JobBuf jb; std::vector<uint8_t> jobout;
	uint64_t total_in = 0;
	// deflate_batch: its output when the caller's buffer may be too small for the whole run
	std::vector<uint8_t> batch_tmp;

	void out_bytes(const uint8_t *p, size_t n)
	{
		if (pend_off == pend.size()) {
			size_t k = std::min<size_t>(n, z->avail_out);
			memcpy(z->next_out, p, k);
			z->next_out += k; z->avail_out -= (uInt)k; z->total_out += k;
			p += k; n -= k;
			if (n) { pend.clear(); pend_off = 0; }
		}
		pend.insert(pend.end(), p, p + n);
	}
	void drain()
	{
		size_t k = std::min<size_t>(pend.size() - pend_off, z->avail_out);
		if (k) {
			memcpy(z->next_out, pend.data() + pend_off, k);
			z->next_out += k; z->avail_out -= (uInt)k; z->total_out += k; pend_off += k;
		}
		if (pend_off == pend.size()) { pend.clear(); pend_off = 0; }
	}
	bool pending() const { return pend_off < pend.size(); }
	void put_bits(uint32_t v, int n)
	{
		uint64_t acc = tail_bits | ((uint64_t)v << tail_n);
		int have = tail_n + n;
		uint8_t tmp[8]; int k = 0;
		for (; have >= 8; have -= 8, acc >>= 8) tmp[k++] = (uint8_t)acc;
		out_bytes(tmp, k);
		tail_bits = (uint32_t)acc; tail_n = have;
	}
	// empty (or len-byte) stored block header after the current tail bits: BFINAL, 00, pad, LEN, NLEN
	// (append_btype00_header / append_sync_flush, lib/nx_deflate.c:175-243)
	void stored_header(int final, uint32_t len)
	{
		put_bits((uint32_t)final & 1, 3);
		if (tail_n) put_bits(0, 8 - tail_n);
		uint8_t h[4] = { (uint8_t)len, (uint8_t)(len >> 8), (uint8_t)~len, (uint8_t)(~len >> 8) };
		out_bytes(h, 4);
	}
};

Deflate *dstate(z_streamp strm)
{
	if (!strm || !strm->state) return nullptr;
	Deflate *s = (Deflate *)strm->state;
	return s->magic == MAGIC_DEF ? s : nullptr;
}

void deflate_header(Deflate *s)
{
	if (s->wrap == HDR_ZLIB) {
		// lib/nx_deflate.c:1428-1459
		uint32_t header = (Z_DEFLATED + ((15 - 8) << 4)) << 8;
		uint32_t lf = s->level < 2 ? 0 : s->level < 6 ? 1 : s->level == 6 ? 2 : 3;
		header |= lf << 6;
		if (s->dict_len) header |= 0x20;
		header += 31 - (header % 31);
		uint8_t h[6] = { (uint8_t)(header >> 8), (uint8_t)header, (uint8_t)(s->dict_id >> 24), (uint8_t)(s->dict_id >> 16),
				 (uint8_t)(s->dict_id >> 8), (uint8_t)s->dict_id };
		s->out_bytes(h, s->dict_len ? 6 : 2);
		s->z->adler = s->adler = 1;
	} else if (s->wrap == HDR_GZIP) {
		gz_headerp g = s->gzhead;
		if (!g) {
			static const uint8_t blank[10] = { 0x1f, 0x8b, 0x08, 0, 0, 0, 0, 0, 0x04, 0x03 };   // :473-488
			s->out_bytes(blank, 10);
		} else {
			// lib/nx_deflate.c:1476-1549: text/extra/name/comment, no header crc, XFL 4
			uint8_t flg = (g->text ? 1 : 0) | (g->extra ? 4 : 0) | (g->name ? 8 : 0) | (g->comment ? 16 : 0);
			uint8_t h[10] = { 0x1f, 0x8b, 0x08, flg, (uint8_t)g->time, (uint8_t)(g->time >> 8), (uint8_t)(g->time >> 16),
					  (uint8_t)(g->time >> 24), 0x04, (uint8_t)g->os };
			s->out_bytes(h, 10);
			if (g->extra) {
				uint8_t l[2] = { (uint8_t)g->extra_len, (uint8_t)(g->extra_len >> 8) };
				s->out_bytes(l, 2);
				s->out_bytes(g->extra, g->extra_len);
			}
			if (g->name) s->out_bytes(g->name, strlen((const char *)g->name) + 1);
			if (g->comment) s->out_bytes(g->comment, strlen((const char *)g->comment) + 1);
		}
		s->z->adler = s->crc = 0;
	}
	s->st = Deflate::BUSY;
}

void deflate_trailer(Deflate *s)
{
	if (s->wrap == HDR_GZIP) {
		uint32_t isize = (uint32_t)s->total_in;
		uint8_t t[8] = { (uint8_t)s->crc, (uint8_t)(s->crc >> 8), (uint8_t)(s->crc >> 16), (uint8_t)(s->crc >> 24),
				 (uint8_t)isize, (uint8_t)(isize >> 8), (uint8_t)(isize >> 16), (uint8_t)(isize >> 24) };
		s->out_bytes(t, 8);
	} else if (s->wrap == HDR_ZLIB) {
		uint8_t t[4] = { (uint8_t)(s->adler >> 24), (uint8_t)(s->adler >> 16), (uint8_t)(s->adler >> 8), (uint8_t)s->adler };
		s->out_bytes(t, 4);
	}
	s->st = Deflate::TRAILER;
}

void publish_cksum(Deflate *s)
{
	if (s->wrap == HDR_ZLIB) s->z->adler = s->adler;
	else if (s->wrap == HDR_GZIP) s->z->adler = s->crc;
}

// consume n source bytes (first from the cache, then from next_in) and keep max_history of them
void deflate_consume(Deflate *s, size_t n)
{
	size_t from_cache = std::min(n, s->used), from_next = n - from_cache;
	// history := last max_history bytes of [history | consumed cache | consumed next_in]
	if (s->max_history == 0) {
		s->fifo.erase(s->fifo.begin(), s->fifo.begin() + s->hist_len + from_cache);
		s->hist_len = 0;
	} else if (from_next >= s->max_history) {
		// (a large call: the new history is the tail of what was taken from next_in, nothing else moves)
		s->fifo.erase(s->fifo.begin(), s->fifo.begin() + s->hist_len + from_cache);
		s->fifo.insert(s->fifo.begin(), s->z->next_in + from_next - s->max_history, s->z->next_in + from_next);
		s->hist_len = s->max_history;
	} else {
		// bring the consumed next_in bytes into the fifo right after the consumed cache part
		if (from_next) s->fifo.insert(s->fifo.begin() + s->hist_len + from_cache, s->z->next_in, s->z->next_in + from_next);
		size_t newhist = s->hist_len + from_cache + from_next;
		size_t drop = newhist > s->max_history ? newhist - s->max_history : 0;
		s->fifo.erase(s->fifo.begin(), s->fifo.begin() + drop);
		s->hist_len = newhist - drop;
	}
	s->used -= from_cache;
	s->z->next_in += from_next; s->z->avail_in -= (uInt)from_next; s->z->total_in += from_next;
	s->total_in += n;
}

// the next job's source: [history (multiple of 16)] [cache] [next_in part]; returns source bytes
uint32_t deflate_source(Deflate *s, nxz_crb_cpb_t *j, uint32_t limit, uint32_t *histuse_out)
{
	uint32_t histuse = (uint32_t)std::min<size_t>(s->hist_len, WINDOW) & ~15u;
	uint32_t room = JOB_UNIT - histuse;
	if (limit && limit < room) room = limit;
	uint32_t from_cache = (uint32_t)std::min<size_t>(s->used, room);
	uint32_t from_next = (uint32_t)std::min<size_t>(s->z->avail_in, room - from_cache);
	nxz_dde_t *l = s->jb.ddl;
	uint32_t cnt = 0, total = 0;
	if (histuse + from_cache) {
		nxz_dde_set_direct(&l[cnt++], s->fifo.data() + (s->hist_len - histuse), histuse + from_cache);
		total += histuse + from_cache;
	}
	if (from_next) { nxz_dde_set_direct(&l[cnt++], s->z->next_in, from_next); total += from_next; }
	if (cnt == 1) j->crb.source = l[0];
	else nxz_dde_set_indirect(&j->crb.source, l, cnt, total);
	if (cnt == 0) nxz_dde_set_direct(&j->crb.source, s->fifo.data(), 0);
	*histuse_out = histuse;
	return from_cache + from_next;
}

void combine_cksum(Deflate *s, uint32_t jcrc, uint32_t jadler, uint32_t n)
{
	// WRAP jobs start from the initial values: combine (lib/nx_deflate.c:1562-1578)
	if (s->cksum_set) {
		s->adler = (uint32_t)nx_adler32_combine(s->adler, jadler, n);
		s->crc = (uint32_t)nx_crc32_combine(s->crc, jcrc, n);
	} else { s->adler = jadler; s->crc = jcrc; }
	s->cksum_set = true;
}

// stored blocks for `n` source bytes (engine WRAP jobs copy and checksum them)
int deflate_stored(Deflate *s, uint32_t n, bool finish)
{
	while (n) {
		uint32_t chunk = std::min(n, STORED_MAX);
		nxz_crb_cpb_t *j = s->jb.job;
		memset(j, 0, sizeof(*j));
		// stored blocks carry no history: gather only cache + next_in
		uint32_t from_cache = (uint32_t)std::min<size_t>(s->used, chunk), from_next = chunk - from_cache;
		nxz_dde_t *l = s->jb.ddl; uint32_t cnt = 0;
		if (from_cache) nxz_dde_set_direct(&l[cnt++], s->fifo.data() + s->hist_len, from_cache);
		if (from_next) nxz_dde_set_direct(&l[cnt++], s->z->next_in, from_next);
		if (cnt == 1) j->crb.source = l[0]; else nxz_dde_set_indirect(&j->crb.source, l, cnt, chunk);
		s->jobout.resize(chunk + 64);
		nxz_dde_set_direct(&j->crb.target, s->jobout.data(), chunk);
		nxz_set_fc(j, NXZ_FC_WRAP);
		if (s->eng.submit(j) != NXZ_CC_OK) return Z_STREAM_ERROR;
		bool last = finish && chunk == n && s->used + s->z->avail_in == chunk;
		s->stored_header(last, chunk);
		s->out_bytes(s->jobout.data(), chunk);
		combine_cksum(s, nxz_out_crc(&j->cpb), nxz_out_adler(&j->cpb), chunk);
		deflate_consume(s, chunk);
		publish_cksum(s);
		if (last) s->st = Deflate::BFINAL;
		n -= chunk;
	}
	return Z_OK;
}

// one engine job = one deflate block (nx_compress_block, lib/nx_deflate.c:1209-1412)
// Dynamic blocks: the reference makes each job's table on the host from the symbol counts of the
// job before (nx_deflate.c:1030-1052, nx_dht.c), because its engine cannot; this engine can make the
// exact table of the job itself (NXZ_FC_COMPRESS_*DHTGEN), which is both smaller output and one
// table parse less per job.  NXZ_DEVICE_DHT=0 keeps the reference's scheme (cached / canned tables,
// nxz_dht.cpp); the CPU model of the test suite always does.
extern "C" int nxz_deflate_host(nxz_ctx_t *, int, const uint8_t *, size_t, int, uint8_t *, size_t, size_t *, uint32_t *, uint32_t *) __attribute__((weak));
bool device_dht_available() { return nxz_deflate_host != nullptr; }     // (the device engine, not the CPU model of the tests)
bool device_dht()
{
	static const bool on = nxz_deflate_host != nullptr && !(getenv("NXZ_DEVICE_DHT") && atoi(getenv("NXZ_DEVICE_DHT")) == 0);
	return on;
}

int deflate_job(Deflate *s, int flush)
{
	nxz_crb_cpb_t *j = s->jb.job;
	memset(j, 0, sizeof(*j));
	if (s->tail_n) s->stored_header(0, 0);            // the engine starts blocks on a byte boundary
	uint32_t histuse, n = deflate_source(s, j, 0, &histuse);
	if (n == 0) return Z_OK;
	const bool dynamic = s->strategy != Z_FIXED, host_table = dynamic && !device_dht();
	uint32_t fc = host_table ? NXZ_FC_COMPRESS_RESUME_DHT_COUNT : dynamic ? NXZ_FC_COMPRESS_RESUME_DHTGEN : NXZ_FC_COMPRESS_RESUME_FHT;
	nxz_set_fc(j, fc);
	nxz_set_in_histlen(&j->cpb, histuse / 16);
	nxz_set_in_crc(&j->cpb, s->crc);
	nxz_set_in_adler(&j->cpb, s->adler);
	if (host_table) {
		uint32_t dhtlen;
		nxz_dht_lookup(s->dht, s->have_counts ? s->counts : nullptr, s->last_job_bytes, j->cpb.in_dht, &dhtlen);
		nxz_set_in_dhtlen(&j->cpb, dhtlen);
	}
	s->jobout.resize((size_t)n * 2 + 1024);
	nxz_dde_set_direct(&j->crb.target, s->jobout.data(), (uint32_t)s->jobout.size());
	int cc = s->eng.submit(j);
	uint32_t spbc = host_table ? nxz_rd32(&j->cpb.out_spbc_with_count_be) : nxz_rd32(&j->cpb.u.out_spbc_be);
	uint32_t tpbc = nxz_csb_tpbc(j);
	if (cc == NXZ_CC_TPBC_GT_SPBC || ((cc == NXZ_CC_OK || cc == NXZ_CC_DATA_LENGTH) && tpbc + histuse > spbc)) {
		// did not shrink: re-emit the same source as stored blocks (:1274-1282, :1377-1389)
		return deflate_stored(s, spbc > histuse ? spbc - histuse : n, flush == Z_FINISH);
	}
	if (cc != NXZ_CC_OK && !(cc == NXZ_CC_DATA_LENGTH && (nxz_csb_ce3(j) & NXZ_CE_PARTIAL) && !(nxz_csb_ce3(j) & NXZ_CE_TERMINATE)))
		return Z_STREAM_ERROR;
	if (spbc <= histuse) return Z_OK;                 // no progress (:981-986)
	uint32_t took = spbc - histuse, tebc = nxz_out_tebc(&j->cpb);
	s->crc = nxz_out_crc(&j->cpb); s->adler = nxz_out_adler(&j->cpb); s->cksum_set = true;
	if (host_table) {
		for (int i = 0; i < 316; i++) s->counts[i] = nxz_rd32(&j->cpb.u.out_lzcount_be[i]);
		s->have_counts = true; s->last_job_bytes = took;
	}
	deflate_consume(s, took);
	publish_cksum(s);
	const bool final = flush == Z_FINISH && s->used == 0 && s->z->avail_in == 0;
	uint8_t *o = s->jobout.data();
	o[0] = (uint8_t)((o[0] & ~1) | (final ? 1 : 0));     // set_bfinal, :158-168
	if (final) {
		s->out_bytes(o, tpbc);                        // the partial last byte goes out zero padded
		s->st = Deflate::BFINAL;
		return Z_OK;
	}
	if (tebc) { s->out_bytes(o, tpbc - 1); s->tail_bits = o[tpbc - 1] & ((1u << tebc) - 1); s->tail_n = (int)tebc; }
	else s->out_bytes(o, tpbc);
	// flush block rules, lib/nx_deflate.c:1081-1176
	if (flush == Z_PARTIAL_FLUSH) {
		s->stored_header(0, 0);
		s->put_bits(2, 10);                             // empty fixed block: BFINAL 0, BTYPE 01, EOB
	} else if (s->tail_n || flush == Z_SYNC_FLUSH || flush == Z_FULL_FLUSH) {
		s->stored_header(0, 0);
	}
	if (flush == Z_FULL_FLUSH) { s->fifo.erase(s->fifo.begin(), s->fifo.begin() + s->hist_len); s->hist_len = 0; }
	return Z_OK;
}

// A caller that hands over hundreds of kilobytes at once at a level that keeps no history between
// jobs (the default and 1..4: lib/nx_deflate.c:654-680 sets max_history_len = 0, so every job is
// independent) gets all its full 64 KiB blocks -- with Z_FINISH everything it has -- compressed by ONE
// engine call (nxz_deflate_host: blocks side by side, fixed code or an exact dynamic table per block,
// strung together on the device the way deflate_job strings jobs together) instead of one nxu_run_job
// round trip per block.  (Weak references: the CPU model of the test suite has no such entry.)
extern "C" {
size_t nxz_deflate_host_bound_hist(size_t, uint32_t) __attribute__((weak));
int nxz_deflate_host_hist(nxz_ctx_t *, int, const uint8_t *, size_t, int, uint32_t, const uint8_t *, size_t, uint8_t *, size_t, size_t *, uint32_t *, uint32_t *) __attribute__((weak));
}
static size_t batch_min()
{
	static const size_t v = getenv("NXZ_BATCH_MIN") ? (size_t)strtoull(getenv("NXZ_BATCH_MIN"), nullptr, 0) : 0;
	return v ? v : 2 * JOB_UNIT;
}

// true when it took input; what it leaves (less than a block) goes the job-after-job way
bool deflate_batch(Deflate *s, int flush)
{
	z_streamp z = s->z;
	if (!nxz_deflate_host_hist || !nxz_deflate_host_bound_hist) return false;
	if (!s->eng.open || s->used != 0 || s->dict_len != 0 || s->pending() || z->avail_in < batch_min()) return false;
	// At the levels that carry history (5..9, max_history 4..32 KiB) a block's window is the input in front of
	// it: blocks of 64 KiB - max_history, still side by side (nxz_deflate_host_hist); the first block's window
	// is what the stream kept of the calls before (the front of the fifo).
	const uint32_t hmax = s->max_history > 32768 ? 32768u : (s->max_history & ~15u);
	const size_t unit = JOB_UNIT - hmax;
	nxz_ctx_t *ctx = (nxz_ctx_t *)s->eng.dev.paste_addr;
	if (!ctx) return false;
	const bool final = flush == Z_FINISH;
	const size_t take = final ? z->avail_in : (size_t)z->avail_in / unit * unit;
	const size_t bound = nxz_deflate_host_bound_hist(take, hmax);
	if (s->tail_n) s->stored_header(0, 0);                          // the run starts on a byte boundary
	// straight into the caller's buffer when it is sure to fit, else through the pending buffer
	const bool direct = !s->pending() && z->avail_out >= bound;
	uint8_t *out = z->next_out;
	if (!direct) { s->batch_tmp.resize(bound); out = s->batch_tmp.data(); }
	size_t produced = 0;
	uint32_t crc = 0, adler = 1;
	const uint64_t t0 = nxz_ticks();
	if (nxz_deflate_host_hist(ctx, s->strategy == Z_FIXED ? NXZ_FC_COMPRESS_FHT : 0x22 /* NXZ_FC_COMPRESS_DHTGEN */, z->next_in, take, final,
				  hmax, s->hist_len ? s->fifo.data() : nullptr, s->hist_len, out, bound, &produced, &crc, &adler))
		return false;                                               // nothing consumed: the ordinary path takes over
	nxz_device_stats(t0, t0 + (nxz_ticks() - t0) / ((take + unit - 1) / unit));   // per job, as AUTO mode's average counts
	if (direct) { z->next_out += produced; z->avail_out -= (uInt)produced; z->total_out += produced; }
	else s->out_bytes(out, produced);
	deflate_consume(s, take);
	combine_cksum(s, crc, adler, (uint32_t)take);
	publish_cksum(s);
	if (final) s->st = Deflate::BFINAL;
	s->have_counts = false;                                          // (the next single job starts from the default table again)
	// A flush request is honoured by the job that takes the last input (deflate_job).  When the batch left none,
	// the flush rules of lib/nx_deflate.c:1081-1176 are applied here: the run ends on a byte boundary (a block
	// that ended inside a byte is followed by an empty stored block already: pack_stream_kernel); a sync / full
	// flush gets its 00 00 FF FF marker here, ALWAYS -- round 3 looked at the run's last four bytes and left the
	// marker out when they read 00 00 FF FF, which a stored or byte-aligned block's own data may end in (advisor
	// finding: a client that strips the marker, permessage-deflate for one, would then cut real data; a second
	// empty stored block behind one that the run brought along costs five bytes and is valid) --, a partial
	// flush its empty fixed block.
	if (!final && z->avail_in == 0 && s->used == 0 && (flush == Z_SYNC_FLUSH || flush == Z_FULL_FLUSH || flush == Z_PARTIAL_FLUSH)) {
		s->stored_header(0, 0);
		if (flush == Z_PARTIAL_FLUSH) s->put_bits(2, 10);            // empty fixed block: BFINAL 0, BTYPE 01, EOB
		if (flush == Z_FULL_FLUSH) { s->fifo.erase(s->fifo.begin(), s->fifo.begin() + s->hist_len); s->hist_len = 0; }
	}
	return true;
}

int deflate_end_of_stream(Deflate *s)
{
	if (s->st == Deflate::BUSY) {                         // :1594-1604: final empty stored block
		s->stored_header(1, 0);
		s->st = Deflate::BFINAL;
	}
	if (s->st == Deflate::BFINAL) deflate_trailer(s);
	return s->pending() ? Z_OK : Z_STREAM_END;
}

int deflate_reset_keep(z_streamp strm)
{
	Deflate *s = dstate(strm);
	if (!s) return Z_STREAM_ERROR;
	strm->total_in = strm->total_out = 0; strm->msg = Z_NULL; strm->data_type = Z_UNKNOWN;
	s->st = Deflate::INIT; s->pend.clear(); s->pend_off = 0; s->tail_bits = 0; s->tail_n = 0;
	s->fifo.clear(); s->hist_len = s->used = 0; s->dict_len = 0; s->dict_id = 0;
	s->crc = 0; s->adler = 1; s->cksum_set = false; s->have_counts = false; s->last_job_bytes = 0; s->total_in = 0;
	if (s->wrap == HDR_ZLIB) strm->adler = 1; else if (s->wrap == HDR_GZIP) strm->adler = 0;
	if (s->strategy != Z_FIXED) { if (s->dht) nxz_dht_end(s->dht); s->dht = nxz_dht_begin(); }
	return Z_OK;
}

} // namespace

extern "C" int nx_deflateInit_(z_streamp strm, int level, const char *version, int stream_size)
{
	return nx_deflateInit2_(strm, level, Z_DEFLATED, 15, 8, Z_DEFAULT_STRATEGY, version, stream_size);
}

extern "C" int nx_deflateInit2_(z_streamp strm, int level, int method, int windowBits, int memLevel,
				int strategy, const char *version, int stream_size)
{
	(void)memLevel; (void)version; (void)stream_size;
	if (strm == Z_NULL) return Z_STREAM_ERROR;
	strm->msg = Z_NULL; strm->total_in = strm->total_out = 0;
	if (windowBits != 15 && windowBits != 31 && windowBits != -15) return Z_STREAM_ERROR;     // :609-613
	if (method != Z_DEFLATED || (strategy != Z_FIXED && strategy != Z_DEFAULT_STRATEGY)) return Z_STREAM_ERROR;
	uint32_t maxhist;
	const int level_in = level;
	switch (level) {                                                                           // :654-680
	case 0: level = 6; maxhist = 0; break;                                                    // (Q2)
	case Z_DEFAULT_COMPRESSION: case 1: case 2: case 3: case 4: maxhist = 0; break;
	case 5: case 6: case 7: maxhist = 1u << (level + 7); break;
	case 8: case 9: maxhist = WINDOW; break;
	default: return Z_STREAM_ERROR;
	}
	Deflate *s = new (std::nothrow) Deflate();
	if (!s || !s->jb.job || !s->jb.ddl) { delete s; return Z_MEM_ERROR; }
	if (!s->eng.begin()) { delete s; return Z_STREAM_ERROR; }                                  // "cannot open NX device"
	s->z = strm;
	s->wrap = windowBits < 0 ? HDR_RAW : windowBits > 15 ? HDR_GZIP : HDR_ZLIB;
	s->init_wbits = windowBits; s->init_level = level_in; s->init_memlevel = memLevel;
	s->level = level; s->max_history = maxhist;
	// NX_GZIP_STRATEGY=0 forces fixed Huffman whatever the caller asked for (lib/nx_deflate.c:648-652)
	s->strategy = (strategy == Z_FIXED || nxz_config()->strategy_override == 0) ? Z_FIXED : Z_DEFAULT_STRATEGY;
	strm->state = (struct internal_state *)s;
	return deflate_reset_keep(strm);
}

extern "C" int nx_deflateResetKeep(z_streamp strm) { return deflate_reset_keep(strm); }
extern "C" int nx_deflateReset(z_streamp strm) { return deflate_reset_keep(strm); }

extern "C" int nx_deflateEnd(z_streamp strm)
{
	Deflate *s = dstate(strm);
	if (!s) return Z_STREAM_ERROR;
	int busy = s->st == Deflate::BUSY;
	if (s->dht) nxz_dht_end(s->dht);
	s->eng.end();
	s->magic = 0;
	delete s;
	strm->state = Z_NULL;
	return busy ? Z_DATA_ERROR : Z_OK;                                                        // :576-577
}

extern "C" unsigned long nx_deflateBound(z_streamp strm, unsigned long sourceLen)
{
	(void)strm;
	nxz_stats_inc("deflateBound");                                                            // :1920
	long pg = sysconf(_SC_PAGESIZE);
	return sourceLen * 2 + (unsigned long)std::min<long>(pg, 1 << 16);                         // :1922 (Q1)
}

extern "C" int nx_deflateSetHeader(z_streamp strm, gz_headerp head)
{
	Deflate *s = dstate(strm);
	if (!s || s->wrap != HDR_GZIP) return Z_STREAM_ERROR;
	s->gzhead = head;
	return Z_OK;
}

extern "C" int nx_deflateSetDictionary(z_streamp strm, const unsigned char *dictionary, unsigned int dictLength)
{
	Deflate *s = dstate(strm);
	if (!s || !dictionary) return Z_STREAM_ERROR;
	if (s->wrap == HDR_GZIP || (s->wrap == HDR_ZLIB && s->st != Deflate::INIT) || s->used) return Z_STREAM_ERROR;
	uint32_t keep = std::min<uint32_t>(dictLength, WINDOW);
	s->fifo.assign(dictionary + dictLength - keep, dictionary + dictLength);
	s->hist_len = keep;
	s->dict_len = dictLength;
	s->dict_id = sw_adler32(1, dictionary, dictLength);
	if (s->wrap == HDR_ZLIB) strm->adler = s->dict_id;
	// the dictionary must stay reachable for the first job even at the levels that keep no history
	if (s->max_history < keep) s->max_history = keep;
	return Z_OK;
}

extern "C" int nx_deflateCopy(z_streamp dest, z_streamp source)
{
	Deflate *s = dstate(source);
	if (!s || !dest) return Z_STREAM_ERROR;
	*dest = *source;
	Deflate *d = new (std::nothrow) Deflate();
	if (!d || !d->jb.job || !d->eng.begin()) { delete d; return Z_MEM_ERROR; }
	d->z = dest; d->wrap = s->wrap; d->level = s->level; d->strategy = s->strategy; d->max_history = s->max_history;
	d->init_wbits = s->init_wbits; d->init_level = s->init_level; d->init_memlevel = s->init_memlevel;
	d->st = s->st; d->pend = s->pend; d->pend_off = s->pend_off; d->tail_bits = s->tail_bits; d->tail_n = s->tail_n;
	d->fifo = s->fifo; d->hist_len = s->hist_len; d->used = s->used; d->gzhead = s->gzhead;
	d->dict_id = s->dict_id; d->dict_len = s->dict_len; d->crc = s->crc; d->adler = s->adler; d->cksum_set = s->cksum_set;
	d->dht = s->dht ? nxz_dht_copy(s->dht) : nullptr; memcpy(d->counts, s->counts, sizeof(d->counts));
	d->have_counts = s->have_counts; d->last_job_bytes = s->last_job_bytes; d->total_in = s->total_in;
	dest->state = (struct internal_state *)d;
	return Z_OK;
}

extern "C" int nx_deflate(z_streamp strm, int flush)
{
	if (flush > Z_BLOCK || flush < 0) return Z_STREAM_ERROR;
	Deflate *s = dstate(strm);
	if (!s) return Z_STREAM_ERROR;
	if (strm->next_out == Z_NULL || (strm->avail_in != 0 && strm->next_in == Z_NULL)) return Z_STREAM_ERROR;
	if (strm->avail_out == 0) return Z_BUF_ERROR;
	if (s->st == Deflate::INIT) deflate_header(s);
	if (s->st >= Deflate::BFINAL && flush != Z_FINISH) return Z_STREAM_ERROR;
	if (s->st >= Deflate::BFINAL && strm->avail_in != 0) return Z_BUF_ERROR;

	const bool had_pending = s->pending();
	s->drain();
	if (s->pending()) return Z_OK;                          // need more output space
	if (!had_pending && strm->avail_in == 0 && (flush < Z_PARTIAL_FLUSH || flush > Z_FINISH))
		return Z_BUF_ERROR;                                 // nothing to do (:1721-1722)

	for (int guard = 0; guard < 0xffff; guard++) {
		if (s->st >= Deflate::BFINAL) return deflate_end_of_stream(s);
		const size_t avail = s->used + strm->avail_in;
		if (avail <= nxz_config()->cache_threshold && flush == Z_NO_FLUSH && s->dict_len == 0 && !s->pending()) {
			// small input, no flush requested: just remember it (:1729-1741, cache_input :783)
			s->fifo.insert(s->fifo.end(), strm->next_in, strm->next_in + strm->avail_in);
			s->used += strm->avail_in;
			strm->total_in += strm->avail_in; strm->next_in += strm->avail_in; strm->avail_in = 0;
			return Z_OK;
		}
		if (avail) {
			if (!deflate_batch(s, flush)) {
				int rc = deflate_job(s, flush);
				if (rc != Z_OK) return rc;
			}
			s->dict_len = 0;
		}
		const bool more_in = s->used + strm->avail_in != 0;
		if (strm->avail_out == 0 || s->pending()) return Z_OK;     // buffer_state 0b0xxx / fifo_out not empty
		if (!more_in) {
			if (flush != Z_FINISH) return Z_OK;
			return deflate_end_of_stream(s);
		}
	}
	return Z_STREAM_ERROR;
}

// ---------------------------------------------------------------------------
// inflate
// ---------------------------------------------------------------------------
namespace {

// a byte vector whose resize() does not zero what it adds (the output of a part of a stream that waits for next_out is
// copied over it at once: 3.7 MiB of memset per 1 MiB step of inflate() otherwise)
template <class T> struct default_init_alloc : std::allocator<T> {
	template <class U> struct rebind { using other = default_init_alloc<U>; };
	template <class U, class... A> void construct(U *p, A &&... a)
	{
		if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
		else ::new ((void *)p) U(std::forward<A>(a)...);
	}
};
typedef std::vector<uint8_t, default_init_alloc<uint8_t>> raw_bytes;

struct Inflate {
	uint64_t magic = MAGIC_INF;
	z_streamp z = nullptr;
	int wrap = HDR_ZLIB | HDR_GZIP, window_bits = 15;
	int init_wbits = 47;                               // as the caller gave it
	enum St { HEADER, GZ_ID2, GZ_CM, GZ_FLG, GZ_MTIME, GZ_XFL, GZ_OS, GZ_XLEN, GZ_EXTRA, GZ_NAME, GZ_COMMENT, GZ_HCRC,
		  ZL_CMF, ZL_FLG, ZL_DICTID, NEED_DICT, BODY, TRAILER, DONE, BAD } st = HEADER;
	uint32_t held = 0, nheld = 0, gzflags = 0, xlen = 0, zcmf = 0, dictid = 0;
	gz_headerp gzhead = nullptr;
	uint32_t hcrc = 0;                                 // running crc of the gzip header bytes
	raw_bytes pend; size_t pend_off = 0;               // produced bytes waiting for next_out (resize() leaves new bytes as they are: megabytes per call)
	std::vector<uint8_t> hist;                         // last <= 32 KiB of output
	std::vector<uint8_t> carry;                        // source bytes taken from next_in but not yet consumed by the engine
	std::vector<uint8_t> unget;                        // bytes taken from next_in that lie BEHIND the stream's trailer (the next member's):
							   // read first by whatever follows, and kept across inflateReset
	uint32_t sfbt = 0, subc = 0, rem = 0, dhtlen = 0; uint8_t dht[NXZ_DHT_MAXSZ]; bool resuming = false;
	uint32_t crc = 0, adler = 1;
	uint64_t total_out = 0;
	uint8_t trailer[8]; uint32_t ntrailer = 0;
	bool sync_point = false, have_dict = false;
	uint32_t par_skip = 0, par_declined = 0;           // calls for which the parallel decode is not tried again; declines in a row
	bool whole_at_once = false;                        // the call in work was told Z_FINISH before anything of the stream had been taken: a one-shot call
	bool one_shot_hint = false;                        // ... or comes from nx_uncompress2, which holds the whole source
	uint32_t ratio = 250;                              // last compressed/uncompressed per mille (:1234-1250)
	Engine eng; JobBuf jb; std::vector<uint8_t> src, out;

	bool pending() const { return pend_off < pend.size(); }
	void drain()
	{
		size_t k = std::min<size_t>(pend.size() - pend_off, z->avail_out);
		if (k) { memcpy(z->next_out, pend.data() + pend_off, k); z->next_out += k; z->avail_out -= (uInt)k; z->total_out += k; pend_off += k; }
		if (pend_off == pend.size()) { pend.clear(); pend_off = 0; }
	}
	void produce(const uint8_t *p, size_t n)
	{
		total_out += n;
		// history = last 32 KiB of everything produced
		if (n >= WINDOW) hist.assign(p + n - WINDOW, p + n);
		else {
			size_t drop = hist.size() + n > WINDOW ? hist.size() + n - WINDOW : 0;
			hist.erase(hist.begin(), hist.begin() + drop);
			hist.insert(hist.end(), p, p + n);
		}
		if (!pending()) {
			size_t k = std::min<size_t>(n, z->avail_out);
			memcpy(z->next_out, p, k); z->next_out += k; z->avail_out -= (uInt)k; z->total_out += k;
			p += k; n -= k;
		}
		pend.insert(pend.end(), p, p + n);
	}
};

Inflate *istate(z_streamp strm)
{
	if (!strm || !strm->state) return nullptr;
	Inflate *s = (Inflate *)strm->state;
	return s->magic == MAGIC_INF ? s : nullptr;
}

// next header / trailer byte: what an earlier call took beyond its stream's end first (already
// counted in total_in), then next_in.  Gzip header bytes run through the header CRC (FHCRC, RFC 1952).
bool get_byte(Inflate *s, uint32_t &c)
{
	if (!s->unget.empty()) { c = s->unget.front(); s->unget.erase(s->unget.begin()); }
	else {
		if (s->z->avail_in == 0) return false;
		c = *s->z->next_in++;
		s->z->avail_in--; s->z->total_in++;
	}
	if (s->st >= Inflate::GZ_ID2 && s->st <= Inflate::GZ_COMMENT) { const uint8_t b = (uint8_t)c; s->hcrc = sw_crc32(s->hcrc, &b, 1); }
	return true;
}

void publish(Inflate *s)
{
	if (s->wrap == HDR_GZIP) s->z->adler = s->crc;
	else if (s->wrap == HDR_ZLIB) s->z->adler = s->adler;
}

// A caller that hands over a good deal of deflate data at once (nx_uncompress, inflate() on a whole file
// or with buffers of a megabyte) gets the engine's parallel decode of ONE stream (nxz_inflate_stream_part:
// block starts are found by speculation, the pieces are inflated side by side) instead of the
// job-after-job loop below, which is what the reference runs (lib/nx_inflate.c:1143-1744) and is only
// as fast as one wavefront.  What the caller holds need not be the whole stream: the part may begin
// inside a block (the resume fields of whatever ran before) and ends where the source ends, with the
// same resume fields a suspended job reports; output beyond avail_out waits in `pend` like a job's.
// Everything the engine declines (short input, hardly any dynamic blocks) falls through to that loop.
// (Weak references: the CPU model of the test suite has no such entry points.)
extern "C" {
int nxz_inflate_stream_part(nxz_ctx_t *, const uint8_t *, uint64_t, uint64_t, const uint8_t *, uint32_t, uint8_t *, uint64_t,
			    uint64_t *, uint32_t *, uint32_t *, uint64_t *, nxz_stream_resume_t *, uint32_t *, void *) __attribute__((weak));
void *nxz_dev_malloc(nxz_ctx_t *, size_t) __attribute__((weak));
void nxz_dev_free(nxz_ctx_t *, void *) __attribute__((weak));
int nxz_copy_to_device(nxz_ctx_t *, void *, const void *, size_t, void *) __attribute__((weak));
int nxz_copy_to_host(nxz_ctx_t *, void *, const void *, size_t, void *) __attribute__((weak));
void *nxz_pinned_malloc(nxz_ctx_t *, size_t) __attribute__((weak));
void nxz_pinned_free(nxz_ctx_t *, void *) __attribute__((weak));
int nxz_ctx_sync(nxz_ctx_t *, void *) __attribute__((weak));
void *nxz_stream_create(nxz_ctx_t *) __attribute__((weak));
int nxz_ctx_device(nxz_ctx_t *) __attribute__((weak));
int nxz_engine_usable(void) __attribute__((weak));
}
// 12 KiB -- but 112 KiB (round 6; 64 before: one workgroup decodes a whole stream of up to 350 KiB of output faster than the
// one-stream pipeline starts up: a 256 KiB buffer 0.80 -> 0.71 ms) for a stream that comes in ONE call (nx_uncompress, inflate() with Z_FINISH on a fresh stream: the shape of
// samples/compdecomp_th.c): such a buffer is a job, and the rounds of nxu_run_job cut their jobs into pieces (nxz_inflate_cut.hip,
// one sequence of launches for all callers of a round) -- the better the more threads call at once: 64 threads x 64 KiB buffers 1.7
// against 0.40 GiB/s.  Not for a stream that comes in steps: what a step leaves over is often just short of the step, and as a
// job it is cut in its first blocks only (inflate() in 64 KiB steps: 0.27 against 0.23 GiB/s).  NXZ_PARALLEL_INFLATE_MIN: both, bytes.
static size_t parallel_inflate_min(bool whole_at_once)
{
	static const size_t v = getenv("NXZ_PARALLEL_INFLATE_MIN") ? (size_t)strtoull(getenv("NXZ_PARALLEL_INFLATE_MIN"), nullptr, 0) : 0;
	return v ? v : whole_at_once ? (size_t)112 << 10 : (size_t)12 << 10;
}
static std::atomic<int> g_inflate_callers{0};    // threads inside nx_inflate right now
constexpr size_t CARRY_KEEP = 1024;                // unconsumed source kept between calls at most: a dynamic block header (<= 290 bytes) and a token

bool parallel_inflate(Inflate *s)
{
	z_streamp z = s->z;
	const auto t_entry = std::chrono::steady_clock::now();
	if (!nxz_inflate_stream_part || !nxz_dev_malloc || !nxz_dev_free || !nxz_copy_to_device || !nxz_copy_to_host || !nxz_ctx_sync) return false;
	if (s->pending() || !s->eng.open) return false;
	if (nxz_engine_usable && !nxz_engine_usable()) return false;       // (forked after the engine was opened: the job loop reports it)
	static const bool off = getenv("NXZ_PARALLEL_INFLATE") && atoi(getenv("NXZ_PARALLEL_INFLATE")) == 0;   // 0: always job after job
	const size_t nc = s->carry.size();
	if (off || nc + z->avail_in < parallel_inflate_min(s->whole_at_once)) return false;
	// A whole stream of up to a few hundred KiB from a caller who is not alone: as ONE job -- the rounds of nxu_run_job take the
	// callers' jobs together, a stream per workgroup in one launch (nxz_inflate_wg.hip), where the parts of this function run their
	// sequences of small kernels one caller after the other: 16 threads x 256 KiB buffers 1.7 -> 2.9 GiB/s, x 512 KiB 2.5 -> 3.0.
	// (A caller alone is better off here from 256 KiB on: one workgroup makes 0.3-0.4 GiB/s of one stream.)
	if (s->whole_at_once && g_inflate_callers.load(std::memory_order_relaxed) > 1 && nc + z->avail_in <= (size_t)384 << 10 &&
	    (size_t)z->avail_out + WINDOW + (WINDOW >> 2) <= ((size_t)1 << 20)) return false;
	if (s->par_skip) { s->par_skip--; return false; }           // (declined a moment ago: this stream is not the kind)
	nxz_ctx_t *ctx = (nxz_ctx_t *)s->eng.dev.paste_addr;
	if (!ctx) return false;
	// all of next_in when the target can take what it makes; otherwise about a megabyte at a time (the surplus
	// output waits in `pend`: room for 32 x the part's size, so that -- whatever next_in moves over being what the
	// engine has used -- a part is as good as always used up and the caller's next call brings a whole new part
	// instead of the shrinking remainder of this one)
	size_t take = z->avail_in;
	if (z->avail_out < 2 * (nc + take)) take = std::min<size_t>(take, std::max<size_t>(1u << 20, z->avail_out));
	const size_t nin = nc + take, nh = s->hist.size();
	const size_t cap = z->avail_out >= 2 * nin ? (size_t)z->avail_out : std::max<size_t>(z->avail_out, 32 * nin);
	// device buffers for the stream and its output and a HIP stream to work on: a few sets, kept from call to call
	// (grow only); callers on different threads take different sets and run side by side
	// The sets belong to a DEVICE, not to a context: what nxz_dev_malloc hands out is plain device memory that outlives
	// the context it was asked through, and since nxz_ctx_create(-1) gives every thread a GPU of its own in turn
	// (nxz_pick_device) callers on different devices must never take over each other's set -- round 3 kept ONE pool for
	// all of them and dropped a set's buffers and stream, unfreed, whenever the caller's context was another one
	// (advisor finding: device memory leaked without bound on a multi-GPU host).
	struct Slot {
		std::mutex mtx;
		uint8_t *src = nullptr, *dst = nullptr, *hist = nullptr;
		size_t src_cap = 0, dst_cap = 0;
		void *stream = nullptr;
		// pinned staging for parts of a few MiB: a copy straight from (to) the caller's pages makes the runtime pin and
		// unpin them per call under the process's memory-map lock, which is what sixteen threads of 1 MiB calls spent
		// their time in; the calling thread's own memcpy to and from pinned memory costs a tenth of that
		uint8_t *pin_in = nullptr, *pin_out = nullptr;
		size_t pin_in_cap = 0, pin_out_cap = 0;
	};
	constexpr size_t STAGE_IN_MAX = (size_t)8 << 20, STAGE_OUT_MAX = (size_t)16 << 20;  // (twelve callers at a time, see below: 288 MiB of pinned memory if every one of them stages as much as it may)
	static const bool stage_on = !(getenv("NXZ_HOST_STAGE") && atoi(getenv("NXZ_HOST_STAGE")) == 0) && nxz_pinned_malloc && nxz_pinned_free;
	constexpr int NSLOT = 32, NDEV = 64;
	static Slot slots[NDEV][NSLOT];
	static std::atomic<unsigned> turn{0};
	int dev = nxz_ctx_device ? nxz_ctx_device(ctx) : 0;
	if (dev < 0 || dev >= NDEV) return false;
	// Twelve of these calls at a time (NXZ_PARALLEL_INFLATE_MAX; 0: as many as there are callers): the device runs the callers'
	// small kernels one after the other whatever streams they come on, and beyond a dozen callers the streams only get in each
	// other's way -- 64 threads of 1 MiB calls 3.4 -> 4.9 GiB/s, of 256 KiB calls 1.1 -> 1.7, of 4 MiB calls 7.8 -> 11.0; 16 threads as before.
	static const int gate_max = getenv("NXZ_PARALLEL_INFLATE_MAX") ? atoi(getenv("NXZ_PARALLEL_INFLATE_MAX")) : 12;
	static std::mutex gate_m; static std::condition_variable gate_cv; static int gate_n = 0;
	struct Gate { bool on; Gate(bool o) : on(o) { if (on) { std::unique_lock<std::mutex> g(gate_m); gate_cv.wait(g, [] { return gate_n < gate_max; }); gate_n++; } }
		      ~Gate() { if (on) { { std::lock_guard<std::mutex> g(gate_m); gate_n--; } gate_cv.notify_one(); } } } gate(gate_max > 0);
	// (how many callers are in here right now: the pinned staging below is for company -- a caller that is alone copies straight
	// from and to its own pages, which the runtime pins faster than one core copies them: one thread's 16 MiB nx_uncompress calls
	// 4.2 -> 5.5 GiB/s, 8 MiB 3.3 -> 3.7; below a few MiB the staging is as fast)
	static std::atomic<int> inside{0};
	struct Inside { int n; Inside() : n(++inside) {} ~Inside() { --inside; } } in_here;
	const bool company = in_here.n > 1;
	constexpr size_t STAGE_ALONE_MAX = (size_t)2 << 20;
	Slot *const pool = slots[dev];
	Slot *slot = nullptr;
	for (int k = 0; k < NSLOT && !slot; k++) if (pool[k].mtx.try_lock()) slot = &pool[k];
	if (!slot) { slot = &pool[turn.fetch_add(1) % NSLOT]; slot->mtx.lock(); }
	std::lock_guard<std::mutex> pool_guard(slot->mtx, std::adopt_lock);
	uint8_t *&pool_src = slot->src, *&pool_dst = slot->dst, *&pool_hist = slot->hist;
	size_t &pool_src_cap = slot->src_cap, &pool_dst_cap = slot->dst_cap;
	if (!slot->stream && nxz_stream_create) slot->stream = nxz_stream_create(ctx);
	void *const hs = slot->stream;               // (NULL, the default stream, if none could be made)
	// (an eighth more than asked for, in whole MiB: the streams of one caller differ by a few bytes, and a step up is a free and an allocation through the runtime)
	auto roomy = [](size_t n) { return (n + n / 8 + (((size_t)1 << 20) - 1)) & ~(((size_t)1 << 20) - 1); };
	if (pool_src_cap < nin + 64) { if (pool_src) nxz_dev_free(ctx, pool_src); pool_src = (uint8_t *)nxz_dev_malloc(ctx, roomy(nin + 64)); pool_src_cap = pool_src ? roomy(nin + 64) : 0; }
	if (pool_dst_cap < cap + 64) { if (pool_dst) nxz_dev_free(ctx, pool_dst); pool_dst = (uint8_t *)nxz_dev_malloc(ctx, roomy(cap + 64)); pool_dst_cap = pool_dst ? roomy(cap + 64) : 0; }
	if (!pool_hist) pool_hist = (uint8_t *)nxz_dev_malloc(ctx, WINDOW);
	uint8_t *d_src = pool_src, *d_dst = pool_dst, *d_hist = nh ? pool_hist : nullptr;
	bool ok = d_src && d_dst && (!nh || d_hist);
	uint64_t out_len = 0, end_bit = 0;
	uint32_t crc = 0, adler = 1;
	static const bool trace = getenv("NXZ_API_TRACE") != nullptr;
	auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = trace ? now() : 0;
	auto pin_need = [&](uint8_t *&p, size_t &have, size_t want, size_t most) {
		if (!stage_on || want > most || (!company && want > STAGE_ALONE_MAX)) return false;
		if (have >= want) return true;
		size_t to = std::max<size_t>((size_t)256 << 10, have);
		while (to < want) to <<= 1;
		if (p) nxz_pinned_free(ctx, p);
		p = (uint8_t *)nxz_pinned_malloc(ctx, to);
		have = p ? to : 0;
		return p != nullptr;
	};
	if (ok && pin_need(slot->pin_in, slot->pin_in_cap, nin + nh, STAGE_IN_MAX)) {
		uint8_t *const pi = slot->pin_in;
		if (nc) memcpy(pi, s->carry.data(), nc);
		if (take) memcpy(pi + nc, z->next_in, take);
		if (nh) memcpy(pi + nin, s->hist.data(), nh);
		ok = nxz_copy_to_device(ctx, d_src, pi, nin, hs) == 0 && (!nh || nxz_copy_to_device(ctx, d_hist, pi + nin, nh, hs) == 0);
	} else if (ok) ok = (!nc || nxz_copy_to_device(ctx, d_src, s->carry.data(), nc, hs) == 0) && (!take || nxz_copy_to_device(ctx, d_src + nc, z->next_in, take, hs) == 0) &&
		     (!nh || nxz_copy_to_device(ctx, d_hist, s->hist.data(), nh, hs) == 0);
	if (ok && trace) (void)nxz_ctx_sync(ctx, hs);
	const double t1 = trace ? now() : 0;
	// where the stream stands at the first byte: the fields of the last suspension (a job's or a part's)
	nxz_stream_resume_t st;
	memset(&st, 0, sizeof(st));
	uint64_t first_bit = 0;
	if (s->resuming) {
		st.sfbt = s->sfbt; st.rem = s->rem; st.dhtlen = s->dhtlen; memcpy(st.dht, s->dht, NXZ_DHT_MAXSZ);
		first_bit = s->subc ? 8 - s->subc : 0;
	}
	int prc = -1;
	if (ok) prc = nxz_inflate_stream_part(ctx, d_src, nin, first_bit, d_hist, (uint32_t)nh, d_dst, cap, &out_len, &crc, &adler, &end_bit, &st, nullptr, hs);
	ok = ok && prc == 0 && end_bit >= first_bit && end_bit <= (uint64_t)nin * 8 && (out_len || (end_bit >> 3));
	const double t2 = trace ? now() : 0;
	// the output: as much as the caller has room for goes straight to next_out, the rest waits; the last 32 KiB are the next history
	const size_t direct = (size_t)std::min<uint64_t>(out_len, z->avail_out), later = ok ? (size_t)out_len - direct : 0;
	std::vector<uint8_t> tail;
	if (ok) {
		s->pend.resize(later); s->pend_off = 0;
		tail.resize((size_t)std::min<uint64_t>(out_len, WINDOW));
		if (out_len && pin_need(slot->pin_out, slot->pin_out_cap, (size_t)out_len, STAGE_OUT_MAX)) {
			ok = nxz_copy_to_host(ctx, slot->pin_out, d_dst, (size_t)out_len, hs) == 0 && nxz_ctx_sync(ctx, hs) == 0;
			if (ok) {
				if (direct) memcpy(z->next_out, slot->pin_out, direct);
				if (later) memcpy(s->pend.data(), slot->pin_out + direct, later);
				if (!tail.empty()) memcpy(tail.data(), slot->pin_out + out_len - tail.size(), tail.size());
			}
		} else
		ok = (!direct || nxz_copy_to_host(ctx, z->next_out, d_dst, direct, hs) == 0) && (!later || nxz_copy_to_host(ctx, s->pend.data(), d_dst + direct, later, hs) == 0) &&
		     (tail.empty() || nxz_copy_to_host(ctx, tail.data(), d_dst + out_len - tail.size(), tail.size(), hs) == 0) && nxz_ctx_sync(ctx, hs) == 0;
		if (!ok) s->pend.clear();
	}
	if (trace) fprintf(stderr, "nxz parallel_inflate: %zu bytes in (%zu carried), %.2f ms for a turn and buffers, copy in %.2f ms, inflate %.2f ms, copy out %.2f ms (%llu bytes, %zu wait)%s\n", nin, nc,
			   t0 - std::chrono::duration<double, std::milli>(t_entry.time_since_epoch()).count(), t1 - t0, t2 - t1, now() - t2,
			   (unsigned long long)out_len, later, ok ? (st.final ? " -- final" : "") : " -- declined");
	if (!ok) {
		// declined: once is chance (the next part is tried again), again and again is the kind of stream
		s->par_declined = std::min<uint32_t>(s->par_declined + 1, 4);
		s->par_skip = (1u << (s->par_declined - 1)) - 1;
		return false;
	}
	s->par_declined = 0;
	// the source: [carry][next_in .. take); whole bytes used, and the byte the part stopped in (supplied again)
	auto at = [&](size_t i) -> uint8_t { return i < nc ? s->carry[i] : z->next_in[i - nc]; };
	const size_t consumed = st.final ? (size_t)((end_bit + 7) >> 3) : (size_t)(end_bit >> 3);
	bool sync_point = false;
	if (!st.final) {
		const uint64_t subc = (uint64_t)nin * 8 - end_bit;            // bits not processed, as a suspended job counts them
		if ((st.sfbt & 0xe) == 0xe && subc >= 3 && subc <= 10 && nin >= 2)      // right behind an empty stored block's header (lib/nx_inflate.c:1563-1583)
			sync_point = subc <= 8 ? !(at(nin - 1) & (uint8_t)(0xff << (8 - subc))) : (at(nin - 1) == 0 && !(at(nin - 2) & (uint8_t)(0xff << (16 - subc))));
	}
	std::vector<uint8_t> rest;
	rest.reserve(nin - consumed);
	for (size_t i = consumed; i < nc; i++) rest.push_back(s->carry[i]);
	const size_t from_next0 = consumed > nc ? consumed - nc : 0;
	rest.insert(rest.end(), z->next_in + from_next0, z->next_in + take);
	z->next_in += take; z->avail_in -= (uInt)take; z->total_in += take;
	if (st.final || rest.size() > CARRY_KEEP) {
		// what is not this stream's, or was left for want of room in the target, goes back to the caller's view as far
		// as it came from next_in: next_in moves only over what the engine has used (lib/nx_inflate.c:1614-1623), so at
		// Z_STREAM_END it stands right behind the stream.  Only the few bytes of a token or header that the source
		// ended in stay here (the caller's next call brings what follows them).
		const size_t giveback = std::min<size_t>(rest.size(), take - from_next0);
		z->next_in -= giveback; z->avail_in += (uInt)giveback; z->total_in -= giveback;
		rest.resize(rest.size() - giveback);
	}
	s->carry.swap(rest);
	z->next_out += direct; z->avail_out -= (uInt)direct; z->total_out += direct;
	s->crc = (uint32_t)nx_crc32_combine(s->crc, crc, (off_t)out_len);
	s->adler = (uint32_t)nx_adler32_combine(s->adler, adler, (off_t)out_len);
	s->total_out += out_len;
	if (tail.size() >= WINDOW) s->hist.swap(tail);
	else {
		const size_t drop = s->hist.size() + tail.size() > WINDOW ? s->hist.size() + tail.size() - WINDOW : 0;
		s->hist.erase(s->hist.begin(), s->hist.begin() + drop);
		s->hist.insert(s->hist.end(), tail.begin(), tail.end());
	}
	if (out_len) s->ratio = std::max<uint32_t>(1, std::min<uint32_t>(1000, (uint32_t)((1000ull * (consumed + 1)) / (out_len + 1))));
	s->sync_point = sync_point;
	if (st.final) { s->st = Inflate::TRAILER; s->resuming = false; }
	else {
		const uint32_t part = (uint32_t)(end_bit & 7);
		s->resuming = true;
		s->sfbt = st.sfbt ? st.sfbt : 0xe; s->subc = part ? 8 - part : 0; s->rem = st.rem;
		if ((st.sfbt & 0xe) == 0xc) { s->dhtlen = st.dhtlen; memcpy(s->dht, st.dht, NXZ_DHT_MAXSZ); }
	}
	publish(s);
	return true;
}

// one engine job (nx_inflate_, lib/nx_inflate.c:1143-1744).  Returns Z_OK / error.
int inflate_job(Inflate *s)
{
	z_streamp z = s->z;
	if (parallel_inflate(s)) return Z_OK;
	// source = [history rounded up to 16 B][carry][part of next_in]; size it from the last ratio
	uint32_t want_out = (uint32_t)std::min<size_t>((size_t)z->avail_out + WINDOW + (WINDOW >> 2), 1u << 20);
	// (a job that overflows its target is run again with a quarter of the source -- the engine reports no
	// state to resume from then, as the reference's does not, lib/nx_inflate.c:1399-1424 -- so the source is
	// cut for 5/8 of the room: the last ratio is only an estimate of the next stretch)
	uint32_t src_want = (uint32_t)(((uint64_t)want_out * s->ratio * 5 / 8 + 1000) / 1000);
	src_want = std::max<uint32_t>(src_want, 16);
	// This engine can do better (NXZ_JOB_SUSPEND_WHEN_FULL, include/nxz_engine.h): a job whose target fills
	// up suspends there and is resumed, nothing is decoded twice, so the source need not be rationed.
	const bool suspend_when_full = device_dht_available();
	if (suspend_when_full) src_want = 1u << 20;
	for (int attempt = 0; attempt < 24; attempt++) {
		uint32_t histlen = (uint32_t)s->hist.size();
		uint32_t pad = (16 - (histlen & 15)) & 15;                 // history prefix is given in 16-byte units
		uint32_t from_carry = (uint32_t)s->carry.size();
		uint32_t from_next = (uint32_t)std::min<size_t>(z->avail_in, src_want > from_carry ? src_want - from_carry : 0);
		if (from_carry + from_next == 0) return Z_OK;
		s->src.resize(pad + histlen + from_carry + from_next);
		memset(s->src.data(), 0, pad);
		memcpy(s->src.data() + pad, s->hist.data(), histlen);
		memcpy(s->src.data() + pad + histlen, s->carry.data(), from_carry);
		memcpy(s->src.data() + pad + histlen + from_carry, z->next_in, from_next);
		uint32_t cap = std::max<uint32_t>(want_out, 65536);
		s->out.resize(cap);
		nxz_crb_cpb_t *j = s->jb.job;
		memset(j, 0, sizeof(*j));
		const bool resume = s->resuming || histlen;
		nxz_set_fc(j, resume ? NXZ_FC_DECOMPRESS_RESUME : NXZ_FC_DECOMPRESS);
		if (suspend_when_full) nxz_wr32(&j->crb.reserved1, NXZ_JOB_SUSPEND_WHEN_FULL);
		nxz_set_in_histlen(&j->cpb, (pad + histlen) / 16);
		nxz_set_in_crc(&j->cpb, s->crc); nxz_set_in_adler(&j->cpb, s->adler);
		if (s->resuming) {
			nxz_set_in_sfbt(&j->cpb, s->sfbt); nxz_set_in_subc(&j->cpb, s->subc);
			if ((s->sfbt & 0xe) == 0x8) nxz_set_in_rembytecnt(&j->cpb, s->rem);
			if ((s->sfbt & 0xe) == 0xc) { nxz_set_in_dhtlen(&j->cpb, s->dhtlen); memcpy(j->cpb.in_dht, s->dht, NXZ_DHT_MAXSZ); }
		}
		nxz_dde_set_direct(&j->crb.source, s->src.data(), (uint32_t)s->src.size());
		nxz_dde_set_direct(&j->crb.target, s->out.data(), cap);
		int cc = s->eng.submit(j);
		if (cc == NXZ_CC_TARGET_SPACE) {
			// halve the source and retry (:1399-1424)
			uint32_t have = from_carry + from_next;
			if (have <= 1) { want_out = std::min<uint32_t>(want_out * 4, 4u << 20); if (want_out >= (4u << 20)) return Z_BUF_ERROR; }
			src_want = std::max<uint32_t>(have / 4, 1);
			if (src_want < from_carry) { want_out = std::min<uint32_t>(want_out * 4, 4u << 20); src_want = from_carry; }
			continue;
		}
		if (cc != NXZ_CC_OK && cc != NXZ_CC_DATA_LENGTH) return Z_DATA_ERROR;
		uint32_t tpbc = nxz_csb_tpbc(j), spbc = nxz_rd32(&j->cpb.u.d.out_spbc_decomp_be);
		uint32_t sfbt = nxz_out_sfbt(&j->cpb), subc = nxz_out_subc(&j->cpb);
		uint32_t given = from_carry + from_next;
		if (spbc < pad + histlen) spbc = pad + histlen;           // (an engine that reports less than the history it was given)
		uint32_t used = spbc - (pad + histlen);                   // source bytes the engine looked at
		uint32_t consumed;
		s->crc = nxz_out_crc(&j->cpb); s->adler = nxz_out_adler(&j->cpb);
		bool final = false;
		if (cc == NXZ_CC_OK) { consumed = used; final = true; }
		else if (sfbt == 0) { consumed = used - subc / 8; final = true; }
		else {
			consumed = used - (subc + 7) / 8;                      // re-supply the partial byte (:1464-1609)
			s->sfbt = sfbt; s->subc = subc % 8; s->resuming = true;
			if ((sfbt & 0xe) == 0x8) s->rem = nxz_out_rembytecnt(&j->cpb);
			if ((sfbt & 0xe) == 0xc) { s->dhtlen = nxz_out_dhtlen(&j->cpb); memcpy(s->dht, j->cpb.u.d.out_dht, NXZ_DHT_MAXSZ); }
			// sync point: source ended right after an empty stored block header (:1563-1583)
			s->sync_point = false;
			if ((sfbt & 0xe) == 0xe && subc >= 3 && subc <= 10 && used >= 2) {
				const uint8_t *last = s->src.data() + pad + histlen + used - 1;
				s->sync_point = subc <= 8 ? !(*last & (uint8_t)(0xff << (8 - subc)))
							   : (*last == 0 && !(*(last - 1) & (uint8_t)(0xff << (16 - subc))));
			}
		}
		// what stays for the next job: unconsumed bytes taken from carry / next_in
		std::vector<uint8_t> rest(s->src.begin() + pad + histlen + consumed, s->src.begin() + pad + histlen + given);
		z->next_in += from_next; z->avail_in -= from_next; z->total_in += from_next;
		if (final || rest.size() > CARRY_KEEP) {
			// unconsumed bytes go back to the caller's view as far as they came from next_in (all of them behind a
			// finished stream or when the job stopped for want of room in the target; the tail of a token or header
			// that the source ended in stays here)
			size_t giveback = std::min<size_t>(rest.size(), from_next);
			z->next_in -= giveback; z->avail_in += (uInt)giveback; z->total_in -= giveback;
			rest.resize(rest.size() - giveback);
		}
		s->carry.swap(rest);
		if (consumed || tpbc) s->ratio = std::max<uint32_t>(1, std::min<uint32_t>(1000, (uint32_t)((1000ull * (consumed + 1)) / (tpbc + 1ull))));
		else if (cc == NXZ_CC_DATA_LENGTH && sfbt == 0xe && subc > 0) s->ratio = 1000;
		s->produce(s->out.data(), tpbc);
		publish(s);
		if (final) s->st = Inflate::TRAILER;
		return Z_OK;
	}
	return Z_BUF_ERROR;
}

int inflate_reset(z_streamp strm)
{
	Inflate *s = istate(strm);
	if (!s) return Z_STREAM_ERROR;
	strm->total_in = strm->total_out = 0; strm->msg = Z_NULL;
	if (s->st != Inflate::DONE) s->unget.clear();           // (behind a finished stream: the next member's first bytes stay)
	s->st = Inflate::HEADER; s->held = s->nheld = 0; s->gzflags = 0; s->pend.clear(); s->pend_off = 0;
	s->hist.clear(); s->carry.clear(); s->resuming = false; s->sfbt = s->subc = s->rem = s->dhtlen = 0;
	s->crc = 0; s->adler = 1; s->total_out = 0; s->ntrailer = 0; s->sync_point = false; s->have_dict = false; s->ratio = 250; s->par_skip = 0; s->par_declined = 0;
	s->hcrc = 0;
	return Z_OK;
}

} // namespace

extern "C" int nx_inflateReset(z_streamp strm) { return inflate_reset(strm); }

// (for the gz-file layer, which puts them back in front of its own buffer: nxz_gzfile.cpp)
extern "C" size_t nxz_inflate_unget_size(z_streamp strm) { Inflate *s = istate(strm); return s && s->st == Inflate::DONE ? s->unget.size() : 0; }
extern "C" void nxz_inflate_take_unget(z_streamp strm, unsigned char *dst)
{
	Inflate *s = istate(strm);
	if (!s || s->unget.empty()) return;
	memcpy(dst, s->unget.data(), s->unget.size());
	// (they were counted as consumed when they were taken)
	strm->total_in -= std::min<uLong>(strm->total_in, (uLong)s->unget.size());
	s->unget.clear();
}

// The reference's inflateResetKeep is the same reset (lib/nx_inflate.c: resets state, totals and history).
extern "C" int nx_inflateResetKeep(z_streamp strm) { return inflate_reset(strm); }

// Deep copy of a decompression stream (lib/nx_inflate.c:1876-1942): state, pending output,
// history, carried source bytes and the resume fields; the copy gets its own engine handle.
extern "C" int nx_inflateCopy(z_streamp dest, z_streamp source)
{
	Inflate *s = istate(source);
	if (!s || !dest) return Z_STREAM_ERROR;
	Inflate *d = new (std::nothrow) Inflate();
	if (!d || !d->jb.job || !d->eng.begin()) { delete d; return Z_MEM_ERROR; }
	*dest = *source;
	d->z = dest; d->wrap = s->wrap; d->window_bits = s->window_bits; d->st = s->st; d->init_wbits = s->init_wbits;
	d->held = s->held; d->nheld = s->nheld; d->gzflags = s->gzflags; d->xlen = s->xlen; d->zcmf = s->zcmf; d->dictid = s->dictid;
	d->gzhead = s->gzhead; d->hcrc = s->hcrc;
	d->pend = s->pend; d->pend_off = s->pend_off; d->hist = s->hist; d->carry = s->carry; d->unget = s->unget;
	d->sfbt = s->sfbt; d->subc = s->subc; d->rem = s->rem; d->dhtlen = s->dhtlen; memcpy(d->dht, s->dht, sizeof(d->dht));
	d->resuming = s->resuming; d->crc = s->crc; d->adler = s->adler; d->total_out = s->total_out;
	memcpy(d->trailer, s->trailer, sizeof(d->trailer)); d->ntrailer = s->ntrailer;
	d->sync_point = s->sync_point; d->have_dict = s->have_dict; d->ratio = s->ratio;
	dest->state = (struct internal_state *)d;
	return Z_OK;
}

extern "C" int nx_inflateReset2(z_streamp strm, int windowBits)
{
	Inflate *s = istate(strm);
	if (!s) return Z_STREAM_ERROR;
	const int wbits_given = windowBits;
	int wrap;                                                    // lib/nx_inflate.c:147-164
	if (windowBits < 0) { wrap = HDR_RAW; windowBits = -windowBits; if (windowBits < 8 || windowBits > 15) return Z_STREAM_ERROR; }
	else if (windowBits >= 8 && windowBits <= 15) wrap = HDR_ZLIB;
	else if (windowBits >= 24 && windowBits <= 31) wrap = HDR_GZIP;
	else if (windowBits >= 40 && windowBits <= 47) wrap = HDR_ZLIB | HDR_GZIP;
	else if (windowBits == 0) { wrap = HDR_ZLIB; windowBits = 15; }
	else return Z_STREAM_ERROR;
	s->wrap = wrap; s->window_bits = windowBits;
	s->init_wbits = wbits_given;                                 // what AUTO mode reopens the stream with (nxz_inflate_pristine)
	return inflate_reset(strm);
}

extern "C" int nx_inflateInit2_(z_streamp strm, int windowBits, const char *version, int stream_size)
{
	if (version == Z_NULL || version[0] != ZLIB_VERSION[0] || stream_size != (int)sizeof(z_stream)) return Z_VERSION_ERROR;
	if (strm == Z_NULL) return Z_STREAM_ERROR;
	strm->msg = Z_NULL;
	Inflate *s = new (std::nothrow) Inflate();
	if (!s || !s->jb.job) { delete s; return Z_MEM_ERROR; }
	if (!s->eng.begin()) { delete s; return Z_STREAM_ERROR; }
	s->z = strm;
	strm->state = (struct internal_state *)s;
	int rc = nx_inflateReset2(strm, windowBits);
	if (rc != Z_OK) { s->eng.end(); delete s; strm->state = Z_NULL; }
	return rc;
}

// AUTO mode's switchable streams (nxz_host.h)
extern "C" int nxz_inflate_pristine(z_streamp strm, int *wbits)
{
	Inflate *s = istate(strm);
	if (!s || s->st != Inflate::HEADER || s->nheld || s->have_dict || s->gzhead || strm->total_in || strm->total_out || !s->carry.empty() ||
	    !s->unget.empty())            // (the next gzip member's first bytes, kept over inflateReset: a reopened stream would lose them)
		return 0;
	*wbits = s->init_wbits;
	return 1;
}
extern "C" int nxz_deflate_pristine(z_streamp strm, int *level, int *wbits, int *strategy, int *memlevel)
{
	Deflate *s = dstate(strm);
	if (!s || s->st != Deflate::INIT || s->used || s->dict_len || s->gzhead || s->hist_len || strm->total_in || strm->total_out) return 0;
	*level = s->init_level; *wbits = s->init_wbits; *strategy = s->strategy; *memlevel = s->init_memlevel;
	return 1;
}

extern "C" int nx_inflateInit_(z_streamp strm, const char *version, int stream_size)
{
	return nx_inflateInit2_(strm, 47, version, stream_size);      // DEF_WBITS, lib/nx_zlib.h:117-119
}

extern "C" int nx_inflateEnd(z_streamp strm)
{
	Inflate *s = istate(strm);
	if (!s) return Z_STREAM_ERROR;
	s->eng.end(); s->magic = 0;
	delete s;
	strm->state = Z_NULL;
	return Z_OK;
}

extern "C" int nx_inflateSyncPoint(z_streamp strm)
{
	Inflate *s = istate(strm);
	if (!s) return Z_STREAM_ERROR;
	return s->sync_point ? 1 : 0;
}

extern "C" int nx_inflateGetHeader(z_streamp strm, gz_headerp head)
{
	Inflate *s = istate(strm);
	if (!s || !(s->wrap & HDR_GZIP)) return Z_STREAM_ERROR;
	s->gzhead = head;
	if (head) head->done = 0;
	return Z_OK;
}

extern "C" int nx_inflateSetDictionary(z_streamp strm, const unsigned char *dictionary, unsigned int dictLength)
{
	Inflate *s = istate(strm);
	if (!s || !dictionary) return Z_STREAM_ERROR;
	if (s->wrap == HDR_ZLIB) {
		if (s->st != Inflate::NEED_DICT) return Z_STREAM_ERROR;
		if (sw_adler32(1, dictionary, dictLength) != s->dictid) return Z_DATA_ERROR;
		s->st = Inflate::BODY;
	} else if (s->wrap != HDR_RAW) return Z_STREAM_ERROR;
	uint32_t keep = std::min<uint32_t>(dictLength, WINDOW);
	s->hist.assign(dictionary + dictLength - keep, dictionary + dictLength);
	s->have_dict = true;
	return Z_OK;
}

extern "C" int nx_inflate(z_streamp strm, int flush)
{
	Inflate *s = istate(strm);
	if (!s) return Z_STREAM_ERROR;
	if (flush == Z_BLOCK || flush == Z_TREES) { strm->msg = (char *)"Z_BLOCK or Z_TREES not implemented"; return Z_STREAM_ERROR; }
	if (strm->next_out == Z_NULL && strm->avail_out) return Z_STREAM_ERROR;
	const uInt in0 = strm->avail_in, out0 = strm->avail_out;
	struct Caller { Caller() { g_inflate_callers.fetch_add(1, std::memory_order_relaxed); } ~Caller() { g_inflate_callers.fetch_sub(1, std::memory_order_relaxed); } } caller_here;
	s->whole_at_once = (flush == Z_FINISH || s->one_shot_hint) && strm->total_in == 0;
	int rc = Z_OK;
	uint32_t c;
#define NEXT(state) do { s->st = Inflate::state; } while (0)
#define NEEDBYTE() do { if (!get_byte(s, c)) goto out; } while (0)
	for (;;) {
		switch (s->st) {
		case Inflate::HEADER:
			if (s->wrap == (HDR_ZLIB | HDR_GZIP)) {
				NEEDBYTE();
				if (c == 0x1f) { s->wrap = HDR_GZIP; const uint8_t b = 0x1f; s->hcrc = sw_crc32(0, &b, 1); NEXT(GZ_ID2); }
				else if ((c & 0x0f) == 0x08 && ((c >> 4) & 0x0f) < 8) { s->wrap = HDR_ZLIB; s->zcmf = c; NEXT(ZL_FLG); }
				else { strm->msg = (char *)"incorrect header"; NEXT(BAD); }
			} else if (s->wrap == HDR_ZLIB) NEXT(ZL_CMF);
			else if (s->wrap == HDR_GZIP) {
				NEEDBYTE();
				if (c != 0x1f) { strm->msg = (char *)"incorrect gzip header"; NEXT(BAD); } else { const uint8_t b = 0x1f; s->hcrc = sw_crc32(0, &b, 1); NEXT(GZ_ID2); }
			} else { s->crc = 0; s->adler = 1; NEXT(BODY); }
			break;
		case Inflate::GZ_ID2: NEEDBYTE(); if (c != 0x8b) { strm->msg = (char *)"incorrect gzip header"; NEXT(BAD); } else NEXT(GZ_CM); break;
		case Inflate::GZ_CM: NEEDBYTE(); if (c != 8) { strm->msg = (char *)"unknown compression method"; NEXT(BAD); } else NEXT(GZ_FLG); break;
		case Inflate::GZ_FLG:
			NEEDBYTE(); s->gzflags = c;
			if (c & 0xe0) { strm->msg = (char *)"unknown header flags set"; NEXT(BAD); break; }
			if (s->gzhead) { s->gzhead->text = c & 1; s->gzhead->time = 0; }
			s->nheld = 0; NEXT(GZ_MTIME); break;
		case Inflate::GZ_MTIME:
			while (s->nheld < 4) { NEEDBYTE(); if (s->gzhead) s->gzhead->time |= (uLong)c << (8 * s->nheld); s->nheld++; }
			s->nheld = 0; NEXT(GZ_XFL); break;
		case Inflate::GZ_XFL: NEEDBYTE(); if (s->gzhead) s->gzhead->xflags = (int)c; NEXT(GZ_OS); break;
		case Inflate::GZ_OS: NEEDBYTE(); if (s->gzhead) s->gzhead->os = (int)c; s->nheld = 0; s->held = 0; NEXT(GZ_XLEN); break;
		case Inflate::GZ_XLEN:
			if (s->gzflags & 4) {
				while (s->nheld < 2) { NEEDBYTE(); s->held |= c << (8 * s->nheld); s->nheld++; }
				s->xlen = s->held; if (s->gzhead) s->gzhead->extra_len = s->xlen;
			} else { s->xlen = 0; if (s->gzhead) s->gzhead->extra = Z_NULL; }
			s->nheld = 0; NEXT(GZ_EXTRA); break;
		case Inflate::GZ_EXTRA:
			while (s->nheld < s->xlen) {
				NEEDBYTE();
				if (s->gzhead && s->gzhead->extra && s->nheld < s->gzhead->extra_max) s->gzhead->extra[s->nheld] = (Bytef)c;
				s->nheld++;
			}
			s->nheld = 0; NEXT(GZ_NAME); break;
		case Inflate::GZ_NAME:
			if (s->gzflags & 8) {
				do { NEEDBYTE(); if (s->gzhead && s->gzhead->name && s->nheld < s->gzhead->name_max) s->gzhead->name[s->nheld++] = (Bytef)c; } while (c);
			} else if (s->gzhead) s->gzhead->name = Z_NULL;
			s->nheld = 0; NEXT(GZ_COMMENT); break;
		case Inflate::GZ_COMMENT:
			if (s->gzflags & 16) {
				do { NEEDBYTE(); if (s->gzhead && s->gzhead->comment && s->nheld < s->gzhead->comm_max) s->gzhead->comment[s->nheld++] = (Bytef)c; } while (c);
			} else if (s->gzhead) s->gzhead->comment = Z_NULL;
			s->nheld = 0; NEXT(GZ_HCRC); break;
		case Inflate::GZ_HCRC:
			if (s->gzflags & 2) {
				while (s->nheld < 2) { NEEDBYTE(); s->held = s->nheld ? s->held | c << 8 : c; s->nheld++; }
				if (s->held != (s->hcrc & 0xffff)) { strm->msg = (char *)"header crc mismatch"; NEXT(BAD); break; }
			}
			if (s->gzhead) { s->gzhead->hcrc = (s->gzflags >> 1) & 1; s->gzhead->done = 1; }
			s->crc = 0; s->adler = 1; strm->adler = 0; NEXT(BODY); break;
		case Inflate::ZL_CMF:
			NEEDBYTE(); s->zcmf = c;
			if ((c & 0x0f) != 8) { strm->msg = (char *)"unknown compression method"; NEXT(BAD); break; }
			if (((c >> 4) & 0x0f) >= 8) { strm->msg = (char *)"invalid window size"; NEXT(BAD); break; }
			NEXT(ZL_FLG); break;
		case Inflate::ZL_FLG:
			NEEDBYTE();
			if (((s->zcmf << 8) + c) % 31) { strm->msg = (char *)"incorrect header check"; NEXT(BAD); break; }
			s->nheld = 0; s->held = 0;
			if (c & 0x20) NEXT(ZL_DICTID); else { s->crc = 0; s->adler = 1; strm->adler = 1; NEXT(BODY); }
			break;
		case Inflate::ZL_DICTID:
			while (s->nheld < 4) { NEEDBYTE(); s->held = (s->held << 8) | c; s->nheld++; }
			s->dictid = s->held; strm->adler = s->dictid; s->crc = 0; s->adler = 1;
			NEXT(NEED_DICT);
			/* fall through */
		case Inflate::NEED_DICT:
			if (!s->have_dict) { rc = Z_NEED_DICT; goto out; }
			NEXT(BODY); break;
		case Inflate::BODY:
			if (!s->unget.empty()) { s->carry.insert(s->carry.begin(), s->unget.begin(), s->unget.end()); s->unget.clear(); }
			s->drain();
			if (s->pending()) goto out;
			if (strm->avail_out == 0) goto out;
			if (strm->avail_in == 0 && s->carry.empty()) goto out;
			// (The reference gathers inputs below its cache threshold here and returns Z_OK without decoding,
			// lib/nx_inflate.c:1184-1205.  Not replicated: a caller in the canonical zlib loop -- zpipe.c: feed what
			// fread() gave, stop at Z_STREAM_END, end of file otherwise means a truncated stream -- never gets the end
			// of a stream whose last chunk, or whole length, is shorter than the threshold.  Found by the interoperability
			// matrix, tests/test_gpu_oct.py; inflate() decodes what it is given.)
			rc = inflate_job(s);
			if (rc != Z_OK) { if (rc == Z_DATA_ERROR) NEXT(BAD); goto out; }
			if (s->st == Inflate::BODY && s->pending()) goto out;
			if (s->st == Inflate::BODY && strm->avail_in == 0 && !s->carry.empty() && strm->avail_out) {
				// the engine wants more source than we hold: wait for the caller
				goto out;
			}
			break;
		case Inflate::TRAILER: {
			uint32_t need = s->wrap == HDR_GZIP ? 8 : s->wrap == HDR_ZLIB ? 4 : 0;
			while (s->ntrailer < need) {
				if (!s->carry.empty()) { s->trailer[s->ntrailer++] = s->carry.front(); s->carry.erase(s->carry.begin()); continue; }
				NEEDBYTE(); s->trailer[s->ntrailer++] = (uint8_t)c;
			}
			const uint8_t *t = s->trailer;
			bool ok = true;
			if (s->wrap == HDR_GZIP) {
				uint32_t ck = t[0] | t[1] << 8 | t[2] << 16 | (uint32_t)t[3] << 24, isz = t[4] | t[5] << 8 | t[6] << 16 | (uint32_t)t[7] << 24;
				ok = ck == s->crc && isz == (uint32_t)s->total_out;
			} else if (s->wrap == HDR_ZLIB) {
				uint32_t ck = (uint32_t)t[0] << 24 | t[1] << 16 | t[2] << 8 | t[3];
				ok = ck == s->adler;
			}
			if (!ok) { strm->msg = (char *)"incorrect data check"; NEXT(BAD); break; }
			// what was taken from the caller beyond this stream (small inputs are gathered before the engine
			// sees them) is the start of whatever follows: kept for it
			if (!s->carry.empty()) { s->unget.insert(s->unget.end(), s->carry.begin(), s->carry.end()); s->carry.clear(); }
			NEXT(DONE);
			break;
		}
		case Inflate::DONE:
			s->drain();
			rc = s->pending() ? Z_OK : Z_STREAM_END;
			goto out;
		case Inflate::BAD:
			rc = Z_DATA_ERROR;
			goto out;
		}
	}
out:
#undef NEXT
#undef NEEDBYTE
	if (rc == Z_OK) s->drain();
	// zlib's progress rule (lib/nx_inflate.c:738-746)
	if (in0 == strm->avail_in && out0 == strm->avail_out && rc == Z_OK) return Z_BUF_ERROR;
	if (flush == Z_FINISH && rc == Z_OK) return Z_BUF_ERROR;
	return rc;
}

// ---------------------------------------------------------------------------
// one-shot (lib/nx_compress.c:26-75, lib/nx_uncompr.c:32-88)
// ---------------------------------------------------------------------------
namespace {
struct ApiTrace {
	std::atomic<uint64_t> calls{0}, ns_init{0}, ns_body{0}, ns_end{0};
	bool on = getenv("NXZ_API_TRACE") != nullptr;
	static uint64_t now() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
	~ApiTrace()
	{
		if (on && calls) fprintf(stderr, "nxz api trace: %llu nx_compress2 calls; per call: init %.1f us, deflate %.1f us, end %.1f us\n", (unsigned long long)calls,
					 ns_init / (double)calls * 1e-3, ns_body / (double)calls * 1e-3, ns_end / (double)calls * 1e-3);
	}
} g_api;
struct ApiTraceU {
	std::atomic<uint64_t> calls{0}, ns_init{0}, ns_body{0}, ns_end{0};
	~ApiTraceU()
	{
		if (g_api.on && calls) fprintf(stderr, "nxz api trace: %llu nx_uncompress2 calls; per call: init %.1f us, inflate %.1f us, end %.1f us\n", (unsigned long long)calls,
					       ns_init / (double)calls * 1e-3, ns_body / (double)calls * 1e-3, ns_end / (double)calls * 1e-3);
	}
} g_apiu;
}

extern "C" int nx_compress2(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen, int level)
{
	z_stream st;
	memset(&st, 0, sizeof(st));
	const uint64_t t0 = g_api.on ? ApiTrace::now() : 0;
	int rc = nx_deflateInit(&st, level);
	if (rc != Z_OK) return rc;
	const uint64_t t1 = g_api.on ? ApiTrace::now() : 0;
	const uInt maxu = 1u << 30;
	uLong remaining = *destLen;
	*destLen = 0;
	st.next_out = dest; st.avail_out = 0; st.next_in = (z_const Bytef *)source; st.avail_in = 0;
	do {
		if (st.avail_out == 0) { st.avail_out = remaining > maxu ? maxu : (uInt)remaining; remaining -= st.avail_out; }
		if (st.avail_in == 0) { st.avail_in = sourceLen > maxu ? maxu : (uInt)sourceLen; sourceLen -= st.avail_in; }
		rc = nx_deflate(&st, sourceLen ? Z_NO_FLUSH : Z_FINISH);
	} while (rc == Z_OK);
	*destLen = st.total_out;
	const uint64_t t2 = g_api.on ? ApiTrace::now() : 0;
	nx_deflateEnd(&st);
	if (g_api.on) { g_api.calls++; g_api.ns_init += t1 - t0; g_api.ns_body += t2 - t1; g_api.ns_end += ApiTrace::now() - t2; }
	return rc == Z_STREAM_END ? Z_OK : rc;
}

extern "C" int nx_compress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen)
{
	return nx_compress2(dest, destLen, source, sourceLen, Z_DEFAULT_COMPRESSION);
}

extern "C" uLong nx_compressBound(uLong sourceLen) { return nx_deflateBound(NULL, sourceLen); }

extern "C" int nx_uncompress2(Bytef *dest, uLongf *destLen, const Bytef *source, uLong *sourceLen)
{
	z_stream st;
	memset(&st, 0, sizeof(st));
	const uInt maxu = 1u << 30;
	uLong len = *sourceLen, left;
	Byte buf[1];
	if (*destLen) { left = *destLen; *destLen = 0; } else { left = 1; dest = buf; }
	st.next_in = (z_const Bytef *)source; st.avail_in = 0;
	const uint64_t t0 = g_api.on ? ApiTrace::now() : 0;
	int rc = nx_inflateInit(&st);
	if (rc != Z_OK) return rc;
	const uint64_t t1 = g_api.on ? ApiTrace::now() : 0;
	if (Inflate *is = istate(&st)) is->one_shot_hint = len <= maxu;      // (the whole source is in the first call)
	st.next_out = dest; st.avail_out = 0;
	do {
		if (st.avail_out == 0) { st.avail_out = left > maxu ? maxu : (uInt)left; left -= st.avail_out; }
		if (st.avail_in == 0) { st.avail_in = len > maxu ? maxu : (uInt)len; len -= st.avail_in; }
		rc = nx_inflate(&st, Z_NO_FLUSH);
	} while (rc == Z_OK);
	*sourceLen -= len + st.avail_in;
	if (dest != buf) *destLen = st.total_out;
	else if (st.total_out && rc == Z_BUF_ERROR) left = 1;
	const uint64_t t2 = g_api.on ? ApiTrace::now() : 0;
	nx_inflateEnd(&st);
	if (g_api.on) { g_apiu.calls++; g_apiu.ns_init += t1 - t0; g_apiu.ns_body += t2 - t1; g_apiu.ns_end += ApiTrace::now() - t2; }
	return rc == Z_STREAM_END ? Z_OK : rc == Z_NEED_DICT ? Z_DATA_ERROR : rc == Z_BUF_ERROR && left + st.avail_out ? Z_DATA_ERROR : rc;
}

extern "C" int nx_uncompress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen)
{
	return nx_uncompress2(dest, destLen, source, &sourceLen);
}

// ---------------------------------------------------------------------------
// software checksums
// ---------------------------------------------------------------------------
extern "C" unsigned long nx_crc32(unsigned long crc, const unsigned char *buf, size_t len)
{
	if (buf == Z_NULL) return 0;
	return ~__crc32_vpmsum(~(uint32_t)crc, buf, len) & 0xffffffffu;                 // lib/crc32_ppc.c:22-67
}

extern "C" unsigned long nx_adler32(unsigned long adler, const unsigned char *buf, size_t len)
{
	if (buf == Z_NULL) return 1;
	return sw_adler32((uint32_t)adler, buf, len);
}

// lib/nx_adler32.c:150 (lib/Versions:29)
extern "C" unsigned long nx_adler32_z(unsigned long adler, const unsigned char *buf, size_t len) { return nx_adler32(adler, buf, len); }
extern "C" unsigned long nx_crc32_combine64(unsigned long crc1, unsigned long crc2, off_t len2) { return nx_crc32_combine(crc1, crc2, len2); }
extern "C" unsigned long nx_adler32_combine64(unsigned long a1, unsigned long a2, off_t len2) { return nx_adler32_combine(a1, a2, len2); }

extern "C" unsigned long nx_crc32_combine(unsigned long crc1, unsigned long crc2, off_t len2)
{
	if (len2 <= 0) return crc1;
	uint32_t r = 0x80000000u, sq = 0x00800000u;                                     // x^(8*len2) mod P
	for (uint64_t n = (uint64_t)len2; n; n >>= 1) { if (n & 1) r = gf2_mul(r, sq); sq = gf2_mul(sq, sq); }
	return gf2_mul((uint32_t)crc1, r) ^ (uint32_t)crc2;
}

extern "C" unsigned long nx_adler32_combine(unsigned long adler1, unsigned long adler2, off_t len2)
{
	if (len2 < 0) return 0xffffffffUL;
	const uint64_t B = 65521;
	uint64_t rem = (uint64_t)len2 % B, s1 = adler1 & 0xffff;
	uint64_t sum1 = (s1 + (adler2 & 0xffff) + B - 1) % B;
	uint64_t sum2 = (rem * s1 + ((adler1 >> 16) & 0xffff) + ((adler2 >> 16) & 0xffff) + B - rem) % B;
	return (unsigned long)((sum2 << 16) | sum1);
}
