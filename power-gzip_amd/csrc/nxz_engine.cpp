// nxz_engine.cpp -- host side of libnxz_engine.so: the C ABI of include/nxz_engine.h
// on top of the HIP kernels (nxz_lz77.hip, nxz_dhtgen.hip, nxz_encode.hip, nxz_inflate*.hip, nxz_misc.hip).
//
// What it replaces in the reference (paths relative to the libnxz tree):
//   lib/gzip_vas.c  -- open /dev/crypto/nx-gzip, VAS window, copy/paste of the
//                      CRB, CSB polling (:94-417).  Here: a HIP stream per job
//                      slot, pinned staging of the DDE gather/scatter lists,
//                      kernel launch, and completion written to the CSB.
//   lib/crc32_power.c -- __crc32_vpmsum (vector CRC on POWER).  Here: slice-by-8.
// There is NO CPU fallback for the engine ops: without a gfx950 device
// nx_function_begin fails with ENODEV and nxu_run_job completes jobs with
// CC = NXZ_CC_NO_HW.
#include <hip/hip_runtime.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <map>
#include <deque>
#include <mutex>
#include <vector>
#include "nxz_device.h"
#include "../../include/nxz_wire.h"

#define NXZ_VERSION "nxz-engine 0.1 (gfx950)"
#define SUBBLOCK 65536u
#define SLOTS 32

static thread_local char g_err[256];
static void set_err(const char *what, hipError_t e)
{
	snprintf(g_err, sizeof(g_err), "%s: %s", what, e == hipSuccess ? "error" : hipGetErrorString(e));
}
#define HIPCHK(x, fail) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(#x, e_); fail; } } while (0)

extern "C" const char *nxz_last_error(void) { return g_err; }
extern "C" size_t nxz_pinflate_trim(void);
static size_t trim_compress_scratch();
extern "C" size_t nxz_trim(void) { return nxz_pinflate_trim() + trim_compress_scratch(); }
extern "C" const char *nxz_engine_version(void) { return NXZ_VERSION; }
extern "C" size_t nxz_compress_bound(size_t n) { return ((n * 9 + 7) / 8 + 16 + 15) & ~(size_t)15; }

// one in-flight single job (nxu_run_job)
struct Slot {
	hipStream_t stream = nullptr;
	uint8_t *h_in = nullptr, *h_out = nullptr;      // pinned
	uint8_t *d_in = nullptr, *d_out = nullptr;
	nxz_batch_job_t *h_job = nullptr, *d_job = nullptr;
	nxz_batch_result_t *h_res = nullptr, *d_res = nullptr;
	nxz_batch_dht_t *h_dht = nullptr, *d_dht = nullptr;
	nxz_dht_prepared_t *d_prep = nullptr;
	uint32_t *h_cnt = nullptr, *d_cnt = nullptr;
	bool busy = false;
};

#define OUT_CAP (SUBBLOCK * 2 + 4096)    /* staging for one job's target */
#define INF_SRC_CAP (1u << 20)           /* decompress: source bytes taken per job */
#define INF_OUT_CAP (4u << 20)

constexpr int HOST_PAIRS = 16;
struct nxz_ctx {
	int device = 0;
	int refs = 0;
	hipStream_t stream = nullptr;                 // default stream for batch calls
	std::mutex mtx;
	std::condition_variable cv;
	Slot slots[SLOTS];
	// batch scratch, one set per stream the caller launches on: launches on different streams may
	// run at the same time, so they must not share the prepared tables or the decode workspace
	uint32_t *h_sample = nullptr;                 // pinned words the block-type sample of a large inflate batch lands in
	unsigned sample_turn = 0;
	struct Scratch {
		nxz_dht_prepared_t *d_prepared = nullptr;
		size_t prepared_cap = 0;
		uint8_t *d_lanes_ws = nullptr;            // per-lane decode tables of the batched inflate kernel
		size_t lanes_cap = 0;
		uint8_t *d_order_ws = nullptr;            // the jobs' order by length for the stream-per-wave kernel's larger batches
		size_t order_cap = 0;
		uint8_t *d_cut_ws = nullptr;              // small inflate batches cut into pieces (nxz_inflate_cut.hip): control arrays + the pieces' elements
		size_t cut_cap = 0;
		uint8_t *d_wg_ws = nullptr;               // a stream per workgroup (nxz_inflate_wg.hip): job counter, reasons, hand-back list
		size_t wg_cap = 0;
		// compress: what the LZ77 kernel hands to the entropy kernel, for one chunk of jobs
		uint8_t *d_tokens = nullptr;              // chunk x NXZ_TOK_STRIDE
		nxz_dht_prepared_t *d_gen = nullptr;      // tables the device generated, one per job of the chunk
		uint32_t *d_counts = nullptr;             // symbol counts when the caller did not ask for them
		uint16_t *d_cand2 = nullptr;              // LZ77 kernel: second bucket entries in transit, 32 KiB per workgroup
		uint8_t *d_fuse = nullptr;                // the fused dynamic-Huffman form: two token slots and two table slots per workgroup (nxz_lz77.hip gen::)
		size_t chunk_cap = 0;
		size_t chunk_limit = 0;                   // jobs per chunk the device had room for when a larger chunk could not be had (0: no such failure yet)
		void release_chunk() {
			if (d_tokens) (void)hipFree(d_tokens);                       // (d_gen and d_counts lie inside it)
			d_tokens = nullptr; d_gen = nullptr; d_counts = nullptr; chunk_cap = 0;
		}
		// the three buffers of a chunk, all or none
		bool alloc_chunk(size_t chunk) {
			// (one allocation: the tokens, then the tables, then the counts)
			const size_t tb = (chunk * (size_t)NXZ_TOK_STRIDE + 255) & ~(size_t)255, gb = (chunk * sizeof(nxz_dht_prepared_t) + 255) & ~(size_t)255;
			if (hipMalloc((void **)&d_tokens, tb + gb + chunk * 316 * sizeof(uint32_t)) == hipSuccess) {
				d_gen = (nxz_dht_prepared_t *)(d_tokens + tb);
				d_counts = (uint32_t *)(d_tokens + tb + gb);
				chunk_cap = chunk;
				return true;
			}
			(void)hipGetLastError();
			d_tokens = nullptr; d_gen = nullptr; d_counts = nullptr;
			return false;
		}
		void release() {
			if (d_prepared) (void)hipFree(d_prepared);
			if (d_lanes_ws) (void)hipFree(d_lanes_ws);
			if (d_order_ws) (void)hipFree(d_order_ws);
			if (d_cut_ws) (void)hipFree(d_cut_ws);
			if (d_wg_ws) (void)hipFree(d_wg_ws);
			if (d_tokens) (void)hipFree(d_tokens);                       // (d_gen and d_counts lie inside it)
			if (d_cand2) (void)hipFree(d_cand2);
			if (d_fuse) (void)hipFree(d_fuse);
			*this = Scratch();
		}
	};
	std::map<hipStream_t, Scratch> scratch;
	std::map<hipStream_t, std::mutex> scratch_use;   // held by a batch call from sizing its stream's scratch to its last launch
	// nxz_deflate_host: a call works on two lanes, each with its own stream, so that the copies of one group of
	// blocks run while the other group is in the kernels; HOST_PAIRS such pairs (made when first used, 100 MiB of device
	// memory each), for callers on different threads (four pairs: 16 threads spent three quarters of a call waiting for one)
	struct HostLane {
		hipStream_t stream = nullptr;
		uint8_t *d_src = nullptr, *d_dst = nullptr, *d_packed = nullptr;
		nxz_batch_job_t *d_jobs = nullptr, *h_jobs = nullptr;
		nxz_batch_result_t *d_res = nullptr, *h_res = nullptr;
		uint64_t *d_off = nullptr, *h_total = nullptr;
		uint8_t *h_src = nullptr, *h_packed = nullptr; // pinned staging for calls of a few MiB (null above STAGE_MAX_BLOCKS per group)
		uint8_t *d_base = nullptr, *h_base = nullptr; // ONE device and ONE pinned allocation hold all of the above (sixteen threads' first calls queue for the runtime's allocator)
		size_t n = 0; uint64_t bytes = 0;
		size_t cap = 0;                           // blocks per group the buffers hold
	} lanes[2 * HOST_PAIRS];
	std::mutex lanes_mtx[HOST_PAIRS];
	std::atomic<unsigned> lanes_turn{0};
	// nxz_deflate_host calls of a few MiB from many threads: the callers that are there at the same time put their blocks into
	// ONE batch (a launch of each kernel for all of them, on one stream), as the rounds below do for single-block jobs --
	// a HIP stream per caller does not get them side by side: the runtime maps the streams onto four hardware queues, and
	// sixteen threads of 1 MiB calls ran at 4 GiB/s, two or three calls at a time (merged_deflate)
	struct Merge {
		enum State { FREE, OPEN, RUNNING, DONE } state = FREE;
		hipStream_t stream = nullptr;
		uint8_t *h_src = nullptr, *h_packed = nullptr;       // pinned: the callers copy their source in and their stream out themselves
		uint8_t *d_src = nullptr, *d_dst = nullptr;
		nxz_batch_job_t *h_jobs = nullptr;                   // pinned, read and written by the kernels in place, as a round's
		nxz_batch_result_t *h_res = nullptr;
		uint64_t *h_off = nullptr;
		nxz_pack_member_t *h_mem = nullptr;
		uint16_t *h_member_of = nullptr;
		uint8_t *h_base = nullptr, *d_base = nullptr;
		int fc = 0; uint32_t H = 0;
		uint32_t slots = 0, jobs = 0, members = 0, filled = 0, left = 0;   // slots: 64 KiB units of staging (windows too); jobs: blocks
		int rc = 0;
		bool ready = false;
	} merges[3];
	std::mutex mm;
	std::condition_variable mcv;
	std::atomic<int> host_callers{0};                 // callers inside nxz_deflate_host at this moment
	hipStream_t split_stream = nullptr;               // nxz_batch_decompress: a large batch of streams that bring tables, shared out between two kernels
	hipEvent_t split_ev[2] = { nullptr, nullptr };
	std::mutex split_mtx;
	// nxu_run_job, compress: callers that arrive while a launch is in flight are gathered and go out
	// together as one launch of each kernel (run_compress / round_run)
	struct Round {
		hipStream_t stream = nullptr;
		nxz_batch_job_t *h_jobs = nullptr;        // pinned, read by the kernels in place
		nxz_batch_result_t *h_res = nullptr;      // pinned, written by the kernels in place
		nxz_batch_dht_t *h_dht = nullptr;
		uint32_t *h_cnt = nullptr;
		nxz_dht_prepared_t *d_prep = nullptr;
		uint8_t *d_tok = nullptr;
		uint16_t *d_cand2 = nullptr;
		uint8_t *d_src = nullptr;                 // the sources, brought over by one copy kernel (two kernels read them)
		uint8_t *d_cut = nullptr;                 // decompress rounds: the workspace of nxz_inflate_cut.hip (made when first used)
		size_t cut_arena = 0;
		uint8_t *d_wg = nullptr;                  // ... or of nxz_inflate_wg.hip (rounds of fresh streams of at most 64 KiB either side)
		uint8_t **h_targets = nullptr;            // ... and where the jobs' outputs go from the device buffers they are decoded into
		struct Item { const uint8_t *src; uint8_t *dst; uint64_t bytes; } *h_items = nullptr;
		bool busy = false, ready = false;
	} rounds[16];
	std::mutex qm;
	std::condition_variable qcv;
	std::deque<struct CompressReq *> q;
	std::deque<struct InflateReq *> qi;           // the same for decompress jobs
	uint32_t *d_job_counters = nullptr;           // job counters of the batched deflate launches (ring)
	unsigned next_counter = 0;
	// measurement aid (nxz_ctx_stage_timing): events around every kernel of the compress batches
	bool timing = false;
	std::vector<hipEvent_t> tev;                  // per chunk: before LZ77, after it, after dhtgen, after the entropy kernel
};
static constexpr unsigned JOB_COUNTERS = 256;
// Which inflate kernel a batch gets (profiles/r01c_inflate_by_batch_size.txt, 64 KiB streams):
//   up to NXZ_WINDOW_LDS_MAX streams   a stream per wave, window in LDS (4 per CU): 3.7-4.4 ms a round
//   below NXZ_LANES_MIN streams        a stream per wave, the target as window (20 per CU): 7.5 ms for
//                                      4096 streams, 40 GiB/s at 65 536
//   from NXZ_LANES_MIN streams on      a stream per lane: 55-60 ms however few streams, 51 GiB/s at 65 536,
//                                      110 at 262 144
// The wave kernels need 16-byte aligned sources, as the batch interface demands.
// Round 3 (profiles/r03_inflate_by_batch_size.txt): the lane kernel wins only on streams of fixed-Huffman (or stored)
// blocks -- one table for all lanes -- from about 100 000 streams on (91 against 44 GiB/s at 262 144); streams that
// bring a table each (zlib's, the engine's own exact-table output) run twice as fast a stream per wave at every batch
// size (81 against 40).  So a batch of NXZ_LANES_MIN streams or more is sampled first: 256 of its streams, the type
// of their first block.
// Round 4 (profiles/r04c_inflate_by_batch_size.txt): with its memory instructions issued where all lanes pass together
// the lane kernel does fixed-code streams at 36 GiB/s at 16 384 streams, 62 at 32 768, 105 at 65 536, 154 at 131 072
// (a stream per wave: 38, 42, 43, 44); streams with tables of their own are still the wave kernel's at every size.
// ... on the bench's synthetic blocks (ratio 1.75, a token every 2.3 bytes).  The corpus' blocks as fixed-code streams (ratio
// 2.9) go through the wave kernel at 73 GiB/s from 32 768 streams on and through the lane kernel at 51 / 88 / 164 at 32 768 /
// 65 536 / 262 144: the switch-over lies between the two kinds' break-evens.
#define NXZ_LANES_MIN 49152
#define NXZ_LANES_TABLES_MIN 163840   /* streams that bring tables: the lane kernel from here on */
#define NXZ_WINDOW_LDS_MAX 1024

static std::mutex g_mtx;
static nxz_ctx *g_ctx[64];
// The HIP runtime does not survive fork(): a child that inherits contexts must not touch them (the
// reference re-opens its device in the child, lib/nx_zlib.c:529-551; here the child is refused and
// the dispatch layer sends its streams to software zlib).
static pid_t g_creator_pid = 0;
static bool forked_child() { return g_creator_pid != 0 && getpid() != g_creator_pid; }

static bool slot_init(Slot &s)
{
	HIPCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking), return false);
	HIPCHK(hipHostMalloc((void **)&s.h_in, INF_SRC_CAP + 64), return false);
	HIPCHK(hipHostMalloc((void **)&s.h_out, INF_OUT_CAP), return false);
	HIPCHK(hipMalloc((void **)&s.d_in, INF_SRC_CAP + 64), return false);
	HIPCHK(hipMalloc((void **)&s.d_out, INF_OUT_CAP), return false);
	HIPCHK(hipHostMalloc((void **)&s.h_job, sizeof(nxz_batch_job_t)), return false);
	HIPCHK(hipMalloc((void **)&s.d_job, sizeof(nxz_batch_job_t)), return false);
	HIPCHK(hipHostMalloc((void **)&s.h_res, sizeof(nxz_batch_result_t)), return false);
	HIPCHK(hipMalloc((void **)&s.d_res, sizeof(nxz_batch_result_t)), return false);
	HIPCHK(hipHostMalloc((void **)&s.h_dht, sizeof(nxz_batch_dht_t)), return false);
	HIPCHK(hipMalloc((void **)&s.d_dht, sizeof(nxz_batch_dht_t)), return false);
	HIPCHK(hipMalloc((void **)&s.d_prep, sizeof(nxz_dht_prepared_t)), return false);
	HIPCHK(hipHostMalloc((void **)&s.h_cnt, 316 * 4), return false);
	HIPCHK(hipMalloc((void **)&s.d_cnt, 316 * 4), return false);
	return true;
}

static void slot_free(Slot &s)
{
	if (s.stream) (void)hipStreamDestroy(s.stream);
	(void)hipHostFree(s.h_in); (void)hipHostFree(s.h_out); (void)hipFree(s.d_in); (void)hipFree(s.d_out);
	(void)hipHostFree(s.h_job); (void)hipFree(s.d_job); (void)hipHostFree(s.h_res); (void)hipFree(s.d_res);
	(void)hipHostFree(s.h_dht); (void)hipFree(s.d_dht); (void)hipFree(s.d_prep);
	(void)hipHostFree(s.h_cnt); (void)hipFree(s.d_cnt);
	s = Slot();
}

extern "C" int nxz_engine_usable(void) { return forked_child() ? 0 : 1; }

// Which device a caller that names none (NX_GZIP_DEV_NUM = -1, the default) gets.  The reference opens the
// NX unit nearest the calling CPU, or the one NX_GZIP_DEV_NUM names (lib/nx_zlib.c:568-576,1081,1281-1287), so a
// process with many threads uses every engine of the machine.  Here: NXZ_DEVICE names one; else the calling
// THREAD keeps one device for all its streams -- the first thread of the process the current HIP device (what a
// single-threaded caller has always got), every further thread the next visible device in turn -- so that T
// threads spread over min(T, ndev) GPUs.  NXZ_DEVICE_POLICY=current: every thread the current device.
// Pure function of its arguments (tests/test_config.py calls it with made-up device counts).
extern "C" int nxz_pick_device(int requested, int ndev, int current, unsigned thread_index, int spread)
{
	if (ndev <= 0) return -1;
	if (requested >= 0) return requested < ndev ? requested : -1;      // an explicit ordinal pins (out of range: no such device)
	if (current < 0 || current >= ndev) current = 0;
	if (!spread) return current;
	return (int)(((unsigned)current + thread_index) % (unsigned)ndev);
}

static unsigned calling_thread_index()
{
	static std::atomic<unsigned> next{0};
	static thread_local unsigned mine = ~0u;
	if (mine == ~0u) mine = next.fetch_add(1);
	return mine;
}

extern "C" nxz_ctx_t *nxz_ctx_create(int device)
{
	if (forked_child()) {
		snprintf(g_err, sizeof(g_err), "this process was forked after the engine was opened: the HIP runtime does not survive fork()");
		errno = ENODEV;
		return nullptr;
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		snprintf(g_err, sizeof(g_err), "no HIP device: the DEFLATE engine needs a gfx950 GPU (no CPU fallback)");
		errno = ENODEV;
		return nullptr;
	}
	if (device < 0) {
		const char *e = getenv("NXZ_DEVICE");
		if (e) device = atoi(e);
		else {
			static const bool spread = !(getenv("NXZ_DEVICE_POLICY") && !strcmp(getenv("NXZ_DEVICE_POLICY"), "current"));
			int cur = 0;
			(void)hipGetDevice(&cur);
			device = nxz_pick_device(-1, ndev < 64 ? ndev : 64, cur, calling_thread_index(), spread);
		}
	}
	if (device < 0 || device >= ndev || device >= 64) { errno = ENODEV; snprintf(g_err, sizeof(g_err), "device %d out of range", device); return nullptr; }
	std::lock_guard<std::mutex> g(g_mtx);
	if (g_ctx[device]) { g_ctx[device]->refs++; (void)hipSetDevice(device); return g_ctx[device]; }
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, device), { errno = ENODEV; return nullptr; });
	if (!strstr(prop.gcnArchName, "gfx950")) {
		snprintf(g_err, sizeof(g_err), "device %d is %s: this engine is built for gfx950 only", device, prop.gcnArchName);
		errno = ENODEV;
		return nullptr;
	}
	HIPCHK(hipSetDevice(device), { errno = ENODEV; return nullptr; });
	nxz_ctx *c = new nxz_ctx();
	c->device = device;
	c->refs = 1;
	g_creator_pid = getpid();
	HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), { delete c; errno = ENODEV; return nullptr; });
	g_ctx[device] = c;
	return c;
}

// (for the engine's other translation units: the HIP device of a context)
extern "C" int nxz_ctx_device(nxz_ctx_t *c) { return c ? c->device : -1; }

extern "C" void nxz_ctx_destroy(nxz_ctx_t *c)
{
	if (!c || forked_child()) return;                     // the parent owns the device objects
	std::lock_guard<std::mutex> g(g_mtx);
	if (--c->refs > 0) return;
	(void)hipSetDevice(c->device);
	for (auto &s : c->slots) if (s.stream) slot_free(s);
	for (auto &kv : c->scratch) kv.second.release();
	if (c->d_job_counters) (void)hipFree(c->d_job_counters);
	if (c->h_sample) (void)hipHostFree(c->h_sample);
	for (auto &l : c->lanes) {
		if (!l.stream) continue;
		(void)hipStreamSynchronize(l.stream);
		{
			std::lock_guard<std::mutex> g2(c->mtx);
			auto it = c->scratch.find(l.stream);
			if (it != c->scratch.end()) { it->second.release(); c->scratch.erase(it); }
		}
		(void)hipFree(l.d_base); (void)hipHostFree(l.h_base);
		(void)hipStreamDestroy(l.stream);
		l = nxz_ctx::HostLane();
	}
	for (auto &m : c->merges) {
		if (!m.stream) continue;
		(void)hipStreamSynchronize(m.stream);
		{
			std::lock_guard<std::mutex> g2(c->mtx);
			auto it = c->scratch.find(m.stream);
			if (it != c->scratch.end()) { it->second.release(); c->scratch.erase(it); }
		}
		(void)hipFree(m.d_base); (void)hipHostFree(m.h_base);
		(void)hipStreamDestroy(m.stream);
		m = nxz_ctx::Merge();
	}
	for (auto &r : c->rounds) {
		if (r.stream) { (void)hipStreamSynchronize(r.stream); (void)hipStreamDestroy(r.stream); }
		(void)hipHostFree(r.h_jobs); (void)hipHostFree(r.h_res); (void)hipHostFree(r.h_dht); (void)hipHostFree(r.h_cnt);
		(void)hipFree(r.d_prep); (void)hipFree(r.d_tok); (void)hipFree(r.d_cand2); (void)hipFree(r.d_src); (void)hipHostFree(r.h_items);
		r = nxz_ctx::Round();
	}
	if (c->split_stream) {
		(void)hipStreamSynchronize(c->split_stream); (void)hipStreamDestroy(c->split_stream);
		if (c->split_ev[0]) (void)hipEventDestroy(c->split_ev[0]);
		if (c->split_ev[1]) (void)hipEventDestroy(c->split_ev[1]);
	}
	if (c->stream) (void)hipStreamDestroy(c->stream);
	g_ctx[c->device] = nullptr;
	delete c;
}

extern "C" int nxz_ctx_sync(nxz_ctx_t *c, void *stream)
{
	hipStream_t s = (hipStream_t)stream;   // NULL = the HIP default stream
	HIPCHK(hipStreamSynchronize(s), return -EIO);
	return 0;
}

// ---------------------------------------------------------------------------
// batched, device-resident interface
// ---------------------------------------------------------------------------
// Jobs per launch of the three compress kernels: bounds the token scratch (104 KiB per job: 832 MiB for 8192
// jobs, 6.6 GiB for 65536).  Larger chunks cost memory, smaller ones time: the LZ77 kernel is one persistent
// workgroup per CU, at the end of a launch CUs idle until the last job is done, and every chunk is three launches
// (the corpus, 262144 jobs: 92.8 GiB/s at 8192 jobs per launch, 94.7 at 16384, 95.8 at 32768, 96.4 at 65536).  So the
// chunk grows with the batch -- a quarter to an eighth of it, 8192 at least and 65536 at most: a caller with a few thousand jobs
// never pays gigabytes for them -- and falls back to 8192 when the device has no room for more.
// NXZ_COMPRESS_CHUNK fixes it.
static size_t compress_chunk(size_t n)
{
	static const size_t v = [] { const char *e = getenv("NXZ_COMPRESS_CHUNK"); size_t x = e ? (size_t)strtoull(e, nullptr, 0) : 0; return x >= 256 ? x : (size_t)0; }();
	if (v) return v;
	size_t c = 8192;
	while (c < 65536 && n >= 8 * c) c *= 2;
	return c;
}

// The compress function codes: LZ77 kernel (tokens, counts, checksums) -> [table generator] ->
// entropy kernel, chunk after chunk on the caller's stream.
extern "C" int nxz_batch_compress(nxz_ctx_t *c, int fc, const nxz_batch_job_t *jobs, size_t n,
				  const nxz_batch_dht_t *dht, size_t ntables, nxz_batch_result_t *results,
				  uint32_t *counts, void *stream)
{
	if (!c || !nxz_fc_is_compress((uint32_t)fc) || (fc & 1) || (fc & ~0x2e)) return -EINVAL;
	if (forked_child()) return -ENODEV;
	const bool gen = nxz_fc_is_dhtgen((uint32_t)fc);
	const bool isdht = nxz_fc_is_dht((uint32_t)fc), count = nxz_fc_has_count((uint32_t)fc);
	if (gen && !isdht) return -EINVAL;
	if (count && !counts) return -EINVAL;
	if (isdht && !gen && (!dht || !ntables)) return -EINVAL;
	if (n == 0) return 0;
	hipStream_t s = (hipStream_t)stream;   // NULL = the HIP default stream
	(void)hipSetDevice(c->device);
	nxz_dht_prepared_t *prepared = nullptr;
	// equal chunks (a last chunk of a few jobs would cost three launches for nothing)
	// the fixed code without counts: the LZ77 kernel writes the finished block itself (no tokens in device scratch,
	// no entropy launch, and so no reason to cut the batch into chunks: one launch, one tail)
	// ... and so it can for the additive DHTGEN function codes (round 5, NXZ_FUSED_GEN=1): the table of a block is made, and the block
	// encoded, inside the LZ77 kernel, a job behind the parse (nxz_lz77.hip gen::) -- one launch, 110 MB of scratch whatever the batch
	// instead of 104 KiB a job of a chunk.  Not the default: 75.7 against 103.1 GiB/s on the corpus (profiles/r05_fused_dhtgen.txt) -- the
	// table generator is a chain of dependent steps that ONE wavefront works through while fifteen wait at their barriers, where the
	// kernel of nxz_dhtgen.hip has thirty tables in flight on a CU and hides every one's latency behind the others'.
	const char *fge = getenv("NXZ_FUSED_GEN");                        // (read at every call: the tests switch it)
	const bool fused_gen_on = fge && atoi(fge) != 0;
	const bool fused_gen = gen && fused_gen_on;
	const bool fused = (!isdht && !count) || fused_gen;
	size_t want = compress_chunk(n);
	size_t nchunks = fused ? 1 : (n + want - 1) / want;
	size_t chunk = (n + nchunks - 1) / nchunks;
	nxz_ctx::Scratch sc;
	// Calls on ONE stream share that stream's scratch (tokens, tables): their launches must not interleave, and a call
	// that needs more room must not free what another has just handed to its kernels (round 2's advisor finding).  One
	// call at a time per stream from sizing to the last launch; the stream's order does the rest.
	std::mutex *use_mtx;
	{
		std::lock_guard<std::mutex> g(c->mtx);
		use_mtx = &c->scratch_use[s];
	}
	std::lock_guard<std::mutex> use(*use_mtx);
	{
		std::lock_guard<std::mutex> g(c->mtx);
		nxz_ctx::Scratch &r = c->scratch[s];
		if (!fused && r.chunk_limit && want > r.chunk_limit) {
			// a larger chunk could not be had on this device a call ago: not tried again before nxz_trim()
			want = r.chunk_limit; nchunks = (n + want - 1) / want; chunk = (n + nchunks - 1) / nchunks;
		}
		if (!fused && r.chunk_cap < chunk) {
			// grows only: warm up once with the largest batch before timing a loop.  The scratch of a chunk (104 KiB of
			// tokens + a table + the counts per job) takes no more than a quarter of what the device has free right now.
			if (r.d_tokens) { (void)hipStreamSynchronize(s); r.release_chunk(); }
			const size_t per_job = (size_t)NXZ_TOK_STRIDE + sizeof(nxz_dht_prepared_t) + 316 * sizeof(uint32_t);
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > 8192 && chunk * per_job > free_b / 4) {
				while (want > 8192 && want * per_job > free_b / 4) want /= 2;
				nchunks = (n + want - 1) / want; chunk = (n + nchunks - 1) / nchunks;
				r.chunk_limit = want;
			}
			while (!r.alloc_chunk(chunk)) {
				if (want <= 1024) return -ENOMEM;
				want = want > 8192 ? 8192 : want / 2;          // no room for the large chunk: the small one, then halves of it
				nchunks = (n + want - 1) / want; chunk = (n + nchunks - 1) / nchunks;
				r.chunk_limit = want;
			}
		}
		if (!r.d_cand2) HIPCHK(hipMalloc((void **)&r.d_cand2, nxz_lz77_cand2_bytes()), return -ENOMEM);
		if (fused_gen && !r.d_fuse) HIPCHK(hipMalloc((void **)&r.d_fuse, nxz_lz77_gen_scratch_bytes()), return -ENOMEM);
		if (isdht && !gen && r.prepared_cap < ntables) {
			if (r.d_prepared) { (void)hipStreamSynchronize(s); (void)hipFree(r.d_prepared); }
			r.d_prepared = nullptr; r.prepared_cap = 0;
			HIPCHK(hipMalloc((void **)&r.d_prepared, ntables * sizeof(nxz_dht_prepared_t)), return -ENOMEM);
			r.prepared_cap = ntables;
		}
		if (!c->d_job_counters && hipMalloc((void **)&c->d_job_counters, JOB_COUNTERS * sizeof(uint32_t)) != hipSuccess) c->d_job_counters = nullptr;
		sc = r;
	}
	if (isdht && !gen) {
		prepared = sc.d_prepared;
		int rc = nxz_launch_dht_prepare(dht, ntables, prepared, s);
		if (rc) { set_err("dht prepare launch", (hipError_t)rc); return -EIO; }
	}
	for (size_t off = 0; off < n; off += chunk) {
		const size_t m = n - off < chunk ? n - off : chunk;
		uint32_t *jc = nullptr;
		{
			std::lock_guard<std::mutex> g(c->mtx);
			if (c->d_job_counters) jc = c->d_job_counters + (c->next_counter++ % JOB_COUNTERS);
		}
		uint32_t *cnt = count ? counts + off * 316 : gen ? sc.d_counts : nullptr;
		auto stamp = [&]() {
			if (!c->timing) return;
			hipEvent_t e;
			if (hipEventCreate(&e) != hipSuccess) return;
			(void)hipEventRecord(e, s);
			std::lock_guard<std::mutex> g(c->mtx);
			c->tev.push_back(e);
		};
		stamp();
		int rc = fused_gen ? nxz_launch_lz77(NXZ_LZ77_FUSED_GEN, jobs + off, m, sc.d_fuse, sc.d_cand2, results + off, count ? counts + off * 316 : nullptr, jc, s)
				   : nxz_launch_lz77(fused ? NXZ_LZ77_FUSED_FHT : cnt != nullptr, jobs + off, m, sc.d_tokens, sc.d_cand2, results + off, cnt, jc, s);
		if (rc) { set_err("lz77 launch", (hipError_t)rc); return -EIO; }
		stamp();
		if (fused) { stamp(); stamp(); continue; }
		if (gen) {
			rc = nxz_launch_dhtgen(cnt, m, sc.d_gen, nullptr, s);
			if (rc) { set_err("dhtgen launch", (hipError_t)rc); return -EIO; }
		}
		stamp();
		rc = nxz_launch_encode(isdht, gen, jobs + off, m, sc.d_tokens, gen ? sc.d_gen : prepared, results + off, s);
		if (rc) { set_err("encode launch", (hipError_t)rc); return -EIO; }
		stamp();
	}
	return 0;
}

// nxz_trim(): the token scratch of every stream no batch call is working on goes back to the device (a chunk of 65536 jobs
// is 6.6 GiB), and a remembered "no room for more than N jobs a chunk" is forgotten.  Returns the bytes freed.
static size_t trim_compress_scratch()
{
	size_t freed = 0;
	std::lock_guard<std::mutex> g(g_mtx);
	for (nxz_ctx *c : g_ctx) {
		if (!c) continue;
		(void)hipSetDevice(c->device);
		std::vector<std::pair<hipStream_t, std::mutex *>> streams;
		{
			std::lock_guard<std::mutex> g2(c->mtx);
			for (auto &kv : c->scratch) streams.emplace_back(kv.first, &c->scratch_use[kv.first]);
		}
		for (auto &sm : streams) {
			if (!sm.second->try_lock()) continue;              // a call is sizing or launching on that stream
			// (a stream the caller has destroyed meanwhile -- nxz_stream_destroy drops its entry, a stream of the caller's own may be gone
			// without a word: a failed wait means "leave it alone")
			bool there;
			{
				std::lock_guard<std::mutex> g2(c->mtx);
				there = c->scratch.find(sm.first) != c->scratch.end();
			}
			if (there && hipStreamSynchronize(sm.first) != hipSuccess) { (void)hipGetLastError(); there = false; }
			if (there) {
				std::lock_guard<std::mutex> g2(c->mtx);
				auto it = c->scratch.find(sm.first);               // (find, not []: an entry that went away in between stays away)
				if (it != c->scratch.end()) {
					nxz_ctx::Scratch &r = it->second;
					if (r.d_tokens) freed += r.chunk_cap * ((size_t)NXZ_TOK_STRIDE + sizeof(nxz_dht_prepared_t) + 316 * sizeof(uint32_t));
					r.release_chunk();
					r.chunk_limit = 0;
				}
			}
			sm.second->unlock();
		}
	}
	return freed;
}

// Measurement aid: with timing on, every compress batch records events around its kernels;
// nxz_ctx_stage_ms waits for them and returns the milliseconds spent in the LZ77, dhtgen and
// entropy kernels since the last call (and the number of launches of each).
extern "C" void nxz_ctx_stage_timing(nxz_ctx_t *c, int on)
{
	if (!c) return;
	std::lock_guard<std::mutex> g(c->mtx);
	c->timing = on != 0;
}

extern "C" int nxz_ctx_stage_ms(nxz_ctx_t *c, double ms[3], unsigned *launches)
{
	if (!c || !ms) return -EINVAL;
	std::vector<hipEvent_t> ev;
	{
		std::lock_guard<std::mutex> g(c->mtx);
		ev.swap(c->tev);
	}
	ms[0] = ms[1] = ms[2] = 0;
	if (launches) *launches = (unsigned)(ev.size() / 4);
	for (size_t i = 0; i + 3 < ev.size(); i += 4) {
		(void)hipEventSynchronize(ev[i + 3]);
		for (int k = 0; k < 3; k++) {
			float f = 0;
			if (hipEventElapsedTime(&f, ev[i + k], ev[i + k + 1]) == hipSuccess) ms[k] += f;
		}
	}
	for (auto e : ev) (void)hipEventDestroy(e);
	return 0;
}

// The reference's dhtgen() (lib/nx_dhtgen.c:945-1034) for a batch of count arrays on the device.
extern "C" int nxz_batch_dhtgen(nxz_ctx_t *c, const uint32_t *counts, size_t n, nxz_batch_dht_t *tables, void *stream)
{
	if (!c || !counts || !tables) return -EINVAL;
	if (forked_child()) return -ENODEV;
	(void)hipSetDevice(c->device);
	int rc = nxz_launch_dhtgen(counts, n, nullptr, tables, (hipStream_t)stream);
	if (rc) { set_err("dhtgen launch", (hipError_t)rc); return -EIO; }
	return 0;
}

// force: 0 -- the kernel by the batch's size and kind; 1 -- a stream per lane, any block type; 2 -- a stream per wavefront
static int batch_decompress(nxz_ctx_t *c, const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io, void *stream, int force);
static hipError_t stream_create_spread(hipStream_t *s, unsigned turn);
extern "C" int nxz_batch_decompress(nxz_ctx_t *c, const nxz_batch_job_t *jobs, size_t n,
				    nxz_batch_result_t *results, nxz_batch_dht_t *dht_io, void *stream)
{
	return batch_decompress(c, jobs, n, results, dht_io, stream, 0);
}
static int batch_decompress(nxz_ctx_t *c, const nxz_batch_job_t *jobs, size_t n, nxz_batch_result_t *results, nxz_batch_dht_t *dht_io, void *stream, int force)
{
	if (!c) return -EINVAL;
	if (forked_child()) return -ENODEV;
	(void)hipSetDevice(c->device);         // scratch is allocated on, and kernels go to, the context's device
	hipStream_t s = (hipStream_t)stream;   // NULL = the HIP default stream
	int rc;
	const char *lm = getenv("NXZ_INFLATE_LANES_MIN");                    // tuning / test knob
	const size_t lanes_min = lm ? (size_t)strtoull(lm, nullptr, 0) : (size_t)NXZ_LANES_MIN;
	bool lanes = force ? (force & 3) == 1 : n >= lanes_min, by_len = (force & 4) != 0, no_tables = false;
	bool split = false;
	// A stream per WORKGROUP, source, output and tables in LDS (nxz_inflate_wg.hip): every batch, unless one of the older routes' knobs
	// is set (the tests' way to name a route) -- except, from 98 304 streams on, the batches whose sampled streams begin with fixed-code
	// or stored blocks (the fixed-code lane kernel's: 158-177 GiB/s against 151).  That kernel runs at one rate from a few thousand
	// streams on (a CU a stream; profiles/r06_inflate_by_batch_size.txt: zlib -6 streams of the corpus 96-99 GiB/s from 4096 streams on,
	// own exact-table streams 115-120, fixed-code synthetic blocks 144-151), where a stream per wavefront needs 16 384 streams for 56
	// and levels off at 66, and a stream per lane needs 100 000 (zlib -6 streams at 262 144, both older kernels side by side: 87).
	// Streams of any length are its own (in spans, the output flushed in halves); what it does not do -- streams that resume or bring
	// a history, end early or are damaged -- it hands back, and those go a stream per wavefront behind it.
	// NXZ_INFLATE_WG=0 / 1: never / always; NXZ_INFLATE_WG_MAX: batches up to that size only.
	const char *wge = getenv("NXZ_INFLATE_WG");                         // (read at every call: the tests switch it)
	const char *wgm = getenv("NXZ_INFLATE_WG_MAX");
	const size_t wg_max = wgm ? (size_t)strtoull(wgm, nullptr, 0) : ~(size_t)0;
	bool wg = !force && (wge ? atoi(wge) != 0 : (!lm && !getenv("NXZ_INFLATE_CUT") && n <= wg_max));
	if (wg && !wge && n >= 98304) {
		// (the sample the older routes take below: here only "do these streams bring tables?")
		uint32_t *h = nullptr;
		{
			std::lock_guard<std::mutex> g(c->mtx);
			if (!c->h_sample) (void)hipHostMalloc((void **)&c->h_sample, 64 * sizeof(uint32_t));
			h = c->h_sample ? c->h_sample + 4 * (c->sample_turn++ & 15) : nullptr;
		}
		if (h) {
			h[0] = 0; h[1] = 0; h[2] = 0;
			if (nxz_launch_sample_btype(jobs, n, h, s) == 0 && hipStreamSynchronize(s) == hipSuccess && h[0] <= 16) wg = false;
		}
	}
	if (wg) {
		std::mutex *use_mtx;
		{
			std::lock_guard<std::mutex> g(c->mtx);
			use_mtx = &c->scratch_use[s];
		}
		std::lock_guard<std::mutex> use(*use_mtx);                     // (one call at a time per stream's scratch)
		uint8_t *wws = nullptr, *ows = nullptr;
		{
			std::lock_guard<std::mutex> g(c->mtx);
			nxz_ctx::Scratch &sc = c->scratch[s];
			const size_t need = nxz_inflate_wg_workspace(n), oneed = n >= 128 ? nxz_order_workspace(n) : 0;
			if (sc.wg_cap < need) {
				if (sc.d_wg_ws) { (void)hipStreamSynchronize(s); (void)hipFree(sc.d_wg_ws); }
				sc.d_wg_ws = nullptr; sc.wg_cap = 0;
				HIPCHK(hipMalloc((void **)&sc.d_wg_ws, need), return -ENOMEM);
				sc.wg_cap = need;
			}
			if (sc.order_cap < oneed) {
				if (sc.d_order_ws) { (void)hipStreamSynchronize(s); (void)hipFree(sc.d_order_ws); }
				sc.d_order_ws = nullptr; sc.order_cap = 0;
				if (hipMalloc((void **)&sc.d_order_ws, oneed) == hipSuccess) sc.order_cap = oneed; else (void)hipGetLastError();
			}
			wws = sc.d_wg_ws;
			ows = oneed && sc.order_cap >= oneed ? sc.d_order_ws : nullptr;
		}
		const uint32_t *order = ows ? nxz_launch_order_by_length(jobs, n, ows, s) : nullptr;   // (a workgroup draws stream after stream: the long ones first)
		rc = nxz_launch_inflate_wg(jobs, n, results, dht_io, wws, order, nullptr, s);
		if (rc) { set_err("inflate launch", (hipError_t)rc); return -EIO; }
		return 0;
	}
	if (lanes && !lm && !force) {
		// what kind of streams?  (one small launch and a wait for it: nothing next to the tens of milliseconds such a batch takes)
		uint32_t *h = nullptr;
		{
			std::lock_guard<std::mutex> g(c->mtx);
			if (!c->h_sample) (void)hipHostMalloc((void **)&c->h_sample, 64 * sizeof(uint32_t));
			h = c->h_sample ? c->h_sample + 4 * (c->sample_turn++ & 15) : nullptr;
		}
		if (h) {
			h[0] = 0; h[1] = 0; h[2] = 0;
			if (nxz_launch_sample_btype(jobs, n, h, s) == 0 && hipStreamSynchronize(s) == hipSuccess) {
				// a quarter or more with tables: the wave kernel's, unless the batch is so large that the general lane kernel
				// overtakes it (zlib -6 streams of the corpus: 75 against 86 GiB/s at 131 072 streams, 95 against 86 at 196 608, 102 at
				// 262 144, 117 at 524 288; profiles/r04c_inflate_by_batch_size.txt)
				if (h[0] > 64 && n < NXZ_LANES_TABLES_MIN) lanes = false;
				// ... and from there on BOTH, side by side on two HIP streams, each on its share of the batch (NXZ_INFLATE_SPLIT_PCT: the
				// wavefront kernel's share, 40; 0: the lane kernel alone, as up to round 5): the lane kernel waits for memory three quarters
				// of its time, the wavefront kernel is bound by what it issues -- 262 548 zlib -6 streams of the corpus 81.6 -> 84.4 GiB/s,
				// of the round-4 classes 97.4 -> 110
				else if (h[0] > 64 && n >= NXZ_LANES_TABLES_MIN) split = true;
				// streams of very different lengths (zeros beside text: BASELINE configs[4]): a wavefront takes as long as its
				// longest stream, so the lane kernel gets them ordered by length; much of a size they stay as they come
				// (neighbours in memory: ordering the bench's synthetic blocks cost 5 %)
				by_len = h[2] > 8 * (uint64_t)h[1] + 4096;
				// few of the sampled streams begin with a dynamic block: the fixed-code-only lane kernel first, which hands the
				// streams it cannot do -- those, and any with a dynamic block further in -- to the general one, stream by stream
				no_tables = h[0] <= 16;
			}
		}
	}
	static const int split_pct = getenv("NXZ_INFLATE_SPLIT_PCT") ? atoi(getenv("NXZ_INFLATE_SPLIT_PCT")) : 40;
	if (split && split_pct > 0 && split_pct < 100) {
		{
			std::lock_guard<std::mutex> g(c->mtx);
			if (!c->split_stream) {
				if (stream_create_spread(&c->split_stream, 1) != hipSuccess || hipEventCreateWithFlags(&c->split_ev[0], hipEventDisableTiming) != hipSuccess ||
				    hipEventCreateWithFlags(&c->split_ev[1], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); c->split_stream = nullptr; }
			}
		}
		if (c->split_stream) {
			std::lock_guard<std::mutex> one(c->split_mtx);              // (one split batch at a time: the second stream and the events are the context's)
			const size_t k = ((n * (size_t)(100 - split_pct) / 100) + 63) & ~(size_t)63;
			if (k > 0 && k < n) {
				HIPCHK(hipEventRecord(c->split_ev[0], s), return -EIO);
				HIPCHK(hipStreamWaitEvent(c->split_stream, c->split_ev[0], 0), return -EIO);
				const int r2 = batch_decompress(c, jobs + k, n - k, results + k, dht_io ? dht_io + k : nullptr, c->split_stream, 2);
				const int r1 = batch_decompress(c, jobs, k, results, dht_io, s, 1 | (by_len ? 4 : 0));
				HIPCHK(hipEventRecord(c->split_ev[1], c->split_stream), return -EIO);
				HIPCHK(hipStreamWaitEvent(s, c->split_ev[1], 0), return -EIO);
				return r1 ? r1 : r2;
			}
		}
	}
	if (lanes) {
		// many streams: one stream per lane (nxz_inflate_lanes.hip); the table workspace is made once
		int init = 0;
		uint8_t *ws;
		std::mutex *use_mtx;                                           // (one call at a time per stream's table workspace, as in nxz_batch_compress)
		{
			std::lock_guard<std::mutex> g(c->mtx);
			use_mtx = &c->scratch_use[s];
		}
		std::lock_guard<std::mutex> use(*use_mtx);
		{
			std::lock_guard<std::mutex> g(c->mtx);
			nxz_ctx::Scratch &sc = c->scratch[s];
			const size_t need = nxz_inflate_lanes_workspace(n);
			if (sc.lanes_cap < need) {
				// grows only (3.6 KiB per lane in flight, 0.9 GiB for the largest grid)
				if (sc.d_lanes_ws) { (void)hipStreamSynchronize(s); (void)hipFree(sc.d_lanes_ws); }
				sc.d_lanes_ws = nullptr; sc.lanes_cap = 0;
				HIPCHK(hipMalloc((void **)&sc.d_lanes_ws, need), return -ENOMEM);
				sc.lanes_cap = need;
				init = 1;
			}
			ws = sc.d_lanes_ws;
		}
		rc = nxz_launch_inflate_lanes(jobs, n, results, dht_io, ws, init | (by_len ? 2 : 0) | (no_tables ? 4 : 0), s);
	} else if ([&]() -> bool {
		// A batch that does not fill the device a stream per wavefront (5120 at a time, each as slow as 20-100 MB/s): every stream
		// is cut inside its first block and the pieces go side by side (nxz_inflate_cut.hip; zlib -6 streams of the corpus, 4096
		// of them: 28.6 GiB/s a stream per wavefront).  NXZ_INFLATE_CUT=0 / 1: never / whenever two pieces a stream are allowed.
		const char *ce = getenv("NXZ_INFLATE_CUT");
		const int cut_env = ce ? atoi(ce) : -1;
		if (cut_env == 0) return false;
		unsigned P = nxz_inflate_cut_pieces(n);
		// (left to itself: batches of 64 streams at most, where a call takes as long as its slowest stream and
		// the pieces of all of them are resident at once; larger ones lose more to the rounds -- each as long as ITS
		// slowest piece -- than the cuts win: profiles/r05_inflate_cut_by_batch_size.txt)
		static const size_t auto_max = getenv("NXZ_INFLATE_CUT_MAX") ? (size_t)strtoull(getenv("NXZ_INFLATE_CUT_MAX"), nullptr, 0) : 64;
		if (cut_env < 0 && (P < 4 || n > auto_max)) return false;
		if (P < 2) { if (cut_env <= 0) return false; P = 2; }
		// room for the pieces' 16-bit elements: half a megabyte a stream, a quarter of what the device has free at most
		size_t arena = n * ((size_t)512 << 10), free_b = 0, total_b = 0;
		if (arena < ((size_t)256 << 20)) arena = (size_t)256 << 20;
		if (arena > ((size_t)8 << 30)) arena = (size_t)8 << 30;
		std::mutex *use_mtx;
		{
			std::lock_guard<std::mutex> g(c->mtx);
			use_mtx = &c->scratch_use[s];
		}
		std::lock_guard<std::mutex> use(*use_mtx);                         // (one call at a time per stream's scratch)
		uint8_t *ws = nullptr;
		{
			std::lock_guard<std::mutex> g(c->mtx);
			nxz_ctx::Scratch &sc = c->scratch[s];
			size_t need = nxz_inflate_cut_workspace(n, P, arena);
			if (sc.cut_cap < need) {
				if (sc.d_cut_ws) { (void)hipStreamSynchronize(s); (void)hipFree(sc.d_cut_ws); }
				sc.d_cut_ws = nullptr; sc.cut_cap = 0;
				if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && arena > free_b / 4) { arena = free_b / 4; need = nxz_inflate_cut_workspace(n, P, arena); }
				if (arena < ((size_t)16 << 20) || hipMalloc((void **)&sc.d_cut_ws, need) != hipSuccess) { (void)hipGetLastError(); return false; }
				sc.cut_cap = need;
			} else arena += sc.cut_cap - need;                             // (what a larger batch left: the arena takes it)
			ws = sc.d_cut_ws;
		}
		rc = nxz_launch_inflate_cut(jobs, n, results, dht_io, P, ws, arena, s);
		return true;
	}()) {
	} else {
		const char *wm = getenv("NXZ_INFLATE_LDS_MAX");                 // tuning / test knob
		const size_t lds_max = wm ? (size_t)strtoull(wm, nullptr, 0) : (size_t)NXZ_WINDOW_LDS_MAX;
		// A launch ends with its slowest stream, and the corpus' slowest block takes a wavefront 8 ms where the average takes 4:
		// the long ones start first -- the jobs' indices by falling source length (zlib -6 streams of the corpus: 53.9 -> 77.1
		// GiB/s at 16 384 streams, 67.4 -> 85.0 at 32 768, 83.1 -> 86.5 at 262 144, 26.1 -> 28.5 at 4096 where all are resident
		// at once; profiles/r04c_inflate_by_batch_size.txt).  Not for the few streams that get the window in LDS.
		// (NXZ_INFLATE_ORDER=0 / 1: never / always)
		const uint32_t *order = nullptr;
		const char *oe = getenv("NXZ_INFLATE_ORDER");
		const int order_env = oe ? atoi(oe) : -1;
		if (order_env < 0 ? n > lds_max : order_env != 0) {
			std::mutex *use_mtx;
			{
				std::lock_guard<std::mutex> g(c->mtx);
				use_mtx = &c->scratch_use[s];
			}
			std::lock_guard<std::mutex> use(*use_mtx);                     // (one call at a time per stream's scratch: the kernel reads the order)
			uint8_t *ows = nullptr;
			{
				std::lock_guard<std::mutex> g(c->mtx);
				nxz_ctx::Scratch &sc = c->scratch[s];
				const size_t need = nxz_order_workspace(n);
				if (sc.order_cap < need) {
					if (sc.d_order_ws) { (void)hipStreamSynchronize(s); (void)hipFree(sc.d_order_ws); }
					sc.d_order_ws = nullptr; sc.order_cap = 0;
					if (hipMalloc((void **)&sc.d_order_ws, need) == hipSuccess) sc.order_cap = need;
				}
				ows = sc.d_order_ws;
			}
			order = nxz_launch_order_by_length(jobs, n, ows, s);              // (NULL: in the caller's order)
			rc = nxz_launch_inflate(jobs, n, results, dht_io, n <= lds_max, order, s);
		} else rc = nxz_launch_inflate(jobs, n, results, dht_io, n <= lds_max, nullptr, s);
	}
	if (rc) { set_err("inflate launch", (hipError_t)rc); return -EIO; }
	return 0;
}

// (diagnostic / tests: how many streams of the last batch of n that `stream` ran through the lane kernels the fixed-code-only
// kernel handed back to the general one; waits for the stream)
extern "C" int nxz_inflate_lanes_handed_back(const uint8_t *workspace, size_t n, uint32_t *count);
extern "C" int nxz_ctx_lanes_handed_back(nxz_ctx_t *c, void *stream, size_t n, uint32_t *count)
{
	if (!c || !count) return -EINVAL;
	(void)hipSetDevice(c->device);
	hipStream_t s = (hipStream_t)stream;
	if (hipStreamSynchronize(s) != hipSuccess) return -EIO;
	const uint8_t *ws = nullptr;
	{
		std::lock_guard<std::mutex> g(c->mtx);
		auto it = c->scratch.find(s);
		if (it != c->scratch.end()) ws = it->second.d_lanes_ws;
	}
	if (!ws) return -ENOENT;
	return nxz_inflate_lanes_handed_back(ws, n, count) ? -EIO : 0;
}

// (diagnostic / tests: why the workgroup-per-stream kernel handed streams of the last batch on `stream` back: out16[1..14] by reason
// (nxz_inflate_wg.hip R_*), out16[15] the streams handed back; waits for the stream)
extern "C" int nxz_ctx_wg_reasons(nxz_ctx_t *c, void *stream, uint32_t *out16)
{
	if (!c || !out16) return -EINVAL;
	(void)hipSetDevice(c->device);
	hipStream_t s = (hipStream_t)stream;
	if (hipStreamSynchronize(s) != hipSuccess) return -EIO;
	const uint8_t *ws = nullptr;
	{
		std::lock_guard<std::mutex> g(c->mtx);
		auto it = c->scratch.find(s);
		if (it != c->scratch.end()) ws = it->second.d_wg_ws;
	}
	if (!ws) return -ENOENT;
	return nxz_inflate_wg_reasons(ws, out16) ? -EIO : 0;
}
extern "C" int nxz_ctx_wg_prof(nxz_ctx_t *c, void *stream, unsigned long long *out12)
{
	if (!c || !out12) return -EINVAL;
	(void)hipSetDevice(c->device);
	hipStream_t s = (hipStream_t)stream;
	if (hipStreamSynchronize(s) != hipSuccess) return -EIO;
	const uint8_t *ws = nullptr;
	{
		std::lock_guard<std::mutex> g(c->mtx);
		auto it = c->scratch.find(s);
		if (it != c->scratch.end()) ws = it->second.d_wg_ws;
	}
	if (!ws) return -ENOENT;
	return nxz_inflate_wg_prof(ws, out12) ? -EIO : 0;
}

extern "C" int nxz_batch_wrap(nxz_ctx_t *c, const nxz_batch_job_t *jobs, size_t n,
			      nxz_batch_result_t *results, void *stream)
{
	if (!c) return -EINVAL;
	(void)hipSetDevice(c->device);
	hipStream_t s = (hipStream_t)stream;   // NULL = the HIP default stream
	static const bool old_wrap = getenv("NXZ_WRAP_OLD") && atoi(getenv("NXZ_WRAP_OLD")) != 0;
	int rc = old_wrap ? nxz_launch_wrap(jobs, n, results, s) : nxz_launch_wrap_sliced(jobs, n, results, s);
	if (rc) { set_err("wrap launch", (hipError_t)rc); return -EIO; }
	return 0;
}


// Gzip members from the results of a compress batch (nxz_misc.hip): offsets[n + 1] and `packed`
// are device memory; offsets[n] is the number of bytes written to `packed`.
extern "C" int nxz_batch_pack_gzip(nxz_ctx_t *c, const nxz_batch_job_t *jobs, const nxz_batch_result_t *results, size_t n,
				   uint64_t *offsets, uint8_t *packed, void *stream)
{
	if (!c || !jobs || !results || !offsets || !packed || n > 0xffffffffu) return -EINVAL;
	(void)hipSetDevice(c->device);
	int rc = nxz_launch_pack_members(jobs, results, n, offsets, packed, (hipStream_t)stream);
	if (rc) { set_err("pack launch", (hipError_t)rc); return -EIO; }
	return 0;
}

// ---------------------------------------------------------------------------
// nxz_deflate_host: a long HOST buffer -> one raw deflate stream in a HOST buffer
// ---------------------------------------------------------------------------
#define HOST_GROUP 256u                    /* blocks per group: 16 MiB in, one launch of each kernel */
#define STAGE_MAX_BLOCKS 64u               /* groups up to this many blocks go through the lane's pinned staging */
#define HOST_SLOT 73856u                   /* room for one block's output (nxz_compress_bound(65536) rounded) */

static uint32_t gf2_mul32(uint32_t a, uint32_t b)
{
	uint32_t p = 0;
	for (uint32_t m = 0x80000000u; m; m >>= 1) {
		if (a & m) p ^= b;
		b = (b >> 1) ^ ((b & 1) ? 0xedb88320u : 0);
	}
	return p;
}
static uint32_t crc_shift_op(uint64_t nbytes)                       // x^(8 nbytes) mod P, reflected
{
	uint32_t r = 0x80000000u, sq = 0x00800000u;
	for (uint64_t n = nbytes; n; n >>= 1) { if (n & 1) r = gf2_mul32(r, sq); sq = gf2_mul32(sq, sq); }
	return r;
}
static uint32_t adler_join(uint32_t a1, uint32_t a2, uint64_t len2)
{
	const uint64_t B = 65521, rem = len2 % B, s1 = a1 & 0xffff;
	const uint64_t sum1 = (s1 + (a2 & 0xffff) + B - 1) % B;
	const uint64_t sum2 = (rem * s1 + (a1 >> 16) + (a2 >> 16) + B - rem) % B;
	return (uint32_t)((sum2 << 16) | sum1);
}

// the source bytes a block takes when hist_max bytes of what lies in front of it are its window
// (window + block <= 64 KiB, both multiples of 16)
static inline size_t host_block_bytes(uint32_t hist_max)
{
	const uint32_t h = hist_max > 32768u ? 32768u : hist_max & ~15u;
	return SUBBLOCK - h;
}
extern "C" size_t nxz_deflate_host_bound_hist(size_t src_len, uint32_t hist_max)
{
	const size_t B = host_block_bytes(hist_max);
	return src_len + ((src_len + B - 1) / B) * 10 + 16;
}
extern "C" size_t nxz_deflate_host_bound(size_t src_len) { return nxz_deflate_host_bound_hist(src_len, 0); }

// (the two lanes of a pair get streams of different priority: the runtime maps streams onto a few hardware
// queues, and two streams of one priority may share a queue, depending on what other streams the process has
// made before -- then the copies of one lane and the kernels of the other run one after the other)
// a non-blocking stream whose priority goes round (low, normal, high) with `turn`: streams that are to run side by
// side -- rounds, the callers' streams of nxz_stream_create -- spread over the hardware queues of all three levels
static hipError_t stream_create_spread(hipStream_t *s, unsigned turn)
{
	static const bool spread = !(getenv("NXZ_STREAM_PRIORITIES") && atoi(getenv("NXZ_STREAM_PRIORITIES")) == 0);
	int least = 0, greatest = 0;
	(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
	const int prio = !spread ? 0 : turn % 3 == 0 ? 0 : turn % 3 == 1 ? greatest : least;
	return hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio);
}

// a lane's stream (made once) and its buffers for `blocks` blocks per group (grow only, a power of two from 32 up to
// HOST_GROUP: a caller of megabyte-sized calls holds 2 x 6 MiB, not 2 x 50 -- with 16 pairs of lanes that is what a
// process of many threads pays when each makes its first call)
static bool lane_need(nxz_ctx::HostLane &l, bool high, size_t blocks)
{
	if (!l.stream) {
		int least = 0, greatest = 0;
		(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
		HIPCHK(hipStreamCreateWithPriority(&l.stream, hipStreamNonBlocking, high ? greatest : least), return false);
	}
	if (blocks <= l.cap) return true;
	size_t cap = 32;
	while (cap < blocks) cap <<= 1;
	if (cap > HOST_GROUP) cap = HOST_GROUP;
	if (l.cap) {
		(void)hipStreamSynchronize(l.stream);
		(void)hipFree(l.d_base); (void)hipHostFree(l.h_base);
		l.d_base = l.h_base = nullptr;
		l.d_src = l.d_dst = l.d_packed = nullptr; l.d_jobs = l.h_jobs = nullptr; l.d_res = l.h_res = nullptr; l.d_off = nullptr; l.h_total = nullptr;
		l.h_src = l.h_packed = nullptr;
		l.cap = 0;
	}
	// one allocation on either side (round 4 made nine: with sixteen threads at their first call 30 streams and some 400
	// allocations went through the runtime's lock one after the other -- 480 ms before the first call came back)
	auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
	const size_t o_src = 0, o_dst = o_src + up(cap * SUBBLOCK), o_packed = o_dst + up(cap * HOST_SLOT), o_jobs = o_packed + up(cap * (SUBBLOCK + 16)),
		     o_res = o_jobs + up(cap * sizeof(nxz_batch_job_t)), o_off = o_res + up(cap * sizeof(nxz_batch_result_t)), d_total = o_off + up((cap + 1) * sizeof(uint64_t));
	// Calls of a few MiB from many threads: the caller's pages are not pinned, and a copy straight from them makes the
	// runtime pin and unpin them per call under the process's memory-map lock -- sixteen threads of 1 MiB calls ran at
	// 4 GiB/s that way.  Up to STAGE_MAX_BLOCKS per group the lane has pinned staging of its own: the calling thread
	// copies in and out of it (its own core's time), the DMA runs from pinned memory.
	static const bool stage_on = !(getenv("NXZ_HOST_STAGE") && atoi(getenv("NXZ_HOST_STAGE")) == 0);
	const bool stage = stage_on && cap <= STAGE_MAX_BLOCKS;
	const size_t p_jobs = 0, p_res = p_jobs + up(cap * sizeof(nxz_batch_job_t)), p_total = p_res + up(cap * sizeof(nxz_batch_result_t)),
		     p_src = p_total + 256, p_packed = p_src + (stage ? up(cap * SUBBLOCK) : 0), h_total_bytes = p_packed + (stage ? up(cap * (SUBBLOCK + 16)) : 0);
	HIPCHK(hipMalloc((void **)&l.d_base, d_total), return false);
	HIPCHK(hipHostMalloc((void **)&l.h_base, h_total_bytes), { (void)hipFree(l.d_base); l.d_base = nullptr; return false; });
	l.d_src = l.d_base + o_src; l.d_dst = l.d_base + o_dst; l.d_packed = l.d_base + o_packed;
	l.d_jobs = (nxz_batch_job_t *)(l.d_base + o_jobs); l.d_res = (nxz_batch_result_t *)(l.d_base + o_res); l.d_off = (uint64_t *)(l.d_base + o_off);
	l.h_jobs = (nxz_batch_job_t *)(l.h_base + p_jobs); l.h_res = (nxz_batch_result_t *)(l.h_base + p_res); l.h_total = (uint64_t *)(l.h_base + p_total);
	l.h_src = stage ? l.h_base + p_src : nullptr; l.h_packed = stage ? l.h_base + p_packed : nullptr;
	l.cap = cap;
	return true;
}


// ---- calls of a few MiB from many threads: one batch for the callers that are there together ------------------
// A caller takes room for its blocks in the merge that is open (same function code and window), copies its source into
// the merge's pinned staging and writes its job records -- every caller on its own core, side by side -- and waits.  The
// merge goes out when all who took room have filled it and fewer than NXZ_MERGE_RUNNING (2) merges are in flight (so the
// first caller goes alone at once and those who come while it is in flight go together, as in round_submit): whoever
// sees that first queues one copy of the staging, nxz_batch_compress over all blocks and the two kernels that pack every
// member's blocks as that member's stream straight into pinned memory, waits for the stream, and wakes the rest.  Each
// caller then joins its blocks' checksums and copies its stream out.  Returns -EAGAIN when the merges cannot be set up
// (the caller's own pair of lanes takes the call).
#define MERGE_CAP 512u                     /* block slots of a merge: 32 MiB in */
#define MERGE_MEMBERS 64u
#define MERGE_OUT_STRIDE (SUBBLOCK + 32)   /* room per block in the packed staging: nxz_deflate_host_bound of a member fits its blocks' room */
static uint32_t merge_max_blocks()
{
	const char *e = getenv("NXZ_MERGE_MAX_BLOCKS");                 // (read at every call: the tests switch it; 0: never)
	const long x = e ? atol(e) : 128;
	return (uint32_t)(x < 0 ? 0 : x > 256 ? 256 : x);
}
static bool merge_init(nxz_ctx *c, nxz_ctx::Merge &m)
{
	if (m.ready) return true;
	static std::atomic<unsigned> turn{0};
	auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
	const size_t p_src = 0, p_packed = p_src + up((size_t)MERGE_CAP * SUBBLOCK), p_jobs = p_packed + up((size_t)MERGE_CAP * MERGE_OUT_STRIDE),
		     p_res = p_jobs + up(MERGE_CAP * sizeof(nxz_batch_job_t)), p_off = p_res + up(MERGE_CAP * sizeof(nxz_batch_result_t)),
		     p_mem = p_off + up((MERGE_CAP + MERGE_MEMBERS) * sizeof(uint64_t)), p_of = p_mem + up(MERGE_MEMBERS * sizeof(nxz_pack_member_t)),
		     p_total = p_of + up(MERGE_CAP * sizeof(uint16_t));
	const size_t o_src = 0, o_dst = up((size_t)MERGE_CAP * SUBBLOCK), d_total = o_dst + up((size_t)MERGE_CAP * HOST_SLOT);
	if (!m.stream) { HIPCHK(stream_create_spread(&m.stream, turn.fetch_add(1)), return false); }
	HIPCHK(hipMalloc((void **)&m.d_base, d_total), return false);
	HIPCHK(hipHostMalloc((void **)&m.h_base, p_total), { (void)hipFree(m.d_base); m.d_base = nullptr; return false; });
	m.h_src = m.h_base + p_src; m.h_packed = m.h_base + p_packed;
	m.h_jobs = (nxz_batch_job_t *)(m.h_base + p_jobs); m.h_res = (nxz_batch_result_t *)(m.h_base + p_res); m.h_off = (uint64_t *)(m.h_base + p_off);
	m.h_mem = (nxz_pack_member_t *)(m.h_base + p_mem); m.h_member_of = (uint16_t *)(m.h_base + p_of);
	m.d_src = m.d_base + o_src; m.d_dst = m.d_base + o_dst;
	{
		// the stream's token scratch for a full merge at once (it grows only, and every step up is a free and an allocation)
		std::lock_guard<std::mutex> g(c->mtx);
		nxz_ctx::Scratch &r = c->scratch[m.stream];
		if (r.chunk_cap < MERGE_CAP) { if (r.d_tokens) r.release_chunk(); (void)r.alloc_chunk(MERGE_CAP); }
	}
	m.ready = true;
	return true;
}

static int merged_deflate(nxz_ctx_t *c, int fc, const uint8_t *src, size_t src_len, int final, uint32_t H, size_t B,
			  const uint8_t *prev, size_t prev_len, uint8_t *dst, size_t *out_len, uint32_t *crc, uint32_t *adler)
{
	typedef nxz_ctx::Merge Merge;
	static const unsigned running_max = [] { const char *e = getenv("NXZ_MERGE_RUNNING"); int v = e ? atoi(e) : 2; return (unsigned)(v < 1 ? 1 : v > 3 ? 3 : v); }();
	const uint32_t nblk = (uint32_t)((src_len + B - 1) / B);
	const uint32_t h0 = !H ? 0 : (uint32_t)std::min<size_t>(H, prev_len) & ~15u;     // the window in front of the call's first block
	const uint32_t need = nblk + (h0 ? 1 : 0);                                     // staging slots: [window][blocks]
	std::unique_lock<std::mutex> lk(c->mm);
	Merge *M = nullptr;
	for (;;) {
		Merge *fresh = nullptr;
		for (auto &m : c->merges) {
			if (m.state == Merge::OPEN && m.fc == fc && m.H == H && m.slots + need <= MERGE_CAP && m.members < MERGE_MEMBERS) { M = &m; break; }
			if (m.state == Merge::FREE && !fresh) fresh = &m;
		}
		if (!M && fresh) {
			if (!merge_init(c, *fresh)) return -EAGAIN;
			M = fresh; M->state = Merge::OPEN; M->fc = fc; M->H = H; M->slots = M->jobs = M->members = M->filled = M->left = 0; M->rc = 0;
		}
		if (M) break;
		c->mcv.wait(lk);
	}
	const uint32_t me = M->members++, s0 = M->slots, j0 = M->jobs;
	M->slots += need; M->jobs += nblk; M->left++;
	lk.unlock();

	// my part of the staging, my job records, my line of the member table
	uint8_t *const stage = M->h_src + (size_t)s0 * SUBBLOCK;
	uint8_t *const dsrc = M->d_src + (size_t)s0 * SUBBLOCK;
	if (h0) memcpy(stage, prev + prev_len - h0, h0);
	memcpy(stage + h0, src, src_len);
	for (uint32_t k = 0; k < nblk; k++) {
		nxz_batch_job_t &j = M->h_jobs[j0 + k];
		memset(&j, 0, sizeof(j));
		const uint32_t hk = (uint32_t)std::min<uint64_t>(H, h0 + (uint64_t)k * B);     // (a multiple of 16: h0, B and H are)
		j.src = dsrc + h0 + (size_t)k * B - hk; j.dst = M->d_dst + (size_t)(j0 + k) * HOST_SLOT;
		j.hist_len = hk;
		j.src_len = hk + (uint32_t)std::min<uint64_t>(B, src_len - (uint64_t)k * B);
		j.dst_cap = HOST_SLOT; j.in_crc = 0; j.in_adler = 1;
		M->h_member_of[j0 + k] = (uint16_t)me;
	}
	nxz_pack_member_t &pm = M->h_mem[me];
	pm.b0 = j0; pm.n = nblk; pm.fin = final ? j0 + nblk - 1 : 0xffffffffu; pm.off0 = j0 + me;
	pm.packed = M->h_packed + (size_t)j0 * MERGE_OUT_STRIDE;

	lk.lock();
	M->filled++;
	while (M->state == Merge::OPEN) {
		unsigned running = 0;
		for (auto &m : c->merges) if (m.state == Merge::RUNNING) running++;
		if (M->filled < M->members || running >= running_max) { c->mcv.wait(lk); continue; }
		// it goes out, and I am the one to send it
		M->state = Merge::RUNNING;
		const uint32_t nj = M->jobs, ns = M->slots, nm = M->members;
		lk.unlock();
		int rc = 0;
		(void)hipSetDevice(c->device);
		if (hipMemcpyAsync(M->d_src, M->h_src, (size_t)ns * SUBBLOCK, hipMemcpyHostToDevice, M->stream) != hipSuccess) rc = -EIO;
		if (!rc) rc = nxz_batch_compress(c, M->fc, M->h_jobs, nj, nullptr, 0, M->h_res, nullptr, M->stream);
		if (!rc && nxz_launch_pack_member_streams(M->h_jobs, M->h_res, nj, M->h_mem, nm, M->h_member_of, M->h_off, M->stream)) rc = -EIO;
		if (hipStreamSynchronize(M->stream) != hipSuccess && !rc) rc = -EIO;
		lk.lock();
		M->rc = rc;
		M->state = Merge::DONE;
		c->mcv.notify_all();
	}
	while (M->state != Merge::DONE) c->mcv.wait(lk);
	const int rc = M->rc;
	lk.unlock();

	if (!rc) {
		const uint64_t total = M->h_off[pm.off0 + nblk];
		memcpy(dst, pm.packed, total);
		const uint32_t op_block = crc_shift_op(B);
		uint32_t run_crc = 0, run_adler = 1;
		for (uint32_t k = 0; k < nblk; k++) {
			const nxz_batch_job_t &j = M->h_jobs[j0 + k];
			const uint32_t len = j.src_len - j.hist_len;
			run_crc = gf2_mul32(run_crc, len == B ? op_block : crc_shift_op(len)) ^ M->h_res[j0 + k].crc;
			run_adler = adler_join(run_adler, M->h_res[j0 + k].adler, len);
		}
		*out_len = total;
		if (crc) *crc = run_crc;
		if (adler) *adler = run_adler;
	}
	lk.lock();
	if (--M->left == 0) { M->state = Merge::FREE; c->mcv.notify_all(); }
	lk.unlock();
	return rc;
}

static inline uint64_t trace_ns();
extern "C" int nxz_deflate_host(nxz_ctx_t *c, int fc, const uint8_t *src, size_t src_len, int final,
				uint8_t *dst, size_t dst_cap, size_t *out_len, uint32_t *crc, uint32_t *adler)
{
	return nxz_deflate_host_hist(c, fc, src, src_len, final, 0, nullptr, 0, dst, dst_cap, out_len, crc, adler);
}

// The same with a window: every block sees the hist_max bytes of the INPUT in front of it (the levels that carry
// history from job to job, lib/nx_deflate.c:654-680,845-862: the history of a job is just the bytes in front of
// it, known up front, so the jobs do not depend on each other); the first block's window is the tail of `prev`
// (what the stream kept of earlier calls).  Blocks are 64 KiB - hist_max long (window + block <= 64 KiB).
extern "C" int nxz_deflate_host_hist(nxz_ctx_t *c, int fc, const uint8_t *src, size_t src_len, int final, uint32_t hist_max,
				     const uint8_t *prev, size_t prev_len, uint8_t *dst, size_t dst_cap, size_t *out_len, uint32_t *crc, uint32_t *adler)
{
	if (!c || !src || !dst || !out_len || !src_len) return -EINVAL;
	if (fc != NXZ_FC_COMPRESS_FHT && fc != NXZ_FC_COMPRESS_DHTGEN) return -EINVAL;
	if (forked_child()) return -ENODEV;
	if (dst_cap < nxz_deflate_host_bound_hist(src_len, hist_max)) return -E2BIG;
	const uint32_t H = (uint32_t)(SUBBLOCK - host_block_bytes(hist_max));      // window bytes per block
	const size_t B = SUBBLOCK - H;
	if (H) fc |= 0x08;                                                  // the RESUME forms take hist_len
	if (!prev) prev_len = 0;
	(void)hipSetDevice(c->device);
	// (a call of more than 64 blocks that is alone takes its own pair of lanes, whose groups overlap copies and kernels: 16 MiB
	// on one thread 8.1 against 6.8 GiB/s merged; with others about, merged: sixteen threads of 8 MiB calls 12.4 -> 28-29 GiB/s.
	// Not beyond 128 blocks: two members of 16 MiB fill a merge, 17 GiB/s either way.)
	struct InCall { std::atomic<int> &n; int mine; InCall(std::atomic<int> &a) : n(a), mine(a.fetch_add(1) + 1) {} ~InCall() { n.fetch_sub(1); } } in_call(c->host_callers);
	const size_t nblk_all = (src_len + B - 1) / B;
	const uint32_t mmax = merge_max_blocks();
	if (nblk_all <= mmax && (nblk_all <= 64 || in_call.mine > 1)) {
		const int r = merged_deflate(c, fc, src, src_len, final, H, B, prev, prev_len, dst, out_len, crc, adler);
		if (r != -EAGAIN) return r;
	}
	// A call beyond the merge's limit while other callers are about goes through the merges in slices of that many blocks, one
	// after the other (every block of a stream starts on a byte boundary and sees the input in front of it as its window, so the
	// slices' streams laid end to end ARE the call's stream, byte for byte): sixteen threads of 16 MiB calls on four pairs of
	// lanes of their own ran at half the rate of 8 MiB calls (14 against 30 GiB/s).  NXZ_MERGE_SLICES=0: own lanes as before.
	const char *sle = getenv("NXZ_MERGE_SLICES");                       // (read at every call: the tests switch it)
	if (!(sle && atoi(sle) == 0) && nblk_all > mmax && mmax >= 64 && in_call.mine > 1) {
		const size_t S = (size_t)mmax * B;
		size_t off = 0, pos = 0;
		uint32_t run_crc = 0, run_adler = 1;
		while (off < src_len) {
			const size_t len = std::min<size_t>(S, src_len - off);
			size_t got = 0;
			uint32_t ck = 0, ak = 1;
			const int r = merged_deflate(c, fc, src + off, len, final && off + len == src_len, H, B, off ? src : prev, off ? off : prev_len, dst + pos, &got, &ck, &ak);
			if (r == -EAGAIN && !off) break;                            // (no merges to be had: the lanes below, nothing is done yet)
			if (r) return r == -EAGAIN ? -EIO : r;
			run_crc = gf2_mul32(run_crc, crc_shift_op(len)) ^ ck;
			run_adler = adler_join(run_adler, ak, len);
			pos += got; off += len;
		}
		if (off == src_len) {
			*out_len = pos;
			if (crc) *crc = run_crc;
			if (adler) *adler = run_adler;
			return 0;
		}
	}
	int pair = -1;
	for (int k = 0; k < HOST_PAIRS && pair < 0; k++) if (c->lanes_mtx[k].try_lock()) pair = k;
	if (pair < 0) { pair = (int)(c->lanes_turn.fetch_add(1) % HOST_PAIRS); c->lanes_mtx[pair].lock(); }
	std::lock_guard<std::mutex> g(c->lanes_mtx[pair], std::adopt_lock);
	nxz_ctx::HostLane *const lanes = c->lanes + 2 * pair;
	const size_t nblk = (src_len + B - 1) / B;
	// groups: at least four when the input allows it, so that copies and kernels overlap
	size_t group = std::min<size_t>(HOST_GROUP - 1, std::max<size_t>(31, (nblk + 3) / 4));   // (- 1: the window in front of a group's first block)
	const size_t ngroups = (nblk + group - 1) / group;
	group = (nblk + ngroups - 1) / ngroups;
	for (int k = 0; k < 2; k++)
		if (!lane_need(lanes[k], k == 1, group + 1)) return -ENOMEM;
	const uint32_t op_block = crc_shift_op(B);
	uint32_t run_crc = 0, run_adler = 1;
	size_t pos = 0;
	int rc = 0;

	auto queue = [&](size_t gi) -> int {
		nxz_ctx::HostLane &l = lanes[gi & 1];
		const size_t b0 = gi * group, n = std::min(group, nblk - b0);
		const uint64_t first = (uint64_t)b0 * B;                      // offset of the group's first block in src
		const uint64_t bytes = std::min<uint64_t>((uint64_t)n * B, src_len - first);
		// the window in front of the group: from src itself, for the call's first block from `prev`
		const uint32_t h0 = !H ? 0 : first ? (uint32_t)std::min<uint64_t>(H, first) & ~15u : (uint32_t)std::min<size_t>(H, prev_len) & ~15u;
		for (size_t k = 0; k < n; k++) {
			nxz_batch_job_t &j = l.h_jobs[k];
			memset(&j, 0, sizeof(j));
			const uint32_t hk = (uint32_t)std::min<uint64_t>(H, h0 + (uint64_t)k * B);   // (a multiple of 16: h0, B and H are)
			j.src = l.d_src + h0 + k * B - hk; j.dst = l.d_dst + k * HOST_SLOT;
			j.hist_len = hk;
			j.src_len = hk + (uint32_t)std::min<uint64_t>(B, bytes - (uint64_t)k * B);
			j.dst_cap = HOST_SLOT; j.in_crc = 0; j.in_adler = 1;
		}
		l.n = n; l.bytes = bytes;
		HIPCHK(hipMemcpyAsync(l.d_jobs, l.h_jobs, n * sizeof(nxz_batch_job_t), hipMemcpyHostToDevice, l.stream), return -EIO);
		const uint8_t *from = src + first - (first ? h0 : 0);
		const size_t from_bytes = bytes + (first ? h0 : 0), at = first ? 0 : h0;
		if (l.h_src) {                                                // the window and the blocks through the lane's pinned staging: one copy
			if (at) memcpy(l.h_src, prev + prev_len - h0, h0);
			memcpy(l.h_src + at, from, from_bytes);
			HIPCHK(hipMemcpyAsync(l.d_src, l.h_src, at + from_bytes, hipMemcpyHostToDevice, l.stream), return -EIO);
		} else {
			if (at) { HIPCHK(hipMemcpyAsync(l.d_src, prev + prev_len - h0, h0, hipMemcpyHostToDevice, l.stream), return -EIO); }
			HIPCHK(hipMemcpyAsync(l.d_src + at, from, from_bytes, hipMemcpyHostToDevice, l.stream), return -EIO);
		}
		int r = nxz_batch_compress(c, fc, l.d_jobs, n, nullptr, 0, l.d_res, nullptr, l.stream);
		if (r) return r;
		const uint32_t fin = final && gi == ngroups - 1 ? (uint32_t)(n - 1) : 0xffffffffu;
		if (nxz_launch_pack_stream(l.d_jobs, l.d_res, n, fin, l.d_off, l.d_packed, l.stream)) return -EIO;
		HIPCHK(hipMemcpyAsync(l.h_res, l.d_res, n * sizeof(nxz_batch_result_t), hipMemcpyDeviceToHost, l.stream), return -EIO);
		HIPCHK(hipMemcpyAsync(l.h_total, l.d_off + n, sizeof(uint64_t), hipMemcpyDeviceToHost, l.stream), return -EIO);
		return 0;
	};
	auto collect = [&](size_t gi) -> int {
		nxz_ctx::HostLane &l = lanes[gi & 1];
		HIPCHK(hipStreamSynchronize(l.stream), return -EIO);
		const uint64_t total = *l.h_total;
		if (pos + total > dst_cap) return -E2BIG;                   // cannot happen: the bound was checked
		HIPCHK(hipMemcpyAsync(l.h_packed ? l.h_packed : dst + pos, l.d_packed, total, hipMemcpyDeviceToHost, l.stream), return -EIO);
		for (size_t k = 0; k < l.n; k++) {                           // meanwhile: checksums of the run
			const uint32_t len = l.h_jobs[k].src_len - l.h_jobs[k].hist_len;
			const uint32_t op = len == B ? op_block : crc_shift_op(len);
			run_crc = gf2_mul32(run_crc, op) ^ l.h_res[k].crc;
			run_adler = adler_join(run_adler, l.h_res[k].adler, len);
		}
		HIPCHK(hipStreamSynchronize(l.stream), return -EIO);
		if (l.h_packed) memcpy(dst + pos, l.h_packed, total);
		pos += total;
		return 0;
	};
	size_t queued = 0, done = 0;
	static const bool htrace = getenv("NXZ_API_TRACE") != nullptr;
	uint64_t tq = 0, tc = 0, t_ = 0;
	auto tick = [&]() { return htrace ? trace_ns() : 0; };
	while (!rc && queued < ngroups && queued < 2) { t_ = tick(); rc = queue(queued++); tq += tick() - t_; }
	while (!rc && done < queued) {
		t_ = tick(); rc = collect(done++); tc += tick() - t_;
		if (!rc && queued < ngroups) { t_ = tick(); rc = queue(queued++); tq += tick() - t_; }
	}
	if (htrace && src_len >= (64u << 20))
		fprintf(stderr, "nxz_deflate_host: %zu bytes in %zu groups on lanes %d/%d: %.2f ms queueing (copies in, launches), %.2f ms collecting (waits, copies out)\n",
			src_len, ngroups, 2 * pair, 2 * pair + 1, tq * 1e-6, tc * 1e-6);
	if (rc) {
		for (int k = 0; k < 2; k++) (void)hipStreamSynchronize(lanes[k].stream);
		return rc;
	}
	*out_len = pos;
	if (crc) *crc = run_crc;
	if (adler) *adler = run_adler;
	return 0;
}

// Device memory, pinned host memory, streams and copies for callers that hold HOST buffers and do
// not link the HIP runtime themselves (cgo / JNI / ctypes hosts; libnxz_amd.so's blocked-gzip layer).
extern "C" void *nxz_dev_malloc(nxz_ctx_t *c, size_t bytes)
{
	void *p = nullptr;
	if (!c || hipSetDevice(c->device) != hipSuccess) return nullptr;
	HIPCHK(hipMalloc(&p, bytes ? bytes : 16), return nullptr);
	return p;
}
extern "C" void nxz_dev_free(nxz_ctx_t *c, void *p) { if (c && p) { (void)hipSetDevice(c->device); (void)hipFree(p); } }
extern "C" void *nxz_pinned_malloc(nxz_ctx_t *c, size_t bytes)
{
	void *p = nullptr;
	if (!c || hipSetDevice(c->device) != hipSuccess) return nullptr;
	HIPCHK(hipHostMalloc(&p, bytes ? bytes : 16), return nullptr);
	return p;
}
extern "C" void nxz_pinned_free(nxz_ctx_t *c, void *p) { if (c && p) { (void)hipSetDevice(c->device); (void)hipHostFree(p); } }
extern "C" void *nxz_stream_create(nxz_ctx_t *c)
{
	hipStream_t s = nullptr;
	if (!c || hipSetDevice(c->device) != hipSuccess) return nullptr;
	static std::atomic<unsigned> turn{0};
	HIPCHK(stream_create_spread(&s, turn.fetch_add(1)), return nullptr);
	return (void *)s;
}
extern "C" void nxz_stream_destroy(nxz_ctx_t *c, void *stream)
{
	if (!c || !stream) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize((hipStream_t)stream);
	{
		std::lock_guard<std::mutex> g(c->mtx);
		auto it = c->scratch.find((hipStream_t)stream);
		if (it != c->scratch.end()) {
			it->second.release();
			c->scratch.erase(it);
		}
	}
	(void)hipStreamDestroy((hipStream_t)stream);
}
extern "C" int nxz_copy_to_device(nxz_ctx_t *c, void *dst_dev, const void *src_host, size_t bytes, void *stream)
{
	if (!c) return -EINVAL;
	if (!bytes) return 0;
	HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream), return -EIO);
	return 0;
}
// Measurement aid (bench.py's peak_measured): one device-to-device copy of `bytes` (a multiple of 16, both 16-byte aligned) by a
// 16-bytes-a-lane kernel, asynchronous on `stream`
extern "C" int nxz_copy_device(nxz_ctx_t *c, void *dst_dev, const void *src_dev, size_t bytes, void *stream)
{
	if (!c) return -EINVAL;
	(void)hipSetDevice(c->device);
	return nxz_launch_copy16(src_dev, dst_dev, bytes, (hipStream_t)stream) ? -EIO : 0;
}

extern "C" int nxz_copy_to_host(nxz_ctx_t *c, void *dst_host, const void *src_dev, size_t bytes, void *stream)
{
	if (!c) return -EINVAL;
	if (!bytes) return 0;
	HIPCHK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream), return -EIO);
	return 0;
}

// ---------------------------------------------------------------------------
// the reference's transport symbols
// ---------------------------------------------------------------------------
extern "C" uint64_t tb_freq = 512000000ull;

static uint64_t tb_now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (uint64_t)ts.tv_sec * 512000000ull + (uint64_t)ts.tv_nsec * 512ull / 1000ull;
}

extern "C" uint64_t nx_wait_ticks(uint64_t ticks, uint64_t accumulated, int do_sleep)
{
	uint64_t t0 = tb_now(), t1 = t0;
	if (do_sleep && accumulated > 110000) {
		uint64_t us = accumulated / 512;
		usleep(us > 1000 ? 1000 : (useconds_t)us);
		t1 = tb_now();
	} else {
		while (t1 - t0 <= ticks) t1 = tb_now();
	}
	return accumulated + (t1 - t0);
}

extern "C" int nx_function_begin(int function, int pri, void *handle)
{
	nxz_dev_t *h = (nxz_dev_t *)handle;
	if (function != NXZ_FUNC_COMP_GZIP || !h) { errno = EINVAL; return -1; }
	nxz_ctx_t *c = nxz_ctx_create(pri);
	if (!c) { if (!errno) errno = ENODEV; fprintf(stderr, "nxz: %s\n", g_err); return -1; }
	h->function = function;
	h->paste_addr = c;
	h->fd = c->device + 1;
	return 0;
}

extern "C" int nx_function_end(void *handle)
{
	nxz_dev_t *h = (nxz_dev_t *)handle;
	if (!h) return -1;
	if (h->paste_addr) nxz_ctx_destroy((nxz_ctx_t *)h->paste_addr);
	h->paste_addr = nullptr;
	h->fd = 0;
	return 0;
}

static Slot *slot_acquire(nxz_ctx *c)
{
	std::unique_lock<std::mutex> g(c->mtx);
	for (;;) {
		for (auto &s : c->slots) {
			if (!s.busy) {
				s.busy = true;
				if (!s.stream) {
					g.unlock();
					(void)hipSetDevice(c->device);
					bool ok = slot_init(s);
					if (!ok) slot_free(s);                        // nothing half built stays behind (the next caller starts over)
					g.lock();
					if (!ok) { s.busy = false; return nullptr; }
				}
				return &s;
			}
		}
		c->cv.wait(g);
	}
}

static void slot_release(nxz_ctx *c, Slot *s)
{
	{ std::lock_guard<std::mutex> g(c->mtx); s->busy = false; }
	c->cv.notify_one();
}

// gather `want` bytes starting `skip` bytes into the DDE list
static uint32_t dde_gather(const nxz_dde_t *d, uint32_t skip, uint8_t *dst, uint32_t want)
{
	uint32_t total = nxz_dde_bytes(d), cnt = nxz_dde_count(d), got = 0;
	if (cnt == 0) {
		if (skip >= total) return 0;
		uint32_t k = total - skip < want ? total - skip : want;
		memcpy(dst, (const uint8_t *)nxz_dde_addr(d) + skip, k);
		return k;
	}
	const nxz_dde_t *l = (const nxz_dde_t *)nxz_dde_addr(d);
	uint32_t off = 0;
	for (uint32_t i = 0; i < cnt && off < total && got < want; i++) {
		uint32_t k = nxz_dde_bytes(&l[i]);
		if (k > total - off) k = total - off;
		uint32_t lo = off, hi = off + k;
		if (hi > skip) {
			uint32_t from = lo > skip ? 0 : skip - lo;
			uint32_t take = k - from < want - got ? k - from : want - got;
			memcpy(dst + got, (const uint8_t *)nxz_dde_addr(&l[i]) + from, take);
			got += take;
		}
		off += k;
	}
	return got;
}

static uint32_t dde_capacity(const nxz_dde_t *d)
{
	uint32_t total = nxz_dde_bytes(d), cnt = nxz_dde_count(d), sum = 0;
	if (cnt == 0) return total;
	const nxz_dde_t *l = (const nxz_dde_t *)nxz_dde_addr(d);
	for (uint32_t i = 0; i < cnt; i++) sum += nxz_dde_bytes(&l[i]);
	return sum < total ? sum : total;
}

static void dde_scatter(const nxz_dde_t *d, const uint8_t *src, uint32_t n)
{
	uint32_t cnt = nxz_dde_count(d), off = 0;
	if (cnt == 0) { if (n) memcpy(nxz_dde_addr(d), src, n); return; }
	const nxz_dde_t *l = (const nxz_dde_t *)nxz_dde_addr(d);
	for (uint32_t i = 0; i < cnt && off < n; i++) {
		uint32_t k = nxz_dde_bytes(&l[i]);
		if (k > n - off) k = n - off;
		memcpy(nxz_dde_addr(&l[i]), src + off, k);
		off += k;
	}
}

static void put_cksums(nxz_crb_cpb_t *j, uint32_t crc, uint32_t adler)
{
	nxz_wr32(&j->cpb.out_adler_be, adler);
	j->cpb.out_crc_le = htole32(crc);
}

// NXZ_JOB_TRACE=1: where the time of the single-job interface goes, printed when the process ends
#include <atomic>
#include <chrono>
static struct JobTrace {
	std::atomic<uint64_t> jobs{0}, rounds{0}, ns_acquire{0}, ns_gather{0}, ns_wait{0}, ns_finish{0}, ns_issue{0}, ns_sync{0};
	bool on = false;
	JobTrace() { on = getenv("NXZ_JOB_TRACE") != nullptr; }
	~JobTrace()
	{
		if (!on || !jobs) return;
		const double j = (double)jobs, r = (double)(rounds ? rounds.load() : 1);
		fprintf(stderr, "nxz job trace: %llu compress jobs in %llu rounds (%.1f per round); per job: slot %.1f us, gather %.1f us, in the round %.1f us, "
			"scatter %.1f us; per round: launch calls %.1f us, synchronise %.1f us\n", (unsigned long long)jobs, (unsigned long long)rounds, j / r,
			ns_acquire / j * 1e-3, ns_gather / j * 1e-3, ns_wait / j * 1e-3, ns_finish / j * 1e-3, ns_issue / r * 1e-3, ns_sync / r * 1e-3);
	}
} g_trace;
static inline uint64_t trace_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// One caller's compress job on its way through a round.
struct CompressReq {
	uint32_t fc = 0;
	nxz_batch_job_t job;                  // src / dst: the caller's slot, pinned host memory the kernels work on in place
	const nxz_cpb_t *cpb = nullptr;       // the caller's table (DHT function codes)
	nxz_batch_result_t res;
	uint32_t cnt[316];
	int rc = 0;
	bool taken = false, done = false;
};
#define ROUND_MAX 32u

static void round_free(nxz_ctx::Round &r)
{
	if (r.stream) (void)hipStreamDestroy(r.stream);
	if (r.h_jobs) (void)hipHostFree(r.h_jobs);
	if (r.h_res) (void)hipHostFree(r.h_res);
	if (r.h_dht) (void)hipHostFree(r.h_dht);
	if (r.h_cnt) (void)hipHostFree(r.h_cnt);
	if (r.h_items) (void)hipHostFree(r.h_items);
	if (r.d_prep) (void)hipFree(r.d_prep);
	if (r.d_tok) (void)hipFree(r.d_tok);
	if (r.d_cand2) (void)hipFree(r.d_cand2);
	if (r.d_src) (void)hipFree(r.d_src);
	if (r.d_cut) (void)hipFree(r.d_cut);
	if (r.d_wg) (void)hipFree(r.d_wg);
	if (r.h_targets) (void)hipHostFree(r.h_targets);
	const bool busy = r.busy;                    // (the caller's claim on the round stands)
	r = nxz_ctx::Round();
	r.busy = busy;
}

static bool round_init(nxz_ctx::Round &r)
{
	if (r.ready) return true;
	static std::atomic<unsigned> round_turn{0};
	// (what a failed attempt got so far goes back: the next attempt would write over the pointers -- advisor, round 2)
	auto fail = [&]() { round_free(r); return false; };
	HIPCHK(stream_create_spread(&r.stream, round_turn.fetch_add(1)), return fail());
	HIPCHK(hipHostMalloc((void **)&r.h_jobs, ROUND_MAX * sizeof(nxz_batch_job_t)), return fail());
	HIPCHK(hipHostMalloc((void **)&r.h_res, ROUND_MAX * sizeof(nxz_batch_result_t)), return fail());
	HIPCHK(hipHostMalloc((void **)&r.h_dht, ROUND_MAX * sizeof(nxz_batch_dht_t)), return fail());
	HIPCHK(hipHostMalloc((void **)&r.h_cnt, ROUND_MAX * 316 * sizeof(uint32_t)), return fail());
	HIPCHK(hipMalloc((void **)&r.d_prep, ROUND_MAX * sizeof(nxz_dht_prepared_t)), return fail());
	HIPCHK(hipMalloc((void **)&r.d_tok, (size_t)ROUND_MAX * NXZ_TOK_STRIDE), return fail());
	HIPCHK(hipMalloc((void **)&r.d_cand2, nxz_lz77_cand2_bytes() / NXZ_LZ77_MAX_GRID * ROUND_MAX), return fail());
	HIPCHK(hipMalloc((void **)&r.d_src, (size_t)ROUND_MAX * SUBBLOCK), return fail());
	HIPCHK(hipHostMalloc((void **)&r.h_items, ROUND_MAX * sizeof(nxz_ctx::Round::Item)), return fail());
	r.ready = true;
	return true;
}

// The jobs of one round: a launch of each kernel for all of them.  Targets, job, table and result
// records are pinned host memory that the kernels read and write in place; the sources, which two
// kernels read (the second one in small pieces), are first brought over by a copy kernel (one
// workgroup per job, 16 bytes per lane straight from the callers' pinned staging).  No copy to
// queue, one synchronisation per round.
static int round_run(nxz_ctx *c, nxz_ctx::Round &R, std::vector<CompressReq *> &v)
{
	const uint32_t fc = v[0]->fc;
	const size_t n = v.size();
	const bool dht = nxz_fc_is_dht(fc), count = nxz_fc_has_count(fc), gen = nxz_fc_is_dhtgen(fc);
	(void)hipSetDevice(c->device);
	const uint64_t t0 = g_trace.on ? trace_ns() : 0;
	for (size_t k = 0; k < n; k++) {
		R.h_jobs[k] = v[k]->job;
		R.h_jobs[k].dht_index = (uint32_t)k;
		R.h_items[k].src = v[k]->job.src; R.h_items[k].dst = R.d_src + k * (size_t)SUBBLOCK; R.h_items[k].bytes = v[k]->job.src_len;
		R.h_jobs[k].src = R.h_items[k].dst;
		if (dht && !gen) {
			R.h_dht[k].dhtlen = nxz_in_dhtlen(v[k]->cpb);
			memcpy(R.h_dht[k].dht, v[k]->cpb->in_dht, NXZ_DHT_MAXSZ);
		}
	}
	static const bool each = getenv("NXZ_JOB_TRACE") && atoi(getenv("NXZ_JOB_TRACE")) > 1;   // 2: time every kernel on its own
	static std::atomic<uint64_t> kns[4];
	auto lap = [&](int i, uint64_t from) { if (each) { (void)hipStreamSynchronize(R.stream); kns[i] += trace_ns() - from; } };
	uint64_t k0 = each ? trace_ns() : 0;
	if (nxz_launch_copy_items(R.h_items, (uint32_t)n, R.stream)) return -EIO;
	if (dht && !gen && nxz_launch_dht_prepare(R.h_dht, n, R.d_prep, R.stream)) return -EIO;
	lap(0, k0); k0 = each ? trace_ns() : 0;
	const bool fused = !dht && !count && !gen;             // the fixed code: the LZ77 kernel writes the block itself
	if (nxz_launch_lz77(fused ? NXZ_LZ77_FUSED_FHT : count || gen, R.h_jobs, n, R.d_tok, R.d_cand2, R.h_res, R.h_cnt, nullptr, R.stream)) return -EIO;
	lap(1, k0); k0 = each ? trace_ns() : 0;
	if (gen && nxz_launch_dhtgen(R.h_cnt, n, R.d_prep, nullptr, R.stream)) return -EIO;
	lap(2, k0); k0 = each ? trace_ns() : 0;
	if (!fused && nxz_launch_encode(dht, gen, R.h_jobs, n, R.d_tok, R.d_prep, R.h_res, R.stream)) return -EIO;
	lap(3, k0);
	if (each && (g_trace.rounds & 255) == 255)
		fprintf(stderr, "nxz round kernels (sum so far, us): dht_prepare %.0f lz77 %.0f dhtgen %.0f encode %.0f over %llu rounds\n",
			kns[0] * 1e-3, kns[1] * 1e-3, kns[2] * 1e-3, kns[3] * 1e-3, (unsigned long long)g_trace.rounds + 1);
	const uint64_t t1 = g_trace.on ? trace_ns() : 0;
	HIPCHK(hipStreamSynchronize(R.stream), return -EIO);
	if (g_trace.on) { g_trace.rounds++; g_trace.ns_issue += t1 - t0; g_trace.ns_sync += trace_ns() - t1; }
	for (size_t k = 0; k < n; k++) {
		v[k]->res = R.h_res[k];
		if (count) memcpy(v[k]->cnt, R.h_cnt + k * 316, sizeof(v[k]->cnt));
	}
	return 0;
}

// A round that is free, as long as fewer than NXZ_ROUNDS (default 6) are in flight: few rounds in flight
// make the callers that arrive meanwhile wait and go out TOGETHER, which is what the device wants (it
// runs only a handful of small launches side by side); c->qm is held.
static nxz_ctx::Round *free_round(nxz_ctx *c, bool inflate = false)
{
	// (decompress rounds are a dozen dependent launches each where a compress round is four: three of them in flight serve
	// sixteen and sixty-four threads of 64 KiB calls best -- 1.07 / 2.0 GiB/s against 0.89 / 1.7 with six; NXZ_ROUNDS_INFLATE)
	static const unsigned limit_c = [] { const char *e = getenv("NXZ_ROUNDS"); unsigned v = e ? (unsigned)atoi(e) : 6; return v < 1 ? 1u : v > 16 ? 16u : v; }();
	static const unsigned limit_i = [] { const char *e = getenv("NXZ_ROUNDS_INFLATE"); unsigned v = e ? (unsigned)atoi(e) : getenv("NXZ_ROUNDS") ? (unsigned)atoi(getenv("NXZ_ROUNDS")) : 3; return v < 1 ? 1u : v > 16 ? 16u : v; }();
	const unsigned limit = inflate ? limit_i : limit_c;
	unsigned busy = 0;
	nxz_ctx::Round *f = nullptr;
	for (auto &r : c->rounds) { if (r.busy) busy++; else if (!f) f = &r; }
	return busy < limit ? f : nullptr;
}

// Queue the job; whoever finds a free round takes the job at the head of the queue and every queued
// job with the same function code, runs them, and wakes their owners.  The first caller goes out
// alone at once; those that arrive while it is in flight form the next round.
static int round_submit(nxz_ctx *c, CompressReq *me)
{
	std::unique_lock<std::mutex> lk(c->qm);
	c->q.push_back(me);
	while (!me->done) {
		nxz_ctx::Round *R = nullptr;
		if (!me->taken) R = free_round(c);
		if (!R) { c->qcv.wait(lk); continue; }
		R->busy = true;
		std::vector<CompressReq *> v;
		const uint32_t fc = c->q.front()->fc;
		for (auto it = c->q.begin(); it != c->q.end() && v.size() < ROUND_MAX; ) {
			if ((*it)->fc == fc) { (*it)->taken = true; v.push_back(*it); it = c->q.erase(it); }
			else ++it;
		}
		lk.unlock();
		(void)hipSetDevice(c->device);
		int rc = round_init(*R) ? round_run(c, *R, v) : -ENOMEM;
		if (rc && R->stream) (void)hipStreamSynchronize(R->stream);   // nothing of a failed round may still be in flight
		lk.lock();
		for (auto *r : v) { r->rc = rc; r->done = true; }
		R->busy = false;
		c->qcv.notify_all();
	}
	return me->rc;
}

static int run_compress(nxz_ctx *c, Slot *s, nxz_crb_cpb_t *j, uint32_t fc)
{
	uint32_t srctotal = nxz_dde_bytes(&j->crb.source);
	uint32_t hist = nxz_fc_is_resume(fc) ? nxz_in_histlen(&j->cpb) * 16 : 0;
	if (hist > srctotal) hist = srctotal;
	// only the last 32 KiB of history can be referenced (inc_nx/nxu.h:303-317)
	uint32_t skip = hist > 32768 ? hist - 32768 : 0;
	uint32_t h = hist - skip;
	uint32_t n = srctotal - hist;
	bool partial = false;
	if (h + n > SUBBLOCK) { n = SUBBLOCK - h; partial = true; }     // byte-count limit -> CC 3 partial
	const uint64_t t0 = g_trace.on ? trace_ns() : 0;
	uint32_t got = dde_gather(&j->crb.source, skip, s->h_in, h + n);
	if (got < h) { h = got; n = 0; } else n = got - h;
	uint32_t cap = dde_capacity(&j->crb.target);
	uint32_t dcap = cap < OUT_CAP ? cap & ~3u : OUT_CAP;
	if (cap < 4) dcap = 0;
	const bool count = nxz_fc_has_count(fc);

	CompressReq req;
	req.fc = fc;
	req.cpb = &j->cpb;
	memset(&req.job, 0, sizeof(req.job));
	req.job.src = s->h_in; req.job.dst = s->h_out; req.job.src_len = h + n; req.job.hist_len = h;
	req.job.dst_cap = dcap; req.job.in_crc = nxz_in_crc(&j->cpb); req.job.in_adler = nxz_in_adler(&j->cpb);
	const uint64_t t1 = g_trace.on ? trace_ns() : 0;
	int rc = round_submit(c, &req);
	if (rc) return rc;
	const uint64_t t2 = g_trace.on ? trace_ns() : 0;

	const nxz_batch_result_t r = req.res;
	uint32_t cc = r.cc, ce = 0, tpbc = 0;
	if (cc == NXZ_CC_TARGET_SPACE || cc == NXZ_CC_MISSING_CODE || cc == NXZ_CC_INVALID_DHT) {
		ce = NXZ_CE_TERMINATE;
	} else {
		tpbc = r.tpbc;
		if (tpbc > cap) { cc = NXZ_CC_TARGET_SPACE; ce = NXZ_CE_TERMINATE; tpbc = 0; }
	}
	if (ce != NXZ_CE_TERMINATE) {
		dde_scatter(&j->crb.target, s->h_out, tpbc);
		nxz_putf(&j->cpb.out_w2_be, 16, 3, r.tebc);
		put_cksums(j, r.crc, r.adler);
		uint32_t spbc = skip + r.spbc;
		if (count) {
			for (int i = 0; i < 316; i++) nxz_wr32(&j->cpb.u.out_lzcount_be[i], req.cnt[i]);
			nxz_wr32(&j->cpb.out_spbc_with_count_be, spbc);
		} else {
			nxz_wr32(&j->cpb.u.out_spbc_be, spbc);
		}
		if (cc == 0 && partial) { cc = NXZ_CC_DATA_LENGTH; ce = NXZ_CE_PARTIAL | NXZ_CE_TPBC_VALID; }
	}
	nxz_csb_complete(j, cc, ce, tpbc);
	if (g_trace.on) { g_trace.jobs++; g_trace.ns_gather += t1 - t0; g_trace.ns_wait += t2 - t1; g_trace.ns_finish += trace_ns() - t2; }
	return 0;
}

static int run_wrap(nxz_ctx *c, Slot *s, nxz_crb_cpb_t *j)
{
	uint32_t n = nxz_dde_bytes(&j->crb.source), cap = dde_capacity(&j->crb.target);
	if (n > INF_SRC_CAP) n = INF_SRC_CAP;
	if (n > cap) { nxz_csb_complete(j, NXZ_CC_TARGET_SPACE, NXZ_CE_TERMINATE, 0); return 0; }
	n = dde_gather(&j->crb.source, 0, s->h_in, n);
	nxz_batch_job_t *bj = s->h_job;
	memset(bj, 0, sizeof(*bj));
	bj->src = s->d_in; bj->dst = s->d_out; bj->src_len = n; bj->dst_cap = INF_OUT_CAP;
	HIPCHK(hipMemcpyAsync(s->d_in, s->h_in, n, hipMemcpyHostToDevice, s->stream), return -EIO);
	HIPCHK(hipMemcpyAsync(s->d_job, bj, sizeof(*bj), hipMemcpyHostToDevice, s->stream), return -EIO);
	if (nxz_launch_wrap_sliced(s->d_job, 1, s->d_res, s->stream)) return -EIO;
	HIPCHK(hipMemcpyAsync(s->h_res, s->d_res, sizeof(nxz_batch_result_t), hipMemcpyDeviceToHost, s->stream), return -EIO);
	HIPCHK(hipMemcpyAsync(s->h_out, s->d_out, n, hipMemcpyDeviceToHost, s->stream), return -EIO);
	HIPCHK(hipStreamSynchronize(s->stream), return -EIO);
	dde_scatter(&j->crb.target, s->h_out, n);
	put_cksums(j, s->h_res->crc, s->h_res->adler);
	nxz_wr32(&j->cpb.u.out_spbc_be, n);
	nxz_csb_complete(j, s->h_res->cc, 0, n);
	return 0;
}

// One caller's decompress job on its way through a round (the rounds of the compress jobs, above: the
// callers that arrive while a launch is in flight go out together -- here as one launch of the
// stream-per-wave inflate kernel with a wavefront per job, instead of a launch per job on a stream of
// its own, of which the device runs only a few at a time).
struct InflateReq {
	uint8_t *d_out = nullptr;             // the slot's device buffer for the output (rounds that cut the streams into pieces decode into it)
	nxz_batch_job_t job;                  // src: the slot's device buffer (filled by the round's copy kernel from h_in); dst: the slot's pinned h_out
	const uint8_t *h_in = nullptr;        // the slot's pinned staging of the source
	nxz_batch_dht_t *dht = nullptr;       // the slot's pinned table: in when the job resumes inside a dynamic block, out when it suspends in one
	nxz_batch_result_t res;
	int rc = 0;
	bool taken = false, done = false;
};

static int round_run_inflate(nxz_ctx *c, nxz_ctx::Round &R, std::vector<InflateReq *> &v)
{
	const size_t n = v.size();
	(void)hipSetDevice(c->device);
	for (size_t k = 0; k < n; k++) {
		R.h_jobs[k] = v[k]->job;
		R.h_items[k].src = v[k]->h_in; R.h_items[k].dst = (uint8_t *)v[k]->job.src; R.h_items[k].bytes = v[k]->job.src_len;
		R.h_dht[k] = *v[k]->dht;
	}
	// A wavefront per job takes 0.6-3 ms for 64 KiB however few jobs there are.  Jobs of a few KiB and more are cut into
	// pieces instead (nxz_inflate_cut.hip: 16 streams of 64 KiB 1.0-1.4 GiB/s against 0.2-0.3); those decode into the slots'
	// device buffers -- the pieces' elements are resolved against what is already there, which pinned host memory is too
	// far away for -- and one more kernel takes the outputs to the callers' pinned targets.  NXZ_ROUND_CUT=0: never.
	static const bool cut_on = !(getenv("NXZ_ROUND_CUT") && atoi(getenv("NXZ_ROUND_CUT")) == 0);
	bool cut = cut_on;
	if (cut) {
		size_t longish = 0;
		for (size_t k = 0; k < n; k++) {
			const nxz_batch_job_t &j = v[k]->job;
			// (long enough, and standing where a stream can be cut: at a block header or inside a dynamic block)
			const uint32_t sfbt = (j.resume >> 16) & 15;
			if (j.src_len - (j.hist_len < j.src_len ? j.hist_len : j.src_len) >= 4096 && v[k]->d_out && (sfbt == 0 || (sfbt & 0xe) == 0xe || (sfbt & 0xe) == 0xc)) longish++;
		}
		cut = longish > 0;
	}
	// A round of fresh streams -- what nx_uncompress / inflate(Z_FINISH) of whole buffers are -- goes a stream per WORKGROUP
	// (nxz_inflate_wg.hip: 0.2-0.4 ms for buffers of 64 KiB whatever the round holds, 0.8 ms for 256 KiB, one launch; what that
	// kernel hands back -- a stream that ends early, a target that is too small -- goes a wavefront each behind it).
	// NXZ_ROUND_WG=0: never; NXZ_ROUND_WG_MAX: source bytes of a job at most.
	static const bool wg_on = !(getenv("NXZ_ROUND_WG") && atoi(getenv("NXZ_ROUND_WG")) == 0);
	static const uint32_t wg_src_max = getenv("NXZ_ROUND_WG_MAX") ? (uint32_t)strtoul(getenv("NXZ_ROUND_WG_MAX"), nullptr, 0) : 400000u;
	bool wg = wg_on;
	for (size_t k = 0; k < n && wg; k++) {
		const nxz_batch_job_t &j = v[k]->job;
		if (j.resume || j.hist_len || j.src_len > wg_src_max || !v[k]->d_out) wg = false;
	}
	if (wg && !R.h_targets && hipHostMalloc((void **)&R.h_targets, ROUND_MAX * sizeof(uint8_t *)) != hipSuccess) { (void)hipGetLastError(); R.h_targets = nullptr; wg = false; }
	if (wg && !R.d_wg && hipMalloc((void **)&R.d_wg, nxz_inflate_wg_workspace(ROUND_MAX)) != hipSuccess) { (void)hipGetLastError(); R.d_wg = nullptr; wg = false; }
	const unsigned P = 32;
	if (wg) cut = false;
	if (cut && !R.d_cut) {
		const size_t arena = (size_t)96 << 20;
		if (hipMalloc((void **)&R.d_cut, nxz_inflate_cut_workspace(ROUND_MAX, P, arena)) != hipSuccess || (!R.h_targets && hipHostMalloc((void **)&R.h_targets, ROUND_MAX * sizeof(uint8_t *)) != hipSuccess)) {
			(void)hipGetLastError();
			if (R.d_cut) (void)hipFree(R.d_cut);
			R.d_cut = nullptr; cut = false;
		} else R.cut_arena = arena;
	}
	// (the sources to the device buffers -- except for a round that goes a stream per workgroup: that kernel reads a stream once,
	// 16 bytes a lane, and does so straight from the callers' pinned staging; and its checksum pass takes the outputs to the
	// pinned targets as it reads them: two launches a round less, 158 -> 145 us for a call of 64 KiB)
	if (!wg && nxz_launch_copy_items(R.h_items, (uint32_t)n, R.stream)) return -EIO;
	if (wg) {
		for (size_t k = 0; k < n; k++) { R.h_targets[k] = R.h_jobs[k].dst; R.h_jobs[k].dst = v[k]->d_out; R.h_jobs[k].src = v[k]->h_in; }
		if (nxz_launch_inflate_wg(R.h_jobs, n, R.h_res, R.h_dht, R.d_wg, nullptr, R.h_targets, R.stream)) return -EIO;
	} else if (cut) {
		for (size_t k = 0; k < n; k++) { R.h_targets[k] = R.h_jobs[k].dst; R.h_jobs[k].dst = v[k]->d_out; }
		if (nxz_launch_inflate_cut(R.h_jobs, n, R.h_res, R.h_dht, P, R.d_cut, R.cut_arena, R.stream)) return -EIO;
		if (nxz_launch_copy_out(R.h_jobs, R.h_res, R.h_targets, n, R.stream)) return -EIO;
	} else if (nxz_launch_inflate(R.h_jobs, n, R.h_res, R.h_dht, 1, nullptr, R.stream)) return -EIO;
	HIPCHK(hipStreamSynchronize(R.stream), return -EIO);
	for (size_t k = 0; k < n; k++) {
		v[k]->res = R.h_res[k];
		if ((R.h_res[k].sfbt & 0xe) == 0xc) *v[k]->dht = R.h_dht[k];
	}
	return 0;
}

static int round_submit_inflate(nxz_ctx *c, InflateReq *me)
{
	std::unique_lock<std::mutex> lk(c->qm);
	c->qi.push_back(me);
	while (!me->done) {
		nxz_ctx::Round *R = nullptr;
		if (!me->taken) R = free_round(c, true);
		if (!R) { c->qcv.wait(lk); continue; }
		R->busy = true;
		std::vector<InflateReq *> v;
		while (!c->qi.empty() && v.size() < ROUND_MAX) { c->qi.front()->taken = true; v.push_back(c->qi.front()); c->qi.pop_front(); }
		lk.unlock();
		(void)hipSetDevice(c->device);
		int rc = round_init(*R) ? round_run_inflate(c, *R, v) : -ENOMEM;
		if (rc && R->stream) (void)hipStreamSynchronize(R->stream);
		lk.lock();
		for (auto *r : v) { r->rc = rc; r->done = true; }
		R->busy = false;
		c->qcv.notify_all();
	}
	return me->rc;
}

static int run_decompress(nxz_ctx *c, Slot *s, nxz_crb_cpb_t *j, uint32_t fc)
{
	uint32_t srctotal = nxz_dde_bytes(&j->crb.source);
	bool resume = nxz_fc_is_resume(fc);
	uint32_t hist = resume ? nxz_in_histlen(&j->cpb) * 16 : 0;
	if (hist > srctotal) hist = srctotal;
	uint32_t take = srctotal > INF_SRC_CAP ? INF_SRC_CAP : srctotal;
	if (take < hist) take = hist;
	uint32_t got = dde_gather(&j->crb.source, 0, s->h_in, take);
	if (got < hist) hist = got;
	uint32_t cap = dde_capacity(&j->crb.target);
	uint32_t dcap = cap < INF_OUT_CAP ? cap : INF_OUT_CAP;
	InflateReq req;
	nxz_batch_job_t *bj = &req.job;
	memset(bj, 0, sizeof(*bj));
	bj->src = s->d_in; bj->dst = s->h_out; bj->src_len = got; bj->hist_len = hist; bj->dst_cap = dcap;
	bj->in_crc = nxz_in_crc(&j->cpb); bj->in_adler = nxz_in_adler(&j->cpb);
	bj->reserved = nxz_rd32(&j->crb.reserved1) & NXZ_JOB_SUSPEND_WHEN_FULL;
	s->h_dht->dhtlen = 0;
	if (resume) {
		uint32_t sfbt = nxz_in_sfbt(&j->cpb);
		bj->resume = (sfbt << 16) | (nxz_in_subc(&j->cpb) << 20);
		if ((sfbt & 0xe) == 0x8) bj->resume |= nxz_in_rembytecnt(&j->cpb);
		if ((sfbt & 0xe) == 0xc) {
			s->h_dht->dhtlen = nxz_in_dhtlen(&j->cpb);
			memcpy(s->h_dht->dht, j->cpb.in_dht, NXZ_DHT_MAXSZ);
		}
	}
	req.h_in = s->h_in; req.dht = s->h_dht; req.d_out = s->d_out;
	const uint64_t td0 = g_trace.on ? trace_ns() : 0;
	{
		const int rrc = round_submit_inflate(c, &req);
		if (rrc) return rrc;
	}
	nxz_batch_result_t r = req.res;
	if (g_trace.on) {
		static std::atomic<uint64_t> dj{0}, dns{0}, din{0}, dout{0};
		dj++; dns += trace_ns() - td0; din += got; dout += r.tpbc;
		if ((dj & 4095) == 0) fprintf(stderr, "nxz decompress jobs: %llu, %.1f us each in the round, %.0f source bytes in, %.0f bytes out each\n",
					      (unsigned long long)dj, dns / (double)dj * 1e-3, din / (double)dj, dout / (double)dj);
	}
	uint32_t cc = r.cc, ce = 0, tpbc = 0;
	if (cc != 0 && cc != NXZ_CC_DATA_LENGTH) {
		ce = NXZ_CE_TERMINATE;
	} else {
		tpbc = r.tpbc;
		uint32_t sfbt = r.sfbt & 0xf;
		dde_scatter(&j->crb.target, s->h_out, tpbc);
		put_cksums(j, r.crc, r.adler);
		nxz_wr32(&j->cpb.out_w2_be, r.subc & 0xffff);
		nxz_wr32(&j->cpb.out_w3_be, 0);
		nxz_putf(&j->cpb.out_w3_be, 16, 4, sfbt);
		if ((sfbt & 0xe) == 0x8) nxz_putf(&j->cpb.out_w3_be, 0, 16, r.tebc);
		else if ((sfbt & 0xe) == 0xc) {
			nxz_putf(&j->cpb.out_w3_be, 0, 12, s->h_dht->dhtlen);
			memset(j->cpb.u.d.out_dht, 0, NXZ_DHT_MAXSZ);
			memcpy(j->cpb.u.d.out_dht, s->h_dht->dht, (s->h_dht->dhtlen + 7) / 8);
		}
		nxz_wr32(&j->cpb.u.d.out_spbc_decomp_be, r.spbc);
		if (cc == NXZ_CC_DATA_LENGTH) ce = NXZ_CE_PARTIAL | NXZ_CE_TPBC_VALID;
	}
	nxz_csb_complete(j, cc, ce, tpbc);
	return 0;
}

extern "C" int nxu_run_job(nxz_crb_cpb_t *j, void *handle)
{
	nxz_dev_t *h = (nxz_dev_t *)handle;
	nxz_ctx *c = h ? (nxz_ctx *)h->paste_addr : nullptr;
	if (!j) return -EINVAL;
	if (!c || forked_child()) { nxz_csb_complete(j, NXZ_CC_NO_HW, NXZ_CE_TERMINATE, 0); return 0; }
	uint32_t fc = nxz_fc(j);
	const uint64_t ta = g_trace.on ? trace_ns() : 0;
	Slot *s = slot_acquire(c);
	if (!s) { nxz_csb_complete(j, NXZ_CC_NO_HW, NXZ_CE_TERMINATE, 0); return 0; }
	if (g_trace.on) g_trace.ns_acquire += trace_ns() - ta;
	if (fc == NXZ_FC_WRAP) (void)hipSetDevice(c->device);     // (compress and decompress jobs: the thread that runs the round does)
	int rc;
	if (fc == NXZ_FC_WRAP) rc = run_wrap(c, s, j);
	else if (nxz_fc_is_compress(fc) && !(fc & 1) && !(fc & ~0x2eu) && (!nxz_fc_is_dhtgen(fc) || nxz_fc_is_dht(fc))) rc = run_compress(c, s, j, fc);
	else if (fc == NXZ_FC_DECOMPRESS || fc == NXZ_FC_DECOMPRESS_RESUME) rc = run_decompress(c, s, j, fc);
	else { nxz_csb_complete(j, NXZ_CC_INVALID_OP, NXZ_CE_TERMINATE, 0); rc = 0; }
	if (rc) (void)hipStreamSynchronize(s->stream);            // nothing of a failed job may still be in flight when the slot is reused
	slot_release(c, s);
	if (rc) { fprintf(stderr, "nxz: job failed: %s\n", g_err); return -EAGAIN; }
	return 0;
}

// ---------------------------------------------------------------------------
// __crc32_vpmsum: raw CRC-32 register update (no pre/post inversion), slice-by-8
// ---------------------------------------------------------------------------
static uint32_t crc_t[8][256];
static std::once_flag crc_once;
static void crc_tables(void)
{
	for (uint32_t i = 0; i < 256; i++) {
		uint32_t c = i;
		for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1) ? 0xedb88320u : 0);
		crc_t[0][i] = c;
	}
	for (uint32_t i = 0; i < 256; i++)
		for (int k = 1; k < 8; k++) crc_t[k][i] = crc_t[0][crc_t[k - 1][i] & 0xff] ^ (crc_t[k - 1][i] >> 8);
}

extern "C" unsigned int __crc32_vpmsum(unsigned int crc, const unsigned char *p, unsigned long len)
{
	std::call_once(crc_once, crc_tables);
	while (len && ((uintptr_t)p & 7)) { crc = crc_t[0][(crc ^ *p++) & 0xff] ^ (crc >> 8); len--; }
	while (len >= 8) {
		uint64_t v; memcpy(&v, p, 8);
		v = le64toh(v) ^ crc;
		crc = crc_t[7][v & 0xff] ^ crc_t[6][(v >> 8) & 0xff] ^ crc_t[5][(v >> 16) & 0xff] ^ crc_t[4][(v >> 24) & 0xff] ^
		      crc_t[3][(v >> 32) & 0xff] ^ crc_t[2][(v >> 40) & 0xff] ^ crc_t[1][(v >> 48) & 0xff] ^ crc_t[0][v >> 56];
		p += 8; len -= 8;
	}
	while (len--) crc = crc_t[0][(crc ^ *p++) & 0xff] ^ (crc >> 8);
	return crc;
}
