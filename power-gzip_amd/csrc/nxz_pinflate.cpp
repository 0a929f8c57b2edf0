// nxz_pinflate.cpp -- one deflate stream, or the part of one a caller of inflate() holds, decoded in parallel:
// block-boundary speculation, and cuts at token boundaries inside the blocks.
//
// The reference inflates a stream job after job (/root/reference lib/nx_inflate.c:1060-1762:
// every job resumes where the last one stopped, with the last 32 KiB of output as its history),
// which is fast on an engine that is fast on ONE stream (7.16 GB/s for silesia.tar on POWER9,
// samples/simpleapi/README:27-30).  A wavefront decodes one stream at 15-18 MB/s; the GPU is fast
// on MANY streams.  So the stream is cut where deflate blocks start (nxz_blockfind.hip finds
// the headers of dynamic blocks by trying every bit position) and, inside the blocks, at token
// boundaries (nxz_inflate.hip token_sync_kernel: 64 lanes decode token lengths from neighbouring bits
// until they fall in step), and the pieces -- a block, or a few KiB of one -- are decoded side by side
// with the engine's ordinary decompress jobs (DECOMPRESS_RESUME: source bit offset in in_subc; a
// piece at a cut resumes inside a dynamic block with the block's table, one behind a run of stored
// blocks inside a stored block -- the resume states the reference's engine defines).
// What a piece does not know is its history: the 32 KiB of output in front of it.  But which
// history byte an output byte is a copy of (directly or through copies of copies) does not depend
// on what the history holds.  So a piece is decoded into 16-bit elements (nxz_inflate.hip, W16): the
// byte itself, or 0x8000 | k for "whatever byte k of the 32 KiB in front of me is".  Then
//   - every piece's length, hence its place in the output, is known, and so is whether the piece
//     in front of it ends exactly at its header (a wrong guess of a block start shows here: the
//     boundary is dropped and the merged piece -- only that one -- is decoded again);
//   - one workgroup walks the pieces in order and makes the true 32 KiB window behind each
//     (32 KiB of look-ups per piece);
//   - all pieces are resolved into place at once: final byte = the element, or window[k].
// A part of a stream (nxz_inflate_stream_part) is the same with a resume state in and out: the first piece
// begins wherever the last call stopped, the last one runs out of source like any suspended job.
// The CRC-32 / Adler-32 of the output are computed over 64 KiB slices and combined (zlib's
// crc32_combine idea); the caller checks them against the trailer as for any stream.
#include <hip/hip_runtime.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <vector>
#include "nxz_device.h"

extern "C" {
uint32_t nxz_blockfind_segment(uint64_t srclen);
size_t nxz_blockfind_scratch(uint32_t nseg);
int nxz_launch_find_blocks(const uint8_t *src, uint64_t srclen, uint64_t first_bit, uint64_t *first, uint32_t nseg, void *scratch, hipStream_t stream);
int nxz_launch_copy_items(const void *items, uint32_t n, hipStream_t stream);
int nxz_ctx_device(nxz_ctx_t *c);
int nxz_engine_usable(void);
uint32_t nxz_window_chain_group(uint32_t n);
int nxz_launch_window_chain(const void *pieces, uint32_t n, const uint8_t *win0, uint8_t *windows,
			    uint16_t *gmaps, uint8_t *gwin, hipStream_t stream);
int nxz_launch_resolve(const void *pieces, uint32_t n, const uint8_t *win0, const uint8_t *windows, uint8_t *dst, hipStream_t stream);
}

namespace {

struct CopyItem { const uint8_t *src; uint8_t *dst; uint64_t bytes; };      // src below 16: a fill (nxz_blockfind.hip)
struct Piece { const uint16_t *o; uint64_t len, place; };

constexpr uint32_t WINDOW = 32768;

// GF(2) helpers for crc32_combine: multiply modulo the reflected CRC-32 polynomial
uint32_t gf_mul(uint32_t a, uint32_t b)
{
	uint32_t r = 0;
	for (int i = 0; i < 32; i++) {
		if (b & 0x80000000u) r ^= a;
		a = (a >> 1) ^ ((a & 1) ? 0xedb88320u : 0);
		b <<= 1;
	}
	return r;
}
uint32_t gf_xpow8(uint64_t n)            // x^(8n)
{
	uint32_t r = 0x80000000u, sq = 0x00800000u;
	for (; n; n >>= 1) { if (n & 1) r = gf_mul(r, sq); sq = gf_mul(sq, sq); }
	return r;
}
uint32_t adler_combine(uint32_t a1, uint32_t a2, uint64_t len2)
{
	const uint32_t BASE = 65521;
	const uint32_t rem = (uint32_t)(len2 % BASE);
	uint32_t s1 = a1 & 0xffff, s2 = (uint32_t)(((uint64_t)rem * s1) % BASE);
	s1 += (a2 & 0xffff) + BASE - 1;
	s2 += (a1 >> 16) + (a2 >> 16) + BASE - rem;
	if (s1 >= BASE) s1 -= BASE;
	if (s1 >= BASE) s1 -= BASE;
	if (s2 >= (BASE << 1)) s2 -= (BASE << 1);
	if (s2 >= BASE) s2 -= BASE;
	return s1 | (s2 << 16);
}

// grow-only device/pinned workspaces, a few per device: callers on different threads (each with a stream of
// its own) decode side by side; a caller takes one that is free, or waits for the one its turn falls on
// device bytes all workspaces hold (the figure the idle budget below is kept against)
std::atomic<size_t> g_ws_bytes{0};
// ... and those of them that calls are working in right now: what is IDLE is the difference
std::atomic<size_t> g_ws_inuse{0};
inline size_t keep_bytes();
struct Workspace {
	void *dev = nullptr; size_t dev_cap = 0;
	void *pin = nullptr; size_t pin_cap = 0;
	hipStream_t own = nullptr;                  // for callers that name no stream
	void *built = nullptr; size_t built_cap = 0; // the blocks' decode tables (nxz_inflate.hip Built), device memory
	void *rc_pin = nullptr, *rc_dev = nullptr;  // requests, results and tables of the token boundaries asked for in later rounds (RECUT_BYTES each, made once)
	std::mutex mtx;
	int64_t last_used_ms = 0;                   // when the last call that worked in it ended (read and written under mtx)
	bool need(size_t d, size_t p)
	{
		if (d > dev_cap) {                          // (only ever called by the call that holds the workspace: its bytes are in use)
			const size_t cap_before = dev_cap;
			if (dev) (void)hipFree(dev);
			g_ws_inuse -= dev_cap;
			g_ws_bytes -= dev_cap;
			dev = nullptr; dev_cap = 0;
			// An eighth more than asked for, in whole 2 MiB pages: callers of streams of one kind need a few bytes more or less from one
			// call to the next, and every step up is a hipFree and a hipMalloc of the whole workspace -- gigabytes, a few hundred
			// milliseconds, one after the other through the runtime for all callers (sixteen threads of 64 MiB streams, 8 GB a
			// workspace: 0.5 GiB/s, calls that waited four seconds for their buffers).
			static const bool trace = getenv("NXZ_PINFLATE_TRACE") != nullptr;
			if (trace) fprintf(stderr, "nxz_inflate_stream: a workspace grows from %zu to %zu MiB and more (all workspaces: %zu MiB, in use %zu MiB)\n", cap_before >> 20, d >> 20, g_ws_bytes.load() >> 20, g_ws_inuse.load() >> 20);
			size_t want = (d + d / 8 + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);
			if (d <= keep_bytes() && want > keep_bytes()) want = keep_bytes();     // (a workspace sized to stay inside what one may keep -- capmul_for() -- stays inside it)
			if (hipMalloc(&dev, want) != hipSuccess) {
				(void)hipGetLastError();
				want = d;
				if (hipMalloc(&dev, want) != hipSuccess) return false;
			}
			d = want;
			dev_cap = d;
			g_ws_bytes += d;
			g_ws_inuse += d;
		}
		if (p > pin_cap) {
			if (pin) (void)hipHostFree(pin);
			pin = nullptr; pin_cap = 0;
			p = (p + p / 8 + 0xffff) & ~(size_t)0xffff;           // (likewise)
			if (hipHostMalloc(&pin, p) != hipSuccess) return false;
			pin_cap = p;
		}
		return true;
	}
};
constexpr int NWS = 32;           // (as many callers side by side; 16 threads on 8 of them spent two thirds of a call waiting for one)
Workspace g_ws[64][NWS];
// A workspace that a call has grown beyond this is given back to the device when the call ends (NXZ_PINFLATE_KEEP_MB,
// default 8192 MiB): the engine shares the device with its caller (torch, other libraries).
inline size_t keep_bytes()
{
	static const size_t v = (size_t)(getenv("NXZ_PINFLATE_KEEP_MB") ? atoll(getenv("NXZ_PINFLATE_KEEP_MB")) : 8192) << 20;
	return v;
}
// Room for a piece's 16-bit output: capmul elements per compressed byte to begin with (NXZ_PINFLATE_CAPMUL; a
// piece that runs out of room -- CC 13 -- is decoded again from its start with 8 x as much, up to deflate's own
// limit of 1032, and a repeated piece costs its whole decode time once more: 256 MiB of the corpus at zlib -6
// take 11.2 ms at 100 x, 12.8 at 32 x, 15.7 at 16 x), and a floor that spares short pieces of very repetitive
// data the repeat.  (Round 2 used 100 x with a floor of 2 Mi elements per block piece whatever the stream's
// length: ~11 GB + 9 GB of floors for that stream, per workspace, kept for good -- the advisor's finding.)
// The multiplier of a call: 100 x while the whole stream's buffers stay inside the memory a workspace may keep
// (keep_bytes(): 200 bytes of device memory per compressed byte), less for longer streams, 32 x at least.
static const uint32_t CAPMUL_ENV = getenv("NXZ_PINFLATE_CAPMUL") ? (uint32_t)atoi(getenv("NXZ_PINFLATE_CAPMUL")) : 0;
static inline uint32_t capmul_for(uint64_t src_len)
{
	if (CAPMUL_ENV) return CAPMUL_ENV;
	// (2 bytes an element, a quarter more kept free for the pieces that are decoded again, and the windows)
	const uint64_t fit = keep_bytes() * 10 / (27 * (src_len ? src_len : 1));
	return (uint32_t)(fit > 100 ? 100 : fit < 32 ? 32 : fit);
}
constexpr uint64_t CAP_FLOOR_CUT = 32u << 10, CAP_FLOOR_BLOCK = 128u << 10;
// ... and so is one that would take what all the workspaces hold between calls beyond a budget (NXZ_PINFLATE_IDLE_MB,
// default 16384 MiB or a quarter of the device's memory, at least twice what ONE workspace may keep: 32 callers of 1 MiB parts hold 32 x 32 MiB and never get here; 32
// callers of 64 MiB streams would otherwise keep 32 x 9 GiB for as long as the process lives -- round 3 left that to a manual nxz_trim()).
inline size_t idle_bytes()
{
	// (a quarter of the device's memory where that is more -- an eighth up to round 5's last session: the call that leaves the device
	// idle between two bursts of sixteen 32 MiB streams, 3.6 GiB a workspace, gave six of them back, and the next burst began with
	// their allocation -- : sixteen threads of 16 MiB streams hold 22 GiB, and under the
	// flat 16 GiB every call gave its 1.4 GiB back at its end and asked for them again at the next one's start -- one
	// allocation after the other through the runtime: 0.4 GiB/s all told, single calls of three seconds)
	static const size_t v = [] {
		if (getenv("NXZ_PINFLATE_IDLE_MB")) return (size_t)atoll(getenv("NXZ_PINFLATE_IDLE_MB")) << 20;
		size_t free_b = 0, total_b = 0;
		const size_t flat = (size_t)16384 << 20;
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return flat; }
		return std::max(flat, total_b / 4);
	}();
	return v;
}
inline int64_t ws_now_ms() { return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr int64_t WS_LINGER_MS = 2000;
struct TrimOnExit {
	Workspace &w;
	int dev;
	~TrimOnExit()
	{
		// Both limits -- what ONE workspace may keep, what ALL may hold -- are about memory that lies idle BETWEEN calls.  While
		// other calls are at work a workspace stays as it is: callers of streams of one size come in company, and the next of them
		// takes this workspace a moment later (up to round 5's end the total, busy workspaces and all, was held against the budget:
		// sixteen threads of 32 / 64 MiB nx_uncompress calls, 2.8 / 8 GiB a workspace, gave every workspace back at the call's end and
		// asked for it again at the next one's start -- hipFree and hipMalloc of gigabytes, a few hundred milliseconds each, one after the
		// other through the runtime for all callers: 0.4 GiB/s).  What has lain idle for two seconds while others work goes if it is
		// over a limit; the call that leaves the device idle applies both limits to everything at once.
		const int64_t now = ws_now_ms();
		w.last_used_ms = now;
		g_ws_inuse -= w.dev_cap;
		static const bool trace = getenv("NXZ_PINFLATE_TRACE") != nullptr;
		const size_t busy = g_ws_inuse.load();
		auto idle_now = [&] { const size_t all = g_ws_bytes.load(), b = g_ws_inuse.load(); return all > b ? all - b : (size_t)0; };
		auto drop = [&](Workspace &x) {
			if (trace) fprintf(stderr, "nxz_inflate_stream: a workspace of %zu MiB is given back (all workspaces: %zu MiB, in use %zu MiB, idle budget %zu MiB)\n", x.dev_cap >> 20, g_ws_bytes.load() >> 20, g_ws_inuse.load() >> 20, idle_bytes() >> 20);
			(void)hipFree(x.dev); g_ws_bytes -= x.dev_cap; x.dev = nullptr; x.dev_cap = 0;
		};
		const bool quiet = busy == 0;
		if (quiet && w.dev_cap && (w.dev_cap > keep_bytes() || idle_now() > idle_bytes())) drop(w);
		// (others at work: still a walk -- a workspace beyond what ONE may keep that has lain idle for two seconds goes whatever the
		// total is: under steady load from many threads the call that leaves the device quiet may never come, and the 8-9 GiB
		// workspaces of callers long gone stayed for good)
		for (int k = 0; k < NWS; k++) {
			Workspace &o = g_ws[dev][k];
			if (&o == &w || !o.dev_cap || !o.mtx.try_lock()) continue;
			if (o.dev_cap && (quiet || now - o.last_used_ms > WS_LINGER_MS) && (o.dev_cap > keep_bytes() || idle_now() > idle_bytes())) drop(o);
			o.mtx.unlock();
		}
	}
};
std::atomic<unsigned> g_ws_turn{0};

inline size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

} // namespace

// Gives the device memory of every workspace that no call is using back to the device (include/nxz_engine.h nxz_trim).
extern "C" size_t nxz_pinflate_trim(void)
{
	size_t freed = 0;
	for (int d = 0; d < 64; d++)
		for (int k = 0; k < NWS; k++) {
			Workspace &w = g_ws[d][k];
			if (!w.dev && !w.built && !w.rc_pin && !w.rc_dev) continue;
			if (!w.mtx.try_lock()) continue;
			if (hipSetDevice(d) == hipSuccess) {
				if (w.dev) { freed += w.dev_cap; g_ws_bytes -= w.dev_cap; (void)hipFree(w.dev); w.dev = nullptr; w.dev_cap = 0; }
				if (w.built) { freed += w.built_cap; (void)hipFree(w.built); w.built = nullptr; w.built_cap = 0; }
				if (w.rc_pin) { (void)hipHostFree(w.rc_pin); w.rc_pin = nullptr; }
				if (w.rc_dev) { (void)hipFree(w.rc_dev); w.rc_dev = nullptr; }
			}
			w.mtx.unlock();
		}
	return freed;
}

// Raw deflate stream at src (DEVICE memory, src_len bytes; it starts at bit first_bit and, unless
// `st` is given, must run to a final block) -> dst (DEVICE).  hist (may be NULL): up to 32 KiB of DEVICE bytes that precede
// the output (a preset dictionary or what was inflated before).  Synchronous.
//   0        done: *out_len bytes, checksums of them continued from 0 / 1 style initial values
//            (crc of the data alone, adler of the data alone: combine with yours), *end_bit = first
//            bit behind the final block
//   -ENOTSUP the stream does not lend itself to this (too short, too few dynamic blocks, a piece that
//            blows its buffer, no final block inside src): use the ordinary resume loop
//   -E2BIG   dst_cap is too small (*out_len = bytes needed)        -EILSEQ  the data is not deflate
// st (nxz_inflate_stream_part): a PART of a stream -- what a caller of inflate() holds at one time.  In:
// where the stream stands at first_bit (inside a block: the resume fields a suspended job reported);
// out: where it stands at *end_bit, which is the end of src -- the last piece simply runs out of source
// like any suspended job -- or the header of the first piece whose output no longer fits dst_cap.
static int inflate_stream(nxz_ctx_t *c, const uint8_t *src, uint64_t src_len, uint64_t first_bit,
			  const uint8_t *hist, uint32_t hist_len,
			  uint8_t *dst, uint64_t dst_cap, uint64_t *out_len, uint32_t *crc, uint32_t *adler,
			  uint64_t *end_bit, nxz_stream_resume_t *st, uint32_t *pieces, uint32_t *rounds, void *stream_)
{
	if (!c || !src || !dst || !out_len) return -EINVAL;
	if (!nxz_engine_usable()) return -ENODEV;                  // (a process forked after the engine was opened: no HIP calls there)
	hipStream_t s = (hipStream_t)stream_;
	int dev = nxz_ctx_device(c);                              // the context's device, whatever the calling thread's current one is
	if (dev < 0 || dev >= 64 || hipSetDevice(dev) != hipSuccess) return -ENODEV;
	Workspace *wsp = nullptr;
	for (int k = 0; k < NWS && !wsp; k++) if (g_ws[dev][k].mtx.try_lock()) wsp = &g_ws[dev][k];
	if (!wsp) { wsp = &g_ws[dev][g_ws_turn.fetch_add(1) % NWS]; wsp->mtx.lock(); }
	Workspace &ws = *wsp;
	std::lock_guard<std::mutex> guard(ws.mtx, std::adopt_lock);
	g_ws_inuse += ws.dev_cap;
	TrimOnExit trim{ws, dev};                                       // (destroyed before the guard: still under the lock)
	static const bool own_stream = !(getenv("NXZ_PINFLATE_OWN_STREAM") && atoi(getenv("NXZ_PINFLATE_OWN_STREAM")) == 0);
	if (!s && own_stream) {
		if (!ws.own) {
			// (priorities in turn: streams of one priority may share a hardware queue)
			int least = 0, greatest = 0;
			(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
			const int k = (int)(wsp - g_ws[dev]);
			if (hipStreamCreateWithPriority(&ws.own, hipStreamNonBlocking, k % 3 == 0 ? 0 : k % 3 == 1 ? greatest : least) != hipSuccess) return -EIO;
		}
		s = ws.own;
	}
	if (hist_len > WINDOW) { hist += hist_len - WINDOW; hist_len = WINDOW; }
	if (src_len < (12u << 10) || first_bit / 8 >= src_len) return -ENOTSUP;   // (half a dozen pieces' worth: below that one wavefront is as fast)

	static const bool trace = getenv("NXZ_PINFLATE_TRACE") != nullptr;
	// The small records the kernels are steered by and answer with (jobs, results, requests, tables, pieces) stay
	// in pinned host memory, which the device reads and writes in place: no copy calls, a call of this function
	// is a dozen launches and four waits.  (NXZ_PINFLATE_ZEROCOPY=0: copies to and from device memory instead.)
	static const bool zc = !(getenv("NXZ_PINFLATE_ZEROCOPY") && atoi(getenv("NXZ_PINFLATE_ZEROCOPY")) == 0);
	struct timespec ts0;
	clock_gettime(CLOCK_MONOTONIC, &ts0);
	auto lap = [&](const char *what) {
		if (!trace) return;
		(void)hipStreamSynchronize(s);
		struct timespec t1;
		clock_gettime(CLOCK_MONOTONIC, &t1);
		fprintf(stderr, "nxz_inflate_stream: %-28s %8.3f ms\n", what, (t1.tv_sec - ts0.tv_sec) * 1e3 + (t1.tv_nsec - ts0.tv_nsec) * 1e-6);
		ts0 = t1;
	};
	const uint32_t SEG = nxz_blockfind_segment(src_len);
	const uint32_t nseg = (uint32_t)((src_len + SEG - 1) / SEG);
	// ---- block starts ----
	const size_t nfirst = (size_t)nseg * (1 + NXZ_BLOCKFIND_MORE);         // the first header of every segment, then up to three more
	const size_t o_left = up(nfirst * sizeof(uint64_t), 256);           // (what the search's first kernel leaves for its second)
	if (!ws.need(o_left + nxz_blockfind_scratch(nseg), nfirst * sizeof(uint64_t))) return -ENOMEM;
	if (nxz_launch_find_blocks(src, src_len, first_bit, (uint64_t *)(zc ? ws.pin : ws.dev), nseg, (uint8_t *)ws.dev + o_left, s)) return -EIO;
	if (!zc && hipMemcpyAsync(ws.pin, ws.dev, nfirst * sizeof(uint64_t), hipMemcpyDeviceToHost, s) != hipSuccess) return -EIO;
	if (hipStreamSynchronize(s) != hipSuccess) return -EIO;
	lap("block starts");
	std::vector<uint64_t> B;
	B.push_back(first_bit);
	{
		static const bool more_on = !(getenv("NXZ_PINFLATE_MORE_STARTS") && atoi(getenv("NXZ_PINFLATE_MORE_STARTS")) == 0);
		const uint64_t *F = (const uint64_t *)ws.pin;
		size_t extra = 0;
		for (uint32_t i = 0; i < nseg; i++) {
			uint64_t v[1 + NXZ_BLOCKFIND_MORE];
			uint32_t m = 0;
			if (F[i] != ~0ull) v[m++] = F[i];
			for (uint32_t j = 0; more_on && j < NXZ_BLOCKFIND_MORE; j++) if (F[nseg + (size_t)i * NXZ_BLOCKFIND_MORE + j] != ~0ull) v[m++] = F[nseg + (size_t)i * NXZ_BLOCKFIND_MORE + j];
			std::sort(v, v + m);
			for (uint32_t j = 0; j < m; j++) if (v[j] > first_bit + 64 && v[j] > B.back()) { B.push_back(v[j]); extra += j != 0; }
		}
		if (trace) fprintf(stderr, "nxz_inflate_stream: %zu block starts, %zu of them not the first of their segment\n", B.size(), extra);
	}

	// ---- a stream (or part of one) of a few hundred blocks at most: more places to cut it at ----
	// With that few pieces the call takes as long as the longest piece, milliseconds for a block of 100 KiB.
	// Inside a dynamic block any token boundary will do as a cut (the piece there is a job that resumes
	// inside a dynamic block, with the block's table); token_sync_kernel finds boundaries near the bits asked for.
	struct Sub { uint64_t bit; uint32_t tab, fin, per256; };        // per256: bytes of output per 256 bits of source around the cut (token_sync_kernel's estimate)
	std::vector<Sub> subs;
	std::vector<nxz_batch_dht_t> tabs;
	std::vector<uint64_t> blk_base;              // per table: the bit of the stream its requests count from
	{
		static const int split_max = getenv("NXZ_PINFLATE_SPLIT") ? atoi(getenv("NXZ_PINFLATE_SPLIT")) : 64;      // pieces per block at most (0, 1: blocks only)
		static const uint64_t sub_min = getenv("NXZ_PINFLATE_PIECE_BITS") ? (uint64_t)atoll(getenv("NXZ_PINFLATE_PIECE_BITS")) : 4096;   // the shortest piece
		const uint64_t all_bits = src_len * 8 - first_bit;
		// How many pieces: each costs its share of the window chain (64 KiB of map, 32 KiB of window) and a request for a
		// token boundary, and a wavefront alone decodes 10-15 MB/s -- about 2.5 KiB of the stream per piece is where the
		// two meet (16 MiB of the corpus at zlib -6: 3.2 ms with 657 pieces, 2.5 with 1414, 3.3 with 5839; 64 MiB: 6.2 ms
		// with 7664, 5.4 with 4543, 8.1 with 15566), between 768 pieces (a part of a megabyte or two) and 8192 (a long
		// stream: more than the device holds wavefronts of the decode kernel at one time, 5120).
		static const uint64_t many_env = getenv("NXZ_PINFLATE_PIECES") ? std::max<uint64_t>(1, (uint64_t)atoll(getenv("NXZ_PINFLATE_PIECES"))) : 0;
		// (8192 up to 64 MiB of stream, 16384 beyond: 256 MiB of the round-4 corpus are 53 MiB of stream -- 7.4 ms with 8192 pieces, 8.9
		// with 14692 --, with the image-like and packed classes of round 5 they are 86 MiB of literal-heavy blocks of 20 KB that
		// hardly ever got a second piece at 8192 and take a wavefront 3.3 ms each: 16.3 ms with 8192, 14.1 with 12288, 13.9 with 16384,
		// 15.8 with 24576; with four headers a segment from the block search and the later rounds' pieces cut: 10.5 ms with 16384,
		// 9.65 with 12288, 9.94 with 10240)
		const uint64_t many = many_env ? many_env : all_bits > ((uint64_t)64 << 23) ? 12288 : 8192;
		static const uint64_t few = getenv("NXZ_PINFLATE_PIECES_SHORT") ? std::max<uint64_t>(1, (uint64_t)atoll(getenv("NXZ_PINFLATE_PIECES_SHORT"))) : 768;
		static const uint64_t per_piece = getenv("NXZ_PINFLATE_PIECE_BYTES") ? std::max<uint64_t>(256, (uint64_t)atoll(getenv("NXZ_PINFLATE_PIECE_BYTES"))) : 2560;
		const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(few, all_bits / (8 * per_piece)), std::max<uint64_t>(few, many));
		const uint64_t sub_bits = std::max<uint64_t>(std::max<uint64_t>(sub_min, 2048), all_bits / want);
		std::vector<nxz_sync_req_t> rq, bq;          // requests; the blocks they lie in (bq: src, srclen, header_bit)
		std::vector<uint64_t> rq_base;
		blk_base.clear();
		// (the first piece begins wherever the caller's part of the stream does: inside a dynamic block that is not
		// the last -- then the table is the one the last suspension handed back --, or at a header of whatever kind)
		const bool given0 = st && (st->sfbt & 8) && (st->sfbt & 0xe) != 0xe;
		bool given = false;
		for (size_t i = 0; i < B.size() && split_max > 1; i++) {
			const uint64_t end = i + 1 < B.size() ? B[i + 1] : src_len * 8, span = end - B[i];
			const uint64_t base = (B[i] >> 3) & ~3ull;
			if (end - base * 8 >= 0xffffffffull) continue;
			if (i == 0 && given0 && (st->sfbt & 0xe) != 0xc) continue;
			const uint32_t nsub = (uint32_t)std::min<uint64_t>((uint64_t)split_max, span / sub_bits);
			if (nsub < 2) continue;
			nxz_sync_req_t r;
			r.src = src + base;
			r.srclen = (uint32_t)std::min<uint64_t>(src_len - base, 0x7fffffffull);
			r.header_bit = (uint32_t)(B[i] - base * 8);
			r.guess_bit = 0; r.limit_bit = (uint32_t)(end - base * 8);
			if (i == 0 && given0) { r.header_bit = 0xffffffffu; given = true; }
			bq.push_back(r);
			blk_base.push_back(base * 8);
			for (uint32_t k = 1; k < nsub; k++) {
				r.header_bit = (uint32_t)(bq.size() - 1);         // (which block's tables)
				r.guess_bit = (uint32_t)(B[i] + span * k / nsub - base * 8);
				rq.push_back(r); rq_base.push_back(base * 8);
			}
		}
		const size_t nr = rq.size(), nb = bq.size();
		if (nr) {
			const size_t o_rq = 0, o_bq = up(nr * sizeof(nxz_sync_req_t), 256), o_rs = o_bq + up(nb * sizeof(nxz_sync_req_t), 256),
				     o_tb = o_rs + up(nr * sizeof(nxz_sync_res_t), 256), tot = o_tb + up(nb * sizeof(nxz_batch_dht_t), 256);
			if (!ws.need(tot, tot)) return -ENOMEM;
			// the blocks' decode tables as they stand in LDS: device memory of their own, they are still wanted when the pieces are decoded
			const size_t bbytes = nxz_built_tables_bytes() * nb;
			if (bbytes > ws.built_cap) {
				if (ws.built) (void)hipFree(ws.built);
				ws.built = nullptr; ws.built_cap = 0;
				if (hipMalloc(&ws.built, bbytes + bbytes / 2) != hipSuccess) return -ENOMEM;
				ws.built_cap = bbytes + bbytes / 2;
			}
			uint8_t *Pn = (uint8_t *)ws.pin, *Dv = zc ? Pn : (uint8_t *)ws.dev;
			memcpy(Pn + o_rq, rq.data(), nr * sizeof(nxz_sync_req_t));
			memcpy(Pn + o_bq, bq.data(), nb * sizeof(nxz_sync_req_t));
			if (!zc && hipMemcpyAsync(Dv + o_rq, Pn + o_rq, o_rs, hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
			if (given) {
				nxz_batch_dht_t *tg = (nxz_batch_dht_t *)(Pn + o_tb);            // (it is the first block)
				memset(&tg[0], 0, sizeof(tg[0]));
				tg[0].dhtlen = st->dhtlen;
				memcpy(tg[0].dht, st->dht, NXZ_DHT_MAXSZ);
				if (!zc && hipMemcpyAsync(Dv + o_tb, Pn + o_tb, sizeof(nxz_batch_dht_t), hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
			}
			if (nxz_launch_token_sync((const nxz_sync_req_t *)(Dv + o_bq), (uint32_t)nb, (nxz_batch_dht_t *)(Dv + o_tb), ws.built,
						  (const nxz_sync_req_t *)(Dv + o_rq), (uint32_t)nr, (nxz_sync_res_t *)(Dv + o_rs), s)) return -EIO;
			if (!zc && hipMemcpyAsync(Pn + o_rs, Dv + o_rs, tot - o_rs, hipMemcpyDeviceToHost, s) != hipSuccess) return -EIO;
			if (hipStreamSynchronize(s) != hipSuccess) return -EIO;
			const nxz_sync_res_t *rs = (const nxz_sync_res_t *)(Pn + o_rs);
			const nxz_batch_dht_t *tb = (const nxz_batch_dht_t *)(Pn + o_tb);
			tabs.assign(tb, tb + nb);                        // (a table per block; a cut's `tab` is its block's)
			for (size_t k = 0; k < nr; k++) {
				if (rs[k].bit == 0xffffffffu) continue;
				const uint32_t blk = rq[k].header_bit;
				subs.push_back(Sub{ rq_base[k] + rs[k].bit, blk, blk == 0 && given ? (st->sfbt & 1u) : rs[k].lanes >> 31, (rs[k].lanes >> 8) & 0xffff });
			}
			if (trace) fprintf(stderr, "nxz_inflate_stream: %zu blocks, %zu token boundaries asked for in %zu of them, %zu found\n", B.size(), nr, nb, subs.size());
			lap("token boundaries");
		}
	}

	// a piece of the stream: from a block start (or a token boundary inside a dynamic block) to the next
	struct P {
		uint64_t bit, cstart, cbytes;           // first bit; first byte and byte count of its part of the stream
		uint64_t stop;                          // the next piece starts inside a block: its first bit, counted from cstart (else 0)
		int tab;                                // starts inside a dynamic block: which table (else -1)
		uint32_t hdr0;                          // block headers it has read while it is still in the block it began in
		uint32_t srem, sfin;                    // starts inside a stored block: bytes of it still to come (else 0), its BFINAL
		uint32_t cfin;                          // starts inside a dynamic block: its BFINAL
		uint32_t per256;                        // bytes of output per 256 bits of source, estimated (0: not known)
		size_t stage_off, out_off;              // its 16-byte aligned copy of those bytes / its 16-bit output, in the bump area
		uint64_t cap;                           // output elements it may produce
		uint32_t capmul;
		nxz_batch_result_t res;
		bool done;
	};
	std::vector<P> pc;
	pc.reserve(B.size() + subs.size());
	for (size_t i = 0, k = 0; i < B.size(); i++) {
		P p = P();
		p.bit = B[i]; p.capmul = capmul_for(src_len); p.tab = -1;      // buffer: capmul x the compressed size to begin with (x 8 per repeat), a floor at least
		p.hdr0 = i == 0 && st && (st->sfbt & 8) && (st->sfbt & 0xe) != 0xe ? 0 : 1;
		pc.push_back(p);
		p.hdr0 = 0;
		const uint64_t end = i + 1 < B.size() ? B[i + 1] : src_len * 8;
		for (; k < subs.size() && subs[k].bit < end; k++) {
			if (subs[k].bit < pc.back().bit + 2048 || subs[k].bit + 2048 > end) continue;     // (too close to its neighbours to be worth a piece)
			p.bit = subs[k].bit; p.tab = (int)subs[k].tab; p.cfin = subs[k].fin; p.per256 = subs[k].per256;
			pc.push_back(p);
		}
	}
	// (a whole stream with hardly anything to do side by side: the job loop serves as well.  A part of a stream is
	// taken whatever it is made of -- stored blocks, fixed-Huffman blocks: the caller's alternative is one wavefront too)
	if (pc.size() < 3 && !st) return -ENOTSUP;
	// (n0: how many pieces the control arrays hold -- the first list, and an eighth more for the pieces that later rounds cut)
	const size_t n0 = pc.size() + std::max<size_t>(64, pc.size() / 8), ng0 = n0 / nxz_window_chain_group(0) + n0 / (nxz_window_chain_group(0) * 16) + 4;   // (groups of the smallest size, and the groups of 16 of them)
	// control arrays (sized for the first, longest list of pieces), then a bump area for copies and outputs
	const size_t o_jobs = 0, o_res = o_jobs + up(n0 * sizeof(nxz_batch_job_t), 256), o_items = o_res + up(n0 * sizeof(nxz_batch_result_t), 256),
		     o_pieces = o_items + up((n0 + 4) * sizeof(CopyItem), 256), o_win0 = o_pieces + up(n0 * sizeof(Piece), 256),
		     o_windows = o_win0 + WINDOW, o_gmaps = o_windows + n0 * (size_t)WINDOW,
		     o_gwin = o_gmaps + ng0 * (size_t)WINDOW * 2, o_dht = o_gwin + ng0 * (size_t)WINDOW,
		     o_walk = o_dht + up(n0 * sizeof(nxz_batch_dht_t), 256),
		     o_bump = o_walk + up(n0 * (sizeof(nxz_walk_req_t) + sizeof(nxz_walk_res_t)), 256);
	const size_t pin_jobs = 0, pin_res = pin_jobs + up(n0 * sizeof(nxz_batch_job_t), 256), pin_items = pin_res + up(n0 * sizeof(nxz_batch_result_t), 256),
		     pin_pieces = pin_items + up((n0 + 4) * sizeof(CopyItem), 256), pin_dht = pin_pieces + up(n0 * sizeof(Piece), 256),
		     pin_walk = pin_dht + up((n0 + 1) * sizeof(nxz_batch_dht_t), 256),    // (the tables the jobs start with; the last piece's, out)
		     pin_total = pin_walk + up(n0 * (sizeof(nxz_walk_req_t) + sizeof(nxz_walk_res_t)), 256);
	size_t bump = 0, reserved = 0;
	// The pieces read the caller's stream where it lies: a piece's source starts at the 16-byte boundary in front of
	// its first bit (the decode kernel loads 16 bytes a lane) and the job names the bit.  (A stream that does not
	// itself start on such a boundary: every piece gets an aligned copy of its bytes, as all did up to round 3 --
	// 0.9 ms of copies for 256 MiB.  NXZ_PINFLATE_STAGE=1 forces that.)
	static const bool stage_env = getenv("NXZ_PINFLATE_STAGE") && atoi(getenv("NXZ_PINFLATE_STAGE")) != 0;
	const bool direct = !stage_env && ((uintptr_t)src & 15) == 0;
	static const uint32_t est_room = getenv("NXZ_PINFLATE_EST_ROOM") ? (uint32_t)atoi(getenv("NXZ_PINFLATE_EST_ROOM")) : 3;     // (halves: 1.5 x; 0: the multiplier alone)
	auto size_piece = [&](P &p, const P *next) -> bool {
		const uint64_t next_bit = next ? next->bit : 0;
		p.cstart = direct ? (p.bit >> 3) & ~15ull : p.bit >> 3;
		const uint64_t cend = next_bit ? (next_bit + 7) >> 3 : src_len;
		p.cbytes = cend - p.cstart;
		if (p.cbytes > 0xfffffff0ull - 64) return false;
		p.stop = next && next->tab >= 0 ? next_bit - p.cstart * 8 : 0;
		if (p.stop > 0xffffffffull) return false;
		uint64_t cp = std::max<uint64_t>(p.cbytes * p.capmul, p.tab >= 0 ? CAP_FLOOR_CUT : CAP_FLOOR_BLOCK);
		// (where token_sync_kernel saw the data make much more than the call's multiplier allows for -- long runs: 6 KB of a
		// stream that make 300 KB --, EST_ROOM / 2 times what it saw: such a piece decoded again costs the call a round)
		uint32_t r256 = p.per256;
		if (!r256 && next && next->tab >= 0) r256 = next->per256;
		if (est_room && r256 && p.capmul < 1032) cp = std::max<uint64_t>(cp, std::min<uint64_t>(p.cbytes * 1032, p.cbytes * (uint64_t)r256 * est_room / 64));    // (est_room: in halves)
		if (cp > 0xfff00000ull) return false;
		p.cap = up(cp, 256);
		p.stage_off = bump; if (!direct) bump += up(p.cbytes + 64, 256);
		p.out_off = bump; bump += p.cap * 2;
		p.done = false;
		return true;
	};
	for (size_t i = 0; i < pc.size(); i++)
		if (!size_piece(pc[i], i + 1 < pc.size() ? &pc[i + 1] : nullptr)) return -ENOTSUP;
	uint8_t *D = nullptr, *PN = nullptr;
	bool win0_made = false;
	nxz_batch_dht_t last_dht;                       // the table in force where the last piece stopped
	last_dht.dhtlen = 0;
	for (int attempt = 0; ; attempt++) {
		if (bump > reserved || !ws.dev) {
			// (first attempt, or the repeats outgrew the room left for them: everything is placed and decoded anew)
			if (attempt) {
				bump = 0;
				for (size_t i = 0; i < pc.size(); i++)
					if (!size_piece(pc[i], i + 1 < pc.size() ? &pc[i + 1] : nullptr)) return -ENOTSUP;
			}
			reserved = bump + std::max<size_t>(bump / 4, (size_t)32 << 20);
			if (o_bump + reserved > (200ull << 30)) return -ENOTSUP;
			if (!ws.need(o_bump + reserved, pin_total)) return -ENOMEM;
			win0_made = false;
		}
		D = (uint8_t *)ws.dev; PN = (uint8_t *)ws.pin;
		nxz_batch_job_t *h_jobs = (nxz_batch_job_t *)(PN + pin_jobs), *d_jobs = zc ? h_jobs : (nxz_batch_job_t *)(D + o_jobs);
		nxz_batch_result_t *h_res = (nxz_batch_result_t *)(PN + pin_res), *d_res = zc ? h_res : (nxz_batch_result_t *)(D + o_res);
		CopyItem *h_items = (CopyItem *)(PN + pin_items), *d_items = zc ? h_items : (CopyItem *)(D + o_items);
		size_t nj = 0, ni = 0;
		// the pieces to decode, the longest first (a piece's time goes with its compressed bytes): the launch ends with
		// its last piece, and the device holds 5120 of them at a time (NXZ_PINFLATE_LPT=0: in stream order; 256 MiB of
		// the corpus: the decode launch 5.3 -> 4.8 ms)
		std::vector<size_t> who;
		who.reserve(pc.size());
		for (size_t i = 0; i < pc.size(); i++) if (!pc[i].done) who.push_back(i);
		static const bool lpt = !(getenv("NXZ_PINFLATE_LPT") && atoi(getenv("NXZ_PINFLATE_LPT")) == 0);
		if (lpt && who.size() > 1024) {
			// (by counting, 64 bytes a bucket: a comparison sort of 7000 pieces is a quarter of a millisecond the device waits for)
			constexpr size_t NB = 2048;
			std::vector<uint32_t> cnt(NB + 1, 0);
			// (a piece's time, in cycles of a wavefront on its own: 365 per byte of source + 5.4 per byte it makes -- DESIGN.md
			// section 6 --, the latter from token_sync_kernel's estimate around the piece's first bit; a piece that starts at a
			// block header takes the estimate of the first cut behind it, one without any the average)
			static const bool by_time = !(getenv("NXZ_PINFLATE_LPT_TIME") && atoi(getenv("NXZ_PINFLATE_LPT_TIME")) == 0);
			uint64_t sum256 = 0, n256 = 0;
			for (size_t k = 0; k < pc.size(); k++) if (pc[k].per256) { sum256 += pc[k].per256; n256++; }
			const uint32_t avg256 = n256 ? (uint32_t)(sum256 / n256) : 0;
			auto bucket = [&](size_t i) -> size_t {
				uint32_t r = pc[i].per256;
				if (!r && i + 1 < pc.size() && pc[i + 1].tab >= 0) r = pc[i + 1].per256;
				if (!r) r = avg256;
				const uint64_t eq = by_time ? pc[i].cbytes * (23360 + 11 * (uint64_t)r) / 23360 : pc[i].cbytes;      // in bytes of source without output
				const uint64_t b = eq >> 7;
				return NB - 1 - (size_t)std::min<uint64_t>(b, NB - 1);
			};
			for (size_t k = 0; k < who.size(); k++) cnt[bucket(who[k]) + 1]++;
			for (size_t b = 0; b < NB; b++) cnt[b + 1] += cnt[b];
			std::vector<size_t> tw(who.size());
			for (size_t k = 0; k < who.size(); k++) tw[cnt[bucket(who[k])]++] = who[k];
			who.swap(tw);
		}
		for (size_t k = 0; k < who.size(); k++) {
			const size_t i = who[k];
			P &p = pc[i];
			uint8_t *stg = D + o_bump + p.stage_off;
			if (!direct) h_items[ni++] = CopyItem{ src + p.cstart, stg, p.cbytes };
			nxz_batch_job_t &j = h_jobs[nj++];
			memset(&j, 0, sizeof(j));
			j.src = direct ? src + p.cstart : stg; j.src_len = (uint32_t)p.cbytes;
			j.hist_len = (uint32_t)(p.bit - p.cstart * 8);                               // (a piece: the bit it starts at)
			j.dst = D + o_bump + p.out_off; j.dst_cap = (uint32_t)p.cap;
			j.resume = 0xeu << 16;
			if (i == 0 && st && (st->sfbt & 8)) j.resume = (st->rem & 0xffff) | ((st->sfbt & 0xf) << 16);
			if (p.tab >= 0) j.resume = (0xcu | p.cfin) << 16;                            // inside a dynamic block
			if (p.srem) j.resume = p.srem | ((0x8u | p.sfin) << 16);                     // inside a stored block (on a byte boundary)
			j.in_adler = (uint32_t)p.stop;
			j.in_crc = p.tab >= 0 ? (uint32_t)p.tab + 1 : 0;                             // (which block's ready-made tables)
			// (what of the stream lies behind the piece's range: a piece whose range ends inside a stored block runs on)
			static const bool run_on = !(getenv("NXZ_PINFLATE_RUN_ON") && atoi(getenv("NXZ_PINFLATE_RUN_ON")) == 0);
			j.dht_index = direct && run_on ? (uint32_t)std::min<uint64_t>(src_len - (p.cstart + p.cbytes), 0x7fffffffull) : 0;
		}
		if (!win0_made) {
			// the window in front of the whole output: zeros, the caller's history at its end
			h_items[ni++] = CopyItem{ (const uint8_t *)(uintptr_t)0, D + o_win0, WINDOW - hist_len };
			if (hist_len) h_items[ni++] = CopyItem{ hist, D + o_win0 + (WINDOW - hist_len), hist_len };
			win0_made = true;
		}
		if (!zc && hipMemcpyAsync(d_items, h_items, ni * sizeof(CopyItem), hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
		if (!zc && hipMemcpyAsync(d_jobs, h_jobs, nj * sizeof(nxz_batch_job_t), hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
		// tables: in for a first piece that resumes inside a dynamic block, out for a last piece that stops inside one
		const bool use_dht = st || !tabs.empty();
		nxz_batch_dht_t *h_dht = (nxz_batch_dht_t *)(PN + pin_dht), *d_dht = !use_dht ? nullptr : zc ? h_dht : (nxz_batch_dht_t *)(D + o_dht);
		if (use_dht && nj) {
			bool any = false;
			for (size_t k = 0; k < nj; k++) {
				const P &p = pc[who[k]];
				// (a piece that starts at a cut loads its block's ready-made tables and wants the table's length only;
				// the stream's last piece hands the table bits on when it stops inside that block)
				if (p.tab >= 0) {
					if (who[k] + 1 == pc.size()) h_dht[k] = tabs[(size_t)p.tab];
					else h_dht[k].dhtlen = tabs[(size_t)p.tab].dhtlen;
					any = true;
				}
				else if (who[k] == 0 && st && (st->sfbt & 0xe) == 0xc) {
					memset(&h_dht[k], 0, sizeof(h_dht[k]));
					h_dht[k].dhtlen = st->dhtlen;
					memcpy(h_dht[k].dht, st->dht, NXZ_DHT_MAXSZ);
					any = true;
				}
			}
			if (!zc && any && hipMemcpyAsync(d_dht, h_dht, nj * sizeof(nxz_batch_dht_t), hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
		}
		if (nxz_launch_copy_items(d_items, (uint32_t)ni, s)) return -EIO;
		lap("staging");
		if (nxz_launch_inflate_w16(d_jobs, nj, d_res, d_dht, tabs.empty() ? nullptr : ws.built, attempt == 0, s)) return -EIO;
		lap("decode");
		if (!zc && hipMemcpyAsync(h_res, d_res, nj * sizeof(nxz_batch_result_t), hipMemcpyDeviceToHost, s) != hipSuccess) return -EIO;
		size_t klast = nj;                          // which job the stream's last piece is, if it ran
		for (size_t k = 0; k < nj && st; k++) if (who[k] == pc.size() - 1) klast = k;
		const bool last_ran = klast < nj;
		if (!zc && last_ran && hipMemcpyAsync(&h_dht[n0], d_dht + klast, sizeof(nxz_batch_dht_t), hipMemcpyDeviceToHost, s) != hipSuccess) return -EIO;
		if (hipStreamSynchronize(s) != hipSuccess) return -EIO;
		if (last_ran) last_dht = zc ? h_dht[klast] : h_dht[n0];
		for (size_t k = 0; k < nj; k++) { pc[who[k]].res = h_res[k]; pc[who[k]].done = true; }
		if (trace) {
			size_t slow = 0;
			double sum_us = 0;
			for (size_t k = 0; k < nj; k++) sum_us += h_res[k].crc * 0.01;
			for (size_t k = 1; k < nj; k++) if (h_res[k].crc > h_res[slow].crc) slow = k;
			fprintf(stderr, "nxz_inflate_stream: round %d: the pieces' decode times add up to %.0f us (%.1f us each; / 5120 wavefronts at a time: %.0f us)\n", attempt, sum_us, sum_us / nj, sum_us / 5120);
			{
				std::vector<uint32_t> tt(nj);
				for (size_t k = 0; k < nj; k++) tt[k] = h_res[k].crc;
				std::sort(tt.begin(), tt.end());
				fprintf(stderr, "nxz_inflate_stream: round %d: piece times, us: p10 %.0f  p50 %.0f  p75 %.0f  p90 %.0f  p99 %.0f  max %.0f\n", attempt,
					tt[nj / 10] * 0.01, tt[nj / 2] * 0.01, tt[nj * 3 / 4] * 0.01, tt[nj * 9 / 10] * 0.01, tt[nj * 99 / 100] * 0.01, tt[nj - 1] * 0.01);
				// time per compressed byte, by decile of the pieces in launch order (longest first)
				for (size_t d = 0; d < 10 && nj >= 100; d++) {
					double us = 0, by = 0, ob = 0;
					for (size_t k = d * nj / 10; k < (d + 1) * nj / 10; k++) { us += h_res[k].crc * 0.01; by += h_jobs[k].src_len; ob += h_res[k].tpbc; }
					fprintf(stderr, "nxz_inflate_stream:   launch decile %zu: %.0f us, %.0f bytes in, %.0f out a piece\n", d, us / (nj / 10), by / (nj / 10), ob / (nj / 10));
				}
			}
			const P &q = pc[who[slow]];
			fprintf(stderr, "nxz_inflate_stream: round %d: %zu pieces; the longest took %.1f us (piece %zu, %s%s: %llu bytes in, %u out, cc %u)\n", attempt, nj, h_res[slow].crc * 0.01,
				who[slow], q.tab >= 0 ? "cut" : "block", q.srem ? ", starts in a stored block" : "", (unsigned long long)q.cbytes, h_res[slow].tpbc, h_res[slow].cc);
		}
		// pieces that stopped inside a stored block: where the run of stored blocks they are in ends (stored data may
		// hold anything, things that look like block starts included: they are continued behind the run)
		std::vector<uint64_t> run_end(pc.size(), 0);
		{
			nxz_walk_req_t *h_wq = (nxz_walk_req_t *)(PN + pin_walk);
			nxz_walk_res_t *h_wr = (nxz_walk_res_t *)(PN + pin_walk + up(n0 * sizeof(nxz_walk_req_t), 16));
			nxz_walk_req_t *d_wq = zc ? h_wq : (nxz_walk_req_t *)(D + o_walk);
			nxz_walk_res_t *d_wr = zc ? h_wr : (nxz_walk_res_t *)(D + o_walk + up(n0 * sizeof(nxz_walk_req_t), 16));
			std::vector<size_t> wi;
			for (size_t i = 0; i + 1 < pc.size() && wi.size() < n0; i++) {
				const P &p = pc[i];
				const nxz_batch_result_t &r = p.res;
				if (!p.done || r.cc != NXZ_CC_DATA_LENGTH) continue;
				if ((r.sfbt & 0xe) == 0xe && r.spbc > p.cbytes) {
					// it ran on through stored blocks behind its range and stands at a header, E (see below).  A piece starts
					// there: nothing to ask.  Else: is that header a stored block's (one that did not fit), and where does its run end?
					const uint64_t E = (p.cstart + r.spbc) * 8 - r.subc;
					size_t jn = i + 1;
					while (jn < pc.size() && pc[jn].bit < E) jn++;
					if (jn < pc.size() && pc[jn].bit == E) continue;
					nxz_walk_req_t &w = h_wq[wi.size()];
					w.src = src; w.src_len = src_len;
					w.bit = E; w.rem = 0; w.bfinal = 0;
					wi.push_back(i);
					continue;
				}
				if ((r.sfbt & 0xe) != 0x8 || !r.tebc || pc[i + 1].srem) continue;
				nxz_walk_req_t &w = h_wq[wi.size()];
				w.src = src; w.src_len = src_len;
				w.bit = p.cstart * 8 + (p.stop ? p.stop : p.cbytes * 8) - r.subc;
				w.rem = r.tebc; w.bfinal = r.sfbt & 1;
				wi.push_back(i);
			}
			if (!wi.empty()) {
				if (!zc && hipMemcpyAsync(d_wq, h_wq, wi.size() * sizeof(nxz_walk_req_t), hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
				if (nxz_launch_stored_walk(d_wq, (uint32_t)wi.size(), d_wr, s)) return -EIO;
				if (!zc && hipMemcpyAsync(h_wr, d_wr, wi.size() * sizeof(nxz_walk_res_t), hipMemcpyDeviceToHost, s) != hipSuccess) return -EIO;
				if (hipStreamSynchronize(s) != hipSuccess) return -EIO;
				for (size_t k = 0; k < wi.size(); k++) run_end[wi[k]] = h_wr[k].bit;
				lap("runs of stored blocks");
			}
		}
		// every piece but the last must stop at the header of the next one; the last at the final block's end
		bool again = false;
		std::vector<P> nx;
		nx.reserve(pc.size());
		bool swallow = false, swallow_cuts = false, drop_cuts = false;
		uint64_t swallow_until = 0;
		// `confirmed`: the start of the piece under inspection is known to be a block start -- it is the
		// first piece, or the piece in front is final and ends exactly there.  Only such a piece can say
		// that the DATA is bad; an error in any other piece more likely means that its start is none
		// (deflate output carried as data inside a stored block has block headers that fit each other).
		bool confirmed = true;
		for (size_t i = 0; i < pc.size(); i++) {
			// this piece's start was refuted: it goes into the piece in front (sized below); so do all starts
			// inside a stored block that the piece in front was in the middle of
			// (a piece that did not arrive at a cut inside its block: the block's other cuts go as well)
			// (and a block start that goes takes the cuts inside its block along)
			if (drop_cuts && pc[i].tab >= 0) continue;
			if (swallow || (swallow_cuts && pc[i].tab >= 0) || pc[i].bit < swallow_until) { if (pc[i].tab < 0) drop_cuts = true; swallow = false; continue; }
			swallow_cuts = false; drop_cuts = false;
			P p = pc[i];
			const nxz_batch_result_t &r = p.res;
			const bool conf = confirmed;
			if (r.cc == NXZ_CC_TARGET_SPACE) {
				if (p.capmul >= 1032 * 2) return -ENOTSUP;
				p.capmul *= 8; p.done = false;                     // same piece, more room
				again = true; confirmed = false;
				nx.push_back(p);
				continue;
			}
			const bool err = r.cc != NXZ_CC_DATA_LENGTH && r.cc != 0;
			// the final block ended in this piece: what lies behind it (a trailer, another gzip member, anything)
			// is not this stream's, and block starts seen there are no concern of ours
			const bool fin = !err && (r.sfbt & 0x100) && conf;
			if (!err && (r.sfbt & 0x100) && !conf) {
				// (a final block seen from a start that is not settled yet -- a whole deflate stream carried as data inside a
				// stored block has one: like an error in such a piece below, it waits for what the pieces in front turn out to be)
				confirmed = false;
				nx.push_back(p);
				continue;
			}
			bool ends_well = !err;
			if (!fin && i + 1 < pc.size()) {
				// (r.spbc beyond the piece's own bytes: it ran on through stored blocks behind its range, see below)
				const uint64_t used = (r.spbc > p.cbytes ? (uint64_t)r.spbc * 8 : p.stop ? p.stop : p.cbytes * 8) - r.subc, want = pc[i + 1].bit - p.cstart * 8;
				const uint32_t kind = r.sfbt & 0xe;
				ends_well = r.cc == NXZ_CC_DATA_LENGTH && ((kind == 0xe && used == want) || (kind == 0xa && used == want + 3));
				bool ran_on = false;
				if (r.cc == NXZ_CC_DATA_LENGTH && kind == 0xe && r.spbc > p.cbytes && used > want && p.done) {
					// Its range ended inside a stored block (the start it was cut at is stored data that looks like a
					// header) and it went on: through that block and the stored ones behind it that fit its room, to a
					// header, E.  The starts in between are none.  A piece starts at E: all is well.  None does (a block the
					// search did not see, a fixed one, a stored one that did not fit): one is made, which takes what lies
					// between E -- or the end of the stored run that begins at E -- and the next start, and is decoded next round.
					const uint64_t E = p.cstart * 8 + used;
					size_t jn = i + 1;
					while (jn < pc.size() && pc[jn].bit < E) jn++;
					swallow_until = E;
					const bool at_end = E + 8 > src_len * 8;                       // (it stands where the source ends: a part of a stream)
					if ((jn == pc.size() && at_end) || (jn < pc.size() && pc[jn].bit == E && pc[jn].tab < 0 && !pc[jn].srem)) ran_on = true;
					else {
						P q = P();
						q.bit = E; q.capmul = capmul_for(src_len); q.tab = -1; q.hdr0 = 1; q.done = false;
						swallow_until = std::max<uint64_t>(E + 1, run_end[i]);      // (+ 1: a cut that happens to stand on E goes as well)
						nx.push_back(p);
						nx.push_back(q);
						again = true; confirmed = false;
						continue;
					}
				}
				// a cut inside a dynamic block: this piece must stand exactly there, still in the block it began in (it has
				// read that block's header and no other, or none if it began at a cut itself)
				// the next piece is the rest of the stored block this one stopped in (see below)
				if (pc[i + 1].srem) ends_well = r.cc == NXZ_CC_DATA_LENGTH && kind == 0x8 && used == want && r.tebc == pc[i + 1].srem;
				if (pc[i + 1].tab >= 0) ends_well = r.cc == NXZ_CC_DATA_LENGTH && kind == 0xc && (r.sfbt & 1) == pc[i + 1].cfin && used == want && r.adler == p.hdr0;
				if (ran_on) ends_well = true;
				if (!ends_well && trace)
					fprintf(stderr, "nxz_inflate_stream: piece %zu (bit %llu, %llu bytes in, %u out, start %s) cc %u sfbt %#x subc %u: used %llu bits, next header thought at %llu\n",
						i, (unsigned long long)p.bit, (unsigned long long)p.cbytes, r.tpbc, conf ? "confirmed" : "open", r.cc, r.sfbt, r.subc, (unsigned long long)used, (unsigned long long)want);
			}
			if (err) {
				if (conf) return -EILSEQ;                          // a block that starts here is damaged
				// Its start is not settled yet -- something in front of it is being decoded again (else `conf` would hold),
				// and it is looked at here only because the piece in front arrived exactly at it.  It keeps its result:
				// either the pieces in front turn out to be none and take it along, or its start is confirmed next
				// time round and the data is bad.
				confirmed = false;
				nx.push_back(p);
				continue;
			}
			if (!ends_well && !err && r.cc == NXZ_CC_DATA_LENGTH && (r.sfbt & 0xe) == 0x8 && r.tebc && p.done) {
				// It stopped inside a stored block: the start it was cut at is none (stored data that looks like a
				// header).  What it has decoded stands; what is left of the stored block, and whatever follows up to
				// the next start behind that block, becomes a piece of its own -- a job that resumes inside a stored block.
				P q = P();
				q.bit = p.cstart * 8 + (p.stop ? p.stop : p.cbytes * 8) - r.subc;
				q.capmul = capmul_for(src_len); q.tab = -1; q.hdr0 = 0; q.srem = r.tebc; q.sfin = r.sfbt & 1; q.done = false;
				swallow_until = std::max<uint64_t>(q.bit + (uint64_t)r.tebc * 8, run_end[i]);
				if (q.bit > p.bit && !(q.bit & 7)) {
					nx.push_back(p);
					nx.push_back(q);
					again = true; confirmed = false;
					continue;
				}
				swallow_until = 0;
			}
			if (!ends_well) {
				// the piece did not end where the next was thought to start: that start is wrong
				p.done = false; swallow = true;
				if (pc[i + 1].tab >= 0) swallow_cuts = true;
				if ((r.sfbt & 0xe) == 0x8) swallow_until = (p.cstart + p.cbytes) * 8 + (uint64_t)r.tebc * 8;
				again = true;
			}
			confirmed = conf && ends_well && p.done;
			nx.push_back(p);
			if (fin) break;
		}
		pc.swap(nx);
		if (!again) break;
		if (attempt >= 23 || pc.size() < 2) return -ENOTSUP;

		// ---- the pieces of a later round get token boundaries of their own ----
		// A round lasts as long as its longest piece, and what is decoded again is few and long: the block behind a start that
		// turned out to be none (deflate streams carried inside stored blocks have headers that pass every test: a block of
		// 16384 literals as ONE piece is 1.6 ms), pieces whose output outgrew its room (6 KB of the stream that make 300 KB:
		// 2 ms).  Such a piece is cut the way the blocks were cut before the first round -- with its block's table if it
		// starts at a cut (the piece lies inside that block), else with the table of the header it starts at (what it holds
		// behind that block's end, if anything, yields boundaries that the arrival check refutes: a round more, which is why
		// this is done in the first rounds only).  NXZ_PINFLATE_RECUT=0: never.
		static const bool recut_on = !(getenv("NXZ_PINFLATE_RECUT") && atoi(getenv("NXZ_PINFLATE_RECUT")) == 0);
		if (recut_on && attempt < 3) {
			constexpr uint64_t MIN_BITS = 6144 * 8, PIECE_BITS = 2048 * 8;
			constexpr uint32_t MAXSUB = 8, MAXREQ = 2048, MAXBLK = 256;
			constexpr size_t RECUT_BYTES = (size_t)256 << 10;
			struct Cand { size_t i; uint32_t blk; bool fresh; uint64_t base_bits, end; size_t r0, r1; };
			std::vector<Cand> cands;
			std::vector<nxz_sync_req_t> rq, bqn;
			const size_t built_room = ws.built ? ws.built_cap / nxz_built_tables_bytes() : 0;
			size_t add = 0;
			for (size_t i = 0; i < pc.size(); i++) {
				const P &p = pc[i];
				if (p.done || p.srem || (i == 0 && st && (st->sfbt & 8))) continue;
				const uint64_t end = i + 1 < pc.size() ? pc[i + 1].bit : src_len * 8;
				// (a piece that outgrew its room is long by what it makes, not by what it reads: 32 Ki elements a piece)
				const bool grew = p.res.cc == NXZ_CC_TARGET_SPACE;
				if (end <= p.bit || end - p.bit < (grew ? MIN_BITS / 3 : MIN_BITS)) continue;
				const uint64_t span = end - p.bit;
				const uint64_t want = std::max<uint64_t>(span / PIECE_BITS, grew ? p.cap / 32768 : 0);
				const uint32_t nsub = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(MAXSUB, want), span / 4096);
				if (nsub < 2) continue;
				if (pc.size() + add + nsub - 1 > n0 || rq.size() + nsub - 1 > MAXREQ) break;
				Cand c;
				c.i = i; c.end = end; c.r0 = rq.size();
				if (p.tab >= 0) {
					if ((size_t)p.tab >= blk_base.size()) continue;
					c.blk = (uint32_t)p.tab; c.fresh = false; c.base_bits = blk_base[(size_t)p.tab];
				} else {
					if (p.hdr0 != 1 || tabs.size() != blk_base.size() || tabs.size() + bqn.size() >= built_room || bqn.size() >= MAXBLK) continue;
					const uint64_t base = (p.bit >> 3) & ~3ull;
					nxz_sync_req_t r;
					r.src = src + base;
					r.srclen = (uint32_t)std::min<uint64_t>(src_len - base, 0x7fffffffull);
					r.header_bit = (uint32_t)(p.bit - base * 8);
					r.guess_bit = 0; r.limit_bit = (uint32_t)std::min<uint64_t>(end - base * 8, 0xfffffff0ull);
					c.blk = (uint32_t)(tabs.size() + bqn.size()); c.fresh = true; c.base_bits = base * 8;
					if (end - c.base_bits >= 0xffffffffull) continue;
					bqn.push_back(r);
				}
				if (end - c.base_bits >= 0xffffffffull) continue;
				for (uint32_t k = 1; k < nsub; k++) {
					nxz_sync_req_t r;
					r.src = src + c.base_bits / 8;
					r.srclen = (uint32_t)std::min<uint64_t>(src_len - c.base_bits / 8, 0x7fffffffull);
					r.header_bit = c.blk;
					r.guess_bit = (uint32_t)(p.bit + span * k / nsub - c.base_bits);
					r.limit_bit = (uint32_t)(end - c.base_bits);
					rq.push_back(r);
				}
				c.r1 = rq.size();
				cands.push_back(c);
				add += nsub - 1;
			}
			const size_t nr = rq.size(), nbn = bqn.size();
			const size_t o_rq = 0, o_bq = up(nr * sizeof(nxz_sync_req_t), 256), o_rs = o_bq + up(nbn * sizeof(nxz_sync_req_t), 256),
				     o_tb = o_rs + up(nr * sizeof(nxz_sync_res_t), 256), tot = o_tb + up(nbn * sizeof(nxz_batch_dht_t), 256);
			bool go = nr != 0 && tot <= RECUT_BYTES;
			if (go && !ws.rc_pin) {
				if (hipHostMalloc(&ws.rc_pin, RECUT_BYTES) != hipSuccess) { (void)hipGetLastError(); ws.rc_pin = nullptr; go = false; }
				else if (!zc && hipMalloc(&ws.rc_dev, RECUT_BYTES) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(ws.rc_pin); ws.rc_pin = nullptr; ws.rc_dev = nullptr; go = false; }
			}
			if (go && !zc && !ws.rc_dev) go = false;
			if (go) {
				uint8_t *Pn = (uint8_t *)ws.rc_pin, *Dv = zc ? Pn : (uint8_t *)ws.rc_dev;
				memcpy(Pn + o_rq, rq.data(), nr * sizeof(nxz_sync_req_t));
				if (nbn) memcpy(Pn + o_bq, bqn.data(), nbn * sizeof(nxz_sync_req_t));
				if (!zc && hipMemcpyAsync(Dv + o_rq, Pn + o_rq, o_rs, hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
				if (nxz_launch_token_sync_more((const nxz_sync_req_t *)(Dv + o_bq), (uint32_t)nbn, (nxz_batch_dht_t *)(Dv + o_tb), ws.built, (uint32_t)tabs.size(),
							       (const nxz_sync_req_t *)(Dv + o_rq), (uint32_t)nr, (nxz_sync_res_t *)(Dv + o_rs), s)) return -EIO;
				if (!zc && hipMemcpyAsync(Pn + o_rs, Dv + o_rs, tot - o_rs, hipMemcpyDeviceToHost, s) != hipSuccess) return -EIO;
				if (hipStreamSynchronize(s) != hipSuccess) return -EIO;
				const nxz_sync_res_t *rs = (const nxz_sync_res_t *)(Pn + o_rs);
				const nxz_batch_dht_t *tb = (const nxz_batch_dht_t *)(Pn + o_tb);
				for (size_t k = 0; k < nbn; k++) tabs.push_back(tb[k]);
				for (const Cand &c : cands) if (c.fresh) blk_base.push_back(c.base_bits);
				std::vector<P> out;
				out.reserve(pc.size() + add);
				size_t ci = 0, made = 0;
				for (size_t i = 0; i < pc.size(); i++) {
					out.push_back(pc[i]);
					if (ci >= cands.size() || cands[ci].i != i) continue;
					const Cand &c = cands[ci++];
					uint64_t last = pc[i].bit;
					for (size_t k = c.r0; k < c.r1; k++) {
						if (rs[k].bit == 0xffffffffu) continue;
						const uint64_t bit = c.base_bits + rs[k].bit;
						if (bit < last + 2048 || bit + 2048 > c.end) continue;
						P q = P();
						q.bit = bit; q.tab = (int)c.blk; q.cfin = c.fresh ? rs[k].lanes >> 31 : pc[i].cfin; q.per256 = (rs[k].lanes >> 8) & 0xffff;
						q.capmul = pc[i].capmul; q.hdr0 = 0; q.done = false;
						out.push_back(q);
						last = bit; made++;
					}
				}
				if (trace) fprintf(stderr, "nxz_inflate_stream: round %d: %zu pieces to decode again asked for %zu token boundaries (%zu tables more): %zu found\n", attempt, cands.size(), nr, nbn, made);
				pc.swap(out);
				lap("token boundaries again");
			}
		}
		for (size_t i = 0; i < pc.size(); i++)
			if (!pc[i].done && !size_piece(pc[i], i + 1 < pc.size() ? &pc[i + 1] : nullptr)) return -ENOTSUP;
	}
	size_t n = pc.size();
	if (!n) return -ENOTSUP;
	// where the decoded part ends: the last piece must have seen the final block, or (a part of a stream)
	// stopped at the end of the source like any suspended job; pieces whose output no longer fits are left
	// for the next call, which then starts at a block header
	{
		uint64_t sum = 0;
		size_t m = 0;
		while (m < n && sum + pc[m].res.tpbc <= dst_cap) sum += pc[m++].res.tpbc;
		if (m < n && (!st || m == 0)) {
			for (sum = 0, m = 0; m < n; m++) sum += pc[m].res.tpbc;
			*out_len = sum;
			return -E2BIG;
		}
		const bool cut = m < n;
		const P &lp = pc[m - 1];
		const nxz_batch_result_t &r = lp.res;
		if (r.cc != 0 && r.cc != NXZ_CC_DATA_LENGTH) return -EILSEQ;
		const bool fin = (r.sfbt & 0x100) != 0;
		if (!fin && !st) return -ENOTSUP;
		const uint64_t stop = cut ? pc[m].bit : (lp.cstart + r.spbc) * 8 - r.subc;
		if (end_bit) *end_bit = stop;
		if (st) {
			memset(st, 0, sizeof(*st));
			st->final = fin && !cut;
			if (!fin && !cut) {
				st->sfbt = r.sfbt & 0xf; st->rem = r.tebc;
				if ((r.sfbt & 0xe) == 0xc) { st->dhtlen = (r.sfbt >> 16) & 0xfff; memcpy(st->dht, last_dht.dht, NXZ_DHT_MAXSZ); }
			}
			if (cut) {
				// the first piece left out begins at a block header (all zero), or inside a block: stored, or dynamic at a cut
				const P &np = pc[m];
				if (np.srem) { st->sfbt = 0x8 | np.sfin; st->rem = np.srem; }
				else if (np.tab >= 0) {
					st->sfbt = 0xc | np.cfin; st->dhtlen = tabs[(size_t)np.tab].dhtlen;
					memcpy(st->dht, tabs[(size_t)np.tab].dht, NXZ_DHT_MAXSZ);
				}
			}
		}
		n = m;
	}
	// ---- places, true windows, resolution ----
	uint8_t *P_ = (uint8_t *)ws.pin;
	Piece *h_pieces = (Piece *)(P_ + pin_pieces);
	Piece *d_pieces = zc ? h_pieces : (Piece *)(D + o_pieces);
	uint8_t *d_windows = D + o_windows;
	nxz_batch_job_t *d_jobs = (nxz_batch_job_t *)(D + o_jobs);
	nxz_batch_result_t *d_res = (nxz_batch_result_t *)(D + o_res);
	uint64_t total = 0;
	for (size_t i = 0; i < n; i++) {
		h_pieces[i] = Piece{ (const uint16_t *)(D + o_bump + pc[i].out_off), pc[i].res.tpbc, total };
		total += pc[i].res.tpbc;
	}
	*out_len = total;
	if (total > dst_cap) return -E2BIG;
	if (!zc && hipMemcpyAsync(d_pieces, h_pieces, n * sizeof(Piece), hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
	const uint8_t *win0 = D + o_win0;
	if (nxz_launch_window_chain(d_pieces, (uint32_t)n, win0, d_windows, (uint16_t *)(D + o_gmaps), D + o_gwin, s)) return -EIO;
	lap("tail maps + window chain");
	if (nxz_launch_resolve(d_pieces, (uint32_t)n, win0, d_windows, dst, s)) return -EIO;
	lap("resolve");
	// ---- checksums: 256 KiB slices of the output (the job / result arrays are free again) ----
	// (a workgroup per slice, and a workgroup takes 64 KiB at a time: 64 KiB slices fill the device up to half a GiB)
	static const uint64_t slice_env = getenv("NXZ_PINFLATE_CKSUM_SLICE") ? (uint64_t)atoll(getenv("NXZ_PINFLATE_CKSUM_SLICE")) : 0;
	const uint64_t SLICE = slice_env ? slice_env : total <= (512u << 20) ? 64 << 10 : 256 << 10;
	const size_t nsl = (size_t)((total + SLICE - 1) / SLICE);
	std::vector<nxz_batch_result_t> sres(nsl);
	if (!nsl) { if (crc) *crc = 0; if (adler) *adler = 1; }
	for (size_t o = 0; o < nsl; o += n0) {
		const size_t m = std::min(nsl - o, n0);
		nxz_batch_job_t *h_jobs = (nxz_batch_job_t *)(P_ + pin_jobs);
		nxz_batch_result_t *h_res = (nxz_batch_result_t *)(P_ + pin_res);
		for (size_t k = 0; k < m; k++) {
			memset(&h_jobs[k], 0, sizeof(nxz_batch_job_t));
			memset(&h_res[k], 0, sizeof(nxz_batch_result_t));
			h_jobs[k].dst = dst + (o + k) * SLICE;
			// (what of the caller's buffer lies behind the slice's first byte: the kernel loads 16 bytes a lane where
			// that stays inside it -- left at 0, as up to round 3, every byte was loaded on its own: 0.58 ms for 256 MiB)
			h_jobs[k].dst_cap = (uint32_t)std::min<uint64_t>(dst_cap - (o + k) * SLICE, 0xffffffffull);
			h_jobs[k].in_crc = 0; h_jobs[k].in_adler = 1;
			h_res[k].tpbc = (uint32_t)std::min<uint64_t>(SLICE, total - (o + k) * SLICE);
		}
		if (!zc && hipMemcpyAsync(d_jobs, h_jobs, m * sizeof(nxz_batch_job_t), hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
		if (!zc && hipMemcpyAsync(d_res, h_res, m * sizeof(nxz_batch_result_t), hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
		if (nxz_launch_cksum(zc ? h_jobs : d_jobs, m, zc ? h_res : d_res, s)) return -EIO;
		if (!zc && hipMemcpyAsync(h_res, d_res, m * sizeof(nxz_batch_result_t), hipMemcpyDeviceToHost, s) != hipSuccess) return -EIO;
		if (hipStreamSynchronize(s) != hipSuccess) return -EIO;
		memcpy(&sres[o], h_res, m * sizeof(nxz_batch_result_t));
	}
	lap("checksums");
	if (trace) {
		uint64_t mx = 0;
		for (size_t i = 0; i < n; i++) mx = std::max<uint64_t>(mx, pc[i].res.tpbc);
		fprintf(stderr, "nxz_inflate_stream: %zu pieces, largest %llu bytes out, mean %llu\n", n, (unsigned long long)mx, (unsigned long long)(total / n));
		// the pieces that took longest (the kernel reports 10 ns ticks in the CRC's place)
		std::vector<size_t> ord(n);
		for (size_t i = 0; i < n; i++) ord[i] = i;
		std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return pc[a].res.crc > pc[b].res.crc; });
		for (size_t k = 0; k < std::min<size_t>(n, 6); k++) {
			const P &q = pc[ord[k]];
			fprintf(stderr, "nxz_inflate_stream:   piece %zu (%s): %llu bytes in, %u out, %.1f us\n", ord[k], q.tab >= 0 ? "cut" : "block", (unsigned long long)q.cbytes, q.res.tpbc, q.res.crc * 0.01);
		}
	}
	const uint32_t nround = 1;
	uint32_t cr = 0, ad = 1;
	if (!nsl) return 0;
	const uint32_t xslice = gf_xpow8(SLICE);                 // (all slices but the last are this long)
	for (size_t k = 0; k < nsl; k++) {
		cr = k ? gf_mul(cr, sres[k].tpbc == SLICE ? xslice : gf_xpow8(sres[k].tpbc)) ^ sres[k].crc : sres[k].crc;
		ad = k ? adler_combine(ad, sres[k].adler, sres[k].tpbc) : sres[k].adler;
	}
	if (crc) *crc = cr;
	if (adler) *adler = ad;
	if (pieces) *pieces = (uint32_t)n;
	if (rounds) *rounds = nround;
	return 0;
}

extern "C" int nxz_inflate_stream(nxz_ctx_t *c, const uint8_t *src, uint64_t src_len, uint64_t first_bit,
				  const uint8_t *hist, uint32_t hist_len,
				  uint8_t *dst, uint64_t dst_cap, uint64_t *out_len, uint32_t *crc, uint32_t *adler,
				  uint64_t *end_bit, uint32_t *pieces, uint32_t *rounds, void *stream)
{
	return inflate_stream(c, src, src_len, first_bit, hist, hist_len, dst, dst_cap, out_len, crc, adler, end_bit, nullptr, pieces, rounds, stream);
}

// A part of a stream (include/nxz_engine.h): what inflate() holds at one time.
extern "C" int nxz_inflate_stream_part(nxz_ctx_t *c, const uint8_t *src, uint64_t src_len, uint64_t first_bit,
				       const uint8_t *hist, uint32_t hist_len,
				       uint8_t *dst, uint64_t dst_cap, uint64_t *out_len, uint32_t *crc, uint32_t *adler,
				       uint64_t *end_bit, nxz_stream_resume_t *state, uint32_t *pieces, void *stream)
{
	if (!state) return -EINVAL;
	return inflate_stream(c, src, src_len, first_bit, hist, hist_len, dst, dst_cap, out_len, crc, adler, end_bit, state, pieces, nullptr, stream);
}
