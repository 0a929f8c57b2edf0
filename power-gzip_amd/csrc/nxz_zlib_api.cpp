// nxz_zlib_api.cpp -- the unprefixed zlib entry points (LD_PRELOAD drop-in), dispatching per
// stream between the MI355X engine (nx_*, libnxz_amd.so) and software zlib.
//
// Counterpart of the reference's dispatch layer: the unprefixed functions at the bottom of
// lib/nx_deflate.c:2236-2580, lib/nx_inflate.c:1982-2364, lib/nx_compress.c:77-117,
// lib/nx_uncompr.c:90-149 and the dlopen trampolines of lib/sw_zlib.c:57-336.
//   NX_GZIP_TYPE_SELECTOR = 0 auto (default), 1 software zlib, 2 engine, 3 engine deflate + zlib
//   inflate; NX_GZIP_COMP_MODE / NX_GZIP_DEC_MODE per direction; the same keys in the file named by
//   NX_GZIP_CONFIG (lib/nx_zlib.c:1067-1217; parsed by nxz_config.cpp).  NX_GZIP_TRACE=8 gathers the
//   reference's call statistics here, in the dispatch layer, as lib/nx_deflate.c:2472-2520 does.
// Auto: one-shot calls and streams whose first call brings less than the measured break-even (nxz_config
// auto_comp_min / auto_dec_min; the reference: <= 1024 bytes, lib/nx_zlib.h:88-89) go to zlib; everything goes to zlib
// when no engine can be opened.  A stream stays with the backend that initialised it; which one
// that was is read from the state tag, so no stream map is needed (the reference keeps one for
// its switchable AUTO streams, lib/nx_map.c).
#include <zlib.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <set>
#include "../../include/nxz_engine.h"
#include "../../include/nxz_zlib.h"
#include "../../include/nxz_config.h"
#include "nxz_host.h"
#include <time.h>

namespace {

enum { MODE_AUTO = 0, MODE_SW = 1, MODE_NX = 2 };
constexpr uint64_t MAGIC_DEF = 0x6e787a2d64656621ull, MAGIC_INF = 0x6e787a2d696e6621ull;   // nxz_stream.cpp

struct Sw {
	void *h = nullptr;
#define SWF(ret, name, args) ret (*name) args = nullptr;
	SWF(const char *, zlibVersion, (void))
	SWF(int, deflateInit_, (z_streamp, int, const char *, int))
	SWF(int, deflateInit2_, (z_streamp, int, int, int, int, int, const char *, int))
	SWF(int, deflate, (z_streamp, int))
	SWF(int, deflateEnd, (z_streamp))
	SWF(int, deflateReset, (z_streamp))
	SWF(int, deflateResetKeep, (z_streamp))
	SWF(uLong, deflateBound, (z_streamp, uLong))
	SWF(int, deflateSetHeader, (z_streamp, gz_headerp))
	SWF(int, deflateSetDictionary, (z_streamp, const Bytef *, uInt))
	SWF(int, deflateCopy, (z_streamp, z_streamp))
	SWF(int, deflateParams, (z_streamp, int, int))
	SWF(int, inflateInit_, (z_streamp, const char *, int))
	SWF(int, inflateInit2_, (z_streamp, int, const char *, int))
	SWF(int, inflate, (z_streamp, int))
	SWF(int, inflateEnd, (z_streamp))
	SWF(int, inflateReset, (z_streamp))
	SWF(int, inflateReset2, (z_streamp, int))
	SWF(int, inflateSetDictionary, (z_streamp, const Bytef *, uInt))
	SWF(int, inflateGetHeader, (z_streamp, gz_headerp))
	SWF(int, inflateSyncPoint, (z_streamp))
	SWF(int, inflateCopy, (z_streamp, z_streamp))
	SWF(int, inflateResetKeep, (z_streamp))
	SWF(gzFile, gzopen, (const char *, const char *))
	SWF(gzFile, gzdopen, (int, const char *))
	SWF(int, gzread, (gzFile, voidp, unsigned))
	SWF(int, gzwrite, (gzFile, voidpc, unsigned))
	SWF(int, gzclose, (gzFile))
	SWF(int, compress, (Bytef *, uLongf *, const Bytef *, uLong))
	SWF(int, compress2, (Bytef *, uLongf *, const Bytef *, uLong, int))
	SWF(uLong, compressBound, (uLong))
	SWF(int, uncompress, (Bytef *, uLongf *, const Bytef *, uLong))
	SWF(int, uncompress2, (Bytef *, uLongf *, const Bytef *, uLong *))
	SWF(uLong, crc32, (uLong, const Bytef *, uInt))
	SWF(uLong, adler32, (uLong, const Bytef *, uInt))
	SWF(uLong, crc32_combine, (uLong, uLong, z_off_t))
	SWF(uLong, adler32_combine, (uLong, uLong, z_off_t))
#undef SWF
} sw;

int g_mode_def = MODE_AUTO, g_mode_inf = MODE_AUTO;      // nx_config.mode.deflate / .inflate
bool g_engine = false;
std::once_flag g_once;

void init_once()
{
	const nxz_config_t *cfg = nxz_config();
	g_mode_def = cfg->mode_deflate;
	g_mode_inf = cfg->mode_inflate;
	const char *path = getenv("NXZ_ZLIB_PATH");
	// RTLD_DEEPBIND: the real zlib must bind its own internal calls to itself, not to this shim
	sw.h = dlopen(path ? path : "libz.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
	if (sw.h) {
#define REG(name) *(void **)(&sw.name) = dlsym(sw.h, #name);
		REG(zlibVersion) REG(deflateInit_) REG(deflateInit2_) REG(deflate) REG(deflateEnd) REG(deflateReset)
		REG(deflateResetKeep) REG(deflateBound) REG(deflateSetHeader) REG(deflateSetDictionary) REG(deflateCopy)
		REG(deflateParams) REG(inflateInit_) REG(inflateInit2_) REG(inflate) REG(inflateEnd) REG(inflateReset)
		REG(inflateReset2) REG(inflateSetDictionary) REG(inflateGetHeader) REG(inflateSyncPoint) REG(inflateCopy)
		REG(compress) REG(compress2) REG(compressBound) REG(uncompress) REG(uncompress2) REG(crc32) REG(adler32)
		REG(crc32_combine) REG(adler32_combine) REG(inflateResetKeep) REG(gzopen) REG(gzdopen) REG(gzread) REG(gzwrite) REG(gzclose)
#undef REG
	} else if (g_mode_def != MODE_NX || g_mode_inf != MODE_NX) {
		fprintf(stderr, "nxz: cannot dlopen software zlib (%s): forcing engine mode\n", dlerror());
		g_mode_def = g_mode_inf = MODE_NX;                 // lib/nx_zlib.c:1357-1360
	}
	if (g_mode_def != MODE_SW || g_mode_inf != MODE_SW) {
		nxz_dev_t probe;
		memset(&probe, 0, sizeof(probe));
		if (nx_function_begin(NXZ_FUNC_COMP_GZIP, cfg->dev_num, &probe) == 0) { g_engine = true; nx_function_end(&probe); }
		else if (g_mode_def == MODE_NX || g_mode_inf == MODE_NX) fprintf(stderr, "nxz: engine mode selected but no engine is available\n");
	}
	nxz_log(2, "nxz preload: deflate mode %d, inflate mode %d, engine %s\n", g_mode_def, g_mode_inf, g_engine ? "open" : "absent");
}

inline void init() { std::call_once(g_once, init_once); }
// a forked child cannot use the parent's engine: its new streams go to software zlib
inline bool want_nx(int mode) { return nxz_engine_usable() && (mode == MODE_NX || (mode == MODE_AUTO && g_engine)); }
// AUTO also looks at how the engine has been doing (lib/nx_deflate.c:714, lib/nx_inflate.c: s->use_nx =
// avg_delay <= nx_config.[de]compress_delay; lib/nx_zlib.h:376-422): a stream that is opened while the
// average job delay is above the threshold is served by software zlib, which lets the average fade
// (decrease_delay, :443-449) until the engine is tried again.
inline bool slow(int mode, uint64_t threshold)
{
	if (mode != MODE_AUTO || nxz_avg_delay() <= threshold) return false;
	nxz_decrease_delay();
	return true;
}
inline bool want_nx_def() { init(); return want_nx(g_mode_def) && !slow(g_mode_def, nxz_config()->compress_delay); }
inline bool want_nx_inf() { init(); return want_nx(g_mode_inf) && !slow(g_mode_inf, nxz_config()->decompress_delay); }
inline bool is_nx(z_streamp s, uint64_t magic) { return s && s->state && *(const uint64_t *)s->state == magic; }
inline uint64_t now_ns() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; }

} // namespace

#define EXPORT extern "C" __attribute__((visibility("default")))

EXPORT const char *zlibVersion(void) { init(); return sw.zlibVersion ? sw.zlibVersion() : ZLIB_VERSION; }

// ---- deflate ----
EXPORT int deflateInit2_(z_streamp s, int level, int method, int wbits, int memLevel, int strategy, const char *ver, int size)
{
	const bool nx = want_nx_def();
	nxz_stats_inc("deflateInit");
	if (nx) {
		int rc = nx_deflateInit2_(s, level, method, wbits, memLevel, strategy, ver, size);
		// parameters the engine does not take (small windows, Z_FILTERED, ...) go to software in auto mode
		if (rc == Z_OK || g_mode_def == MODE_NX || !sw.deflateInit2_) return rc;
	}
	return sw.deflateInit2_ ? sw.deflateInit2_(s, level, method, wbits, memLevel, strategy, ver, size) : Z_STREAM_ERROR;
}
EXPORT int deflateInit_(z_streamp s, int level, const char *ver, int size)
{
	return deflateInit2_(s, level, Z_DEFLATED, 15, 8, Z_DEFAULT_STRATEGY, ver, size);
}
#define DISPATCH_DEF(call_nx, call_sw, err) do { init(); if (is_nx(s, MAGIC_DEF)) return call_nx; return sw.deflate ? call_sw : err; } while (0)
// AUTO mode, first call of a stream: a caller that brings ALL its input (Z_FINISH) and less of it than the break-even
// (nxz_config auto_comp_min / auto_dec_min: measured, tools/api_sweep.py) is better served by software zlib -- the engine
// stream, to which nothing has happened yet, is closed and the same z_stream reopened there with the parameters it
// was made with.  Only with Z_FINISH, as the reference (lib/nx_zlib.h:389-419): "even when compressing a large amount
// of data, the first call may not have enough input" -- a client that streams through 16-64 KiB buffers (zpipe,
// minigz, gzip, CPython's zlib) stays on the engine, whose stream layer gathers small inputs before it runs a job.
static bool auto_to_sw_deflate(z_streamp s, int flush)
{
	if (g_mode_def != MODE_AUTO || !sw.deflateInit2_ || !s || flush != Z_FINISH || s->avail_in >= nxz_config()->auto_comp_min) return false;
	int level, wbits, strategy, memlevel;
	if (!nxz_deflate_pristine(s, &level, &wbits, &strategy, &memlevel)) return false;
	z_stream keep = *s;
	nx_deflateEnd(s);
	if (sw.deflateInit2_(s, level, Z_DEFLATED, wbits, memlevel, strategy, ZLIB_VERSION, (int)sizeof(z_stream)) != Z_OK) {
		// (cannot happen with parameters the engine took; back to an engine stream)
		*s = keep; s->state = Z_NULL;
		(void)nx_deflateInit2_(s, level, Z_DEFLATED, wbits, memlevel, strategy, ZLIB_VERSION, (int)sizeof(z_stream));
		return false;
	}
	s->next_in = keep.next_in; s->avail_in = keep.avail_in; s->next_out = keep.next_out; s->avail_out = keep.avail_out;
	return true;
}
static bool auto_to_sw_inflate(z_streamp s, int flush)
{
	if (g_mode_inf != MODE_AUTO || !sw.inflateInit2_ || !s || flush != Z_FINISH || s->avail_in == 0 || s->avail_in >= nxz_config()->auto_dec_min) return false;
	int wbits;
	if (!nxz_inflate_pristine(s, &wbits)) return false;
	z_stream keep = *s;
	nx_inflateEnd(s);
	if (sw.inflateInit2_(s, wbits, ZLIB_VERSION, (int)sizeof(z_stream)) != Z_OK) {
		*s = keep; s->state = Z_NULL;
		(void)nx_inflateInit2_(s, wbits, ZLIB_VERSION, (int)sizeof(z_stream));
		return false;
	}
	s->next_in = keep.next_in; s->avail_in = keep.avail_in; s->next_out = keep.next_out; s->avail_out = keep.avail_out;
	return true;
}

EXPORT int deflate(z_streamp s, int flush)
{
	init();
	bool nx = is_nx(s, MAGIC_DEF);
	if (nx && s->total_in == 0 && auto_to_sw_deflate(s, flush)) nx = false;
	if (!nx && !sw.deflate) return Z_STREAM_ERROR;
	if (!nxz_stats_enabled()) return nx ? nx_deflate(s, flush) : sw.deflate(s, flush);
	const unsigned ai = s ? s->avail_in : 0, ao = s ? s->avail_out : 0;
	const uint64_t t0 = now_ns();
	int rc = nx ? nx_deflate(s, flush) : sw.deflate(s, flush);
	nxz_stats_call(0, nx, ai, ao, now_ns() - t0, rc == Z_OK || rc == Z_STREAM_END);
	return rc;
}
EXPORT int deflateEnd(z_streamp s) { init(); nxz_stats_inc("deflateEnd"); DISPATCH_DEF(nx_deflateEnd(s), sw.deflateEnd(s), Z_STREAM_ERROR); }
EXPORT int deflateReset(z_streamp s) { DISPATCH_DEF(nx_deflateReset(s), sw.deflateReset(s), Z_STREAM_ERROR); }
EXPORT int deflateResetKeep(z_streamp s) { DISPATCH_DEF(nx_deflateResetKeep(s), sw.deflateResetKeep(s), Z_STREAM_ERROR); }
EXPORT int deflateSetHeader(z_streamp s, gz_headerp h) { DISPATCH_DEF(nx_deflateSetHeader(s, h), sw.deflateSetHeader(s, h), Z_STREAM_ERROR); }
EXPORT int deflateSetDictionary(z_streamp s, const Bytef *d, uInt n) { DISPATCH_DEF(nx_deflateSetDictionary(s, d, n), sw.deflateSetDictionary(s, d, n), Z_STREAM_ERROR); }
EXPORT int deflateParams(z_streamp s, int level, int strategy)
{
	init();
	if (is_nx(s, MAGIC_DEF)) return Z_OK;                     // the engine has one speed; accepted and ignored
	return sw.deflateParams ? sw.deflateParams(s, level, strategy) : Z_STREAM_ERROR;
}
EXPORT int deflateCopy(z_streamp d, z_streamp s) { DISPATCH_DEF(nx_deflateCopy(d, s), sw.deflateCopy(d, s), Z_STREAM_ERROR); }
EXPORT uLong deflateBound(z_streamp s, uLong n)
{
	init();
	if (is_nx(s, MAGIC_DEF)) return nx_deflateBound(s, n);
	uLong a = nx_deflateBound(nullptr, n), b = sw.deflateBound ? sw.deflateBound(s, n) : 0;   // lib/nx_compress.c:113-117
	return a > b ? a : b;
}

// ---- inflate ----
EXPORT int inflateInit2_(z_streamp s, int wbits, const char *ver, int size)
{
	const bool nx = want_nx_inf();
	nxz_stats_inc("inflateInit");
	if (nx) {
		int rc = nx_inflateInit2_(s, wbits, ver, size);
		if (rc == Z_OK || g_mode_inf == MODE_NX || !sw.inflateInit2_) return rc;
	}
	return sw.inflateInit2_ ? sw.inflateInit2_(s, wbits, ver, size) : Z_STREAM_ERROR;
}
EXPORT int inflateInit_(z_streamp s, const char *ver, int size) { return inflateInit2_(s, 15, ver, size); }
#define DISPATCH_INF(call_nx, call_sw) do { init(); if (is_nx(s, MAGIC_INF)) return call_nx; return sw.inflate ? call_sw : Z_STREAM_ERROR; } while (0)
EXPORT int inflate(z_streamp s, int flush)
{
	init();
	bool nx = is_nx(s, MAGIC_INF);
	if (nx && s->total_in == 0 && auto_to_sw_inflate(s, flush)) nx = false;
	if (!nx && !sw.inflate) return Z_STREAM_ERROR;
	if (!nxz_stats_enabled()) return nx ? nx_inflate(s, flush) : sw.inflate(s, flush);
	const unsigned ai = s ? s->avail_in : 0, ao = s ? s->avail_out : 0;
	const uint64_t t0 = now_ns();
	int rc = nx ? nx_inflate(s, flush) : sw.inflate(s, flush);
	nxz_stats_call(1, nx, ai, ao, now_ns() - t0, rc == Z_OK || rc == Z_STREAM_END);
	return rc;
}
EXPORT int inflateEnd(z_streamp s) { init(); nxz_stats_inc("inflateEnd"); DISPATCH_INF(nx_inflateEnd(s), sw.inflateEnd(s)); }
EXPORT int inflateReset(z_streamp s) { DISPATCH_INF(nx_inflateReset(s), sw.inflateReset(s)); }
EXPORT int inflateReset2(z_streamp s, int w) { DISPATCH_INF(nx_inflateReset2(s, w), sw.inflateReset2(s, w)); }
EXPORT int inflateSetDictionary(z_streamp s, const Bytef *d, uInt n) { DISPATCH_INF(nx_inflateSetDictionary(s, d, n), sw.inflateSetDictionary(s, d, n)); }
EXPORT int inflateGetHeader(z_streamp s, gz_headerp h) { DISPATCH_INF(nx_inflateGetHeader(s, h), sw.inflateGetHeader(s, h)); }
EXPORT int inflateSyncPoint(z_streamp s) { DISPATCH_INF(nx_inflateSyncPoint(s), (sw.inflateSyncPoint ? sw.inflateSyncPoint(s) : Z_STREAM_ERROR)); }
EXPORT int inflateCopy(z_streamp d, z_streamp s) { DISPATCH_INF(nx_inflateCopy(d, s), (sw.inflateCopy ? sw.inflateCopy(d, s) : Z_STREAM_ERROR)); }
EXPORT int inflateResetKeep(z_streamp s) { DISPATCH_INF(nx_inflateResetKeep(s), (sw.inflateResetKeep ? sw.inflateResetKeep(s) : Z_STREAM_ERROR)); }

// ---- one-shot (lib/nx_compress.c:77-117, lib/nx_uncompr.c:90-149) ----
EXPORT int compress2(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen, int level)
{
	init();
	nxz_stats_inc("compress");
	bool nx = nxz_engine_usable() && (g_mode_def == MODE_NX || (g_mode_def == MODE_AUTO && g_engine && sourceLen >= nxz_config()->auto_comp_min));
	if (nx && slow(g_mode_def, nxz_config()->compress_delay)) nx = false;     // as deflateInit decides (lib/nx_deflate.c:714)
	if (nx) return nx_compress2(dest, destLen, source, sourceLen, level);
	return sw.compress2 ? sw.compress2(dest, destLen, source, sourceLen, level) : Z_STREAM_ERROR;
}
EXPORT int compress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen)
{
	return compress2(dest, destLen, source, sourceLen, Z_DEFAULT_COMPRESSION);
}
EXPORT uLong compressBound(uLong n)
{
	init();
	uLong a = nx_compressBound(n), b = sw.compressBound ? sw.compressBound(n) : 0;
	return a > b ? a : b;
}
EXPORT int uncompress2(Bytef *dest, uLongf *destLen, const Bytef *source, uLong *sourceLen)
{
	init();
	nxz_stats_inc("uncompress");
	bool nx = nxz_engine_usable() && (g_mode_inf == MODE_NX || (g_mode_inf == MODE_AUTO && g_engine && *sourceLen >= nxz_config()->auto_dec_min));
	if (nx && slow(g_mode_inf, nxz_config()->decompress_delay)) nx = false;
	if (nx) return nx_uncompress2(dest, destLen, source, sourceLen);
	if (sw.uncompress2) return sw.uncompress2(dest, destLen, source, sourceLen);
	return sw.uncompress ? sw.uncompress(dest, destLen, source, *sourceLen) : Z_STREAM_ERROR;
}
EXPORT int uncompress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen)
{
	return uncompress2(dest, destLen, source, &sourceLen);
}

// ---- checksums: host code either way (lib/nx_crc.c:437-446 routes crc32 to the vector CRC) ----
EXPORT uLong crc32(uLong crc, const Bytef *buf, uInt len) { return nx_crc32(crc, buf, len); }
EXPORT uLong adler32(uLong adler, const Bytef *buf, uInt len) { return nx_adler32(adler, buf, len); }
EXPORT uLong crc32_z(uLong crc, const Bytef *buf, z_size_t len) { return nx_crc32(crc, buf, len); }
EXPORT uLong adler32_z(uLong adler, const Bytef *buf, z_size_t len) { return nx_adler32(adler, buf, len); }
EXPORT uLong crc32_combine(uLong a, uLong b, z_off_t n) { return nx_crc32_combine(a, b, n); }
EXPORT uLong adler32_combine(uLong a, uLong b, z_off_t n) { return nx_adler32_combine(a, b, n); }
EXPORT uLong crc32_combine64(uLong a, uLong b, z_off64_t n) { return nx_crc32_combine(a, b, (off_t)n); }
EXPORT uLong adler32_combine64(uLong a, uLong b, z_off64_t n) { return nx_adler32_combine(a, b, (off_t)n); }

// ---- gz files (lib/nx_gzlib.c:331-354): the engine's layer or software zlib's, per handle ----
namespace {
std::mutex g_gz_mu;
std::set<void *> g_gz_nx;
bool gz_is_nx(void *f) { std::lock_guard<std::mutex> l(g_gz_mu); return g_gz_nx.count(f) != 0; }
void *gz_track(void *f) { if (f) { std::lock_guard<std::mutex> l(g_gz_mu); g_gz_nx.insert(f); } return f; }
// a file opened for writing follows the deflate mode, one opened for reading the inflate mode
int gz_mode_of(const char *mode) { return mode && (strchr(mode, 'w') || strchr(mode, 'a')) ? g_mode_def : g_mode_inf; }
}
EXPORT gzFile gzopen(const char *path, const char *mode)
{
	init();
	const int m = gz_mode_of(mode);
	if (want_nx(m)) { void *f = gz_track(nx_gzopen(path, mode)); if (f || m == MODE_NX || !sw.gzopen) return (gzFile)f; }
	return sw.gzopen ? sw.gzopen(path, mode) : nullptr;
}
EXPORT gzFile gzdopen(int fd, const char *mode)
{
	init();
	const int m = gz_mode_of(mode);
	if (want_nx(m)) { void *f = gz_track(nx_gzdopen(fd, mode)); if (f || m == MODE_NX || !sw.gzdopen) return (gzFile)f; }
	return sw.gzdopen ? sw.gzdopen(fd, mode) : nullptr;
}
EXPORT int gzread(gzFile f, voidp buf, unsigned len) { init(); return gz_is_nx(f) ? nx_gzread(f, buf, len) : sw.gzread ? sw.gzread(f, buf, len) : -1; }
EXPORT int gzwrite(gzFile f, voidpc buf, unsigned len) { init(); return gz_is_nx(f) ? nx_gzwrite(f, buf, len) : sw.gzwrite ? sw.gzwrite(f, buf, len) : 0; }
EXPORT int gzclose(gzFile f)
{
	init();
	if (gz_is_nx(f)) { { std::lock_guard<std::mutex> l(g_gz_mu); g_gz_nx.erase(f); } return nx_gzclose(f); }
	return sw.gzclose ? sw.gzclose(f) : Z_STREAM_ERROR;
}
