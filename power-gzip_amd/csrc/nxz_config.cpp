// nxz_config.cpp -- configuration (environment + key=value file), log file and call statistics of
// the stream layer.  See include/nxz_config.h for the reference lines each part stands in for.
#include "../../include/nxz_config.h"
#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <map>
#include <mutex>
#include <string>

namespace {

std::mutex g_cfg_mtx, g_stat_mtx, g_log_mtx;
nxz_config_t g_cfg;
bool g_cfg_ready = false;
nxz_stats_t g_stats;
FILE *g_log = nullptr;
bool g_log_tried = false;

std::string trimmed(const char *b, const char *e)
{
	while (b < e && isspace((unsigned char)*b)) b++;
	while (e > b && isspace((unsigned char)e[-1])) e--;
	return std::string(b, e);
}

// key = value lines; '#' starts a comment; a repeated key keeps its last value (lib/nx_utils.c:185-250)
bool read_cfg_file(const char *path, std::map<std::string, std::string> &kv)
{
	FILE *f = fopen(path, "r");
	if (!f) return false;
	char line[1024];
	while (fgets(line, sizeof(line), f)) {
		if (char *h = strchr(line, '#')) *h = 0;
		char *eq = strchr(line, '=');
		if (!eq) continue;
		std::string k = trimmed(line, eq), v = trimmed(eq + 1, eq + 1 + strlen(eq + 1));
		if (k.empty() || v.empty() || k.size() >= 64 || v.size() >= 1024) continue;
		kv[k] = v;
	}
	fclose(f);
	return !kv.empty();
}

void load_locked()
{
	nxz_config_t c;
	memset(&c, 0, sizeof(c));
	c.strategy_override = 1;                       // dynamic Huffman unless the caller asks for Z_FIXED (:1129)
	c.dev_num = -1;
	c.def_buf_size = 1u << 20;                     // lib/nx_zlib.c:1115
	c.cache_threshold = 8192;                      // :1116
	c.mode_deflate = c.mode_inflate = NXZ_MODE_AUTO;
	// AUTO mode's break-even, measured on MI355X against system zlib on the host's cores with the reference's own
	// harness shape (tools/api_sweep.py, profiles/r03_api_sweep.txt): below these sizes a call costs its fixed
	// launches and copies (0.2 - 0.3 ms) and software zlib is faster
	c.auto_comp_min = 128u << 10;
	c.auto_dec_min = 1u << 20;
	c.decompress_delay = 17000000;                 // lib/nx_zlib.c:1121-1122 (timebase ticks)
	c.compress_delay = 100000000;

	const char *cfgfile = getenv("NX_GZIP_CONFIG");
	if (!cfgfile) cfgfile = "./nx-zlib.conf";
	snprintf(c.cfgfile, sizeof(c.cfgfile), "%s", cfgfile);
	std::map<std::string, std::string> kv;
	c.cfgfile_loaded = read_cfg_file(cfgfile, kv) ? 1 : 0;

	// the environment wins over the file (:1139-1181)
	auto get = [&](const char *env, const char *key) -> const char * {
		const char *e = env ? getenv(env) : nullptr;
		if (e) return e;
		auto it = kv.find(key);
		return it == kv.end() ? nullptr : it->second.c_str();
	};
	const char *logfile = get("NX_GZIP_LOGFILE", "logfile");
	snprintf(c.logfile, sizeof(c.logfile), "%s", logfile ? logfile : "/tmp/nx.log");

	const char *sel = get("NX_GZIP_TYPE_SELECTOR", "nx_selector");
	if (sel) {                                     // :1190-1203
		unsigned v = (unsigned)(uint8_t)nxz_str_to_num(sel);
		if (v < 3) c.mode_deflate = c.mode_inflate = (int)v;
		else if (v == 3) { c.mode_deflate = NXZ_MODE_NX; c.mode_inflate = NXZ_MODE_SW; }
	} else {                                       // :1204-1217
		if (const char *m = get("NX_GZIP_COMP_MODE", "comp_mode")) { unsigned v = (unsigned)(uint8_t)nxz_str_to_num(m); c.mode_deflate = v < 3 ? (int)v : NXZ_MODE_AUTO; }
		if (const char *m = get("NX_GZIP_DEC_MODE", "dec_mode")) { unsigned v = (unsigned)(uint8_t)nxz_str_to_num(m); c.mode_inflate = v < 3 ? (int)v : NXZ_MODE_AUTO; }
	}
	if (const char *t = get("NX_GZIP_TRACE", "trace")) c.trace = (int)strtol(t, nullptr, 0);
	if (const char *v = get("NX_GZIP_VERBOSE", "verbose")) c.verbose = (int)(nxz_str_to_num(v) & 0xff);
	if (const char *b = get("NX_GZIP_DEF_BUF_SIZE", "def_buf_size")) {         // 64 KiB .. 8 MiB (:1253-1262)
		uint64_t sz = nxz_str_to_num(b);
		if (sz > (1ull << 23)) sz = 1ull << 23;
		else if (sz < 65536) sz = 65536;
		c.def_buf_size = (uint32_t)sz;
	}
	if (const char *s = get("NX_GZIP_STRATEGY", "strategy")) {                // :1265-1271
		uint64_t v = nxz_str_to_num(s);
		c.strategy_override = (v == 0 || v == 1) ? (int)v : 0;
	}
	if (const char *d = get("NX_GZIP_DHT_CONFIG", "dht_config")) c.dht = (int)nxz_str_to_num(d);
	if (const char *d = get("NX_GZIP_DEV_NUM", "dev_num")) c.dev_num = atoi(d);
	if (const char *v = get("NX_GZIP_AUTO_COMP_MIN", "auto_comp_min")) c.auto_comp_min = nxz_str_to_num(v);
	if (const char *v = get("NX_GZIP_AUTO_DEC_MIN", "auto_dec_min")) c.auto_dec_min = nxz_str_to_num(v);
	if (const char *t = get(nullptr, "cache_threshold")) {                     // file only (:1166, 1295-1299)
		uint64_t v = nxz_str_to_num(t);
		long pg = sysconf(_SC_PAGESIZE);
		if (v > (uint64_t)pg) v = (uint64_t)pg;
		c.cache_threshold = (uint32_t)v;
	}
	if (const char *d = get(nullptr, "delay_threshold")) c.compress_delay = c.decompress_delay = nxz_str_to_num(d);   // :1306-1318
	else {
		if (const char *d = get(nullptr, "decompress_delay")) c.decompress_delay = nxz_str_to_num(d);
		if (const char *d = get(nullptr, "compress_delay")) c.compress_delay = nxz_str_to_num(d);
	}
	g_cfg = c;
	g_cfg_ready = true;
}

FILE *logfile_locked()
{
	if (g_log || g_log_tried) return g_log;
	g_log_tried = true;
	const nxz_config_t *c = nxz_config();
	g_log = fopen(c->logfile, "a+");               // open_logfile, lib/nx_zlib.c:957-992
	if (g_log) chmod(c->logfile, 0666);
	else if ((g_log = fopen("/tmp/nx.log", "a+"))) chmod("/tmp/nx.log", 0666);
	return g_log;
}

uint64_t *counter(const char *name)
{
	static const struct { const char *n; uint64_t nxz_stats_t::*p; } tab[] = {
		{"deflateInit", &nxz_stats_t::deflateInit}, {"deflateEnd", &nxz_stats_t::deflateEnd},
		{"deflateBound", &nxz_stats_t::deflateBound}, {"compress", &nxz_stats_t::compress},
		{"inflateInit", &nxz_stats_t::inflateInit}, {"inflateEnd", &nxz_stats_t::inflateEnd},
		{"uncompress", &nxz_stats_t::uncompress},
	};
	for (auto &t : tab) if (!strcmp(t.n, name)) return &(g_stats.*(t.p));
	return nullptr;
}

struct AtExit { ~AtExit() { if (nxz_stats_enabled()) nxz_stats_print(); std::lock_guard<std::mutex> g(g_log_mtx); if (g_log) { fclose(g_log); g_log = nullptr; } } } g_at_exit;

} // namespace

extern "C" uint64_t nxz_str_to_num(const char *str)
{
	if (!str) return 0;
	char *s = nullptr;
	uint64_t num = strtoull(str, &s, 0);
	if (*s == 0) return num;
	if (!strcmp(s, "KiB")) return num * 1024;
	if (!strcmp(s, "MiB")) return num * 1024 * 1024;
	if (!strcmp(s, "GiB")) return num * 1024 * 1024 * 1024;
	return UINT64_MAX;
}

extern "C" const nxz_config_t *nxz_config(void)
{
	std::lock_guard<std::mutex> g(g_cfg_mtx);
	if (!g_cfg_ready) load_locked();
	return &g_cfg;
}

extern "C" void nxz_config_reload(void)
{
	{ std::lock_guard<std::mutex> g(g_cfg_mtx); load_locked(); }
	std::lock_guard<std::mutex> g(g_log_mtx);
	if (g_log) { fclose(g_log); g_log = nullptr; }
	g_log_tried = false;
}

extern "C" int nxz_stats_enabled(void) { return (nxz_config()->trace & 0x8) != 0; }

extern "C" void nxz_stats_get(nxz_stats_t *out)
{
	std::lock_guard<std::mutex> g(g_stat_mtx);
	*out = g_stats;
}

extern "C" void nxz_stats_reset(void)
{
	std::lock_guard<std::mutex> g(g_stat_mtx);
	memset(&g_stats, 0, sizeof(g_stats));
}

extern "C" void nxz_stats_inc(const char *name)
{
	if (!nxz_stats_enabled()) return;
	std::lock_guard<std::mutex> g(g_stat_mtx);
	if (uint64_t *c = counter(name)) ++*c;
}

// lib/nx_deflate.c:2472-2520, lib/nx_inflate.c:2264-2310: sizes are bucketed in 4 KiB slots and the
// call is counted only when it returned Z_OK or Z_STREAM_END
extern "C" void nxz_stats_call(int which, int engine, unsigned avail_in, unsigned avail_out, uint64_t ns, int ok)
{
	if (!ok || !nxz_stats_enabled()) return;
	unsigned si = avail_in / 4096, so = avail_out / 4096;
	if (si >= NXZ_STAT_SLOTS) si = NXZ_STAT_SLOTS - 1;
	if (so >= NXZ_STAT_SLOTS) so = NXZ_STAT_SLOTS - 1;
	std::lock_guard<std::mutex> g(g_stat_mtx);
	if (which == 0) {
		g_stats.deflate_avail_in[si]++; g_stats.deflate_avail_out[so]++; g_stats.deflate++;
		if (engine) g_stats.deflate_nx++; else g_stats.deflate_sw++;
		g_stats.deflate_len += avail_in; g_stats.deflate_ns += ns;
	} else {
		g_stats.inflate_avail_in[si]++; g_stats.inflate_avail_out[so]++; g_stats.inflate++;
		if (engine) g_stats.inflate_nx++; else g_stats.inflate_sw++;
		g_stats.inflate_len += avail_in; g_stats.inflate_ns += ns;
	}
}

extern "C" void nxz_log(int level, const char *fmt, ...)
{
	if (level > nxz_config()->verbose) return;
	std::lock_guard<std::mutex> g(g_log_mtx);
	FILE *f = logfile_locked();
	if (!f) return;
	fprintf(f, "[%d] ", (int)getpid());
	va_list ap;
	va_start(ap, fmt);
	vfprintf(f, fmt, ap);
	va_end(ap);
	fflush(f);
}

// the lines of print_stats (lib/nx_zlib.c:876-955)
extern "C" void nxz_stats_print(void)
{
	nxz_stats_t s;
	nxz_stats_get(&s);
	std::lock_guard<std::mutex> g(g_log_mtx);
	FILE *f = logfile_locked();
	if (!f) return;
	fprintf(f, "API call statistic:\n");
	fprintf(f, "deflateInit: %llu\n", (unsigned long long)s.deflateInit);
	fprintf(f, "deflate: %llu\n", (unsigned long long)s.deflate);
	fprintf(f, "\tdeflate(sw): %llu\n", (unsigned long long)s.deflate_sw);
	fprintf(f, "\tdeflate(nx): %llu\n", (unsigned long long)s.deflate_nx);
	for (int i = 0; i < NXZ_STAT_SLOTS; i++) if (s.deflate_avail_in[i]) fprintf(f, "  deflate_avail_in %4i KiB: %llu\n", (i + 1) * 4, (unsigned long long)s.deflate_avail_in[i]);
	for (int i = 0; i < NXZ_STAT_SLOTS; i++) if (s.deflate_avail_out[i]) fprintf(f, "  deflate_avail_out %4i KiB: %llu\n", (i + 1) * 4, (unsigned long long)s.deflate_avail_out[i]);
	fprintf(f, "deflateBound: %llu\n", (unsigned long long)s.deflateBound);
	fprintf(f, "deflateEnd: %llu\n", (unsigned long long)s.deflateEnd);
	fprintf(f, "compress: %llu\n", (unsigned long long)s.compress);
	fprintf(f, "inflateInit: %llu\n", (unsigned long long)s.inflateInit);
	fprintf(f, "inflate: %llu\n", (unsigned long long)s.inflate);
	fprintf(f, "\tinflate(sw): %llu\n", (unsigned long long)s.inflate_sw);
	fprintf(f, "\tinflate(nx): %llu\n", (unsigned long long)s.inflate_nx);
	for (int i = 0; i < NXZ_STAT_SLOTS; i++) if (s.inflate_avail_in[i]) fprintf(f, "  inflate_avail_in %4i KiB: %llu\n", (i + 1) * 4, (unsigned long long)s.inflate_avail_in[i]);
	for (int i = 0; i < NXZ_STAT_SLOTS; i++) if (s.inflate_avail_out[i]) fprintf(f, "  inflate_avail_out %4i KiB: %llu\n", (i + 1) * 4, (unsigned long long)s.inflate_avail_out[i]);
	fprintf(f, "inflateEnd: %llu\n", (unsigned long long)s.inflateEnd);
	fprintf(f, "uncompress: %llu\n", (unsigned long long)s.uncompress);
	double ds = (double)s.deflate_ns * 1e-9, is = (double)s.inflate_ns * 1e-9;
	fprintf(f, "deflate data length: %llu KiB\n", (unsigned long long)(s.deflate_len / 1024));
	fprintf(f, "deflate time: %1.2f secs\n", ds);
	fprintf(f, "deflate rate: %1.2f MiB/s\n", ds > 0 ? (double)(s.deflate_len / (1024 * 1024)) / ds : 0.0);
	fprintf(f, "inflate data length: %llu KiB\n", (unsigned long long)(s.inflate_len / 1024));
	fprintf(f, "inflate time: %1.2f secs\n", is);
	fprintf(f, "inflate rate: %1.2f MiB/s\n", is > 0 ? (double)(s.inflate_len / (1024 * 1024)) / is : 0.0);
	fflush(f);
}

// ---- the engine's average job delay (AUTO mode's "is the device slow" input) ----
#include <atomic>
#include <time.h>
static std::atomic<uint64_t> g_avg_delay{0};
extern "C" uint64_t nxz_ticks(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (uint64_t)ts.tv_sec * 512000000ull + (uint64_t)ts.tv_nsec * 512ull / 1000ull;
}
extern "C" void nxz_device_stats(uint64_t start, uint64_t end)
{
	const uint64_t tps = 512000000ull, last = end - start;
	if (last > tps || last < tps / 10000000ull) return;           // the process slept, or the clock is broken (:1493-1497)
	uint64_t d = g_avg_delay.load(std::memory_order_relaxed);
	if (d == 0) d = last;
	g_avg_delay.store((last + 4 * d) / 5, std::memory_order_relaxed);
}
extern "C" uint64_t nxz_avg_delay(void) { return g_avg_delay.load(std::memory_order_relaxed); }
extern "C" void nxz_decrease_delay(void)
{
	const uint64_t d = g_avg_delay.load(std::memory_order_relaxed);
	g_avg_delay.store(d - d / 4, std::memory_order_relaxed);
}
extern "C" void nxz_set_avg_delay(uint64_t t) { g_avg_delay.store(t, std::memory_order_relaxed); }
