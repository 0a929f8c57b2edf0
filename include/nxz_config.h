/* nxz_config.h -- run-time configuration and call statistics of the stream layer (libnxz_amd.so)
 * and of the LD_PRELOAD dispatch (libnxz_preload.so).
 *
 * Counterpart of the reference's nx_config / nx_hw_init() (lib/nx_zlib.c:1065-1347: environment
 * variables override the keys of the file named by NX_GZIP_CONFIG, default ./nx-zlib.conf;
 * sample file test/nx-zlib.conf), of str_to_num (lib/nx_zlib.c:849-869), and of struct zlib_stats
 * / print_stats (lib/nx_zlib.h:560-603, lib/nx_zlib.c:876-955; gathered when trace bit 0x8 is
 * set, inc_nx/nx_dbg.h:53-57, printed when the library is unloaded, lib/nx_zlib.c:1381-1391).
 *
 *   variable                 file key        meaning
 *   NX_GZIP_TYPE_SELECTOR    nx_selector     0 auto, 1 software zlib, 2 engine, 3 engine deflate + zlib inflate
 *   NX_GZIP_COMP_MODE        comp_mode       0/1/2 for deflate only  (ignored when the selector is set)
 *   NX_GZIP_DEC_MODE         dec_mode        0/1/2 for inflate only  (ignored when the selector is set)
 *   NX_GZIP_STRATEGY         strategy        0 = fixed Huffman always, 1 = dynamic unless Z_FIXED (default)
 *   NX_GZIP_DHT_CONFIG       dht_config      bit 0: table-cache keys from literals and lengths (default literals only)
 *   NX_GZIP_TRACE            trace           bit 0x8: gather and print statistics
 *   NX_GZIP_VERBOSE          verbose         0 errors, 1 warnings, 2 info (to the log file)
 *   NX_GZIP_LOGFILE          logfile         default /tmp/nx.log (opened only when something is logged)
 *   NX_GZIP_DEV_NUM          dev_num         device ordinal, -1 = current/default (NXZ_DEVICE still works)
 *   NX_GZIP_DEF_BUF_SIZE     def_buf_size    accepted (64 KiB..8 MiB, KiB/MiB/GiB suffixes); the stream layer sizes its buffers on demand
 *   file only                compress_delay, decompress_delay, delay_threshold (both): AUTO mode's "device is slow" rule
 *   (keys of the POWER transport -- mlock_csb, timeout_pgfaults, max_vas_reuse_count,
 *    soft_copy_threshold -- are read and ignored; cache_threshold is honoured, <= one page)
 */
#ifndef NXZ_CONFIG_H
#define NXZ_CONFIG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { NXZ_MODE_AUTO = 0, NXZ_MODE_SW = 1, NXZ_MODE_NX = 2 };     /* GZIP_AUTO/SW/NX, lib/nx_zlib.h:376-381 */

typedef struct nxz_config {
	int      verbose;
	int      trace;
	int      dht;                 /* nx_config.dht */
	int      strategy_override;   /* nx_config.strategy_override */
	int      dev_num;
	int      mode_deflate, mode_inflate;
	uint32_t def_buf_size;
	uint32_t cache_threshold;
	uint64_t auto_comp_min, auto_dec_min;        /* AUTO mode: a one-shot call, or the first deflate() / inflate() call of a
						      * stream, with fewer input bytes than this is served by software zlib
						      * (NX_GZIP_AUTO_COMP_MIN / NX_GZIP_AUTO_DEC_MIN, keys auto_comp_min /
						      * auto_dec_min; the reference's rule is a fixed 1024 bytes, lib/nx_zlib.h:88-89) */
	uint64_t compress_delay, decompress_delay;   /* AUTO mode: average job delay (512 MHz ticks) above which new
						      * streams go to software zlib (lib/nx_zlib.c:1121-1122,1306-1318) */
	char     logfile[256];
	char     cfgfile[256];
	int      cfgfile_loaded;      /* 1 when the file was read and had at least one key */
} nxz_config_t;

#define NXZ_STAT_SLOTS 256        /* ZLIB_SIZE_SLOTS: 4 KiB buckets of avail_in / avail_out */
typedef struct nxz_stats {
	uint64_t deflateInit, deflate, deflate_sw, deflate_nx, deflateBound, deflateEnd, compress;
	uint64_t inflateInit, inflate, inflate_sw, inflate_nx, inflateEnd, uncompress;
	uint64_t deflate_len, deflate_ns, inflate_len, inflate_ns;
	uint64_t deflate_avail_in[NXZ_STAT_SLOTS], deflate_avail_out[NXZ_STAT_SLOTS];
	uint64_t inflate_avail_in[NXZ_STAT_SLOTS], inflate_avail_out[NXZ_STAT_SLOTS];
} nxz_stats_t;

const nxz_config_t *nxz_config(void);          /* parsed once, on first use */
void     nxz_config_reload(void);               /* parse environment and file again (tests) */
uint64_t nxz_str_to_num(const char *s);         /* "64KiB" -> 65536; UINT64_MAX when the suffix is unknown */

/* the engine's average job delay, in 512 MHz ticks (lib/nx_zlib.c:1487-1511 nx_device_stats: exponential
 * moving average, decay 4; lib/nx_zlib.h:443-449 decrease_delay: streams served in software let it fade) */
void     nxz_device_stats(uint64_t start_ticks, uint64_t end_ticks);
uint64_t nxz_avg_delay(void);
void     nxz_decrease_delay(void);
void     nxz_set_avg_delay(uint64_t ticks);     /* tests */
uint64_t nxz_ticks(void);                       /* the 512 MHz timebase the thresholds are written for */

int      nxz_stats_enabled(void);               /* trace & 0x8 */
void     nxz_stats_get(nxz_stats_t *out);
void     nxz_stats_reset(void);
void     nxz_stats_print(void);                 /* to the log file, in the reference's format */
/* one call of deflate()/inflate(): which = 0 deflate, 1 inflate; engine = 1 nx, 0 software;
 * avail_in/avail_out as they were before the call (avail_in is also what the *_len totals add up) */
void     nxz_stats_call(int which, int engine, unsigned avail_in, unsigned avail_out, uint64_t ns, int ok);
/* the other counters: name is one of "deflateInit", "deflateEnd", "deflateBound", "compress",
 * "inflateInit", "inflateEnd", "uncompress" */
void     nxz_stats_inc(const char *name);
void     nxz_log(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#ifdef __cplusplus
}
#endif
#endif
