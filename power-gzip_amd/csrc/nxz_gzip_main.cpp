// nxz_gzip -- gzip-compatible command line tool on the MI355X engine.
//
// Counterpart of the reference's samples/nx_gzip.c (same option letters where they apply).
// Compression cuts the input into 65 280-byte blocks, compresses them as GPU batches and writes
// one gzip member per block (include/nxz_blocked.h): every gzip reader accepts the file, and this
// tool -- or bgzip/htslib -- decompresses the members in parallel.  Decompression of ordinary
// gzip files (one member, any producer) goes through the stream layer (nx_inflate).
#include <errno.h>
#include <fcntl.h>
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <string>
#include <vector>
#include "../../include/nxz_blocked.h"
#include "../../include/nxz_engine.h"
#include "../../include/nxz_zlib.h"

namespace {

struct Opts {
	bool to_stdout = false, decompress = false, force = false, keep = false, list = false, quiet = false, test = false;
	int verbose = 0;
	bool fixed = false;
	std::string suffix = ".gz";
};

struct Input {
	const uint8_t *p = nullptr; size_t len = 0;
	void *map = nullptr; std::vector<uint8_t> buf;
	~Input() { if (map) munmap(map, len); }
};

bool read_input(const char *path, Input &in)
{
	int fd = path ? open(path, O_RDONLY) : 0;
	if (fd < 0) { fprintf(stderr, "nxz_gzip: %s: %s\n", path, strerror(errno)); return false; }
	struct stat st;
	if (path && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
		void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
		if (m != MAP_FAILED) { in.map = m; in.p = (const uint8_t *)m; in.len = (size_t)st.st_size; close(fd); return true; }
	}
	uint8_t tmp[1 << 16];
	for (;;) {
		ssize_t n = read(fd, tmp, sizeof(tmp));
		if (n < 0) { if (errno == EINTR) continue; fprintf(stderr, "nxz_gzip: read: %s\n", strerror(errno)); if (path) close(fd); return false; }
		if (n == 0) break;
		in.buf.insert(in.buf.end(), tmp, tmp + n);
	}
	if (path) close(fd);
	in.p = in.buf.data(); in.len = in.buf.size();
	return true;
}

struct Out { FILE *f = nullptr; uint64_t bytes = 0; };
int sink_file(void *user, const void *buf, size_t len)
{
	Out *o = (Out *)user;
	o->bytes += len;
	if (!o->f) return 0;                                  // -t
	return fwrite(buf, 1, len, o->f) == len ? 0 : 1;
}

// ordinary gzip / zlib data through the stream layer, member after member (samples/nx_gzip.c:150-210)
int stream_inflate(const uint8_t *p, size_t len, Out &out)
{
	z_stream s;
	memset(&s, 0, sizeof(s));
	if (nx_inflateInit2_(&s, 47, ZLIB_VERSION, (int)sizeof(s)) != Z_OK) { fprintf(stderr, "nxz_gzip: cannot open the engine\n"); return -1; }
	std::vector<uint8_t> o(64u << 20);
	size_t pos = 0;
	int rc = Z_OK;
	while (pos < len) {
		const size_t take = len - pos < (1u << 30) ? len - pos : (1u << 30);
		const int flush = pos + take == len ? Z_SYNC_FLUSH : Z_NO_FLUSH;   // the last piece: nothing may stay cached
		s.next_in = (Bytef *)(p + pos); s.avail_in = (uInt)take;
		for (;;) {
			const uInt in0 = s.avail_in;
			s.next_out = o.data(); s.avail_out = (uInt)o.size();
			rc = nx_inflate(&s, flush);
			if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) { nx_inflateEnd(&s); return -1; }
			const size_t got = o.size() - s.avail_out;
			if (sink_file(&out, o.data(), got)) { nx_inflateEnd(&s); return -2; }
			if (rc == Z_STREAM_END) {
				bool pad = true;                                   // only zero padding may follow the last member
				for (uInt i = 0; i < s.avail_in; i++) if (s.next_in[i]) { pad = false; break; }
				if (pad) { s.avail_in = 0; break; }
				nx_inflateReset(&s);                               // another member follows
				rc = Z_OK;
				continue;
			}
			if (got == 0 && s.avail_in == in0) break;              // no progress: needs more input
		}
		pos += take - s.avail_in;
		if (rc == Z_STREAM_END) break;
	}
	nx_inflateEnd(&s);
	return rc == Z_STREAM_END ? 0 : -1;
}

int do_file(const char *path, const Opts &op)
{
	Input in;
	if (!read_input(path, in)) return 1;
	const bool from_stdin = path == nullptr;
	std::string outname;
	Out out;
	if (op.list) {
		uint64_t members = 0, usize = 0; size_t used = 0;
		nxz_blocked_scan(in.p, in.len, &members, &usize, &used);
		if (used != in.len || !members) {                    // ordinary gzip: ISIZE of the last member
			usize = in.len >= 4 ? in.p[in.len - 4] | (uint64_t)in.p[in.len - 3] << 8 | (uint64_t)in.p[in.len - 2] << 16 | (uint64_t)in.p[in.len - 1] << 24 : 0;
			members = 1;
		}
		printf("%12s %12s %8s %6s %s\n", "compressed", "uncompressed", "members", "ratio", "name");
		printf("%12zu %12llu %8llu %5.1f%% %s\n", in.len, (unsigned long long)usize, (unsigned long long)members,
		       usize ? 100.0 * (1.0 - (double)in.len / (double)usize) : 0.0, path ? path : "stdin");
		return 0;
	}
	if (!op.test) {
		if (op.to_stdout || from_stdin) out.f = stdout;
		else {
			if (op.decompress) {
				std::string p = path;
				if (p.size() > op.suffix.size() && p.compare(p.size() - op.suffix.size(), op.suffix.size(), op.suffix) == 0) outname = p.substr(0, p.size() - op.suffix.size());
				else { if (!op.quiet) fprintf(stderr, "nxz_gzip: %s: unknown suffix -- ignored\n", path); return 2; }
			} else {
				outname = std::string(path) + op.suffix;
			}
			if (!op.force && access(outname.c_str(), F_OK) == 0) { fprintf(stderr, "nxz_gzip: %s already exists; not overwritten\n", outname.c_str()); return 1; }
			out.f = fopen(outname.c_str(), "wb");
			if (!out.f) { fprintf(stderr, "nxz_gzip: %s: %s\n", outname.c_str(), strerror(errno)); return 1; }
		}
	}
	int rc = 0;
	nxz_blocked_opts_t bo;
	memset(&bo, 0, sizeof(bo));
	bo.device = -1;
	bo.fixed = op.fixed;
	if (op.decompress || op.test) {
		size_t used = 0;
		uint64_t n = 0;
		rc = nxz_blocked_inflate(in.p, in.len, &bo, sink_file, &out, &n, &used);
		if (rc == 1) { used = 0; rc = 0; }
		if (rc == 0 && used < in.len) rc = stream_inflate(in.p + used, in.len - used, out);
		if (rc) { fprintf(stderr, "nxz_gzip: %s: invalid compressed data%s\n", path ? path : "stdin", rc == -EILSEQ ? " -- crc or length error" : ""); rc = 1; }
	} else {
		rc = nxz_blocked_deflate(in.p, in.len, &bo, sink_file, &out, nullptr);
		if (!rc) rc = nxz_blocked_end_marker(sink_file, &out);
		if (rc) { fprintf(stderr, "nxz_gzip: %s: compression failed (%d): %s\n", path ? path : "stdin", rc, nxz_last_error()); rc = 1; }
	}
	if (out.f && out.f != stdout) { if (fclose(out.f)) rc = 1; }
	else if (out.f) fflush(out.f);
	if (rc) { if (!outname.empty()) unlink(outname.c_str()); return 1; }
	if (op.verbose) {
		const double a = (double)in.len, b = (double)out.bytes;
		fprintf(stderr, "%s:\t%5.1f%%%s%s\n", path ? path : "stdin", op.decompress || op.test ? (b ? 100.0 * (1.0 - a / b) : 0.0) : (a ? 100.0 * (1.0 - b / a) : 0.0),
			outname.empty() ? "" : " -- replaced with ", outname.c_str());
	}
	if (!outname.empty() && !op.keep) unlink(path);
	return 0;
}

void usage(FILE *fp)
{
	fprintf(fp, "Usage: nxz_gzip [OPTION]... [FILE]...\n"
		"Compress or uncompress FILEs on the MI355X DEFLATE engine (by default, compress FILEs in place).\n\n"
		"  -c, --stdout      write on standard output, keep original files unchanged\n"
		"  -d, --decompress  decompress\n"
		"  -f, --force       force overwrite of output file\n"
		"  -h, --help        give this help\n"
		"  -k, --keep        keep (don't delete) input files\n"
		"  -l, --list        list compressed file contents\n"
		"  -q, --quiet       suppress all warnings\n"
		"  -S, --suffix=SUF  use suffix SUF on compressed files\n"
		"  -t, --test        test compressed file integrity\n"
		"  -v, --verbose     verbose mode\n"
		"  -V, --version     display version number\n"
		"  -F, --fixed       fixed Huffman codes only (default: dynamic, one table per 64 blocks)\n"
		"  -1 .. -9          accepted; the engine has one speed\n\n"
		"With no FILE, or when FILE is -, read standard input.\n"
		"Output is a multi-member gzip file, one member per 65280-byte block (BGZF layout).\n");
}

} // namespace

int main(int argc, char **argv)
{
	Opts op;
	static const struct option lo[] = {
		{"stdout", 0, 0, 'c'}, {"decompress", 0, 0, 'd'}, {"force", 0, 0, 'f'}, {"help", 0, 0, 'h'}, {"keep", 0, 0, 'k'},
		{"list", 0, 0, 'l'}, {"quiet", 0, 0, 'q'}, {"suffix", 1, 0, 'S'}, {"test", 0, 0, 't'}, {"verbose", 0, 0, 'v'},
		{"version", 0, 0, 'V'}, {"fixed", 0, 0, 'F'}, {"fast", 0, 0, '1'}, {"best", 0, 0, '9'}, {0, 0, 0, 0}};
	int ch;
	while ((ch = getopt_long(argc, argv, "cdfhklqS:tvVF123456789", lo, nullptr)) != -1) {
		switch (ch) {
		case 'c': op.to_stdout = true; break;
		case 'd': op.decompress = true; break;
		case 'f': op.force = true; break;
		case 'k': op.keep = true; break;
		case 'l': op.list = true; break;
		case 'q': op.quiet = true; break;
		case 'S': op.suffix = optarg; break;
		case 't': op.test = true; break;
		case 'v': op.verbose++; break;
		case 'F': op.fixed = true; break;
		case 'V': printf("nxz_gzip (%s)\n", nxz_engine_version()); return 0;
		case 'h': usage(stdout); return 0;
		case '1': case '2': case '3': case '4': case '5': case '6': case '7': case '8': case '9': break;
		default: usage(stderr); return 2;
		}
	}
	int worst = 0;
	if (optind >= argc) return do_file(nullptr, op);
	for (int i = optind; i < argc; i++) {
		int rc = do_file(strcmp(argv[i], "-") ? argv[i] : nullptr, op);
		if (rc > worst) worst = rc;
	}
	return worst;
}
