// nxz_gzfile.cpp -- the gz file calls of the nx_ API (reference lib/nx_gzlib.c:68-329): a thin
// layer over nx_deflate / nx_inflate and a file descriptor.
//   nx_gzopen / nx_gzdopen   "w" (+ level digit, 'h' Z_HUFFMAN_ONLY, 'f' Z_FILTERED, 'R' Z_RLE) opens a
//                            gzip writer (windowBits 31), anything else a reader
//   nx_gzwrite               returns the number of uncompressed bytes taken, 0 on error
//   nx_gzread                returns the number of uncompressed bytes delivered (0 at the end / on error)
//   nx_gzclose               finishes the stream, returns the zlib code of the End call
// Unlike the reference, reader/writer is a property of the handle (the reference keeps one global
// flag, lib/nx_gzlib.c:54) and the reader fills a 64 KiB buffer instead of 1..10 bytes per read().
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <cstring>
#include <unistd.h>
#include <algorithm>
#include "../../include/nxz_zlib.h"
#include "nxz_host.h"

// The reference's gz layer keeps this flag in a global that ends up in its ABI (lib/nx_gzlib.c:55,
// test/libnxz.abi): set once a file has been opened for writing.
extern "C" { bool is_deflate = false; }

namespace {

struct GzState {
	int fd = -1;
	FILE *fp = nullptr;
	bool writer = false;
	int err = Z_OK;
	z_stream strm;
	unsigned char *buf = nullptr;      // reader: source buffer
	unsigned used = 0;                 // reader: bytes of buf not yet consumed
	unsigned char *cur = nullptr;
	bool eof = false;                  // reader: the file has been read to its end
	bool done = false;                 // reader: ... and everything in it has been handed out
};
constexpr unsigned RBUF = 1u << 20, CHUNK = 65536;       // (reads of a MiB: nx_inflate decodes what it is handed side by side)

GzState *gz_open(const char *path, int fd, const char *mode)
{
	if (!mode) { errno = EINVAL; return nullptr; }
	GzState *g = new (std::nothrow) GzState();
	if (!g) return nullptr;
	if (path) {
		g->fp = fopen(path, strchr(mode, 'w') ? "wb" : strchr(mode, 'a') ? "ab" : "rb");
		g->fd = g->fp ? fileno(g->fp) : -1;
	} else g->fd = fd;
	if (g->fd < 0) { delete g; return nullptr; }
	memset(&g->strm, 0, sizeof(g->strm));
	int rc;
	if (strchr(mode, 'w') || strchr(mode, 'a')) {
		int strategy = strchr(mode, 'h') ? Z_HUFFMAN_ONLY : strchr(mode, 'f') ? Z_FILTERED : strchr(mode, 'R') ? Z_RLE : Z_DEFAULT_STRATEGY;
		const char *digit = strpbrk(mode, "0123456789");
		int level = digit ? *digit - '0' : Z_DEFAULT_COMPRESSION;
		rc = nx_deflateInit2_(&g->strm, level, Z_DEFLATED, 31, 8, strategy, ZLIB_VERSION, (int)sizeof(z_stream));
		g->writer = true;
		is_deflate = true;
	} else {
		rc = nx_inflateInit2_(&g->strm, 47, ZLIB_VERSION, (int)sizeof(z_stream));
		if (rc == Z_OK && !(g->buf = (unsigned char *)malloc(RBUF))) { nx_inflateEnd(&g->strm); rc = Z_MEM_ERROR; }
	}
	if (rc != Z_OK) {
		if (g->fp) fclose(g->fp);
		if (rc == Z_STREAM_ERROR) errno = EINVAL;
		delete g;
		return nullptr;
	}
	return g;
}

bool write_all(int fd, const unsigned char *p, size_t n)
{
	while (n) {
		ssize_t w = write(fd, p, n);
		if (w < 0) { if (errno == EINTR) continue; return false; }
		p += w; n -= (size_t)w;
	}
	return true;
}

} // namespace

extern "C" void *nx_gzopen(const char *path, const char *mode) { return path ? gz_open(path, -1, mode) : nullptr; }
extern "C" void *nx_gzdopen(int fd, const char *mode) { return gz_open(nullptr, fd, mode); }

extern "C" int nx_gzwrite(void *file, const void *buf, unsigned len)
{
	GzState *g = (GzState *)file;
	if (!g || !g->writer || g->err != Z_OK) return 0;
	if (len == 0) return 0;
	unsigned char *out = (unsigned char *)malloc(CHUNK);
	if (!out) { g->err = Z_MEM_ERROR; return 0; }
	g->strm.next_in = (z_const Bytef *)buf;
	g->strm.avail_in = len;
	while (g->strm.avail_in) {
		g->strm.next_out = out; g->strm.avail_out = CHUNK;
		int rc = nx_deflate(&g->strm, Z_NO_FLUSH);
		if (!write_all(g->fd, out, CHUNK - g->strm.avail_out)) { g->err = Z_ERRNO; break; }
		if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) { g->err = rc; break; }
	}
	free(out);
	return g->err == Z_OK ? (int)len : 0;
}

extern "C" int nx_gzread(void *file, void *buf, unsigned len)
{
	GzState *g = (GzState *)file;
	if (!g || g->writer) return -1;
	if (g->err != Z_OK) return -1;                       // (zlib's gzread: an error stays an error, -1, not "end of file")
	if (len == 0 || g->done) return 0;
	uLong before = g->strm.total_out, produced = 0;
	g->strm.next_out = (Bytef *)buf;
	g->strm.avail_out = len;
	while (g->strm.avail_out) {
		if (g->used == 0 && !g->eof) {
			ssize_t r;
			do r = read(g->fd, g->buf, RBUF); while (r < 0 && errno == EINTR);
			if (r < 0) { g->err = Z_ERRNO; break; }
			g->cur = g->buf; g->used = (unsigned)r;
			if (r == 0) g->eof = true;
		}
		int rc;
		if (g->used == 0) {                                 // end of file: let the stream finish (it may hold output, and source, yet)
			const uInt had = g->strm.avail_out;
			g->strm.next_in = g->buf; g->strm.avail_in = 0;
			rc = nx_inflate(&g->strm, Z_FINISH);
			if (rc != Z_STREAM_END) {
				if (rc != Z_OK && rc != Z_BUF_ERROR) { g->err = rc; g->done = true; break; }
				if (g->strm.avail_out == had) { g->done = true; break; }     // nothing more to come: a truncated file
				continue;
			}
		} else {
			g->strm.next_in = g->cur; g->strm.avail_in = g->used;
			rc = nx_inflate(&g->strm, Z_NO_FLUSH);
			g->cur = (unsigned char *)g->strm.next_in; g->used = g->strm.avail_in;
		}
		if (rc == Z_STREAM_END) {
			// a gzip file is a sequence of members (RFC 1952 2.2; this library's own nxz_gzip and bgzip
			// write one per block): go on with the next one, as zlib's gzread does.  Zero bytes between
			// or behind members are padding.
			produced += g->strm.total_out - before;
			// what the stream took beyond its own end (it gathers small inputs, and hands the engine more than one
			// member's worth) comes back in front of what we still hold
			if (const size_t ung = nxz_inflate_unget_size(&g->strm)) {
				unsigned char *nb = (unsigned char *)malloc(std::max<size_t>(RBUF, ung + g->used));
				if (!nb) { g->err = Z_MEM_ERROR; break; }
				nxz_inflate_take_unget(&g->strm, nb);
				if (g->used) memcpy(nb + ung, g->cur, g->used);
				free(g->buf);
				g->buf = g->cur = nb; g->used += (unsigned)ung;
			}
			while (g->used && *g->cur == 0) { g->cur++; g->used--; }
			if (g->used == 0 && !g->eof) {
				ssize_t r;
				do r = read(g->fd, g->buf, RBUF); while (r < 0 && errno == EINTR);
				if (r < 0) { g->err = Z_ERRNO; break; }
				g->cur = g->buf; g->used = (unsigned)r;
				while (g->used && *g->cur == 0) { g->cur++; g->used--; }
				if (r == 0) g->eof = true;
			}
			Bytef *no = g->strm.next_out; uInt ao = g->strm.avail_out;
			if (nx_inflateReset(&g->strm) != Z_OK) { g->err = Z_STREAM_ERROR; break; }
			g->strm.next_out = no; g->strm.avail_out = ao;
			before = 0;
			if (g->eof && g->used == 0) { g->done = true; break; }
			continue;
		}
		// a damaged member: what this call has made so far -- earlier members' bytes included -- is the caller's (zlib's
		// gzread returns it; the error is what the NEXT call reports), -1 only when there is nothing
		if (rc != Z_OK && rc != Z_BUF_ERROR) { g->err = rc; break; }
	}
	const uLong n = produced + g->strm.total_out - before;
	return n == 0 && g->err != Z_OK ? -1 : (int)n;
}

extern "C" int nx_gzclose(void *file)
{
	GzState *g = (GzState *)file;
	if (!g) { errno = EINVAL; return Z_STREAM_ERROR; }
	int rc;
	if (g->writer) {
		unsigned char *out = (unsigned char *)malloc(CHUNK);
		if (!out) return Z_MEM_ERROR;
		g->strm.next_in = Z_NULL; g->strm.avail_in = 0;
		int r;
		do {
			g->strm.next_out = out; g->strm.avail_out = CHUNK;
			r = nx_deflate(&g->strm, Z_FINISH);
			if (!write_all(g->fd, out, CHUNK - g->strm.avail_out)) { g->err = Z_ERRNO; break; }
		} while (r == Z_OK || r == Z_BUF_ERROR);
		free(out);
		rc = nx_deflateEnd(&g->strm);
		if (r != Z_STREAM_END && g->err == Z_OK) g->err = r < 0 ? r : Z_BUF_ERROR;
	} else rc = nx_inflateEnd(&g->strm);
	if ((g->fp ? fclose(g->fp) : close(g->fd)) != 0 && g->err == Z_OK) g->err = Z_ERRNO;
	if (g->err != Z_OK && rc == Z_OK) rc = g->writer ? g->err : rc;         // a failed write / close is the caller's to know
	free(g->buf);
	delete g;
	return rc;
}
