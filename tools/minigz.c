/* minigz.c -- a plain zlib client for the interoperability matrix (tests/test_gpu_oct.py): stdin -> stdout
 * through the UNPREFIXED zlib API, nothing of this repository in it.  Under LD_PRELOAD=libnxz_preload.so its
 * calls land in the engine; without, in system zlib.  The two clients of the reference's oct/ matrix in one
 * (oct/generate-test.sh:11-30): minigzip (gzip files through the gz* calls) and minideflate (zlib streams
 * through deflate() / inflate() with fixed-size buffers).
 *   minigz [-d] [-1 .. -9] [-z]      -z: zlib format with deflate()/inflate(); default: gzip through gzdopen()
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define CHUNK (256 * 1024)
static unsigned char in[CHUNK], out[CHUNK];

static int zdeflate(int level)
{
	z_stream s;
	memset(&s, 0, sizeof(s));
	if (deflateInit(&s, level) != Z_OK) return 1;
	int flush;
	do {
		s.avail_in = (uInt)fread(in, 1, CHUNK, stdin);
		if (ferror(stdin)) return 1;
		flush = feof(stdin) ? Z_FINISH : Z_NO_FLUSH;
		s.next_in = in;
		do {
			s.avail_out = CHUNK; s.next_out = out;
			if (deflate(&s, flush) == Z_STREAM_ERROR) return 1;
			size_t have = CHUNK - s.avail_out;
			if (fwrite(out, 1, have, stdout) != have) return 1;
		} while (s.avail_out == 0);
		if (s.avail_in != 0) return 1;
	} while (flush != Z_FINISH);
	return deflateEnd(&s) == Z_OK ? 0 : 1;
}

static int zinflate(void)
{
	z_stream s;
	memset(&s, 0, sizeof(s));
	if (inflateInit(&s) != Z_OK) return 1;
	int rc = Z_OK;
	do {
		s.avail_in = (uInt)fread(in, 1, CHUNK, stdin);
		if (ferror(stdin)) return 1;
		if (s.avail_in == 0) break;
		s.next_in = in;
		do {
			s.avail_out = CHUNK; s.next_out = out;
			rc = inflate(&s, Z_NO_FLUSH);
			if (rc == Z_NEED_DICT || rc == Z_DATA_ERROR || rc == Z_MEM_ERROR || rc == Z_STREAM_ERROR) return 1;
			size_t have = CHUNK - s.avail_out;
			if (fwrite(out, 1, have, stdout) != have) return 1;
		} while (s.avail_out == 0);
	} while (rc != Z_STREAM_END);
	inflateEnd(&s);
	return rc == Z_STREAM_END ? 0 : 1;
}

static int gzcomp(int level)
{
	char mode[8];
	snprintf(mode, sizeof(mode), "wb%d", level);
	gzFile g = gzdopen(1, mode);
	if (!g) return 1;
	size_t n;
	while ((n = fread(in, 1, CHUNK, stdin)) > 0)
		if (gzwrite(g, in, (unsigned)n) != (int)n) return 1;
	return gzclose(g) == Z_OK ? 0 : 1;
}

static int gzdecomp(void)
{
	gzFile g = gzdopen(0, "rb");
	if (!g) return 1;
	int n;
	while ((n = gzread(g, out, CHUNK)) > 0)
		if (fwrite(out, 1, (size_t)n, stdout) != (size_t)n) return 1;
	if (n < 0) return 1;
	return gzclose(g) == Z_OK ? 0 : 1;
}

int main(int argc, char **argv)
{
	int d = 0, z = 0, level = 6;
	for (int i = 1; i < argc; i++) {
		if (!strcmp(argv[i], "-d")) d = 1;
		else if (!strcmp(argv[i], "-z")) z = 1;
		else if (argv[i][0] == '-' && argv[i][1] >= '1' && argv[i][1] <= '9' && !argv[i][2]) level = argv[i][1] - '0';
		else { fprintf(stderr, "usage: minigz [-d] [-1..-9] [-z] < in > out\n"); return 2; }
	}
	int rc = z ? (d ? zinflate() : zdeflate(level)) : (d ? gzdecomp() : gzcomp(level));
	if (fflush(stdout)) rc = 1;
	return rc;
}
