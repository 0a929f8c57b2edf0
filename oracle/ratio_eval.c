/* ratio_eval.c -- compare the oracle's compressed size with zlib -1 on 64 KiB chunks.
 * TEST INFRASTRUCTURE ONLY.  usage: ratio_eval file... */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "nxz_oracle.h"

static size_t zsize(const uint8_t *p, size_t n, int level, int strategy)
{
	static uint8_t out[200000];
	z_stream s; memset(&s, 0, sizeof(s));
	deflateInit2(&s, level, Z_DEFLATED, -15, 8, strategy);
	s.next_in = (Bytef *)p; s.avail_in = n; s.next_out = out; s.avail_out = sizeof(out);
	deflate(&s, Z_FINISH);
	size_t r = s.total_out; deflateEnd(&s); return r;
}

static int check(const uint8_t *comp, size_t clen, const uint8_t *orig, size_t n)
{
	static uint8_t out[70000];
	z_stream s; memset(&s, 0, sizeof(s));
	inflateInit2(&s, -15);
	s.next_in = (Bytef *)comp; s.avail_in = clen; s.next_out = out; s.avail_out = sizeof(out);
	int rc = inflate(&s, Z_FINISH);
	int ok = rc == Z_STREAM_END && s.total_out == n && !memcmp(out, orig, n);
	inflateEnd(&s);
	return ok;
}

int main(int argc, char **argv)
{
	size_t T[6] = {0};
	for (int a = 1; a < argc; a++) {
		FILE *f = fopen(argv[a], "rb"); if (!f) { perror(argv[a]); continue; }
		static uint8_t buf[65536], out[200000]; static uint32_t tok[65536];
		size_t n, tot = 0, z1 = 0, z1f = 0, z6 = 0, of = 0, od = 0; int bad = 0;
		while ((n = fread(buf, 1, sizeof(buf), f)) > 0) {
			tot += n;
			z1 += zsize(buf, n, 1, Z_DEFAULT_STRATEGY);
			z1f += zsize(buf, n, 1, Z_FIXED);
			z6 += zsize(buf, n, 6, Z_DEFAULT_STRATEGY);
			size_t nt = nxo_lz77(buf, 0, n, tok);
			uint64_t bits = nxo_encode_fixed(tok, nt, out, sizeof(out));
			of += (bits + 7) / 8;
			if (!check(out, (bits + 7) / 8, buf, n)) bad++;
			uint32_t ll[286], d[30]; uint8_t dht[320]; int nb, vb;
			nxo_count(tok, nt, ll, d);
			nxo_dhtgen(ll, 286, d, 30, dht, &nb, &vb);
			int dhtlen = nb * 8 - (vb ? 8 - vb : 0);
			bits = nxo_encode_dynamic(tok, nt, dht, dhtlen, out, sizeof(out));
			if (bits == (uint64_t)-1) { bad++; continue; }
			od += (bits + 7) / 8;
			if (!check(out, (bits + 7) / 8, buf, n)) bad++;
		}
		fclose(f);
		printf("%-16s %9zu | zlib1 %.3f fix %.3f z6 %.3f | ours dht %.3f (%.3fx) fht %.3f (%.3fx) %s\n",
		       strrchr(argv[a], '/') ? strrchr(argv[a], '/') + 1 : argv[a], tot,
		       (double)tot / z1, (double)tot / z1f, (double)tot / z6,
		       (double)tot / od, (double)z1 / od, (double)tot / of, (double)z1f / of, bad ? "BAD" : "ok");
		T[0] += tot; T[1] += z1; T[2] += z1f; T[3] += od; T[4] += of; T[5] += bad;
	}
	printf("TOTAL %zu | zlib1 %.3f fix %.3f | ours dht %.3f (%.3fx) fht %.3f (%.3fx) bad=%zu\n", T[0],
	       (double)T[0] / T[1], (double)T[0] / T[2], (double)T[0] / T[3], (double)T[1] / T[3],
	       (double)T[0] / T[4], (double)T[2] / T[4], T[5]);
	return 0;
}
