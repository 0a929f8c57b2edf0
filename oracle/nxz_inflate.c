/*
 * nxz_inflate.c -- CPU model of the engine's DECOMPRESS / DECOMPRESS_RESUME
 * function codes (0x10 / 0x14).  TEST INFRASTRUCTURE ONLY (see nxz_oracle.h).
 *
 * No reference source exists for the decoder (NX silicon).  The suspend /
 * resume contract restated here is the one the reference's consumer code
 * relies on (paths relative to /root/reference):
 *   - resume inputs  in_sfbt/in_subc/in_rembytecnt/in_dht: inc_nx/nxu.h:296-393
 *   - outputs out_sfbt/out_subc/out_rembytecnt/out_dht:    inc_nx/nxu.h:403-541
 *   - how each SFBT value is consumed:                    lib/nx_inflate.c:1464-1609
 *   - CC=3+partial is the normal "source ran out / trailer follows" result,
 *     CC=0 only when the source ends exactly with the final EOB,
 *     CC=13 when the target is too small:                 lib/nx_inflate.c:1372-1445
 * The decode itself is RFC1951; it is pinned by round trips against system
 * zlib and by the reference's golden zlib stream (test/test_buf_error.c:107-183).
 */
#include <string.h>
#include "nxz_oracle.h"

typedef struct {
	const uint8_t *p;
	uint64_t nbits;   /* total bits in source */
	uint64_t pos;     /* next bit */
} br_t;

static inline int64_t br_avail(const br_t *b) { return (int64_t)(b->nbits - b->pos); }

static inline uint32_t br_peekbit(const br_t *b, uint64_t pos)
{
	return (b->p[pos >> 3] >> (pos & 7)) & 1u;
}

/* returns -1 if not enough bits */
static int32_t br_get(br_t *b, int n)
{
	uint32_t v = 0;
	int i;
	if (br_avail(b) < n)
		return -1;
	for (i = 0; i < n; i++, b->pos++)
		v |= br_peekbit(b, b->pos) << i;
	return (int32_t)v;
}

typedef struct {
	uint16_t count[16];
	uint16_t symbol[288];
} htab_t;

static void htab_build(htab_t *h, const uint8_t *len, int n)
{
	uint16_t offs[16];
	int i;
	memset(h->count, 0, sizeof(h->count));
	for (i = 0; i < n; i++)
		h->count[len[i]]++;
	h->count[0] = 0;
	offs[1] = 0;
	for (i = 1; i < 15; i++)
		offs[i + 1] = offs[i] + h->count[i];
	for (i = 0; i < n; i++)
		if (len[i])
			h->symbol[offs[len[i]]++] = (uint16_t)i;
}

/* canonical decode; returns symbol, -1 out of bits, -2 invalid code */
static int htab_decode(br_t *b, const htab_t *h)
{
	int code = 0, first = 0, index = 0, len;
	for (len = 1; len <= 15; len++) {
		int count;
		if (br_avail(b) < 1)
			return -1;
		code |= (int)br_peekbit(b, b->pos++);
		count = h->count[len];
		if (code - count < first)
			return h->symbol[index + (code - first)];
		index += count;
		first += count;
		first <<= 1;
		code <<= 1;
	}
	return -2;
}

static const uint16_t len_base[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
	35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint8_t len_extra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
	3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t dist_base[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
	257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
static const uint8_t dist_extra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
	7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

enum { ST_HEADER, ST_STORED, ST_CODED };

static void set_suspend(nxo_inflate_state_t *st, const br_t *b, uint64_t from, uint32_t sfbt)
{
	st->out_sfbt = sfbt;
	st->out_subc = (uint32_t)(b->nbits - from);
}

int nxo_inflate(const uint8_t *src, size_t srclen, uint8_t *dst, size_t dstcap,
		size_t hist, nxo_inflate_state_t *st)
{
	br_t b;
	htab_t hl, hd;
	nxo_codes_t codes;
	size_t out = 0;
	int state = ST_HEADER, bfinal = 0, btype = 0;
	uint32_t rem = 0;

	st->out_sfbt = st->out_subc = st->out_rembytecnt = 0;
	st->out_dhtlen = 0;
	st->final_eob = 0;
	st->err = 0;
	st->spbc = srclen;
	st->tpbc = 0;

	b.p = src;
	b.nbits = (uint64_t)srclen * 8;
	b.pos = 0;
	if (srclen && (st->subc & 7))
		b.pos = 8 - (st->subc & 7);

	/* resume state */
	if (st->sfbt & 0x8) {
		uint32_t kind = (st->sfbt >> 1) & 0x7;
		bfinal = st->sfbt & 1;
		if (kind == 4) {                 /* 100x stored */
			state = ST_STORED; btype = 0; rem = st->rembytecnt;
		} else if (kind == 5) {          /* 101x fixed */
			state = ST_CODED; btype = 1;
			nxo_codes_fixed(&codes);
			htab_build(&hl, codes.ll_len, 288);
			htab_build(&hd, codes.d_len, 30);
		} else if (kind == 6) {          /* 110x dynamic */
			state = ST_CODED; btype = 2;
			if (nxo_dht_parse(st->dht, st->dhtlen, &codes) != st->dhtlen) {
				st->err = 68;
				return st->err;
			}
			htab_build(&hl, codes.ll_len, 286);
			htab_build(&hd, codes.d_len, 30);
			memcpy(st->out_dht, st->dht, (size_t)(st->dhtlen + 7) / 8);
			st->out_dhtlen = st->dhtlen;
		} else {                          /* 111x header */
			state = ST_HEADER;
		}
	}

	for (;;) {
		if (state == ST_HEADER) {
			uint64_t hdr = b.pos;
			int32_t v = br_get(&b, 3);
			if (v < 0) { set_suspend(st, &b, hdr, 0xe); goto suspended; }
			bfinal = v & 1;
			btype = v >> 1;
			if (btype == 0) {
				uint32_t len, nlen;
				b.pos = (b.pos + 7) & ~7ull;
				if (br_avail(&b) < 32) { set_suspend(st, &b, hdr, 0xe | bfinal); goto suspended; }
				len = (uint32_t)br_get(&b, 16);
				nlen = (uint32_t)br_get(&b, 16);
				if ((len ^ nlen) != 0xffff) { st->err = 68; goto done; }
				rem = len;
				state = ST_STORED;
			} else if (btype == 1) {
				nxo_codes_fixed(&codes);
				htab_build(&hl, codes.ll_len, 288);
				htab_build(&hd, codes.d_len, 30);
				state = ST_CODED;
			} else if (btype == 2) {
				/* copy the table bits out byte-aligned, then parse */
				uint8_t tmp[320];
				int64_t avail = br_avail(&b);
				int nb = avail > 320 * 8 ? 320 * 8 : (int)avail, i, used;
				memset(tmp, 0, sizeof(tmp));
				for (i = 0; i < nb; i++)
					tmp[i >> 3] |= (uint8_t)(br_peekbit(&b, b.pos + i) << (i & 7));
				used = nxo_dht_parse(tmp, nb, &codes);
				if (used == -1) { set_suspend(st, &b, hdr, 0xe | bfinal); goto suspended; }
				if (used < 0 || used > 288 * 8) { st->err = 68; goto done; }
				if (!codes.ll_len[256]) { st->err = 68; goto done; }
				b.pos += used;
				memset(st->out_dht, 0, sizeof(st->out_dht));
				memcpy(st->out_dht, tmp, (size_t)(used + 7) / 8);
				if (used & 7)
					st->out_dht[used >> 3] &= (uint8_t)((1u << (used & 7)) - 1);
				st->out_dhtlen = used;
				htab_build(&hl, codes.ll_len, 286);
				htab_build(&hd, codes.d_len, 30);
				state = ST_CODED;
			} else {
				st->err = 68;
				goto done;
			}
		} else if (state == ST_STORED) {
			/* byte aligned here */
			size_t srcleft = (size_t)(br_avail(&b) >> 3);
			size_t n = rem < srcleft ? rem : srcleft;
			if (n > dstcap - out) { st->err = 13; goto done; }
			memcpy(dst + out, src + (b.pos >> 3), n);
			out += n; b.pos += (uint64_t)n * 8; rem -= (uint32_t)n;
			if (rem) {
				st->out_sfbt = 0x8 | bfinal;
				st->out_subc = 0;
				st->out_rembytecnt = rem;
				goto suspended;
			}
			if (bfinal) goto final;
			state = ST_HEADER;
		} else {
			uint64_t sym_start = b.pos;
			uint32_t sfbt = (btype == 1 ? 0xa : 0xc) | bfinal;
			int sym = htab_decode(&b, &hl);
			if (sym == -1) { set_suspend(st, &b, sym_start, sfbt); goto suspended; }
			if (sym < 0) { st->err = 66; goto done; }
			if (sym < 256) {
				if (out >= dstcap) { st->err = 13; goto done; }
				dst[out++] = (uint8_t)sym;
			} else if (sym == 256) {
				if (bfinal) goto final;
				state = ST_HEADER;
			} else {
				int32_t e; uint32_t len, dist; int ds; size_t i;
				sym -= 257;
				if (sym >= 29) { st->err = 66; goto done; }
				e = br_get(&b, len_extra[sym]);
				if (e < 0) { set_suspend(st, &b, sym_start, sfbt); goto suspended; }
				len = len_base[sym] + (uint32_t)e;
				ds = htab_decode(&b, &hd);
				if (ds == -1) { set_suspend(st, &b, sym_start, sfbt); goto suspended; }
				if (ds < 0 || ds >= 30) { st->err = 67; goto done; }
				e = br_get(&b, dist_extra[ds]);
				if (e < 0) { set_suspend(st, &b, sym_start, sfbt); goto suspended; }
				dist = dist_base[ds] + (uint32_t)e;
				if (dist > out + hist) { st->err = 67; goto done; }
				if (len > dstcap - out) { st->err = 13; goto done; }
				for (i = 0; i < len; i++, out++)
					dst[out] = dst[(ptrdiff_t)out - (ptrdiff_t)dist];
			}
		}
	}

final:
	st->final_eob = 1;
	st->out_sfbt = 0;
	st->out_subc = (uint32_t)br_avail(&b);
suspended:
done:
	st->tpbc = out;
	return st->err;
}
