/*
 * nxz_lz77.c -- deterministic LZ77 match finder + greedy parser, CPU restatement.
 * TEST INFRASTRUCTURE ONLY (see nxz_oracle.h).
 *
 * The reference has no LZ77 source (POWER NX silicon; its output is not even
 * reproducible run to run: doc/power_nx_gzip_um.pdf 2.5.9.5, mirrored by
 * lib/nx_dhtgen.c:252-256).  PARITY UNPINNED: this file DEFINES the token
 * choice that the HIP kernel (power-gzip_amd/csrc/nxz_lz77.hip) must
 * reproduce bit-for-bit.  The definition is position-parallel by construction
 * so that a 64-lane-wavefront implementation gives identical results:
 *
 *  1. The job source is cut into sub-blocks of NXO_SUBBLOCK (64 KiB) bytes.
 *     A sub-block sees at most NXO_WINDOW (32 KiB) bytes before it (job
 *     history + earlier source).  Sub-blocks are matched independently.
 *  2. Every position p with >= 4 bytes left hashes its next 4 bytes
 *     (little-endian load * 0x9E3779B1 >> (32-HBITS)).
 *  3. Every hash value owns a bucket of two entries: the NEWEST position
 *     inserted so far with that hash and the one that was the newest before
 *     it (2-way, oldest out).  Window bytes are inserted first, in order.
 *     Block positions are handled in chunks of CHUNK positions: all
 *     positions of a chunk look their bucket up (state before the chunk),
 *     then all are inserted: the largest position of the chunk with that
 *     hash becomes the newest entry (other positions of the same chunk with
 *     the same hash leave no trace).  The second entries follow a PIECE of
 *     512 positions late (pieces are counted from the start of the PTILE
 *     tile; the tile's last piece may be shorter): chunk after chunk of a
 *     piece, a bucket that a chunk inserted into is noted with what was its
 *     newest entry before that chunk; the notes of piece c become the second
 *     entries after the lookups of piece c + 1 (all notes at the end of a
 *     tile).  So lookups of piece c see the second entries as the inserts
 *     up to piece c - 2 left them: the kernel's chain wave must not wait
 *     for a lookup before it inserts, and its second entries live in device
 *     memory, a piece of latency away.
 *     Positions deep inside a byte run (the 12 bytes p-8 .. p+3 all equal)
 *     take no part in the table, neither lookup nor insert.
 *  3b. The second entries are used in the first tile of a sub-block and, in
 *     its later tiles, only if the first tile's parse made NXO_SECOND_MIN_TOKENS
 *     tokens or more (a tile of 16384 positions: fewer than 3072 tokens is
 *     data that compresses 5:1 and better).  Easy data does not need them:
 *     there they cost the kernel a fifth of its match phase -- a second
 *     probe per position, the second entries' upkeep in the chain, and
 *     matches that take turns between two distances -- for 1 to 5 % of an
 *     output that is 4 to 12 % smaller than zlib -1's anyway (JSON, msgpack,
 *     XML, tables); text, binaries and fonts, where the second candidate
 *     is worth 2 to 4 % and the margin to zlib -1 is thin, keep it.
 *  3c. Text that is not easy in the sense of 3b -- fewer than one byte in
 *     NXO_TEXT_HIGH_DIV (16) of the first tile has its top bit set, and that
 *     tile's parse made NXO_SECOND_MIN_TOKENS tokens or more: prose, program
 *     sources, base64 -- does without the second entries in its later tiles
 *     too, and without the lazy step of 5 there.  Its literals are cheap
 *     (5 bits and fewer under the block's own table), so zlib -1, which spends
 *     its 3-byte matches on them, is beaten by 7-8 % with everything on; the
 *     two together cost such a block 4 % of its output and a tenth of the
 *     kernel's time.  Binaries, fonts and text in scripts that live above
 *     0x7f, where the margin to zlib -1 is thin, keep both.  (Per class on
 *     the fallback corpus: text 1.08 -> 1.03-1.05 x zlib -1, sources 1.076 ->
 *     1.036, base64 1.027 -> 1.014; overall 0.996 -> 0.993.)
 *  4. A bucket entry q is a candidate if dist = p-q <= 32768 and >= 4 bytes
 *     agree.  Of two candidates the one with more equal bytes among the
 *     first 8 is taken, the newest on a tie (8 bytes decide: what a lane can
 *     compare in one step); it is extended to at most 258 bytes / end of
 *     sub-block.  Another candidate at distance 1 is tried when
 *     load32(p-1) == load32(p) (byte runs); the longer wins, ties go to the
 *     smaller distance.  (Single entry per bucket: 0.911 x zlib -1 on
 *     repetitive tables, 0.945 x Z_FIXED on the bench blocks; two: 0.99 / 0.98.)
 *  5. Parsing, per tile of PTILE positions, in two passes over segments of
 *     PSEG positions.  Pass 1: every segment is walked greedily from its own
 *     start until the walk leaves the segment (one-step lazy evaluation: a
 *     match of length L < LAZY_MAX at p is replaced by a literal when
 *     position p+1 has a match longer than L; matches are truncated at the
 *     tile's end and dropped if fewer than 3 bytes remain); X[s] is where
 *     it leaves.  Pass 2: the chain of ENTERED segments -- the one that holds
 *     the tile's first position, then the one that holds X of the one before
 *     -- is walked for real from its entry: step by step with matches
 *     truncated at X[s] until the walk stands on a position that pass 1's
 *     walk of this segment visited; from there on it IS pass 1's walk (its
 *     tokens are taken as they are, they end at X[s]).  A walk that has
 *     stepped over the start of pass 1's last match without visiting it
 *     stands inside that match: what is left of it -- same distance, up to
 *     X[s] -- is the last token (literals if fewer than 3 bytes are left).
 *     So every step of pass 2 is taken inside the segment: a lane of the
 *     kernel walks on the 16 lengths it holds in registers, keeps pass 1's
 *     visits in a mask, and most segments cost no second walk at all.
 */
#include <string.h>
#include "nxz_oracle.h"

#ifndef NXO_HBITS
#define NXO_HBITS 13
#endif
#ifndef NXO_CHUNK
#define NXO_CHUNK 64
#endif
#ifndef NXO_PSEG
#define NXO_PSEG 16
#endif
#ifndef NXO_PTILE
#define NXO_PTILE 16384
#endif
#ifndef NXO_LAZY_MAX
#define NXO_LAZY_MAX 32     /* 0 disables lazy evaluation */
#endif
#ifndef NXO_RLE
#define NXO_RLE 1
#endif
#ifndef NXO_PIECE
#define NXO_PIECE 512
#endif
#ifndef NXO_TEXT_HIGH_DIV
#define NXO_TEXT_HIGH_DIV 16         /* step 3c; 0 = no such rule */
#endif
#ifndef NXO_SECOND_MIN_TOKENS
#define NXO_SECOND_MIN_TOKENS 3072   /* step 3b: tokens of a sub-block's first tile below which the later tiles do without second entries */
#endif
#define MINMATCH 4
#define MAXMATCH 258

static inline uint32_t ld32(const uint8_t *p)
{
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

static inline uint32_t hash4(uint32_t v)
{
	return (v * 0x9E3779B1u) >> (32 - NXO_HBITS);
}

static inline uint32_t match_len(const uint8_t *a, const uint8_t *b, uint32_t maxlen)
{
	uint32_t l = 0;
	while (l < maxlen && a[l] == b[l])
		l++;
	return l;
}

/* Greedy walk with one-step lazy evaluation from position p (relative to the
 * sub-block start) until p >= stop.  Matches are truncated at `limit`
 * (dropped if fewer than 3 bytes remain).  Returns the exit position.  With
 * tok != NULL the tokens are appended at tok[*ntok]. */
static uint32_t walk(const uint8_t *w, uint32_t h, const uint16_t *mlen, const uint16_t *mdist,
		     uint32_t p, uint32_t stop, uint32_t limit, uint32_t *tok, size_t *ntok, uint32_t *visited, uint32_t vbase,
		     uint32_t *last_match, uint32_t lazy_max)
{
	while (p < stop) {
		uint32_t len = mlen[p];
		if (visited)
			*visited |= 1u << (p - vbase);      /* a token starts here (stop - vbase <= 32) */
		if (last_match)
			*last_match = 0xffffffffu;          /* start of the walk's last token if that is a match */
		if (len > limit - p)
			len = limit - p;
		if (len >= MINMATCH - 1 && mlen[p] >= MINMATCH) {
			if (len < lazy_max && p + 1 < limit) {
				uint32_t l2 = mlen[p + 1];
				if (l2 > limit - p - 1)
					l2 = limit - p - 1;
				if (l2 > len) {
					if (tok) tok[(*ntok)++] = w[h + p];
					p++;
					continue;
				}
			}
			if (tok) tok[(*ntok)++] = NXO_TOK_MATCH | ((uint32_t)mdist[p] << 8) | (len - 3);
			if (last_match) *last_match = p;
			p += len;
		} else {
			if (tok) tok[(*ntok)++] = w[h + p];
			p++;
		}
	}
	return p;
}

/* debug taps */
uint16_t *nxo_dbg_mlen, *nxo_dbg_mdist; uint32_t *nxo_dbg_x;

/* one sub-block: w[0..h) window, w[h..h+n) block.  n <= NXO_SUBBLOCK, h <= NXO_WINDOW */
/* Positions deep inside a run of one byte value -- the 12 bytes w[r-8 .. r+3] are all the same --
 * are left out of the hash table (neither looked up nor inserted): there the distance-1 candidate is
 * what counts, and a table in which thousands of positions share one slot is of no use to anyone. */
static inline int deep_in_run(const uint8_t *w, uint32_t r)
{
	uint32_t k;
	if (r < 8)
		return 0;
	for (k = r - 8; k < r + 3; k++)
		if (w[k] != w[k + 1])
			return 0;
	return 1;
}

static size_t lz77_subblock(const uint8_t *w, uint32_t h, uint32_t n, uint32_t *tok)
{
	static __thread uint32_t head[1u << NXO_HBITS];    /* newest entry (position + 1; 0 == empty) */
	static __thread uint32_t head2[1u << NXO_HBITS];   /* the one before it */
	static __thread uint16_t pend_h[2][NXO_PIECE], pend_v[2][NXO_PIECE];   /* notes of the current piece [0] and of the one before [1] */
	uint32_t npend[2] = { 0, 0 };
	static __thread uint16_t mlen[NXO_SUBBLOCK];
	static __thread uint16_t mdist[NXO_SUBBLOCK];
	const uint32_t end = h + n;
	uint32_t c, r, ntok = 0, t0, t1;
	int use_second = 1, text = 0;
	uint32_t lazy_max = NXO_LAZY_MAX;

#if NXO_TEXT_HIGH_DIV
	{
		/* 3c. text: fewer than one byte in NXO_TEXT_HIGH_DIV of the first tile has its top bit set */
		uint32_t t0n = n < NXO_PTILE ? n : NXO_PTILE, high = 0;
		for (r = 0; r < t0n; r++)
			high += w[h + r] >> 7;
		text = high * NXO_TEXT_HIGH_DIV < t0n;
	}
#endif

	memset(head, 0, sizeof(head));
	memset(head2, 0, sizeof(head2));
	/* 3. window seeding (entries are position+1; 0 == empty) */
	for (r = 0; r < h && r + 4 <= end; r++) {
		uint32_t hv = hash4(ld32(w + r));
		if (deep_in_run(w, r))
			continue;
		if (head[hv] < r + 1) {
			head2[hv] = head[hv];
			head[hv] = r + 1;
		}
	}
	/* tile after tile: match finding, then the parse (the parse of a tile has no say in the match finding of the
	 * next one -- except through the count of step 3b) */
	for (t0 = 0; t0 < n; t0 = t1) {
	t1 = t0 + NXO_PTILE < n ? t0 + NXO_PTILE : n;
	/* 2-4. match finding per chunk */
	for (c = t0; c < t1; c += NXO_CHUNK) {
		uint32_t cend = c + NXO_CHUNK < n ? c + NXO_CHUNK : n;
		uint32_t j;
		for (j = c; j < cend; j++) {
			uint32_t len = 0, dist = 0;
			r = h + j;
			if (r + 4 <= end) {
				uint32_t v = ld32(w + r);
				int deep = deep_in_run(w, r);
				uint32_t cand = deep ? 0 : head[hash4(v)], cand2 = deep || !use_second ? 0 : head2[hash4(v)];
				uint32_t maxlen = end - r < MAXMATCH ? end - r : MAXMATCH;
				uint32_t cap8 = maxlen < 8 ? maxlen : 8, a1 = 0, a2 = 0;
				if (cand != 0 && r - (cand - 1) <= NXO_WINDOW && ld32(w + cand - 1) == v)
					a1 = match_len(w + cand - 1, w + r, cap8);
				if (cand2 != 0 && r - (cand2 - 1) <= NXO_WINDOW && ld32(w + cand2 - 1) == v)
					a2 = match_len(w + cand2 - 1, w + r, cap8);
				if (a2 > a1)
					cand = cand2;
				if (a1 || a2) {
					len = match_len(w + cand - 1, w + r, maxlen);
					dist = r - (cand - 1);
				}
#if NXO_RLE
				if (r >= 1 && ld32(w + r - 1) == v) {
					uint32_t l1 = match_len(w + r - 1, w + r, maxlen);
					if (l1 >= len) {
						len = l1;
						dist = 1;
					}
				}
#endif
			}
			mlen[j] = (uint16_t)len;
			mdist[j] = (uint16_t)(dist - 1);
		}
		for (j = c; j < cend; j++) {
			r = h + j;
			if (r + 4 <= end && !deep_in_run(w, r)) {
				uint32_t hv = hash4(ld32(w + r));
				if (head[hv] < r + 1) {
					if (head[hv] < h + c + 1) {   /* the newest entry from before this chunk: second entry, later */
						pend_h[0][npend[0]] = (uint16_t)hv;
						pend_v[0][npend[0]++] = (uint16_t)head[hv];
					}
					head[hv] = r + 1;
				}
			}
		}
		/* end of a piece: the notes of the piece before become second entries; end of the tile
		 * (or of the data): those of this piece too */
		if ((cend % NXO_PTILE) % NXO_PIECE == 0 || cend == n) {
			int tile_end = cend % NXO_PTILE == 0 || cend == n;
			for (j = 0; j < npend[1]; j++)
				head2[pend_h[1][j]] = pend_v[1][j];
			if (tile_end) {
				for (j = 0; j < npend[0]; j++)
					head2[pend_h[0][j]] = pend_v[0][j];
				npend[0] = 0;
			}
			memcpy(pend_h[1], pend_h[0], npend[0] * sizeof(uint16_t));
			memcpy(pend_v[1], pend_v[0], npend[0] * sizeof(uint16_t));
			npend[1] = npend[0];
			npend[0] = 0;
		}
	}
	/* 5. two-pass segment parse of the tile */
	for (c = t0; c < t1; c += NXO_PTILE) {
		uint32_t tend = c + NXO_PTILE < n ? c + NXO_PTILE : n;
		uint32_t nseg = (tend - c + NXO_PSEG - 1) / NXO_PSEG, s;
		static __thread uint32_t X[NXO_PTILE / NXO_PSEG], V[NXO_PTILE / NXO_PSEG], A[NXO_PTILE / NXO_PSEG];
		uint32_t entry = c;
		/* pass 1: speculative walk of every segment from its own start; X = exit */
		for (s = 0; s < nseg; s++) {
			uint32_t sb = c + s * NXO_PSEG;
			uint32_t se = sb + NXO_PSEG < tend ? sb + NXO_PSEG : tend;
			V[s] = 0;
			X[s] = walk(w, h, mlen, mdist, sb, se, tend, NULL, NULL, &V[s], sb, &A[s], lazy_max);
		}
		if (nxo_dbg_x) for (s = 0; s < nseg; s++) nxo_dbg_x[c / NXO_PSEG + s] = X[s];
		/* chain of entered segments: the segment containing `entry` is walked for real from `entry`:
		 * step by step (matches truncated at the segment's speculative exit) until it stands on a
		 * position the speculative walk visited, then along that walk to its exit, which is the entry of
		 * the next entered segment */
		while (entry < tend) {
			size_t k = ntok;
			uint32_t sb, se, p = entry;
			s = (entry - c) / NXO_PSEG;
			sb = c + s * NXO_PSEG;
			se = sb + NXO_PSEG < tend ? sb + NXO_PSEG : tend;
			while (p < X[s]) {
				if (p < se && ((V[s] >> (p - sb)) & 1)) {
					p = walk(w, h, mlen, mdist, p, se, tend, tok, &k, NULL, 0, NULL, lazy_max);
					break;
				}
				if (A[s] != 0xffffffffu && p > A[s]) {
					/* inside the speculative walk's last match: the rest of it */
					if (X[s] - p >= 3)
						tok[k++] = NXO_TOK_MATCH | ((uint32_t)mdist[A[s]] << 8) | (X[s] - p - 3);
					else
						while (p < X[s])
							tok[k++] = w[h + p++];
					p = X[s];
					break;
				}
				p = walk(w, h, mlen, mdist, p, p + 1, X[s], tok, &k, NULL, 0, NULL, lazy_max);
			}
			ntok = (uint32_t)k;
			entry = X[s];
		}
	}
	/* 3b. what the first tile came to decides about the second entries for the rest of the sub-block */
	if (t0 == 0) {
		use_second = ntok >= NXO_SECOND_MIN_TOKENS && !text;
		if (text && ntok >= NXO_SECOND_MIN_TOKENS)
			lazy_max = 0;                      /* 3c */
	}
	}
	if (nxo_dbg_mlen) { memcpy(nxo_dbg_mlen, mlen, n * 2); memcpy(nxo_dbg_mdist, mdist, n * 2); }
	return ntok;
}

size_t nxo_lz77(const uint8_t *buf, size_t hist, size_t n, uint32_t *tok)
{
	size_t off = 0, ntok = 0;
	while (off < n) {
		uint32_t bn = n - off < NXO_SUBBLOCK ? (uint32_t)(n - off) : NXO_SUBBLOCK;
		size_t before = hist + off;
		uint32_t h = before < NXO_WINDOW ? (uint32_t)before : NXO_WINDOW;
		ntok += lz77_subblock(buf + before - h, h, bn, tok + ntok);
		off += bn;
	}
	return ntok;
}

/* ---- symbol mapping (RFC1951 3.2.5) ------------------------------------ */
static const uint16_t len_base[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
	35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint8_t len_extra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
	3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t dist_base[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
	257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
static const uint8_t dist_extra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
	7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

static inline uint32_t len_sym(uint32_t len)   /* returns index 0..28 */
{
	uint32_t s = 28;
	while (len_base[s] > len)
		s--;
	return s;
}

static inline uint32_t dist_sym(uint32_t dist)
{
	uint32_t s = 29;
	while (dist_base[s] > dist)
		s--;
	return s;
}

void nxo_count(const uint32_t *tok, size_t ntok, uint32_t ll[286], uint32_t d[30])
{
	size_t i;
	memset(ll, 0, 286 * sizeof(uint32_t));
	memset(d, 0, 30 * sizeof(uint32_t));
	for (i = 0; i < ntok; i++) {
		uint32_t t = tok[i];
		if (t & NXO_TOK_MATCH) {
			ll[257 + len_sym((t & 0xff) + 3)]++;
			d[dist_sym(((t >> 8) & 0x7fff) + 1)]++;
		} else {
			ll[t & 0xff]++;
		}
	}
	ll[256] = 1;
}

/* ---- bit writer -------------------------------------------------------- */
typedef struct {
	uint8_t *out;
	size_t cap;
	uint64_t bits;    /* total bits written */
	uint64_t acc;
	unsigned nacc;
	int ovf;
} bitw_t;

static void bw_put(bitw_t *b, uint32_t v, unsigned n)
{
	b->acc |= (uint64_t)v << b->nacc;
	b->nacc += n;
	b->bits += n;
	while (b->nacc >= 8) {
		size_t idx = (size_t)((b->bits - b->nacc) >> 3);
		if (idx < b->cap)
			b->out[idx] = (uint8_t)b->acc;
		else
			b->ovf = 1;
		b->acc >>= 8;
		b->nacc -= 8;
	}
}

static void bw_flush(bitw_t *b)
{
	if (b->nacc) {
		size_t idx = (size_t)(b->bits >> 3);
		if (idx < b->cap)
			b->out[idx] = (uint8_t)b->acc;
		else
			b->ovf = 1;
	}
}

static int put_tokens(bitw_t *b, const uint32_t *tok, size_t ntok, const nxo_codes_t *c)
{
	size_t i;
	for (i = 0; i < ntok; i++) {
		uint32_t t = tok[i];
		if (t & NXO_TOK_MATCH) {
			uint32_t len = (t & 0xff) + 3, dist = ((t >> 8) & 0x7fff) + 1;
			uint32_t ls = len_sym(len), ds = dist_sym(dist);
			if (!c->ll_len[257 + ls] || !c->d_len[ds])
				return -1;
			bw_put(b, c->ll_code[257 + ls], c->ll_len[257 + ls]);
			if (len_extra[ls])
				bw_put(b, len - len_base[ls], len_extra[ls]);
			bw_put(b, c->d_code[ds], c->d_len[ds]);
			if (dist_extra[ds])
				bw_put(b, dist - dist_base[ds], dist_extra[ds]);
		} else {
			if (!c->ll_len[t & 0xff])
				return -1;
			bw_put(b, c->ll_code[t & 0xff], c->ll_len[t & 0xff]);
		}
	}
	if (!c->ll_len[256])
		return -1;
	bw_put(b, c->ll_code[256], c->ll_len[256]);
	return 0;
}

uint64_t nxo_encode_fixed(const uint32_t *tok, size_t ntok, uint8_t *out, size_t out_cap)
{
	nxo_codes_t c;
	bitw_t b = { out, out_cap, 0, 0, 0, 0 };
	nxo_codes_fixed(&c);
	bw_put(&b, 1, 1);  /* BFINAL as the engine emits it (UM 5.1.1; host rewrites, nx_deflate.c:158) */
	bw_put(&b, 1, 2);  /* BTYPE 01 */
	put_tokens(&b, tok, ntok, &c);
	bw_flush(&b);
	return b.ovf ? (uint64_t)-2 : b.bits;
}

uint64_t nxo_encode_dynamic(const uint32_t *tok, size_t ntok, const uint8_t *dht, int dhtlen,
			    uint8_t *out, size_t out_cap)
{
	nxo_codes_t c;
	bitw_t b = { out, out_cap, 0, 0, 0, 0 };
	int i;
	if (nxo_dht_parse(dht, dhtlen, &c) != dhtlen)
		return (uint64_t)-1;
	bw_put(&b, 1, 1);
	bw_put(&b, 2, 2);  /* BTYPE 10 */
	for (i = 0; i + 8 <= dhtlen; i += 8)
		bw_put(&b, dht[i >> 3], 8);
	if (dhtlen & 7)
		bw_put(&b, dht[dhtlen >> 3] & ((1u << (dhtlen & 7)) - 1), dhtlen & 7);
	if (put_tokens(&b, tok, ntok, &c))
		return (uint64_t)-1;
	bw_flush(&b);
	return b.ovf ? (uint64_t)-2 : b.bits;
}
