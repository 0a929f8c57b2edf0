/*
 * nxz_huff.c -- Huffman tables: fixed codes, canonical codes, DHT header parse,
 * and a restatement of the reference's dhtgen().  TEST INFRASTRUCTURE ONLY.
 *
 * Follows (paths relative to /root/reference):
 *   nxo_dhtgen             lib/nx_dhtgen.c:945-1034
 *     length_limit         lib/nx_dhtgen.c:295-317
 *     sort order           lib/nx_dhtgen.c:323-348  (count asc, zero counts last, symbol asc)
 *     two-queue Huffman    lib/nx_dhtgen.c:418-571  (ties prefer the leaf queue, :484)
 *     depth assignment     lib/nx_dhtgen.c:360-394  (root's children have depth 1)
 *     retry schedule       lib/nx_dhtgen.c:576-595  (limit 2^14, x3/4 per pass, until depth<=15)
 *     header encoder       lib/nx_dhtgen.c:709-915  (fixed code-length code :628-648, RLE 16/17/18)
 *   nxo_dht_parse          lib/nx_dht_decomp.c:255-617, RFC1951 3.2.7
 *   nxo_codes_from_lengths lib/nx_dht_decomp.c:619-653, RFC1951 3.2.2
 * Pinned by tests/golden/dhtgen_*.json (made by oracle/_ref/dhtgen_ref built
 * from the reference file in place) and tests/golden/builtin_dht.json.
 */
#include <string.h>
#include <stdlib.h>
#include "nxz_oracle.h"

static uint16_t bitrev(uint32_t code, int len)
{
	uint32_t r = 0;
	int i;
	for (i = 0; i < len; i++)
		r |= ((code >> i) & 1u) << (len - 1 - i);
	return (uint16_t)r;
}

static int canon(const uint8_t *len, uint16_t *code, int n)
{
	uint32_t bl_count[16] = { 0 }, next[16], c = 0;
	int i, b;
	uint64_t kraft = 0;
	for (i = 0; i < n; i++)
		bl_count[len[i]]++;
	bl_count[0] = 0;
	for (b = 1; b <= 15; b++) {
		c = (c + bl_count[b - 1]) << 1;
		next[b] = c;
		kraft += (uint64_t)bl_count[b] << (15 - b);
	}
	for (i = 0; i < n; i++)
		code[i] = len[i] ? bitrev(next[len[i]]++, len[i]) : 0;
	return kraft > (1u << 15) ? -1 : 0;
}

int nxo_codes_from_lengths(nxo_codes_t *c)
{
	int a = canon(c->ll_len, c->ll_code, 288);
	int b = canon(c->d_len, c->d_code, 32);
	return (a || b) ? -1 : 0;
}

void nxo_codes_fixed(nxo_codes_t *c)
{
	int i;
	memset(c, 0, sizeof(*c));
	for (i = 0; i < 144; i++) c->ll_len[i] = 8;
	for (; i < 256; i++) c->ll_len[i] = 9;
	for (; i < 280; i++) c->ll_len[i] = 7;
	for (; i < 288; i++) c->ll_len[i] = 8;
	for (i = 0; i < 32; i++) c->d_len[i] = 5;
	nxo_codes_from_lengths(c);
}

/* ---- DHT header parse -------------------------------------------------- */
typedef struct { const uint8_t *p; int nbits, pos; } bitr_t;

static int br_get(bitr_t *b, int n)
{
	int v = 0, i;
	if (b->pos + n > b->nbits)
		return -1;
	for (i = 0; i < n; i++, b->pos++)
		v |= ((b->p[b->pos >> 3] >> (b->pos & 7)) & 1) << i;
	return v;
}

int nxo_dht_parse(const uint8_t *dht, int dhtlen, nxo_codes_t *c)
{
	static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
	bitr_t b = { dht, dhtlen, 0 };
	uint8_t cl_len[19] = { 0 }, lens[288 + 32];
	uint16_t cl_code[19];
	int hlit, hdist, hclen, i, n, prev = 0;

	memset(c, 0, sizeof(*c));
	if ((hlit = br_get(&b, 5)) < 0 || (hdist = br_get(&b, 5)) < 0 || (hclen = br_get(&b, 4)) < 0)
		return -1;
	hlit += 257; hdist += 1; hclen += 4;
	if (hlit > 286 || hdist > 30)
		return -2;
	for (i = 0; i < hclen; i++) {
		int v = br_get(&b, 3);
		if (v < 0) return -1;
		cl_len[order[i]] = (uint8_t)v;
	}
	if (canon(cl_len, cl_code, 19))
		return -3;
	n = 0;
	while (n < hlit + hdist) {
		/* decode one code-length symbol, bit by bit (codes are stored bit-reversed) */
		int sym = -1, len, code = 0;
		for (len = 1; len <= 7 && sym < 0; len++) {
			int bit = br_get(&b, 1);
			if (bit < 0) return -1;
			code |= bit << (len - 1);
			for (i = 0; i < 19; i++)
				if (cl_len[i] == len && cl_code[i] == code) { sym = i; break; }
		}
		if (sym < 0) return -4;
		if (sym < 16) {
			lens[n++] = (uint8_t)sym; prev = sym;
		} else {
			int rep, val = 0;
			if (sym == 16) { if (n == 0) return -5; rep = br_get(&b, 2); if (rep < 0) return -1; rep += 3; val = prev; }
			else if (sym == 17) { rep = br_get(&b, 3); if (rep < 0) return -1; rep += 3; }
			else { rep = br_get(&b, 7); if (rep < 0) return -1; rep += 11; }
			if (n + rep > hlit + hdist) return -6;
			while (rep--) lens[n++] = (uint8_t)val;
			if (sym != 16) prev = 0;
		}
	}
	memcpy(c->ll_len, lens, hlit);
	memcpy(c->d_len, lens + hlit, hdist);
	if (nxo_codes_from_lengths(c))
		return -7;
	return b.pos;
}

/* ---- dhtgen ------------------------------------------------------------ */
#define NLEN 286
#define NDIS 30

void nxo_fill_zero_lzcounts(uint32_t *ll, uint32_t *d, uint32_t val)
{
	int i;
	if (ll) for (i = 0; i < NLEN; i++) if (!ll[i]) ll[i] = val;
	if (d)  for (i = 0; i < NDIS; i++) if (!d[i])  d[i] = val;
}

typedef struct { uint32_t sym, cnt; } leaf_t;

static int leaf_cmp(const void *a, const void *b)
{
	const leaf_t *x = a, *y = b;
	/* counts are compared as int with 0 -> INT_MAX (nx_dhtgen.c:327-334) */
	int cx = x->cnt ? (int)x->cnt : 0x7fffffff, cy = y->cnt ? (int)y->cnt : 0x7fffffff;
	if (cx != cy) return cx < cy ? -1 : 1;
	return x->sym < y->sym ? -1 : 1;
}

/* returns max depth; depth[] filled for symbols with non-zero count */
static int huffman_depths(const uint32_t *hist, int nsym, uint32_t *depth)
{
	leaf_t leaf[NLEN];
	struct { uint32_t cnt; int child[2], is_leaf[2]; } node[NLEN];
	int stack[2 * NLEN][2];
	int nz = 0, lh = 0, nh = 0, nt = 0, i, sp = 0, maxd = 0;

	for (i = 0; i < nsym; i++) {
		leaf[i].sym = i; leaf[i].cnt = hist[i];
		if (hist[i]) nz++;
	}
	qsort(leaf, nsym, sizeof(leaf_t), leaf_cmp);
	if (nz == 0)
		return 0;
	if (nz == 1) {
		/* the reference walks an uninitialised node here (nx_dhtgen.c:466,555-568);
		 * its callers always pre-fill counts.  Defined here as: the lone symbol gets 1 bit. */
		depth[leaf[0].sym] = 1;
		return 1;
	}
	while ((nz - lh) + (nt - nh) > 1) {
		int k;
		uint32_t cnt[2]; int ch[2], il[2];
		for (k = 0; k < 2; k++) {
			int have_leaf = lh < nz, have_node = nh < nt;
			if (have_leaf && (!have_node || leaf[lh].cnt <= node[nh].cnt)) {
				cnt[k] = leaf[lh].cnt; ch[k] = (int)leaf[lh].sym; il[k] = 1; lh++;
			} else {
				cnt[k] = node[nh].cnt; ch[k] = nh; il[k] = 0; nh++;
			}
		}
		node[nt].cnt = cnt[0] + cnt[1];
		node[nt].child[0] = ch[0]; node[nt].child[1] = ch[1];
		node[nt].is_leaf[0] = il[0]; node[nt].is_leaf[1] = il[1];
		nt++;
	}
	/* root = last node; children of a node visited at `d` have depth d (root: 1) */
	stack[sp][0] = nt - 1; stack[sp][1] = 1; sp++;
	while (sp) {
		int nd, d, k;
		sp--; nd = stack[sp][0]; d = stack[sp][1];
		if (d > maxd) maxd = d;
		for (k = 0; k < 2; k++) {
			if (node[nd].is_leaf[k]) depth[node[nd].child[k]] = (uint32_t)d;
			else { stack[sp][0] = node[nd].child[k]; stack[sp][1] = d < 31 ? d + 1 : 31; sp++; }
		}
	}
	return maxd;
}

static void huffmanize(uint32_t *hist, int nsym, uint32_t *depth)
{
	int limit = 1 << 14, maxd, i;
	do {
		uint64_t sum = 0, divisor;
		for (i = 0; i < nsym; i++) sum += hist[i];
		divisor = (sum + limit - 1) / limit;
		if (divisor)    /* all-zero histogram: the reference divides by zero here */
			for (i = 0; i < nsym; i++) hist[i] = (uint32_t)((hist[i] + divisor - 1) / divisor);
		limit = (limit * 3) / 4;
		maxd = huffman_depths(hist, nsym, depth);
	} while (maxd > 15);
}

int nxo_dhtgen(uint32_t *lhist, int num_lhist, uint32_t *dhist, int num_dhist,
	       uint8_t *dht, int *dht_num_bytes, int *dht_num_valid_bits)
{
	/* fixed code-length code of the reference (lengths from nx_dhtgen.c:628-648; the codes
	 * there are the RFC1951 canonical codes of these lengths, bit-reversed) */
	static const uint8_t cl_len[19] = { 5, 7, 6, 5, 5, 4, 4, 3, 3, 3, 3, 4, 5, 5, 4, 7, 6, 5, 6 };
	static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
	uint32_t lens[NLEN + NDIS + 2];
	uint16_t cl_code[19];
	int nhlit, nhdist, total, i, state = 0, count = 0;
	uint32_t cur = 0xffffffffu;
	uint64_t acc = 0; int nacc = 0, nbits = 0;
	uint8_t *o = dht;

#define PUT(v, n) do { acc |= (uint64_t)(v) << nacc; nacc += (n); nbits += (n); \
	while (nacc >= 8) { *o++ = (uint8_t)acc; acc >>= 8; nacc -= 8; } } while (0)
#define PUTSYM(s) PUT(cl_code[s], cl_len[s])

	memset(lens, 0, sizeof(lens));
	nhlit = num_lhist < 257 ? 257 : num_lhist;
	huffmanize(lhist, num_lhist, lens);
	if (num_dhist > 1) {
		huffmanize(dhist, num_dhist, lens + nhlit);
	} else {
		lens[nhlit] = 1; num_dhist = 1; dhist[0] = 1;
	}
	nhdist = num_dhist < 1 ? 1 : num_dhist;
	total = nhlit + nhdist;
	canon(cl_len, cl_code, 19);

	PUT(nhlit - 257, 5); PUT(nhdist - 1, 5); PUT(19 - 4, 4);
	for (i = 0; i < 19; i++) PUT(cl_len[order[i]], 3);

	/* RLE state machine, nx_dhtgen.c:758-910.  state: 0 start, 1 one pending, 16 repeating
	 * non-zero, 17/18 repeating zero */
	for (i = 0; i <= total; i++) {
		uint32_t nl = (i == total) ? 0xfffffffeu : lens[i];
		int j;
		switch (state) {
		case 0:
			state = 1;
			break;
		case 1:
			if (cur != nl) { PUTSYM(cur); }
			else if (nl != 0) { PUTSYM(cur); state = 16; count = 1; }
			else { state = 17; count = 2; }
			break;
		case 16:
			if (nl != cur) {
				if (count < 3) { for (j = 0; j < count; j++) PUTSYM(cur); }
				else { PUTSYM(16); PUT(count - 3, 2); }
				state = 1;
			} else if (count == 6) {
				PUTSYM(16); PUT(3, 2); count = 1;
			} else count++;
			break;
		case 17:
			if (nl != 0) {
				if (count < 3) { for (j = 0; j < count; j++) PUTSYM(0); }
				else { PUTSYM(17); PUT(count - 3, 3); }
				state = 1;
			} else {
				state = (count == 10) ? 18 : 17;
				count++;
			}
			break;
		case 18:
			if (nl != 0) {
				PUTSYM(18); PUT(count - 11, 7);
				state = 1;
			} else if (count == 138) {
				PUTSYM(18); PUT(138 - 11, 7);
				state = 17; count = 1;
			} else count++;
			break;
		}
		cur = nl;
	}
	if (nacc) *o++ = (uint8_t)acc;
	*dht_num_bytes = (nbits + 7) / 8;
	*dht_num_valid_bits = nbits % 8;
	return 0;
#undef PUT
#undef PUTSYM
}
