/*
 * nxz_cksum.c -- CRC-32 (IEEE 802.3, reflected, as gzip) and Adler-32 with
 * their combine operations.  TEST INFRASTRUCTURE ONLY (see nxz_oracle.h).
 *
 * Behaviour follows /root/reference lib/nx_crc.c:215-345 (nx_crc32),
 * :374-434 (crc32_combine), lib/nx_adler32.c:81-148 (nx_adler32), :154-177
 * (combine).  The arithmetic is restated from the definitions (RFC1952 8,
 * RFC1950 9), not from those files: combine is done by multiplying with
 * x^(8*len2) mod P in GF(2)[x] instead of zlib's matrix squaring.
 * Pinned by tests/golden/crc32_kat.json and adler32_kat.json
 * (values from test/test_crc32.c:38-180, test/test_adler32.c:38-179).
 */
#include "nxz_oracle.h"

#define POLY 0xedb88320u   /* reflected 0x04c11db7 */

static uint32_t crc_tab[256];
static int crc_tab_ready;

static void crc_init(void)
{
	uint32_t i, k, c;
	for (i = 0; i < 256; i++) {
		c = i;
		for (k = 0; k < 8; k++)
			c = (c & 1) ? (c >> 1) ^ POLY : c >> 1;
		crc_tab[i] = c;
	}
	crc_tab_ready = 1;
}

uint32_t nxo_crc32(uint32_t crc, const uint8_t *p, size_t n)
{
	if (!crc_tab_ready)
		crc_init();
	if (!p)
		return 0;
	crc = ~crc;
	while (n--)
		crc = crc_tab[(crc ^ *p++) & 0xff] ^ (crc >> 8);
	return ~crc;
}

/* a(x)*b(x) mod P in the reflected representation (bit 31 = x^0) */
static uint32_t gf_mul(uint32_t a, uint32_t b)
{
	uint32_t r = 0;
	int i;
	for (i = 0; i < 32; i++) {
		if (b & 0x80000000u)
			r ^= a;
		a = (a & 1) ? (a >> 1) ^ POLY : a >> 1;   /* a *= x */
		b <<= 1;
	}
	return r;
}

/* x^(8*n) mod P */
static uint32_t gf_xpow8(uint64_t n)
{
	uint32_t r = 0x80000000u;        /* 1 */
	uint32_t sq = 0x00800000u;       /* x^8 */
	while (n) {
		if (n & 1)
			r = gf_mul(r, sq);
		sq = gf_mul(sq, sq);
		n >>= 1;
	}
	return r;
}

uint32_t nxo_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
	return gf_mul(crc1, gf_xpow8(len2)) ^ crc2;
}

#define ADLER_BASE 65521u

uint32_t nxo_adler32(uint32_t adler, const uint8_t *p, size_t n)
{
	uint32_t a = adler & 0xffff, b = (adler >> 16) & 0xffff;
	if (!p)
		return 1;
	while (n) {
		size_t k = n < 5552 ? n : 5552;   /* largest k with no u32 overflow */
		n -= k;
		while (k--) {
			a += *p++;
			b += a;
		}
		a %= ADLER_BASE;
		b %= ADLER_BASE;
	}
	return (b << 16) | a;
}

uint32_t nxo_adler32_combine(uint32_t a1, uint32_t a2, uint64_t len2)
{
	uint64_t rem = len2 % ADLER_BASE;
	uint64_t s1 = a1 & 0xffff, s2;
	uint64_t sum1 = (s1 + (a2 & 0xffff) + ADLER_BASE - 1) % ADLER_BASE;
	s2 = (rem * s1) % ADLER_BASE;
	s2 = (s2 + ((a1 >> 16) & 0xffff) + ((a2 >> 16) & 0xffff) + ADLER_BASE - rem) % ADLER_BASE;
	return (uint32_t)((s2 << 16) | sum1);
}
