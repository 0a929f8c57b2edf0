/*
 * nxz_engine_model.c -- CPU model of one engine job (what nxu_run_job makes
 * the accelerator do).  TEST INFRASTRUCTURE ONLY (see nxz_oracle.h): it is
 * the checker for the HIP engine and the engine under the reference's own
 * host code in oracle/_ref/libnxz_ref.so; the product never calls it.
 *
 * Contract restated from the consumers (paths relative to /root/reference):
 *   submit/complete        lib/nx_zlib.c:469-501, lib/gzip_vas.c:262-417
 *   compress results       lib/nx_deflate.c:919-927 (spbc location), :969-1078
 *                          (spbc incl. history, tpbc, tebc 0==8), :1274-1282 (CC 64)
 *   checksums in CPB       lib/nx_deflate.c:1562-1590, :428-468 (crc little-endian in
 *                          memory, adler big-endian)
 *   lzcounts               lib/nx_dht.c:169-237 (316 x u32 big-endian, EOB count 1)
 *   wrap                   lib/nx_zlib.c:1398-1443, lib/nx_deflate.c:1749-1800
 *   decompress             lib/nx_inflate.c:927-1053 (inputs), :1308-1609 (outputs)
 */
#include <stdlib.h>
#include <string.h>
#include "nxz_oracle.h"
#include "../include/nxz_wire.h"

/* gather up to `want` bytes described by a DDE into a fresh buffer */
static uint8_t *gather(const nxz_dde_t *d, uint32_t *total)
{
	uint32_t n = nxz_dde_bytes(d), cnt = nxz_dde_count(d), i, off = 0;
	uint8_t *buf = malloc((size_t)n + 64);
	if (!buf) return NULL;
	if (cnt == 0) {
		if (n) memcpy(buf, nxz_dde_addr(d), n);
	} else {
		const nxz_dde_t *l = nxz_dde_addr(d);
		for (i = 0; i < cnt && off < n; i++) {
			uint32_t k = nxz_dde_bytes(&l[i]);
			if (k > n - off) k = n - off;
			memcpy(buf + off, nxz_dde_addr(&l[i]), k);
			off += k;
		}
		n = off;
	}
	*total = n;
	return buf;
}

static uint32_t target_cap(const nxz_dde_t *d)
{
	uint32_t n = nxz_dde_bytes(d), cnt = nxz_dde_count(d), i, sum = 0;
	if (cnt == 0) return n;
	for (i = 0; i < cnt; i++) sum += nxz_dde_bytes(&((const nxz_dde_t *)nxz_dde_addr(d))[i]);
	return sum < n ? sum : n;
}

static void scatter(const nxz_dde_t *d, const uint8_t *src, uint32_t n)
{
	uint32_t cnt = nxz_dde_count(d), i, off = 0;
	if (cnt == 0) {
		if (n) memcpy(nxz_dde_addr(d), src, n);
		return;
	}
	for (i = 0; i < cnt && off < n; i++) {
		const nxz_dde_t *e = &((const nxz_dde_t *)nxz_dde_addr(d))[i];
		uint32_t k = nxz_dde_bytes(e);
		if (k > n - off) k = n - off;
		memcpy(nxz_dde_addr(e), src + off, k);
		off += k;
	}
}

static void put_checksums(nxz_crb_cpb_t *j, uint32_t crc, uint32_t adler)
{
	nxz_wr32(&j->cpb.out_adler_be, adler);
	j->cpb.out_crc_le = htole32(crc);
}

int nxo_run_job(nxz_crb_cpb_t *j)
{
	uint32_t fc = nxz_fc(j), srclen = 0, cap = target_cap(&j->crb.target);
	uint8_t *src = gather(&j->crb.source, &srclen);
	uint32_t cc = NXZ_CC_OK, ce = 0, tpbc = 0;

	if (!src) return -1;

	if (fc == NXZ_FC_WRAP) {
		if (srclen > cap) { cc = NXZ_CC_TARGET_SPACE; ce = NXZ_CE_TERMINATE; goto out; }
		scatter(&j->crb.target, src, srclen);
		put_checksums(j, nxo_crc32(0, src, srclen), nxo_adler32(1, src, srclen));
		nxz_wr32(&j->cpb.u.out_spbc_be, srclen);
		tpbc = srclen;
	} else if (nxz_fc_is_compress(fc)) {
		uint32_t hist = nxz_fc_is_resume(fc) ? nxz_in_histlen(&j->cpb) * 16 : 0;
		uint32_t n, *tok, skip, partial = 0;
		uint8_t *out, *srcbase = src;
		uint64_t bits;
		size_t ntok, ocap;
		if (hist > srclen) hist = srclen;
		n = srclen - hist;
		/* only the last 32 KiB of history can be referenced (inc_nx/nxu.h:303-317); one job
		 * covers one 64 KiB sub-block incl. its window, the rest is left to the caller through
		 * the engine's byte-count-limit completion (CC 3 + partial, lib/nx_deflate.c:1341-1359) */
		skip = hist > NXO_WINDOW ? hist - NXO_WINDOW : 0;
		src += skip; hist -= skip;
		if (hist + n > NXO_SUBBLOCK) { n = NXO_SUBBLOCK - hist; partial = 1; }
		srclen = skip + hist + n;
		tok = malloc(((size_t)n + 1) * sizeof(uint32_t));
		ocap = (size_t)n * 2 + 1024;
		out = calloc(1, ocap);
		ntok = nxo_lz77(src, hist, n, tok);
		if (nxz_fc_is_dht(fc))
			bits = nxo_encode_dynamic(tok, ntok, j->cpb.in_dht, (int)nxz_in_dhtlen(&j->cpb), out, ocap);
		else
			bits = nxo_encode_fixed(tok, ntok, out, ocap);
		if (bits == (uint64_t)-1) {
			cc = NXZ_CC_MISSING_CODE; ce = NXZ_CE_TERMINATE;
		} else {
			tpbc = (uint32_t)((bits + 7) / 8);
			if (tpbc > cap) {
				cc = NXZ_CC_TARGET_SPACE; ce = NXZ_CE_TERMINATE; tpbc = 0;
			} else {
				scatter(&j->crb.target, out, tpbc);
				nxz_putf(&j->cpb.out_w2_be, 16, 3, (uint32_t)(bits & 7));
				put_checksums(j, nxo_crc32(nxz_in_crc(&j->cpb), src + hist, n),
					      nxo_adler32(nxz_in_adler(&j->cpb), src + hist, n));
				if (nxz_fc_has_count(fc)) {
					uint32_t ll[286], d[30], i;
					nxo_count(tok, ntok, ll, d);
					for (i = 0; i < 286; i++) nxz_wr32(&j->cpb.u.out_lzcount_be[i], ll[i]);
					for (i = 0; i < 30; i++) nxz_wr32(&j->cpb.u.out_lzcount_be[286 + i], d[i]);
					nxz_wr32(&j->cpb.out_spbc_with_count_be, srclen);
				} else {
					nxz_wr32(&j->cpb.u.out_spbc_be, srclen);
				}
				if (tpbc > hist + n)
					cc = NXZ_CC_TPBC_GT_SPBC;
				else if (partial) {
					cc = NXZ_CC_DATA_LENGTH; ce = NXZ_CE_PARTIAL | NXZ_CE_TPBC_VALID;
				}
			}
		}
		free(tok); free(out);
		src = srcbase;
	} else if (fc == NXZ_FC_DECOMPRESS || fc == NXZ_FC_DECOMPRESS_RESUME) {
		nxo_inflate_state_t st;
		uint32_t hist = nxz_fc_is_resume(fc) ? nxz_in_histlen(&j->cpb) * 16 : 0;
		uint8_t *buf;
		if (hist > srclen) hist = srclen;
		memset(&st, 0, sizeof(st));
		if (nxz_fc_is_resume(fc)) {
			st.subc = nxz_in_subc(&j->cpb);
			st.sfbt = nxz_in_sfbt(&j->cpb);
			st.rembytecnt = nxz_in_rembytecnt(&j->cpb);
			st.dht = j->cpb.in_dht;
			st.dhtlen = (int)nxz_in_dhtlen(&j->cpb);
		}
		buf = malloc((size_t)hist + cap + 64);
		memcpy(buf, src, hist);
		nxo_inflate(src + hist, srclen - hist, buf + hist, cap, hist, &st);
		if (st.err) {
			cc = (uint32_t)st.err; ce = NXZ_CE_TERMINATE;
		} else {
			uint32_t spbc = srclen, subc = st.out_subc;
			tpbc = (uint32_t)st.tpbc;
			if (st.final_eob && subc > 0xfff8) {
				/* the 16-bit SUBC cannot describe more trailing source: leave it unread */
				uint32_t drop = (subc - 0xfff8 + 7) / 8;
				spbc -= drop; subc -= drop * 8;
			}
			scatter(&j->crb.target, buf + hist, tpbc);
			put_checksums(j, nxo_crc32(nxz_in_crc(&j->cpb), buf + hist, tpbc),
				      nxo_adler32(nxz_in_adler(&j->cpb), buf + hist, tpbc));
			nxz_wr32(&j->cpb.out_w2_be, subc & 0xffff);
			nxz_wr32(&j->cpb.out_w3_be, 0);
			nxz_putf(&j->cpb.out_w3_be, 16, 4, st.out_sfbt);
			if ((st.out_sfbt & 0xe) == 0x8)
				nxz_putf(&j->cpb.out_w3_be, 0, 16, st.out_rembytecnt);
			else if ((st.out_sfbt & 0xe) == 0xc) {
				nxz_putf(&j->cpb.out_w3_be, 0, 12, (uint32_t)st.out_dhtlen);
				memcpy(j->cpb.u.d.out_dht, st.out_dht, sizeof(st.out_dht));
			}
			nxz_wr32(&j->cpb.u.d.out_spbc_decomp_be, spbc);
			if (!(st.final_eob && subc < 8)) {
				cc = NXZ_CC_DATA_LENGTH; ce = NXZ_CE_PARTIAL | NXZ_CE_TPBC_VALID;
			}
		}
		free(buf);
	} else {
		cc = NXZ_CC_INVALID_OP; ce = NXZ_CE_TERMINATE;
	}
out:
	free(src);
	nxz_csb_complete(j, cc, ce, tpbc);
	return 0;
}
