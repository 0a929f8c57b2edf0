/*
 * nxz_wire.h -- endian-safe accessors for the job wire format of nxz_engine.h.
 * Header-only (static inline); the equivalents in the reference are the
 * getnn/putnn/get32/put32/get64 macros of inc_nx/nxu.h:719-751 with the
 * (mask, offset) tables of inc_nx/nxu.h:626-707.
 */
#ifndef NXZ_WIRE_H
#define NXZ_WIRE_H
#include <endian.h>
#include <string.h>
#include "nxz_engine.h"

static inline uint32_t nxz_rd32(const volatile uint32_t *p) { return be32toh(*p); }
static inline void nxz_wr32(volatile uint32_t *p, uint32_t v) { *p = htobe32(v); }
static inline uint64_t nxz_rd64(const volatile uint64_t *p) { return be64toh(*p); }
static inline void nxz_wr64(volatile uint64_t *p, uint64_t v) { *p = htobe64(v); }

/* bit field [shift, shift+bits) of the decoded big-endian word */
static inline uint32_t nxz_getf(const volatile uint32_t *p, unsigned shift, unsigned bits)
{
	return (nxz_rd32(p) >> shift) & ((1u << bits) - 1);
}
static inline void nxz_putf(volatile uint32_t *p, unsigned shift, unsigned bits, uint32_t v)
{
	uint32_t m = ((1u << bits) - 1) << shift;
	nxz_wr32(p, (nxz_rd32(p) & ~m) | ((v << shift) & m));
}

/* CRB / DDE */
static inline uint32_t nxz_fc(const nxz_crb_cpb_t *j) { return nxz_getf(&j->crb.fc_be, 0, 8); }
static inline void nxz_set_fc(nxz_crb_cpb_t *j, uint32_t fc) { nxz_putf(&j->crb.fc_be, 0, 8, fc); }
static inline uint32_t nxz_dde_count(const nxz_dde_t *d) { return nxz_getf(&d->count_be, 8, 8); }
static inline uint32_t nxz_dde_bytes(const nxz_dde_t *d) { return nxz_rd32(&d->bytes_be); }
static inline void *nxz_dde_addr(const nxz_dde_t *d) { return (void *)(uintptr_t)nxz_rd64(&d->addr_be); }
static inline void nxz_dde_set_direct(nxz_dde_t *d, const void *addr, uint32_t len)
{
	d->count_be = 0;
	nxz_wr32(&d->bytes_be, len);
	nxz_wr64(&d->addr_be, (uint64_t)(uintptr_t)addr);
}
static inline void nxz_dde_set_indirect(nxz_dde_t *d, const nxz_dde_t *list, uint32_t count, uint32_t total)
{
	d->count_be = 0;
	nxz_putf(&d->count_be, 8, 8, count);
	nxz_wr32(&d->bytes_be, total);
	nxz_wr64(&d->addr_be, (uint64_t)(uintptr_t)list);
}

/* CPB input */
static inline uint32_t nxz_in_histlen(const nxz_cpb_t *c) { return nxz_getf(&c->in_w2_be, 20, 12); }
static inline uint32_t nxz_in_subc(const nxz_cpb_t *c) { return nxz_getf(&c->in_w2_be, 0, 3); }
static inline uint32_t nxz_in_sfbt(const nxz_cpb_t *c) { return nxz_getf(&c->in_w3_be, 16, 4); }
static inline uint32_t nxz_in_rembytecnt(const nxz_cpb_t *c) { return nxz_getf(&c->in_w3_be, 0, 16); }
static inline uint32_t nxz_in_dhtlen(const nxz_cpb_t *c) { return nxz_getf(&c->in_w3_be, 0, 12); }
static inline uint32_t nxz_in_adler(const nxz_cpb_t *c) { return nxz_rd32(&c->in_adler_be); }
static inline uint32_t nxz_in_crc(const nxz_cpb_t *c) { return le32toh(c->in_crc_le); }
static inline void nxz_set_in_histlen(nxz_cpb_t *c, uint32_t qw) { nxz_putf(&c->in_w2_be, 20, 12, qw); }
static inline void nxz_set_in_subc(nxz_cpb_t *c, uint32_t v) { nxz_putf(&c->in_w2_be, 0, 3, v); }
static inline void nxz_set_in_sfbt(nxz_cpb_t *c, uint32_t v) { nxz_putf(&c->in_w3_be, 16, 4, v); }
static inline void nxz_set_in_rembytecnt(nxz_cpb_t *c, uint32_t v) { nxz_putf(&c->in_w3_be, 0, 16, v); }
static inline void nxz_set_in_dhtlen(nxz_cpb_t *c, uint32_t v) { nxz_putf(&c->in_w3_be, 0, 12, v); }
static inline void nxz_set_in_adler(nxz_cpb_t *c, uint32_t v) { nxz_wr32(&c->in_adler_be, v); }
static inline void nxz_set_in_crc(nxz_cpb_t *c, uint32_t v) { c->in_crc_le = htole32(v); }

/* CPB output */
static inline uint32_t nxz_out_tebc(const nxz_cpb_t *c) { return nxz_getf(&c->out_w2_be, 16, 3); }
static inline uint32_t nxz_out_subc(const nxz_cpb_t *c) { return nxz_getf(&c->out_w2_be, 0, 16); }
static inline uint32_t nxz_out_sfbt(const nxz_cpb_t *c) { return nxz_getf(&c->out_w3_be, 16, 4); }
static inline uint32_t nxz_out_rembytecnt(const nxz_cpb_t *c) { return nxz_getf(&c->out_w3_be, 0, 16); }
static inline uint32_t nxz_out_dhtlen(const nxz_cpb_t *c) { return nxz_getf(&c->out_w3_be, 0, 12); }
static inline uint32_t nxz_out_adler(const nxz_cpb_t *c) { return nxz_rd32(&c->out_adler_be); }
static inline uint32_t nxz_out_crc(const nxz_cpb_t *c) { return le32toh(c->out_crc_le); }

/* CSB */
static inline uint32_t nxz_csb_valid(const nxz_crb_cpb_t *j) { return nxz_getf(&j->crb.csb.flags_be, 31, 1); }
static inline uint32_t nxz_csb_cc(const nxz_crb_cpb_t *j) { return nxz_getf(&j->crb.csb.flags_be, 8, 8); }
static inline uint32_t nxz_csb_ce3(const nxz_crb_cpb_t *j) { return nxz_getf(&j->crb.csb.flags_be, 5, 3); }
static inline uint32_t nxz_csb_tpbc(const nxz_crb_cpb_t *j) { return nxz_rd32(&j->crb.csb.tpbc_be); }
static inline void nxz_csb_complete(nxz_crb_cpb_t *j, uint32_t cc, uint32_t ce3, uint32_t tpbc)
{
	nxz_wr32(&j->crb.csb.tpbc_be, tpbc);
	j->crb.csb.fsaddr_be = 0;
	__atomic_thread_fence(__ATOMIC_RELEASE);
	nxz_wr32(&j->crb.csb.flags_be, (1u << 31) | (cc << 8) | (ce3 << 5));
}

static inline int nxz_fc_is_compress(uint32_t fc) { return (fc & 0x10) == 0; }
static inline int nxz_fc_has_count(uint32_t fc) { return (fc & 0x10) == 0 && (fc & 0x4); }
static inline int nxz_fc_is_dht(uint32_t fc) { return (fc & 0x10) == 0 && (fc & 0x2); }
static inline int nxz_fc_is_dhtgen(uint32_t fc) { return (fc & 0x30) == 0x20; }   /* additive: NXZ_FC_COMPRESS_*_DHTGEN */
static inline int nxz_fc_is_resume(uint32_t fc) { return (fc & 0x10) ? (fc & 0x4) != 0 : (fc & 0x8) != 0; }

#endif
