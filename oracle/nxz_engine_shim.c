/*
 * nxz_engine_shim.c -- the six transport symbols on top of the CPU engine model
 * (TEST INFRASTRUCTURE ONLY, see nxz_oracle.h).  Linked with the product's host
 * sources into oracle/libnxz_amd_model.so so that the stream layer (framing,
 * flush rules, return codes) can be tested where there is no GPU.  The product
 * library links the HIP engine instead and never sees this file.
 */
#include <time.h>
#include <unistd.h>
#include "nxz_oracle.h"
#include "../include/nxz_engine.h"

int nxo_run_job(nxz_crb_cpb_t *j);

uint64_t tb_freq = 512000000ull;

int nx_function_begin(int function, int pri, void *handle)
{
	nxz_dev_t *h = handle;
	(void)pri;
	if (function != NXZ_FUNC_COMP_GZIP || !h) return -1;
	h->function = function; h->paste_addr = (void *)1; h->fd = 1;
	return 0;
}

int nx_function_end(void *handle) { (void)handle; return 0; }
int nxu_run_job(nxz_crb_cpb_t *j, void *handle) { (void)handle; return nxo_run_job(j); }

uint64_t nx_wait_ticks(uint64_t ticks, uint64_t acc, int do_sleep)
{
	(void)do_sleep;
	usleep(1);
	return acc + ticks;
}

unsigned int __crc32_vpmsum(unsigned int crc, const unsigned char *p, unsigned long len)
{
	return ~nxo_crc32(~crc, p, len);
}
