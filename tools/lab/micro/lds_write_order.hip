// Which lane wins when several lanes of ONE ds_write_b16 store to the same LDS address?  The LZ77
// kernel's chain wave relies on the highest lane (largest position) landing last.  Random collision
// patterns, with the other waves of the workgroup hammering LDS at the same time.
// hipcc --offload-arch=gfx950 -O3 -o lds_write_order lds_write_order.hip && ./lds_write_order
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(1024) void k(const uint16_t *__restrict__ addr, uint32_t nsteps, uint32_t *__restrict__ bad)
{
	__shared__ uint16_t tab[16384];
	__shared__ uint32_t noise[8192];
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	for (int i = t; i < 16384; i += 1024) tab[i] = 0xffff;
	__syncthreads();
	if (wave == 0) {
		for (uint32_t s = 0; s < nsteps; s++) {
			const uint32_t a = addr[(size_t)blockIdx.x * nsteps * 64 + s * 64 + lane];
			tab[a] = (uint16_t)(s * 64 + lane);               // one ds_write_b16 for the 64 lanes
			__builtin_amdgcn_wave_barrier();
			// the winner must be the highest lane with this address
			const uint32_t got = __hip_atomic_load(&tab[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (a plain load would be forwarded from my own store)
			uint32_t want = lane;
			for (int o = 0; o < 64; o++) {
				const uint32_t ao = __shfl(a, o, 64);
				if (ao == a && (uint32_t)o > want) want = o;
			}
			if (got != ((s * 64 + want) & 0xffffu)) atomicAdd(bad, 1u);
		}
	} else {
		for (uint32_t s = 0; s < nsteps * 8; s++) atomicMax(&noise[(t * 37 + s * 101) & 8191], s);
	}
}

int main()
{
	const uint32_t nsteps = 4096, nblk = 256;
	uint16_t *h = (uint16_t *)malloc((size_t)nblk * nsteps * 64 * 2), *d;
	srand(1);
	for (size_t i = 0; i < (size_t)nblk * nsteps * 64; i++) {
		const int mode = (i / 64) % 4;
		h[i] = mode == 0 ? rand() % 8 : mode == 1 ? rand() % 64 : mode == 2 ? (rand() % 4) * 2 + (rand() & 1) * 4096 : rand() % 16384;
	}
	uint32_t *bad, hb = 0;
	hipMalloc(&d, (size_t)nblk * nsteps * 64 * 2); hipMalloc(&bad, 4);
	hipMemcpy(d, h, (size_t)nblk * nsteps * 64 * 2, hipMemcpyHostToDevice); hipMemset(bad, 0, 4);
	hipLaunchKernelGGL(k, dim3(nblk), dim3(1024), 0, 0, d, nsteps, bad);
	hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
	printf("steps %u x %u workgroups: %u lanes saw a winner other than the highest lane\n", nsteps, nblk, hb);
	return hb != 0;
}
