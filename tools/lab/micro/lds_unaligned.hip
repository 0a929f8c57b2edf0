// Does the LDS serve ds_read_b64 / ds_read_b32 at addresses that are not multiples of their size?
// (gfx950: SH_MEM_CONFIG.alignment_mode as the HSA runtime sets it)  Prints mismatches.
// hipcc --offload-arch=gfx950 -O2 -o lds_unaligned lds_unaligned.hip && ./lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t *bad)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[4096];
	const uint32_t t = threadIdx.x;
	for (uint32_t i = t; i < 4096; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
	__syncthreads();
	uint32_t nbad = 0;
	for (uint32_t base = 0; base < 2048; base += 64) {
		const uint32_t a = base + t * 9 % 1500;          // all alignments
		uint64_t v64; uint32_t v32;
		asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v64) : "v"(a) : "memory");
		asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v32) : "v"(a) : "memory");
		uint64_t w64 = 0; uint32_t w32 = 0;
		for (int k = 0; k < 8; k++) w64 |= (uint64_t)lds[a + k] << (8 * k);
		for (int k = 0; k < 4; k++) w32 |= (uint32_t)lds[a + k] << (8 * k);
		nbad += (v64 != w64) + (v32 != w32);
	}
	atomicAdd(bad, nbad);
}
int main()
{
	uint32_t *d, h = 0;
	hipMalloc(&d, 4); hipMemset(d, 0, 4);
	hipLaunchKernelGGL(k, dim3(4), dim3(64), 0, 0, d);
	hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
	printf("unaligned LDS reads: %u mismatches\n", h);
	return h != 0;
}
