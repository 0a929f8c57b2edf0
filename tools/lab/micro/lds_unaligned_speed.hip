// How fast does the LDS serve unaligned ds_read_b64 / b128 against the aligned-dwords-and-alignbyte form?
// 16 waves per CU (one workgroup of 1024), every lane a pseudo-random address; prints cycles per wave-level read.
// hipcc --offload-arch=gfx950 -O2 -o lds_unaligned_speed lds_unaligned_speed.hip && ./lds_unaligned_speed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint64_t *out, uint32_t *sink)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[65536 + 64];
	const uint32_t t = threadIdx.x;
	for (uint32_t i = t; i < 65536 + 64; i += 1024) lds[i] = (uint8_t)(i * 7 + 3);
	__syncthreads();
	uint32_t a = (t * 2654435761u) >> 16, acc = 0;
	const uint32_t *w = (const uint32_t *)lds;
	const uint64_t t0 = clock64();
	for (int it = 0; it < 256; it++) {
		a = (a * 1664525u + 1013904223u);
		const uint32_t q = (a >> 8) & 0xffff;                    // random byte address
		if (MODE == 0) {                                         // three aligned dwords + two alignbytes: 8 bytes at q
			const uint32_t qa = q >> 2, u0 = w[qa], u1 = w[qa + 1], u2 = w[qa + 2];
			acc += __builtin_amdgcn_alignbyte(u1, u0, q & 3) ^ __builtin_amdgcn_alignbyte(u2, u1, q & 3);
		} else if (MODE == 1) {                                  // one unaligned 8-byte read
			uint64_t v;
			asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(q) : "memory");
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			acc += (uint32_t)v ^ (uint32_t)(v >> 32);
		} else if (MODE == 2) {                                  // aligned 8-byte read (for reference)
			const uint2 v = *(const uint2 *)(lds + (q & ~7u));
			acc += v.x ^ v.y;
		} else if (MODE == 3) {                                  // unaligned 16-byte read
			uint4 v;
			asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(q) : "memory");
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			acc += v.x ^ v.y ^ v.z ^ v.w;
		} else if (MODE == 4) {                                  // five aligned dwords + four alignbytes: 16 bytes at q
			const uint32_t qa = q >> 2, u0 = w[qa], u1 = w[qa + 1], u2 = w[qa + 2], u3 = w[qa + 3], u4 = w[qa + 4];
			acc += __builtin_amdgcn_alignbyte(u1, u0, q & 3) ^ __builtin_amdgcn_alignbyte(u2, u1, q & 3) ^ __builtin_amdgcn_alignbyte(u3, u2, q & 3) ^ __builtin_amdgcn_alignbyte(u4, u3, q & 3);
		} else if (MODE == 5) {                                  // unaligned 4-byte read
			uint32_t v;
			asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(q) : "memory");
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			acc += v;
		}
	}
	const uint64_t t1 = clock64();
	if (t == 0) out[blockIdx.x] = t1 - t0;
	sink[blockIdx.x * 1024 + t] = acc;
}
// correctness of the 16-byte form
__global__ void chk(uint32_t *bad)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[4096];
	const uint32_t t = threadIdx.x;
	for (uint32_t i = t; i < 4096; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
	__syncthreads();
	uint32_t nbad = 0;
	for (uint32_t base = 0; base < 2048; base += 64) {
		const uint32_t a = base + t * 9 % 1500;
		uint4 v;
		asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
		const uint32_t vv[4] = { v.x, v.y, v.z, v.w };
		for (int k = 0; k < 16; k++) nbad += ((vv[k >> 2] >> (8 * (k & 3))) & 0xff) != lds[a + k];
	}
	atomicAdd(bad, nbad);
}
int main()
{
	uint64_t *d; uint32_t *s, *b;
	hipMalloc(&d, 8 * 256); hipMalloc(&s, 4 * 1024 * 256); hipMalloc(&b, 4); hipMemset(b, 0, 4);
	hipLaunchKernelGGL(chk, dim3(4), dim3(64), 0, 0, b);
	uint32_t hb; hipMemcpy(&hb, b, 4, hipMemcpyDeviceToHost);
	printf("unaligned ds_read_b128: %u wrong bytes\n", hb);
	const char *names[6] = { "3 dwords + 2 alignbyte (8 B)", "ds_read_b64 unaligned", "ds_read_b64 aligned", "ds_read_b128 unaligned", "5 dwords + 4 alignbyte (16 B)", "ds_read_b32 unaligned" };
	for (int m = 0; m < 6; m++) {
		for (int rep = 0; rep < 2; rep++) {
			if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, d, s);
			if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, d, s);
			if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, d, s);
			if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, d, s);
			if (m == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(1024), 0, 0, d, s);
			if (m == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(1024), 0, 0, d, s);
		}
		uint64_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
		double sum = 0; for (int i = 0; i < 256; i++) sum += (double)h[i];
		printf("%-32s %8.1f cycles per iteration of a wave (16 waves on the CU), %6.1f per CU and wave-read\n", names[m], sum / 256 / 256, sum / 256 / 256 / 16);
	}
	return 0;
}
