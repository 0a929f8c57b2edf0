// What does an LDS atomic add cost when many lanes of a wave hit the same word?  (the literal histogram of the
// LZ77 kernel: in binaries a third of the literals are zero bytes)  16 waves on the CU; K distinct words per wave-instruction.
// hipcc --offload-arch=gfx950 -O2 -o lds_atomic_same lds_atomic_same.hip && ./lds_atomic_same
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(1024) void k(uint64_t *out, uint32_t *sink, uint32_t K)
{
	__shared__ uint32_t hist[512];
	const uint32_t t = threadIdx.x, lane = t & 63;
	if (t < 512) hist[t] = 0;
	__syncthreads();
	uint32_t a = (lane * 2654435761u) >> 7;
	const uint64_t t0 = clock64();
	for (int it = 0; it < 256; it++) {
		a = a * 1664525u + 1013904223u;
		const uint32_t slot = K >= 64 ? (a >> 9) & 255 : ((lane % K) * 37 + (it & 3)) & 255;
		atomicAdd(&hist[slot], 1u);
	}
	__syncthreads();
	const uint64_t t1 = clock64();
	if (t == 0) out[blockIdx.x] = t1 - t0;
	sink[blockIdx.x * 1024 + t] = hist[t & 511];
}
int main()
{
	uint64_t *d; uint32_t *s;
	(void)hipMalloc(&d, 8 * 256); (void)hipMalloc(&s, 4 * 1024 * 256);
	const uint32_t ks[] = { 1, 2, 4, 8, 16, 32, 64 };
	for (uint32_t K : ks) {
		for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, s, K);
		uint64_t h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
		double sum = 0; for (int i = 0; i < 256; i++) sum += (double)h[i];
		printf("%2u distinct words among 64 lanes%s: %7.1f CU-cycles per wave-level atomic add\n", K, K >= 64 ? " (random of 256)" : "", sum / 256 / 256 / 16);
	}
	return 0;
}
