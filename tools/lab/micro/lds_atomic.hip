// micro-benchmark: cost of LDS operations issued by ONE wave (random addresses vs. linear)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE> __global__ void k(unsigned long long *out, const unsigned *idx, unsigned *sink)
{
	__shared__ unsigned head[8192];
	const unsigned lane = threadIdx.x;
	for (int i = lane; i < 8192; i += 64) head[i] = 0;
	unsigned p[16];
	for (int i = 0; i < 16; i++) p[i] = idx[lane * 16 + i] & 8191;
	__syncthreads();
	unsigned acc = 0;
	unsigned long long t0 = clock64();
	for (int rep = 0; rep < 64; rep++) {
#pragma unroll
		for (int u = 0; u < 16; u++) {
			unsigned *slot = &head[(p[u] + rep * 17) & 8191];
			if (MODE == 0) atomicMax(slot, lane + rep);
			else if (MODE == 1) *(volatile unsigned *)slot = lane + rep;
			else if (MODE == 2) acc += *(volatile unsigned *)slot;
			else if (MODE == 3) acc += atomicMax(slot, lane + rep);
			else if (MODE == 4) { acc += __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __builtin_amdgcn_wave_barrier(); atomicMax(slot, lane + rep); __builtin_amdgcn_wave_barrier(); }
		}
	}
	__syncthreads();
	unsigned long long t1 = clock64();
	if (lane == 0) out[MODE] = t1 - t0;
	sink[lane] = acc + head[lane];
}
int main()
{
	unsigned long long *out; unsigned *idx, *sink;
	hipMalloc(&out, 64); hipMalloc(&idx, 64 * 16 * 4); hipMalloc(&sink, 256);
	unsigned h[1024];
	for (int lin = 0; lin < 2; lin++) {
		for (int i = 0; i < 1024; i++) h[i] = lin ? (unsigned)((i / 16) + (i % 16) * 64) : (unsigned)rand();
		hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice);
		k<0><<<1, 64>>>(out, idx, sink); k<1><<<1, 64>>>(out, idx, sink); k<2><<<1, 64>>>(out, idx, sink);
		k<3><<<1, 64>>>(out, idx, sink); k<4><<<1, 64>>>(out, idx, sink);
		unsigned long long r[8];
		hipMemcpy(r, out, 64, hipMemcpyDeviceToHost);
		const char *names[5] = { "ds_max (no return)", "ds_write_b32", "ds_read_b32", "ds_max_rtn", "read + max pair" };
		for (int m = 0; m < 5; m++) printf("%s  %-20s %.1f cycles per wave-op\n", lin ? "linear" : "random", names[m], (double)r[m] / (64 * 16));
	}
	return 0;
}
