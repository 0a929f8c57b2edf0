// hip_cpu_shim.h -- TEST INFRASTRUCTURE: just enough of the HIP device language to run ONE workgroup of a kernel on the
// CPU, a real OS thread per lane (so races and missing barriers show, also under -fsanitize=thread / address), for the
// kernels that are written against it (power-gzip_amd/csrc/nxz_inflate_wg.hip).  Nothing in the product includes this.
//   - __syncthreads / __syncthreads_or: a barrier over the workgroup's threads
//   - __ballot / __any / __shfl / __shfl_up / readlane / readfirstlane: through a barrier over the 64 threads of a wave;
//     every lane of the wave must take part (the kernels written against this shim call them in wave-uniform flow only)
//   - atomics on "LDS" (static storage: one workgroup at a time) and on "global" memory: the __atomic builtins
#ifndef HIP_CPU_SHIM_H
#define HIP_CPU_SHIM_H
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <thread>
#include <vector>

#define NXZ_CPU_SIM 1
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct __attribute__((aligned(16))) uint4 { uint32_t x, y, z, w; };
struct sim_dim3 { unsigned x, y, z; };
static thread_local sim_dim3 threadIdx, blockIdx;
static sim_dim3 blockDim, gridDim;

namespace hipsim {
constexpr int WAVE = 64, MAXW = 16;
static pthread_barrier_t block_bar;
static pthread_barrier_t wave_bar[MAXW];
static uint64_t ballot_slot[MAXW][2];
static uint32_t shfl_slot[MAXW][2][WAVE];
static uint32_t or_slot[2];
static thread_local unsigned ballot_n, shfl_n, or_n;
inline int wave() { return (int)(threadIdx.x / WAVE); }
inline int lane() { return (int)(threadIdx.x % WAVE); }
inline void wave_sync() { pthread_barrier_wait(&wave_bar[wave()]); }
}

static inline void __syncthreads() { pthread_barrier_wait(&hipsim::block_bar); }
static inline int __syncthreads_or(int p)
{
	const unsigned k = hipsim::or_n++ & 1;
	if (p) __atomic_fetch_or(&hipsim::or_slot[k], 1u, __ATOMIC_SEQ_CST);
	__syncthreads();
	const int v = (int)__atomic_load_n(&hipsim::or_slot[k], __ATOMIC_SEQ_CST);
	if (threadIdx.x == 0) __atomic_store_n(&hipsim::or_slot[k ^ 1], 0u, __ATOMIC_SEQ_CST);
	__syncthreads();
	return v;
}
static inline uint64_t __ballot(int p)
{
	using namespace hipsim;
	const unsigned k = ballot_n++ & 1;
	const int w = wave();
	if (p) __atomic_fetch_or(&ballot_slot[w][k], 1ull << lane(), __ATOMIC_SEQ_CST);
	wave_sync();
	const uint64_t v = __atomic_load_n(&ballot_slot[w][k], __ATOMIC_SEQ_CST);
	if (lane() == 0) __atomic_store_n(&ballot_slot[w][k ^ 1], 0ull, __ATOMIC_SEQ_CST);
	wave_sync();
	return v;
}
static inline int __any(int p) { return __ballot(p) != 0; }
static inline uint32_t sim_shfl_idx(uint32_t v, int src)
{
	using namespace hipsim;
	const unsigned k = shfl_n++ & 1;
	const int w = wave();
	shfl_slot[w][k][lane()] = v;
	wave_sync();
	const uint32_t r = shfl_slot[w][k][src & 63];
	wave_sync();
	return r;
}
static inline uint32_t __shfl(uint32_t v, int src, int = 64) { return sim_shfl_idx(v, src); }
static inline int __shfl(int v, int src, int = 64) { return (int)sim_shfl_idx((uint32_t)v, src); }
static inline uint32_t __shfl_up(uint32_t v, unsigned d, int = 64)
{
	const int l = hipsim::lane();
	const uint32_t r = sim_shfl_idx(v, l >= (int)d ? l - (int)d : l);
	return r;
}
static inline int __builtin_amdgcn_readlane(int v, int l) { return (int)sim_shfl_idx((uint32_t)v, l); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return (int)sim_shfl_idx((uint32_t)v, 0); }
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
static inline uint32_t __builtin_amdgcn_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3))); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline long long clock64() { return 0; }
static inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }

template <typename T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <typename T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <typename T> static inline T atomicAnd(T *p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
template <typename T> static inline T atomicMax(T *p, T v)
{
	T o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
	while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) { }
	return o;
}
template <typename T> static inline T atomicMin(T *p, T v)
{
	T o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
	while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) { }
	return o;
}

// run one workgroup of `threads` threads (a multiple of 64) as block `b` of a grid of `nblocks`
static inline void hipsim_run_block(unsigned b, unsigned nblocks, unsigned threads, const std::function<void()> &kernel)
{
	using namespace hipsim;
	blockDim = { threads, 1, 1 };
	gridDim = { nblocks, 1, 1 };
	pthread_barrier_init(&block_bar, nullptr, threads);
	for (unsigned w = 0; w < threads / WAVE; w++) pthread_barrier_init(&wave_bar[w], nullptr, WAVE);
	memset(ballot_slot, 0, sizeof(ballot_slot));
	memset(or_slot, 0, sizeof(or_slot));
	std::vector<std::thread> th;
	th.reserve(threads);
	for (unsigned t = 0; t < threads; t++)
		th.emplace_back([=, &kernel] {
			threadIdx = { t, 0, 0 };
			blockIdx = { b, 0, 0 };
			ballot_n = shfl_n = or_n = 0;
			kernel();
		});
	for (auto &t : th) t.join();
	pthread_barrier_destroy(&block_bar);
	for (unsigned w = 0; w < threads / WAVE; w++) pthread_barrier_destroy(&wave_bar[w]);
}
#endif
