// nxz_dhtgen.hip -- dynamic-Huffman table generation on the device (gfx950, wave64).
//
// The reference builds the table of a dynamic-Huffman block on the host: dhtgen()
// (/root/reference lib/nx_dhtgen.c:945-1034) turns the 286 + 30 LZ symbol counts that a
// COMPRESS_*_COUNT job returned into code lengths and the RFC 1951 3.2.7 header, and the next
// job is encoded with it.  Here the same function runs on the GPU, one WAVEFRONT per table,
// so that a batch gets an exact table per block without a host round trip:
//
//   length_limit :295      counts scaled so that their sum is <= limit (2^14, x3/4 per retry)
//   ordering :323-348      rank sort of the used symbols by (count, symbol): every lane counts
//                          the keys below its own (keys are broadcast 16 bytes at a time)
//   two-queue merge :418   one lane; leaf preferred on ties (:484); a node remembers nothing
//                          but its weight, the children remember their parent
//   depth :360-394         pointer jumping over the parent links: 5 rounds cover depth < 32,
//                          deeper trees are retried anyway (limit schedule :576-595)
//   canonical codes        RFC 1951 3.2.2; a symbol's code = first code of its length + the
//                          number of lower symbols of that length (ballots)
//   header :610-915        fixed code-length code (:628-648); the run-length coder is position
//                          parallel: from the start and the end of its run every position of the
//                          length array knows which symbol -- if any -- the reference's state
//                          machine (:758-910) emits there
//
// The result must equal nxz_dhtgen() (csrc/nxz_dht.cpp) / the reference bit for bit:
// tests/test_gpu_dhtgen.py against tests/golden/dhtgen_vectors.json and the host generator;
// tests/dht_model.py restates this file's formulation for the CPU suite.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nxz_device.h"
#include "nxz_dhtgen_dev.h"

namespace nxzd {

struct NoBar { static constexpr bool none = true; __device__ __forceinline__ void operator()() const {} };

// One wavefront per table: counts[316] (LL then D; EOB forced to 1) -> prepared table and/or
// the caller-visible bit string.
__global__ __launch_bounds__(64) void dhtgen_kernel(const uint32_t *__restrict__ counts_, uint32_t n,
						    nxz_dht_prepared_t *__restrict__ prepared_, nxz_batch_dht_t *__restrict__ tables_)
{
	__shared__ __attribute__((aligned(16))) uint8_t w[WAVE_LDS];
	const int lane = threadIdx.x;
	const uint32_t bid = blockIdx.x;
	if (bid >= n) return;
	// the reference's callers count EOB once (lib/nx_dht.c:189-195): hist[256] is rewritten by the kernels that count, a
	// caller's array is taken as it is
	const uint32_t NXZ_GLOBAL_AS *hist = (const uint32_t NXZ_GLOBAL_AS *)counts_ + (size_t)bid * NTOT;
	NoBar nb;
	dhtgen_wave(w, hist, prepared_ ? prepared_ + bid : nullptr, tables_ ? tables_ + bid : nullptr, lane, nb);
}

} // namespace nxzd

extern "C" int nxz_launch_dhtgen(const uint32_t *counts, size_t n, nxz_dht_prepared_t *prepared,
				 nxz_batch_dht_t *tables, hipStream_t stream)
{
	if (n == 0) return 0;
	hipLaunchKernelGGL(nxzd::dhtgen_kernel, dim3((unsigned)n), dim3(64), 0, stream, counts, (uint32_t)n, prepared, tables);
	return (int)hipGetLastError();
}
