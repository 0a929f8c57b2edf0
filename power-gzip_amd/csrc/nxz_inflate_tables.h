// nxz_inflate_tables.h -- the decode tables of the stream-per-wave inflate kernel (nxz_inflate.hip) as other
// translation units see them: the batch path that cuts streams into pieces (nxz_inflate_cut.hip) reads what
// block_tables_kernel leaves about a block.
#ifndef NXZ_INFLATE_TABLES_H
#define NXZ_INFLATE_TABLES_H
#include <stdint.h>

namespace nxzi {

constexpr int LBITS = 11, DBITS = 9;

struct Huff {
	uint16_t fast[1 << LBITS];   // symbol | len << 12 ; 0 = use slow path
	uint16_t sym[288];           // symbols sorted by (len, symbol)
	uint16_t count[16];
};
struct HuffD {
	uint16_t fast[1 << DBITS];
	uint16_t sym[32];
	uint16_t count[16];
};

// The decode tables of a dynamic block as they stand in LDS, kept in device memory: built once per block
// (block_tables_kernel) for everything that starts inside the block -- the requests of token_sync_kernel,
// the pieces that begin at a cut -- to load instead of reading the header and building them again.
struct __attribute__((aligned(16))) Built {
	Huff hl;
	HuffD hd;
	uint32_t ok, bfinal, end_bit, pad;         // end_bit: first bit behind the header, counted from the request's src (0: header not in the source)
};

} // namespace nxzi
#endif
