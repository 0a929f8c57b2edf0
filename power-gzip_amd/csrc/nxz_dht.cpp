// nxz_dht.cpp -- host-side dynamic-Huffman table (DHT) generation and selection policy.
//
// Product code (C++), the counterpart of the reference's lib/nx_dhtgen.c and lib/nx_dht.c:
//   nxz_dhtgen()           same result, bit for bit, as dhtgen() (lib/nx_dhtgen.c:945-1034):
//                          length_limit :295, ordering :323-348, two-queue merge with leaf
//                          preference :418-571, depth limit retry schedule :576-595, header
//                          encoder with the fixed code-length code :610-915.
//   nxz_dht_state / nxz_dht_lookup()
//                          table selection for the NEXT job from the counts of the LAST job
//                          (lib/nx_dht.c:568-676): default table for the first job, reuse of
//                          the last table for 512 KiB of source, cache keyed by the two most
//                          frequent literals and the most frequent length symbol, clock
//                          replacement, else generate.
// Parity is pinned by tests/golden/dhtgen_vectors.json (made with the reference's own file).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#include "nxz_host.h"
#include "../../include/nxz_config.h"

namespace {

struct BitSink {
	uint8_t *p; uint64_t acc = 0; int nacc = 0; int total = 0;
	explicit BitSink(uint8_t *out) : p(out) {}
	void put(uint32_t v, int n)
	{
		acc |= (uint64_t)v << nacc; nacc += n; total += n;
		for (; nacc >= 8; nacc -= 8, acc >>= 8) *p++ = (uint8_t)acc;
	}
	void flush() { if (nacc) *p++ = (uint8_t)acc; }
};

// Code lengths by Huffman's two-queue construction.  Returns the deepest leaf.
int code_lengths(const uint32_t *hist, int nsym, uint32_t *len_out)
{
	struct Leaf { uint32_t sym, cnt; };
	std::vector<Leaf> leaves;
	leaves.reserve(nsym);
	for (int i = 0; i < nsym; i++) if (hist[i]) leaves.push_back({(uint32_t)i, hist[i]});
	// counts compare as int (the reference casts them), symbol breaks ties
	std::sort(leaves.begin(), leaves.end(), [](const Leaf &a, const Leaf &b) {
		return (int)a.cnt != (int)b.cnt ? (int)a.cnt < (int)b.cnt : a.sym < b.sym;
	});
	const int n = (int)leaves.size();
	if (n == 0) return 0;
	if (n == 1) { len_out[leaves[0].sym] = 1; return 1; }   // the reference is undefined here

	// internal nodes are created in non-decreasing weight order: a plain array is the second queue
	struct Node { uint32_t cnt; int kid[2]; bool leaf[2]; int depth; };
	std::vector<Node> nodes;
	nodes.reserve(n);
	int li = 0, ni = 0;
	while ((n - li) + ((int)nodes.size() - ni) > 1) {
		Node nd{};
		for (int k = 0; k < 2; k++) {
			bool take_leaf = li < n && (ni >= (int)nodes.size() || leaves[li].cnt <= nodes[ni].cnt);
			if (take_leaf) { nd.cnt += leaves[li].cnt; nd.kid[k] = (int)leaves[li].sym; nd.leaf[k] = true; li++; }
			else { nd.cnt += nodes[ni].cnt; nd.kid[k] = ni; nd.leaf[k] = false; ni++; }
		}
		nodes.push_back(nd);
	}
	// children are always created before their parent: walk from the root (last node) down
	int deepest = 0;
	nodes.back().depth = 1;
	for (int i = (int)nodes.size() - 1; i >= 0; i--) {
		int d = nodes[i].depth;
		deepest = std::max(deepest, d);
		for (int k = 0; k < 2; k++) {
			if (nodes[i].leaf[k]) len_out[nodes[i].kid[k]] = (uint32_t)d;
			else nodes[nodes[i].kid[k]].depth = d < 31 ? d + 1 : 31;
		}
	}
	return deepest;
}

void limited_lengths(uint32_t *hist, int nsym, uint32_t *len_out)
{
	int limit = 1 << 14, deepest;
	do {
		uint64_t sum = 0;
		for (int i = 0; i < nsym; i++) sum += hist[i];
		uint64_t div = (sum + limit - 1) / limit;
		if (div) for (int i = 0; i < nsym; i++) hist[i] = (uint32_t)((hist[i] + div - 1) / div);
		limit = limit * 3 / 4;
		deepest = code_lengths(hist, nsym, len_out);
	} while (deepest > 15);
}

uint16_t reverse_bits(uint32_t v, int n)
{
	uint32_t r = 0;
	for (int i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i);
	return (uint16_t)r;
}

} // namespace

extern "C" void nxz_fill_zero_lzcounts(uint32_t *ll, uint32_t *d, uint32_t val)
{
	if (ll) for (int i = 0; i < 286; i++) if (!ll[i]) ll[i] = val;
	if (d) for (int i = 0; i < 30; i++) if (!d[i]) d[i] = val;
}

extern "C" int nxz_dhtgen(uint32_t *lhist, int num_lhist, uint32_t *dhist, int num_dhist,
			  uint8_t *dht, int *dht_num_bytes, int *dht_num_valid_bits)
{
	// the code-length alphabet uses one fixed code (lengths as in lib/nx_dhtgen.c:628-648)
	static const uint8_t kClLen[19] = { 5, 7, 6, 5, 5, 4, 4, 3, 3, 3, 3, 4, 5, 5, 4, 7, 6, 5, 6 };
	static const uint8_t kOrder[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
	uint16_t clcode[19];
	{
		uint32_t cnt[8] = {0}, next[8] = {0}, c = 0;
		for (int i = 0; i < 19; i++) cnt[kClLen[i]]++;
		for (int b = 1; b <= 7; b++) { c = (c + cnt[b - 1]) << 1; next[b] = c; }
		for (int i = 0; i < 19; i++) clcode[i] = reverse_bits(next[kClLen[i]]++, kClLen[i]);
	}
	const int nhlit = std::max(num_lhist, 257);
	std::vector<uint32_t> lens(286 + 30 + 2, 0);
	limited_lengths(lhist, num_lhist, lens.data());
	if (num_dhist > 1) limited_lengths(dhist, num_dhist, lens.data() + nhlit);
	else { lens[nhlit] = 1; num_dhist = 1; dhist[0] = 1; }      // lib/nx_dhtgen.c:985-1014
	const int total = nhlit + num_dhist;

	BitSink out(dht);
	out.put((uint32_t)(nhlit - 257), 5);
	out.put((uint32_t)(num_dhist - 1), 5);
	out.put(19 - 4, 4);
	for (int i = 0; i < 19; i++) out.put(kClLen[kOrder[i]], 3);
	auto sym = [&](int s) { out.put(clcode[s], kClLen[s]); };

	// run-length coder: emit runs exactly as the reference's state machine does
	// (:758-910): a non-zero run is value + 16(3..6)*; a zero run uses 17 for 3..10 and 18 for
	// 11..138, where a run that outgrows 138 restarts counting at 1.
	int i = 0;
	while (i < total) {
		uint32_t v = lens[i];
		int run = 1;
		while (i + run < total && lens[i + run] == v) run++;
		if (v != 0) {
			sym((int)v);
			int rest = run - 1;
			while (rest >= 6) { sym(16); out.put(3, 2); rest -= 6; }
			if (rest >= 3) { sym(16); out.put((uint32_t)(rest - 3), 2); }
			else for (int k = 0; k < rest; k++) sym((int)v);
		} else {
			int rest = run;
			while (rest > 138) { sym(18); out.put(138 - 11, 7); rest -= 138; }
			if (rest >= 11) { sym(18); out.put((uint32_t)(rest - 11), 7); }
			else if (rest >= 3) { sym(17); out.put((uint32_t)(rest - 3), 3); }
			else for (int k = 0; k < rest; k++) sym(0);
		}
		i += run;
	}
	out.flush();
	*dht_num_bytes = (out.total + 7) / 8;
	*dht_num_valid_bits = out.total % 8;
	return 0;
}

// ---------------------------------------------------------------------------
// selection policy
// ---------------------------------------------------------------------------
// The reference's 35 canned tables (lib/nx_dht_builtin.c:104-840), carried as data: tables made
// from typical inputs that serve a job whose two most frequent literal/length symbols are a
// table's keys (dht_search_builtin, lib/nx_dht.c:401-432); entry 0 is the default of a stream's
// first job (lib/nx_dht.c:578-583).
namespace {
struct Builtin { uint32_t dhtlen; int key[3]; uint8_t dht[NXZ_DHT_MAXSZ]; };
const Builtin kBuiltin[] = {
#include "nxz_dht_builtin.inc"
};
constexpr int kNumBuiltin = (int)(sizeof(kBuiltin) / sizeof(kBuiltin[0]));
static_assert(kNumBuiltin == 35, "lib/nx_dht_builtin.c has 35 tables");
}

extern "C" int nxz_dht_builtin_count(void) { return kNumBuiltin; }
extern "C" int nxz_dht_builtin_get(int i, uint8_t *dht_out, uint32_t *dhtlen_out, int key[3])
{
	if (i < 0 || i >= kNumBuiltin) return -1;
	if (dht_out) memcpy(dht_out, kBuiltin[i].dht, (kBuiltin[i].dhtlen + 7) / 8);
	if (dhtlen_out) *dhtlen_out = kBuiltin[i].dhtlen;
	if (key) memcpy(key, kBuiltin[i].key, sizeof(kBuiltin[i].key));
	return 0;
}

struct nxz_dht_state {
	struct Entry { bool valid = false; int accessed = 0; int key[3] = {-1, -1, -1}; uint32_t dhtlen = 0; uint8_t dht[NXZ_DHT_MAXSZ]; };
	Entry builtin[kNumBuiltin];
	int last_builtin = 0;                          // where the next search of the canned tables starts (:408-411)
	Entry cache[128];
	int clock = 0;
	const Entry *last = nullptr;
	long bytes_since_refresh = 0;
};

// The cache keys of a table: the three most frequent symbols among the literals (dht_config bit 0
// clear, the default) or among literals and lengths (bit 0 set), found with the reference's single
// scan (lib/nx_dht.c:169-237): a new maximum pushes the old one to second place but leaves the
// third untouched, and only strictly greater counts move anything (so ties keep the lower symbol).
extern "C" void nxz_dht_top_keys(const uint32_t *ll, int lit_and_len, int key[3])
{
	const int scan = lit_and_len ? 286 : 256;
	uint32_t cnt[3] = {0, 0, 0};
	key[0] = key[1] = key[2] = -1;
	for (int i = 0; i < scan; i++) {
		const uint32_t c = ll[i];
		if (c > cnt[0]) { cnt[1] = cnt[0]; key[1] = key[0]; cnt[0] = c; key[0] = i; }
		else if (c > cnt[1]) { cnt[2] = cnt[1]; key[2] = key[1]; cnt[1] = c; key[1] = i; }
		else if (c > cnt[2]) { cnt[2] = c; key[2] = i; }
	}
}

extern "C" nxz_dht_state *nxz_dht_begin(void)
{
	auto *s = new nxz_dht_state();
	for (int i = 0; i < kNumBuiltin; i++) {
		auto &e = s->builtin[i];
		e.valid = true;
		e.dhtlen = kBuiltin[i].dhtlen;
		memcpy(e.key, kBuiltin[i].key, sizeof(e.key));
		memset(e.dht, 0, sizeof(e.dht));
		memcpy(e.dht, kBuiltin[i].dht, (e.dhtlen + 7) / 8);
	}
	return s;
}

extern "C" void nxz_dht_end(nxz_dht_state *s) { delete s; }

extern "C" nxz_dht_state *nxz_dht_copy(const nxz_dht_state *s)
{
	auto *c = new nxz_dht_state(*s);
	if (s->last) {
		if (s->last >= s->builtin && s->last < s->builtin + kNumBuiltin) c->last = &c->builtin[s->last - s->builtin];
		else c->last = &c->cache[s->last - s->cache];
	}
	return c;
}

// Chooses the table for the next job.  counts == nullptr: first job -> default table.
// source_bytes: size of the job the counts came from.
extern "C" void nxz_dht_lookup(nxz_dht_state *s, const uint32_t *counts, long source_bytes,
			       uint8_t *dht_out, uint32_t *dhtlen_out)
{
	const nxz_dht_state::Entry *e = nullptr;
	if (!counts) {
		e = &s->builtin[0];                                // the first canned table is the default (lib/nx_dht.c:578-583)
		s->bytes_since_refresh = 0;
	} else {
		s->bytes_since_refresh += source_bytes;
		if (s->last && source_bytes != 0 && s->bytes_since_refresh < 512 * 1024) {
			e = s->last;                                   // amortise the lookup (lib/nx_dht.c:480-566)
		} else {
			s->bytes_since_refresh = source_bytes;         // :543
			int key[3];
			nxz_dht_top_keys(counts, nxz_config()->dht & 1, key);
			// a cached table serves when its two top symbols are the job's (dht_search_cache, :434-478)
			for (auto &c : s->cache)
				if (c.valid && c.key[0] == key[0] && c.key[1] == key[1]) { c.accessed = 1; e = &c; break; }
			// ... or a canned one with the job's two top symbols (dht_search_builtin, :401-432), the search
			// starting at the table that served last
			if (!e) {
				for (int i = 0, k = s->last_builtin % kNumBuiltin; i < kNumBuiltin; i++, k = (k + 1) % kNumBuiltin) {
					auto &b = s->builtin[k];
					if (b.valid && b.key[0] == key[0] && b.key[1] == key[1]) { e = &b; s->last_builtin = k; break; }
				}
			}
			if (!e) {
				// clock replacement, then generate a universal table (no missing codes)
				for (;;) {
					auto &c = s->cache[s->clock];
					s->clock = (s->clock + 1) % 128;
					if (!c.valid || !c.accessed) {
						uint32_t ll[286], d[30];
						memcpy(ll, counts, sizeof(ll));
						memcpy(d, counts + 286, sizeof(d));
						nxz_fill_zero_lzcounts(ll, d, 1);
						int nb = 0, vb = 0;
						nxz_dhtgen(ll, 286, d, 30, c.dht, &nb, &vb);
						c.dhtlen = (uint32_t)(nb * 8 - (vb ? 8 - vb : 0));
						memcpy(c.key, key, sizeof(key));
						c.valid = true; c.accessed = 1;
						e = &c;
						break;
					}
					c.accessed = 0;
				}
			}
		}
	}
	s->last = e;
	memcpy(dht_out, e->dht, (e->dhtlen + 7) / 8);
	*dhtlen_out = e->dhtlen;
}

// ---------------------------------------------------------------------------------------------
// Batched table builder for the device-resident path: counts[n][316] (what the COUNT function
// codes write: 286 literal/length + 30 distance counts) -> one table per job, on `nthreads` host
// threads.  Zero counts are raised to 1 first (fill_zero_lzcounts, lib/nx_dhtgen.c:235) so that
// a table can also code a block it was not made from.
#include <thread>
extern "C" int nxz_dhtgen_batch(const uint32_t *counts, size_t n, nxz_batch_dht_t *tables, int nthreads)
{
	if (!counts || !tables) return -1;
	if (nthreads < 1) nthreads = 1;
	auto work = [&](size_t lo, size_t hi) {
		for (size_t i = lo; i < hi; i++) {
			uint32_t ll[286], d[30];
			memcpy(ll, counts + i * 316, sizeof(ll));
			memcpy(d, counts + i * 316 + 286, sizeof(d));
			ll[256] = 1;
			nxz_fill_zero_lzcounts(ll, d, 1);
			int nb = 0, vb = 0;
			memset(tables[i].dht, 0, sizeof(tables[i].dht));
			nxz_dhtgen(ll, 286, d, 30, tables[i].dht, &nb, &vb);
			tables[i].dhtlen = (uint32_t)(nb * 8 - (vb ? 8 - vb : 0));
		}
	};
	std::vector<std::thread> th;
	const size_t per = (n + (size_t)nthreads - 1) / (size_t)nthreads;
	for (int k = 0; k < nthreads; k++) {
		size_t lo = (size_t)k * per, hi = lo + per < n ? lo + per : n;
		if (lo < hi) th.emplace_back(work, lo, hi);
	}
	for (auto &t : th) t.join();
	return 0;
}
