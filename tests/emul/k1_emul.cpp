// Runs the two K1 kernels (ngs-bits_amd/csrc/k1_kernels.h, the text the GPU library compiles) under the wave emulator on
// the CPU: tests/test_k1_emul.py compares the result with zlib. Test infrastructure - see wave_emul.h.
#include "wave_emul.h"
#include "../../ngs-bits_amd/csrc/k1_kernels.h"
#include <algorithm>
#include <numeric>

using namespace ngsqc;

extern "C" {

// comp: compressed image with at least 64 readable bytes behind the last payload. blocks[i]: payload position / output position.
// tok_mode: 0 = the library's pool size (k1_pool_pages), 1 = worst-case pool (second chance), -1 = the absolute bound (third chance), >= 2: a pool of exactly tok_mode pages (overflow tests).
// p1_wgs / p2_wgs: grid sizes (0 = one lane / one wave per member). order_mode: 1 = members handed out largest compressed size first
// (what the library does).
// stats[0] = rendezvous count, stats[1] = token words written, stats[2] = no-op words among them, stats[3] = rendezvous count of phase 1,
// stats[4] = pool pages used, stats[5] = wave iterations of the symbol loop (two trips each), stats[6] = lane trips that decoded,
// stats[7] = wave iterations of the slow section, stats[8] = lane trips that waited for input, stats[13] = members whose token groups fill their last page exactly.
int k1_emul_inflate(const uint8_t* comp, const BlockDesc* blocks, int64_t n, uint8_t* out, BlockStatus* st, int park_hi, int tok_mode,
                    int p1_wgs, int p2_wgs, int order_mode, uint64_t* stats)
{
	if (n <= 0) return 0;
	uint64_t sum_c = 0, sum_u = 0;
	for (int64_t i = 0; i < n; ++i) { sum_c += blocks[i].clen; sum_u += blocks[i].usize; }
	const uint64_t pages = tok_mode >= 2 ? (uint64_t)tok_mode : (tok_mode < 0 ? k1_pool_pages_absolute(sum_c, sum_u, (uint64_t)n) : k1_pool_pages(sum_c, sum_u, (uint64_t)n, tok_mode == 1));
	std::vector<uint32_t> pool((size_t)pages * K1_PAGE_WORDS + 16, 0xdeadbeefu), tok_first((size_t)n + 8, 0xffffffffu), tok_count((size_t)n + 8, 0), order((size_t)n);
	uint32_t pool_ctr = 0;
	std::iota(order.begin(), order.end(), 0u);
	if (order_mode) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return blocks[a].clen > blocks[b].clen; });
	for (int64_t i = 0; i < n; ++i) { st[i].produced = 0; st[i].error = 0xffffffffu; }
	unsigned long long work = 0;
	const int64_t need1 = (n + 63) / 64, g1 = p1_wgs > 0 ? std::min<int64_t>(p1_wgs, need1) : need1;
	wv::emu().n_sync = 0; memset(wv::emu_stats(), 0, 8 * sizeof(uint64_t));   // (8 counters)
	for (int64_t blk = 0; blk < g1; ++blk)
		wv::run_block(blk, g1, [&] {
			k1::huff_tokens_kernel(comp, blocks, n, pool.data(), (uint32_t)pages, &pool_ctr, tok_first.data(), tok_count.data(), st, &work, order_mode ? order.data() : nullptr, park_hi);
		});
	const uint64_t sync1 = wv::emu().n_sync;
	uint64_t words = 0, noops = 0, full_pages = 0;
	for (int64_t i = 0; i < n; ++i)
	{
		if (st[i].error) continue;
		words += 4ull * tok_count[(size_t)i];
		if (tok_count[(size_t)i] && tok_count[(size_t)i] % (K1_PAGE_GROUPS - 1) == 0) ++full_pages;   // members whose token stream ends with its page
		uint32_t page = tok_first[(size_t)i];
		for (uint32_t g = 0; g < tok_count[(size_t)i]; ++g)
		{
			const uint32_t r = g % (K1_PAGE_GROUPS - 1);
			if (g && r == 0) page = pool[(size_t)page * K1_PAGE_WORDS + K1_PAGE_WORDS - 3];
			const uint32_t* w = &pool[(size_t)page * K1_PAGE_WORDS + 4 * r];
			if (w[0] == K1_TOK_TABLE) { noops += 4; continue; }
			for (int k = 0; k < 4; ++k) noops += w[k] == K1_TOK_NOOP;
		}
	}
	const int64_t g2 = p2_wgs > 0 ? std::min<int64_t>(p2_wgs, n) : n;
	for (int64_t blk = 0; blk < g2; ++blk)
		wv::run_block(blk, g2, [&] {
			k1::lz77_groups_kernel(pool.data(), tok_first.data(), tok_count.data(), blocks, n, out, st, comp);
		});
	if (stats)
	{
		stats[0] = wv::emu().n_sync; stats[1] = words; stats[2] = noops; stats[3] = sync1; stats[4] = pool_ctr;
		stats[5] = wv::emu_stats()[0]; stats[6] = wv::emu_stats()[1]; stats[7] = wv::emu_stats()[2]; stats[8] = wv::emu_stats()[3]; stats[9] = wv::emu_stats()[4]; stats[10] = wv::emu_stats()[5]; stats[11] = wv::emu_stats()[6]; stats[12] = wv::emu_stats()[7]; stats[13] = full_pages;
	}
	return 0;
}

}
