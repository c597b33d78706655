// Runs the two K1 kernels (ngs-bits_amd/csrc/k1_kernels.h, the text the GPU library compiles) under the wave emulator on
// the CPU: tests/test_k1_emul.py compares the result with zlib. Test infrastructure - see wave_emul.h.
#include "wave_emul.h"
#include "../../ngs-bits_amd/csrc/k1_kernels.h"
#include <algorithm>
#include <numeric>

using namespace ngsqc;

extern "C" {

// comp: compressed image with at least 64 readable bytes behind the last payload. blocks[i]: payload position / output position.
// tok_mode: 0 = the library's budget (clen + 64 words per member), 1 = worst-case budget. p1_wgs / p2_wgs: grid sizes (0 = one lane /
// one wave per member). order_mode: 1 = members handed out largest compressed size first (what the library does).
// stats[0] = rendezvous count, stats[1] = token words written, stats[2] = no-op words among them, stats[3] = rendezvous count of phase 1
// (2.25 per decoder trip and wave: two ballots per trip, one more per service block).
int k1_emul_inflate(const uint8_t* comp, const BlockDesc* blocks, int64_t n, uint8_t* out, BlockStatus* st, int park_hi, int tok_mode,
                    int p1_wgs, int p2_wgs, int order_mode, uint64_t* stats)
{
	if (n <= 0) return 0;
	std::vector<uint64_t> tok_off((size_t)n + 1, 0);
	for (int64_t i = 0; i < n; ++i)
	{
		const uint64_t cap = tok_mode ? 4ull * blocks[i].usize + 64 : (uint64_t)blocks[i].clen + 64;
		tok_off[(size_t)i + 1] = tok_off[(size_t)i] + ((cap + 3) & ~3ull);
	}
	std::vector<uint32_t> tok((size_t)tok_off[(size_t)n] + 16, 0xdeadbeefu), tok_count((size_t)n + 8, 0), order((size_t)n);
	std::iota(order.begin(), order.end(), 0u);
	if (order_mode) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return blocks[a].clen > blocks[b].clen; });
	for (int64_t i = 0; i < n; ++i) { st[i].produced = 0; st[i].error = 0xffffffffu; }
	unsigned long long work = 0;
	const int64_t need1 = (n + 63) / 64, g1 = p1_wgs > 0 ? std::min<int64_t>(p1_wgs, need1) : need1;
	wv::emu().n_sync = 0;
	for (int64_t blk = 0; blk < g1; ++blk)
		wv::run_block(blk, g1, [&] {
			k1::huff_tokens_kernel(comp, blocks, n, tok_off.data(), tok.data(), tok_count.data(), st, &work, order_mode ? order.data() : nullptr, park_hi);
		});
	const uint64_t sync1 = wv::emu().n_sync;
	uint64_t words = 0, noops = 0;
	for (int64_t i = 0; i < n; ++i)
	{
		if (st[i].error) continue;
		words += tok_count[(size_t)i];
		for (uint32_t k = 0; k < tok_count[(size_t)i]; ++k) noops += tok[(size_t)tok_off[(size_t)i] + k] == K1_TOK_NOOP;
	}
	const int64_t g2 = p2_wgs > 0 ? std::min<int64_t>(p2_wgs, n) : n;
	for (int64_t blk = 0; blk < g2; ++blk)
		wv::run_block(blk, g2, [&] {
			k1::lz77_groups_kernel(tok.data(), tok_off.data(), tok_count.data(), blocks, n, out, st);
		});
	if (stats) { stats[0] = wv::emu().n_sync; stats[1] = words; stats[2] = noops; stats[3] = sync1; }
	return 0;
}

}
