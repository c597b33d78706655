// K2's guess of a record start (ngs-bits_amd/csrc/k2_guess.h: the text the GPU library compiles into index_guess_kernel / index_guess_wide_kernel / lane_guess) on
// the CPU: tests/test_k2_guess_emul.py runs it over real inflated BAM streams. Test infrastructure, never linked into the library.
#include <cstddef>
#include <cstdint>
#include <cstring>
#define NGSQC_K2_GUESS_ON_CPU
#define __device__
#define __forceinline__ inline
#define __noinline__
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t shift) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (shift & 31u)); }   // v_alignbit_b32
#include "../../ngs-bits_amd/csrc/k2_guess.h"

using namespace ngsqc;

extern "C" {

// for every offset of `at` (n of them): bit 0 = the cheap window test lets it through (as one of the four offsets of the window that starts t bytes in front of it, every
// t = 0 .. 3 must agree), bit 1 = plausible(), bit 2 = plausible_chain()
void k2_guess_classify(const uint8_t* infl, int64_t total, int32_t n_ref, const int64_t* at, int64_t n, uint8_t* out)
{
	for (int64_t i = 0; i < n; ++i)
	{
		const int64_t o = at[i]; bool cheap = true;
		for (int t = 0; t < 4; ++t)
		{
			const int64_t o0 = o - t; if (o0 < 0) continue;
			uint32_t w[8]; load_window(infl, total, o0, w);
			cheap = cheap && ((cheap_candidates(w, o0, total, total, n_ref) >> t) & 1u);
		}
		out[i] = (uint8_t)((cheap ? 1 : 0) | (plausible(infl, total, o, n_ref) ? 2 : 0) | (plausible_chain(infl, total, o, n_ref) ? 4 : 0));
	}
}

// the first guessed record start in [lo[i], hi[i]) for every i, as an absolute offset (-1: none) - what a walker of that piece would start from
void k2_guess_first(const uint8_t* infl, int64_t total, int32_t n_ref, const int64_t* lo, const int64_t* hi, int64_t n, int64_t* out)
{
	for (int64_t i = 0; i < n; ++i) { const int32_t g = lane_guess(infl, total, lo[i], hi[i], n_ref); out[i] = g < 0 ? -1 : lo[i] + g; }
}

// every offset in [lo, hi) that the cheap test lets through / that passes the chain test: how selective the two stages are (counts)
void k2_guess_counts(const uint8_t* infl, int64_t total, int32_t n_ref, int64_t lo, int64_t hi, int64_t* n_cheap, int64_t* n_chain)
{
	*n_cheap = 0; *n_chain = 0;
	for (int64_t o0 = lo; o0 < hi; o0 += 4)
	{
		uint32_t w[8]; load_window(infl, total, o0, w);
		const uint32_t cand = cheap_candidates(w, o0, hi, total, n_ref);
		for (int t = 0; t < 4; ++t) if ((cand >> t) & 1u) { ++*n_cheap; if (plausible_chain(infl, total, o0 + t, n_ref)) ++*n_chain; }
	}
}

} // extern "C"
