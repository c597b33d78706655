// K2's guess of a record start in the middle of the inflated stream (index.hip: the wave-cooperative guess kernel; scan.hip / index.hip: the walkers of the pieces of
// a member guess their own first record). Guessing is never trusted: the chain check (index_chain_kernel / the host's verification) accepts a tile only if every
// walker's exit is the next walker's start.
#pragma once
#ifndef NGSQC_K2_GUESS_ON_CPU   // (tests/emul/k2_guess_emul.cpp compiles this text for the CPU, with the three device keywords and v_alignbit defined away)
#include "common.h"
#endif

namespace ngsqc {

__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// cheap structural plausibility of a record header at absolute offset o (used for guessing only, never for correctness)
__device__ __noinline__ static bool plausible(const uint8_t* infl, int64_t total, int64_t o, int32_t n_ref)
{
	if (o + 36 > total) return false;
	const uint8_t* r = infl + o;
	uint32_t bs = ld32u(r);
	if (bs < 32 || bs > (1u << 28) || o + 4 + (int64_t)bs > total) return false;
	int32_t tid = (int32_t)ld32u(r + 4), pos = (int32_t)ld32u(r + 8);
	uint32_t w = ld32u(r + 12), w2 = ld32u(r + 16);
	int32_t l_seq = (int32_t)ld32u(r + 20), mtid = (int32_t)ld32u(r + 24), mpos = (int32_t)ld32u(r + 28);
	uint32_t l_name = w & 0xff, n_cigar = w2 & 0xffff;
	if (tid < -1 || tid >= n_ref || mtid < -1 || mtid >= n_ref || pos < -1 || mpos < -1 || l_seq < 0 || l_name == 0) return false;
	uint64_t need = 32ull + l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
	if (need > bs) return false;
	if (r[36 + l_name - 1] != 0) return false; // qname is NUL-terminated
	return true;
}
// The optional fields of a record must parse, tag by tag, to exactly the record's end (tag[2] type[1] value: SAM spec 4.2.4). A header check alone is not enough
// for a guess INSIDE a member (round 5): two bytes in front of a true record on the first reference the length word reads as (true block_size << 16 | 2 bytes of
// the record in front) - 20 MB - and every field test passes against so large a block_size; with records of ~330 bytes one such leap in 330 lands on a true
// record, from where the chain looks perfect (measured on the generator's data: 0.24 % of the pieces on chr1). A false header's optional fields do not parse.
// (Used for guessing only: a record htslib would read but this refuses costs the tile the general path, never a wrong result.)
__device__ static bool aux_parses(const uint8_t* p, const uint8_t* end)
{
	int budget = 4096;   // bytes of text tags looked at (long MM / MD strings: not worth a lane's time - accept)
	while (p < end)
	{
		if (p + 3 > end) return false;
		const uint8_t type = p[2]; p += 3; size_t sz;
		switch (type)
		{
			case 'A': case 'c': case 'C': sz = 1; break;
			case 's': case 'S': sz = 2; break;
			case 'i': case 'I': case 'f': sz = 4; break;
			case 'd': sz = 8; break;
			case 'Z': case 'H': { const uint8_t* q = p; while (q < end && *q && --budget > 0) ++q; if (budget <= 0) return true; if (q >= end) return false; sz = (size_t)(q - p) + 1; break; }
			case 'B':
			{
				if (p + 5 > end) return false;
				const uint8_t st = p[0]; const uint32_t n = ld32u(p + 1);
				const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
				if (!es) return false;
				sz = 5 + es * (size_t)n; break;
			}
			default: return false;
		}
		if (sz > (size_t)(end - p)) return false;
		p += sz;
	}
	return true;
}
// a plausible header whose optional fields parse and whose two successors (as far as they lie inside the tile) are plausible too
__device__ static bool plausible_chain(const uint8_t* infl, int64_t total, int64_t o, int32_t n_ref)
{
	if (!plausible(infl, total, o, n_ref)) return false;
	{
		// (only a header that claims more than 1 KiB of optional fields: the false ones claim megabytes, and a short-read record's 50 bytes are not worth ten round trips)
		const uint8_t* r = infl + o; const uint32_t bs = ld32u(r), l_name = r[12], n_cigar = ld32u(r + 16) & 0xffffu, l_seq = ld32u(r + 20);
		const uint8_t* aux = r + 36 + l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + l_seq;
		if (r + 4 + bs - aux > 1024 && !aux_parses(aux, r + 4 + bs)) return false;
	}
	for (int k = 0; k < 2; ++k)
	{
		o += 4 + (int64_t)ld32u(infl + o);       // (plausible: the record ends inside the tile)
		if (o + 36 > total) return true;          // the tile ends here, or inside the next header: nothing more to check
		const uint32_t bs = ld32u(infl + o);
		if (o + 4 + (int64_t)bs > total) return bs >= 32 && bs <= (1u << 28);   // a record cut by the tile end (plausible() refuses it for that alone)
		if (!plausible(infl, total, o, n_ref)) return false;
	}
	return true;
}


// The cheap test of the four offsets o0 .. o0 + 3 on the 32 bytes behind o0 (w[0..7]; the fields of offset o0 + t are funnel shifts of neighbouring words): the
// length word, refID, the mate's refID and position in range, a read name, and the fixed part + name + CIGAR + bases + qualities no longer than the record says. Round 5: the
// last test was added when the scan of a long read's CG:B,I array turned out to call the full test (a dozen dependent loads) on every word - small integers pass
// as length word and refID; as l_seq and block_size of one record they almost never fit. Bit t of the result: offset o0 + t is worth the full test.
__device__ __forceinline__ uint32_t cheap_candidates(const uint32_t (&w)[8], int64_t o0, int64_t hi, int64_t total, int32_t n_ref)
{
	uint32_t cand = 0;
	#pragma unroll
	for (int t = 0; t < 4; ++t)
	{
		uint32_t f[7];
		#pragma unroll
		for (int i = 0; i < 7; ++i) f[i] = t ? __builtin_amdgcn_alignbit(w[i + 1], w[i], 8u * t) : w[i];
		const uint32_t bs = f[0], l_name = f[3] & 0xffu, n_cig = f[4] & 0xffffu; const int32_t tid = (int32_t)f[1], pos = (int32_t)f[2], l_seq = (int32_t)f[5], mtid = (int32_t)f[6];
		// (the mate's refID, the window's last word: of the 7 072 offsets in 8 MiB of long reads that passed the other tests - single-base operations of a CG:B,I array
		// in front of small integers - 531 pass this one, 199 of them true records; measured on the CPU with this text, tests/test_k2_guess_emul.py)
		const bool ok = o0 + t < hi && o0 + t + 36 <= total && bs >= 32 && bs <= (1u << 28) && tid >= -1 && tid < n_ref && mtid >= -1 && mtid < n_ref && pos >= -1 && l_name != 0 && l_seq >= 0
		                && 32ull + l_name + 4ull * n_cig + ((uint64_t)(uint32_t)l_seq + 1) / 2 + (uint64_t)(uint32_t)l_seq <= (uint64_t)bs;
		cand |= ok ? 1u << t : 0u;
	}
	return cand;
}
__device__ __forceinline__ void load_window(const uint8_t* infl, int64_t total, int64_t o0, uint32_t (&w)[8])
{
	if (o0 + 32 <= total) __builtin_memcpy(w, infl + o0, 32);
	else for (int k = 0; k < 8; ++k) w[k] = o0 + 4 * k + 4 <= total ? ld32u(infl + o0 + 4 * k) : 0u;
}

// One lane looks for the first record of its piece [lo, hi) by itself (round 5: a walker of the second half of a short-read member finds its first record ~170
// bytes in - twenty 16-byte loads of three lines that it reads anyway - where the separate guess kernel cost a wave and a dozen dependent round trips per piece,
// 0.35 ms per tile and walker). Four offsets per load, the same tests as the guess kernel. -1: no record starts in the piece.
__device__ static int32_t lane_guess(const uint8_t* infl, int64_t total, int64_t lo, int64_t hi, int32_t n_ref)
{
	for (int64_t o0 = lo; o0 < hi; o0 += 4)
	{
		uint32_t w[8]; load_window(infl, total, o0, w);
		const uint32_t cand = cheap_candidates(w, o0, hi, total, n_ref);
		for (int t = 0; t < 4; ++t) if (((cand >> t) & 1u) && plausible_chain(infl, total, o0 + t, n_ref)) return (int32_t)(o0 + t - lo);
	}
	return -1;
}

} // namespace ngsqc
