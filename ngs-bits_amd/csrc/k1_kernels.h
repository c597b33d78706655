// K1 - BGZF inflate in two phases (the kernels; inflate.hip holds the launchers).
//
//   phase 1  huff_tokens_kernel : ONE LANE PER BGZF MEMBER. Every lane Huffman-decodes its own raw-DEFLATE stream into tokens
//            (literal byte | match{len,dist}); 64 members advance per wave instruction. Canonical Huffman decode runs out of
//            REGISTERS (per code length one limit|delta word, the length found by a 4-level binary search with v_cndmask-selected
//            pivots); LDS holds, per lane, the symbol-order planes and a 32-byte window of the compressed input, laid out
//            element-major (word k of lane l at k*64+l: any per-lane access pattern is bank-conflict free), and once per
//            workgroup the base|extra-bits tables of the length and distance symbols.
//            Bit reader: a bit cursor into the input window; a decode reads the two window words under the cursor and funnels
//            them (v_alignbit) into 32 fresh bits - no bit buffer to shift, no refill state.
//            Tokens: every trip shifts the lane's token (or a no-op) into a four-register group; the service block that runs
//            every four trips stores the group with one 16-byte store when it holds a real token. No compaction, no
//            per-lane queue: phase 2 reads the same groups, one per lane, and skips the no-ops.
//   phase 2  lz77_groups_kernel : ONE WAVE PER MEMBER. A batch is up to 56 token groups (<= 224 tokens, <= P2_BMAX bytes): a
//            wave prefix sum places every token and sets a flag on its last byte; per 64-byte chunk the flags become a lane
//            mask (ballot) from which every OUTPUT BYTE gets its owner token (v_mbcnt), and its source in periodic form
//            (i mod dist). Pass 1 classifies all bytes of the batch and issues every gather that reaches behind the batch
//            (HBM) in one go; pass 2 resolves the chunks front to back: a source in an earlier chunk comes from the LDS
//            staging bytes, one in the same chunk from the source LANE. One HBM round trip per batch instead of one per 64 bytes.
//
// Written against the wave vocabulary of wave.h only (see there). Integer work, no MFMA. RFC 1951; the reference reaches
// zlib's inflate through htslib's bgzf.c under BamReader::getNextAlignment (src/cppNGS/BamReader.h:386-392).
#pragma once
#include "k1_types.h"

namespace ngsqc { namespace k1 {

// ---------------------------------------------------------------------------------------------------------------- phase 1
constexpr int P1_SYM_W = 81;     // lit_sym : 288 x 9 bit as a byte plane (72 words) + a bit plane (9 words)
constexpr int P1_RING_W = 8;     // compressed input window (32 B)
constexpr int P1_LANE_W = P1_SYM_W + P1_RING_W;   // 89 words per lane (22.8 KB per wave); tokens and the distance symbols stay in registers
constexpr int P1_PAD_W = 192;    // + 768 B: 23 KB per one-wave workgroup (exactly six fit a CU's 160 KB, see the kernel): the window's mirror slot and the constant tables
constexpr int P1_LDS_W = P1_LANE_W * 64 + P1_PAD_W;
constexpr int P1_SERVICE = 4;    // trips between service blocks = tokens per group

enum { S_NEXT = 0, S_HDR = 1, S_P1 = 2, S_P2 = 3, S_SYM = 4, S_STORED = 5, S_FINISH = 6, S_DONE = 7 };

struct P1Lds
{
	uint32_t* base; int lane;
	K1_DEV uint32_t& at(int k) const { return base[k * 64 + lane]; }
	K1_DEV uint32_t litsym(uint32_t i) const
	{
		uint32_t lo = at((int)(i >> 2)), hi = at(72 + (int)(i >> 5));
		return ((lo >> (8 * (i & 3))) & 255u) | (((hi >> (i & 31)) & 1u) << 8);
	}
	K1_DEV void set_litsym(uint32_t i, uint32_t s) const
	{
		uint32_t& lo = at((int)(i >> 2)); uint32_t sh = 8 * (i & 3); lo = (lo & ~(255u << sh)) | ((s & 255u) << sh);
		uint32_t& hi = at(72 + (int)(i >> 5)); hi = (hi & ~(1u << (i & 31))) | ((s >> 8) << (i & 31));
	}
	// input window: word k of the member's piece stream sits in slot k & 7; slot 8 (the first 64 words of the workgroup's pad area)
	// mirrors slot 0, so that the two words under a bit cursor are always slots s and s + 1: one address, two reads
	K1_DEV uint32_t* slot(uint32_t i) const { return &at(P1_SYM_W + (int)(i & (P1_RING_W - 1))); }
	K1_DEV void stage(uint32_t wr, uint32_t a, uint32_t b, uint32_t c, uint32_t d) const   // a piece = four words, wr a multiple of 4
	{
		uint32_t* p = slot(wr & 4u); p[0] = a; p[64] = b; p[128] = c; p[192] = d;
		if ((wr & 4u) == 0) at(P1_SYM_W + P1_RING_W) = a;
	}
};

// packed per-length counters: FW bits per field, 32/FW fields per register
template <int FW, int NREG> struct Packed
{
	uint32_t r[NREG];
	K1_DEV void clear() { for (int i = 0; i < NREG; ++i) r[i] = 0; }
	K1_DEV uint32_t get_const(int idx) const { constexpr int PER = 32 / FW; return (r[idx / PER] >> (FW * (idx % PER))) & ((1u << FW) - 1u); }   // idx compile-time after unrolling
	K1_DEV uint32_t get(uint32_t idx) const
	{
		// mask-select (not an indexed read: keeps the counters in VGPRs instead of scratch memory)
		constexpr int PER = 32 / FW; uint32_t reg = idx / PER, sh = FW * (idx % PER); uint32_t v = 0;
		#pragma unroll
		for (int i = 0; i < NREG; ++i) v |= r[i] & (0u - (uint32_t)(reg == (uint32_t)i));
		return (v >> sh) & ((1u << FW) - 1u);
	}
	K1_DEV void add(uint32_t idx, uint32_t delta)
	{
		constexpr int PER = 32 / FW; uint32_t reg = idx / PER, inc = delta << (FW * (idx % PER));
		#pragma unroll
		for (int i = 0; i < NREG; ++i) r[i] += reg == (uint32_t)i ? inc : 0u;
	}
	K1_DEV void set(uint32_t idx, uint32_t v)
	{
		constexpr int PER = 32 / FW; uint32_t reg = idx / PER, sh = FW * (idx % PER), m = ((1u << FW) - 1u) << sh;
		#pragma unroll
		for (int i = 0; i < NREG; ++i) r[i] = reg == (uint32_t)i ? ((r[i] & ~m) | (v << sh)) : r[i];
	}
};
using LitCnt = Packed<10, 5>;   // indices 0..14 <-> code lengths 1..15, values <= 288
using DistCnt = Packed<6, 3>;   // values <= 32

// Distance symbols sorted by (len, sym), 30 x 5 bit in three 64-bit registers (mask-select, no indexed access). Keeping them in LDS
// instead (5 more words per lane) was measured: phase 1 did not get faster, and the 512 B it costs per wave take a workgroup slot
// from phase 2 on every CU.
struct DistSyms
{
	uint64_t q[3];
	K1_DEV void clear() { q[0] = q[1] = q[2] = 0; }
	K1_DEV uint32_t get(uint32_t i) const
	{
		uint32_t reg = (i * 43u) >> 9, sh = 5u * (i - reg * 12u);   // i / 12 for i < 36
		uint64_t v = (q[0] & (0ull - (uint64_t)(reg == 0))) | (q[1] & (0ull - (uint64_t)(reg == 1))) | (q[2] & (0ull - (uint64_t)(reg == 2)));
		return (uint32_t)(v >> sh) & 31u;
	}
	K1_DEV void set(uint32_t i, uint32_t s)
	{
		uint32_t reg = (i * 43u) >> 9, sh = 5u * (i - reg * 12u); uint64_t m = 31ull << sh, val = (uint64_t)s << sh;
		#pragma unroll
		for (int k = 0; k < 3; ++k) q[k] = reg == (uint32_t)k ? ((q[k] & ~m) | val) : q[k];
	}
};

// Branch-free canonical decode out of REGISTERS (occupancy is LDS-bound at 1.5 waves per SIMD, so VGPRs are free).
// For code length l (1..15) the word holds
//   limit_l = (first_code_l + count_l) << (15 - l)   (upper bound, left-aligned to 15 bits; non-decreasing in l)
//   delta_l = offset_l - first_code_l                (index of the length's first symbol in the sorted array minus its first code)
// With v = the next 15 stream bits MSB-first: length = 1 + #{l : v >= limit_l}, index = (v >> (15 - length)) + delta_length.
struct LimTab
{
	uint32_t w[15];   // (limit_l << 16) | (delta_l & 0xffff) for l = 1..15
	// With vx = (v << 16) | 0xffff a plain 32-bit compare vx >= w[l] is v >= limit_l. The limits are non-decreasing, so
	// n = #{l : v >= limit_l} is found by a 4-level binary search whose pivots are picked with v_cndmask from the 15
	// registers (4 compares + 11 selects instead of 15 compares + 30 selects); the last pivot the search went LEFT of
	// is w[n], the word of the code's own length, which carries the delta.
	K1_DEV int decode(uint32_t bits, uint32_t& len_out) const
	{
		const uint32_t vx = ((wv::brev(bits) >> 1) & 0x7fff0000u) | 0xffffu;
		const bool c1 = vx >= w[7];
		const uint32_t p2 = c1 ? w[11] : w[3];
		const bool c2 = vx >= p2;
		const uint32_t p3a = c2 ? w[5] : w[1], p3b = c2 ? w[13] : w[9];
		const uint32_t p3 = c1 ? p3b : p3a;
		const bool c3 = vx >= p3;
		const uint32_t q0 = c3 ? w[2] : w[0], q1 = c3 ? w[6] : w[4], q2 = c3 ? w[10] : w[8], q3 = c3 ? w[14] : w[12];
		const uint32_t r0 = c2 ? q1 : q0, r1 = c2 ? q3 : q2;
		const uint32_t p4 = c1 ? r1 : r0;
		const bool c4 = vx >= p4;
		uint32_t n = c1 ? 1u : 0u; n = 2 * n + (c2 ? 1u : 0u); n = 2 * n + (c3 ? 1u : 0u); n = 2 * n + (c4 ? 1u : 0u);
		uint32_t sel = c1 ? 0u : w[7]; sel = c2 ? sel : p2; sel = c3 ? sel : p3; sel = c4 ? sel : p4;
		len_out = n + 1;
		const int idx = (int)((vx >> 16) >> (14 - (n & 15u) < 15u ? 14 - (n & 15u) : 0u)) + (int)(int16_t)(sel & 0xffffu);
		return n >= 15 ? -1 : idx;
	}
	template <class CNT> K1_DEV void build(const CNT& c)
	{
		uint32_t code = 0, o = 0;
		#pragma unroll
		for (int l = 1; l <= 15; ++l)
		{
			const uint32_t cnt = c.get_const(l - 1);
			uint32_t lim = (code + cnt) << (15 - l); if (lim > 0x8000u) lim = 0x8000u;   // over-subscribed codes are rejected by the index checks
			w[l - 1] = (lim << 16) | ((o - code) & 0xffffu);
			o += cnt; code = (code + cnt) << 1;
		}
	}
};

K1_KERNEL(64) void huff_tokens_kernel(const uint8_t* __restrict__ comp, const BlockDesc* __restrict__ blocks, int64_t n_blocks,
                                      const uint64_t* __restrict__ tok_off, uint32_t* __restrict__ tok, uint32_t* __restrict__ tok_count,
                                      BlockStatus* __restrict__ status, unsigned long long* __restrict__ work_counter, const uint32_t* __restrict__ order, int park_hi)
{
	// 23 KB per one-wave workgroup: exactly six fit a CU's 160 KB and a seventh does not, so the workgroups of the NEXT chunk's
	// launch (queued on a second stream) take over the slots of this launch's finished waves without ever squeezing the LDS that
	// the phase-2 / CRC / scan workgroups need beside them
	K1_SHARED uint32_t lds[P1_LDS_W];
	const int lane = wv::lane();
	P1Lds L{lds, lane};
	wv::set_priority(park_hi >> 8); park_hi &= 255;   // (upper bits: wave priority, a measurement switch)
	const wv::u32x4* const comp_q = (const wv::u32x4*)comp;
	// base | extra-bits << 16 of the 29 length symbols (words 0..28) and the 30 distance symbols (words 32..61) of RFC 1951 §3.2.5
	uint32_t* const tab = lds + P1_LANE_W * 64 + 64;   // (behind the window's mirror slot)
	{
		const uint32_t i = (uint32_t)lane;
		if (i < 29)
		{
			const uint32_t eb = i < 8 ? 0u : (i == 28 ? 0u : (i - 4) >> 2);
			const uint32_t base = i < 8 ? i + 3 : (i == 28 ? 258u : ((4u + ((i - 4) & 3u)) << eb) + 3u);
			tab[i] = base | (eb << 16);
		}
		if (i < 30)
		{
			const uint32_t eb = i < 4 ? 0u : (i >> 1) - 1u;
			const uint32_t base = i < 4 ? i + 1 : ((2u + (i & 1u)) << eb) + 1u;
			tab[32 + i] = base | (eb << 16);
		}
	}
	wv::barrier();

	// ---- per-lane decoder state ----
	int state = S_NEXT;
	int64_t b = -1;                                        // members are handed out one at a time from a global counter
	uint64_t q0 = 0; uint32_t n_q = 0, next_q = 0;         // 16-byte pieces of this member: comp_q[q0 + i], i < n_q
	uint32_t abit = 0, abit_end = 0;                       // bit cursor / end of the payload, both counted from the start of piece 0
	uint32_t wr = 0;                                       // window words staged so far (word k of the piece stream sits in ring slot k & 7)
	uint32_t usize = 0;
	wv::u32x4 pf = wv::make4(0, 0, 0, 0); bool pf_valid = false;
	uint32_t* tok_ptr = nullptr; uint32_t tok_cap = 0, tok_n = 0;
	uint32_t g0 = K1_TOK_NOOP, g1 = K1_TOK_NOOP, g2 = K1_TOK_NOOP, g3 = K1_TOK_NOOP;   // the tokens of the last four trips, newest first
	uint32_t tk = K1_TOK_NOOP;                               // this trip's token
	uint32_t out_n = 0, err = 0; int bfinal = 0;
	DistSyms dsym; dsym.clear();                             // distance symbols sorted by (len, sym)
	LimTab limL, limD;                                       // decode tables of the current deflate block (registers)
	#pragma unroll
	for (int i = 0; i < 15; ++i) { limL.w[i] = 0; limD.w[i] = 0; }
	LitCnt cl; DistCnt cd; cl.clear(); cd.clear();          // code-length counts
	LitCnt ol; DistCnt od; ol.clear(); od.clear();          // placement cursors of pass 2
	uint64_t ccl_lo = 0, ccl_hi = 0; uint32_t limC[7];       // code-length alphabet: sorted symbols (19 x 5 bit), limit/delta words per length 1..7
	#pragma unroll
	for (int i = 0; i < 7; ++i) limC[i] = 0;
	uint32_t h_i = 0, h_n = 0, h_nlit = 0, h_prev = 0, hdr_abit = 0; uint32_t stored_left = 0;

	auto exhausted = [&]() -> bool { return next_q >= n_q && !pf_valid; };   // every piece of the member is in the window (what lies behind it is never consumed by a valid stream)
	auto ready = [&](uint32_t bits) -> bool { return (int)(wr * 32u - abit) >= (int)bits || exhausted(); };   // `bits` stream bits from the cursor on are staged (a window may also read a stale word behind them: those bits are never used)
	// (Reading the next trip's window words at the end of a trip - taking their LDS latency out of the dependent chain - was measured: no gain,
	// phase 1 alone 94.1 vs 91.1 ms per 96 M reads. A lone wave is bound by the issue latency of its dependent VALU chain, not by the LDS round trips.)
	auto window = [&](uint32_t ab) -> uint32_t { const uint32_t* p = L.slot(ab >> 5); return wv::alignbit(p[64], p[0], ab & 31u); };   // the 32 stream bits at bit position ab
	auto seek = [&](uint32_t target) {   // synchronous restart of the reader at bit position `target`
		next_q = target >> 7; wr = next_q * 4; pf_valid = false;
		wv::u32x4 c0 = next_q < n_q ? comp_q[q0 + next_q] : wv::make4(0, 0, 0, 0); ++next_q;
		wv::u32x4 c1 = next_q < n_q ? comp_q[q0 + next_q] : wv::make4(0, 0, 0, 0); ++next_q;
		L.stage(wr, c0.x, c0.y, c0.z, c0.w); L.stage(wr + 4, c1.x, c1.y, c1.z, c1.w); wr += 8;
		abit = target;
	};

	bool slow_mode = false;
	for (uint32_t trip = 0;; ++trip)
	{
		g3 = g2; g2 = g1; g1 = g0; g0 = tk; tk = K1_TOK_NOOP;
		// ================= service block: commit prefetched input, store the token group, issue the next prefetch =================
		if ((trip & (P1_SERVICE - 1)) == 0)
		{
			if (wv::ballot(state != S_DONE) == 0) break;
			// a piece is always in flight; it enters the window as soon as four slots in front of the cursor's word are free
			if (pf_valid && (int)(wr - (abit >> 5)) <= P1_RING_W - 4) { L.stage(wr, pf.x, pf.y, pf.z, pf.w); wr += 4; pf_valid = false; }
			if (state != S_DONE && state != S_NEXT)
			{
				if ((g0 & g1 & g2 & g3) != K1_TOK_NOOP)   // g3..g0 are exactly the four trips since the last service block
				{
					if (tok_n + 4 > tok_cap) { err = K1_ERR_TOKEN_OVERFLOW; state = S_FINISH; }
					else { *(wv::u32x4*)(tok_ptr + tok_n) = wv::make4(g3, g2, g1, g0); tok_n += 4; }
				}
				if (!pf_valid && next_q < n_q) { pf = comp_q[q0 + next_q]; ++next_q; pf_valid = true; }
			}
		}

		// Symbol decode (the common state) and header / bookkeeping states never run in the same trip: a lane that reaches
		// a header parks until park_hi lanes are parked (or no lane decodes symbols), then the wave runs ONLY the slow states
		// until every parked lane is back in S_SYM. Otherwise nearly every trip would pay for both code paths.
		// park_hi == 0: no parking (both per trip).
		{
			const uint64_t slow_m = wv::ballot(state != S_SYM && state != S_STORED && state != S_DONE);
			if (!slow_mode)
			{
				if (slow_m != 0 && ((int)wv::popc64(slow_m) >= park_hi || wv::ballot(state == S_SYM || state == S_STORED) == 0)) slow_mode = true;
			}
			else if (slow_m == 0) slow_mode = false;
		}
		const bool run_fast = !slow_mode || park_hi == 0, run_slow = slow_mode || park_hi == 0;

		if (run_fast && state == S_SYM && ready(48))   // a trip consumes at most 20 + 28 bits
		{
			// few exec regions: everything is computed unconditionally (indices clamped); the common path only ORs one error flag
			const uint32_t win = window(abit);
			uint32_t len;
			const int idx = limL.decode(win, len);
			bool bad = (uint32_t)idx >= 288u;
			const uint32_t s = L.litsym((uint32_t)idx < 288u ? (uint32_t)idx : 287u);
			uint32_t used = len, tokv = s, add = 1, ls = 0, ds = 0, mdist = 0; int di = 0;
			if (s > 256)
			{
				ls = s - 257;
				const uint32_t lt = tab[ls < 29 ? ls : 28];
				const uint32_t eb = lt >> 16, mlen = (lt & 0xffffu) + wv::bfe(win, len, eb);   // len <= 16, eb <= 5
				const uint32_t win2 = window(abit + len + eb);
				uint32_t dl;
				di = limD.decode(win2, dl);
				ds = dsym.get((uint32_t)di < 30u ? (uint32_t)di : 29u);
				const uint32_t dt = tab[32 + (ds < 30 ? ds : 29u)];
				const uint32_t deb = dt >> 16; mdist = (dt & 0xffffu) + wv::bfe(win2, dl, deb);   // dl <= 16, deb <= 13
				used = len + eb + dl + deb;
				bad = bad || ls >= 29 || (uint32_t)di >= 30u || ds >= 30 || mdist > out_n;
				tokv = (mlen << 23) + mdist + (0x80000000u - (3u << 23) - 1u); add = mlen;   // = bit 31 | (mlen - 3) << 23 | (mdist - 1): 3 <= mlen <= 258, 1 <= mdist <= 32768 on the good path
			}
			abit += used;
			bad = bad || (s != 256 && out_n + add > usize);
			if (bad)
			{
				err = (uint32_t)idx >= 288u ? 9u : s <= 256 ? 3u : ls >= 29 ? 10u : (uint32_t)di >= 30u ? 11u : ds >= 30 ? 12u : 13u;
				state = S_FINISH;
			}
			else if (s == 256) state = bfinal ? S_FINISH : S_HDR;
			else { tk = tokv; out_n += add; }
		}
		else if (run_fast && state == S_STORED)
		{
			if (ready(8))
			{
				if (out_n >= usize) { err = 3; state = S_FINISH; }
				else
				{
					tk = window(abit) & 255u; abit += 8; ++out_n;
					if (--stored_left == 0) state = bfinal ? S_FINISH : S_HDR;
				}
			}
		}
		if (!run_slow) continue;
		if (state == S_P1 || state == S_P2)
		{
			// one code-length-alphabet symbol per trip (RFC 1951 §3.2.7); pass 1 counts, pass 2 places symbols
			if (ready(14))
			{
				const uint32_t win = window(abit);
				// canonical decode of the 19-symbol alphabet (lengths 1..7) from registers
				int sym = -1; uint32_t len = 0;
				{
					const uint32_t v = wv::brev(win) >> 25;   // next 7 bits, MSB-first
					uint32_t n = 0, sel = 0;
					#pragma unroll
					for (int l = 6; l >= 0; --l) { const bool ge = v >= (limC[l] & 0xffffu); n += ge ? 1u : 0u; sel = ge ? sel : limC[l]; }
					if (n < 7)
					{
						const uint32_t k = (v >> (6 - n)) + (uint32_t)(int)(int16_t)(sel >> 16);
						if (k < 19) { sym = (int)(k < 12 ? (ccl_lo >> (5 * k)) & 31u : (ccl_hi >> (5 * (k - 12))) & 31u); len = n + 1; }
					}
				}
				if (sym < 0) { err = 6; state = S_FINISH; }
				else
				{
					const uint32_t x = win >> len;   // the repeat count's extra bits (len <= 7, at most 7 more)
					uint32_t rep = 1, val = (uint32_t)sym, used = len;
					if (sym == 16) { if (h_i == 0) { err = 7; state = S_FINISH; } rep = 3 + (x & 3u); used += 2; val = h_prev; }
					else if (sym == 17) { rep = 3 + (x & 7u); used += 3; val = 0; }
					else if (sym == 18) { rep = 11 + (x & 127u); used += 7; val = 0; }
					abit += used;
					if (h_i + rep > h_n) { err = 8; state = S_FINISH; }
					else if (err == 0)
					{
						if (val != 0)
						{
							if (state == S_P1)
							{
								const uint32_t n_lit = h_i >= h_nlit ? 0u : (h_nlit - h_i < rep ? h_nlit - h_i : rep);
								cl.add(val - 1, n_lit); cd.add(val - 1, rep - n_lit);
							}
							else for (uint32_t k = 0; k < rep; ++k)
							{
								uint32_t i = h_i + k;
								if (i < h_nlit) { uint32_t o = ol.get(val - 1); ol.add(val - 1, 1); L.set_litsym(o, i); }
								else { uint32_t o = od.get(val - 1); od.add(val - 1, 1); dsym.set(o, i - h_nlit); }
							}
						}
						h_i += rep; if (sym < 16) h_prev = (uint32_t)sym; else if (sym != 16) h_prev = 0;
						if (h_i == h_n)
						{
							if (state == S_P1)
							{
								// start offsets of every code length in the sorted symbol arrays, then re-read the header for pass 2
								uint32_t o = 0;
								#pragma unroll
								for (int l = 0; l < 15; ++l) { ol.set(l, o); o += cl.get_const(l); }
								if (o > 288) { err = 5; state = S_FINISH; }
								o = 0;
								#pragma unroll
								for (int l = 0; l < 15; ++l) { od.set(l, o); o += cd.get_const(l); }
								if (o > 32) { err = 5; state = S_FINISH; }
								if (err == 0) { limL.build(cl); limD.build(cd); seek(hdr_abit); h_i = 0; h_prev = 0; state = S_P2; }
							}
							else state = S_SYM;
						}
					}
				}
			}
		}
		else if (state == S_HDR)
		{
			if (ready(80))   // enough input staged for the fixed part of the header (<= 74 bits) or a stored-block header
			{
				if (abit > abit_end) { err = 15; state = S_FINISH; }   // the stream ran past its payload
				else
				{
					uint32_t win = window(abit);
					bfinal = (int)(win & 1u); const uint32_t btype = (win >> 1) & 3u; abit += 3;
					cl.clear(); cd.clear();
					if (btype == 0)
					{
						abit = (abit + 7u) & ~7u;   // piece 0 starts on a byte boundary of the stream, so this is the stream's byte alignment
						win = window(abit); abit += 32;
						const uint32_t lo = win & 0xffffu, hi = win >> 16;
						if ((lo ^ hi) != 0xffffu) { err = 2; state = S_FINISH; }
						else { stored_left = lo; state = lo ? S_STORED : (bfinal ? S_FINISH : S_HDR); }
					}
					else if (btype == 1)
					{
						// fixed Huffman code: lengths 7 (256..279), 8 (0..143, 280..287), 9 (144..255); 30 distance codes of length 5
						cl.set(6, 24); cl.set(7, 152); cl.set(8, 112); cd.set(4, 30);
						uint32_t k = 0;
						for (uint32_t s = 256; s < 280; ++s) L.set_litsym(k++, s);
						for (uint32_t s = 0; s < 144; ++s) L.set_litsym(k++, s);
						for (uint32_t s = 280; s < 288; ++s) L.set_litsym(k++, s);
						for (uint32_t s = 144; s < 256; ++s) L.set_litsym(k++, s);
						for (uint32_t s = 0; s < 30; ++s) dsym.set(s, s);
						limL.build(cl); limD.build(cd);
						state = S_SYM;
					}
					else if (btype == 2)
					{
						h_nlit = ((win >> 3) & 31u) + 257; const uint32_t ndist = ((win >> 8) & 31u) + 1, ncl = ((win >> 13) & 15u) + 4; abit += 14;
						h_n = h_nlit + ndist; h_i = 0; h_prev = 0;
						if (h_nlit > 286 || ndist > 30) { err = 5; state = S_FINISH; }
						else
						{
							// 19 code-length code lengths (3 bits each, permuted order)
							const uint64_t ORD_LO = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
							const uint64_t ORD_HI = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
							uint64_t cll = 0;   // 3 bits per symbol index
							for (uint32_t i = 0; i < ncl; ++i)
							{
								const uint32_t v = window(abit) & 7u; abit += 3;
								const uint32_t s = (uint32_t)((i < 12 ? ORD_LO >> (5 * i) : ORD_HI >> (5 * (i - 12))) & 31u);
								cll |= (uint64_t)v << (3 * s);
							}
							ccl_lo = 0; ccl_hi = 0; uint32_t k = 0, ccode = 0;
							#pragma unroll
							for (int l = 1; l <= 7; ++l)
							{
								uint32_t cnt = 0; const uint32_t o = k;
								for (uint32_t s = 0; s < 19; ++s)
									if (((cll >> (3 * s)) & 7u) == (uint32_t)l)
									{
										if (k < 12) ccl_lo |= (uint64_t)s << (5 * k); else ccl_hi |= (uint64_t)s << (5 * (k - 12));
										++k; ++cnt;
									}
								uint32_t lim = (ccode + cnt) << (7 - l); if (lim > 0x80u) lim = 0x80u;
								limC[l - 1] = lim | (((o - ccode) & 0xffffu) << 16);
								ccode = (ccode + cnt) << 1;
							}
							hdr_abit = abit;
							state = S_P1;
						}
					}
					else { err = 4; state = S_FINISH; }
				}
			}
		}
		else if (state == S_FINISH)
		{
			// publish once every token of the member has been stored: the group registers hold the last four trips (tk: this trip, when
			// a stored block's last byte and the finish fall into one trip), and a finishing lane emits nothing, so they run empty
			// within one service period
			if ((g0 & g1 & g2 & g3 & tk) == K1_TOK_NOOP)
			{
				if (!err && out_n != usize) err = 14;
				if (!err && abit > abit_end) err = 15;   // consumed bits behind the payload: a truncated stream
				tok_count[b] = tok_n; status[b].produced = out_n; status[b].error = err;
				state = S_NEXT;
			}
		}
		else if (state == S_NEXT)
		{
			// members leave the queue in the caller's order (largest compressed size first): the 64 lanes of a wave decode members of
			// nearly equal size and finish together, and the launch ends with its smallest members
			b = (int64_t)wv::atomic_inc(work_counter);
			if (b >= n_blocks) state = S_DONE;
			else
			{
				if (order) b = (int64_t)order[b];
				const BlockDesc bd = blocks[b];
				const uint64_t to = tok_off[b], to1 = tok_off[b + 1];
				q0 = bd.cpos >> 4; const uint32_t mis16 = (uint32_t)(bd.cpos & 15); usize = bd.usize;
				n_q = (mis16 + bd.clen + 15) / 16 + 1;
				abit_end = (mis16 + bd.clen) * 8;
				tok_ptr = tok + to; tok_cap = (uint32_t)(to1 - to); tok_n = 0;
				out_n = 0; err = 0; bfinal = 0;
				seek(mis16 * 8);
				state = S_HDR;
			}
		}
		// S_DONE: idle until every lane of the wave is done (checked in the service block)
	}
}

// ---------------------------------------------------------------------------------------------------------------- phase 2
constexpr int P2_BMAX = 1056;                  // output bytes resolved per batch: at least one whole group (4 x 258 bytes) always fits
constexpr int P2_NCH = (P2_BMAX + 63) / 64;    // 64-byte chunks per batch
constexpr int P2_GROUPS = 56;                  // groups per batch: 224 tokens are ~1000 bytes of BAM, so the byte limit binds first anyway - and pk + val stay under 2 KB
struct P2Lds { uint32_t pk[P2_GROUPS * 4]; alignas(8) uint8_t val[P2_NCH * 64]; };   // 1984 B per wave (eleven waves fit beside the six decoder waves of a CU); val: token-end flags during pass 1, the staged output bytes from pass 2 on

K1_DEV uint32_t tok_len(uint32_t t) { return t == K1_TOK_NOOP ? 0u : ((t >> 31) ? ((t >> 23) & 255u) + 3u : 1u); }

// Register budget of five waves per SIMD (70 VGPRs instead of 98, no spills): next to the decoder waves (163 VGPRs each, one or two per SIMD) the
// register file, not LDS, decides how many phase-2 waves a CU holds - 12 instead of 8.
K1_KERNEL_OCC(64, 5) void lz77_groups_kernel(const uint32_t* __restrict__ tok, const uint64_t* __restrict__ tok_off, const uint32_t* __restrict__ tok_count,
                                      const BlockDesc* __restrict__ blocks, int64_t n_blocks, uint8_t* __restrict__ out_base, BlockStatus* __restrict__ status)
{
	K1_SHARED P2Lds S;
	const int lane = wv::lane();
	const wv::u32x4 noop4 = wv::make4(K1_TOK_NOOP, K1_TOK_NOOP, K1_TOK_NOOP, K1_TOK_NOOP);
	for (int64_t b = wv::block_id(); b < n_blocks; b += wv::grid_size())
	{
		if (status[b].error) continue;
		const uint32_t ngroups = tok_count[b] >> 2;
		const wv::u32x4* T4 = (const wv::u32x4*)(tok + tok_off[b]);
		const uint32_t usize = blocks[b].usize;
		const wv::ByteBuf out = wv::ByteBuf::make(out_base + blocks[b].upos, usize);
		uint32_t P = 0, fail = 0;   // bytes written so far
		wv::u32x4 nxt = (lane < P2_GROUPS && (uint32_t)lane < ngroups) ? T4[lane] : noop4;
		for (uint32_t g0 = 0; g0 < ngroups;)
		{
			// ---- place the batch: one group per lane, a prefix sum over (bytes | real tokens << 20) ----
			const bool valid = lane < P2_GROUPS && g0 + (uint32_t)lane < ngroups;
			const uint32_t t0 = nxt.x, t1 = nxt.y, t2 = nxt.z, t3 = nxt.w;
			const uint32_t l0 = valid ? tok_len(t0) : 0u, l1 = valid ? tok_len(t1) : 0u, l2 = valid ? tok_len(t2) : 0u, l3 = valid ? tok_len(t3) : 0u;
			const uint32_t s = l0 + l1 + l2 + l3, c = (l0 ? 1u : 0u) + (l1 ? 1u : 0u) + (l2 ? 1u : 0u) + (l3 ? 1u : 0u);
			const uint32_t E = wv::scan_incl(s | (c << 20));
			const uint32_t Eb = E & 0xfffffu, Ec = E >> 20;
			// the longest prefix of groups whose output fits the staging buffer (the sums are non-decreasing: the ballot is a prefix mask)
			const uint32_t ng = wv::popc64(wv::ballot(valid && Eb <= (uint32_t)P2_BMAX));
			if (ng == 0) { fail = 18; break; }   // a group longer than 4 x 258 bytes: not a token stream of phase 1
			const uint32_t B = wv::readlane(Eb, (int)ng - 1);
			if (P + B > usize) { fail = 16; break; }
			// per real token: match flag | start inside the batch << 20 | dist-1 or the literal; a flag on the token's last byte
			{
				unsigned long long* z = (unsigned long long*)S.val;
				z[lane] = 0ull; z[64 + lane] = 0ull; if (lane < P2_NCH * 8 - 128) z[128 + lane] = 0ull;
			}
			wv::barrier();
			if ((uint32_t)lane < ng)
			{
				uint32_t st = Eb - s, rk = Ec - c;
				const uint32_t tt[4] = {t0, t1, t2, t3}, ll[4] = {l0, l1, l2, l3};
				#pragma unroll
				for (int k = 0; k < 4; ++k)
					if (ll[k])
					{
						const uint32_t t = tt[k];
						S.pk[rk] = (t & 0x80000000u) | (st << 20) | ((t >> 31) ? (t & 0x7fffu) : (t & 255u));
						st += ll[k]; ++rk;
						S.val[st - 1] = 1;
					}
			}
			wv::barrier();
			// stores of earlier batches must be complete before this batch gathers from the window behind P
			wv::wait_vm0();
			// the next batch's groups are requested now; they arrive while this batch is resolved
			{ const uint32_t i2 = g0 + ng + (uint32_t)lane; nxt = (lane < P2_GROUPS && i2 < ngroups) ? T4[i2] : noop4; }

			// ---- pass 1: classify every byte of the batch, issue all gathers that reach behind the batch ----
			// inf: bits 0..7 value, bits 8..9 kind (0 value known, 1 gathered from HBM, 2 staged byte of an earlier chunk, 3 a lower lane of the same chunk), bits 10.. source
			uint32_t inf[P2_NCH], gth[P2_NCH];
			uint32_t ta = 0;   // tokens that end before the current chunk
			#pragma unroll
			for (int ch = 0; ch < P2_NCH; ++ch)
			{
				inf[ch] = 0; gth[ch] = 0;
				if ((uint32_t)(ch * 64) < B)
				{
					const uint32_t j0 = (uint32_t)(ch * 64), j = j0 + (uint32_t)lane;
					// owner token of byte j = ta + #tokens ending inside the chunk before j: the end flags of the chunk as a lane mask
					const uint64_t m = wv::ballot(S.val[j] != 0);
					const uint32_t o = ta + wv::mbcnt(m);
					ta += wv::popc64(m);
					const uint32_t pko = S.pk[o < (uint32_t)(P2_GROUPS * 4) ? o : 0u];   // (only lanes behind the batch's last byte can run past the table)
					uint32_t f = pko & 255u;
					if (j < B && (pko >> 31))
					{
						const uint32_t sto = (pko >> 20) & 0x7ffu, d = (pko & 0x7fffu) + 1u;
						int src = (int)j - (int)d;   // relative to P
						if (src >= (int)sto)         // the match reaches into its own output (distance < length): periodic form sto - d + (j - sto) mod d
						{
							const uint32_t off = j - sto;
							uint32_t q = (uint32_t)((float)off * wv::rcp((float)d)); int rr = (int)off - (int)(q * d);   // q is off by at most 1
							if (rr < 0) rr += (int)d; else if (rr >= (int)d) rr -= (int)d;
							src = (int)sto - (int)d + rr;
						}
						if (src < 0) { gth[ch] = out.load(P + (uint32_t)src); f = 0x100u; }
						else f = ((uint32_t)src < j0 ? 0x200u : 0x300u) | ((uint32_t)src << 10);   // (same chunk: the source lane is src & 63)
					}
					inf[ch] = f;
				}
			}
			// every gather has landed (one wait for the whole batch: pass 2 below issues stores only, and must not wait for them chunk by chunk)
			wv::wait_vm0();
			// ---- pass 2: resolve front to back; every byte is written once to LDS and once to HBM (64 consecutive bytes per store instruction) ----
			#pragma unroll
			for (int ch = 0; ch < P2_NCH; ++ch)
			{
				if ((uint32_t)(ch * 64) < B)
				{
					const uint32_t j = (uint32_t)(ch * 64) + (uint32_t)lane, f = inf[ch], kind = (f >> 8) & 3u;
					uint32_t vv = f & 255u, rel = (uint32_t)lane;
					if (kind == 1) vv = gth[ch] & 255u;
					else if (kind == 2) vv = S.val[(f >> 10) & 0x7ffu];
					else if (kind == 3) { vv = 0x100u; rel = (f >> 10) & 63u; }   // bit 8 = still waiting for a lower lane of this chunk
					// sources are always lower lanes, so the loop terminates; the periodic form makes its depth the number of chained TOKENS, not bytes
					uint64_t pend = wv::ballot((vv & 0x100u) != 0);
					while (pend)
					{
						const uint32_t sv = wv::shfl(vv, (int)rel);
						if ((vv & 0x100u) && !(sv & 0x100u)) vv = sv;
						pend = wv::ballot((vv & 0x100u) != 0);
					}
					if (j < B) { S.val[j] = (uint8_t)vv; out.store(P + j, vv); }
					wv::barrier();
				}
			}
			P += B; g0 += ng;
		}
		if (lane == 0)
		{
			if (fail) status[b].error = fail;
			else if (P != usize) status[b].error = 17;
			status[b].produced = P;
		}
	}
}

} } // namespace ngsqc::k1
