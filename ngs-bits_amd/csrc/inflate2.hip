// K1 (two-phase form) — BGZF inflate split into the part that is serial per member and the part that is not.
//
//   phase 1  huff_tokens_kernel : ONE LANE PER BGZF MEMBER. Every lane Huffman-decodes its own raw-DEFLATE stream into a
//            token stream (literal byte | match{len,dist}); 64 members advance per wave instruction, so the VALU issue
//            slots that the group kernel (inflate.hip) spends on 16 redundant lanes carry 64 independent decoders.
//            Canonical Huffman decode runs out of REGISTERS (per code length one limit|delta word, the length found by a
//            4-level binary search with v_cndmask-selected pivots); LDS only holds the symbol-order planes and a 32-byte
//            input ring per lane, laid out element-major (word k of lane l at k*64+l) so that any per-lane access
//            pattern is bank-conflict free; waiting tokens sit in a 7-register shift register.
//            Input refill and token flush happen in a "service" block every 4 symbols: 16-byte loads prefetched one
//            service ahead, 16-byte token stores, one s_waitcnt per service. Header states are parked (see the loop).
//   phase 2  lz77_chunk_kernel : ONE WAVE PER MEMBER. Tokens are taken 64 at a time; a wave prefix-sum gives every
//            token its output range, then the batch is resolved front to back in 64-byte chunks: every OUTPUT BYTE gets a
//            lane (owner token by popcount over a token-end bitmap), its source in periodic form (i mod dist) is
//            gathered from HBM (before the batch), read from the LDS staging bytes (earlier chunk) or taken from the
//            source lane (same chunk).
//   The host side (api.hip:inflate_members) cuts the members into chunks of one decoder "round" and overlaps phase 2 of a
//   chunk with phase 1 of the next one on a second stream.
//
// Same output as inflate.hip (bit-exact; tests/test_gpu_parity.py, tests/test_gpu_inflate.py). Integer work, no MFMA.
#include "common.h"
#include <cstdlib>
#include <algorithm>

namespace ngsqc {

// ---------------------------------------------------------------------------------------------------------------- phase 1
constexpr int P1_SYM_W = 81;     // lit_sym : 288 x 9 bit as a byte plane (72 words) + a bit plane (9 words)
constexpr int P1_RING_W = 8;     // compressed input ring (32 B)
constexpr int P1_LANE_W = P1_SYM_W + P1_RING_W;   // 89 words per lane (22.8 KB per wave); tokens and the distance symbols wait in registers
constexpr int P1_PAD_W = 192;    // + 768 B: 23 KB per one-wave workgroup (see the kernel)
constexpr int P1_SERVICE = 4;    // symbols between service blocks

enum { S_NEXT = 0, S_HDR = 1, S_P1 = 2, S_P2 = 3, S_SYM = 4, S_STORED = 5, S_FINISH = 6, S_DONE = 7 };
enum { TOK_ERR_OVERFLOW = K1_ERR_TOKEN_OVERFLOW };

struct P1Lds
{
	uint32_t* base; int lane;
	__device__ __forceinline__ uint32_t& at(int k) const { return base[k * 64 + lane]; }
	__device__ __forceinline__ uint32_t litsym(uint32_t i) const
	{
		uint32_t lo = at((int)(i >> 2)), hi = at(72 + (int)(i >> 5));
		return ((lo >> (8 * (i & 3))) & 255u) | (((hi >> (i & 31)) & 1u) << 8);
	}
	__device__ __forceinline__ void set_litsym(uint32_t i, uint32_t s) const
	{
		uint32_t& lo = at((int)(i >> 2)); uint32_t sh = 8 * (i & 3); lo = (lo & ~(255u << sh)) | ((s & 255u) << sh);
		uint32_t& hi = at(72 + (int)(i >> 5)); hi = (hi & ~(1u << (i & 31))) | ((s >> 8) << (i & 31));
	}
	__device__ __forceinline__ uint32_t& ring(uint32_t i) const { return at(P1_SYM_W + (int)(i & (P1_RING_W - 1))); }
};

// packed per-length counters: FW bits per field, 32/FW fields per register (FW = 10 for lit/len, 5+1 for dist/CL -> use 6)
template <int FW, int NREG> struct Packed
{
	uint32_t r[NREG];
	__device__ __forceinline__ void clear() { for (int i = 0; i < NREG; ++i) r[i] = 0; }
	__device__ __forceinline__ uint32_t get_const(int idx) const { constexpr int PER = 32 / FW; return (r[idx / PER] >> (FW * (idx % PER))) & ((1u << FW) - 1u); }   // idx compile-time after unrolling
	__device__ __forceinline__ uint32_t get(uint32_t idx) const
	{
		// mask-select (not an indexed read: keeps the counters in VGPRs instead of scratch memory)
		constexpr int PER = 32 / FW; uint32_t reg = idx / PER, sh = FW * (idx % PER); uint32_t v = 0;
		#pragma unroll
		for (int i = 0; i < NREG; ++i) v |= r[i] & (0u - (uint32_t)(reg == (uint32_t)i));
		return (v >> sh) & ((1u << FW) - 1u);
	}
	__device__ __forceinline__ void add(uint32_t idx, uint32_t delta)
	{
		constexpr int PER = 32 / FW; uint32_t reg = idx / PER, inc = delta << (FW * (idx % PER));
		#pragma unroll
		for (int i = 0; i < NREG; ++i) r[i] += reg == (uint32_t)i ? inc : 0u;
	}
	__device__ __forceinline__ void set(uint32_t idx, uint32_t v)
	{
		constexpr int PER = 32 / FW; uint32_t reg = idx / PER, sh = FW * (idx % PER), m = ((1u << FW) - 1u) << sh;
		#pragma unroll
		for (int i = 0; i < NREG; ++i) r[i] = reg == (uint32_t)i ? ((r[i] & ~m) | (v << sh)) : r[i];
	}
};
using LitCnt = Packed<10, 5>;   // indices 0..14 <-> code lengths 1..15, values <= 288
using DistCnt = Packed<6, 3>;   // values <= 32

// canonical (puff-style) decode: bits LSB-first; returns index into the (len,sym)-sorted symbol array, or -1
template <class CNT>
__device__ __forceinline__ int canon_decode(uint32_t bits, const CNT& c, uint32_t& len_out)
{
	int code = 0, first = 0, index = 0;
	#pragma unroll
	for (int len = 1; len <= 15; ++len)
	{
		code |= (int)(bits & 1u); bits >>= 1;
		int count = (int)c.get_const(len - 1);
		if (code - count < first) { len_out = (uint32_t)len; return index + (code - first); }
		index += count; first += count; first <<= 1; code <<= 1;
	}
	len_out = 15; return -1;
}

// Branch-free canonical decode out of REGISTERS (occupancy is LDS-bound at < 1 wave per SIMD, so VGPRs are free).
// For code length l (1..15) the word holds
//   limit_l = (first_code_l + count_l) << (15 - l)   (upper bound, left-aligned to 15 bits; non-decreasing in l)
//   delta_l = offset_l - first_code_l                (index of the length's first symbol in the sorted array minus its first code)
// With v = the next 15 stream bits MSB-first: length = 1 + #{l : v >= limit_l}, index = (v >> (15 - length)) + delta_length.
// Distance symbols sorted by (len, sym), 30 x 5 bit in three 64-bit registers (mask-select, no indexed access). Keeping them in LDS
// instead (5 more words per lane) was measured: phase 1 did not get faster, and the 512 B it costs per wave push the CU from four to
// three resident phase-2 workgroups (lz77 53 -> 71 ms per 48 M reads).
struct DistSyms
{
	uint64_t q[3];
	__device__ __forceinline__ void clear() { q[0] = q[1] = q[2] = 0; }
	__device__ __forceinline__ uint32_t get(uint32_t i) const
	{
		uint32_t reg = (i * 43u) >> 9, sh = 5u * (i - reg * 12u);   // i / 12 for i < 36
		uint64_t v = (q[0] & (0ull - (uint64_t)(reg == 0))) | (q[1] & (0ull - (uint64_t)(reg == 1))) | (q[2] & (0ull - (uint64_t)(reg == 2)));
		return (uint32_t)(v >> sh) & 31u;
	}
	__device__ __forceinline__ void set(uint32_t i, uint32_t s)
	{
		uint32_t reg = (i * 43u) >> 9, sh = 5u * (i - reg * 12u); uint64_t m = 31ull << sh, val = (uint64_t)s << sh;
		#pragma unroll
		for (int k = 0; k < 3; ++k) q[k] = reg == (uint32_t)k ? ((q[k] & ~m) | val) : q[k];
	}
};

struct LimTab
{
	uint32_t w[15];   // (limit_l << 16) | (delta_l & 0xffff) for l = 1..15
	// With vx = (v << 16) | 0xffff a plain 32-bit compare vx >= w[l] is v >= limit_l. The limits are non-decreasing, so
	// n = #{l : v >= limit_l} is found by a 4-level binary search whose pivots are picked with v_cndmask from the 15
	// registers (4 compares + 11 selects instead of 15 compares + 30 selects); the last pivot the search went LEFT of
	// is w[n], the word of the code's own length, which carries the delta.
	__device__ __forceinline__ int decode(uint32_t bits, uint32_t& len_out) const
	{
		const uint32_t vx = ((__brev(bits) >> 1) & 0x7fff0000u) | 0xffffu;
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
	template <class CNT> __device__ __forceinline__ void build(const CNT& c)
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

template <int PAD_W>
__global__ __launch_bounds__(64) void huff_tokens_kernel(const uint8_t* __restrict__ comp, const BlockDesc* __restrict__ blocks, int64_t n_blocks,
                                                          const uint64_t* __restrict__ tok_off, uint32_t* __restrict__ tok, uint32_t* __restrict__ tok_count,
                                                          BlockStatus* __restrict__ status, unsigned long long* __restrict__ work_counter, const uint32_t* __restrict__ order, int park_hi)
{
	// 23 KB per one-wave workgroup: exactly six fit a CU's 160 KB and a seventh does not, so the workgroups of the NEXT chunk's
	// launch (queued on a second stream) take over the slots of this launch's finished waves without ever squeezing the LDS that
	// the phase-2 / CRC / scan workgroups need beside them
	__shared__ uint32_t lds[P1_LANE_W * 64 + PAD_W];
	const int lane = threadIdx.x;
	P1Lds L{lds, lane};
	// Phase 1 is the long pole of K1 (alone 16.8 ms per round of members, 25 ms next to phase 2 / CRC / the previous tile's consumers) and its
	// waves run a dependent instruction stream: every issue slot lost to a co-resident wave delays them. park_hi's upper bits carry a wave
	// priority: the other kernels' waves then only get the slots phase 1 leaves (s_setprio, VALU arbitration is priority first, then age).
	{
		const int prio = park_hi >> 8;
		if (prio == 1) __builtin_amdgcn_s_setprio(1); else if (prio == 2) __builtin_amdgcn_s_setprio(2); else if (prio >= 3) __builtin_amdgcn_s_setprio(3);
		park_hi &= 255;
	}
	const uint4* const comp_q = (const uint4*)comp;
	constexpr int WAIT_VM0 = 0x0F70;

	// ---- per-lane decoder state ----
	int state = S_NEXT;
	int64_t b = -1;                                        // members are handed out one at a time from a global counter (even finish times)
	uint64_t q0 = 0; uint32_t n_q = 0, next_q = 0;         // 16-byte chunks of this member: comp_q[q0 + i], i < n_q
	uint32_t mis16 = 0, clen = 0, usize = 0;
	uint64_t bitbuf = 0; uint32_t bitcnt = 0;              // LSB-first bit buffer
	uint32_t rd = 0, wr = 0;                               // ring word counters
	uint32_t bits_used = 0;                                // payload bits consumed so far (for byte alignment / re-seek)
	uint4 pf = make_uint4(0, 0, 0, 0); bool pf_valid = false;
	uint32_t* tok_ptr = nullptr; uint32_t tok_cap = 0, tok_n = 0, tok_flushed = 0;
	uint32_t tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0, tq5 = 0, tq6 = 0;   // unflushed tokens, newest first (a shift register: at most 3 left over + 4 new between services)
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
	uint32_t h_i = 0, h_n = 0, h_nlit = 0, h_prev = 0, hdr_bits = 0; uint32_t stored_left = 0;

	auto refill = [&]() {   // top the bit buffer up from the ring (branch-free); past the end of the member zero bits are appended
		const bool t = bitcnt <= 32, a = rd != wr, ex = next_q >= n_q && !pf_valid;
		uint32_t w = L.ring(rd);
		w = (t && a) ? w : 0u;
		bitbuf |= (uint64_t)w << (bitcnt & 63u);
		bitcnt += (t && (a || ex)) ? 32u : 0u;
		rd += (t && a) ? 1u : 0u;
	};
	auto ready = [&](uint32_t words) -> bool { return wr - rd >= words || (next_q >= n_q && !pf_valid); };
	auto take = [&](uint32_t n) -> uint32_t { uint32_t v = (uint32_t)bitbuf & ((1u << n) - 1u); bitbuf >>= n; bitcnt -= n; bits_used += n; return v; };   // n <= 16
	auto seek = [&](uint32_t bitpos) {   // synchronous restart of the reader at payload bit position `bitpos`
		uint32_t abs_byte = mis16 + (bitpos >> 3);
		next_q = abs_byte >> 4; rd = wr = 0; pf_valid = false;
		uint4 c0 = next_q < n_q ? comp_q[q0 + next_q] : make_uint4(0, 0, 0, 0); ++next_q;
		uint4 c1 = next_q < n_q ? comp_q[q0 + next_q] : make_uint4(0, 0, 0, 0); ++next_q;
		__builtin_amdgcn_s_waitcnt(WAIT_VM0);
		L.ring(0) = c0.x; L.ring(1) = c0.y; L.ring(2) = c0.z; L.ring(3) = c0.w; L.ring(4) = c1.x; L.ring(5) = c1.y; L.ring(6) = c1.z; L.ring(7) = c1.w; wr = 8;
		rd = (abs_byte & 15u) >> 2;
		uint32_t sh = (abs_byte & 3u) * 8 + (bitpos & 7u);
		bitbuf = (uint64_t)L.ring(rd) >> sh; bitcnt = 32 - sh; ++rd; bits_used = bitpos;
		refill();
	};
	auto emit = [&](uint32_t t) { if (tok_n >= tok_cap) { err = TOK_ERR_OVERFLOW; state = S_FINISH; } else { tq6 = tq5; tq5 = tq4; tq4 = tq3; tq3 = tq2; tq2 = tq1; tq1 = tq0; tq0 = t; ++tok_n; } };

	int trip = 0; bool slow_mode = false;
	while (true)
	{
		// ================= service block: commit prefetched input, flush tokens, issue the next prefetch =================
		if ((trip & (P1_SERVICE - 1)) == 0)
		{
			if (__builtin_amdgcn_ballot_w64(state != S_DONE) == 0) break;
			__builtin_amdgcn_s_waitcnt(WAIT_VM0);
			if (pf_valid) { L.ring(wr) = pf.x; L.ring(wr + 1) = pf.y; L.ring(wr + 2) = pf.z; L.ring(wr + 3) = pf.w; wr += 4; pf_valid = false; }
			if (state != S_DONE && state != S_NEXT)
			{
				if (tok_n - tok_flushed >= 4)
				{
					// the four oldest of cnt = 4..7 waiting tokens are tq[cnt-1] .. tq[cnt-4]
					const uint32_t k = tok_n - tok_flushed - 4;
					uint4 t4;
					t4.x = k == 0 ? tq3 : (k == 1 ? tq4 : (k == 2 ? tq5 : tq6));
					t4.y = k == 0 ? tq2 : (k == 1 ? tq3 : (k == 2 ? tq4 : tq5));
					t4.z = k == 0 ? tq1 : (k == 1 ? tq2 : (k == 2 ? tq3 : tq4));
					t4.w = k == 0 ? tq0 : (k == 1 ? tq1 : (k == 2 ? tq2 : tq3));
					*(uint4*)(tok_ptr + tok_flushed) = t4; tok_flushed += 4;
				}
				if (wr - rd <= (uint32_t)(P1_RING_W - 4) && next_q < n_q) { pf = comp_q[q0 + next_q]; ++next_q; pf_valid = true; }
			}
		}
		++trip;

		// Symbol decode (the common state) and header / bookkeeping states never run in the same trip: a lane that reaches
		// a header parks until park_hi lanes are parked (or no lane decodes symbols), then the wave runs ONLY the slow states
		// until every parked lane is back in S_SYM. Otherwise nearly every trip would pay for both code paths (with 64 lanes
		// each ~4 % of its time in a header, some lane is in one ~90 % of the time). park_hi == 0: no parking (both per trip).
		{
			const uint64_t slow_m = __builtin_amdgcn_ballot_w64(state != S_SYM && state != S_STORED && state != S_DONE);
			if (!slow_mode)
			{
				if (slow_m != 0 && ((int)__popcll(slow_m) >= park_hi || __builtin_amdgcn_ballot_w64(state == S_SYM || state == S_STORED) == 0)) slow_mode = true;
			}
			else if (slow_m == 0) slow_mode = false;
		}
		const bool run_fast = !slow_mode || park_hi == 0, run_slow = slow_mode || park_hi == 0;

		if (run_fast && state == S_SYM && ready(2))   // both refills of this symbol are guaranteed (or the stream is exhausted: zeros follow)
		{
			// few exec regions: everything is computed unconditionally (indices clamped), errors are collected in e
			refill();
			uint32_t len;
			const int idx = limL.decode((uint32_t)bitbuf, len);
			uint32_t e = (uint32_t)idx < 288u ? 0u : 9u;
			const uint32_t s = L.litsym((uint32_t)idx < 288u ? (uint32_t)idx : 287u);
			bitbuf >>= len; bitcnt -= len; bits_used += len;
			uint32_t tokv = s, add = 1;
			if (s > 256)
			{
				const uint32_t ls = s - 257;
				if (ls >= 29) e = 10;
				const uint32_t eb = ls < 8 ? 0u : (ls == 28 ? 0u : (ls - 4) >> 2);
				const uint32_t base = ls < 8 ? ls + 3 : (ls == 28 ? 258u : ((4u + ((ls - 4) & 3u)) << eb) + 3u);
				const uint32_t mlen = base + take(eb);
				refill();
				uint32_t dl;
				const int di = limD.decode((uint32_t)bitbuf, dl);
				if ((uint32_t)di >= 30u) e = 11;
				const uint32_t ds = dsym.get((uint32_t)di < 30u ? (uint32_t)di : 29u);
				bitbuf >>= dl; bitcnt -= dl; bits_used += dl;
				if (ds >= 30) e = 12;
				const uint32_t deb = ds < 4 ? 0u : (ds >> 1) - 1u;
				const uint32_t dbase = ds < 4 ? ds + 1 : ((2u + (ds & 1u)) << deb) + 1u;
				const uint32_t mdist = dbase + take(deb);
				if (mdist > out_n) e = 13;
				tokv = 0x80000000u | (((mlen - 3) & 255u) << 23) | ((mdist - 1) & 0x7fffu); add = mlen;
			}
			if (e == 0 && s != 256 && out_n + add > usize) e = s < 256 ? 3u : 13u;
			if (e) { err = e; state = S_FINISH; }
			else if (s == 256) state = bfinal ? S_FINISH : S_HDR;
			else { emit(tokv); out_n += add; }
		}
		else if (run_fast && state == S_STORED)
		{
			if (ready(1))
			{
				refill();
				if (out_n >= usize) { err = 3; state = S_FINISH; }
				else { emit(take(8)); ++out_n; if (--stored_left == 0 && state == S_STORED) state = bfinal ? S_FINISH : S_HDR; }
			}
		}
		if (!run_slow) continue;
		if (state == S_P1 || state == S_P2)
		{
			// one code-length-alphabet symbol per trip (RFC 1951 §3.2.7); pass 1 counts, pass 2 places symbols
			if (ready(1))
			{
				refill();
				// canonical decode of the 19-symbol alphabet (lengths 1..7) from registers
				int sym = -1; uint32_t len = 0;
				{
					const uint32_t v = __brev((uint32_t)bitbuf) >> 25;   // next 7 bits, MSB-first
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
					bitbuf >>= len; bitcnt -= len; bits_used += len;
					uint32_t rep = 1, val = (uint32_t)sym;
					if (sym == 16) { if (h_i == 0) { err = 7; state = S_FINISH; } rep = 3 + take(2); val = h_prev; }
					else if (sym == 17) { rep = 3 + take(3); val = 0; }
					else if (sym == 18) { rep = 11 + take(7); val = 0; }
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
								if (err == 0) { limL.build(cl); limD.build(cd); seek(hdr_bits); h_i = 0; h_prev = 0; state = S_P2; }
							}
							else state = S_SYM;
						}
					}
				}
			}
		}
		else if (state == S_HDR)
		{
			if (ready(4))   // enough input staged for the fixed part of the header (<= 74 bits) or a stored-block header
			{
				refill();
				bfinal = (int)take(1); uint32_t btype = take(2);
				cl.clear(); cd.clear();
				if (btype == 0)
				{
					uint32_t pad = (0u - bits_used) & 7u; take(pad);
					refill();
					uint32_t lo = take(16); refill(); uint32_t hi = take(16);
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
					h_nlit = take(5) + 257; uint32_t ndist = take(5) + 1, ncl = take(4) + 4;
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
							refill();
							uint32_t v = take(3);
							uint32_t s = (uint32_t)((i < 12 ? ORD_LO >> (5 * i) : ORD_HI >> (5 * (i - 12))) & 31u);
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
						hdr_bits = bits_used;
						state = S_P1;
					}
				}
				else { err = 4; state = S_FINISH; }
			}
		}
		else if (state == S_FINISH)
		{
			// flush the tail of the token ring, publish counts
			if (err != TOK_ERR_OVERFLOW)
			{
				const uint32_t cnt = tok_n - tok_flushed;   // <= 7; tq_i is the token at stream position tok_n - 1 - i
				if (cnt > 0) tok_ptr[tok_n - 1] = tq0;
				if (cnt > 1) tok_ptr[tok_n - 2] = tq1;
				if (cnt > 2) tok_ptr[tok_n - 3] = tq2;
				if (cnt > 3) tok_ptr[tok_n - 4] = tq3;
				if (cnt > 4) tok_ptr[tok_n - 5] = tq4;
				if (cnt > 5) tok_ptr[tok_n - 6] = tq5;
				if (cnt > 6) tok_ptr[tok_n - 7] = tq6;
			}
			if (!err && out_n != usize) err = 14;
			tok_count[b] = tok_n; status[b].produced = out_n; status[b].error = err;
			state = S_NEXT;
		}
		else if (state == S_NEXT)
		{
			// members leave the queue in the caller's order (largest compressed size first): the 64 lanes of a wave decode members of
			// nearly equal size and finish together, and the launch ends with its smallest members
			b = (int64_t)atomicAdd(work_counter, 1ull);
			if (b >= n_blocks) state = S_DONE;
			else
			{
				if (order) b = (int64_t)order[b];
				const BlockDesc bd = blocks[b];
				const uint64_t to = tok_off[b], to1 = tok_off[b + 1];
				__builtin_amdgcn_s_waitcnt(WAIT_VM0);
				q0 = bd.cpos >> 4; mis16 = (uint32_t)(bd.cpos & 15); clen = bd.clen; usize = bd.usize;
				n_q = (mis16 + clen + 15) / 16 + 1;
				tok_ptr = tok + to; tok_cap = (uint32_t)(to1 - to); tok_n = 0; tok_flushed = 0;
				out_n = 0; err = 0; bfinal = 0;
				seek(0);
				state = S_HDR;
			}
		}
		// S_DONE: idle until every lane of the wave is done (checked in the service block)
	}
}

// ---------------------------------------------------------------------------------------------------------------- phase 2
constexpr int P2_BMAX = 1024;   // max output bytes resolved per batch (LDS staging)

// Phase 2. Tokens are taken in batches of at most 64 tokens / P2_BMAX output bytes; a batch is resolved front to back in
// 64-byte chunks so that no separate dependency passes are needed: a byte whose source lies
//   * before the batch            -> gathered from HBM (earlier batches of the same wave; stores drained by vmcnt(0)),
//   * in an earlier chunk         -> read from the LDS staging bytes (already final),
//   * in the same chunk           -> taken from the source LANE (ds_bpermute), iterating only while some lane's source is
//                                    itself still pending (sources are always lower lanes, so the loop terminates; the
//                                    periodic form i mod dist makes the depth the number of chained TOKENS, not bytes).
// Every byte is written once to LDS and once to HBM (64 consecutive bytes per store instruction).
struct P2bLds { unsigned long long endmask[P2_BMAX / 64]; alignas(8) uint8_t val[P2_BMAX + 64]; };

// Inclusive wave prefix sum on the DPP network (no LDS round trips): Hillis-Steele inside each 16-lane row
// (row_shr:1/2/4/8, out-of-row sources read as 0), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2-3.
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x)
{
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
	return x;
}

// Buffer resource over one member's output: 32-bit offsets (one VALU add per address instead of a 64-bit add chain) and
// hardware bounds clamping. The descriptor lives in SGPRs, so its inputs are made provably wave-uniform first.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t member_rsrc(uint8_t* p, uint32_t bytes)
{
	const uint64_t a = (uint64_t)p;
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
	return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

__global__ __launch_bounds__(256) void lz77_chunk_kernel(const uint32_t* __restrict__ tok, const uint64_t* __restrict__ tok_off, const uint32_t* __restrict__ tok_count,
                                                         const BlockDesc* __restrict__ blocks, int64_t n_blocks, uint8_t* __restrict__ out_base, BlockStatus* __restrict__ status)
{
	__shared__ P2bLds lds[4];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	P2bLds& S = lds[wv];
	constexpr int WAIT_VM0 = 0x0F70;
	const uint64_t lane_lt = (1ull << lane) - 1ull;
	const int64_t n_waves = (int64_t)gridDim.x * 4;
	for (int64_t b = (int64_t)blockIdx.x * 4 + wv; b < n_blocks; b += n_waves)
	{
		if (status[b].error) continue;
		const uint32_t n = tok_count[b];
		const uint32_t* T = tok + tok_off[b];
		uint8_t* out = out_base + blocks[b].upos;
		const uint32_t usize = blocks[b].usize;
		const __amdgpu_buffer_rsrc_t rs = member_rsrc(out, usize);
		uint32_t P = 0;   // bytes written so far
		uint32_t tk_next = (uint32_t)lane < n ? T[lane] : 0u;
		for (uint32_t t0 = 0; t0 < n;)
		{
			const uint32_t i = t0 + (uint32_t)lane;
			const uint32_t tk = tk_next;
			const bool is_m = tk >> 31;
			const uint32_t len = i < n ? (is_m ? ((tk >> 23) & 255u) + 3u : 1u) : 0u;
			const uint32_t end = wave_scan_incl(len);
			// take the longest token prefix whose output fits the staging buffer
			const uint64_t fit = __builtin_amdgcn_ballot_w64(i < n && end <= (uint32_t)P2_BMAX);
			uint32_t ntake = (uint32_t)__popcll(fit); if (ntake == 0) ntake = 1;
			const uint32_t B = (uint32_t)__builtin_amdgcn_readlane((int)end, (int)ntake - 1);
			const uint32_t start = end - len;
			if (P + B > usize) { if (lane == 0) status[b].error = 16; break; }
			// what a byte needs from its owner token: match flag, the token's start inside the batch, dist-1 or the literal
			const uint32_t pk = (tk & 0x80000000u) | ((start & 0x7ffu) << 20) | (is_m ? (tk & 0x7fffu) : (tk & 255u));
			// token-end bitmap of the batch: bit (e-1) set when a token ends at byte e (ends are strictly increasing)
			if (lane < P2_BMAX / 64) S.endmask[lane] = 0ull;
			__builtin_amdgcn_wave_barrier();
			if ((uint32_t)lane < ntake && len != 0) atomicOr(&S.endmask[(end - 1) >> 6], 1ull << ((end - 1) & 63u));
			__builtin_amdgcn_wave_barrier();
			// stores of earlier batches must be complete before this batch gathers from the window
			__builtin_amdgcn_s_waitcnt(WAIT_VM0);
			// the next batch's tokens are requested now; they arrive while this batch is resolved
			{ const uint32_t i2 = t0 + ntake + (uint32_t)lane; tk_next = i2 < n ? T[i2] : 0u; }
			uint32_t ta = 0;   // tokens that end at or before the current chunk start
			for (uint32_t j0 = 0; j0 < B; j0 += 64)
			{
				const uint32_t j = j0 + (uint32_t)lane;
				// owner token of byte j = ta + #tokens ending inside the chunk before j (popcount over the end bitmap)
				const uint64_t m = S.endmask[j0 >> 6];
				const uint32_t o = ta + (uint32_t)__popcll(m & lane_lt);
				ta += (uint32_t)__popcll(m);
				const uint32_t pko = (uint32_t)__shfl((int)pk, (int)(o & 63u));
				uint32_t vv = pko & 255u;        // bit 8 = still waiting for a lower lane of this chunk
				uint32_t rel = (uint32_t)lane;
				if (j < B && (pko >> 31))
				{
					const uint32_t sto = (pko >> 20) & 0x7ffu, d = (pko & 0x7fffu) + 1u, off = j - sto;
					uint32_t r = off;
					if (off >= d)
					{
						uint32_t q = (uint32_t)((float)off * __builtin_amdgcn_rcpf((float)d)); int rr = (int)off - (int)(q * d);   // q is off by at most 1
						if (rr < 0) rr += (int)d; else if (rr >= (int)d) rr -= (int)d;
						r = (uint32_t)rr;
					}
					const int src = (int)sto - (int)d + (int)r;   // relative to P
					if (src < 0) vv = __builtin_amdgcn_raw_buffer_load_b8(rs, (int)P + src, 0, 0);
					else if ((uint32_t)src < j0) vv = S.val[(uint32_t)src];
					else { vv = 0x100u; rel = (uint32_t)src - j0; }
				}
				uint64_t pend = __builtin_amdgcn_ballot_w64((vv & 0x100u) != 0);
				while (pend)
				{
					const uint32_t sv = (uint32_t)__shfl((int)vv, (int)rel);
					if ((vv & 0x100u) && !(sv & 0x100u)) vv = sv;
					pend = __builtin_amdgcn_ballot_w64((vv & 0x100u) != 0);
				}
				if (j < B) { S.val[j] = (uint8_t)vv; __builtin_amdgcn_raw_buffer_store_b8((uint8_t)vv, rs, (int)(P + j), 0, 0); }
				__builtin_amdgcn_wave_barrier();
			}
			P += B; t0 += ntake;
		}
		if (lane == 0 && status[b].error == 0 && P != usize) status[b].error = 17;
		if (lane == 0) status[b].produced = P;
	}
}

void launch_huff_tokens_v2(const uint8_t* d_comp, const BlockDesc* d_blocks, int64_t n_blocks, BlockStatus* d_status,
                        const uint64_t* d_tok_off, uint32_t* d_tok, uint32_t* d_tok_count, unsigned long long* d_work, const uint32_t* d_order, int max_wgs, hipStream_t s)
{
	if (n_blocks <= 0) return;
	// d_work: the launch's member queue head (zeroed by the caller). One-wave workgroups; 7 fit a CU (22.8 KB LDS each).
	int64_t wgs = (n_blocks + 63) / 64;
	int grid1 = (int)(wgs < max_wgs ? wgs : max_wgs);
	const char* pe = getenv("NGSQC_P1_PARK"); int park_hi = pe ? atoi(pe) : 16;
	const char* pr = getenv("NGSQC_P1_PRIO"); const int prio = pr ? atoi(pr) : 0;   // (measured: no effect on the pipelined K1, 70.8 vs 70.9 ms per 48 M reads)
	park_hi = (park_hi & 255) | (prio << 8);
	const char* epad = getenv("NGSQC_P1_PAD"); const bool pad = !epad || atoi(epad) != 0;   // 0: 22.8 KB workgroups (a seventh may squeeze in beside the phase-2 workgroups)
	if (pad) hipLaunchKernelGGL(huff_tokens_kernel<P1_PAD_W>, dim3(grid1), dim3(64), 0, s, d_comp, d_blocks, n_blocks, d_tok_off, d_tok, d_tok_count, d_status, d_work, d_order, park_hi);
	else hipLaunchKernelGGL(huff_tokens_kernel<0>, dim3(grid1), dim3(64), 0, s, d_comp, d_blocks, n_blocks, d_tok_off, d_tok, d_tok_count, d_status, d_work, d_order, park_hi);
	KCHECK();
}

void launch_lz77_resolve_v2(const BlockDesc* d_blocks, int64_t n_blocks, uint8_t* d_out, BlockStatus* d_status,
                         const uint64_t* d_tok_off, const uint32_t* d_tok, const uint32_t* d_tok_count, hipStream_t s)
{
	if (n_blocks <= 0) return;
	int64_t wg2 = (n_blocks + 3) / 4;
	// one member per wave, handed out by the dispatcher: measured 72 ms per 48 M reads at 16 k workgroups, 75 ms at 2 k - 4 k (every wave strides
	// over ~10 members), 92 ms at 1 k (all workgroups resident from the start: the launch ends with a long ragged tail)
	const char* e2 = getenv("NGSQC_P2_WGS"); const int64_t cap2 = e2 ? std::max<int64_t>(1, atoll(e2)) : (int64_t)32768;
	int grid2 = (int)(wg2 < cap2 ? wg2 : cap2);
	hipLaunchKernelGGL(lz77_chunk_kernel, dim3(grid2), dim3(256), 0, s, d_tok, d_tok_off, d_tok_count, d_blocks, n_blocks, d_out, d_status); KCHECK();
}

} // namespace ngsqc
