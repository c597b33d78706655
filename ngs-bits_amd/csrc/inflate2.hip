// K1 (two-phase form) — BGZF inflate split into the part that is serial per member and the part that is not.
//
//   phase 1  huff_tokens_kernel : ONE LANE PER BGZF MEMBER. Every lane Huffman-decodes its own raw-DEFLATE stream into a
//            token stream (literal byte | match{len,dist}); 64 members advance per wave instruction, so the VALU issue
//            slots that the group kernel (inflate.hip) spends on 16 redundant lanes carry 64 independent decoders.
//            Canonical Huffman decode runs out of REGISTERS (code-length counts packed 10/5 bits per field); LDS only
//            holds the symbol-order arrays, a 64-byte input ring and an 8-token output ring per lane, all laid out
//            element-major (word k of lane l at k*64+l) so that any per-lane access pattern is bank-conflict free.
//            Input refill and token flush happen in a "service" block every 4 symbols: 16-byte loads prefetched one
//            service ahead, 16-byte token stores, one s_waitcnt per service.
//   phase 2  lz77_resolve_kernel : ONE WAVE PER MEMBER. Tokens are taken 64 at a time; a wave prefix-sum gives every
//            token its output range, then every OUTPUT BYTE of the batch gets a lane: owner token by binary search,
//            source position in periodic form (i mod dist), bytes whose source lies before the batch are gathered from
//            HBM, the (few) in-batch dependencies are resolved by iterating in LDS, and the batch is stored coalesced.
//
// Same output as inflate.hip (bit-exact; tests/test_gpu_parity.py). Integer / bit-serial work, no MFMA.
#include "common.h"

namespace ngsqc {

// ---------------------------------------------------------------------------------------------------------------- phase 1
constexpr int P1_SYM_W = 81;     // lit_sym : 288 x 9 bit as a byte plane (72 words) + a bit plane (9 words)
constexpr int P1_RING_W = 8;     // compressed input ring (32 B)
constexpr int P1_TOK_W = 8;      // token ring
constexpr int P1_LANE_W = P1_SYM_W + P1_RING_W + P1_TOK_W;   // 97 words per lane (24.8 KB per wave -> 6 waves per CU)
constexpr int P1_SERVICE = 4;    // symbols between service blocks

enum { S_NEXT = 0, S_HDR = 1, S_P1 = 2, S_P2 = 3, S_SYM = 4, S_STORED = 5, S_FINISH = 6, S_DONE = 7 };
enum { TOK_ERR_OVERFLOW = 100 };

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
	__device__ __forceinline__ uint32_t& tok(uint32_t i) const { return at(P1_SYM_W + P1_RING_W + (int)(i & (P1_TOK_W - 1))); }
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
// With v = the next 15 stream bits MSB-first: length = 1 + #{l : v >= limit_l}, index = (v >> (15 - length)) + delta.
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
	uint32_t w[15];
	__device__ __forceinline__ int decode(uint32_t bits, uint32_t& len_out) const
	{
		const uint32_t v = __brev(bits) >> 17;
		uint32_t n = 0, sel = 0;
		#pragma unroll
		for (int l = 14; l >= 0; --l) { const bool ge = v >= (w[l] & 0xffffu); n += ge ? 1u : 0u; sel = ge ? sel : w[l]; }   // sel ends as the word of the smallest l with v < limit_l
		len_out = n + 1;
		if (n >= 15) return -1;
		return (int)(v >> (14 - n)) + (int)(int16_t)(sel >> 16);
	}
	template <class CNT> __device__ __forceinline__ void build(const CNT& c)
	{
		uint32_t code = 0, o = 0;
		#pragma unroll
		for (int l = 1; l <= 15; ++l)
		{
			const uint32_t cnt = c.get_const(l - 1);
			uint32_t lim = (code + cnt) << (15 - l); if (lim > 0x8000u) lim = 0x8000u;   // over-subscribed codes are rejected by the index checks
			w[l - 1] = lim | (((o - code) & 0xffffu) << 16);
			o += cnt; code = (code + cnt) << 1;
		}
	}
};

__global__ __launch_bounds__(64) void huff_tokens_kernel(const uint8_t* __restrict__ comp, const BlockDesc* __restrict__ blocks, int64_t n_blocks,
                                                          const uint64_t* __restrict__ tok_off, uint32_t* __restrict__ tok, uint32_t* __restrict__ tok_count,
                                                          BlockStatus* __restrict__ status, unsigned long long* __restrict__ work_counter)
{
	__shared__ uint32_t lds[P1_LANE_W * 64];
	const int lane = threadIdx.x;
	P1Lds L{lds, lane};
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
	uint32_t out_n = 0, err = 0; int bfinal = 0;
	LimTab limL, limD;                                       // decode tables of the current deflate block (registers)
	DistSyms dsym; dsym.clear();                             // distance symbols sorted by (len, sym)
	#pragma unroll
	for (int i = 0; i < 15; ++i) { limL.w[i] = 0; limD.w[i] = 0; }
	LitCnt cl; DistCnt cd; cl.clear(); cd.clear();          // code-length counts
	LitCnt ol; DistCnt od; ol.clear(); od.clear();          // placement cursors of pass 2
	uint64_t ccl_lo = 0, ccl_hi = 0; uint32_t limC[7];       // code-length alphabet: sorted symbols (19 x 5 bit), limit/delta words per length 1..7
	#pragma unroll
	for (int i = 0; i < 7; ++i) limC[i] = 0;
	uint32_t h_i = 0, h_n = 0, h_nlit = 0, h_prev = 0, hdr_bits = 0; uint32_t stored_left = 0;

	auto exhausted = [&]() -> bool { return rd == wr && next_q >= n_q && !pf_valid; };
	auto refill = [&]() {   // top the bit buffer up from the ring; past the end of the member zero bits are appended
		if (bitcnt <= 32)
		{
			if (rd != wr) { bitbuf |= (uint64_t)L.ring(rd) << bitcnt; bitcnt += 32; ++rd; }
			else if (next_q >= n_q && !pf_valid) bitcnt += 32;
		}
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
	auto emit = [&](uint32_t t) { if (tok_n >= tok_cap) { err = TOK_ERR_OVERFLOW; state = S_FINISH; } else { L.tok(tok_n) = t; ++tok_n; } };

	int trip = 0;
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
					uint4 t4 = make_uint4(L.tok(tok_flushed), L.tok(tok_flushed + 1), L.tok(tok_flushed + 2), L.tok(tok_flushed + 3));
					*(uint4*)(tok_ptr + tok_flushed) = t4; tok_flushed += 4;
				}
				if (wr - rd <= (uint32_t)(P1_RING_W - 4) && next_q < n_q) { pf = comp_q[q0 + next_q]; ++next_q; pf_valid = true; }
			}
		}
		++trip;

		if (state == S_SYM)
		{
			if (ready(2))   // both refills of this symbol are guaranteed (or the stream is exhausted: zeros follow)
			{
				refill();
				uint32_t len;
				int idx = limL.decode((uint32_t)bitbuf, len);
				if (idx < 0 || idx >= 288) { err = 9; state = S_FINISH; }
				else
				{
					uint32_t s = L.litsym((uint32_t)idx);
					bitbuf >>= len; bitcnt -= len; bits_used += len;
					if (s < 256) { emit(s); ++out_n; }
					else if (s == 256) { state = bfinal ? S_FINISH : S_HDR; }
					else
					{
						s -= 257;
						if (s >= 29) { err = 10; state = S_FINISH; }
						else
						{
							uint32_t eb = s < 8 ? 0u : (s == 28 ? 0u : (s - 4) >> 2);
							uint32_t base = s < 8 ? s + 3 : (s == 28 ? 258u : ((4u + ((s - 4) & 3u)) << eb) + 3u);
							uint32_t mlen = base + take(eb);
							refill();
							uint32_t dl;
							int di = limD.decode((uint32_t)bitbuf, dl);
							if (di < 0 || di >= 30) { err = 11; state = S_FINISH; }
							else
							{
								uint32_t ds = dsym.get((uint32_t)di);
								bitbuf >>= dl; bitcnt -= dl; bits_used += dl;
								if (ds >= 30) { err = 12; state = S_FINISH; }
								else
								{
									uint32_t deb = ds < 4 ? 0u : (ds >> 1) - 1u;
									uint32_t dbase = ds < 4 ? ds + 1 : ((2u + (ds & 1u)) << deb) + 1u;
									uint32_t mdist = dbase + take(deb);
									if (mdist > out_n || out_n + mlen > usize) { err = 13; state = S_FINISH; }
									else { emit(0x80000000u | ((mlen - 3) << 23) | (mdist - 1)); out_n += mlen; }
								}
							}
						}
					}
					if (out_n > usize) { err = 3; state = S_FINISH; }
				}
			}
		}
		else if (state == S_P1 || state == S_P2)
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
		else if (state == S_STORED)
		{
			if (ready(1))
			{
				refill();
				if (out_n >= usize) { err = 3; state = S_FINISH; }
				else { emit(take(8)); ++out_n; if (--stored_left == 0 && state == S_STORED) state = bfinal ? S_FINISH : S_HDR; }
			}
		}
		else if (state == S_FINISH)
		{
			// flush the tail of the token ring, publish counts
			if (err != TOK_ERR_OVERFLOW) for (uint32_t i = tok_flushed; i < tok_n; ++i) tok_ptr[i] = L.tok(i);
			if (!err && out_n != usize) err = 14;
			tok_count[b] = tok_n; status[b].produced = out_n; status[b].error = err;
			state = S_NEXT;
		}
		else if (state == S_NEXT)
		{
			b = (int64_t)atomicAdd(work_counter, 1ull);
			if (b >= n_blocks) state = S_DONE;
			else
			{
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

struct P2Lds { unsigned long long endmask[P2_BMAX / 64]; uint16_t src[P2_BMAX + 64]; uint8_t val[P2_BMAX + 64]; };

__global__ __launch_bounds__(256) void lz77_resolve_kernel(const uint32_t* __restrict__ tok, const uint64_t* __restrict__ tok_off, const uint32_t* __restrict__ tok_count,
                                                           const BlockDesc* __restrict__ blocks, int64_t n_blocks, uint8_t* __restrict__ out_base, BlockStatus* __restrict__ status)
{
	__shared__ P2Lds lds[4];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	P2Lds& S = lds[wv];
	constexpr int WAIT_VM0 = 0x0F70;
	const int64_t n_waves = (int64_t)gridDim.x * 4;
	for (int64_t b = (int64_t)blockIdx.x * 4 + wv; b < n_blocks; b += n_waves)
	{
		if (status[b].error) continue;
		const uint32_t n = tok_count[b];
		const uint32_t* T = tok + tok_off[b];
		uint8_t* out = out_base + blocks[b].upos;
		const uint32_t usize = blocks[b].usize;
		uint32_t P = 0;   // bytes written so far
		for (uint32_t t0 = 0; t0 < n;)
		{
			const uint32_t i = t0 + (uint32_t)lane;
			const uint32_t tk = i < n ? T[i] : 0u;
			const bool is_m = tk >> 31;
			const uint32_t len = i < n ? (is_m ? ((tk >> 23) & 255u) + 3u : 1u) : 0u;
			const uint32_t dist = (tk & 0x7fffu) + 1u;
			uint32_t end = len;
			#pragma unroll
			for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(end, o); if (lane >= o) end += t; }
			// take the longest token prefix whose output fits the staging buffer
			const uint64_t fit = __builtin_amdgcn_ballot_w64(i < n && end <= (uint32_t)P2_BMAX);
			uint32_t ntake = (uint32_t)__popcll(fit); if (ntake == 0) ntake = 1;
			const uint32_t B = (uint32_t)__shfl((int)end, (int)ntake - 1);
			const uint32_t start = end - len;
			if (P + B > usize) { if (lane == 0) status[b].error = 16; break; }
			// stores of earlier batches must be complete before this batch gathers from the window
			__builtin_amdgcn_s_waitcnt(WAIT_VM0);
			bool any_unres = false;
			// token-end bitmap of the batch: bit (e-1) set when a token ends at byte e (ends are strictly increasing)
			if (lane < P2_BMAX / 64) S.endmask[lane] = 0ull;
			__builtin_amdgcn_wave_barrier();
			if ((uint32_t)lane < ntake && len != 0) atomicOr(&S.endmask[(end - 1) >> 6], 1ull << ((end - 1) & 63u));
			__builtin_amdgcn_wave_barrier();
			uint32_t ta = 0;   // tokens that end at or before the current chunk start
			for (uint32_t j0 = 0; j0 < B; j0 += 64)
			{
				const uint32_t j = j0 + (uint32_t)lane;
				// owner token of byte j = ta + #tokens ending inside the chunk at or before j. The token ends inside the chunk
				// are strictly increasing, so they form a 64-bit mask (bit p <-> some token ends at j0 + p + 1), read from the
				// batch's end bitmap; a popcount of the bits below (j - j0) ranks the byte.
				const uint64_t m = S.endmask[j0 >> 6];
				const uint32_t pj = (uint32_t)lane;   // j - j0
				const uint32_t o = ta + (uint32_t)__popcll(m & ((1ull << pj) - 1ull));
				ta += (uint32_t)__popcll(m);
				const uint32_t tko = (uint32_t)__shfl((int)tk, (int)(o & 63u));
				const uint32_t sto = (uint32_t)__shfl((int)start, (int)(o & 63u));
				uint32_t sidx = 0xffffu;
				if (j < B)
				{
					uint32_t v = tko & 255u;
					if (tko >> 31)
					{
						const uint32_t d = (tko & 0x7fffu) + 1u, off = j - sto;
						uint32_t r = off;
						if (off >= d)
						{
							uint32_t q = (uint32_t)((float)off * __frcp_rn((float)d)); int rr = (int)off - (int)(q * d);
							if (rr < 0) rr += (int)d; else if (rr >= (int)d) rr -= (int)d;
							r = (uint32_t)rr;
						}
						const int src = (int)sto - (int)d + (int)r;   // relative to P
						if (src < 0) v = out[(int64_t)P + src];
						else { sidx = (uint32_t)src; any_unres = true; }
					}
					S.val[j] = (uint8_t)v; S.src[j] = (uint16_t)sidx;
				}
			}
			// in-batch dependencies: a byte copies an EARLIER byte of the same batch; iterate until all are resolved
			uint64_t pending = __builtin_amdgcn_ballot_w64(any_unres);
			while (pending)
			{
				bool still = false;
				for (uint32_t j0 = 0; j0 < B; j0 += 64)
				{
					const uint32_t j = j0 + (uint32_t)lane;
					if (j < B)
					{
						const uint32_t s = S.src[j];
						if (s != 0xffffu)
						{
							if (S.src[s] == 0xffffu) { S.val[j] = S.val[s]; S.src[j] = 0xffffu; }
							else still = true;
						}
					}
					__builtin_amdgcn_wave_barrier();
				}
				pending = __builtin_amdgcn_ballot_w64(still);
			}
			for (uint32_t j0 = 0; j0 < B; j0 += 64) { const uint32_t j = j0 + (uint32_t)lane; if (j < B) out[P + j] = S.val[j]; }
			P += B; t0 += ntake;
		}
		if (lane == 0 && status[b].error == 0 && P != usize) status[b].error = 17;
		if (lane == 0) status[b].produced = P;
	}
}

void launch_huff_tokens(const uint8_t* d_comp, const BlockDesc* d_blocks, int64_t n_blocks, BlockStatus* d_status,
                        const uint64_t* d_tok_off, uint32_t* d_tok, uint32_t* d_tok_count, hipStream_t s)
{
	if (n_blocks <= 0) return;
	// d_tok_count has n_blocks + 8 entries: the last 8 bytes (8-byte aligned) are the phase-1 work counter
	unsigned long long* d_work = (unsigned long long*)(d_tok_count + ((n_blocks + 1) & ~1ll));
	hipMemsetAsync(d_work, 0, sizeof(unsigned long long), s);
	int64_t wgs = (n_blocks + 63) / 64;
	int grid1 = (int)(wgs < 256 * 6 ? wgs : 256 * 6);   // 6 one-wave workgroups fit a CU (24.8 KB LDS each)
	hipLaunchKernelGGL(huff_tokens_kernel, dim3(grid1), dim3(64), 0, s, d_comp, d_blocks, n_blocks, d_tok_off, d_tok, d_tok_count, d_status, d_work);
}

void launch_lz77_resolve(const BlockDesc* d_blocks, int64_t n_blocks, uint8_t* d_out, BlockStatus* d_status,
                         const uint64_t* d_tok_off, const uint32_t* d_tok, const uint32_t* d_tok_count, hipStream_t s)
{
	if (n_blocks <= 0) return;
	int64_t wg2 = (n_blocks + 3) / 4;
	int grid2 = (int)(wg2 < 256 * 8 ? wg2 : 256 * 8);
	hipLaunchKernelGGL(lz77_resolve_kernel, dim3(grid2), dim3(256), 0, s, d_tok, d_tok_off, d_tok_count, d_blocks, n_blocks, d_out, d_status);
}

} // namespace ngsqc
