// K1 — BGZF block inflate (RFC 1951 DEFLATE) for gfx950.
//
// Replaces what the reference gets from htslib inside sam_read1() (src/cppNGS/BamReader.h:388): every BGZF member is an
// independent raw-DEFLATE stream of <= 64 KiB output, so the parallelism is ACROSS members.
//
// Mapping to the wave: a 64-lane wave hosts 64/G independent decoder GROUPS of G lanes, one BGZF member per group.
//   * GROUP-REDUNDANT DECODE: every lane of a group carries the same bit-reader / Huffman state and executes the decode
//     redundantly (SIMT executes the lanes together anyway). There is no leader lane, so no exec-mask juggling and no
//     broadcast of decoded symbols; table look-ups are same-address LDS broadcasts.
//   * LOCKSTEP: the wave runs ONE flat loop; each trip decodes one symbol for every group (state machine per group), so
//     the instruction stream is shared by all groups instead of being issued once per member.
//   * the compressed stream lives in REGISTERS: lane j of a group holds word base+j (coalesced dword loads by the whole
//     group, the following G words prefetched one batch ahead); the decoder pulls a word with a cross-lane read
//     (ds_bpermute) — no LDS ring, and HBM/L2 latency never sits on the decode dependency chain;
//   * literals are stored by lane 0 of the group, every LZ77 match is copied cooperatively by the G lanes
//     (periodic-source form, so overlapping matches need no intra-copy ordering);
//   * per-group LDS is only the two 16-bit Huffman look-up tables + the canonical arrays for long codes (~2.4 KiB), which
//     is what bounds the number of members in flight per CU (LDS 160 KiB).
// Integer / bit-serial work: no MFMA; bounded by issue rate and LDS/L2 latency, not by HBM bandwidth.
#include "common.h"
#include <cstdlib>

namespace ngsqc {

template <int LIT_BITS, int DIST_BITS>
struct GroupTables
{
	uint16_t lit_lut[1 << LIT_BITS];   // (sym << 4) | code length ; 0 = not in the fast table
	uint16_t dist_lut[1 << DIST_BITS];
	uint16_t lit_sym[288];             // symbols sorted by (len, sym): canonical decode of codes longer than the LUT
	uint16_t dist_sym[32];
	uint16_t lit_cnt[16];
	uint16_t dist_cnt[16];
	uint16_t offs[16];
	uint16_t next_code[16];
	uint8_t  lens[340];                // [0..20) code-length code lengths, [20..20+316) litlen+dist code lengths
};

// RFC 1951 code-length alphabet order {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15} packed 5 bits each
__device__ __forceinline__ uint32_t clorder(int k)
{
	const uint64_t lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
	const uint64_t hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
	return (uint32_t)((k < 12 ? lo >> (5 * k) : hi >> (5 * (k - 12))) & 31u);
}

// canonical decode for codes longer than the fast table: puff-style, LSB-first bits. returns (sym << 4) | len, 0 = invalid
__device__ static uint32_t slow_decode(uint32_t bits, const uint16_t* cnt, const uint16_t* sym)
{
	int code = 0, first = 0, index = 0;
	for (int len = 1; len <= 15; ++len)
	{
		code |= (int)(bits & 1); bits >>= 1;
		int count = cnt[len];
		if (code - count < first) return ((uint32_t)sym[index + (code - first)] << 4) | (uint32_t)len;
		index += count; first += count; first <<= 1; code <<= 1;
	}
	return 0;
}

// Build canonical tables from code lengths. Executed redundantly by every lane of the group (same values, same
// addresses); the LUT must have been zeroed beforehand.
__device__ static void build_tables(const uint8_t* lens, int n, uint16_t* cnt, uint16_t* sym, uint16_t* lut, int bits, uint16_t* offs, uint16_t* next_code)
{
	for (int i = 0; i < 16; ++i) cnt[i] = 0;
	__builtin_amdgcn_wave_barrier();
	{
		// histogram of code lengths in registers (4 bits per... up to 288 -> 16 counters of 9+ bits): avoid LDS read-modify-write chains
		uint32_t c[16];
		#pragma unroll
		for (int i = 0; i < 16; ++i) c[i] = 0;
		for (int s = 0; s < n; ++s)
		{
			uint32_t l = lens[s];
			#pragma unroll
			for (int i = 1; i < 16; ++i) c[i] += (l == (uint32_t)i);
		}
		uint32_t o = 0, code = 0;
		#pragma unroll
		for (int l = 1; l <= 15; ++l) { cnt[l] = (uint16_t)c[l]; offs[l] = (uint16_t)o; o += c[l]; next_code[l] = (uint16_t)code; code = (code + c[l]) << 1; }
	}
	__builtin_amdgcn_wave_barrier();
	for (int s = 0; s < n; ++s)
	{
		int l = lens[s];
		if (!l) continue;
		uint32_t k = offs[l]; offs[l] = (uint16_t)(k + 1);
		sym[k] = (uint16_t)s;
		uint32_t c = next_code[l]; next_code[l] = (uint16_t)(c + 1);
		if (l <= bits)
		{
			uint32_t rev = __brev(c) >> (32 - l);
			uint16_t e = (uint16_t)((s << 4) | l);
			for (uint32_t i = rev; i < (1u << bits); i += (1u << l)) lut[i] = e;
		}
	}
	__builtin_amdgcn_wave_barrier();
}

enum { ST_NEXT_MEMBER = 0, ST_BLOCK_HEADER = 1, ST_SYMBOLS = 2, ST_DONE = 3 };

template <int G, int LIT_BITS, int DIST_BITS, int DBG = 0>
__global__ __launch_bounds__(256) void bgzf_inflate_kernel(const uint8_t* __restrict__ comp, const BlockDesc* __restrict__ blocks, int64_t n_blocks,
                                                            uint8_t* __restrict__ out_base, BlockStatus* __restrict__ status)
{
	constexpr int GROUPS = 256 / G;
	__shared__ GroupTables<LIT_BITS, DIST_BITS> tabs[GROUPS];
	const int tid = threadIdx.x;
	const int g = tid / G;            // group within workgroup
	const int gl = tid % G;           // lane within group
	const int lane = tid & 63;
	const int gbase = lane - gl;      // wave-relative lane index of this group's lane 0
	GroupTables<LIT_BITS, DIST_BITS>& T = tabs[g];

	// ---- per-group state (identical in all lanes of the group unless noted) ----
	int state = ST_NEXT_MEMBER;
	int64_t b = (int64_t)blockIdx.x * GROUPS + g - (int64_t)gridDim.x * GROUPS;   // advanced before first use
	const uint32_t* const comp_words = (const uint32_t*)comp;   // hipMalloc'ed: 256-byte aligned
	uint64_t word0 = 0; uint32_t n_words = 0, misalign = 0, clen = 0, usize = 0;
	uint8_t* out = nullptr; uint8_t* out_lane = nullptr; const uint8_t* pay = nullptr;   // out_lane = out + gl
	uint4 myw = make_uint4(0, 0, 0, 0);  // per lane: absolute words abase + 4*gl .. +3 of the compressed image (16-byte load)
	uint64_t abase = 0;                  // absolute word index of the group's resident batch (4*G words)
	uint32_t w = 0, w0 = 0, w1 = 0, shift = 0;   // w = payload-relative word index of w0
	uint32_t out_pos = 0, err = 0; int bfinal = 0;
	// Memory ordering: a load that may read bytes stored earlier by this wave is only issued after an explicit
	// s_waitcnt vmcnt(0) (the match path below); literals never read memory.
	constexpr int WAIT_VM0 = 0x0F70;   // s_waitcnt vmcnt(0) (expcnt/lgkmcnt untouched), gfx9 encoding

	const uint4* const comp_q = (const uint4*)comp;
	auto load_batch = [&]() {   // synchronous: nothing stays in flight across the loop back-edge (keeps s_waitcnt out of the hot trips)
		myw = comp_q[(abase >> 2) + gl];
		__builtin_amdgcn_s_waitcnt(WAIT_VM0);
	};
	uint32_t kbase = 0;   // payload-relative word index of the batch's first word (may be "negative": wraps, only differences are used)
	auto fetch = [&](uint32_t k) -> uint32_t {   // payload-relative word k of the compressed stream
		uint32_t idx = k - kbase;
		if (idx >= (uint32_t)(4 * G)) { uint64_t a = word0 + k; abase = a & ~(uint64_t)(4 * G - 1); kbase = k - (uint32_t)(a - abase); load_batch(); idx = k - kbase; }
		uint32_t sel = idx & 3u;
		uint32_t mine = sel == 0 ? myw.x : (sel == 1 ? myw.y : (sel == 2 ? myw.z : myw.w));
		return (uint32_t)__shfl((int)mine, gbase + (int)(idx >> 2));
	};
	auto norm = [&]() { if (shift >= 32) { shift -= 32; ++w; w0 = w1; w1 = fetch(w + 1); } };
	auto peek = [&]() -> uint32_t { return __builtin_amdgcn_alignbit(w1, w0, shift); };  // 32 valid bits when shift < 32
	auto get = [&](uint32_t n) -> uint32_t { norm(); uint32_t v = peek() & ((1u << n) - 1u); shift += n; return v; };   // n <= 16
	auto reader_init = [&](uint32_t byte_rel) {   // restart the bit reader at a payload-relative byte position
		uint32_t abs_byte = misalign + byte_rel;
		w = abs_byte / 4; shift = (abs_byte & 3) * 8;
		abase = (word0 + w) & ~(uint64_t)(4 * G - 1); kbase = w - (uint32_t)(word0 + w - abase); load_batch();
		w0 = fetch(w); w1 = fetch(w + 1);
	};

	while (true)
	{
		if (state == ST_NEXT_MEMBER)
		{
			b += (int64_t)gridDim.x * GROUPS;
			if (b >= n_blocks) state = ST_DONE;
			else
			{
				const BlockDesc bd = blocks[b];
				__builtin_amdgcn_s_waitcnt(WAIT_VM0);
				out = out_base + bd.upos; out_lane = out + gl; usize = bd.usize; clen = bd.clen;
				pay = comp + bd.cpos;
				word0 = bd.cpos >> 2;
				misalign = (uint32_t)(bd.cpos & 3);
				n_words = (misalign + clen + 3) / 4 + 1;   // one word of slack (the compressed image is padded)
				reader_init(0);
				out_pos = 0; err = 0; bfinal = 0;
				state = ST_BLOCK_HEADER;
			}
		}
		else if (state == ST_BLOCK_HEADER)
		{
			if (bfinal || err)
			{
				if (!err && out_pos != usize) err = 14;
				if (gl == 0) { status[b].produced = out_pos; status[b].error = err; }
				state = ST_NEXT_MEMBER;
			}
			else
			{
				bfinal = (int)get(1);
				int btype = (int)get(2);
				if (w > n_words + 2) err = 1;
				else if (btype == 0)
				{
					// stored block: skip to byte boundary, LEN, NLEN, then LEN raw bytes
					norm(); shift = (shift + 7u) & ~7u; norm();
					uint32_t v = peek(); shift += 32; norm();
					uint32_t len = v & 0xffffu;
					uint32_t rel = w * 4 + (shift >> 3) - misalign;   // payload-relative byte position
					if ((len ^ (v >> 16)) != 0xffffu) err = 2;
					else if (out_pos + len > usize || rel + len > clen) err = 3;
					else
					{
						const uint8_t* src = pay + rel;
						for (uint32_t i = gl; i < len; i += G) out[out_pos + i] = src[i];
						out_pos += len;
						reader_init(rel + len);
					}
				}
				else if (btype == 3) err = 4;
				else
				{
					// ---- Huffman tables ----
					for (int i = gl; i < (1 << LIT_BITS); i += G) T.lit_lut[i] = 0;
					for (int i = gl; i < (1 << DIST_BITS); i += G) T.dist_lut[i] = 0;
					__builtin_amdgcn_wave_barrier();
					int nlit, ndist;
					uint8_t* L = T.lens + 20; // litlen code lengths, then dist code lengths
					if (btype == 1)
					{
						for (int i = gl; i < 288; i += G) L[i] = (uint8_t)(i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)));
						for (int i = gl; i < 30; i += G) L[288 + i] = 5;
						nlit = 288; ndist = 30;
						__builtin_amdgcn_wave_barrier();
					}
					else
					{
						nlit = (int)get(5) + 257; ndist = (int)get(5) + 1; int ncl = (int)get(4) + 4;
						if (nlit > 286 || ndist > 30) err = 5;
						uint8_t* cl = T.lens; // 19 code-length code lengths, built into dist_lut (7-bit table) temporarily
						for (int i = gl; i < 19; i += G) cl[i] = 0;
						__builtin_amdgcn_wave_barrier();
						for (int i = 0; i < ncl; ++i) { uint32_t v = get(3); cl[clorder(i)] = (uint8_t)v; }
						__builtin_amdgcn_wave_barrier();
						build_tables(cl, 19, T.dist_cnt, T.dist_sym, T.dist_lut, 7, T.offs, T.next_code);
						int i = 0, n = nlit + ndist; uint32_t prev = 0;
						while (i < n && !err)
						{
							norm();
							uint32_t e = T.dist_lut[peek() & 127u];
							if (!(e & 15u)) { err = 6; break; }
							shift += (e & 15u);
							uint32_t s = e >> 4;
							if (s < 16) { L[i++] = (uint8_t)s; prev = s; }
							else
							{
								uint32_t rep, val = 0;
								if (s == 16) { if (i == 0) { err = 7; break; } rep = 3 + get(2); val = prev; }
								else if (s == 17) { rep = 3 + get(3); prev = 0; }
								else { rep = 11 + get(7); prev = 0; }
								if (i + (int)rep > n) { err = 8; break; }
								for (uint32_t k = 0; k < rep; ++k) L[i++] = (uint8_t)val;
							}
						}
						__builtin_amdgcn_wave_barrier();
						if (!err && L[256] == 0) err = 15; // no end-of-block code
						for (int k = gl; k < (1 << DIST_BITS); k += G) T.dist_lut[k] = 0;
						__builtin_amdgcn_wave_barrier();
					}
					if (!err)
					{
						build_tables(L, nlit, T.lit_cnt, T.lit_sym, T.lit_lut, LIT_BITS, T.offs, T.next_code);
						build_tables(L + nlit, ndist, T.dist_cnt, T.dist_sym, T.dist_lut, DIST_BITS, T.offs, T.next_code);
						state = ST_SYMBOLS;
					}
				}
			}
		}
		else if (state == ST_SYMBOLS)
		{
			// ---- one Huffman symbol per trip ----
			norm();
			uint32_t bits = peek();
			uint32_t e = T.lit_lut[bits & ((1u << LIT_BITS) - 1u)];
			if (!(e & 15u)) e = slow_decode(bits, T.lit_cnt, T.lit_sym);
			uint32_t l = e & 15u, s = e >> 4;
			if (!e) { err = 9; state = ST_BLOCK_HEADER; }
			else if (s < 256)
			{
				shift += l;
				if (out_pos >= usize) { err = 3; state = ST_BLOCK_HEADER; }
				else { if (DBG < 2 && gl == 0) out_lane[out_pos] = (uint8_t)s; ++out_pos; }
			}
			else if (s == 256) { shift += l; state = ST_BLOCK_HEADER; }
			else
			{
				s -= 257;
				if (s >= 29) { err = 10; state = ST_BLOCK_HEADER; }
				else
				{
					// length base / extra bits computed arithmetically (RFC 1951 §3.2.5)
					uint32_t eb = s < 8 ? 0u : (s == 28 ? 0u : (s - 4) >> 2);
					uint32_t base = s < 8 ? s + 3 : (s == 28 ? 258u : ((4u + ((s - 4) & 3u)) << eb) + 3u);
					uint32_t mlen = base + ((bits >> l) & ((1u << eb) - 1u));
					shift += l + eb;
					norm();
					bits = peek();
					e = T.dist_lut[bits & ((1u << DIST_BITS) - 1u)];
					if (!(e & 15u)) e = slow_decode(bits, T.dist_cnt, T.dist_sym);
					l = e & 15u; s = e >> 4;
					if (!e || s >= 30) { err = 11; state = ST_BLOCK_HEADER; }
					else
					{
						eb = s < 4 ? 0u : (s >> 1) - 1u;
						base = s < 4 ? s + 1 : ((2u + (s & 1u)) << eb) + 1u;
						uint32_t mdist = base + ((bits >> l) & ((1u << eb) - 1u));
						shift += l + eb;
						if (mdist > out_pos || out_pos + mlen > usize || w > n_words + 2) { err = 13; state = ST_BLOCK_HEADER; }
						else
						{
							if (DBG < 1)
							{
								// a load may read bytes this wave stored in earlier trips: make those stores complete first
								__builtin_amdgcn_s_waitcnt(WAIT_VM0);
								uint8_t* dstl = out_lane + out_pos;            // this lane's first destination byte
								// source index of byte i is i mod dist (periodic form: every source byte precedes the match)
								uint32_t r = (uint32_t)gl, step = (uint32_t)G;
								if (mdist < (uint32_t)G)
								{
									// floor(x / d) for x <= 16, d < 16 via the 8.8 reciprocal ceil(256/d) (exact in this range)
									const uint64_t RLO = 0ull | (0ull << 8) | (128ull << 16) | (86ull << 24) | (64ull << 32) | (52ull << 40) | (43ull << 48) | (37ull << 56);   // d = 0..7 (d=1 handled below)
									const uint64_t RHI = 32ull | (29ull << 8) | (26ull << 16) | (24ull << 24) | (22ull << 32) | (20ull << 40) | (19ull << 48) | (18ull << 56);   // d = 8..15
									uint32_t m = (uint32_t)((mdist < 8 ? RLO >> (8 * mdist) : RHI >> (8 * (mdist - 8))) & 255u);
									uint32_t q = mdist == 1 ? (uint32_t)gl : ((uint32_t)gl * m) >> 8;
									r = (uint32_t)gl - q * mdist;
									q = mdist == 1 ? (uint32_t)G : ((uint32_t)G * m) >> 8;
									step = (uint32_t)G - q * mdist;
								}
								const uint8_t* srcl = dstl - mdist - gl;       // window start (byte 0 of the period)
								if ((uint32_t)gl < mlen) dstl[0] = srcl[r];
								for (uint32_t i = gl + G; i < mlen; i += G)
								{
									r += step; if (r >= mdist) r -= mdist;
									dstl[i - gl] = srcl[r];
								}
							}
							out_pos += mlen;
						}
					}
				}
			}
		}
		else break;   // ST_DONE (a wave leaves the loop when all its groups are done)
	}
}

template <int G, int LB, int DB, int DBG = 0>
static void launch_inflate_t(const uint8_t* d_comp, const BlockDesc* d_blocks, int64_t n_blocks, uint8_t* d_out, BlockStatus* d_status, hipStream_t s)
{
	constexpr int GROUPS = 256 / G;
	int64_t wgs = (n_blocks + GROUPS - 1) / GROUPS;
	int64_t cap = 256 * 8; // persistent-style grid: the flat loop walks members grid-stride
	int grid = (int)(wgs < cap ? wgs : cap);
	hipLaunchKernelGGL((bgzf_inflate_kernel<G, LB, DB, DBG>), dim3(grid), dim3(256), 0, s, d_comp, d_blocks, n_blocks, d_out, d_status);
}

void launch_inflate(const uint8_t* d_comp, const BlockDesc* d_blocks, int64_t n_blocks, uint8_t* d_out, BlockStatus* d_status, hipStream_t s)
{
	if (n_blocks <= 0) return;
	const char* ev = getenv("NGSQC_INFLATE_VARIANT"); const int variant = ev ? atoi(ev) : 0;
	switch (variant)
	{
		case 1: launch_inflate_t<32, 10, 8>(d_comp, d_blocks, n_blocks, d_out, d_status, s); break;
		case 2: launch_inflate_t<16, 10, 8>(d_comp, d_blocks, n_blocks, d_out, d_status, s); break;
		case 3: launch_inflate_t<16, 9, 7>(d_comp, d_blocks, n_blocks, d_out, d_status, s); break;
		case 4: launch_inflate_t<8, 9, 7>(d_comp, d_blocks, n_blocks, d_out, d_status, s); break;
		case 5: launch_inflate_t<32, 9, 7>(d_comp, d_blocks, n_blocks, d_out, d_status, s); break;
		case 6: launch_inflate_t<64, 10, 8>(d_comp, d_blocks, n_blocks, d_out, d_status, s); break;
		case 7: launch_inflate_t<8, 10, 8>(d_comp, d_blocks, n_blocks, d_out, d_status, s); break;
		case 12: launch_inflate_t<16, 9, 7, 1>(d_comp, d_blocks, n_blocks, d_out, d_status, s); break;
		case 13: launch_inflate_t<16, 9, 7, 2>(d_comp, d_blocks, n_blocks, d_out, d_status, s); break;
		default: launch_inflate_t<16, 9, 7>(d_comp, d_blocks, n_blocks, d_out, d_status, s); break;
	}
}

} // namespace ngsqc
