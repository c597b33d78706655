// K1 — BGZF block inflate (RFC 1951 DEFLATE) for gfx950.
//
// Replaces what the reference gets from htslib inside sam_read1() (src/cppNGS/BamReader.h:388): every BGZF member is an
// independent raw-DEFLATE stream of <= 64 KiB output, so the parallelism is ACROSS members. One decoder group of G lanes
// per member: the group's leader lane runs the bit-serial Huffman decode out of LDS tables; literals are stored by the
// leader, every LZ77 match is copied cooperatively by all G lanes (periodic-source form, so overlapping matches need no
// intra-copy ordering). The bit reader keeps a 128-bit window in registers (w0..w3) that is refilled with aligned dword
// loads two words ahead of use, so the HBM/L2 latency of the compressed stream is off the decode dependency chain.
//
// This kernel is integer / bit-serial work: no MFMA. It is bounded by per-wave issue rate and LDS latency, not HBM.
#include "common.h"

namespace ngsqc {

constexpr int LIT_BITS = 10;
constexpr int DIST_BITS = 8;
constexpr int MAX_LIT_RUN = 16;   // leader re-syncs with its group at least every MAX_LIT_RUN literals

struct GroupTables
{
	uint16_t lit_lut[1 << LIT_BITS];   // (sym << 4) | len, 0 = not in fast table
	uint16_t dist_lut[1 << DIST_BITS];
	uint16_t lit_sym[288];             // symbols sorted by (len, sym) for the canonical slow path
	uint16_t dist_sym[32];
	uint16_t lit_cnt[16];
	uint16_t dist_cnt[16];
	uint16_t offs[16];
	uint16_t next_code[16];
	uint8_t  lens[344];                // [0..20) code-length code lengths, [20..20+316) litlen+dist code lengths
};

__constant__ uint16_t c_lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
__constant__ uint8_t  c_lext[29]  = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
__constant__ uint16_t c_dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
__constant__ uint8_t  c_dext[30]  = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
__constant__ uint8_t  c_clorder[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

struct BitReader
{
	const uint32_t* base; uint32_t idx; uint32_t limit; // idx = next word to load; limit = last word index that may be loaded
	uint32_t w0, w1, w2, w3; uint32_t shift;
	__device__ __forceinline__ uint32_t ld(uint32_t i) const { return i <= limit ? base[i] : 0u; }
	__device__ void init(const uint8_t* p, uint32_t nbytes)
	{
		uintptr_t a = (uintptr_t)p;
		base = (const uint32_t*)(a & ~(uintptr_t)3);
		shift = (uint32_t)(a & 3) * 8;
		limit = (uint32_t)(((a & 3) + nbytes + 3) / 4); // one word of slack; the compressed image is padded
		w0 = ld(0); w1 = ld(1); w2 = ld(2); w3 = ld(3); idx = 4;
	}
	__device__ __forceinline__ void norm() { if (shift >= 32) { shift -= 32; w0 = w1; w1 = w2; w2 = w3; w3 = ld(idx); ++idx; } }
	__device__ __forceinline__ uint32_t peek() const { return __builtin_amdgcn_alignbit(w1, w0, shift); } // 32 valid bits, shift < 32
	__device__ __forceinline__ void consume(uint32_t n) { shift += n; }
	__device__ __forceinline__ uint32_t get(uint32_t n) { norm(); uint32_t v = peek() & ((1u << n) - 1u); consume(n); return v; } // n <= 16
	__device__ __forceinline__ uint64_t bytepos() const { return (uint64_t)(idx - 4) * 4 + (shift >> 3); } // relative to base
	__device__ __forceinline__ bool overrun() const { return idx > limit + 6; }
};

// canonical decode for codes longer than the fast table (and as the general fallback): puff-style, LSB-first bits
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

// Build canonical tables from code lengths (leader lane only; LUT must be zeroed by the group beforehand).
__device__ static void build_tables(const uint8_t* lens, int n, uint16_t* cnt, uint16_t* sym, uint16_t* lut, int bits, uint16_t* offs, uint16_t* next_code)
{
	for (int i = 0; i < 16; ++i) cnt[i] = 0;
	for (int s = 0; s < n; ++s) cnt[lens[s]]++;
	cnt[0] = 0;
	uint32_t o = 0, code = 0;
	for (int l = 1; l <= 15; ++l) { offs[l] = (uint16_t)o; o += cnt[l]; next_code[l] = (uint16_t)code; code = (code + cnt[l]) << 1; }
	for (int s = 0; s < n; ++s)
	{
		int l = lens[s];
		if (!l) continue;
		sym[offs[l]++] = (uint16_t)s;
		uint32_t c = next_code[l]++;
		if (l <= bits)
		{
			uint32_t rev = __brev(c) >> (32 - l);
			uint16_t e = (uint16_t)((s << 4) | l);
			for (uint32_t i = rev; i < (1u << bits); i += (1u << l)) lut[i] = e;
		}
	}
}

template <int G>
__global__ __launch_bounds__(256) void bgzf_inflate_kernel(const uint8_t* __restrict__ comp, const BlockDesc* __restrict__ blocks, int64_t n_blocks,
                                                            uint8_t* __restrict__ out_base, BlockStatus* __restrict__ status)
{
	constexpr int GROUPS = 256 / G;
	__shared__ GroupTables tabs[GROUPS];
	const int tid = threadIdx.x;
	const int g = tid / G;          // group within workgroup
	const int gl = tid % G;         // lane within group
	const int lane = tid & 63;
	const int leader_lane = lane - gl; // wave-relative lane index of this group's leader
	const bool leader = gl == 0;
	GroupTables& T = tabs[g];

	for (int64_t b = (int64_t)blockIdx.x * GROUPS + g; b < n_blocks; b += (int64_t)gridDim.x * GROUPS)
	{
		const BlockDesc bd = blocks[b];
		uint8_t* out = out_base + bd.upos;
		const uint32_t usize = bd.usize;
		BitReader br;
		uint32_t out_pos = 0;
		uint32_t err = 0;
		if (leader) br.init(comp + bd.cpos, bd.clen);
		int bfinal = 0;
		while (!bfinal && !err)
		{
			// ---- block header (leader) ----
			int btype = 0;
			if (leader) { bfinal = (int)br.get(1); btype = (int)br.get(2); if (br.overrun()) err = 1; }
			bfinal = __shfl(bfinal, leader_lane); btype = __shfl(btype, leader_lane); err = __shfl(err, leader_lane);
			if (err) break;
			if (btype == 0)
			{
				// stored block: skip to byte boundary, LEN, NLEN, then LEN raw bytes
				uint32_t len = 0, rel = 0;
				if (leader)
				{
					br.norm(); br.shift = (br.shift + 7u) & ~7u; br.norm();
					uint32_t v = br.peek(); br.consume(32); br.norm();
					len = v & 0xffffu;
					if ((len ^ (v >> 16)) != 0xffffu) err = 2;
					rel = (uint32_t)(((const uint8_t*)br.base + br.bytepos()) - (comp + bd.cpos)); // payload-relative byte position
					if (out_pos + len > usize || rel + len > bd.clen) err = 3;
				}
				len = __shfl(len, leader_lane); err = __shfl(err, leader_lane); rel = __shfl(rel, leader_lane);
				uint32_t opos = __shfl(out_pos, leader_lane);
				if (err) break;
				const uint8_t* src = comp + bd.cpos + rel;
				for (uint32_t i = gl; i < len; i += G) out[opos + i] = src[i];
				if (leader)
				{
					out_pos += len;
					br.init(src + len, bd.clen - (rel + len));
				}
				continue;
			}
			if (btype == 3) { err = 4; break; }

			// ---- Huffman tables ----
			for (int i = gl; i < (1 << LIT_BITS); i += G) T.lit_lut[i] = 0;
			for (int i = gl; i < (1 << DIST_BITS); i += G) T.dist_lut[i] = 0;
			__builtin_amdgcn_wave_barrier();
			if (leader)
			{
				int nlit, ndist;
				uint8_t* L = T.lens + 20; // litlen code lengths, then dist code lengths
				if (btype == 1)
				{
					for (int i = 0; i < 144; ++i) L[i] = 8;
					for (int i = 144; i < 256; ++i) L[i] = 9;
					for (int i = 256; i < 280; ++i) L[i] = 7;
					for (int i = 280; i < 288; ++i) L[i] = 8;
					for (int i = 0; i < 30; ++i) L[288 + i] = 5;
					nlit = 288; ndist = 30;
				}
				else
				{
					nlit = (int)br.get(5) + 257; ndist = (int)br.get(5) + 1; int ncl = (int)br.get(4) + 4;
					if (nlit > 286 || ndist > 30) err = 5;
					uint8_t* cl = T.lens; // 19 code-length code lengths, built into dist_lut (7-bit table) temporarily
					for (int i = 0; i < 19; ++i) cl[i] = 0;
					for (int i = 0; i < ncl; ++i) cl[c_clorder[i]] = (uint8_t)br.get(3);
					build_tables(cl, 19, T.dist_cnt, T.dist_sym, T.dist_lut, 7, T.offs, T.next_code);
					int i = 0, n = nlit + ndist; uint32_t prev = 0;
					while (i < n && !err)
					{
						br.norm();
						uint32_t e = T.dist_lut[br.peek() & 127u];
						if (!(e & 15u)) { err = 6; break; }
						br.consume(e & 15u);
						uint32_t s = e >> 4;
						if (s < 16) { L[i++] = (uint8_t)s; prev = s; }
						else
						{
							uint32_t rep, val = 0;
							if (s == 16) { if (i == 0) { err = 7; break; } rep = 3 + br.get(2); val = prev; }
							else if (s == 17) { rep = 3 + br.get(3); prev = 0; }
							else { rep = 11 + br.get(7); prev = 0; }
							if (i + (int)rep > n) { err = 8; break; }
							for (uint32_t k = 0; k < rep; ++k) L[i++] = (uint8_t)val;
						}
						if (br.overrun()) err = 1;
					}
					if (!err && L[256] == 0) err = 15; // no end-of-block code
					for (int k = 0; k < (1 << DIST_BITS); ++k) T.dist_lut[k] = 0;
				}
				if (!err)
				{
					build_tables(L, nlit, T.lit_cnt, T.lit_sym, T.lit_lut, LIT_BITS, T.offs, T.next_code);
					build_tables(L + nlit, ndist, T.dist_cnt, T.dist_sym, T.dist_lut, DIST_BITS, T.offs, T.next_code);
				}
			}
			err = __shfl(err, leader_lane);
			if (err) break;
			__builtin_amdgcn_wave_barrier();

			// ---- symbols ----
			while (true)
			{
				int kind = 0; uint32_t mlen = 0, mdist = 0;
				if (leader)
				{
					int nlit = 0;
					while (true)
					{
						br.norm();
						uint32_t bits = br.peek();
						uint32_t e = T.lit_lut[bits & ((1u << LIT_BITS) - 1u)];
						if (!(e & 15u)) { e = slow_decode(bits, T.lit_cnt, T.lit_sym); if (!e) { err = 9; kind = 3; break; } }
						uint32_t l = e & 15u, s = e >> 4;
						if (s < 256)
						{
							br.consume(l);
							if (out_pos >= usize) { err = 3; kind = 3; break; }
							out[out_pos++] = (uint8_t)s;
							if (++nlit >= MAX_LIT_RUN) { kind = 0; break; }
							continue;
						}
						if (s == 256) { br.consume(l); kind = 2; break; }
						s -= 257;
						if (s >= 29) { err = 10; kind = 3; break; }
						uint32_t eb = c_lext[s];
						mlen = c_lbase[s] + ((bits >> l) & ((1u << eb) - 1u));
						br.consume(l + eb);
						br.norm();
						bits = br.peek();
						e = T.dist_lut[bits & ((1u << DIST_BITS) - 1u)];
						if (!(e & 15u)) { e = slow_decode(bits, T.dist_cnt, T.dist_sym); if (!e) { err = 11; kind = 3; break; } }
						l = e & 15u; s = e >> 4;
						if (s >= 30) { err = 12; kind = 3; break; }
						eb = c_dext[s];
						mdist = c_dbase[s] + ((bits >> l) & ((1u << eb) - 1u));
						br.consume(l + eb);
						if (mdist > out_pos || out_pos + mlen > usize) { err = 13; kind = 3; break; }
						if (br.overrun()) { err = 1; kind = 3; break; }
						kind = 1; break;
					}
				}
				kind = __shfl(kind, leader_lane);
				if (kind == 1)
				{
					mlen = __shfl(mlen, leader_lane); mdist = __shfl(mdist, leader_lane);
					uint32_t opos = __shfl(out_pos, leader_lane);
					uint8_t* dst = out + opos; const uint8_t* src = dst - mdist;
					if (mdist >= mlen)
					{
						for (uint32_t i = gl; i < mlen; i += G) dst[i] = src[i];
					}
					else
					{
						// overlapping match: out[i] = window[i mod dist]; every source byte precedes the match
						float rcp = __frcp_rn((float)mdist);
						for (uint32_t i = gl; i < mlen; i += G)
						{
							uint32_t q = (uint32_t)((float)i * rcp);
							int r = (int)i - (int)(q * mdist);
							if (r < 0) r += (int)mdist; else if (r >= (int)mdist) r -= (int)mdist;
							dst[i] = src[r];
						}
					}
					if (leader) out_pos += mlen;
				}
				else if (kind >= 2) break;
			}
			err = __shfl(err, leader_lane);
		}
		if (leader)
		{
			if (!err && out_pos != usize) err = 14;
			status[b].produced = out_pos; status[b].error = err;
		}
	}
}

void launch_inflate(const uint8_t* d_comp, const BlockDesc* d_blocks, int64_t n_blocks, uint8_t* d_out, BlockStatus* d_status, hipStream_t s)
{
	if (n_blocks <= 0) return;
	constexpr int G = 64;
	constexpr int GROUPS = 256 / G;
	int64_t wgs = (n_blocks + GROUPS - 1) / GROUPS;
	int64_t cap = 256 * 8 * 4; // enough workgroups to fill the chip several times over; grid-stride beyond
	int grid = (int)(wgs < cap ? wgs : cap);
	hipLaunchKernelGGL(bgzf_inflate_kernel<G>, dim3(grid), dim3(256), 0, s, d_comp, d_blocks, n_blocks, d_out, d_status);
}

} // namespace ngsqc
