// CRC32 of every inflated BGZF member against the value in its gzip trailer. htslib verifies it for every block it inflates
// (bgzf.c, under sam_read1 / sam_itr_next: BamReader::getNextAlignment, src/cppNGS/BamReader.h:386-392, fails with "Could not
// read next alignment"), so a payload that is damaged but still a valid DEFLATE stream of the right length must not pass.
//
// ONE WAVE PER MEMBER, coalesced: the member is (virtually) left-padded with zero bytes to a multiple of 4 KiB - leading
// zeros do not change a CRC state of zero - and read in 4 KiB rounds in which lane l takes the 64 bytes at l * 64. A lane
// therefore checksums a strided sub-message; CRC is linear over GF(2), so between its pieces the lane advances its state over
// the 4032 bytes it skips (a constant linear map: four 256-entry tables, like the slice-by-4 step itself), and at the end
//   crc(M) = XOR_l x^(8 * 64 * (63 - l)) * s_l  +  x^(8 n) * 0xFFFFFFFF  +  0xFFFFFFFF        (products mod the CRC polynomial)
// with one branch-free 32-step GF(2) multiplication per lane and member. Integer work: ~0.05 wave instructions per byte.
//
// Round 5: K CHAINS PER LANE. The slice step is a chain of dependent LDS lookups (state -> four table reads -> next state), and profiles/r05_sq_*counters.txt show
// the kernel waiting on it: waves wait 89 % of their cycles while the LDS array is busy 19 % of the time. A round is therefore K x 4 KiB and lane l walks K pieces
// of it (virtual lanes l, l + 64, ..): K independent chains whose lookups and loads are in flight together. The K states of a lane fold into one with the constant
// map "advance over 4 KiB of zero bytes" (S = adv(adv(adv(s0) ^ s1) ^ s2) ^ s3), so the end of a member costs one GF(2) multiplication per lane as before.
#include "common.h"
#include <mutex>

namespace ngsqc {

namespace {
constexpr uint32_t CRC_POLY = 0xEDB88320u;   // reflected CRC-32 (gzip)
constexpr int CRC_ROUND = 4096, CRC_PIECE = 64;
constexpr int TAB_SLICE = 0, TAB_GAP = 1024, TAB_LANE = 2048, TAB_INIT = 2048 + 64, TAB_GAP2 = TAB_INIT + 65537, TAB_GAP4 = TAB_GAP2 + 1024, TAB_ADV = TAB_GAP4 + 1024, TAB_TOTAL = TAB_ADV + 1024;
// TAB_GAP / TAB_GAP2 / TAB_GAP4: a state advanced over the zero bytes between a chain's pieces in rounds of 4 / 8 / 16 KiB; TAB_ADV: over 4 KiB (folds a lane's chains)

__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b)   // a * b mod P, reflected representation (x^0 = 0x80000000)
{
	uint32_t p = 0;
	#pragma unroll
	for (int i = 31; i >= 0; --i)
	{
		p ^= (a >> i) & 1u ? b : 0u;
		b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
	}
	return p;
}

__global__ __launch_bounds__(256) void crc32_kernel(const BlockDesc* __restrict__ blocks, int64_t n_blocks, const uint8_t* __restrict__ out_base,
                                                    const uint32_t* __restrict__ expected, BlockStatus* __restrict__ status, const uint32_t* __restrict__ tabs)
{
	__shared__ uint32_t T[2048];   // [0,1024) slice-by-4 tables T3..T0 order by byte position, [1024,2048) "skip 4032 zero bytes" tables
	for (int i = threadIdx.x; i < 2048; i += 256) T[i] = tabs[i];
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const uint32_t kl = tabs[TAB_LANE + lane];
	const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
	for (int64_t b = wave; b < n_blocks; b += n_waves)
	{
		if (status[b].error) continue;
		const BlockDesc bd = blocks[b];
		const int n = (int)bd.usize;
		const uint8_t* p = out_base + bd.upos;
		const int rounds = (n + CRC_ROUND - 1) / CRC_ROUND, pad = rounds * CRC_ROUND - n;
		uint32_t s = 0;
		for (int r = 0; r < rounds; ++r)
		{
			if (r) s = T[TAB_GAP + (s & 255u)] ^ T[TAB_GAP + 256 + ((s >> 8) & 255u)] ^ T[TAB_GAP + 512 + ((s >> 16) & 255u)] ^ T[TAB_GAP + 768 + (s >> 24)];
			const int base = r * CRC_ROUND + lane * CRC_PIECE - pad;   // offset of the lane's piece inside the member (negative: inside the padding)
			if (base <= -CRC_PIECE) continue;                           // all padding (first round only, where s == 0 stays 0)
			#pragma unroll
			for (int q = 0; q < CRC_PIECE / 16; ++q)
			{
				uint32_t w[4];
				const int o = base + 16 * q;
				if (o >= 0) __builtin_memcpy(w, p + o, 16);
				else
				{
					#pragma unroll
					for (int k = 0; k < 4; ++k)
					{
						w[k] = 0;
						#pragma unroll
						for (int j = 0; j < 4; ++j) { const int oo = o + 4 * k + j; if (oo >= 0) w[k] |= (uint32_t)p[oo] << (8 * j); }
					}
				}
				#pragma unroll
				for (int k = 0; k < 4; ++k)
				{
					s ^= w[k];
					s = T[TAB_SLICE + 768 + (s & 255u)] ^ T[TAB_SLICE + 512 + ((s >> 8) & 255u)] ^ T[TAB_SLICE + 256 + ((s >> 16) & 255u)] ^ T[TAB_SLICE + (s >> 24)];
				}
			}
		}
		uint32_t v = gf_mul(kl, s);
		#pragma unroll
		for (int o = 32; o > 0; o >>= 1) v ^= (uint32_t)__shfl_xor((int)v, o);
		if (lane == 0)
		{
			const uint32_t crc = v ^ tabs[TAB_INIT + n] ^ 0xFFFFFFFFu;
			if (crc != expected[b]) status[b].error = K1_ERR_CRC;
		}
	}
}

// out = v << (8 * sh) where v is a little-endian 128-bit number in four words and sh = 0 .. 16 bytes (16: nothing is left)
__device__ __forceinline__ void shl128_bytes(const uint32_t (&v)[4], int sh, uint32_t* out)
{
	const int ws = sh >> 2, bs = (sh & 3) * 8;
	uint32_t t[5];   // t[i] = word (i - 1 - ws) of v, zero outside
	#pragma unroll
	for (int i = 0; i < 5; ++i)
	{
		uint32_t x = 0;
		#pragma unroll
		for (int c = 0; c < 4; ++c) if (i - 1 - c >= 0 && i - 1 - c < 4) x = ws == c ? v[i - 1 - c] : x;
		t[i] = x;
	}
	#pragma unroll
	for (int k = 0; k < 4; ++k) out[k] = (uint32_t)((((uint64_t)t[k + 1] << 32) | t[k]) >> (32 - bs));
}

template <int K>
__global__ __launch_bounds__(256) void crc32_chains_kernel(const BlockDesc* __restrict__ blocks, int64_t n_blocks, const uint8_t* __restrict__ out_base,
                                                           const uint32_t* __restrict__ expected, BlockStatus* __restrict__ status, const uint32_t* __restrict__ tabs)
{
	constexpr int ROUND = K * CRC_ROUND, GAP_AT = K == 4 ? TAB_GAP4 : TAB_GAP2;
	__shared__ uint32_t T[2048];   // [0,1024) the slice-by-4 tables, [1024,2048) "skip ROUND - 64 zero bytes"
	for (int i = threadIdx.x; i < 2048; i += 256) T[i] = i < 1024 ? tabs[TAB_SLICE + i] : tabs[GAP_AT + i - 1024];
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const uint32_t kl = tabs[TAB_LANE + lane];
	const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
	for (int64_t b = wave; b < n_blocks; b += n_waves)
	{
		if (status[b].error) continue;
		const BlockDesc bd = blocks[b];
		const int n = (int)bd.usize;
		const uint8_t* p = out_base + bd.upos;
		const int rounds = (n + ROUND - 1) / ROUND, pad = rounds * ROUND - n;
		uint32_t s[K];
		#pragma unroll
		for (int j = 0; j < K; ++j) s[j] = 0;
		for (int r = 0; r < rounds; ++r)
		{
			uint32_t w[K][16];
			const int base0 = r * ROUND + lane * CRC_PIECE - pad;   // the piece of chain 0 (chain j: + j * 4 KiB); negative = inside the padding
			if (r || !pad)   // every piece lies inside the member: K x 4 loads of 16 bytes per lane, all issued before the first table read
			{
				#pragma unroll
				for (int j = 0; j < K; ++j)
				{
					#pragma unroll
					for (int q = 0; q < 4; ++q) __builtin_memcpy(&w[j][4 * q], p + base0 + j * CRC_ROUND + 16 * q, 16);
				}
			}
			else   // the first round of a member whose size is not a multiple of the round: the bytes before the member read as zero (a zero state stays zero over them)
			{
				const int js = pad / CRC_ROUND;   // the chain that holds the member's first byte: the chains below it are padding, the ones above whole pieces
				#pragma unroll
				for (int j = 0; j < K; ++j)
				{
					if (j < js)
					{
						#pragma unroll
						for (int k = 0; k < 16; ++k) w[j][k] = 0;
					}
					else if (j > js)
					{
						#pragma unroll
						for (int q = 0; q < 4; ++q) __builtin_memcpy(&w[j][4 * q], p + base0 + j * CRC_ROUND + 16 * q, 16);
					}
					else if (n >= 16)
					{
						// a 16-byte group that begins before the member is the member's first 16 bytes moved up by the distance (as a little-endian number: shifted left);
						// one load per group for every lane, no loop over bytes (which was a chain of up to 15 dependent loads in one lane, once per member)
						#pragma unroll
						for (int q = 0; q < 4; ++q)
						{
							const int o = base0 + j * CRC_ROUND + 16 * q;
							uint32_t v[4];
							__builtin_memcpy(v, p + (o > 0 ? o : 0), 16);
							shl128_bytes(v, o >= 0 ? 0 : o <= -16 ? 16 : -o, &w[j][4 * q]);
						}
					}
					else
					{
						#pragma unroll
						for (int k = 0; k < 16; ++k)
						{
							uint32_t v = 0;
							#pragma unroll
							for (int i = 0; i < 4; ++i) { const int oo = base0 + j * CRC_ROUND + 4 * k + i; if (oo >= 0) v |= (uint32_t)p[oo] << (8 * i); }
							w[j][k] = v;
						}
					}
				}
			}
			if (r)
			{
				#pragma unroll
				for (int j = 0; j < K; ++j) s[j] = T[1024 + (s[j] & 255u)] ^ T[1024 + 256 + ((s[j] >> 8) & 255u)] ^ T[1024 + 512 + ((s[j] >> 16) & 255u)] ^ T[1024 + 768 + (s[j] >> 24)];
			}
			#pragma unroll
			for (int k = 0; k < 16; ++k)
			{
				uint32_t t[K][4];   // the K chains step together: 4 K lookups that do not depend on each other, all on their way before the first is used
				#pragma unroll
				for (int j = 0; j < K; ++j)
				{
					const uint32_t x = s[j] ^ w[j][k];
					t[j][0] = T[768 + (x & 255u)]; t[j][1] = T[512 + ((x >> 8) & 255u)]; t[j][2] = T[256 + ((x >> 16) & 255u)]; t[j][3] = T[x >> 24];
				}
				__builtin_amdgcn_sched_barrier(0);
				#pragma unroll
				for (int j = 0; j < K; ++j) s[j] = t[j][0] ^ t[j][1] ^ t[j][2] ^ t[j][3];
			}
		}
		uint32_t S = s[0];
		#pragma unroll
		for (int j = 1; j < K; ++j)   // (rare: the table stays in global memory)
			S = tabs[TAB_ADV + (S & 255u)] ^ tabs[TAB_ADV + 256 + ((S >> 8) & 255u)] ^ tabs[TAB_ADV + 512 + ((S >> 16) & 255u)] ^ tabs[TAB_ADV + 768 + (S >> 24)] ^ s[j];
		uint32_t v = gf_mul(kl, S);
		#pragma unroll
		for (int o = 32; o > 0; o >>= 1) v ^= (uint32_t)__shfl_xor((int)v, o);
		if (lane == 0)
		{
			const uint32_t crc = v ^ tabs[TAB_INIT + n] ^ 0xFFFFFFFFu;
			if (crc != expected[b]) status[b].error = K1_ERR_CRC;
		}
	}
}

// ---- host: constant tables, built once and uploaded once per device ----
uint32_t h_z1(const uint32_t* t0, uint32_t s) { return t0[s & 255u] ^ (s >> 8); }
uint32_t h_gf_mul(uint32_t a, uint32_t b)
{
	uint32_t p = 0;
	for (int i = 31; i >= 0; --i) { if ((a >> i) & 1u) p ^= b; b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u); }
	return p;
}
const std::vector<uint32_t>& host_tables()
{
	static std::vector<uint32_t> tab;
	static std::once_flag once;
	std::call_once(once, [] {
		tab.assign(TAB_TOTAL, 0u);
		uint32_t t0[256];
		for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1; t0[i] = c; }
		// slice tables: T[k][x] = state after k more zero bytes behind byte x
		for (uint32_t i = 0; i < 256; ++i)
		{
			uint32_t c = t0[i];
			for (int k = 0; k < 4; ++k) { tab[TAB_SLICE + 256 * k + i] = c; c = h_z1(t0, c); }
		}
		// skip tables: a state byte x at byte position k, advanced over CRC_ROUND - CRC_PIECE zero bytes
		for (int k = 0; k < 4; ++k)
			for (uint32_t i = 0; i < 256; ++i)
			{
				uint32_t c = i << (8 * k);
				for (int z = 0; z < CRC_ROUND - CRC_PIECE; ++z) c = h_z1(t0, c);
				tab[TAB_GAP + 256 * k + i] = c;
			}
		// x^(8 m) for m = 0 .. 65536 (advance x^0 over m zero bytes)
		std::vector<uint32_t> x8(65537); x8[0] = 0x80000000u;
		for (int m = 1; m <= 65536; ++m) x8[(size_t)m] = h_z1(t0, x8[(size_t)m - 1]);
		for (int l = 0; l < 64; ++l) tab[TAB_LANE + l] = x8[(size_t)(CRC_PIECE * (63 - l))];
		// the other skip tables: advancing a state over m zero bytes is the multiplication by x^(8 m)
		auto advance_tables = [&](int at, int zero_bytes) {
			for (int k = 0; k < 4; ++k) for (uint32_t i = 0; i < 256; ++i) tab[(size_t)at + 256 * k + i] = h_gf_mul(x8[(size_t)zero_bytes], i << (8 * k));
		};
		advance_tables(TAB_GAP2, 2 * CRC_ROUND - CRC_PIECE); advance_tables(TAB_GAP4, 4 * CRC_ROUND - CRC_PIECE); advance_tables(TAB_ADV, CRC_ROUND);
		for (int m = 0; m <= 65536; ++m) tab[TAB_INIT + m] = h_gf_mul(x8[(size_t)m], 0xFFFFFFFFu);
	});
	return tab;
}
const uint32_t* device_tables()
{
	static std::mutex mu; static uint32_t* d_tab[64] = {nullptr};
	int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) throw std::runtime_error("invalid HIP device for the CRC tables");
	std::lock_guard<std::mutex> g(mu);
	if (!d_tab[dev])
	{
		const std::vector<uint32_t>& t = host_tables();
		HIPCHK(hipMalloc((void**)&d_tab[dev], t.size() * sizeof(uint32_t)));
		HIPCHK(hipMemcpy(d_tab[dev], t.data(), t.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	return d_tab[dev];
}
} // namespace

void launch_crc32(const BlockDesc* d_blocks, int64_t n_blocks, const uint8_t* d_out, const uint32_t* d_expected, BlockStatus* d_status, hipStream_t s)
{
	if (n_blocks <= 0) return;
	const uint32_t* tabs = device_tables();
	const int64_t wgs = (n_blocks + 3) / 4;
	const char* e = getenv("NGSQC_CRC_CHAINS"); const int chains = e && (atoi(e) == 1 || atoi(e) == 2) ? atoi(e) : 4;   // chains per lane (1: the round-2 kernel); read per launch so that one process can compare them
	const dim3 grid((unsigned)(wgs < 32768 ? wgs : 32768)), wg(256);
	if (chains == 4) hipLaunchKernelGGL(crc32_chains_kernel<4>, grid, wg, 0, s, d_blocks, n_blocks, d_out, d_expected, d_status, tabs);
	else if (chains == 2) hipLaunchKernelGGL(crc32_chains_kernel<2>, grid, wg, 0, s, d_blocks, n_blocks, d_out, d_expected, d_status, tabs);
	else hipLaunchKernelGGL(crc32_kernel, grid, wg, 0, s, d_blocks, n_blocks, d_out, d_expected, d_status, tabs);
	KCHECK();
}

} // namespace ngsqc
