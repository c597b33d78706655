// K2 — BAM record index: find the start of every record in the inflated stream.
//
// A BAM record is {u32 block_size; block_size bytes} (SAM spec §4.2); the stream is a chain of them that ignores BGZF
// member boundaries. htslib-written files start a record at the beginning of (almost) every member, other writers do
// not. The chain is therefore resolved as: GUESS a first-record offset per member (offset 0 first, then a plausibility
// scan like a BAM split guesser) -> walk every member's sub-chain in parallel -> VERIFY on the host side of the library
// that every member's chain exit lands exactly on the next member's guessed start. Verification makes the result exact
// (no heuristic can mis-sync silently); a wrong guess only costs a repair round.
//
// Replaces the sequential record pull of BamReader::getNextAlignment (src/cppNGS/BamReader.h:386-398).
#include "common.h"
#include "k2_guess.h"
#include <algorithm>
#include <cstdlib>

namespace ngsqc {

// (entry_range and record_fields_fit: common.h, shared with the scan that rides the chain walk)
__device__ __forceinline__ bool record_fields_fit(const uint8_t* r, uint32_t bs)
{
	const uint32_t w = ld32u(r + 12), w2 = ld32u(r + 16);
	return record_fields_fit(w & 0xff, w2 & 0xffff, (int32_t)ld32u(r + 20), bs);
}

// Guessing the first record of an entry, one WAVE per entry: a lane takes 16 bytes and tests the four offsets inside its first dword (block_size and refID
// of each candidate are funnel shifts of the loaded words: 256 offsets per step, two rejected almost always before the full plausibility check), so an entry
// that lies inside one long record (ONT: most members) costs 256 steps, and the piece of a short-read member (its first record starts ~170 bytes in) one.
// Entries whose start is already known (>= 0 or -1) are skipped; an entry without any plausible start gets -1.
__global__ __launch_bounds__(256) void index_guess_kernel(const uint8_t* __restrict__ infl, int64_t total, const BlockDesc* __restrict__ blocks, int64_t n_blocks, int64_t prefix, int ksh, int64_t nm, int64_t from,
                                                          int32_t* start, int32_t n_ref)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
	for (int64_t b = from + wave; b < n_blocks; b += n_waves)
	{
		if (start[b] != -2) continue;
		int64_t lo, hi; entry_range(blocks, b, prefix, ksh, nm, lo, hi);
		int32_t found = -1;
		// the first 256 bytes alone (the piece of a short-read member: its first record starts ~170 bytes in), then 1 KiB per step with four loads in flight - an
		// entry inside one long record (ONT: most members) is scanned at four windows per memory round trip
		for (int64_t base = lo; base < hi && found < 0;)
		{
			const int nwin = base == lo ? 1 : 4;
			uint32_t w[4][8];
			#pragma unroll
			for (int q = 0; q < 4; ++q)
			{
				const int64_t o0 = base + 256 * q + 4 * lane;
				if (q < nwin && o0 < hi) load_window(infl, total, o0, w[q]); else for (int k = 0; k < 8; ++k) w[q][k] = 0u;
			}
			#pragma unroll
			for (int q = 0; q < 4; ++q)
			{
				if (q < nwin && found < 0)
				{
					const int64_t o0 = base + 256 * q + 4 * lane;
					const uint32_t cand = cheap_candidates(w[q], o0, hi, total, n_ref);   // bit t: offset o0 + t passes the cheap test
					if (__builtin_amdgcn_ballot_w64(cand != 0) != 0)
					{
						int32_t mine = -1;
						if (cand) for (int t = 0; t < 4 && mine < 0; ++t) if (((cand >> t) & 1u) && plausible_chain(infl, total, o0 + t, n_ref)) mine = t;
						const uint64_t m = __builtin_amdgcn_ballot_w64(mine >= 0);
						if (m) { const int l = __builtin_ctzll(m); found = (int32_t)(base + 256 * q + 4 * l + __builtin_amdgcn_readlane(mine, l) - lo); }
					}
				}
			}
			base += 256 * nwin;
		}
		if (lane == 0) start[b] = found;
	}
}

// The same search with a WORKGROUP per entry, for groups of members of a long-read file: an entry inside one long record is searched for kilobytes (16 KiB on average for
// 30 kb reads, the whole entry inside a 500 kb record), and with one wave that is a chain of memory round trips - the longest of them was the whole K2 stage
// (profiles/r05_scan_probe.txt: 1.14 ms per 150 k reads). Here the waves of the workgroup take interleaved 1 KiB stripes of a round and the round ends with the
// lowest hit of any wave: the same first plausible offset, found in a quarter (an eighth) of the round trips, for at most one round of bytes looked at in vain.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void index_guess_wide_kernel(const uint8_t* __restrict__ infl, int64_t total, const BlockDesc* __restrict__ blocks, int64_t n_blocks, int64_t prefix, int ksh, int64_t nm,
                                                                      int64_t from, int32_t* start, int32_t n_ref)
{
	__shared__ long long s_hit[2][WAVES];   // (two rows: a wave may be a round ahead of one that still reads the row before)
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (int64_t b = from + blockIdx.x; b < n_blocks; b += gridDim.x)
	{
		if (start[b] != -2) continue;   // (the same for every thread of the workgroup)
		int64_t lo, hi; entry_range(blocks, b, prefix, ksh, nm, lo, hi);
		long long found = -1; int row = 0;
		for (int64_t round = lo; round < hi && found < 0; round += 1024 * WAVES, row ^= 1)
		{
			const int64_t base = round + 1024 * wv;
			uint32_t w[4][8];
			#pragma unroll
			for (int q = 0; q < 4; ++q)
			{
				const int64_t o0 = base + 256 * q + 4 * lane;
				if (o0 < hi) load_window(infl, total, o0, w[q]); else for (int k = 0; k < 8; ++k) w[q][k] = 0u;
			}
			long long mine_abs = -1;
			#pragma unroll
			for (int q = 0; q < 4; ++q)
			{
				if (mine_abs < 0)
				{
					const int64_t o0 = base + 256 * q + 4 * lane;
					const uint32_t cand = cheap_candidates(w[q], o0, hi, total, n_ref);
					if (__builtin_amdgcn_ballot_w64(cand != 0) != 0)
					{
						int32_t mine = -1;
						if (cand) for (int t = 0; t < 4 && mine < 0; ++t) if (((cand >> t) & 1u) && plausible_chain(infl, total, o0 + t, n_ref)) mine = t;
						const uint64_t m = __builtin_amdgcn_ballot_w64(mine >= 0);
						if (m) { const int l = __builtin_ctzll(m); mine_abs = base + 256 * q + 4 * l + __builtin_amdgcn_readlane(mine, l); }
					}
				}
			}
			if (lane == 0) s_hit[row][wv] = mine_abs;
			__syncthreads();
			#pragma unroll
			for (int k = 0; k < WAVES; ++k) { const long long x = s_hit[row][k]; if (x >= 0 && (found < 0 || x < found)) found = x; }   // (stripes ascend with k: the first hit is the lowest)
		}
		if (threadIdx.x == 0) start[b] = found < 0 ? -1 : (int32_t)(found - lo);
		__syncthreads();   // (the next entry's first round writes row 0 again)
	}
}

// start[b]: >=0 first-record offset inside entry b; -1 none (a longer record covers the whole entry); -2 guess.
__global__ void index_count_kernel(const uint8_t* __restrict__ infl, int64_t total, const BlockDesc* __restrict__ blocks, int64_t n_blocks, int64_t prefix, int ksh, int64_t nm, int64_t from,
                                   int32_t* start, uint32_t* __restrict__ cnt, int64_t* __restrict__ next_abs,
                                   uint32_t* __restrict__ bad, int32_t n_ref, uint16_t* __restrict__ rel)
{
	int64_t b = from + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blocks) return;
	int64_t lo, hi; entry_range(blocks, b, prefix, ksh, nm, lo, hi);
	int32_t s = start[b];
	if (s == -2) { s = lane_guess(infl, total, lo, hi, n_ref); start[b] = s; }   // (fallback: the guess kernel resolves them wave by wave)
	if (s < 0) { cnt[b] = 0; next_abs[b] = -1; return; }
	// next_abs: >= 0 chain exit; -2 corrupt record; <= -10 a record starts at o = -(next_abs + 10) but extends past the end
	// of the resident tile (it is carried into the next tile, not counted here)
	const uint32_t stride = rel_stride(ksh);
	int64_t o = lo + s; uint32_t n = 0; int64_t res = 0; bool stop = false;
	while (o < hi)
	{
		if (o + 4 > total) { res = -(o + 10); stop = true; break; }
		uint32_t bs = ld32u(infl + o);
		if (bs < 32) { res = -2; stop = true; break; }
		if (o + 4 + (int64_t)bs > total) { res = -(o + 10); stop = true; break; }
		if (!record_fields_fit(infl + o, bs)) { res = -2; stop = true; break; }
		if (n < stride) rel[b * stride + n] = (uint16_t)(o - lo);   // (entry 0, the carried prefix, may exceed 16 bits: it is always walked again)
		++n; o += 4 + (int64_t)bs;
	}
	cnt[b] = n; next_abs[b] = stop ? res : o;
	if (stop && res == -2) atomicAdd(bad, 1u);
}

// start[] of every entry from what the host knows before K2: the tile-local offset exp0 of the first record (start = -2: guess). assume0 (the fast path): a
// member is taken to start with a record, as htslib writes them - no guess kernel for those entries; if it does not, the walker's predecessor ends elsewhere
// and the chain check sends the tile to the general path (which guesses).
__global__ void index_init_kernel(const BlockDesc* __restrict__ blocks, int64_t n_blocks, int64_t prefix, int ksh, int64_t nm, int64_t exp0, int guess_all, int assume0, int32_t* __restrict__ start)
{
	int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blocks) return;
	int64_t lo, hi; entry_range(blocks, b, prefix, ksh, nm, lo, hi);
	const bool first_piece = b > 0 && (ksh <= 0 || ((b - 1) & ((1ll << ksh) - 1)) == 0);   // (ksh < 0: an entry is a group of members - no shift by a negative count)
	start[b] = hi <= lo ? -1 : guess_all ? -2 : (hi <= exp0 ? -1 : (lo <= exp0 ? (int32_t)(exp0 - lo) : (assume0 && first_piece ? 0 : -2)));   // (an empty entry holds no record start)
}

// The exact check of the walked chain on the device. Entries in front of the tile's first record (hi <= exp0) hold nothing; the entry that holds exp0 starts
// there; every other start is a GUESS. The chain is right iff every walker leaves its entry exactly at the start of the next entry that has one, every entry
// without a start lies wholly in front of that exit, and the last walker leaves the tile exactly at its end - or meets a record that the tile end cuts (it is
// carried into the next tile: *straddle = its offset) with no start behind it. By induction from exp0 every start then lies on the true chain, and the host
// skips its sequential verification; viol counts the entries that do not fit (any: the tile takes the general path). For an htslib-written file (a record
// starts at every member's first byte, none straddles) and ksh = 0 this is round 3's "aligned" test.
__global__ void index_chain_kernel(const BlockDesc* __restrict__ blocks, int64_t n_blocks, int64_t prefix, int ksh, int64_t nm, int64_t exp0, int64_t total,
                                   const int32_t* __restrict__ start, const int64_t* __restrict__ next_abs, uint32_t* __restrict__ viol, long long* __restrict__ straddle)
{
	int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	bool bad = false;
	if (b < n_blocks)
	{
		int64_t lo, hi; entry_range(blocks, b, prefix, ksh, nm, lo, hi);
		const int32_t s = start[b];
		if (hi <= exp0) bad = s != -1;
		else if (lo <= exp0) bad = s != (int32_t)(exp0 - lo);
		if (!bad && s >= 0)
		{
			const int64_t nx = next_abs[b];
			if (nx == -2) bad = false;   // (a corrupt record: counted by the walk itself, the host throws)
			else if (nx == -3) bad = true;   // (an entry with more records than a name of the riding scan holds)
			else if (nx < 0)
			{
				// a record cut by the tile end: the rest of the tile belongs to it
				for (int64_t e = b + 1; e < n_blocks && !bad; ++e) bad = start[e] >= 0;
				if (!bad) *straddle = -(nx + 10);
			}
			else
			{
				int64_t e = b + 1;
				for (; e < n_blocks && start[e] < 0; ++e) { int64_t l2, h2; entry_range(blocks, e, prefix, ksh, nm, l2, h2); if (h2 > nx) { bad = true; break; } }
				if (!bad)
				{
					if (e < n_blocks) { int64_t l2, h2; entry_range(blocks, e, prefix, ksh, nm, l2, h2); bad = nx != l2 + start[e]; }
					else bad = nx != total;
				}
			}
		}
	}
	const unsigned long long m = __ballot(bad);
	if ((threadIdx.x & 63) == 0 && m) atomicAdd(viol, (uint32_t)__popcll(m));
}

// Record offsets of every entry, ONE WAVE PER ENTRY: the entry-relative offsets that the count pass stored are expanded with coalesced loads
// and stores (the chain is not walked a second time: that would read a third of the inflated tile again). Entry 0 (the carried prefix, offsets
// may exceed 16 bits) and entries with more records than their share of K2_REL_STRIDE walk their chain on lane 0.
__global__ __launch_bounds__(256) void index_write_kernel(const uint8_t* __restrict__ infl, int64_t total, const BlockDesc* __restrict__ blocks, int64_t n_blocks, int64_t prefix, int ksh, int64_t nm,
                                                          const int32_t* __restrict__ start, const uint32_t* __restrict__ cnt, const int64_t* __restrict__ base,
                                                          const uint16_t* __restrict__ rel, int64_t* __restrict__ recoff)
{
	const int lane = threadIdx.x & 63;
	const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (b >= n_blocks) return;
	const int32_t s = start[b];
	if (s < 0) return;
	const uint32_t n = cnt[b];
	const uint32_t stride = rel_stride(ksh);
	int64_t lo, hi; entry_range(blocks, b, prefix, ksh, nm, lo, hi);
	const int64_t k0 = base[b];
	if (b != 0 && n <= stride)   // (stride 0: a group of members - walked again)
	{
		for (uint32_t k = lane; k < n; k += 64) recoff[k0 + k] = lo + rel[b * stride + k];
		return;
	}
	if (lane != 0) return;
	int64_t o = lo + s; int64_t k = k0;
	while (o < hi)
	{
		if (o + 4 > total) break;
		int64_t nx = o + 4 + (int64_t)ld32u(infl + o);
		if (nx > total) break;   // straddles the tile end: belongs to the next tile
		recoff[k++] = o; o = nx;
	}
}

// ---- generic 3-kernel exclusive scans (u32 -> i64) and in-place inclusive (i32) ----
constexpr int SCAN_T = 256, SCAN_ITEMS = 16, SCAN_TILE = SCAN_T * SCAN_ITEMS;

template <typename TIn>
__global__ void scan_tile_sums(const TIn* __restrict__ in, int64_t n, int64_t* __restrict__ tile_sums)
{
	__shared__ int64_t sh[SCAN_T];
	int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
	int64_t acc = 0;
	for (int i = threadIdx.x; i < SCAN_TILE; i += SCAN_T) { int64_t j = base + i; if (j < n) acc += (int64_t)in[j]; }
	sh[threadIdx.x] = acc; __syncthreads();
	for (int s = SCAN_T / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
	if (threadIdx.x == 0) tile_sums[blockIdx.x] = sh[0];
}

__global__ void scan_tile_prefix(int64_t* tile_sums, int64_t n_tiles) // single workgroup, exclusive in place; total -> tile_sums[n_tiles]
{
	__shared__ int64_t sh[SCAN_T]; __shared__ int64_t carry;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (int64_t base = 0; base < n_tiles; base += SCAN_T)
	{
		int64_t j = base + threadIdx.x;
		int64_t v = j < n_tiles ? tile_sums[j] : 0;
		sh[threadIdx.x] = v; __syncthreads();
		for (int d = 1; d < SCAN_T; d <<= 1) { int64_t t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0; __syncthreads(); sh[threadIdx.x] += t; __syncthreads(); }
		int64_t incl = sh[threadIdx.x]; int64_t c = carry;
		if (j < n_tiles) tile_sums[j] = c + incl - v;
		__syncthreads();
		if (threadIdx.x == SCAN_T - 1) carry = c + incl;
		__syncthreads();
	}
	if (threadIdx.x == 0) tile_sums[n_tiles] = carry;
}

// each thread owns SCAN_ITEMS consecutive items of its tile
template <typename TIn, typename TOut, bool INCLUSIVE>
__global__ void scan_tile_apply(const TIn* __restrict__ in, int64_t n, const int64_t* __restrict__ tile_prefix, TOut* __restrict__ out)
{
	__shared__ int64_t sh[SCAN_T];
	int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
	int64_t v[SCAN_ITEMS]; int64_t acc = 0;
	#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; ++i) { int64_t j = base + i; v[i] = j < n ? (int64_t)in[j] : 0; acc += v[i]; }
	sh[threadIdx.x] = acc; __syncthreads();
	for (int d = 1; d < SCAN_T; d <<= 1) { int64_t t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0; __syncthreads(); sh[threadIdx.x] += t; __syncthreads(); }
	int64_t run = tile_prefix[blockIdx.x] + sh[threadIdx.x] - acc;
	#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; ++i)
	{
		int64_t j = base + i;
		if (INCLUSIVE) { run += v[i]; if (j < n) out[j] = (TOut)run; }
		else { if (j < n) out[j] = (TOut)run; run += v[i]; }
	}
}

size_t scan_tmp_bytes(int64_t n) { int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE; return (size_t)(tiles + 2) * sizeof(int64_t); }

// entries [from, n_entries) of the tile (entry 0 = carried prefix, entry e = piece (e - 1) & (2^ksh - 1) of member (e - 1) >> ksh of d_blocks); arrays are indexed by entry
void launch_index_count(const uint8_t* d_infl, int64_t total, const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, int64_t from, int32_t* d_start,
                        uint32_t* d_cnt, int64_t* d_next_abs, uint32_t* d_bad, int32_t n_ref, uint16_t* d_rel, hipStream_t s, bool wave_guess)
{
	const int64_t n = n_entries - from;
	if (n <= 0) return;
	// starts still to be guessed: by a wave per entry (entries that may lie inside one long record: 64 KiB to look through), or by each walker for itself (the
	// pieces of a short-read member: the first record is a few hundred bytes in)
	if (wave_guess) launch_index_guess(d_infl, total, d_blocks, n_entries, prefix, ksh, nm, from, d_start, n_ref, s);
	int grid = (int)((n + 63) / 64);
	hipLaunchKernelGGL(index_count_kernel, dim3(grid), dim3(64), 0, s, d_infl, total, d_blocks, n_entries, prefix, ksh, nm, from, d_start, d_cnt, d_next_abs, d_bad, n_ref, d_rel); KCHECK();
}

// resolve the guessed first-record offsets (start == -2) wave-cooperatively
void launch_index_guess(const uint8_t* d_infl, int64_t total, const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, int64_t from, int32_t* d_start, int32_t n_ref, hipStream_t s)
{
	const int64_t n = n_entries - from;
	if (n <= 0) return;
	// (Round 5 first tried sixteen waves per group of members, each looking through a contiguous sixteenth to its end: 6.6 ms against the wave-per-entry kernel's 1.6 ms
	// per tile of the ONT-like shard, profiles/r05_scan_probe.txt - removed. Interleaved stripes with a common end of the round are the form that is kept.)
	const char* e = getenv("NGSQC_GUESS_WAVES"); const int waves = e ? atoi(e) : (ksh <= -3 ? 8 : ksh < 0 ? 4 : 1);   // groups of members (long reads): a workgroup per entry - eight waves for groups of 8 and more (0.50 vs 0.69 ms at 16), four below (0.84 vs 1.06 ms at 4); 1: a wave per entry
	if (waves == 4 || waves == 8)
	{
		const dim3 grid((unsigned)(n < 65536 ? n : 65536));
		if (waves == 4) hipLaunchKernelGGL(index_guess_wide_kernel<4>, grid, dim3(256), 0, s, d_infl, total, d_blocks, n_entries, prefix, ksh, nm, from, d_start, n_ref);
		else hipLaunchKernelGGL(index_guess_wide_kernel<8>, grid, dim3(512), 0, s, d_infl, total, d_blocks, n_entries, prefix, ksh, nm, from, d_start, n_ref);
		KCHECK(); return;
	}
	const int64_t wg = (n + 3) / 4;
	hipLaunchKernelGGL(index_guess_kernel, dim3((int)(wg < 256 * 32 ? wg : 256 * 32)), dim3(256), 0, s, d_infl, total, d_blocks, n_entries, prefix, ksh, nm, from, d_start, n_ref); KCHECK();
}

void launch_index_init(const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, int64_t exp0, bool guess_all, bool assume0, int32_t* d_start, hipStream_t s)
{
	if (n_entries <= 0) return;
	hipLaunchKernelGGL(index_init_kernel, dim3((int)((n_entries + 255) / 256)), dim3(256), 0, s, d_blocks, n_entries, prefix, ksh, nm, exp0, guess_all ? 1 : 0, assume0 ? 1 : 0, d_start); KCHECK();
}
void launch_index_chain(const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, int64_t exp0, int64_t total, const int32_t* d_start, const int64_t* d_next, uint32_t* d_viol, long long* d_straddle, hipStream_t s)
{
	if (n_entries <= 0) return;
	hipLaunchKernelGGL(index_chain_kernel, dim3((int)((n_entries + 255) / 256)), dim3(256), 0, s, d_blocks, n_entries, prefix, ksh, nm, exp0, total, d_start, d_next, d_viol, d_straddle); KCHECK();
}

void launch_index_write(const uint8_t* d_infl, int64_t total, const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, const int32_t* d_start,
                        const uint32_t* d_cnt, const int64_t* d_base, const uint16_t* d_rel, int64_t* d_recoff, hipStream_t s)
{
	if (n_entries <= 0) return;
	int grid = (int)((n_entries + 3) / 4);   // one wave per entry
	hipLaunchKernelGGL(index_write_kernel, dim3(grid), dim3(256), 0, s, d_infl, total, d_blocks, n_entries, prefix, ksh, nm, d_start, d_cnt, d_base, d_rel, d_recoff); KCHECK();
}

// exclusive scan of u32 counts into int64 bases; d_base[n] receives the total. d_tmp needs scan_tmp_bytes(n).
void launch_scan_counts(const uint32_t* d_cnt, int64_t n, int64_t* d_base, void* d_tmp, hipStream_t s)
{
	int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
	int64_t* ts = (int64_t*)d_tmp;
	if (tiles > 0)
	{
		hipLaunchKernelGGL(scan_tile_sums<uint32_t>, dim3((int)tiles), dim3(SCAN_T), 0, s, d_cnt, n, ts); KCHECK();
		hipLaunchKernelGGL(scan_tile_prefix, dim3(1), dim3(SCAN_T), 0, s, ts, tiles); KCHECK();
		hipLaunchKernelGGL((scan_tile_apply<uint32_t, int64_t, false>), dim3((int)tiles), dim3(SCAN_T), 0, s, d_cnt, n, ts, d_base); KCHECK();
		HIPCHK(hipMemcpyAsync(d_base + n, ts + tiles, sizeof(int64_t), hipMemcpyDeviceToDevice, s));
	}
	else HIPCHK(hipMemsetAsync(d_base, 0, sizeof(int64_t), s));
}

// in-place inclusive prefix sum of the int32 difference array -> per-base depth
void launch_depth_prefix(int32_t* d_diff, int64_t n_slots, void* d_tmp, hipStream_t s)
{
	int64_t tiles = (n_slots + SCAN_TILE - 1) / SCAN_TILE;
	if (tiles <= 0) return;
	int64_t* ts = (int64_t*)d_tmp;
	hipLaunchKernelGGL(scan_tile_sums<int32_t>, dim3((int)tiles), dim3(SCAN_T), 0, s, d_diff, n_slots, ts); KCHECK();
	hipLaunchKernelGGL(scan_tile_prefix, dim3(1), dim3(SCAN_T), 0, s, ts, tiles); KCHECK();
	hipLaunchKernelGGL((scan_tile_apply<int32_t, int32_t, true>), dim3((int)tiles), dim3(SCAN_T), 0, s, d_diff, n_slots, ts, d_diff); KCHECK();
}

} // namespace ngsqc
