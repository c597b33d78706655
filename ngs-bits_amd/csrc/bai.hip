// BAI index (SAM spec §5.2): reading and region queries on the host. The reference reaches this through htslib's sam_index_load /
// sam_itr_queryi under BamReader::setRegion (src/cppNGS/BamReader.cpp:734-768): a region query touches only the BGZF blocks the index
// names. Here a query turns the regions into ONE virtual-offset range [beg, end) that holds every record overlapping any of them
// (ngsqc_bai_range); ngsqc_open_range then sends only the BGZF members of that range to the device.
//
// Second half: WRITING the index (ngsqc_write_bai). The reference has no call site for that - its tools expect the `<bam>.bai` that
// `samtools index` (htslib sam_index_build) left next to the BAM and fail with "Could not load index" (BamReader.cpp:742-746) without
// it. The per-record part (bin, 16 kb windows, run boundaries, counts) runs on the device over the resident tile; the chunk rules
// of hts_idx_push / hts_idx_finish / compress_binning are applied to the runs on the host.
#include "common.h"
#include <algorithm>
#include <cstring>
#include <fstream>
#include <map>

namespace ngsqc {

namespace {
struct BaiChunk { uint64_t beg, end; };
struct BaiRef { std::vector<std::pair<uint32_t, std::vector<BaiChunk>>> bins; std::vector<uint64_t> ioffset; };
struct Bai { std::vector<BaiRef> refs; };

uint32_t r32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t r64(const uint8_t* p) { return (uint64_t)r32(p) | ((uint64_t)r32(p + 4) << 32); }

bool load_bai(const std::string& path, Bai& out)
{
	std::ifstream f(path, std::ios::binary);
	if (!f) return false;
	std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	if (d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) return false;
	size_t o = 4; const size_t n = d.size();
	auto need = [&](size_t k) { if (o + k > n) throw std::runtime_error("truncated BAI index " + path); };
	need(4); const int32_t n_ref = (int32_t)r32(&d[o]); o += 4;
	if (n_ref < 0) return false;
	out.refs.resize((size_t)n_ref);
	for (int32_t r = 0; r < n_ref; ++r)
	{
		need(4); const int32_t n_bin = (int32_t)r32(&d[o]); o += 4;
		for (int32_t b = 0; b < n_bin; ++b)
		{
			need(8); const uint32_t bin = r32(&d[o]); const int32_t n_chunk = (int32_t)r32(&d[o + 4]); o += 8;
			std::vector<BaiChunk> cs((size_t)std::max(n_chunk, 0));
			for (auto& c : cs) { need(16); c.beg = r64(&d[o]); c.end = r64(&d[o + 8]); o += 16; }
			out.refs[(size_t)r].bins.emplace_back(bin, std::move(cs));
		}
		need(4); const int32_t n_intv = (int32_t)r32(&d[o]); o += 4;
		out.refs[(size_t)r].ioffset.resize((size_t)std::max(n_intv, 0));
		for (auto& v : out.refs[(size_t)r].ioffset) { need(8); v = r64(&d[o]); o += 8; }
	}
	return true;
}

// bins that may hold records overlapping [beg, end) (0-based, half open): SAM spec §5.3 reg2bins
void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t>& bins)
{
	--end;
	bins.push_back(0);
	for (int64_t k = 1 + (beg >> 26); k <= 1 + (end >> 26); ++k) bins.push_back((uint32_t)k);
	for (int64_t k = 9 + (beg >> 23); k <= 9 + (end >> 23); ++k) bins.push_back((uint32_t)k);
	for (int64_t k = 73 + (beg >> 20); k <= 73 + (end >> 20); ++k) bins.push_back((uint32_t)k);
	for (int64_t k = 585 + (beg >> 17); k <= 585 + (end >> 17); ++k) bins.push_back((uint32_t)k);
	for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); ++k) bins.push_back((uint32_t)k);
}
} // namespace

// Virtual-offset range [beg_voff, end_voff) that contains every record overlapping any region (1-based, closed; like the iterator of
// BamReader::setRegion: chunks of the overlapping bins that end behind the linear index' lower bound). found = 0: no record can overlap.
// Returns false when there is no readable BAI next to the BAM (<bam>.bai or <bam without .bam>.bai).
static bool load_bai_of(const std::string& bam_path, Bai& bai)
{
	bool ok = load_bai(bam_path + ".bai", bai);
	if (!ok && bam_path.size() > 4 && bam_path.compare(bam_path.size() - 4, 4, ".bam") == 0) ok = load_bai(bam_path.substr(0, bam_path.size() - 4) + ".bai", bai);
	return ok;
}
// the range of ONE region: false = no record can overlap it
static bool region_range(const Bai& bai, const ngsqc_region& g, int32_t n_ref, uint64_t& rb, uint64_t& re)
{
	if (g.tid < 0 || g.tid >= n_ref || (size_t)g.tid >= bai.refs.size()) return false;
	const BaiRef& R = bai.refs[(size_t)g.tid];
	const int64_t beg = std::max<int64_t>((int64_t)g.start - 1, 0), end = std::max<int64_t>(g.end, beg + 1);
	uint64_t min_off = 0;
	if (!R.ioffset.empty()) { const size_t w = (size_t)(beg >> 14); min_off = w < R.ioffset.size() ? R.ioffset[w] : R.ioffset.back(); }
	std::vector<uint32_t> bins; reg2bins(beg, end, bins);
	std::sort(bins.begin(), bins.end());
	rb = ~0ull; re = 0; uint64_t stop = ~0ull; bool any = false;
	for (const auto& bc : R.bins)
	{
		if (bc.first >= 37449u) continue;   // (37450: the metadata pseudo-bin)
		if (std::binary_search(bins.begin(), bins.end(), bc.first))
		{
			for (const BaiChunk& c : bc.second)
				if (c.end > min_off) { rb = std::min(rb, c.beg); re = std::max(re, c.end); any = true; }
			continue;
		}
		// A bin whose interval starts at or behind the region's end holds only records that start there, so its first chunk starts at such a record. The file is
		// sorted by start: every record that overlaps the region lies in front of that record - where the iterator of the reference stops, too (hts_itr_next:
		// "beg >= iter->end"). Without this bound the range runs to the last chunk of the region's 8 Mb / 64 Mb super-bins.
		int l = 0; uint32_t first = 0;
		while (l < 5 && bc.first >= ((1u << (3 * (l + 1))) - 1u) / 7u) { ++l; first = ((1u << (3 * l)) - 1u) / 7u; }
		const int64_t bin_start = (int64_t)(bc.first - first) << (14 + 3 * (5 - l));
		if (bin_start >= end) for (const BaiChunk& c : bc.second) stop = std::min(stop, c.beg);
	}
	if (!any) return false;
	rb = std::max(rb, min_off);             // every record that overlaps the region's first window starts at or behind the linear index' offset
	if (stop != ~0ull && stop >= rb) re = std::min(re, stop);
	return re > rb;
}
bool bai_range(const std::string& bam_path, const ngsqc_region* regions, int64_t n, int32_t n_ref, uint64_t& beg_voff, uint64_t& end_voff, bool& found)
{
	Bai bai;
	if (!load_bai_of(bam_path, bai)) return false;
	beg_voff = ~0ull; end_voff = 0; found = false;
	for (int64_t i = 0; i < n; ++i)
	{
		uint64_t rb, re;
		if (!region_range(bai, regions[i], n_ref, rb, re)) continue;
		beg_voff = std::min(beg_voff, rb); end_voff = std::max(end_voff, re); found = true;
	}
	return true;
}
// the same per region (one load of the index): beg[i] / end[i], end[i] == 0 when no record can overlap region i. The host layer clusters scattered regions
// with these (one partial handle per cluster instead of one range from the first to the last region).
bool bai_ranges(const std::string& bam_path, const ngsqc_region* regions, int64_t n, int32_t n_ref, uint64_t* beg, uint64_t* end)
{
	Bai bai;
	if (!load_bai_of(bam_path, bai)) return false;
	for (int64_t i = 0; i < n; ++i) { uint64_t rb, re; if (region_range(bai, regions[i], n_ref, rb, re)) { beg[i] = rb; end[i] = re; } else { beg[i] = 0; end[i] = 0; } }
	return true;
}


// ------------------------------------------------------------------------------------------------------------ index construction
namespace {
constexpr int BAI_SHIFT = 14, BAI_LEVELS = 5;
constexpr uint32_t BAI_N_BINS = ((1u << (3 * BAI_LEVELS + 3)) - 1u) / 7u, BAI_META_BIN = BAI_N_BINS + 1u;   // 37449, 37450
constexpr int64_t BAI_MAX_POS = 1ll << (BAI_SHIFT + 3 * BAI_LEVELS);                                    // 2^29

// hts_reg2bin for min_shift 14 / 5 levels (SAM spec §5.3). beg = -1, end = 0 (a read without reference) gives 4680 like htslib's arithmetic shifts.
__host__ __device__ inline uint32_t bai_reg2bin(int64_t beg, int64_t end)
{
	--end;
	if (beg >> 14 == end >> 14) return (uint32_t)(4681 + (beg >> 14));
	if (beg >> 17 == end >> 17) return (uint32_t)(585 + (beg >> 17));
	if (beg >> 20 == end >> 20) return (uint32_t)(73 + (beg >> 20));
	if (beg >> 23 == end >> 23) return (uint32_t)(9 + (beg >> 23));
	if (beg >> 26 == end >> 26) return (uint32_t)(1 + (beg >> 26));
	return 0;
}

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// per record: (reference, bin) key and the 16 kb windows it overlaps; per reference the mapped / unmapped counts (hts_idx_push's n_mapped / n_unmapped)
__global__ __launch_bounds__(256) void bai_keys_kernel(const uint8_t* __restrict__ infl, const int64_t* __restrict__ recoff, int64_t n_rec, int32_t n_ref,
                                                       uint64_t* __restrict__ key, uint32_t* __restrict__ wnd, unsigned long long* __restrict__ counts, unsigned long long* __restrict__ flags)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const bool valid = i < n_rec;
	int32_t tid = -1; bool unmapped = true; uint32_t fl = 0;
	if (valid)
	{
		const uint8_t* p = infl + recoff[i];
		tid = (int32_t)ld32(p + 4); const int32_t pos = (int32_t)ld32(p + 8);
		const uint32_t w = ld32(p + 12), w2 = ld32(p + 16);
		const uint32_t l_name = w & 0xffu, n_cigar = w2 & 0xffffu, flag = w2 >> 16;
		unmapped = (flag & 4u) != 0;
		int64_t rlen = 0;
		if (!unmapped)   // bam_endpos: the reference length of the CIGAR (M, D, N, =, X), 1 if there is none
		{
			const uint8_t* c = p + 36 + l_name;
			for (uint32_t k = 0; k < n_cigar; ++k)
			{
				const uint32_t op = ld32(c + 4 * k);
				if ((0x18du >> (op & 15u)) & 1u) rlen += op >> 4;   // ops 0, 2, 3, 7, 8
			}
		}
		int64_t beg = pos, end = (int64_t)pos + (rlen ? rlen : 1);
		if (tid < -1 || tid >= n_ref) { fl |= BAI_F_BAD_TID; tid = -1; }
		uint32_t wn = 0xffffffffu;
		if (tid < 0) { tid = -1; beg = -1; end = 0; }
		else
		{
			if (beg < 0) beg = 0;
			if (end <= 0) end = 1;
			if (beg > BAI_MAX_POS || end > BAI_MAX_POS) { fl |= BAI_F_TOO_FAR; end = BAI_MAX_POS; if (beg >= end) beg = end - 1; }
			wn = (uint32_t)(beg >> BAI_SHIFT) | ((uint32_t)((end - 1) >> BAI_SHIFT) << 16);
		}
		key[i] = ((uint64_t)(uint32_t)tid << 32) | bai_reg2bin(beg, end);
		wnd[i] = wn;
	}
	// counts: one atomic per wave and kind while the wave stays on one reference (a sorted file: nearly always)
	const uint64_t act = __ballot(valid);
	if (act)
	{
		const int32_t t0 = __builtin_amdgcn_readfirstlane(tid);
		if (__ballot(valid && tid != t0) == 0)
		{
			const uint64_t um = __ballot(valid && unmapped);
			const int64_t slot = (int64_t)(t0 < 0 ? n_ref : t0) * 2;
			if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(act))
			{
				if (act & ~um) atomicAdd(&counts[slot], (unsigned long long)__popcll(act & ~um));
				if (um) atomicAdd(&counts[slot + 1], (unsigned long long)__popcll(um));
			}
		}
		else if (valid) atomicAdd(&counts[(int64_t)(tid < 0 ? n_ref : tid) * 2 + (unmapped ? 1 : 0)], 1ull);
	}
	if (fl) atomicOr(flags, (unsigned long long)fl);
}

// run boundaries (a record whose key differs from its predecessor's starts a run), the linear index (minimum start offset per window; a record
// skips the windows its predecessor already covers - that one starts earlier), the sort order inside the tile
__global__ __launch_bounds__(256) void bai_runs_kernel(const uint8_t* __restrict__ infl, const int64_t* __restrict__ recoff, int64_t n_rec, int64_t u_base, const uint64_t* __restrict__ key,
                                                       const uint32_t* __restrict__ wnd, const int64_t* __restrict__ first, unsigned long long* __restrict__ lidx,
                                                       BaiRun* __restrict__ runs, unsigned long long* __restrict__ n_runs, unsigned long long* __restrict__ flags)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_rec) return;
	const uint64_t k = key[i]; const int32_t tid = (int32_t)(k >> 32);
	const int64_t off = recoff[i]; const int64_t u = u_base + off;
	const int32_t pos = (int32_t)ld32(infl + off + 8);
	bool same_ref = false; uint32_t pb0 = 1, pe0 = 0;
	if (i > 0)
	{
		const uint64_t kp = key[i - 1];
		same_ref = (int32_t)(kp >> 32) == tid;
		if (same_ref && tid >= 0)
		{
			const int32_t pp = (int32_t)ld32(infl + recoff[i - 1] + 8);
			if ((pp < 0 ? 0 : pp) > pos) atomicOr(flags, (unsigned long long)BAI_F_UNSORTED);   // hts_idx_push: "unsorted positions"
			const uint32_t pw = wnd[i - 1]; pb0 = pw & 0xffffu; pe0 = pw >> 16;
		}
		if (kp != k) { const unsigned long long at = atomicAdd(n_runs, 1ull); runs[at] = BaiRun{u, tid, (uint32_t)k, pos, 0u}; }
	}
	else { const unsigned long long at = atomicAdd(n_runs, 1ull); runs[at] = BaiRun{u, tid, (uint32_t)k, pos, 0u}; }
	if (i == n_rec - 1) { const unsigned long long at = atomicAdd(n_runs, 1ull); runs[at] = BaiRun{u, tid, (uint32_t)k, pos < 0 ? 0 : pos, 1u}; }
	if (tid >= 0)
	{
		const uint32_t w = wnd[i], b0 = w & 0xffffu, e0 = w >> 16;
		const int64_t f0 = first[tid], cap = first[tid + 1] - f0;
		for (uint32_t x = b0; x <= e0; ++x)
		{
			if (x >= pb0 && x <= pe0) continue;
			if ((int64_t)x >= cap) { atomicOr(flags, (unsigned long long)BAI_F_WINDOWS); break; }
			atomicMin(&lidx[f0 + x], (unsigned long long)u);
		}
	}
}
} // namespace

void launch_bai_keys(const uint8_t* infl, const int64_t* recoff, int64_t n_rec, int32_t n_ref, uint64_t* d_key, uint32_t* d_wnd, unsigned long long* d_counts, unsigned long long* d_flags, hipStream_t s)
{
	if (n_rec <= 0) return;
	hipLaunchKernelGGL(bai_keys_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, s, infl, recoff, n_rec, n_ref, d_key, d_wnd, d_counts, d_flags); KCHECK();
}
void launch_bai_runs(const uint8_t* infl, const int64_t* recoff, int64_t n_rec, int64_t u_base, const uint64_t* d_key, const uint32_t* d_wnd, const int64_t* d_first,
                     unsigned long long* d_lidx, BaiRun* d_runs, unsigned long long* d_nruns, unsigned long long* d_flags, hipStream_t s)
{
	if (n_rec <= 0) return;
	hipLaunchKernelGGL(bai_runs_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, s, infl, recoff, n_rec, u_base, d_key, d_wnd, d_first, d_lidx, d_runs, d_nruns, d_flags); KCHECK();
}

// hts_idx_push over the runs, hts_idx_finish, update_loff, compress_binning, hts_idx_save (hts.c; restated in oracle/bai_build.py, which is pinned on
// the reference's fixture indices). Bins are written in ascending order (htslib writes them in the order of its hash table; readers do not care).
std::string bai_assemble(const std::string& out_path, int32_t n_ref, uint64_t offset0, uint64_t final_off, const std::vector<BaiRunV>& runs, const std::vector<uint64_t>& lidx_in,
                         const std::vector<int64_t>& first, const std::vector<int64_t>& counts)
{
	typedef std::vector<BaiChunk> Chunks;
	std::vector<std::map<uint32_t, Chunks>> bidx((size_t)n_ref);
	std::vector<char> has((size_t)n_ref, 0);
	bool have = false, no_coor = false; int32_t save_tid = 0, tail_tid = -2, tail_pos = 0; uint32_t save_bin = 0; uint64_t save_off = offset0, off_beg = offset0;
	auto insert = [&](int32_t tid, uint32_t bin, uint64_t a, uint64_t b) { bidx[(size_t)tid][bin].push_back(BaiChunk{a, b}); };
	auto meta = [&](int32_t tid, uint64_t upto) {
		insert(tid, BAI_META_BIN, off_beg, upto); insert(tid, BAI_META_BIN, (uint64_t)counts[(size_t)tid * 2], (uint64_t)counts[(size_t)tid * 2 + 1]); off_beg = upto;
	};
	for (const BaiRunV& r : runs)
	{
		if (r.tid < -1 || r.tid >= n_ref) return "a record names a reference sequence that is not in the BAM header";   // (caller-supplied runs: ngsqc_bai_assemble)
		if (r.kind == 1) { tail_tid = r.tid; tail_pos = r.pos; continue; }
		if (tail_tid == r.tid && r.tid >= 0 && tail_pos > r.pos) return "unsorted positions: the BAM is not sorted by coordinate (a BAI index needs that)";
		tail_tid = -2;
		if (have && r.tid == save_tid && r.bin == save_bin) continue;   // (the first record of a tile continues the run of the previous tile)
		const bool new_ref = !have || r.tid != save_tid;
		if (new_ref)
		{
			if (r.tid >= 0 && no_coor) return "reads without a reference are not in a single block at the end of the BAM";
			if (r.tid >= 0 && has[(size_t)r.tid]) return "the records of a reference are not continuous: the BAM is not sorted by coordinate";
		}
		if (have && save_tid >= 0)
		{
			insert(save_tid, save_bin, save_off, r.voff);
			if (new_ref) meta(save_tid, r.voff);
		}
		if (r.tid >= 0) has[(size_t)r.tid] = 1; else no_coor = true;
		have = true; save_tid = r.tid; save_bin = r.bin; save_off = r.voff;
	}
	if (have && save_tid >= 0) { insert(save_tid, save_bin, save_off, final_off); meta(save_tid, final_off); }

	std::string out; out.reserve(1 << 20);
	auto w32 = [&](uint32_t v) { char b[4]; memcpy(b, &v, 4); out.append(b, 4); };
	auto w64 = [&](uint64_t v) { char b[8]; memcpy(b, &v, 8); out.append(b, 8); };
	out.append("BAI\1", 4); w32((uint32_t)n_ref);
	for (int32_t t = 0; t < n_ref; ++t)
	{
		// ---- linear index: length = last window a record touched + 1; windows nobody touched take the next touched one's offset (update_loff) ----
		const int64_t f0 = first[(size_t)t], f1 = first[(size_t)t + 1];
		int64_t n = 0;
		for (int64_t x = f1 - 1; x >= f0; --x) if (lidx_in[(size_t)x] != ~0ull) { n = x - f0 + 1; break; }
		std::vector<uint64_t> L(lidx_in.begin() + f0, lidx_in.begin() + f0 + n);
		for (int64_t x = n - 2; x >= 0; --x) if (L[(size_t)x] == ~0ull) L[(size_t)x] = L[(size_t)x + 1];
		// ---- compress_binning: a bin whose chunks span less than 64 KiB of the file moves into its parent (if that exists), deepest level first ----
		std::map<uint32_t, Chunks>& B = bidx[(size_t)t];
		auto by_start = [](const BaiChunk& a, const BaiChunk& b) { return a.beg < b.beg; };
		for (int l = BAI_LEVELS; l > 0; --l)
		{
			const uint32_t start = ((1u << (3 * l)) - 1u) / 7u;
			for (auto it = B.lower_bound(start); it != B.end() && it->first < BAI_N_BINS;)
			{
				Chunks& p = it->second;
				if (l < BAI_LEVELS && p.size() > 1) std::sort(p.begin(), p.end(), by_start);
				auto par = B.find((it->first - 1u) >> 3);
				if ((p.back().end >> 16) - (p.front().beg >> 16) < 0x10000ull && par != B.end())
				{
					par->second.insert(par->second.end(), p.begin(), p.end());
					it = B.erase(it);
				}
				else ++it;
			}
		}
		{ auto z = B.find(0u); if (z != B.end()) std::sort(z->second.begin(), z->second.end(), by_start); }
		// chunks of a bin that start in the BGZF block the previous one ends in are merged
		for (auto& kv : B)
		{
			if (kv.first >= BAI_N_BINS) continue;
			Chunks& p = kv.second; size_t m = 0;
			for (size_t l = 1; l < p.size(); ++l)
			{
				if (p[m].end >> 16 >= p[l].beg >> 16) { if (p[m].end < p[l].end) p[m].end = p[l].end; }
				else p[++m] = p[l];
			}
			p.resize(m + 1);
		}
		w32((uint32_t)B.size());
		for (const auto& kv : B)
		{
			w32(kv.first); w32((uint32_t)kv.second.size());
			for (const BaiChunk& c : kv.second) { w64(c.beg); w64(c.end); }
		}
		w32((uint32_t)n);
		for (uint64_t v : L) w64(v);
	}
	w64((uint64_t)(counts[(size_t)n_ref * 2] + counts[(size_t)n_ref * 2 + 1]));   // n_no_coor
	std::ofstream f(out_path, std::ios::binary | std::ios::trunc);
	if (!f) return "cannot write " + out_path;
	f.write(out.data(), (std::streamsize)out.size());
	f.close();
	if (!f) return "cannot write " + out_path;
	return "";
}

} // namespace ngsqc
