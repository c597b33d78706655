// BAI / CSI index (SAM spec 5.2, hts-specs CSIv1): reading and region queries on the host. The reference reaches this through htslib's sam_index_load /
// sam_itr_queryi under BamReader::setRegion (src/cppNGS/BamReader.cpp:734-768; sam_index_load takes `<bam>.csi` before `<bam>.bai`): a region query
// touches only the BGZF blocks the index names. Here a query turns the regions into ONE virtual-offset range [beg, end) that holds every record
// overlapping any of them (ngsqc_bai_range); ngsqc_open_range then sends only the BGZF members of that range to the device.
// A CSI index is the same binning scheme with its own (min_shift, depth), `loff` per bin in place of the linear index, inside a BGZF container.
//
// Second half: WRITING the index (ngsqc_write_bai / ngsqc_write_csi). The reference has no call site for that - its tools expect the index that
// `samtools index` (htslib sam_index_build) left next to the BAM and fail with "Could not load index" (BamReader.cpp:742-746) without
// it. The per-record part (bin, windows, run boundaries, counts) runs on the device over the resident tile; the chunk rules
// of hts_idx_push / hts_idx_finish / compress_binning are applied to the runs on the host.
#include "common.h"
#include <algorithm>
#include <cstring>
#include <fstream>
#include <map>
#include <zlib.h>   // the BGZF container of a .csi FILE on the host (a few MB); BAM data never passes through it

namespace ngsqc {

namespace {
struct BaiChunk { uint64_t beg, end; };
struct IdxBin { uint32_t bin; uint64_t loff; std::vector<BaiChunk> chunks; };
struct BaiRef { std::vector<IdxBin> bins /* ascending */; std::vector<uint64_t> ioffset; };
struct Bai { int min_shift = 14, depth = 5; bool csi = false; std::vector<BaiRef> refs; };

uint32_t r32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t r64(const uint8_t* p) { return (uint64_t)r32(p) | ((uint64_t)r32(p + 4) << 32); }
inline uint32_t bin_first(int l) { return (uint32_t)(((1ull << (3 * l)) - 1ull) / 7ull); }
inline uint32_t n_bins_of(int depth) { return bin_first(depth + 1); }
inline int bin_level(uint32_t bin) { int l = 0; while (bin) { ++l; bin = (bin - 1u) >> 3; } return l; }

bool read_file(const std::string& path, std::vector<uint8_t>& d)
{
	std::ifstream f(path, std::ios::binary);
	if (!f) return false;
	d.assign((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	return true;
}
void sort_bins(Bai& out)
{
	for (BaiRef& R : out.refs) std::sort(R.bins.begin(), R.bins.end(), [](const IdxBin& a, const IdxBin& b) { return a.bin < b.bin; });
}

bool load_bai(const std::string& path, Bai& out)
{
	std::vector<uint8_t> d;
	if (!read_file(path, d)) return false;
	if (d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) return false;
	size_t o = 4; const size_t n = d.size();
	auto need = [&](size_t k) { if (o + k > n) throw std::runtime_error("truncated BAI index " + path); };
	need(4); const int32_t n_ref = (int32_t)r32(&d[o]); o += 4;
	if (n_ref < 0 || (size_t)n_ref > (n - o) / 8) return false;   // (every reference takes at least its two counts)
	out = Bai(); out.refs.resize((size_t)n_ref);
	for (int32_t r = 0; r < n_ref; ++r)
	{
		need(4); const int32_t n_bin = (int32_t)r32(&d[o]); o += 4;
		for (int32_t b = 0; b < n_bin; ++b)
		{
			need(8); const uint32_t bin = r32(&d[o]); const int32_t n_chunk = (int32_t)r32(&d[o + 4]); o += 8;
			need((size_t)std::max(n_chunk, 0) * 16);
			std::vector<BaiChunk> cs((size_t)std::max(n_chunk, 0));
			for (auto& c : cs) { c.beg = r64(&d[o]); c.end = r64(&d[o + 8]); o += 16; }
			out.refs[(size_t)r].bins.push_back(IdxBin{bin, 0, std::move(cs)});
		}
		need(4); const int32_t n_intv = (int32_t)r32(&d[o]); o += 4;
		need((size_t)std::max(n_intv, 0) * 8);
		out.refs[(size_t)r].ioffset.resize((size_t)std::max(n_intv, 0));
		for (auto& v : out.refs[(size_t)r].ioffset) { v = r64(&d[o]); o += 8; }
	}
	sort_bins(out);
	return true;
}

// the payload of a BGZF file (or the bytes themselves when the file is not gzip: htslib's reader accepts an uncompressed index as well)
bool bgzf_unpack(const std::vector<uint8_t>& d, std::vector<uint8_t>& out)
{
	if (d.size() < 2 || d[0] != 0x1f || d[1] != 0x8b) { out = d; return true; }
	size_t pos = 0;
	while (pos + 18 <= d.size())
	{
		if (d[pos] != 0x1f || d[pos + 1] != 0x8b || d[pos + 2] != 8 || !(d[pos + 3] & 4)) return false;
		const size_t xlen = (size_t)d[pos + 10] | ((size_t)d[pos + 11] << 8);
		size_t x = pos + 12, bsize = 0; const size_t xend = x + xlen;
		if (xend > d.size()) return false;
		while (x + 4 <= xend)
		{
			const size_t sl = (size_t)d[x + 2] | ((size_t)d[x + 3] << 8);
			if (d[x] == 'B' && d[x + 1] == 'C' && sl == 2 && x + 6 <= xend) bsize = ((size_t)d[x + 4] | ((size_t)d[x + 5] << 8)) + 1;
			x += 4 + sl;
		}
		if (bsize < xend - pos + 8 || pos + bsize > d.size()) return false;
		const uint32_t isize = r32(&d[pos + bsize - 4]);
		if (isize > 65536u) return false;
		const size_t at = out.size(); out.resize(at + isize);
		z_stream z; memset(&z, 0, sizeof z);
		if (inflateInit2(&z, -15) != Z_OK) return false;
		z.next_in = const_cast<Bytef*>(&d[xend]); z.avail_in = (uInt)(pos + bsize - 8 - xend);
		z.next_out = isize ? &out[at] : nullptr; z.avail_out = isize;
		Bytef none; if (!isize) z.next_out = &none;
		const int rc = inflate(&z, Z_FINISH); const bool ok = rc == Z_STREAM_END && z.total_out == isize;
		inflateEnd(&z);
		if (!ok || (uint32_t)crc32(crc32(0L, Z_NULL, 0), isize ? &out[at] : &none, isize) != r32(&d[pos + bsize - 8])) return false;
		pos += bsize;
	}
	return pos == d.size();
}

bool load_csi(const std::string& path, Bai& out)
{
	std::vector<uint8_t> f, d;
	if (!read_file(path, f)) return false;
	if (!bgzf_unpack(f, d)) throw std::runtime_error("damaged BGZF container of CSI index " + path);
	if (d.size() < 20 || memcmp(d.data(), "CSI\1", 4) != 0) return false;
	size_t o = 4; const size_t n = d.size();
	auto need = [&](size_t k) { if (o + k > n) throw std::runtime_error("truncated CSI index " + path); };
	const int32_t min_shift = (int32_t)r32(&d[4]), depth = (int32_t)r32(&d[8]), l_aux = (int32_t)r32(&d[12]); o = 16;
	if (min_shift < 0 || min_shift > 31 || depth < 0 || depth > 10 || min_shift + 3 * depth > 62 || l_aux < 0) return false;   // (bin numbers are 32 bit: depth <= 10)
	need((size_t)l_aux); o += (size_t)l_aux;
	need(4); const int32_t n_ref = (int32_t)r32(&d[o]); o += 4;
	if (n_ref < 0 || (size_t)n_ref > (n - o) / 4) return false;   // (every reference takes at least its bin count)
	out = Bai(); out.csi = true; out.min_shift = min_shift; out.depth = depth; out.refs.resize((size_t)n_ref);
	for (int32_t r = 0; r < n_ref; ++r)
	{
		need(4); const int32_t n_bin = (int32_t)r32(&d[o]); o += 4;
		for (int32_t b = 0; b < n_bin; ++b)
		{
			need(16); const uint32_t bin = r32(&d[o]); const uint64_t loff = r64(&d[o + 4]); const int32_t n_chunk = (int32_t)r32(&d[o + 12]); o += 16;
			need((size_t)std::max(n_chunk, 0) * 16);
			std::vector<BaiChunk> cs((size_t)std::max(n_chunk, 0));
			for (auto& c : cs) { c.beg = r64(&d[o]); c.end = r64(&d[o + 8]); o += 16; }
			out.refs[(size_t)r].bins.push_back(IdxBin{bin, loff, std::move(cs)});
		}
	}
	sort_bins(out);
	return true;
}

const IdxBin* find_bin(const BaiRef& R, uint32_t bin)
{
	auto it = std::lower_bound(R.bins.begin(), R.bins.end(), bin, [](const IdxBin& a, uint32_t b) { return a.bin < b; });
	return it != R.bins.end() && it->bin == bin ? &*it : nullptr;
}
} // namespace

// Virtual-offset range [beg_voff, end_voff) that contains every record overlapping any region (1-based, closed; like the iterator of
// BamReader::setRegion: chunks of the overlapping bins that end behind the index' lower bound). found = 0: no record can overlap.
// Returns false when there is no readable index next to the BAM: <bam>.csi, <bam without .bam>.csi, <bam>.bai, <bam without .bam>.bai in htslib's order
// (hts_idx_check_local).
static bool load_bai_of(const std::string& bam_path, Bai& bai)
{
	const bool has_ext = bam_path.size() > 4 && bam_path.compare(bam_path.size() - 4, 4, ".bam") == 0;
	const std::string stem = has_ext ? bam_path.substr(0, bam_path.size() - 4) : std::string();
	if (load_csi(bam_path + ".csi", bai)) return true;
	if (has_ext && load_csi(stem + ".csi", bai)) return true;
	if (load_bai(bam_path + ".bai", bai)) return true;
	return has_ext && load_bai(stem + ".bai", bai);
}
// the range of ONE region: false = no record can overlap it
static bool region_range(const Bai& bai, const ngsqc_region& g, int32_t n_ref, uint64_t& rb, uint64_t& re)
{
	if (g.tid < 0 || g.tid >= n_ref || (size_t)g.tid >= bai.refs.size()) return false;
	const BaiRef& R = bai.refs[(size_t)g.tid];
	const int shift = bai.min_shift, depth = bai.depth;
	const int64_t max_pos = 1ll << (shift + 3 * depth);
	const int64_t beg = std::max<int64_t>((int64_t)g.start - 1, 0), end = std::min(std::max<int64_t>(g.end, beg + 1), max_pos);
	if (beg >= max_pos) return false;   // (nothing behind the index' last position can be stored in it)
	uint64_t min_off = 0;
	if (bai.csi)
	{
		// hts_itr_query: loff of the bottom-level bin of the region's start; when that bin does not exist, of the sibling in front of it, else of the parent, ...
		uint32_t b = bin_first(depth) + (uint32_t)(beg >> shift); const IdxBin* hit = nullptr;
		while (b)
		{
			if ((hit = find_bin(R, b))) break;
			const uint32_t first = (((b - 1u) >> 3) << 3) + 1u;
			b = b > first ? b - 1u : (b - 1u) >> 3;
		}
		if (!hit) hit = find_bin(R, 0u);
		min_off = hit ? hit->loff : 0;
	}
	else if (!R.ioffset.empty()) { const size_t w = (size_t)(beg >> shift); min_off = w < R.ioffset.size() ? R.ioffset[w] : R.ioffset.back(); }
	// (the bins of the INDEX are tested against the region - the interval of a bin follows from its number - instead of listing the region's bins as reg2bins does:
	// an index with a tiny min_shift would make that list as long as the region)
	const uint32_t n_bins = n_bins_of(depth);
	rb = ~0ull; re = 0; uint64_t stop = ~0ull; bool any = false;
	for (const IdxBin& bc : R.bins)
	{
		if (bc.bin >= n_bins) continue;   // (n_bins + 1: the metadata pseudo-bin)
		const int l = bin_level(bc.bin);
		const int64_t bin_start = (int64_t)(bc.bin - bin_first(l)) << (shift + 3 * (depth - l)), bin_size = 1ll << (shift + 3 * (depth - l));
		if (bin_start < end && bin_start + bin_size > beg)
		{
			for (const BaiChunk& c : bc.chunks)
				if (c.end > min_off) { rb = std::min(rb, c.beg); re = std::max(re, c.end); any = true; }
			continue;
		}
		// A bin whose interval starts at or behind the region's end holds only records that start there, so its first chunk starts at such a record. The file is
		// sorted by start: every record that overlaps the region lies in front of that record - where the iterator of the reference stops, too (hts_itr_next:
		// "beg >= iter->end"). Without this bound the range runs to the last chunk of the region's super-bins.
		if (bin_start >= end) for (const BaiChunk& c : bc.chunks) stop = std::min(stop, c.beg);
	}
	if (!any) return false;
	rb = std::max(rb, min_off);             // every record that overlaps the region's first window starts at or behind the index' lower bound
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
// hts_reg2bin (SAM spec 5.3; BAI: min_shift 14, 5 levels - bins 4681.., 585.., 73.., 9.., 1.., 0). beg = -1, end = 0 (a read without reference) gives
// 4680 there, like htslib's arithmetic shifts.
__host__ __device__ inline uint32_t bai_reg2bin(int64_t beg, int64_t end, int shift, int levels)
{
	--end;
	int s = shift; int64_t t = (int64_t)(((1ull << (3 * levels)) - 1ull) / 7ull);
	for (int l = levels; l > 0; --l, s += 3, t -= 1ll << (3 * l))
		if (beg >> s == end >> s) return (uint32_t)(t + (beg >> s));
	return 0;
}

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// per record: (reference, bin) key and the 16 kb windows it overlaps; per reference the mapped / unmapped counts (hts_idx_push's n_mapped / n_unmapped)
__global__ __launch_bounds__(256) void bai_keys_kernel(const uint8_t* __restrict__ infl, const int64_t* __restrict__ recoff, int64_t n_rec, int32_t n_ref, int shift, int levels,
                                                       uint64_t* __restrict__ key, uint64_t* __restrict__ wnd, unsigned long long* __restrict__ counts, unsigned long long* __restrict__ flags)
{
	const int64_t max_pos = 1ll << (shift + 3 * levels);
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
		uint64_t wn = ~0ull;
		if (tid < 0) { tid = -1; beg = -1; end = 0; }
		else
		{
			if (beg < 0) beg = 0;
			if (end <= 0) end = 1;
			if (beg > max_pos || end > max_pos) { fl |= BAI_F_TOO_FAR; end = max_pos; if (beg >= end) beg = end - 1; }
			wn = (uint64_t)(uint32_t)(beg >> shift) | ((uint64_t)(uint32_t)((end - 1) >> shift) << 32);
		}
		key[i] = ((uint64_t)(uint32_t)tid << 32) | bai_reg2bin(beg, end, shift, levels);
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
                                                       const uint64_t* __restrict__ wnd, const int64_t* __restrict__ first, unsigned long long* __restrict__ lidx,
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
			const uint64_t pw = wnd[i - 1]; pb0 = (uint32_t)pw; pe0 = (uint32_t)(pw >> 32);
		}
		if (kp != k) { const unsigned long long at = atomicAdd(n_runs, 1ull); runs[at] = BaiRun{u, tid, (uint32_t)k, pos, 0u}; }
	}
	else { const unsigned long long at = atomicAdd(n_runs, 1ull); runs[at] = BaiRun{u, tid, (uint32_t)k, pos, 0u}; }
	if (i == n_rec - 1) { const unsigned long long at = atomicAdd(n_runs, 1ull); runs[at] = BaiRun{u, tid, (uint32_t)k, pos < 0 ? 0 : pos, 1u}; }
	if (tid >= 0)
	{
		const uint64_t w = wnd[i]; const uint32_t b0 = (uint32_t)w, e0 = (uint32_t)(w >> 32);
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

void launch_bai_keys(const uint8_t* infl, const int64_t* recoff, int64_t n_rec, int32_t n_ref, int min_shift, int depth, uint64_t* d_key, uint64_t* d_wnd, unsigned long long* d_counts,
                     unsigned long long* d_flags, hipStream_t s)
{
	if (n_rec <= 0) return;
	hipLaunchKernelGGL(bai_keys_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, s, infl, recoff, n_rec, n_ref, min_shift, depth, d_key, d_wnd, d_counts, d_flags); KCHECK();
}
void launch_bai_runs(const uint8_t* infl, const int64_t* recoff, int64_t n_rec, int64_t u_base, const uint64_t* d_key, const uint64_t* d_wnd, const int64_t* d_first,
                     unsigned long long* d_lidx, BaiRun* d_runs, unsigned long long* d_nruns, unsigned long long* d_flags, hipStream_t s)
{
	if (n_rec <= 0) return;
	hipLaunchKernelGGL(bai_runs_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, s, infl, recoff, n_rec, u_base, d_key, d_wnd, d_first, d_lidx, d_runs, d_nruns, d_flags); KCHECK();
}

// a byte string inside a BGZF container (SAM spec 4.1: members of at most 64 KiB and the empty EOF member), as hts_idx_save writes a .csi
static std::string bgzf_pack(const std::string& raw)
{
	std::string out;
	auto member = [&](const char* p, size_t n) {
		std::vector<uint8_t> z(compressBound((uLong)n) + 64);
		z_stream d; memset(&d, 0, sizeof d);
		if (deflateInit2(&d, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
		Bytef none = 0;
		d.next_in = n ? reinterpret_cast<Bytef*>(const_cast<char*>(p)) : &none; d.avail_in = (uInt)n; d.next_out = z.data(); d.avail_out = (uInt)z.size();
		const int rc = deflate(&d, Z_FINISH); const size_t zn = d.total_out; deflateEnd(&d);
		if (rc != Z_STREAM_END || zn + 26 > 65536) throw std::runtime_error("deflate of an index block failed");
		const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), n ? reinterpret_cast<const Bytef*>(p) : &none, (uInt)n), isize = (uint32_t)n, bsize = (uint32_t)(zn + 25);
		const uint8_t h[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 255u), (uint8_t)(bsize >> 8)};
		out.append(reinterpret_cast<const char*>(h), 18); out.append(reinterpret_cast<const char*>(z.data()), zn);
		char t[8]; memcpy(t, &crc, 4); memcpy(t + 4, &isize, 4); out.append(t, 8);
	};
	for (size_t o = 0; o < raw.size(); o += 0xff00) member(raw.data() + o, std::min<size_t>(0xff00, raw.size() - o));
	member(nullptr, 0);
	return out;
}

// hts_idx_push over the runs, hts_idx_finish, update_loff, compress_binning, hts_idx_save (hts.c; restated in oracle/bai_build.py, which is pinned on
// the reference's fixture indices; CSI: oracle/csi_build.py). Bins are written in ascending order (htslib writes them in the order of its hash table;
// readers do not care). csi: loff per bin in place of the linear index, the file inside a BGZF container.
std::string bai_assemble(const std::string& out_path, int32_t n_ref, uint64_t offset0, uint64_t final_off, const std::vector<BaiRunV>& runs, const std::vector<uint64_t>& lidx_in,
                         const std::vector<int64_t>& first, const std::vector<int64_t>& counts, bool csi, int min_shift, int depth)
{
	if (min_shift < 0 || min_shift > 31 || depth < 0 || depth > 10 || (!csi && (min_shift != 14 || depth != 5))) return "unsupported index geometry";
	const int BAI_LEVELS = depth; const uint32_t BAI_N_BINS = n_bins_of(depth), BAI_META_BIN = BAI_N_BINS + 1u;
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
	if (csi) { out.append("CSI\1", 4); w32((uint32_t)min_shift); w32((uint32_t)depth); w32(0u); }
	else out.append("BAI\1", 4);
	w32((uint32_t)n_ref);
	for (int32_t t = 0; t < n_ref; ++t)
	{
		// ---- linear index: length = last window a record touched + 1; windows nobody touched take the next touched one's offset (update_loff) ----
		const int64_t f0 = first[(size_t)t], f1 = first[(size_t)t + 1];
		int64_t n = 0;
		for (int64_t x = f1 - 1; x >= f0; --x) if (lidx_in[(size_t)x] != ~0ull) { n = x - f0 + 1; break; }
		std::vector<uint64_t> L(lidx_in.begin() + f0, lidx_in.begin() + f0 + n);
		for (int64_t x = n - 2; x >= 0; --x) if (L[(size_t)x] == ~0ull) L[(size_t)x] = L[(size_t)x + 1];
		std::map<uint32_t, Chunks>& B = bidx[(size_t)t];
		// ---- update_loff (before compress_binning, as in hts_idx_finish): the linear index at the bin's first window; 0 for the pseudo-bin and behind the last window ----
		std::map<uint32_t, uint64_t> loff;
		if (csi) for (const auto& kv : B)
		{
			uint64_t v = 0;
			if (kv.first < BAI_N_BINS) { const int l = bin_level(kv.first); const uint64_t bot = (uint64_t)(kv.first - bin_first(l)) << (3 * (depth - l)); if (bot < (uint64_t)n) v = L[(size_t)bot]; }
			loff[kv.first] = v;
		}
		// ---- compress_binning: a bin whose chunks span less than 64 KiB of the file moves into its parent (if that exists), deepest level first ----
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
			w32(kv.first); if (csi) w64(loff[kv.first]); w32((uint32_t)kv.second.size());
			for (const BaiChunk& c : kv.second) { w64(c.beg); w64(c.end); }
		}
		if (!csi) { w32((uint32_t)n); for (uint64_t v : L) w64(v); }
	}
	w64((uint64_t)(counts[(size_t)n_ref * 2] + counts[(size_t)n_ref * 2 + 1]));   // n_no_coor
	if (csi) out = bgzf_pack(out);
	std::ofstream f(out_path, std::ios::binary | std::ios::trunc);
	if (!f) return "cannot write " + out_path;
	f.write(out.data(), (std::streamsize)out.size());
	f.close();
	if (!f) return "cannot write " + out_path;
	return "";
}

} // namespace ngsqc
