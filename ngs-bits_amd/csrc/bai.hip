// BAI index (SAM spec §5.2): reading and region queries on the host. The reference reaches this through htslib's sam_index_load /
// sam_itr_queryi under BamReader::setRegion (src/cppNGS/BamReader.cpp:734-768): a region query touches only the BGZF blocks the index
// names. Here a query turns the regions into ONE virtual-offset range [beg, end) that holds every record overlapping any of them
// (ngsqc_bai_range); ngsqc_open_range then sends only the BGZF members of that range to the device.
#include "common.h"
#include <algorithm>
#include <cstring>
#include <fstream>

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
bool bai_range(const std::string& bam_path, const ngsqc_region* regions, int64_t n, int32_t n_ref, uint64_t& beg_voff, uint64_t& end_voff, bool& found)
{
	Bai bai; bool ok = load_bai(bam_path + ".bai", bai);
	if (!ok && bam_path.size() > 4 && bam_path.compare(bam_path.size() - 4, 4, ".bam") == 0) ok = load_bai(bam_path.substr(0, bam_path.size() - 4) + ".bai", bai);
	if (!ok) return false;
	beg_voff = ~0ull; end_voff = 0; found = false;
	std::vector<uint32_t> bins;
	for (int64_t i = 0; i < n; ++i)
	{
		const ngsqc_region& g = regions[i];
		if (g.tid < 0 || g.tid >= n_ref || (size_t)g.tid >= bai.refs.size()) continue;
		const BaiRef& R = bai.refs[(size_t)g.tid];
		const int64_t beg = std::max<int64_t>((int64_t)g.start - 1, 0), end = std::max<int64_t>(g.end, beg + 1);
		uint64_t min_off = 0;
		if (!R.ioffset.empty()) { const size_t w = (size_t)(beg >> 14); min_off = w < R.ioffset.size() ? R.ioffset[w] : R.ioffset.back(); }
		bins.clear(); reg2bins(beg, end, bins);
		std::sort(bins.begin(), bins.end());
		for (const auto& bc : R.bins)
		{
			if (bc.first == 37450u || !std::binary_search(bins.begin(), bins.end(), bc.first)) continue;   // (37450: the metadata pseudo-bin)
			for (const BaiChunk& c : bc.second)
				if (c.end > min_off) { beg_voff = std::min(beg_voff, c.beg); end_voff = std::max(end_voff, c.end); found = true; }
		}
	}
	return true;
}

} // namespace ngsqc
