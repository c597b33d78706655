// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY (see bamio.hpp header).
// CPU restatement of the ROI containers the hot path uses:
//   Chromosome numbering        src/cppNGS/Chromosome.cpp:133-190, Chromosome.h:26-81
//   BedLine / BedFile           src/cppNGS/BedFile.cpp:29-35,119-202,234-237,252-325,519-567,587-644
//   ChromosomalIndex            src/cppNGS/ChromosomalIndex.h:53-135,172-202
//   Histogram (cppCORE, source absent from /root/reference: un-vendored submodule src/cppCORE,
//              .gitmodules:1-4) restated from its use at Statistics.cpp:401,407,631,707-712,1192
//              and pinned by the known-answer tests (see tests/test_oracle_golden.py).
// ============================================================================
#pragma once
#include <cmath>
#include <cstdint>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include "bamio.hpp"

namespace orc {

static inline std::string trim(const std::string& s)
{
	size_t a = 0, b = s.size();
	while (a<b && isspace((unsigned char)s[a])) ++a;
	while (b>a && isspace((unsigned char)s[b-1])) --b;
	return s.substr(a, b-a);
}

// Chromosome.cpp:133-190 — chr1..22 -> 1..22 (any integer 1..1000 without leading zero), X 1001, Y 1002, M/MT 1003,
// all others 1004+ in first-seen order (process-wide).
static inline int chr_num(const std::string& raw)
{
	std::string s = trim(raw);
	std::string t; for (char c : s) t.push_back((char)toupper((unsigned char)c));
	if (t.rfind("CHR", 0)==0) t = t.substr(3);
	if (t=="M") t = "MT";
	if (t.empty()) return 0;
	if (t=="X") return 1001;
	if (t=="Y") return 1002;
	if (t=="MT") return 1003;
	if (t[0]!='0')
	{
		bool digits = !t.empty() && t.size()<=9;
		for (char c : t) if (!isdigit((unsigned char)c)) digits = false;
		if (digits) { long v = atol(t.c_str()); if (v>0 && v<=1000) return (int)v; }
	}
	static std::mutex m; static std::unordered_map<std::string,int> cache; static int next_num = 1004;
	std::lock_guard<std::mutex> g(m);
	auto it = cache.find(t);
	if (it==cache.end()) it = cache.emplace(t, next_num++).first;
	return it->second;
}
static inline bool chr_non_special(int num) { return num>0 && num<1004; } // Chromosome.h:78-81

struct BedLine
{
	std::string chr; int num = 0; int start = 0; int end = -1; // 1-based closed
	std::vector<std::string> annos;
	int length() const { return end - start + 1; }
	bool operator<(const BedLine& r) const // BedFile.cpp:29-35
	{
		if (num<r.num) return true;
		if (num>r.num) return false;
		if (start==r.start) return end<r.end;
		return start<r.start;
	}
	bool overlaps(int s, int e) const { return start<=e && s<=end; } // BasicStatistics::rangeOverlaps
};

struct BedFile
{
	std::vector<std::string> headers;
	std::vector<BedLine> lines;

	size_t count() const { return lines.size(); }
	long long baseCount() const { long long o=0; for (auto& l : lines) o += l.length(); return o; }

	void append(const BedLine& l)
	{
		if (l.num<=0) throw Error("Invalid BED line chromosome - empty string!");
		if (l.start<1 || l.end<1 || l.start>l.end) throw Error("Invalid BED line range '" + std::to_string(l.start) + "' to '" + std::to_string(l.end) + "'!");
		lines.push_back(l);
	}
	void append(const std::string& chr, int s, int e, std::vector<std::string> annos = {})
	{
		BedLine l; l.chr = trim(chr); l.num = chr_num(chr); l.start = s; l.end = e; l.annos = std::move(annos); append(l);
	}

	// BedFile.cpp:119-176 — 0-based half-open on disk -> 1-based closed in memory
	void loadText(const std::string& text)
	{
		lines.clear(); headers.clear();
		std::istringstream in(text); std::string line;
		while (std::getline(in, line))
		{
			while (!line.empty() && (line.back()=='\n' || line.back()=='\r')) line.pop_back();
			if (line.empty()) continue;
			if (line[0]=='#' || line.rfind("track ",0)==0 || line.rfind("browser ",0)==0 || line.rfind("Chromosome\tStart\tEnd",0)==0) { headers.push_back(line); continue; }
			std::vector<std::string> f; { size_t a=0; while (true) { size_t b = line.find('\t', a); if (b==std::string::npos) { f.push_back(line.substr(a)); break; } f.push_back(line.substr(a, b-a)); a = b+1; } }
			if (f.size()<3) throw Error("BED file line with less than three fields found: '" + trim(line) + "'");
			char* e1; char* e2;
			long s = strtol(f[1].c_str(), &e1, 10); long e = strtol(f[2].c_str(), &e2, 10);
			if (f[1].empty() || *e1) throw Error("BED file line with invalid starts position found: '" + trim(line) + "'");
			if (f[2].empty() || *e2) throw Error("BED file line with invalid end position found: '" + trim(line) + "'");
			std::vector<std::string> annos(f.begin()+3, f.end());
			append(f[0], (int)s + 1, (int)e, annos);
		}
	}
	void load(const std::string& path)
	{
		std::ifstream f(path, std::ios::binary);
		if (!f) throw Error("Could not open file for reading: '" + path + "'!");
		std::stringstream ss; ss << f.rdbuf();
		loadText(ss.str());
	}
	// BedFile.cpp:178-202
	std::string toText(bool with_headers = true) const
	{
		std::string o;
		if (with_headers) for (auto& h : headers) o += trim(h) + "\n";
		for (auto& l : lines)
		{
			o += l.chr + "\t" + std::to_string(l.start-1) + "\t" + std::to_string(l.end);
			for (auto& a : l.annos) o += "\t" + a;
			o += "\n";
		}
		return o;
	}

	bool isSorted() const { for (size_t i=1;i<lines.size();++i) if (lines[i]<lines[i-1]) return false; return true; }
	bool isMergedAndSorted() const // BedFile.cpp:621-637
	{
		for (size_t i=1;i<lines.size();++i)
		{
			if (lines[i]<lines[i-1]) return false;
			if (lines[i-1].num==lines[i].num && lines[i-1].overlaps(lines[i].start, lines[i].end)) return false;
		}
		return true;
	}
	void sort() { std::stable_sort(lines.begin(), lines.end()); }

	// BedFile.cpp:252-325
	void merge(bool merge_back_to_back = true, bool merge_names = false, bool merged_names_unique = false)
	{
		if (lines.empty()) return;
		for (auto& l : lines)
		{
			if (!merge_names) l.annos.clear();
			else { std::string name = l.annos.empty() ? "" : l.annos[0]; l.annos.assign(1, name); }
		}
		if (!isSorted()) sort();
		auto join = [](const std::vector<std::string>& v){ std::string o; for (size_t i=0;i<v.size();++i){ if(i) o += ","; o += v[i]; } return o; };
		BedLine next = lines[0]; size_t out = 0;
		for (size_t i=1;i<lines.size();++i)
		{
			const BedLine line = lines[i];
			bool ov = next.num==line.num && next.overlaps(line.start, line.end);
			bool adj = merge_back_to_back && next.num==line.num && (next.start==line.end+1 || next.end==line.start-1);
			if (ov || adj)
			{
				if (line.end>next.end) next.end = line.end;
				if (merge_names)
				{
					const std::string& a = line.annos[0];
					if (!merged_names_unique || std::find(next.annos.begin(), next.annos.end(), a)==next.annos.end()) next.annos.push_back(a);
				}
			}
			else
			{
				lines[out] = next;
				if (merge_names) lines[out].annos.assign(1, join(next.annos));
				++out; next = line;
			}
		}
		lines[out] = next;
		if (merge_names) lines[out].annos.assign(1, join(next.annos));
		lines.resize(out+1);
	}

	// BedFile.cpp:519-567
	void chunk(int chunk_size)
	{
		std::vector<BedLine> nl; nl.reserve(lines.size());
		for (auto& line : lines)
		{
			if (line.length()>chunk_size)
			{
				double length = line.length();
				int n = (int)floor(length/chunk_size);
				if (fabs(chunk_size-(length/n)) > fabs(chunk_size-(length/(n+1)))) n += 1;
				std::vector<int> sizes((size_t)n, chunk_size);
				int rest = line.length()-n*chunk_size; int cur = 0;
				while (rest!=0) { int sign = rest>0 ? 1 : -1; sizes[cur] += sign; rest -= sign; ++cur; if (cur==n) cur = 0; }
				int start = line.start; BedLine x = line;
				for (int i=0;i<n;++i) { int end = start+sizes[i]-1; x.start = start; x.end = end; nl.push_back(x); start = end+1; }
			}
			else nl.push_back(line);
		}
		lines.swap(nl);
	}
};

// ChromosomalIndex.h:103-135 — semantic result: indices (ascending) of all lines on chr overlapping [start,end].
// The reference walks back from an every-30th-line bin start by max_length_; the result set is identical to this
// binary-search + bounded linear scan (requires a sorted container, as the reference does: ChromosomalIndex.h:57-60).
struct ChrIndex
{
	const BedFile& bed; int max_len = -1;
	std::map<int, std::pair<size_t,size_t>> range; // chr num -> [first,last]
	explicit ChrIndex(const BedFile& b) : bed(b)
	{
		if (!b.isSorted()) throw Error("ChromosomalIndex::createIndex called on unsorted container!");
		for (size_t i=0;i<b.lines.size();++i)
		{
			auto it = range.find(b.lines[i].num);
			if (it==range.end()) range[b.lines[i].num] = {i,i}; else it->second.second = i;
			max_len = std::max(max_len, b.lines[i].length());
		}
	}
	template <class F> void forMatches(int num, int start, int end, F f) const
	{
		auto it = range.find(num); if (it==range.end()) return;
		size_t lo = it->second.first, hi = it->second.second + 1;
		size_t a = lo, b = hi; int key = start - max_len; // first line with line.start >= start-max_len
		while (a<b) { size_t m = (a+b)/2; if (bed.lines[m].start < key) a = m+1; else b = m; }
		for (size_t i=a; i<hi && bed.lines[i].start < end + max_len; ++i)
			if (bed.lines[i].overlaps(start, end)) f((int)i);
	}
	std::vector<int> matchingIndices(int num, int start, int end) const { std::vector<int> v; forMatches(num, start, end, [&](int i){ v.push_back(i); }); return v; }
	int matchingIndex(int num, int start, int end) const { int r=-1; forMatches(num, start, end, [&](int i){ if (r<0) r=i; }); return r; }
};

// cppCORE Histogram restated (see file header).
struct Histogram
{
	double min, max, bin; std::vector<double> bins; double sum = 0;
	Histogram(double mn, double mx, double b) : min(mn), max(mx), bin(b), bins((size_t)ceil((mx-mn)/b), 0.0) {}
	int binCount() const { return (int)bins.size(); }
	int binIndex(double v) const
	{
		if (v<min || v>max) throw Error("Requested position not in range!");
		int i = (int)floor((v-min)/(max-min)*bins.size());
		i = std::max(0, i); i = std::min(i, (int)bins.size()-1);
		return i;
	}
	void inc(double v, bool ignore_bounds)
	{
		if (ignore_bounds) v = std::min(std::max(v, min), max);
		bins[binIndex(v)] += 1; sum += 1;
	}
	double binValue(int i) const { return bins[i]; }
	double binSum() const { return sum; }
};

} // namespace orc
