// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY (see bamio.hpp header).
// CPU restatement of the per-read QC / coverage loops of ngs-bits:
//   Statistics::mapping(bed,...)       src/cppNGS/Statistics.cpp:343-803   -> mapping_roi()
//   Statistics::mapping(bam,...)       src/cppNGS/Statistics.cpp:805-988   -> mapping_noroi(wgs_style=false)
//   Statistics::mapping_wgs            src/cppNGS/Statistics.cpp:990-1359  -> mapping_wgs()
//   Statistics::yxRatio                src/cppNGS/Statistics.cpp:2659-2691 -> yx_ratio()
//   Statistics::avgCoverage + workers  src/cppNGS/Statistics.cpp:2698-2804, WorkerAverageCoverage.cpp:17-173
//   Statistics::lowOrHighCoverage      src/cppNGS/Statistics.cpp:2534-2657, WorkerLowOrHighCoverage.cpp:18-252
//   BamAlignment::qualities            src/cppNGS/BamReader.cpp:210-255
//   BamReader::getPileup + BamAlignment::extractBaseByCIGAR  src/cppNGS/BamReader.cpp:809-885, 307-374 -> site_pileup()
//   Statistics::contamination          src/cppNGS/Statistics.cpp:2333-2386 -> contamination_value()
//   StatisticsReads::update(BamAlignment) src/cppNGS/StatisticsReads.cpp:83-158 -> reads_qc()
//   FastaFileIndex::seq / n            src/cppNGS/FastaFileIndex.cpp:72-131 ; Sequence::gcContent Sequence.cpp:86-101
// Written as the straight sequential loops of the reference on purpose: this is the checker, not the product.
// ============================================================================
#pragma once
#include <array>
#include <cinttypes>
#include <numeric>
#include <limits>
#include <memory>
#include "bed.hpp"

namespace orc {

// ---------------------------------------------------------------- FASTA (.fai) access
struct Fasta
{
	struct Entry { int length; long long offset; int line_blen; int line_len; };
	std::string path; std::map<int, Entry> idx; FILE* f = nullptr;
	std::map<int,int> ncache;
	explicit Fasta(const std::string& p) : path(p)
	{
		f = fopen(p.c_str(), "rb");
		if (!f) throw Error("Could not open FASTA file '" + p + "' for reading!");
		std::ifstream fai(p + ".fai");
		if (!fai) throw Error("Could not open file for reading: '" + p + ".fai'!");
		std::string line; int n = 0;
		while (std::getline(fai, line))
		{
			++n; std::vector<std::string> fl; std::stringstream ss(line); std::string x; while (std::getline(ss, x, '\t')) fl.push_back(x);
			if (fl.size()!=5) throw Error("Malformed FASTA index line " + std::to_string(n));
			idx[chr_num(fl[0])] = Entry{atoi(fl[1].c_str()), atoll(fl[2].c_str()), atoi(fl[3].c_str()), atoi(fl[4].c_str())};
		}
		if (idx.empty()) throw Error("Empty FAI file for " + p + "'!");
	}
	~Fasta() { if (f) fclose(f); }
	std::string readRaw(long long pos, int n) const
	{
		std::string s((size_t)std::max(n,0), '\0');
		fseeko(f, pos, SEEK_SET);
		size_t got = fread(&s[0], 1, s.size(), f); s.resize(got);
		s.erase(std::remove(s.begin(), s.end(), '\n'), s.end());
		return s;
	}
	// FastaFileIndex.cpp:72-115 (including its newline arithmetic, kept as is)
	std::string seq(int num, int start, int length, bool upper = true) const
	{
		start -= 1;
		auto it = idx.find(num); if (it==idx.end()) throw Error("Unknown FASTA index chromosome requested!");
		const Entry& e = it->second;
		if (start > e.length) throw Error("FastaFileIndex::seq: Invalid start position");
		if (start+length > e.length) length = std::min(length, e.length - start);
		int nl_before = start>0 ? (start-1)/e.line_blen : 0;
		long long pos = e.offset + nl_before + start;
		int nl_by_end = (start+length-1)/e.line_blen;
		int seqlen = length + (nl_by_end - nl_before);
		std::string s = readRaw(pos, seqlen);
		if (upper) for (auto& c : s) c = (char)toupper((unsigned char)c);
		return s;
	}
	// FastaFileIndex.cpp:117-131
	int n(int num)
	{
		auto c = ncache.find(num); if (c!=ncache.end()) return c->second;
		auto it = idx.find(num); if (it==idx.end()) throw Error("Unknown FASTA index chromosome requested!");
		const Entry& e = it->second;
		std::string s = readRaw(e.offset, e.length / e.line_blen + e.length);
		int o = 0; for (char ch : s) if (ch=='N' || ch=='n') ++o;
		return ncache[num] = o;
	}
};
static inline double gc_content(const std::string& s) // Sequence.cpp:86-101
{
	int gc=0, at=0;
	for (char b : s) { if (b=='G'||b=='C') ++gc; else if (b=='A'||b=='T') ++at; }
	if (gc+at==0) return std::numeric_limits<double>::quiet_NaN();
	return (double)gc/(gc+at);
}

// ---------------------------------------------------------------- result containers
struct QcLine { std::string accession, name, value; bool is_plot = false; };

static inline std::string fmt(double v, int prec = 2) { if (v != v) return "nan"; char b[64]; snprintf(b, sizeof(b), "%.*f", prec, v); return b; } // QString::number(d,'f',prec): a NaN is "nan" whatever its sign (QCCollection.cpp:121-126)

struct MappingResult
{
	// raw integer state of the loop (everything the GPU path must match bit-exactly)
	int64_t al_total=0, al_mapped=0, al_ontarget=0, al_neartarget=0, al_dup=0, al_proper_paired=0, insert_size_read_count=0;
	int64_t bases_trimmed=0, bases_mapped=0, bases_clipped=0, insert_size_sum=0;
	int64_t bases_usable=0, bases_usable_no_overlap=0, bases_usable_raw=0, bases_usable_roi=0;
	int64_t bases_usable_dp[5] = {0,0,0,0,0};
	int64_t dp_dist[4] = {0,0,0,0};
	int64_t insert_hist[1000];           // count per integer insert size 0..999 (pre-binning)
	int32_t max_length=0, paired_end=0;
	int64_t roi_bases=0, half_depth=0, bases_covered_half=0;
	int64_t reads_x=0, reads_y=0; int32_t yx_valid=0;
	std::vector<int32_t> depth;          // per-base depth, ROI lines concatenated in order
	std::vector<double> gc_roi, gc_reads; // 100 bins each (only when a FASTA is given)
	bool have_gc=false;
	double genome_size=0, no_base=0;
	std::vector<QcLine> lines;           // formatted output in insertion order
	MappingResult() { memset(insert_hist, 0, sizeof(insert_hist)); }
};

static inline void add(MappingResult& r, const char* acc, const std::string& name, double v) { r.lines.push_back({acc, name, fmt(v), false}); }
static inline void adds(MappingResult& r, const char* acc, const std::string& name, const std::string& v) { r.lines.push_back({acc, name, v, false}); }
static inline void addp(MappingResult& r, const char* acc, const std::string& name) { r.lines.push_back({acc, name, "", true}); }

struct RefChroms
{
	std::vector<int> num; // per tid
	explicit RefChroms(const BamFile& b) { for (auto& n : b.ref_names) num.push_back(chr_num(n)); }
	int tidOf(int chrnum) const { for (size_t i=0;i<num.size();++i) if (num[i]==chrnum) return (int)i; return -1; }
};

// Statistics.cpp:2659-2691
static inline void yx_ratio(const BamFile& bam, const RefChroms& rc, MappingResult& r)
{
	int tx = rc.tidOf(1001), ty = rc.tidOf(1002);
	if (tx<0 || ty<0) { r.yx_valid = 0; return; }
	int64_t ny=0, nx=0;
	bam.forRegion(ty, 1, (int)bam.ref_lens[ty], [&](const Rec& al){ if (al.isSecondary()||al.isSupplementary()) return; ++ny; });
	bam.forRegion(tx, 1, (int)bam.ref_lens[tx], [&](const Rec& al){ if (al.isSecondary()||al.isSupplementary()) return; ++nx; });
	r.reads_x = nx; r.reads_y = ny;
	r.yx_valid = nx!=0;
}
static inline void add_yx(MappingResult& r)
{
	if (r.yx_valid) r.lines.push_back({"QC:2000139", "chrY/chrX read ratio", fmt((double)r.reads_y/(double)r.reads_x, 4), false});
}

struct GcBins
{
	BedFile dropout; std::vector<int> bin; std::unique_ptr<ChrIndex> index;
	std::vector<double> gc_roi = std::vector<double>(100, 0.0), gc_reads = std::vector<double>(100, 0.0);
	// Statistics.cpp:363-387 / 1022-1045
	GcBins(const BedFile& roi, Fasta* fa)
	{
		dropout.lines = roi.lines; dropout.chunk(100);
		bin.assign(dropout.count(), -1);
		for (size_t i=0;i<dropout.count();++i)
		{
			if (!fa) continue;
			const BedLine& l = dropout.lines[i];
			double gc = gc_content(fa->seq(l.num, l.start, l.length()));
			if (std::isnan(gc) || std::isinf(gc)) bin[i] = -1;
			else { int b = (int)std::floor(100.0*gc); bin[i] = b; if (b>=0 && b<100) gc_roi[b] += 1.0; else if (b==100) { gc_roi.resize(101); gc_reads.resize(101); gc_roi[100] += 1.0; } }
		}
		index.reset(new ChrIndex(dropout));
	}
	void hit(int num, int start, int end)
	{
		std::vector<int> ind = index->matchingIndices(num, start, end);
		for (int i : ind) { int b = bin[i]; if (b>=0) { if ((size_t)b>=gc_reads.size()) gc_reads.resize(b+1); gc_reads[b] += 1.0/ind.size(); } }
	}
	// Statistics.cpp:576-604
	void dropoutValues(double& at, double& gc) const
	{
		double gc_sum = std::accumulate(gc_roi.begin(), gc_roi.end(), 0.0);
		double roi_sum = std::accumulate(gc_reads.begin(), gc_reads.end(), 0.0);
		at = 0; gc = 0;
		for (int i=0;i<100;++i)
		{
			double diff = 100.0*gc_roi[i]/gc_sum - 100.0*gc_reads[i]/roi_sum;
			if (diff>0) { if (i<=50) at += diff; if (i>=50) gc += diff; }
		}
	}
};

// The per-record counter block shared by all three mapping variants (Statistics.cpp:416-454, 547-573).
struct LoopState { MappingResult& r; bool spliced=false; };

// ---------------------------------------------------------------- Statistics::mapping (ROI)  Statistics.cpp:343-803
static inline MappingResult mapping_roi(const BedFile& bed, const BamFile& bam, Fasta* fa, int min_mapq, bool is_cfdna)
{
	if (!bed.isMergedAndSorted()) throw Error("Merged and sorted BED file required for coverage details statistics!");
	MappingResult r; RefChroms rc(bam);
	ChrIndex roi_index(bed);
	std::vector<size_t> doff(bed.count()+1, 0);
	for (size_t i=0;i<bed.count();++i) { doff[i+1] = doff[i] + bed.lines[i].length(); }
	r.roi_bases = (int64_t)doff.back();
	r.depth.assign(doff.back(), 0);
	GcBins gc(bed, fa);
	Histogram insert_dist(0, 999, 5), dp_dist(0.5, 4.5, 1);

	for (size_t ri=0; ri<bam.count(); ++ri)
	{
		Rec al = bam.rec(ri);
		if (al.isSecondary() || al.isSupplementary()) continue;
		++r.al_total;
		if (al.isPaired()) r.paired_end = 1;
		const int length = al.length();
		r.max_length = std::max(r.max_length, length);
		bool spliced = false;
		if (!al.isUnmapped())
		{
			++r.al_mapped;
			const int start_pos = al.start(), end_pos = al.end();
			r.bases_mapped += length;
			for (uint32_t i=0;i<al.n_cigar;++i)
			{
				uint32_t op = al.cigarOp(i);
				if (op==4 || op==5) r.bases_clipped += al.cigarLen(i);
				else if (op==3) spliced = true;
			}
			int num = (al.tid>=0 && (size_t)al.tid<rc.num.size()) ? rc.num[al.tid] : -1;
			std::vector<int> indices = roi_index.matchingIndices(num, start_pos-250, end_pos+250);
			if (!indices.empty())
			{
				++r.al_neartarget;
				indices = roi_index.matchingIndices(num, start_pos, end_pos);
				if (!indices.empty())
				{
					++r.al_ontarget;
					int dp = aux_tagi(al, "DP");
					if (dp!=0) { dp_dist.inc(std::min(dp,4), true); }
					if (!al.isDuplicate() && al.mapq>=min_mapq)
					{
						for (int index : indices)
						{
							const int ol_start = std::max(bed.lines[index].start, start_pos);
							const int ol_end = std::min(bed.lines[index].end, end_pos);
							const int64_t n = ol_end - ol_start + 1;
							r.bases_usable += n;
							r.bases_usable_dp[std::min(dp,4)] += n;
							r.bases_usable_raw += n * (dp + 1);
							int* d = r.depth.data() + doff[index] - bed.lines[index].start;
							for (int p=ol_start; p<=ol_end; ++p) d[p] += 1;
							r.bases_usable_no_overlap += n;
						}
						const int insert_size = std::abs(al.isize);
						if (al.isRead1() && al.isPaired() && al.isProperPair() && !spliced && 2*length > insert_size)
						{
							const int ovl = 2*length - insert_size;
							int os = (al.isize>0) ? start_pos + length - ovl : start_pos;
							int oe = os + ovl - 1;
							roi_index.forMatches(num, os, oe, [&](int index){
								const int a = std::max(bed.lines[index].start, os), b = std::min(bed.lines[index].end, oe);
								r.bases_usable_no_overlap -= (b - a + 1);
							});
						}
					}
					gc.hit(num, start_pos, end_pos);
				}
			}
		}
		if (al.isPaired() && al.isProperPair())
		{
			++r.al_proper_paired;
			if (!spliced)
			{
				const int insert_size = std::abs(al.isize);
				if (insert_size<1000) { ++r.insert_size_read_count; r.insert_size_sum += insert_size; insert_dist.inc(insert_size, true); r.insert_hist[insert_size]++; }
			}
		}
		if (length<r.max_length && length!=-1) r.bases_trimmed += (r.max_length - length);
		if (al.isDuplicate()) ++r.al_dup;
	}
	for (int i=0;i<4;++i) r.dp_dist[i] = (int64_t)dp_dist.binValue(i);

	double at_dropout=0, gc_dropout=0; gc.dropoutValues(at_dropout, gc_dropout);
	r.gc_roi = gc.gc_roi; r.gc_reads = gc.gc_reads; r.have_gc = fa!=nullptr;

	double avg_depth = (double)r.bases_usable / r.roi_bases;
	int half_depth = (int)std::round(0.5*avg_depth);
	int hist_max = 599, hist_step = 5;
	if (avg_depth>200) { hist_max += 400; hist_step += 5; }
	if (avg_depth>500) hist_max += 500;
	if (avg_depth>1000) hist_max += 1000;
	if (is_cfdna) { hist_max = 20000; hist_step = 500; }
	Histogram depth_dist(0, hist_max, hist_step);
	for (int32_t d : r.depth) { depth_dist.inc(d, true); if (d>=half_depth) ++r.bases_covered_half; }
	r.half_depth = half_depth;

	const double bt = (double)r.bases_trimmed, bc = (double)r.bases_clipped, bm = (double)r.bases_mapped;
	add(r, "QC:2000019", "trimmed base percentage", 100.0 * bt / r.al_total / r.max_length);
	add(r, "QC:2000052", "clipped base percentage", 100.0 * bc / bm);
	add(r, "QC:2000020", "mapped read percentage", 100.0 * r.al_mapped / r.al_total);
	add(r, "QC:2000021", "on-target read percentage", 100.0 * r.al_ontarget / r.al_total);
	add(r, "QC:2000057", "near-target read percentage", 100.0 * r.al_neartarget / r.al_total);
	if (r.paired_end)
	{
		add(r, "QC:2000022", "properly-paired read percentage", 100.0 * r.al_proper_paired / r.al_total);
		add(r, "QC:2000023", "insert size", (double)r.insert_size_sum / r.insert_size_read_count);
		add(r, "QC:2000150", "target region read depth (no ol)", (double)r.bases_usable_no_overlap / r.roi_bases);
	}
	else
	{
		adds(r, "QC:2000022", "properly-paired read percentage", "n/a (single end)");
		adds(r, "QC:2000023", "insert size", "n/a (single end)");
	}
	if (r.al_dup==0) adds(r, "QC:2000024", "duplicate read percentage", "n/a (no duplicates marked or duplicates removed during data analysis)");
	else add(r, "QC:2000024", "duplicate read percentage", 100.0 * r.al_dup / r.al_total);
	add(r, "QC:2000050", "bases usable (MB)", (double)r.bases_usable / 1000000.0);
	add(r, "QC:2000025", "target region read depth", avg_depth);
	if (is_cfdna)
	{
		double cum[5] = {0,0,0,0,0}; double run = 0;
		for (int i=4;i>=0;--i) { run += (double)r.bases_usable_dp[i] / r.roi_bases; cum[i] = run; }
		for (int i=2;i<=4;++i) add(r, ("QC:200007" + std::to_string(i-1)).c_str(), "target region read depth " + std::to_string(i) + "-fold duplication", cum[i]);
		add(r, "QC:2000074", "raw target region read depth", (double)r.bases_usable_raw / r.roi_bases);
	}
	std::vector<int> depths = {10,20,30,50,60,100,200,500};
	std::vector<std::string> acc = {"QC:2000026","QC:2000027","QC:2000028","QC:2000029","QC:2000099","QC:2000030","QC:2000031","QC:2000032"};
	if (is_cfdna) { for (int d : {1000,2500,5000,7500,10000,15000}) depths.push_back(d); for (const char* a : {"QC:2000065","QC:2000066","QC:2000067","QC:2000068","QC:2000069","QC:2000070"}) acc.push_back(a); }
	for (size_t i=0;i<depths.size();++i)
	{
		double cov = 0.0;
		for (int b=depth_dist.binIndex(depths[i]); b<depth_dist.binCount(); ++b) cov += depth_dist.binValue(b);
		add(r, acc[i].c_str(), "target region " + std::to_string(depths[i]) + "x percentage", 100.0 * cov / r.roi_bases);
	}
	add(r, "QC:2000058", "target region half depth percentage", 100.0 * r.bases_covered_half / r.roi_bases);
	add(r, "QC:2000059", "AT dropout", at_dropout);
	add(r, "QC:2000060", "GC dropout", gc_dropout);
	addp(r, "QC:2000037", "depth distribution plot");
	if (r.paired_end) addp(r, "QC:2000038", "insert size distribution plot");
	if (is_cfdna && dp_dist.binSum()!=0) { addp(r, "QC:2000075", "fragment duplication distribution plot"); addp(r, "QC:2000076", "duplication-coverage plot"); }
	addp(r, "QC:2000061", "GC bias plot");
	yx_ratio(bam, rc, r); add_yx(r);
	return r;
}

// ---------------------------------------------------------------- shared pass-1 loop of mapping(bam) / mapping_wgs
// Statistics.cpp:830-916 == :1068-1152 (identical bodies)
static inline void pass1_noroi(const BamFile& bam, const RefChroms& rc, int min_mapq, MappingResult& r)
{
	for (size_t ri=0; ri<bam.count(); ++ri)
	{
		Rec al = bam.rec(ri);
		if (al.isSecondary() || al.isSupplementary()) continue;
		++r.al_total;
		if (al.isPaired()) r.paired_end = 1;
		const int length = al.length();
		r.max_length = std::max(r.max_length, length);
		bool spliced = false;
		if (!al.isUnmapped())
		{
			++r.al_mapped;
			r.bases_mapped += length;
			for (uint32_t i=0;i<al.n_cigar;++i)
			{
				uint32_t op = al.cigarOp(i);
				if (op==4 || op==5) r.bases_clipped += al.cigarLen(i);
				else if (op==3) spliced = true;
			}
			// reader.chromosome(id) throws if id >= size (BamReader.cpp:775-780); negative ids are UB in the reference — treated as special here
			int num = (al.tid>=0 && (size_t)al.tid<rc.num.size()) ? rc.num[al.tid] : 0;
			if (chr_non_special(num))
			{
				++r.al_ontarget;
				if (!al.isDuplicate() && al.mapq>=min_mapq)
				{
					r.bases_usable += length;
					if (r.paired_end) r.bases_usable_no_overlap += length;
				}
			}
		}
		if (al.isPaired() && al.isProperPair())
		{
			++r.al_proper_paired;
			if (!spliced)
			{
				const int insert_size = std::abs(al.isize);
				if (insert_size<1000)
				{
					++r.insert_size_read_count; r.insert_size_sum += insert_size; r.insert_hist[insert_size]++;
					if (al.isRead1() && !al.isDuplicate() && al.mapq>=min_mapq && 2*length > insert_size) r.bases_usable_no_overlap -= (2*length) - insert_size;
				}
			}
		}
		if (length<r.max_length && length!=-1) r.bases_trimmed += (r.max_length - length);
		if (al.isDuplicate()) ++r.al_dup;
	}
	r.bases_usable -= r.bases_clipped; // Statistics.cpp:917 / :1183
}

static inline void genome_denominators(const BamFile& bam, const RefChroms& rc, Fasta* fa, MappingResult& r)
{
	// BamReader::genomeSize(false) BamReader.cpp:789-800 ; N count Statistics.cpp:920-928 / 1237-1245
	r.genome_size = 0; r.no_base = 0;
	for (size_t i=0;i<rc.num.size();++i) if (chr_non_special(rc.num[i])) { r.genome_size += (double)bam.ref_lens[i]; if (fa) r.no_base += fa->n(rc.num[i]); }
}

static inline void noroi_common_output(MappingResult& r, bool wgs_style)
{
	const double bt = (double)r.bases_trimmed, bc = (double)r.bases_clipped, bm = (double)r.bases_mapped;
	if (!wgs_style || r.paired_end) add(r, "QC:2000019", "trimmed base percentage", 100.0 * bt / r.al_total / r.max_length);
	else adds(r, "QC:2000019", "trimmed base percentage", "n/a (single end)");
	add(r, "QC:2000052", "clipped base percentage", 100.0 * bc / bm);
	add(r, "QC:2000020", "mapped read percentage", 100.0 * r.al_mapped / r.al_total);
	add(r, "QC:2000021", "on-target read percentage", 100.0 * r.al_ontarget / r.al_total);
	if (r.paired_end)
	{
		add(r, "QC:2000022", "properly-paired read percentage", 100.0 * r.al_proper_paired / r.al_total);
		add(r, "QC:2000023", "insert size", (double)r.insert_size_sum / r.insert_size_read_count);
		add(r, "QC:2000150", "target region read depth (no ol)", (double)r.bases_usable_no_overlap / (r.genome_size - r.no_base));
	}
	else
	{
		adds(r, "QC:2000022", "properly-paired read percentage", "n/a (single end)");
		adds(r, "QC:2000023", "insert size", "n/a (single end)");
	}
	if (r.al_dup==0) adds(r, "QC:2000024", "duplicate read percentage", "n/a (duplicates not marked or removed during data analysis)");
	else add(r, "QC:2000024", "duplicate read percentage", 100.0 * r.al_dup / r.al_total);
	add(r, "QC:2000050", "bases usable (MB)", (double)r.bases_usable / 1000000.0);
	add(r, "QC:2000025", "target region read depth", (double)r.bases_usable / (r.genome_size - r.no_base));
}

// ---------------------------------------------------------------- Statistics::mapping(bam, ref, min_mapq)  Statistics.cpp:805-988
static inline MappingResult mapping_noroi(const BamFile& bam, Fasta* fa, int min_mapq)
{
	MappingResult r; RefChroms rc(bam);
	pass1_noroi(bam, rc, min_mapq, r);
	genome_denominators(bam, rc, fa, r);
	noroi_common_output(r, false);
	int64_t isum = 0; for (int i=0;i<1000;++i) isum += r.insert_hist[i];
	if (r.paired_end && isum>0) addp(r, "QC:2000038", "insert size distribution plot");
	yx_ratio(bam, rc, r); add_yx(r);
	return r;
}

// ---------------------------------------------------------------- Statistics::mapping_wgs  Statistics.cpp:990-1359
static inline MappingResult mapping_wgs(const BamFile& bam, const BedFile* roi_in, Fasta* fa, int min_mapq)
{
	MappingResult r; RefChroms rc(bam);
	bool roi_available = roi_in!=nullptr;
	BedFile roi; if (roi_in) roi = *roi_in;
	if (roi_available && !roi.isMergedAndSorted()) { roi.sort(); roi.merge(); }
	std::vector<size_t> doff(roi.count()+1, 0);
	for (size_t i=0;i<roi.count();++i) doff[i+1] = doff[i] + roi.lines[i].length();
	r.depth.assign(doff.back(), 0);
	r.roi_bases = (int64_t)doff.back();
	GcBins gc(roi, fa);

	pass1_noroi(bam, rc, min_mapq, r);   // note: bases_usable -= bases_clipped happens after pass 2 in the reference (:1183) — same value

	for (size_t i=0;i<roi.count();++i)
	{
		const BedLine& reg = roi.lines[i];
		int tid = rc.tidOf(reg.num);
		if (tid<0) throw Error("Could not find chromosome '" + reg.chr + "' in BAM/CRAM file " + bam.path); // BamReader.cpp:751-754
		bam.forRegion(tid, reg.start, reg.end, [&](const Rec& al){
			if (al.isSecondary() || al.isSupplementary() || al.isUnmapped()) return;
			gc.hit(rc.num[al.tid], al.start(), al.end());
			if (!al.isDuplicate() && al.mapq>=min_mapq)
			{
				r.bases_usable_roi += al.length();
				int a = std::max(al.start(), reg.start), b = std::min(al.end(), reg.end);
				int* d = r.depth.data() + doff[i] - reg.start;
				for (int p=a; p<=b; ++p) d[p] += 1;
			}
		});
	}

	double avg_depth = (double)r.bases_usable_roi / (double)r.roi_bases;
	int half_depth = (int)std::round(0.5*avg_depth);
	Histogram depth_dist(0, 599, 5);
	for (int32_t d : r.depth) { depth_dist.inc(d, true); if (d>=half_depth) ++r.bases_covered_half; }
	r.half_depth = half_depth;
	double at_dropout=0, gc_dropout=0; gc.dropoutValues(at_dropout, gc_dropout);
	r.gc_roi = gc.gc_roi; r.gc_reads = gc.gc_reads; r.have_gc = fa!=nullptr;
	genome_denominators(bam, rc, fa, r);
	noroi_common_output(r, true);
	if (roi_available)
	{
		const int dv[8] = {10,20,30,50,60,100,200,500};
		const char* acc[8] = {"QC:2000026","QC:2000027","QC:2000028","QC:2000029","QC:2000099","QC:2000030","QC:2000031","QC:2000032"};
		for (int i=0;i<8;++i)
		{
			double cov = 0.0;
			for (int b=depth_dist.binIndex(dv[i]); b<depth_dist.binCount(); ++b) cov += depth_dist.binValue(b);
			add(r, acc[i], "target region " + std::to_string(dv[i]) + "x percentage", 100.0 * cov / (double)r.roi_bases);
		}
		add(r, "QC:2000058", "target region half depth percentage", 100.0 * r.bases_covered_half / (double)r.roi_bases);
		add(r, "QC:2000059", "AT dropout", at_dropout);
		add(r, "QC:2000060", "GC dropout", gc_dropout);
		addp(r, "QC:2000037", "depth distribution plot");
	}
	int64_t isum = 0; for (int i=0;i<1000;++i) isum += r.insert_hist[i];
	if (r.paired_end && isum>0) addp(r, "QC:2000038", "insert size distribution plot");
	if (roi_available) addp(r, "QC:2000061", "GC bias plot");
	yx_ratio(bam, rc, r); add_yx(r);
	return r;
}

// ---------------------------------------------------------------- avgCoverage  Statistics.cpp:2698-2804 + WorkerAverageCoverage.cpp
// Returns the per-line coverage sums (long) and appends the formatted annotation to each line like the reference.
// BedReadCount: src/BedReadCount/main.cpp:33-71 (readCount). The BED must be merged + sorted; every read that is mapped, not
// secondary / supplementary and has MAPQ >= min_mapq counts once for every line its [start, end] overlaps. Appends the count as annotation.
static inline std::vector<int64_t> read_counts(BedFile& bed, const BamFile& bam, int min_mapq)
{
	if (!bed.isMergedAndSorted()) throw Error("Merged and sorted BED file required for coverage calculation!");
	std::vector<int64_t> n(bed.count(), 0);
	ChrIndex index(bed);
	std::vector<int> num; for (auto& nm : bam.ref_names) num.push_back(chr_num(nm));
	for (size_t k=0; k<bam.count(); ++k)
	{
		const Rec al = bam.rec(k);
		if (al.isUnmapped()) continue;
		if (al.isSecondary() || al.isSupplementary()) continue;
		if (al.mapq < min_mapq) continue;
		if (al.tid<0 || (size_t)al.tid>=num.size()) continue;
		index.forMatches(num[al.tid], al.start(), al.end(), [&](int i){ n[i] += 1; });
	}
	for (size_t i=0;i<bed.count();++i) bed.lines[i].annos.push_back(std::to_string(n[i]));
	return n;
}

static inline std::vector<int64_t> avg_coverage(BedFile& bed, const BamFile& bam, int min_mapq, int decimals, bool random_access, bool skip_mismapped)
{
	if (!random_access && !bed.isSorted()) throw Error("Input BED file has to be sorted for sweep algorithm!");
	RefChroms rc(bam);
	std::vector<int64_t> cov(bed.count(), 0);
	auto pass = [&](const Rec& al){
		if (al.isDuplicate() || al.isSecondary() || al.isSupplementary()) return false;
		if (al.isUnmapped() || al.mapq<min_mapq) return false;
		if (skip_mismapped && !al.isProperPair() && al.mapq<20) return false;
		return true;
	};
	if (random_access)
	{
		for (size_t i=0;i<bed.count();++i) // WorkerAverageCoverage.cpp:30-55
		{
			const BedLine& l = bed.lines[i];
			int tid = rc.tidOf(l.num);
			if (tid<0) throw Error("Could not find chromosome '" + l.chr + "' in BAM/CRAM file " + bam.path);
			bam.forRegion(tid, l.start, l.end, [&](const Rec& al){
				if (!pass(al)) return;
				int a = std::max(l.start, al.start()), b = std::min(l.end, al.end());
				if (a<=b) cov[i] += b - a + 1;
			});
		}
	}
	else
	{
		ChrIndex index(bed); // WorkerAverageCoverage.cpp:121
		std::vector<int> chrs; for (auto& l : bed.lines) if (std::find(chrs.begin(), chrs.end(), l.num)==chrs.end()) chrs.push_back(l.num);
		for (int num : chrs)
		{
			int start=-1, end=-1; std::string name;
			for (auto& l : bed.lines) if (l.num==num) { if (start<0) { start = l.start; end = l.end; name = l.chr; } start = std::min(start, l.start); end = std::max(end, l.end); }
			int tid = rc.tidOf(num);
			if (tid<0) throw Error("Could not find chromosome '" + name + "' in BAM/CRAM file " + bam.path);
			bam.forRegion(tid, start, end, [&](const Rec& al){
				if (!pass(al)) return;
				index.forMatches(num, al.start(), al.end(), [&](int i){
					int a = std::max(bed.lines[i].start, al.start()), b = std::min(bed.lines[i].end, al.end());
					if (a<=b) cov[i] += b - a + 1;
				});
			});
		}
	}
	for (size_t i=0;i<bed.count();++i) bed.lines[i].annos.push_back(fmt((double)cov[i] / bed.lines[i].length(), decimals));
	return cov;
}

// BamAlignment::qualities  BamReader.cpp:210-255 — bit per reference offset, default true; only M ops clear bits;
// D and N advance the genome index; I and S advance the read index; =, X, H, P advance nothing.
static inline void base_qual_mask(const Rec& al, int min_baseq, int len, std::vector<uint8_t>& out)
{
	out.assign((size_t)std::max(len,0), 1);
	int ai = 0, gi = 0;
	for (uint32_t i=0;i<al.n_cigar;++i)
	{
		uint32_t op = al.cigarOp(i), l = al.cigarLen(i);
		if (op==0) { for (uint32_t k=0;k<l;++k) { if (al.qual[ai] < min_baseq && gi<len) out[gi] = 0; ++ai; ++gi; } }
		else if (op==2 || op==3) gi += l;
		else if (op==1 || op==4) ai += l;
	}
}

// ---------------------------------------------------------------- lowOrHighCoverage  Statistics.cpp:2534-2657 + WorkerLowOrHighCoverage.cpp
static inline BedFile low_high_coverage(const BedFile& bed, const BamFile& bam, int cutoff, int min_mapq, int min_baseq, bool is_high, bool random_access, std::vector<int32_t>* depth_out = nullptr)
{
	if (!random_access && !bed.isSorted()) throw Error("Input BED file has to be sorted for sweep algorithm!");
	RefChroms rc(bam);
	BedFile output;
	auto pass = [&](const Rec& al){
		if (al.isDuplicate()) return false;
		if (al.isSecondary() || al.isSupplementary()) return false;
		if (al.isUnmapped() || al.mapq<min_mapq) return false;
		return true;
	};
	std::vector<uint8_t> mask;
	if (random_access)
	{
		for (size_t i=0;i<bed.count();++i) // WorkerLowOrHighCoverage.cpp:31-108
		{
			const BedLine& l = bed.lines[i]; const int start = l.start;
			std::vector<int> cov((size_t)l.length(), 0);
			int tid = rc.tidOf(l.num);
			if (tid<0) throw Error("Could not find chromosome '" + l.chr + "' in BAM/CRAM file " + bam.path);
			bam.forRegion(tid, start, l.end, [&](const Rec& al){
				if (!pass(al)) return;
				const int os = std::max(start, al.start()) - start, oe = std::min(l.end, al.end()) - start;
				if (min_baseq>0)
				{
					int qp = std::max(start, al.start()) - al.start();
					base_qual_mask(al, min_baseq, al.end() - al.start() + 1, mask);
					for (int p=os;p<=oe;++p) { if (mask[qp]) ++cov[p]; ++qp; }
				}
				else for (int p=os;p<=oe;++p) ++cov[p];
			});
			if (depth_out) depth_out->insert(depth_out->end(), cov.begin(), cov.end());
			bool open=false; int rs=-1;
			for (int p=0;p<(int)cov.size();++p)
			{
				bool filter = is_high ? cov[p]>=cutoff : cov[p]<cutoff;
				if (open && !filter) { output.append(l.chr, rs+start, p+start-1, l.annos); open=false; rs=-1; }
				if (!open && filter) { open=true; rs=p; }
			}
			if (open) output.append(l.chr, rs+start, l.length()+start-1, l.annos);
		}
	}
	else
	{
		if (cutoff>255) throw Error("Cutoff cannot be bigger than 255!"); // WorkerLowOrHighCoverage.cpp:149
		ChrIndex bed_index(bed);
		// chromosome chunks ordered by base count (desc), Statistics.cpp:2545-2590 — QSet iteration order is unspecified
		// in the reference and the final merge() re-sorts, so order does not matter.
		std::vector<int> chrs; for (auto& l : bed.lines) if (std::find(chrs.begin(), chrs.end(), l.num)==chrs.end()) chrs.push_back(l.num);
		std::vector<int32_t> dcat;
		for (int num : chrs)
		{
			size_t first=bed.count(), last=0; for (size_t i=0;i<bed.count();++i) if (bed.lines[i].num==num) { first = std::min(first,i); last = std::max(last,i); }
			const std::string chr = bed.lines[first].chr;
			int tid = rc.tidOf(num);
			if (tid<0) throw Error("Chromosome '" + chr + "' not known in BAM/CRAM file " + bam.path); // BamReader.cpp:784
			int max_pos = (int)bam.ref_lens[tid];
			std::vector<uint8_t> cov((size_t)max_pos+1, 0);
			bam.forRegion(tid, 0, max_pos, [&](const Rec& al){
				if (!pass(al)) return;
				int start = al.start(), end = al.end();
				if (bed_index.matchingIndex(num, start, end)==-1) return;
				if (min_baseq>0)
				{
					base_qual_mask(al, min_baseq, end-start+1, mask);
					int qp=0; for (int p=start;p<=end;++p) { if (mask[qp] && p<=max_pos) { if (cov[p]<254) ++cov[p]; } ++qp; }
				}
				else for (int p=start;p<=end && p<=max_pos;++p) { if (cov[p]<254) ++cov[p]; }
			});
			for (size_t i=first;i<=last;++i)
			{
				const BedLine& l = bed.lines[i];
				if (l.num!=num) continue;
				bool open=false; int rs=-1;
				for (int p=l.start;p<=l.end;++p)
				{
					uint8_t c = p<=max_pos ? cov[p] : 0;
					if (depth_out) depth_out->push_back(c);
					bool filter = is_high ? c>=cutoff : c<cutoff;
					if (open && !filter) { output.append(chr, rs, p-1, l.annos); open=false; rs=-1; }
					if (!open && filter) { open=true; rs=p; }
				}
				if (open) output.append(chr, rs, l.end, l.annos);
			}
		}
	}
	output.merge(true, true, true); // Statistics.cpp:2655
	return output;
}

// ---------------------------------------------------------------- site pileup (BamReader::getPileup, SNP part only)
// BamAlignment::extractBaseByCIGAR (BamReader.cpp:307-374). Returns the base character ('~' = nothing to count, '-' =
// deleted) and the quality of that base (255 for a deletion, -1 for '~'). pos is 1-based.
static inline std::pair<char,int> extract_base_by_cigar(const Rec& al, int pos)
{
	int read_pos = 0, genome_pos = al.start() - 1;
	{
		// cigarIsOnlyInsertion (BamReader.cpp:90-100) looks at the CORE cigar ops; an empty CIGAR counts as "only insertions"
		bool only = true;
		for (uint32_t i=0; i<al.n_cigar; ++i) { uint32_t op = al.cigarOp(i); if (op!=1 && op!=4) { only = false; break; } }
		if (only) return {'~', -1};
	}
	for (uint32_t i=0; i<al.n_cigar; ++i)
	{
		const uint32_t op = al.cigarOp(i); const int len = (int)al.cigarLen(i);
		if (op==0 || op==7 || op==8) { genome_pos += len; read_pos += len; }
		else if (op==1) read_pos += len;
		else if (op==2) { genome_pos += len; if (genome_pos>=pos) return {'-', 255}; }
		else if (op==3) { genome_pos += len; if (genome_pos>=pos) return {'~', -1}; }
		else if (op==4) { read_pos += len; if (read_pos>=al.length()) return {'~', -1}; }
		else if (op==5) {}
		else throw Error("Unknown CIGAR operation!");
		if (genome_pos>=pos)
		{
			const int actual_pos = read_pos - (genome_pos + 1 - pos);
			const int nib = (al.seq[actual_pos>>1] >> ((~actual_pos & 1) << 2)) & 0xf;
			return {"=ACMGRSVTWYHKDBN"[nib], (int)al.qual[actual_pos]};
		}
	}
	throw Error("Could not find position " + std::to_string(pos) + " in read with start position " + std::to_string(al.start()) + "!");
}

// counts per site: A, C, G, T, N, deletion (Pileup::inc, Pileup.cpp:17-32); any other IUPAC letter throws like the reference
struct SiteCounts { int64_t a=0, c=0, g=0, t=0, n=0, del=0; };
static inline SiteCounts site_pileup(const BamFile& bam, int tid, int pos, int min_mapq, bool include_not_properly_paired, int min_baseq)
{
	SiteCounts out;
	bam.forRegion(tid, pos, pos, [&](const Rec& al) {
		if (al.isSecondary() || al.isSupplementary() || al.isDuplicate() || al.isUnmapped()) return;   // BamReader.cpp:830
		if (!al.isProperPair() && !include_not_properly_paired) return;                                 // :831
		if ((int)al.mapq < min_mapq) return;                                                            // :836
		auto base = extract_base_by_cigar(al, pos);                                                     // :866
		if (base.second >= min_baseq)
		{
			switch (base.first)
			{
				case 'A': ++out.a; break; case 'C': ++out.c; break; case 'G': ++out.g; break; case 'T': ++out.t; break; case 'N': ++out.n; break;
				case '-': ++out.del; break; case '~': break;
				default: throw Error(std::string("Unknown base '") + base.first + "' in pileup!");
			}
		}
	});
	return out;
}

// Statistics::contamination (Statistics.cpp:2333-2386) on a list of known SNVs (tid, pos, ref, alt) already filtered by
// allele frequency / SNV / target region (NGSHelper::getKnownVariants). Returns the QC value string.
struct KnownSnp { int tid; int pos; char ref; char alt; };
static inline std::string contamination_value(const BamFile& bam, const std::vector<KnownSnp>& snps, bool include_not_properly_paired, int min_cov = 20, int min_snps = 50)
{
	Histogram hist(0, 1, 0.05);
	int passed = 0;
	for (const KnownSnp& s : snps)
	{
		SiteCounts p = site_pileup(bam, s.tid, s.pos, 1, include_not_properly_paired, 13);
		const int64_t depth = p.a + p.c + p.g + p.t;                      // Pileup::depth(false)
		if (depth < min_cov) continue;
		auto cnt = [&](char b) -> double { b = (char)toupper(b); return b=='A' ? p.a : b=='C' ? p.c : b=='G' ? p.g : b=='T' ? p.t : b=='N' ? p.n : -1; };
		const double w = cnt(s.ref), m = cnt(s.alt);
		if (w < 0 || m < 0) throw Error("Unknown base in frequency calculation!");
		if (w + m == 0) continue;                                          // NaN frequency: non-informative
		++passed;
		hist.inc(m / (w + m), false);
	}
	double off = 0.0;   // Histogram::binValue(i, true): percentage of all counted values
	for (int i=1; i<=5; ++i) off += 100.0 * hist.binValue(i) / hist.binSum();
	for (int i=14; i<=18; ++i) off += 100.0 * hist.binValue(i) / hist.binSum();
	return passed < min_snps ? std::string("n/a") : fmt(off, 2);
}

// ---------------------------------------------------------------- raw-read QC (StatisticsReads::update(const BamAlignment&))
struct ReadsQc
{
	int64_t c_forward = 0, c_reverse = 0, bases_sequenced = 0, c_read_q20 = 0, c_base_q20 = 0, c_base_q30 = 0;
	std::map<int, int64_t> read_lengths;
	std::vector<std::array<int64_t,5>> pileups;            // per cycle: A, C, G, T, N
	std::vector<double> qualities1, qualities2;            // per cycle quality sums, forward / reverse
	std::vector<int64_t> base_qualities = std::vector<int64_t>(100, 0), read_qualities = std::vector<int64_t>(100, 0);
	Histogram qscore_dist_r1 = Histogram(0, 60, 1), qscore_dist_r2 = Histogram(0, 60, 1);
};
static inline ReadsQc reads_qc(const BamFile& bam, bool single_end)
{
	ReadsQc o;
	for (size_t n = 0; n < bam.count(); ++n)
	{
		const Rec al = bam.rec(n);
		if (al.isSupplementary() || al.isSecondary()) continue;                                   // :86
		bool is_forward;
		if (single_end) { is_forward = true; ++o.c_forward; }
		else { is_forward = al.isRead1(); if (is_forward) ++o.c_forward; else ++o.c_reverse; }    // :90-106
		const int cycles = al.length();
		o.bases_sequenced += cycles; o.read_lengths[cycles]++;
		if (cycles > (int)o.pileups.size()) { o.pileups.resize((size_t)cycles, {0,0,0,0,0}); o.qualities1.resize((size_t)cycles, 0.0); o.qualities2.resize((size_t)cycles, 0.0); }
		for (int i = 0; i < cycles; ++i)
		{
			const int base = (al.seq[i>>1] >> ((~i & 1) << 2)) & 0xf;                              // baseIntegers(): A=1, C=2, G=4, T=8, N=15
			if (base==1) o.pileups[(size_t)i][0]++; else if (base==2) o.pileups[(size_t)i][1]++; else if (base==4) o.pileups[(size_t)i][2]++;
			else if (base==8) o.pileups[(size_t)i][3]++; else if (base==15) o.pileups[(size_t)i][4]++;
			else throw Error("Unknown base '" + std::to_string(base) + "' in StatisticsReads::update!");
		}
		double q_sum = 0.0;
		for (int i = 0; i < cycles; ++i)
		{
			const int q = al.qual[i];
			q_sum += q;
			if (q >= 20.0) ++o.c_base_q20;
			if (q >= 30.0) ++o.c_base_q30;
			if (q >= (int)o.base_qualities.size()) throw Error("Base quality > 100 (" + std::to_string(q) + "). This should not happen!");
			o.base_qualities[(size_t)q]++;
			if (is_forward) o.qualities1[(size_t)i] += q; else o.qualities2[(size_t)i] += q;
		}
		const double mean_qscore = q_sum / cycles;
		if (mean_qscore == mean_qscore)
		{
			o.read_qualities[(size_t)std::round(mean_qscore)]++;
			if (is_forward) o.qscore_dist_r1.inc(mean_qscore, true); else o.qscore_dist_r2.inc(mean_qscore, true);
			if (mean_qscore >= 20.0) ++o.c_read_q20;
		}
	}
	return o;
}

} // namespace orc
