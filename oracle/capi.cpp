// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY (see bamio.hpp header).
// Plain C entry points so tests/ (ctypes) can drive the CPU restatement. Nothing in the product links this.
// Counter vector layout (int64[ORC_NCOUNTERS]) is the same order as include/ngsqc.h NGSQC_C_* so that tests
// can compare the two arrays element by element.
// ============================================================================
#include "stats.hpp"
#include "stream.hpp"
#include <chrono>

using namespace orc;

namespace {
struct Result
{
	MappingResult m;
	std::string text;       // accession \t name \t value \t is_plot per line
	std::string bed_text;   // coverage tools: output BED as written by BedFile::store (without tool headers)
	std::vector<int64_t> cov;
	std::vector<int32_t> depth;
	double seconds_load = 0, seconds_compute = 0;
};
void seterr(char* err, int n, const std::string& s) { if (err && n>0) { snprintf(err, (size_t)n, "%s", s.c_str()); } }
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}

extern "C" {

enum { ORC_NCOUNTERS = 1032 };

void* orc_bam_load(const char* path, char* err, int errlen)
{
	try { auto* b = new BamFile(); b->load(path); return b; }
	catch (std::exception& e) { seterr(err, errlen, e.what()); return nullptr; }
}
void orc_bam_free(void* b) { delete (BamFile*)b; }
int64_t orc_bam_count(void* b) { return (int64_t)((BamFile*)b)->count(); }
int64_t orc_bam_inflated_size(void* b) { return (int64_t)((BamFile*)b)->data.size(); }
int64_t orc_bam_first_record_offset(void* b) { return (int64_t)((BamFile*)b)->first_rec; }
int64_t orc_bam_n_blocks(void* b) { return (int64_t)((BamFile*)b)->n_blocks; }
int orc_bam_sorted(void* b) { return ((BamFile*)b)->sorted ? 1 : 0; }
int orc_bam_n_ref(void* b) { return (int)((BamFile*)b)->ref_names.size(); }
const char* orc_bam_ref_name(void* b, int i) { return ((BamFile*)b)->ref_names[i].c_str(); }
int64_t orc_bam_ref_len(void* b, int i) { return ((BamFile*)b)->ref_lens[i]; }
// copies the inflated stream (for checking the GPU inflate kernel)
int64_t orc_bam_inflated(void* b, uint8_t* out, int64_t cap)
{
	auto* f = (BamFile*)b; int64_t n = std::min<int64_t>(cap, (int64_t)f->data.size());
	memcpy(out, f->data.data(), (size_t)n); return n;
}
int64_t orc_bam_record_offsets(void* b, int64_t* out, int64_t cap)
{
	auto* f = (BamFile*)b; int64_t n = std::min<int64_t>(cap, (int64_t)f->rec_off.size());
	for (int64_t i=0;i<n;++i) out[i] = (int64_t)f->rec_off[i];
	return n;
}

// mode: 0 = Statistics::mapping(bed,...) [ROI], 1 = Statistics::mapping(bam,...) [no ROI], 2 = Statistics::mapping_wgs
// bed: path or NULL ; merge_bed: apply BedFile::merge() after load (MappingQC main.cpp:130-132) ; fasta: path or NULL
void* orc_mapping(void* bam, int mode, const char* bed, int merge_bed, const char* fasta, int min_mapq, int is_cfdna, char* err, int errlen)
{
	try
	{
		auto* r = new Result();
		std::unique_ptr<Fasta> fa; if (fasta && *fasta) fa.reset(new Fasta(fasta));
		BedFile roi; bool have = bed && *bed;
		if (have) { roi.load(bed); if (merge_bed) roi.merge(); }
		double t0 = now();
		if (mode==0) r->m = mapping_roi(roi, *(BamFile*)bam, fa.get(), min_mapq, is_cfdna!=0);
		else if (mode==1) r->m = mapping_noroi(*(BamFile*)bam, fa.get(), min_mapq);
		else r->m = mapping_wgs(*(BamFile*)bam, have ? &roi : nullptr, fa.get(), min_mapq);
		r->seconds_compute = now() - t0;
		for (auto& l : r->m.lines) r->text += l.accession + "\t" + l.name + "\t" + l.value + "\t" + (l.is_plot ? "1" : "0") + "\n";
		return r;
	}
	catch (std::exception& e) { seterr(err, errlen, e.what()); return nullptr; }
}

void orc_result_counters(void* res, int64_t* out)
{
	const MappingResult& m = ((Result*)res)->m;
	int64_t v[32] = { m.al_total, m.al_mapped, m.al_ontarget, m.al_neartarget, m.al_dup, m.al_proper_paired, m.insert_size_read_count,
		m.bases_trimmed, m.bases_mapped, m.bases_clipped, m.insert_size_sum, m.bases_usable, m.bases_usable_no_overlap, m.bases_usable_raw, m.bases_usable_roi,
		m.bases_usable_dp[0], m.bases_usable_dp[1], m.bases_usable_dp[2], m.bases_usable_dp[3], m.bases_usable_dp[4],
		m.dp_dist[0], m.dp_dist[1], m.dp_dist[2], m.dp_dist[3], m.max_length, m.paired_end, m.roi_bases, m.half_depth, m.bases_covered_half,
		m.reads_x, m.reads_y, m.yx_valid };
	memcpy(out, v, sizeof(v));
	memcpy(out+32, m.insert_hist, sizeof(m.insert_hist));
}
const char* orc_result_text(void* res) { return ((Result*)res)->text.c_str(); }
int64_t orc_result_depth(void* res, int32_t* out, int64_t cap)
{
	auto* r = (Result*)res; const std::vector<int32_t>& d = r->depth.empty() ? r->m.depth : r->depth;
	int64_t n = std::min<int64_t>(cap, (int64_t)d.size());
	if (out) memcpy(out, d.data(), (size_t)n*4);
	return (int64_t)d.size();
}
int orc_result_gc(void* res, double* gc_roi101, double* gc_reads101)   // bins 0..100 (bin 100: chunks of pure G/C)
{
	const MappingResult& m = ((Result*)res)->m;
	for (int i=0;i<101;++i) { gc_roi101[i] = (size_t)i<m.gc_roi.size() ? m.gc_roi[i] : 0; gc_reads101[i] = (size_t)i<m.gc_reads.size() ? m.gc_reads[i] : 0; }
	return m.have_gc ? 1 : 0;
}
double orc_result_seconds(void* res) { return ((Result*)res)->seconds_compute; }
void orc_result_free(void* res) { delete (Result*)res; }

// BedCoverage core: Statistics::avgCoverage. merge_bed: 0 none (tool behaviour), 1 = merge() first (as the unit tests do).
// Result: bed_text = lines with the appended coverage column; cov = raw per-line sums.
void* orc_avg_coverage(void* bam, const char* bed, int merge_bed, int min_mapq, int decimals, int random_access, int skip_mismapped, int clear, char* err, int errlen)
{
	try
	{
		auto* r = new Result();
		BedFile f; f.load(bed);
		if (clear) { f.headers.clear(); for (auto& l : f.lines) l.annos.clear(); }
		if (merge_bed) f.merge();
		double t0 = now();
		r->cov = avg_coverage(f, *(BamFile*)bam, min_mapq, decimals, random_access!=0, skip_mismapped!=0);
		r->seconds_compute = now() - t0;
		r->bed_text = f.toText(true);
		return r;
	}
	catch (std::exception& e) { seterr(err, errlen, e.what()); return nullptr; }
}
// BedReadCount core: load -> merge(false) (the tool's main) -> read counts; result: cov = counts, bed_text = the annotated BED
void* orc_read_counts(void* bam, const char* bed, int min_mapq, char* err, int errlen)
{
	try
	{
		auto* r = new Result();
		BedFile f; f.load(bed); f.merge(false);
		double t0 = now();
		r->cov = read_counts(f, *(BamFile*)bam, min_mapq);
		r->seconds_compute = now() - t0;
		r->bed_text = f.toText(false);
		return r;
	}
	catch (std::exception& e) { seterr(err, errlen, e.what()); return nullptr; }
}
int64_t orc_result_cov(void* res, int64_t* out, int64_t cap)
{
	auto* r = (Result*)res; int64_t n = std::min<int64_t>(cap, (int64_t)r->cov.size());
	if (out) memcpy(out, r->cov.data(), (size_t)n*8);
	return (int64_t)r->cov.size();
}
const char* orc_result_bed(void* res) { return ((Result*)res)->bed_text.c_str(); }

// BedLowCoverage / BedHighCoverage core. The tools load the BED and call merge(true,true) (src/BedLowCoverage/main.cpp:47-49).
void* orc_low_high_coverage(void* bam, const char* bed, int tool_merge, int cutoff, int min_mapq, int min_baseq, int is_high, int random_access, char* err, int errlen)
{
	try
	{
		auto* r = new Result();
		BedFile f; f.load(bed);
		if (tool_merge==1) f.merge(true, true); else if (tool_merge==2) f.merge();
		double t0 = now();
		BedFile out = low_high_coverage(f, *(BamFile*)bam, cutoff, min_mapq, min_baseq, is_high!=0, random_access!=0, &r->depth);
		r->seconds_compute = now() - t0;
		r->bed_text = out.toText(false);
		r->m.roi_bases = f.baseCount();
		r->cov.push_back((int64_t)f.count()); r->cov.push_back(f.baseCount()); r->cov.push_back((int64_t)out.count()); r->cov.push_back(out.baseCount());
		return r;
	}
	catch (std::exception& e) { seterr(err, errlen, e.what()); return nullptr; }
}

// bench.py cpu_baseline: streaming MappingQC -wgs loop over a BAM image in memory (see stream.hpp). Returns seconds.
// out_counters: int64[ORC_NCOUNTERS] ; out_stats: {n_records, inflated bytes, compressed bytes consumed}
double orc_baseline_wgs_stream(const uint8_t* image, int64_t n, const char* bed, int min_mapq, int64_t max_records, int64_t* out_counters, int64_t* out_stats, char* err, int errlen)
{
	try
	{
		BedFile roi; bool have = bed && *bed; if (have) roi.load(bed);
		StreamStats st; double t0 = now();
		Result r; r.m = mapping_wgs_stream(image, (size_t)n, have ? &roi : nullptr, min_mapq, max_records, st);
		double secs = now() - t0;
		if (out_counters) orc_result_counters(&r, out_counters);
		if (out_stats) { out_stats[0] = st.n_records; out_stats[1] = st.inflated; out_stats[2] = st.compressed; }
		return secs;
	}
	catch (std::exception& e) { seterr(err, errlen, e.what()); return -1.0; }
}

// The same with the contamination pileup of n_sites known sites (tid, 1-based pos) in the same pass: out_site_counts[6 * i ..] = A, C, G, T, N, deletion of site i
// (site_pileup()'s rules with min_mapq = site_min_mapq). The job the GPU step runs, timed as one loop.
double orc_baseline_wgs_stream_sites(const uint8_t* image, int64_t n, const char* bed, int min_mapq, int64_t max_records, const int32_t* tid, const int32_t* pos, int64_t n_sites,
                                     int site_min_mapq, int min_baseq, int include_not_properly_paired, int64_t* out_counters, int64_t* out_stats, int64_t* out_site_counts, char* err, int errlen)
{
	try
	{
		BedFile roi; bool have = bed && *bed; if (have) roi.load(bed);
		StreamSites ss; ss.min_mapq = site_min_mapq; ss.min_baseq = min_baseq; ss.include_not_properly_paired = include_not_properly_paired != 0; ss.counts = out_site_counts;
		for (int64_t i = 0; i < n_sites; ++i) ss.add(tid[i], pos[i], i);
		ss.finish();
		for (int64_t i = 0; i < 6 * n_sites; ++i) out_site_counts[i] = 0;
		StreamStats st; double t0 = now();
		Result r; r.m = mapping_wgs_stream(image, (size_t)n, have ? &roi : nullptr, min_mapq, max_records, st, 0, (size_t)-1, nullptr, &ss);
		double secs = now() - t0;
		if (out_counters) orc_result_counters(&r, out_counters);
		if (out_stats) { out_stats[0] = st.n_records; out_stats[1] = st.inflated; out_stats[2] = st.compressed; }
		return secs;
	}
	catch (std::exception& e) { seterr(err, errlen, e.what()); return -1.0; }
}

// GC bin of every roi.chunk(100) line (Statistics.cpp:363-387): what the host layer hands to the C ABI as gc_bin. merge_mode as in
// orc_bed_roundtrip (1 = merge(), 3 = sort + merge). Returns the number of chunks; bins[i] = floor(100 * gc) or -1 (no A/C/G/T in the chunk).
int64_t orc_gc_bins(const char* fasta, const char* bed, int merge_mode, int32_t* bins, int64_t cap, char* err, int errlen)
{
	try
	{
		Fasta fa(fasta);
		BedFile f; f.load(bed);
		if (merge_mode==1) f.merge(); else if (merge_mode==3) { f.sort(); f.merge(); }
		GcBins g(f, &fa);
		if (bins) for (int64_t i=0; i<std::min<int64_t>(cap, (int64_t)g.bin.size()); ++i) bins[i] = g.bin[(size_t)i];
		return (int64_t)g.bin.size();
	}
	catch (std::exception& e) { seterr(err, errlen, e.what()); return -1; }
}

// BED helpers for host-logic tests: load -> (merge) -> text
int64_t orc_bed_roundtrip(const char* bed, int merge_mode, char* out, int64_t cap, char* err, int errlen)
{
	try
	{
		BedFile f; f.load(bed);
		if (merge_mode==1) f.merge(); else if (merge_mode==2) f.merge(true, true); else if (merge_mode==3) { f.sort(); f.merge(); } else if (merge_mode==4) { f.merge(); f.chunk(100); } else if (merge_mode==5) { f.sort(); f.merge(); f.chunk(100); }
		std::string t = f.toText(false);
		if (out && cap>0) { size_t n = std::min<size_t>((size_t)cap-1, t.size()); memcpy(out, t.data(), n); out[n] = 0; }
		return (int64_t)t.size();
	}
	catch (std::exception& e) { seterr(err, errlen, e.what()); return -1; }
}

// Site pileup (BamReader::getPileup SNP counts) for a list of (tid, pos): out[6*i .. 6*i+5] = A, C, G, T, N, deletion.
int orc_site_pileup(void* bam, const int32_t* tid, const int32_t* pos, int64_t n, int min_mapq, int include_not_properly_paired, int min_baseq, int64_t* out, char* err, int errlen)
{
	try
	{
		for (int64_t i=0; i<n; ++i)
		{
			SiteCounts c = site_pileup(*(BamFile*)bam, tid[i], pos[i], min_mapq, include_not_properly_paired!=0, min_baseq);
			out[6*i] = c.a; out[6*i+1] = c.c; out[6*i+2] = c.g; out[6*i+3] = c.t; out[6*i+4] = c.n; out[6*i+5] = c.del;
		}
		return 0;
	}
	catch (std::exception& e) { seterr(err, errlen, e.what()); return -1; }
}
// Statistics::contamination on (tid, pos, ref, alt) known SNVs; writes the QC value string ("n/a" or "%.2f") to out.
int orc_contamination(void* bam, const int32_t* tid, const int32_t* pos, const char* ref, const char* alt, int64_t n, int include_not_properly_paired, char* out, int outlen, char* err, int errlen)
{
	try
	{
		std::vector<KnownSnp> snps((size_t)n);
		for (int64_t i=0; i<n; ++i) snps[(size_t)i] = KnownSnp{tid[i], pos[i], ref[i], alt[i]};
		std::string v = contamination_value(*(BamFile*)bam, snps, include_not_properly_paired!=0);
		snprintf(out, (size_t)outlen, "%s", v.c_str());
		return 0;
	}
	catch (std::exception& e) { seterr(err, errlen, e.what()); return -1; }
}

// Raw-read QC: out[0..7] = c_forward, c_reverse, bases_sequenced, c_read_q20, c_base_q20, c_base_q30, max_cycles, n_lengths;
// out[8..107] base_qualities, out[108..207] read_qualities, out[208..267] qscore_dist_r1, out[268..327] qscore_dist_r2,
// out[328..332] total A, C, G, T, N. len_hist[l] for l <= len_cap; cyc[7*i+k] = A,C,G,T,N,qsum fwd,qsum rev for i < n_cyc.
int orc_reads_qc(void* bam, int single_end, int64_t* out, int64_t* len_hist, int64_t len_cap, int64_t* cyc, int64_t n_cyc, char* err, int errlen)
{
	try
	{
		ReadsQc q = reads_qc(*(BamFile*)bam, single_end != 0);
		out[0] = q.c_forward; out[1] = q.c_reverse; out[2] = q.bases_sequenced; out[3] = q.c_read_q20; out[4] = q.c_base_q20; out[5] = q.c_base_q30;
		out[6] = (int64_t)q.pileups.size(); out[7] = (int64_t)q.read_lengths.size();
		for (int i=0;i<100;++i) { out[8+i] = q.base_qualities[(size_t)i]; out[108+i] = q.read_qualities[(size_t)i]; }
		for (int i=0;i<60;++i) { out[208+i] = (int64_t)q.qscore_dist_r1.binValue(i); out[268+i] = (int64_t)q.qscore_dist_r2.binValue(i); }
		for (int k=0;k<5;++k) { int64_t t = 0; for (auto& p : q.pileups) t += p[(size_t)k]; out[328+k] = t; }
		if (len_hist) { for (int64_t l=0;l<=len_cap;++l) len_hist[l] = 0; for (auto& kv : q.read_lengths) if (kv.first <= len_cap) len_hist[kv.first] = kv.second; }
		if (cyc) for (int64_t i=0;i<n_cyc;++i) for (int k=0;k<7;++k)
			cyc[7*i+k] = (size_t)i < q.pileups.size() ? (k<5 ? q.pileups[(size_t)i][(size_t)k] : (int64_t)(k==5 ? q.qualities1[(size_t)i] : q.qualities2[(size_t)i])) : 0;
		return 0;
	}
	catch (std::exception& e) { seterr(err, errlen, e.what()); return -1; }
}

// All-cores form of the bench baseline (see stream.hpp). out_stats: {n_records, inflated, compressed}. Returns seconds.
// out_counters (optional, int64[ORC_NCOUNTERS]): the threads' counters SUMMED - exact for the additive counters of an aligned BAM,
// meaningless for the order-dependent / maximum-like ones (bases_trimmed, bases_usable_no_overlap, max_length, paired_end, roi_bases,
// half_depth, bases_covered_half, yx_valid). out_hist (optional, int64[hist_cap + 1]): per-base depth histogram of the ROI (depths above
// the cap in the last bin) - the bench's parity check of the depth scatter at full size.
double orc_baseline_wgs_stream_mt(const uint8_t* image, int64_t n, const char* bed, int min_mapq, int threads, int64_t* out_stats, int64_t* out_counters, int64_t* out_hist, int hist_cap, char* err, int errlen)
{
	try
	{
		BedFile roi; bool have = bed && *bed; if (have) roi.load(bed);
		StreamStats st; std::vector<MappingResult> per; std::vector<int32_t> depth;
		double secs = mapping_wgs_stream_mt(image, (size_t)n, have ? &roi : nullptr, min_mapq, threads, st, out_counters ? &per : nullptr, out_hist ? &depth : nullptr);
		if (out_stats) { out_stats[0] = st.n_records; out_stats[1] = st.inflated; out_stats[2] = st.compressed; }
		if (out_counters)
		{
			std::vector<int64_t> tmp(ORC_NCOUNTERS);
			for (int i = 0; i < ORC_NCOUNTERS; ++i) out_counters[i] = 0;
			for (auto& m : per) { Result r; r.m = m; orc_result_counters(&r, tmp.data()); for (int i = 0; i < ORC_NCOUNTERS; ++i) out_counters[i] += tmp[(size_t)i]; }
		}
		if (out_hist) { for (int i = 0; i <= hist_cap; ++i) out_hist[i] = 0; for (int32_t d : depth) out_hist[d < hist_cap ? (d < 0 ? 0 : d) : hist_cap]++; }
		return secs;
	}
	catch (std::exception& e) { seterr(err, errlen, e.what()); return -1.0; }
}

} // extern "C"


