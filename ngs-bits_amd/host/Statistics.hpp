// Host mirror of the hot-path part of the reference's Statistics class (src/cppNGS/Statistics.h:40-46,53,66-70):
// same static entry points, argument meaning and error messages; the per-read work is delegated to libngsqc_hip.so.
#pragma once
#include <memory>
#include "core.hpp"
#include "../../include/ngsqc.h"

namespace ngsbits {

// RAII wrapper of one open BAM on the GPU (role of BamReader for this path: BamReader.h:350-455)
// BamReader.h:338-347
struct BamInfo { std::string file_format, build; bool false_duplications_masked = true, contains_alt_chrs = false, paired_end = false; std::string mapper, mapper_version; };

class BamReader
{
public:
	// allow_shards: honour NGSQC_SHARDS=N (an extension): the BAM is split into N BGZF-member ranges, one handle each, spread
	// round-robin over the visible GPUs (mapping scans: shard protocol of include/ngsqc.h; depth scans: summed difference arrays).
	BamReader(const std::string& bam_file, const std::string& ref_genome = "", bool allow_shards = false);
	// Region queries (BamReader::setRegion, BamReader.cpp:734-768): only the BGZF blocks the BAI names for the lines of `regions` are sent to the GPU
	// (ngsqc_open_regions). Falls back to the whole file without an index next to the BAM, with NGSQC_SHARDS or NGSQC_INDEX_SELECT=0.
	BamReader(const std::string& bam_file, const std::string& ref_genome, bool allow_shards, const BedFile& regions);
	// the first records only (BamReader::info): header members + n_members BGZF members
	struct Head { int64_t n_members; };
	BamReader(const std::string& bam_file, const std::string& ref_genome, Head head);
	BamInfo info();   // BamReader.cpp:593-730
	// BamReader::skipTags() of the reference (BamReader.cpp:525-572: htslib's CRAM_OPT_REQUIRED_FIELDS): what the readers opened from now on need of a CRAM's
	// records. No function of this path reads a read name; only Statistics::mapping(bed ...) reads a tag (DP, cfDNA). BAM input is not affected.
	static void requireTags(bool needed);
	~BamReader();
	BamReader(const BamReader&) = delete; BamReader& operator=(const BamReader&) = delete;
	const std::vector<Chromosome>& chromosomes() const { return chrs_; }
	int chromosomeID(const Chromosome& chr) const;              // tid or -1
	int chromosomeSize(const Chromosome& chr) const;
	double genomeSize(bool include_special_chromosomes) const; // BamReader.cpp:789-800
	ngsqc_handle* handle() const { return h_; }                 // shard 0 when sharded (holds the combined depth array after a scan)
	const std::vector<ngsqc_handle*>& shards() const { return shards_; }   // all shard handles in file order (size 1 when not sharded)
	const std::string& fileName() const { return bam_file_; }
	void requireIndex() const;                                  // setRegion's "Could not load index" (BamReader.cpp:742-746)
	void check(int rc) const;
private:
	void init(const std::string& ref_genome, bool allow_shards, const BedFile* regions, int64_t head_members);
	int64_t head_members_ = 0;   // > 0: a head open (BamReader::info grows it when the first members do not answer its question)
	std::string bam_file_, ref_file_; ngsqc_handle* h_ = nullptr; std::vector<ngsqc_handle*> shards_; std::vector<Chromosome> chrs_; std::vector<long long> sizes_;
};

// MappingQC reads the BAM up to four times in the reference (src/MappingQC/main.cpp:83-165: read QC, mapping, contamination, somatic
// sub-panel). A FusedPlan announces the passes that follow the mapping pass: the next Statistics::mapping*() call on that BAM then runs
// ONE GPU job (ngsqc_run_job: every BGZF member is inflated once, all consumers see each tile) and the announced functions take their
// counts from it instead of reading the BAM again. Without a plan (or with NGSQC_FUSED=0, or a sharded reader) each function runs its own pass.
struct FusedPlan
{
	bool contamination = false; std::string build, roi_file; bool include_not_properly_paired = false;
	bool read_qc = false, single_end = false;
	bool somatic = false; BedFile somatic_bed; int somatic_min_mapq = 1;
};

// Statistics.h: result of the gender estimates (SampleGender)
struct GenderEstimate { std::string gender; std::vector<std::pair<std::string, std::string>> add_info; };

class Statistics
{
public:
	// Statistics.cpp:2811 / 2836 / 2885 — SampleGender -method xy | hetx | sry on the same GPU passes (chrX / chrY read counts of the
	// mapping scan, site pileup of the known chrX SNVs outside the pseudo-autosomal regions, depth of the SRY gene)
	static GenderEstimate genderXY(const std::string& bam_file, double max_female = 0.06, double min_male = 0.09, const std::string& ref_file = "");
	static GenderEstimate genderHetX(const std::string& build, const std::string& bam_file, double max_male = 0.05, double min_female = 0.25, const std::string& ref_file = "", bool include_not_properly_paired = false);
	static GenderEstimate genderSRY(const std::string& build, const std::string& bam_file, double min_cov = 20.0, const std::string& ref_file = "");
	static void planFused(const std::string& bam_file, const FusedPlan& plan);
	static void clearFused();
	// Statistics.cpp:343  — target-region mode (MappingQC -roi)
	static QCCollection mapping(const BedFile& bed_file, const std::string& bam_file, const std::string& ref_file, int min_mapq = 1, bool is_cfdna = false);
	// Statistics.cpp:805  — -rna / -wgs -build non_human
	static QCCollection mapping(const std::string& bam_file, const std::string& ref_file, int min_mapq = 1);
	// Statistics.cpp:990  — -wgs with the embedded OMIM ROI
	static QCCollection mapping_wgs(const std::string& bam_file, const std::string& bedpath, int min_mapq, const std::string& ref_file);
	// Statistics.cpp:1574  — MappingQC -somatic_custom_bed: depth metrics on a sub-panel
	static QCCollection somaticCustomDepth(const BedFile& bed_file, const std::string& bam_file, const std::string& ref_file, int min_mapq = 1);
	// Statistics.cpp:2333  — sample contamination check on the known common SNPs (MappingQC default, switched off by -no_cont)
	static QCCollection contamination(const std::string& build, const std::string& bam, const std::string& ref_file, const std::string& roi_file = "", bool debug = false, int min_cov = 20, int min_snps = 50, bool include_not_properly_paired = false);
	// Statistics.cpp:2698 / 2693 / 2806
	static void avgCoverage(BedFile& bed_file, const std::string& bam_file, int min_mapq = 1, int threads = 1, int decimals = 2, const std::string& ref_file = "", bool random_access = false, bool skip_mismapped = false, bool debug = false);
	static BedFile lowCoverage(const BedFile& bed_file, const std::string& bam_file, int cutoff, int min_mapq = 1, int min_baseq = 0, int threads = 1, const std::string& ref_file = "", bool random_access = true, bool debug = false);
	static BedFile highCoverage(const BedFile& bed_file, const std::string& bam_file, int cutoff, int min_mapq = 1, int min_baseq = 0, int threads = 1, const std::string& ref_file = "", bool random_access = true, bool debug = false);
	// Statistics.cpp:2904-2922
	static void addQcValue(QCCollection& output, const std::string& accession, const std::string& name, double value);
	static void addQcValue(QCCollection& output, const std::string& accession, const std::string& name, const std::string& value);
	static void addQcPlot(QCCollection& output, const std::string& accession, const std::string& name, const std::vector<double>& x, const std::vector<std::vector<double>>& lines);
private:
	static BedFile lowOrHighCoverage(const BedFile& bed_file, const std::string& bam_file, int cutoff, int min_mapq, int min_baseq, bool is_high, bool random_access, const std::string& ref_file = "");
};

// Raw-read QC of a BAM (src/cppNGS/StatisticsReads.h; MappingQC -read_qc, src/MappingQC/main.cpp:83-98). The reference feeds
// every alignment to update(const BamAlignment&); here update(reader) runs that loop over the whole BAM on the GPU
// (ngsqc_scan_reads) and getResult() is the reference's post-processing of the same counters.
class StatisticsReads
{
public:
	explicit StatisticsReads(bool single_end = false) : single_end_(single_end) {}
	void update(BamReader& reader);
	bool takeFused(const std::string& bam_file);   // the counts of the fused job announced with Statistics::planFused (false: none, run update())
	QCCollection getResult();
private:
	bool single_end_; ngsqc_read_stats st_{}; std::vector<int64_t> read_lengths_, cycles_; bool have_ = false;
};

// ref_file == NO_REF: run without a reference genome (an extension for genome-less test boxes; GC/AT dropout become
// "n/a" and the N-base correction of the WGS depth denominators is skipped). The reference always needs a FASTA.
extern const char* const NO_REF;

} // namespace ngsbits
