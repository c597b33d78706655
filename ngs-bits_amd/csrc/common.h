// Shared declarations of the HIP hot path (libngsqc_hip.so). gfx950 only — no CUDA paths, no compat shims.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <cctype>
#include <memory>
#include <vector>
#include "../../include/ngsqc.h"
#include "k1_types.h"

namespace ngsqc {

constexpr int K2_REL_STRIDE = 512;   // record offsets (u16, relative to the entry) kept per BGZF member for K2's write pass: an entry of a member cut into 2^ksh pieces owns K2_REL_STRIDE >> ksh of them
constexpr int K2_MAX_KSH = 3, K2_MIN_KSH = -4;   // long reads: groups of sixteen members, the guess of a group by a workgroup of eight waves (index.hip; per 150 k reads K2 takes 0.50 ms - groups of four: 0.84, and 1.26 with a wave per group)        // a member is walked by up to 8 threads (round 5: several walkers per member, see entry_range)
constexpr int NAME_SHIFT = 12;       // a record of a tile scanned by the chain walk is named (entry << NAME_SHIFT | k): a 64 KiB member holds at most 65536 / 36 records

// ---- K1 ----
// two-phase K1 (k1_kernels.h / inflate.hip): lane-per-member Huffman -> token groups in pages of a pool, then wave-per-member LZ77 resolve
void launch_huff_tokens(const uint8_t* d_comp, const BlockDesc* d_blocks, int64_t n_blocks, BlockStatus* d_status,
                        uint32_t* d_pool, uint32_t pool_pages, uint32_t* d_pool_ctr /* zeroed */, uint32_t* d_tok_first, uint32_t* d_tok_count,
                        unsigned long long* d_work /* zeroed */, const uint32_t* d_order /* queue order inside the launch, or null */, int max_wgs, hipStream_t s);
void launch_lz77_resolve(const BlockDesc* d_blocks, int64_t n_blocks, uint8_t* d_out, BlockStatus* d_status,
                         const uint32_t* d_pool, const uint32_t* d_tok_first, const uint32_t* d_tok_count, const uint8_t* d_comp, hipStream_t s);

// BAI index on the host (bai.hip)
bool bai_range(const std::string& bam_path, const ngsqc_region* regions, int64_t n, int32_t n_ref, uint64_t& beg_voff, uint64_t& end_voff, bool& found);
bool bai_ranges(const std::string& bam_path, const ngsqc_region* regions, int64_t n, int32_t n_ref, uint64_t* beg, uint64_t* end);   // per region; end == 0: nothing can overlap

// BAI construction (bai.hip): what `samtools index` (htslib sam_index_build: hts_idx_push / hts_idx_finish) writes for a coordinate-sorted BAM.
// The device turns the records of a tile into RUNS (consecutive records of one (reference, bin)), per-reference mapped / unmapped counts and the
// linear index (minimum record start per 16 kb window, in inflated-stream offsets); the host turns offsets into virtual offsets and applies htslib's
// chunk rules to the runs (bai_assemble).
struct BaiRun { int64_t u; int32_t tid; uint32_t bin; int32_t pos; uint32_t kind; };   // u: inflated offset of the run's first record; kind 1: the LAST record of a tile (pos clamped at 0; for the order check across tiles)
enum { BAI_F_UNSORTED = 1, BAI_F_BAD_TID = 2, BAI_F_TOO_FAR = 4, BAI_F_WINDOWS = 8 };
void launch_bai_keys(const uint8_t* infl, const int64_t* recoff, int64_t n_rec, int32_t n_ref, int min_shift, int depth, uint64_t* d_key, uint64_t* d_wnd /* first | last << 32 window */,
                     unsigned long long* d_counts /* [(n_ref + 1)][2] */, unsigned long long* d_flags, hipStream_t s);
void launch_bai_runs(const uint8_t* infl, const int64_t* recoff, int64_t n_rec, int64_t u_base, const uint64_t* d_key, const uint64_t* d_wnd, const int64_t* d_first /* [n_ref + 1] */,
                     unsigned long long* d_lidx, BaiRun* d_runs, unsigned long long* d_nruns, unsigned long long* d_flags, hipStream_t s);
struct BaiRunV { uint64_t voff; int32_t tid; uint32_t bin; int32_t pos; uint32_t kind; };
// runs in file order; lidx: virtual offsets (or ~0) for the windows [first[t], first[t + 1]) of reference t; counts: (mapped, unmapped) per reference, then the
// reads without a reference. Returns an error text ("" = written).
std::string bai_assemble(const std::string& out_path, int32_t n_ref, uint64_t offset0, uint64_t final_off, const std::vector<BaiRunV>& runs, const std::vector<uint64_t>& lidx,
                         const std::vector<int64_t>& first, const std::vector<int64_t>& counts, bool csi = false, int min_shift = 14, int depth = 5);

// A chromosome name as the reference compares it (Chromosome::normalizedStringRepresentation, src/cppNGS/Chromosome.cpp:133-190): "chr" / "CHR" dropped, M -> MT,
// upper case. Every place that turns a region's name into a reference id of the file uses this one (BAM index queries and CRAM slice selection: ADVICE r04)
inline std::string chr_norm(std::string c)
{
	if (c.size() > 3 && (c.compare(0, 3, "chr") == 0 || c.compare(0, 3, "CHR") == 0)) c = c.substr(3);
	if (c == "M") c = "MT";
	for (auto& ch : c) ch = (char)toupper((unsigned char)ch);
	return c;
}

// CRAM 3.0 input (cram.hip; host only): the file as a BAM stream / as a BGZF image with stored blocks that the BAM path takes like any other BAM
bool is_cram(const uint8_t* d, size_t n);
void cram_set_reference(const char* fasta);
std::string cram_reference();
int cram_set_skip_thread(int flags);   // the calling thread's own choice (-1: none); returns the previous one
void cram_set_skip(int flags);   // bit 0: read names, bit 1: optional fields are not needed (not decoded where their blocks are theirs alone)
int cram_skip();
struct CramSelect { struct Region { std::string chr; int32_t start, end; }; std::vector<Region> regions; int64_t max_slices = 0; };   // regions: only slices that can hold their records; max_slices: the first slices only
// The quality arrays of a CRAM (QS series: one rANS 4x8 block per slice, about half of a BAM record's bytes) can stay compressed on the host: the plan names every such
// block (where its four rANS states start in the CRAM image, its frequency tables in a compact form) and, per record, where its qualities go in the BAM stream; the
// device decodes the blocks and writes the qualities into the uploaded image (cram_dev.hip).
struct CramQualPlan
{
	struct Job { uint64_t in_off; uint64_t out_off; uint32_t in_len, n_out, tab_off, sym_off; uint32_t order, nsym; };   // in_off: the states + byte stream in the CRAM image; out_off: into the decoded quality bytes of all jobs
	struct Patch { uint64_t dst, src; uint32_t len, pad; };                                                       // dst: offset in the BAM stream; src: offset in the decoded quality bytes
	std::vector<Job> jobs; std::vector<uint16_t> tabs; std::vector<uint8_t> syms; std::vector<Patch> patches; uint64_t out_bytes = 0;
	// tabs: per job (order 0: one row; order 1: nsym rows, row = index of the previous symbol) of nsym + 1 cumulative frequencies; syms: per job 64 symbols + 256 bytes "byte -> index"
};
// a byte buffer that is NOT zeroed when it is made (a BAM image of a WGS CRAM is ~100 GB: the workers that fill it touch its pages, in parallel)
struct ByteImage
{
	std::unique_ptr<uint8_t[]> p; size_t n = 0;
	void make(size_t bytes) { p.reset(new uint8_t[bytes ? bytes : 1]); n = bytes; }
	uint8_t* data() { return p.get(); }
	const uint8_t* data() const { return p.get(); }
	size_t size() const { return n; }
};
// the CRAM as a BAM IMAGE: header + records in BGZF members of 65 280 bytes with stored blocks (+ the EOF member), which the BAM path takes like any other BAM
int cram_to_bam_image(const uint8_t* d, size_t n, const std::string& path, ByteImage& image, std::string& err, const CramSelect* sel = nullptr, CramQualPlan* defer = nullptr);   // NGSQC_OK or an NGSQC_E_* code with err
// decodes the plan's blocks on the device and writes the qualities into the BAM image (stored BGZF members of 65 280 bytes, as bgzf_store lays them out) at d_image; returns the kernel time in ms
double cram_device_quals(const uint8_t* cram_image, const CramQualPlan& plan, uint8_t* d_image, size_t image_bytes, hipStream_t s);

void k1_read_switches();   // NGSQC_P1_PARK (read when a handle is opened)

// CRC32 of every inflated member against its BGZF trailer (crc.hip); a mismatch sets status.error = K1_ERR_CRC
void launch_crc32(const BlockDesc* d_blocks, int64_t n_blocks, const uint8_t* d_out, const uint32_t* d_expected, BlockStatus* d_status, hipStream_t s);

// ---- K2 ----
// (n_entries = members << ksh, + 1 for the carried prefix; every array is indexed by entry)
void launch_index_count(const uint8_t* d_infl, int64_t total, const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, int64_t from, int32_t* d_start,
                        uint32_t* d_cnt, int64_t* d_next_abs, uint32_t* d_bad, int32_t n_ref, uint16_t* d_rel, hipStream_t s, bool wave_guess = true);
void launch_index_init(const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, int64_t exp0, bool guess_all, bool assume0 /* members start with a record: no guess for their first piece */, int32_t* d_start, hipStream_t s);
void launch_index_chain(const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, int64_t exp0, int64_t total, const int32_t* d_start, const int64_t* d_next, uint32_t* d_viol, long long* d_straddle /* -1 before the launch */, hipStream_t s);
void launch_index_write(const uint8_t* d_infl, int64_t total, const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, const int32_t* d_start,
                        const uint32_t* d_cnt, const int64_t* d_base, const uint16_t* d_rel, int64_t* d_recoff, hipStream_t s);
void launch_scan_counts(const uint32_t* d_cnt, int64_t n, int64_t* d_base, void* d_tmp, hipStream_t s);
void launch_depth_prefix(int32_t* d_diff, int64_t n_slots, void* d_tmp, hipStream_t s);
size_t scan_tmp_bytes(int64_t n);

// ---- K3-K5 ----
constexpr int GC_NMAX = 64;      // (bin, n) integer hit table for reads overlapping n < GC_NMAX GC chunks
constexpr int LONG_CIGAR = 64;   // records with more CIGAR ops go to the wave-per-record kernel
constexpr int MODE_DEPTH = 3;
constexpr int MODE_COUNT = 4;   // BedReadCount: reads per target region (src/BedReadCount/main.cpp:33-71)

// device accumulator slots (scan.hip); the host side of the library maps them onto NGSQC_C_*
enum { A_TOTAL, A_MAPPED, A_ONTARGET, A_NEAR, A_DUP, A_PP, A_INS_CNT, A_SUM_LEN, A_BASES_MAPPED, A_CLIPPED, A_INS_SUM,
       A_USABLE, A_NO_OVERLAP, A_USABLE_RAW, A_USABLE_ROI, A_DP0, A_DP1, A_DP2, A_DP3, A_DP4, A_DD0, A_DD1, A_DD2, A_DD3,
       A_READS_X, A_READS_Y, A_ALG_BYTES, A_COUNT,
       A_MAX_LEN = A_COUNT, A_FIRST_MAX_KEY, A_FIRST_PAIRED, A_LONG_COUNT, A_FIX_TRIM, A_FIX_LEN, A_FIX_CARRY, A_FIX_CNT,
       A_TILE_KEY, A_TILE_PAIRED,   // the fused walk + scan: (longest read, first record) and first paired record of the tile, in (entry, k) form
       A_HIST0, A_DEV_TOTAL = A_HIST0 + 1000 };

struct ScanParams
{
	int32_t mode;            // NGSQC_MODE_* or MODE_DEPTH
	int32_t min_mapq, min_baseq, skip_mismapped;
	int32_t tid_x, tid_y, n_ref;
	int64_t len_x, len_y;
	const uint8_t* tid_nonspecial;     // [n_ref]
	// target regions (merged, sorted by tid then start) + per-tid index ranges [first,last)
	const int32_t* reg_start; const int32_t* reg_end; const int64_t* reg_doff; // doff: first slot of the region in the diff array (len+1 slots each)
	const int32_t* tid_reg_first; const int32_t* tid_reg_last;
	int64_t n_regions;
	// GC chunks (roi.chunk(100))
	const int32_t* gc_start; const int32_t* gc_end; const int32_t* gc_bin;
	const int32_t* tid_gc_first; const int32_t* tid_gc_last; int64_t n_gc;
	// data
	const uint8_t* infl; int64_t total; const int64_t* recoff; int64_t n_rec;   // the resident tile (offsets are tile-local)
	int64_t ord_base;              // file ordinal of the tile's first record
	// outputs
	unsigned long long* counters;  // [A_DEV_TOTAL]
	int32_t* diff;                 // difference array -> depth
	unsigned long long* region_reads;   // MODE_COUNT: reads overlapping each region
	unsigned long long* gc_tab;    // [101][GC_NMAX]
	double* gc_over;               // [101]
	int64_t* long_list; int64_t long_cap;
	// the scan fused into K2's chain walk (launch_walk_scan): records are named (entry << NAME_SHIFT | k) until the counts are scanned (entry_base != null: the
	// long list holds such names); sgn = -1 takes a tile's contributions back (a tile that turned out not to be laid out like an htslib file)
	// the site pileup of the job riding the same walk: records whose span holds a known site leave their offset in list (list == nullptr: no pileup rides)
	struct Pile { const int32_t* site_pos = nullptr; const int32_t* tid_first = nullptr; const int32_t* tid_last = nullptr; const int32_t* bucket = nullptr; const int64_t* tid_bucket0 = nullptr;
	              int64_t* list = nullptr; unsigned long long* count = nullptr; int64_t cap = 0; int32_t min_mapq = 0, include_npp = 0; } pile;
	// MODE_DEPTH with min_baseq riding the walk (round 5): a record that overlaps a region leaves its offset here, a wave per record masks its low-quality bases afterwards
	int64_t* bq_list = nullptr; unsigned long long* bq_count = nullptr; int64_t bq_cap = 0;
	const int64_t* entry_base = nullptr; int32_t sgn = 1; int32_t tile_slots = 0; int64_t scan_limit = INT64_MAX;   // scan_limit: (a shard) records that start at or behind this tile-local offset are walked, not scanned
};

void launch_scan(const ScanParams& p, hipStream_t s);
// K2's chain walk (what launch_index_count does) with the scan of every record the walk passes: one read of a record's first line instead of two
void launch_walk_scan(const ScanParams& p, const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, int32_t* d_start, uint32_t* d_cnt, int64_t* d_next_abs, uint32_t* d_bad, uint16_t* d_rel, hipStream_t s);
void launch_index_guess(const uint8_t* d_infl, int64_t total, const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, int64_t from, int32_t* d_start, int32_t n_ref, hipStream_t s);
void launch_scan_long(const ScanParams& p, int64_t n_long, hipStream_t s);
void launch_baseq_list(const ScanParams& p, int64_t n, hipStream_t s, int64_t* d_sorted, void* d_tmp, size_t tmp_bytes);   // the min_baseq mask of the n records in p.bq_list (sorted by offset into d_sorted first)
size_t baseq_sort_bytes(int64_t n);
void launch_prefix_fix(const ScanParams& p, int64_t upto_max, int64_t upto_paired, const uint32_t* d_head /* captured records, or null: read the resident tile */, hipStream_t s,
                       uint32_t* d_scratch = nullptr /* prefix_fix_scratch_words(max(upto_max, upto_paired)) words: the parallel form; null: one workgroup */);
size_t prefix_fix_scratch_words(int64_t upto);
void launch_prefix_capture(const ScanParams& p, int64_t n, uint32_t* d_head, hipStream_t s);

// site pileup (BamReader::getPileup SNP counts for a table of sites; counts = u32[n_sites][8])
constexpr int PILEUP_BUCKET_SHIFT = 16;   // 64 kb position buckets per reference: bucket -> first site at or behind the bucket start
void launch_pileup(const uint8_t* infl, const int64_t* recoff, int64_t n_rec, int n_ref, const int32_t* site_pos, const int32_t* tid_first, const int32_t* tid_last,
                   const int32_t* bucket, const int64_t* tid_bucket0, int min_mapq, int min_baseq, int include_npp, uint32_t* counts,
                   int64_t* long_list, unsigned long long* long_count, hipStream_t s);
void launch_pileup_long(const uint8_t* infl, const int64_t* recoff, const int64_t* long_list, const unsigned long long* d_n_long /* count on the device */, int64_t n_long_max, const int32_t* site_pos, const int32_t* tid_last,
                        const int32_t* bucket, const int64_t* tid_bucket0, int min_baseq, uint32_t* counts, hipStream_t s);

// ---- raw-read QC pass (reads.hip): StatisticsReads::update(BamAlignment) ----
constexpr int RQ_PASSES = 5, RQ_CYC = RQ_PASSES * 64;   // per-cycle statistics are kept for the first 320 cycles (every Illumina read length)
enum { RA_FWD, RA_REV, RA_BASES, RA_A, RA_C, RA_G, RA_T, RA_N, RA_BAD_BASE, RA_BAD_QUAL, RA_BQ0, RA_RQ0 = RA_BQ0 + 100, RA_QD0 = RA_RQ0 + 100, RA_TOTAL = RA_QD0 + 120 };
void launch_reads_max(const uint8_t* infl, const int64_t* recoff, int64_t n_rec, unsigned long long* d_max, hipStream_t s);
void launch_reads(const uint8_t* infl, const int64_t* recoff, int64_t n_rec, int single_end, unsigned long long* d_acc, unsigned long long* d_len_hist, int64_t len_cap,
                  unsigned long long* d_cyc /* [RQ_CYC][7]: A,C,G,T,N, quality sum forward, quality sum reverse */, hipStream_t s);

// ---- K6 ----
void launch_depth_add(int32_t* d_dst, const int32_t* d_src, int64_t n, hipStream_t s);
void launch_depth_mark_spare(int32_t* d_depth, const int64_t* d_doff, const int32_t* d_len, int64_t n_regions, hipStream_t s);
void launch_depth_hist(const int32_t* d_depth, int64_t n_slots, int32_t cap, int64_t half, unsigned long long* d_hist, unsigned long long* d_cov, hipStream_t s);
void launch_depth_compact(const int32_t* d_depth, const int64_t* d_doff, const int32_t* d_len, int64_t n_regions, int32_t* d_out, hipStream_t s);
void launch_line_sums(const int32_t* d_depth, const int64_t* d_slot, const int32_t* d_n, int64_t n_lines, long long* d_sums, hipStream_t s);
void launch_line_runs(bool write, const int32_t* d_depth, const int64_t* d_slot, const int32_t* d_n, const int32_t* d_line_start, int64_t n_lines,
                      int32_t cutoff, int32_t is_high, int32_t sat, uint32_t* d_cnt, const int64_t* d_base, ngsqc_run* d_runs, hipStream_t s);

} // namespace ngsqc

#ifdef __HIPCC__
namespace ngsqc {
// Entries of a tile: entry 0 is the pseudo member that covers the bytes carried over from the previous tile ([0, prefix)); the entries behind it are the
// tile's members (static descriptor table, upos relative to the tile's first member), each cut into 2^ksh pieces of equal size: entry e >= 1 is piece
// (e - 1) & (2^ksh - 1) of member (e - 1) >> ksh. Round 5: the chain walk is a dependent chain of sparse line fetches (one 128-byte line per record, ~190
// records per member) - with one thread per member a tile kept 1.9 waves per SIMD busy and 77 % of their cycles waited for memory. A piece in the middle
// of a member does not know where its first record starts: the guess kernel finds the first plausible record header (index.hip), every piece's walker runs
// to the end of its piece, and the chain is accepted only if every walker's exit IS the next walker's start (index_chain_kernel) - by induction from the
// tile's known first record every start then lies on the true chain; anything else takes the general path.
// ksh < 0 (long reads, round 5): an entry is a GROUP of 2^-ksh consecutive members (nm = members of the tile): a 20 kb read spans a third of a member, so
// most members hold no record start at all and a guess per member reads the whole tile once more; a walker per megabyte follows ~35 records and its guess
// looks through half a record.
__device__ __forceinline__ void entry_range(const BlockDesc* __restrict__ blocks, int64_t e, int64_t prefix, int ksh, int64_t nm, int64_t& lo, int64_t& hi)
{
	if (e == 0) { lo = 0; hi = prefix; }
	else if (ksh >= 0)
	{
		const int64_t idx = e - 1; const BlockDesc bd = blocks[idx >> ksh]; const int64_t j = idx & ((1ll << ksh) - 1);
		const int64_t mlo = prefix + (int64_t)bd.upos;
		lo = mlo + (((int64_t)bd.usize * j) >> ksh); hi = mlo + (((int64_t)bd.usize * (j + 1)) >> ksh);
	}
	else
	{
		const int64_t m0 = (e - 1) << -ksh, m1 = min(m0 + (1ll << -ksh), nm) - 1;
		const BlockDesc a = blocks[m0], z = blocks[m1];
		lo = prefix + (int64_t)a.upos; hi = prefix + (int64_t)z.upos + z.usize;
	}
}
// offsets (u16, relative to the entry) the walk keeps per entry for K2's write pass; 0: none (a group of members is longer than 64 KiB - its chain is walked again)
__device__ __forceinline__ uint32_t rel_stride(int ksh) { return ksh >= 0 ? (uint32_t)K2_REL_STRIDE >> ksh : 0u; }
// what htslib's bam_read1 checks before it accepts a record (the reference then throws "Could not read next alignment",
// src/cppNGS/BamReader.h:389-392): the variable-length fields must fit the record. Kernels behind K2 trust these fields.
__device__ __forceinline__ bool record_fields_fit(uint32_t l_name, uint32_t n_cigar, int32_t l_seq, uint32_t bs)
{
	if (l_seq < 0 || l_name == 0) return false;
	return 32ull + l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq <= (uint64_t)bs;
}
}
#endif

// after every kernel launch: a rejected launch (bad grid, too much LDS) must not pass as an empty result
#define KCHECK() do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) { throw std::runtime_error(std::string("HIP kernel launch failed: ") + hipGetErrorString(_e)); } } while (0)
#define HIPCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr); } } while (0)
