/*
 * ngsqc.h — C ABI of the MI355X-native BAM mapping-QC / coverage engine (libngsqc_hip.so).
 *
 * Drop-in boundary for ONE hot path of imgag/ngs-bits: the per-read loop behind
 *   Statistics::mapping(bed,...)      src/cppNGS/Statistics.h:40   (Statistics.cpp:343-803)
 *   Statistics::mapping(bam,...)      src/cppNGS/Statistics.h:42   (Statistics.cpp:805-988)
 *   Statistics::mapping_wgs           src/cppNGS/Statistics.h:44   (Statistics.cpp:990-1359)
 *   Statistics::avgCoverage           src/cppNGS/Statistics.h:70   (Statistics.cpp:2698-2804)
 *   Statistics::lowCoverage/highCoverage  Statistics.h:66,68       (Statistics.cpp:2534-2657,2693-2696,2806-2809)
 *   Statistics::yxRatio               Statistics.cpp:2659-2691
 * and the reader underneath them, BamReader::getNextAlignment / setRegion (src/cppNGS/BamReader.h:386-398,
 * BamReader.cpp:734-768), i.e. what the reference gets from htslib's sam_read1 / sam_itr_next / bam_endpos.
 *
 * The reference has no FFI for this path; the seam is the static C++ API above. The host layer
 * (ngs-bits_amd/host, plain C++17) keeps those function names/arguments/errors and calls into this ABI.
 * Everything here is plain C: pointers + sizes, caller-owned output buffers, int return codes
 * (0 = ok, <0 = error; message via ngsqc_last_error). No torch / HIP types cross the boundary.
 *
 * Coordinates: regions are 1-based, closed [start,end] (the in-memory convention of BedLine, BedFile.cpp:157-159).
 * Chromosomes are addressed by BAM reference id (tid); name -> tid mapping is host logic.
 */
#ifndef NGSQC_H
#define NGSQC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ngsqc_handle ngsqc_handle;

/* ---- error codes ---- */
#define NGSQC_OK            0
#define NGSQC_E_IO         -1   /* "Could not open BAM/CRAM file ..."            BamReader.cpp:465-468 */
#define NGSQC_E_FORMAT     -2   /* not BGZF / not BAM / corrupt record           BamReader.h:389-392  */
#define NGSQC_E_ARG        -3   /* invalid argument (ArgumentException cases)    */
#define NGSQC_E_DEVICE     -4   /* HIP runtime error / no device / out of memory */
#define NGSQC_E_UNSUPPORTED -5  /* CRAM 3.1 codecs other than rANS Nx16            */

/* ---- lifecycle (replaces BamReader ctor/dtor, BamReader.cpp:462-523) ---- */
int  ngsqc_open(const char* bam_path, int device, ngsqc_handle** out);
/* Same, from a BAM image already in host memory (bytes are copied to HBM; caller keeps ownership). */
int  ngsqc_open_memory(const void* bam_bytes, size_t n_bytes, int device, ngsqc_handle** out);
void ngsqc_close(ngsqc_handle* h);
/* ngsqc_open copies the compressed image to the device in the background (pieces in file order; the first job starts on the pieces that have
 * arrived - what the reference overlaps with htslib's reader thread, BamReader.cpp:472). This call returns when the whole image is on the device;
 * ngsqc_timings.h2d_ms is final behind it. NGSQC_ASYNC_H2D=0: ngsqc_open itself waits (ngsqc_open_memory always does: the caller owns the buffer). */
int  ngsqc_upload_wait(ngsqc_handle* h);
/* Message of the last failing call on h (or of the last failing ngsqc_open* when h is NULL). */
const char* ngsqc_last_error(const ngsqc_handle* h);

/* ---- header access (BamReader::chromosomes/chromosomeSize, BamReader.cpp:770-800) ---- */
int         ngsqc_n_ref(const ngsqc_handle* h);
const char* ngsqc_ref_name(const ngsqc_handle* h, int tid);
int64_t     ngsqc_ref_len(const ngsqc_handle* h, int tid);
int64_t     ngsqc_n_records(ngsqc_handle* h);     /* forces inflate + record indexing */
int64_t     ngsqc_inflated_size(ngsqc_handle* h); /* bytes of the inflated BGZF stream */
int64_t     ngsqc_n_bgzf_blocks(const ngsqc_handle* h);
int64_t     ngsqc_compressed_size(const ngsqc_handle* h);

/* ---- stage control ----
 * Inflate (K1) and record indexing (K2) run lazily and their result stays in HBM for subsequent scans on the
 * same handle (the reference re-reads the file for every pass). ngsqc_drop_decoded() frees/invalidates that
 * state so that the next scan redoes the whole job from the compressed bytes (what bench.py times). */
int ngsqc_decode(ngsqc_handle* h);
int ngsqc_drop_decoded(ngsqc_handle* h);
/* test hooks: copy device results back (caller buffers) */
int ngsqc_copy_inflated(ngsqc_handle* h, uint8_t* out, int64_t cap);
int ngsqc_copy_record_offsets(ngsqc_handle* h, int64_t* out, int64_t cap);

/* ---- regions ---- */
typedef struct ngsqc_region { int32_t tid; int32_t start; int32_t end; } ngsqc_region; /* 1-based closed */

/* ---- mapping QC scan (Statistics::mapping x2, mapping_wgs) ---- */
#define NGSQC_MODE_ROI    0   /* Statistics.cpp:343  target-region mode                 */
#define NGSQC_MODE_NOROI  1   /* Statistics.cpp:805  -rna / -wgs -build non_human       */
#define NGSQC_MODE_WGS    2   /* Statistics.cpp:990  -wgs, optional OMIM ROI second pass */

/* counter vector indices (int64). Doubles of the reference that only ever hold integers are kept as int64. */
enum {
	NGSQC_C_AL_TOTAL = 0, NGSQC_C_AL_MAPPED, NGSQC_C_AL_ONTARGET, NGSQC_C_AL_NEARTARGET, NGSQC_C_AL_DUP,
	NGSQC_C_AL_PROPER_PAIRED, NGSQC_C_INSERT_SIZE_READ_COUNT, NGSQC_C_BASES_TRIMMED, NGSQC_C_BASES_MAPPED,
	NGSQC_C_BASES_CLIPPED, NGSQC_C_INSERT_SIZE_SUM, NGSQC_C_BASES_USABLE, NGSQC_C_BASES_USABLE_NO_OVERLAP,
	NGSQC_C_BASES_USABLE_RAW, NGSQC_C_BASES_USABLE_ROI, NGSQC_C_BASES_USABLE_DP0, /* ..DP4 = +4 */
	NGSQC_C_DP_DIST0 = 20, /* ..3 */
	NGSQC_C_MAX_LENGTH = 24, NGSQC_C_PAIRED_END, NGSQC_C_ROI_BASES, NGSQC_C_HALF_DEPTH, NGSQC_C_BASES_COVERED_HALF,
	NGSQC_C_READS_X, NGSQC_C_READS_Y, NGSQC_C_YX_VALID,
	NGSQC_C_INSERT_HIST0 = 32, /* 1000 bins: reads per integer insert size 0..999 (binned on host, Statistics.cpp:401) */
	NGSQC_NCOUNTERS = 1032
};

typedef struct ngsqc_mapping_params {
	int32_t mode;                 /* NGSQC_MODE_*                                                        */
	int32_t min_mapq;             /* Statistics.cpp:343 min_mapq                                          */
	int32_t tid_x, tid_y;         /* tids of chrX / chrY or -1 (yxRatio, Statistics.cpp:2662-2666)        */
	const uint8_t* tid_nonspecial;/* [n_ref] 1 if Chromosome::isNonSpecial (Chromosome.h:78-81)           */
	const ngsqc_region* regions;  /* merged + sorted target regions (NULL for NOROI / WGS without ROI)     */
	int64_t n_regions;
	const ngsqc_region* gc_chunks;/* roi.chunk(100) lines, same order (Statistics.cpp:366) or NULL        */
	const int32_t* gc_bin;        /* per chunk: GC bin 0..100 or -1                                       */
	int64_t n_gc_chunks;
} ngsqc_mapping_params;

/* Runs the scan. counters: int64[NGSQC_NCOUNTERS]. gc_reads: double[101] (may be NULL).
 * HALF_DEPTH / BASES_COVERED_HALF are left 0 here (they need avg depth): use ngsqc_depth_stats. */
int ngsqc_scan_mapping(ngsqc_handle* h, const ngsqc_mapping_params* p, int64_t* counters, double* gc_reads);

/* ---- plain depth scan (avgCoverage / lowOrHighCoverage filters) ---- */
typedef struct ngsqc_depth_params {
	int32_t min_mapq;
	int32_t min_baseq;        /* >0: per-base quality mask of BamAlignment::qualities (BamReader.cpp:210-255) */
	int32_t skip_mismapped;   /* WorkerAverageCoverage.cpp:44                                                 */
	int32_t reserved;
	const ngsqc_region* regions; /* merged + sorted */
	int64_t n_regions;
} ngsqc_depth_params;
int ngsqc_scan_depth(ngsqc_handle* h, const ngsqc_depth_params* p);

/* ---- BedReadCount (src/BedReadCount/main.cpp:33-71): number of reads overlapping each line of a merged + sorted BED. A read counts when it is
 * mapped, not secondary / supplementary and has MAPQ >= min_mapq (duplicates count); overlap is [start, bam_endpos] against the closed line. */
int ngsqc_region_read_counts(ngsqc_handle* h, const ngsqc_region* regions, int64_t n_regions, int32_t min_mapq, int64_t* counts);

/* ---- consumers of the per-base depth left in HBM by the last ngsqc_scan_mapping / ngsqc_scan_depth ---- */
/* hist[d] for d in 0..hist_cap (depths > cap are clamped into hist[cap]); covered = #bases with depth >= half_depth */
int ngsqc_depth_stats(ngsqc_handle* h, int32_t hist_cap, int64_t half_depth, int64_t* hist, int64_t* covered);
/* per-base depth, regions concatenated in order (roi_bases int32) */
int ngsqc_depth_copy(ngsqc_handle* h, int32_t* out, int64_t cap);
/* BedCoverage: sum of depth over each (possibly overlapping, unmerged) line; every line must lie inside the scanned regions */
int ngsqc_region_sums(ngsqc_handle* h, const ngsqc_region* lines, int64_t n_lines, int64_t* sums);
/* BedLow/HighCoverage: maximal runs inside each line with depth<cutoff (is_high=0) or depth>=cutoff (is_high=1).
 * saturate254 reproduces the sweep mode's uchar array (WorkerLowOrHighCoverage.cpp:199-203). Output runs are
 * (line index, start, end) triples in line order; returns number of runs via n_runs (call with cap=0 to size). */
typedef struct ngsqc_run { int64_t line; int32_t start; int32_t end; } ngsqc_run;
int ngsqc_lowhigh_runs(ngsqc_handle* h, const ngsqc_region* lines, int64_t n_lines, int32_t cutoff, int32_t is_high,
                       int32_t saturate254, ngsqc_run* runs, int64_t cap, int64_t* n_runs);

/* ---- site pileup: BamReader::getPileup (src/cppNGS/BamReader.cpp:809-885; SNP counts, indel_window = -1,
 * count_fragments = false) for a table of known sites, e.g. the common SNPs of Statistics::contamination
 * (Statistics.cpp:2333-2386). Sites: (tid, pos) with start == end == pos (1-based), sorted by tid then pos, each tid
 * contiguous. counts[8*i + 0..5] = A, C, G, T, N, deletion at site i after the reference's filters (not secondary /
 * supplementary / duplicate / unmapped, proper pair unless include_not_properly_paired, MAPQ >= min_mapq, base quality
 * >= min_baseq; a deleted base passes with quality 255); [6] = bases Pileup::inc throws on (IUPAC codes other than
 * ACGTN), [7] = reads whose CIGAR does not reach the site (the reference throws "Could not find position"). */
int ngsqc_site_pileup(ngsqc_handle* h, const ngsqc_region* sites, int64_t n_sites, int32_t min_mapq, int32_t min_baseq,
                      int32_t include_not_properly_paired, int64_t* counts);

/* ---- raw-read QC pass: StatisticsReads::update(const BamAlignment&) (src/cppNGS/StatisticsReads.cpp:83-158), the loop
 * of `MappingQC -read_qc` (src/MappingQC/main.cpp:83-98). Secondary / supplementary records are skipped; single_end
 * counts every read as forward (otherwise read1 = forward). The reference throws on bases other than A/C/G/T/N and on
 * qualities >= 100: such records are counted in n_unknown_base / n_quality_out_of_range for the caller to throw. */
typedef struct ngsqc_read_stats {
	int64_t c_forward, c_reverse, bases_sequenced;
	int64_t bases[5];                                   /* A, C, G, T, N over all cycles (sum of the per-cycle pileups) */
	int64_t base_qualities[100];                        /* bases per quality value */
	int64_t read_qualities[100];                        /* reads per round(mean quality)  (StatisticsReads.cpp:151) */
	int64_t qscore_dist_r1[60], qscore_dist_r2[60];     /* Histogram(0,60,1) of the mean quality, forward / reverse reads (:152-153) */
	int64_t max_cycles;                                 /* longest counted read */
	int64_t n_unknown_base, n_quality_out_of_range;
} ngsqc_read_stats;
int ngsqc_scan_reads(ngsqc_handle* h, int32_t single_end, ngsqc_read_stats* st);
/* after ngsqc_scan_reads: reads per length 0..max_cycles (cap >= max_cycles + 1) */
int ngsqc_read_length_hist(ngsqc_handle* h, int64_t* out, int64_t cap);
/* after ngsqc_scan_reads: per cycle A, C, G, T, N counts and the quality sums of forward / reverse reads, out[7 * cycle + k],
 * for the first min(n_cycles, 320) cycles (per-cycle statistics only feed plots; longer reads are covered by the totals) */
int ngsqc_read_cycle_stats(ngsqc_handle* h, int64_t* out, int64_t n_cycles);

/* ---- fused job: ONE pass over the BAM for every consumer --------------------------------------------------------------------
 * The reference re-reads the BAM for every pass of a tool: MappingQC runs Statistics::mapping*, then Statistics::contamination
 * (src/MappingQC/main.cpp:100-151), optionally StatisticsReads (-read_qc, :83-98) and Statistics::somaticCustomDepth
 * (-somatic_custom_bed, :153-165). Here every BGZF member is inflated once per job and each requested consumer sees every tile
 * of the inflated stream while it is resident in HBM. NULL / 0 switches a consumer off. The single-purpose entry points above
 * (ngsqc_scan_mapping, ngsqc_scan_depth, ngsqc_site_pileup, ngsqc_scan_reads) are jobs with one consumer.
 * After the job, depth set 0 holds the mapping scan's per-base depth and depth set 1 the extra depth scan's; ngsqc_depth_select
 * picks the one that ngsqc_depth_stats / _copy / ngsqc_region_sums / ngsqc_lowhigh_runs read (a job leaves set 0 selected when it
 * had a mapping scan, otherwise set 1; ngsqc_scan_depth uses set 0). */
typedef struct ngsqc_job_desc {
	const ngsqc_mapping_params* mapping;     /* Statistics::mapping / mapping_wgs, or NULL                                      */
	const ngsqc_depth_params* depth;         /* extra depth scan on its own regions (somaticCustomDepth), or NULL               */
	const ngsqc_region* sites; int64_t n_sites;   /* site pileup (Statistics::contamination), n_sites == 0: off                 */
	int32_t site_min_mapq, site_min_baseq, site_include_npp;
	int32_t read_qc, read_qc_single_end;     /* StatisticsReads::update for every record                                         */
	int32_t reserved;
} ngsqc_job_desc;
typedef struct ngsqc_job_result {
	int64_t* counters; double* gc_reads;     /* mapping: int64[NGSQC_NCOUNTERS], double[101] (may be NULL)                       */
	int64_t* site_counts;                    /* int64[8 * n_sites]                                                               */
	ngsqc_read_stats* read_stats;            /* (+ ngsqc_read_length_hist / ngsqc_read_cycle_stats afterwards)                   */
} ngsqc_job_result;
int ngsqc_run_job(ngsqc_handle* h, const ngsqc_job_desc* job, ngsqc_job_result* result);
int ngsqc_depth_select(ngsqc_handle* h, int32_t which);

/* ---- one BAM sharded over several handles / GPUs --------------------------------------------------
 * The reference reads a BAM with one sequential reader (BamReader::getNextAlignment, src/cppNGS/BamReader.h:386-398); its
 * loop bodies (Statistics.cpp:416-574, :830-917, :1068-1183) are independent per record except for two carries: the
 * running maximum read length behind bases_trimmed (:428-429,:565-568) and "a paired read has been seen" (:879,:1115).
 * A shard handle owns the records that START inside its contiguous range of BGZF members (equal compressed bytes, cut at
 * member starts); the members behind the range are inflated only to complete its last record. Protocol per shard:
 *   ngsqc_open_shard -> ngsqc_scan_mapping_partial (local scan) -> exchange the summaries (all-gather, 48 bytes each) ->
 *   ngsqc_plan_shard_fix (pure host logic, also verifies the record chain across shards) -> ngsqc_scan_mapping_finish
 *   (local fix-up on the record prefix the carries touch; returns ADDITIVE counters) -> SUM all-reduce of the counters
 *   (MAX for max_length / paired_end / yx_valid) and of the difference array (ngsqc_depth_device, in place) ->
 *   ngsqc_depth_finalize -> ngsqc_depth_stats on the summed array.                                                   */
int ngsqc_open_shard(const char* bam_path, int device, int shard, int n_shards, ngsqc_handle** out);
int ngsqc_open_memory_shard(const void* bam_bytes, size_t n_bytes, int device, int shard, int n_shards, ngsqc_handle** out);

/* ---- index-driven partial decode: what BamReader::setRegion gives the reference (src/cppNGS/BamReader.cpp:734-768, htslib sam_index_load /
 * sam_itr_queryi): a region query reads only the BGZF blocks the BAI names. ngsqc_bai_range turns a set of regions (1-based, closed) into ONE
 * virtual-offset range [beg, end) that holds every record overlapping any of them (found = 0: none can); NGSQC_E_IO when there is no <bam>.bai:
 * "Could not load index of BAM/CRAM file ..." like the reference. ngsqc_open_range opens a handle over the records of such a range: only its BGZF
 * members (and those of the BAM header) are sent to the device and inflated. Scans that only depend on the reads overlapping the regions
 * (depth scans, site pileups, read counts) give the same result as on the whole file. */
int ngsqc_bai_range(const char* bam_path, const ngsqc_region* regions, int64_t n_regions, int32_t n_ref, uint64_t* beg_voff, uint64_t* end_voff, int32_t* found);
/* the same range for EVERY region on its own (one load of the index): beg_voff[i] / end_voff[i], end_voff[i] == 0 when no record can overlap region i. Regions that lie
 * far apart in the file are better served by one partial handle per cluster of ranges than by the one range from the first to the last (host layer: avgCoverage). */
int ngsqc_bai_ranges(const char* bam_path, const ngsqc_region* regions, int64_t n_regions, int32_t n_ref, uint64_t* beg_voff, uint64_t* end_voff);
int ngsqc_open_range(const char* bam_path, int device, uint64_t beg_voff, uint64_t end_voff, ngsqc_handle** out);
/* the same in one call for named regions (reference names as in the BAM header, with or without "chr"): header -> tids -> BAI -> range. Only the BGZF
 * members of the header and of the range are walked on the host and sent to the device. No overlapping record: a handle without records. */
typedef struct ngsqc_named_region { const char* chr; int32_t start, end; } ngsqc_named_region;
int ngsqc_open_regions(const char* bam_path, int device, const ngsqc_named_region* regions, int64_t n_regions, ngsqc_handle** out);
/* the first records of the file: the BGZF members of the BAM header and n_members behind them (what BamReader::info needs, BamReader.cpp:593-730) */
int ngsqc_open_head(const char* bam_path, int device, int64_t n_members, ngsqc_handle** out);
/* SAM header text of the BAM header (NUL-terminated copy into out[cap] when out != NULL); returns its length, -1 without a handle */
int64_t ngsqc_header_text(const ngsqc_handle* h, char* out, int64_t cap);

/* The BGZF member table of a BAM image on the host (SAM spec 4.1; what ngsqc_open builds before anything reaches the device; no device needed): members that
 * inflate to nothing (the EOF block) are left out. n_threads > 1 walks the file in pieces (accepted only when the pieces join exactly; else the sequential walk).
 * NGSQC_E_FORMAT with the message of the first broken member. */
typedef struct ngsqc_bgzf_member { uint64_t file_offset, payload_offset, inflated_offset; uint32_t payload_bytes, inflated_bytes, crc32, walked_in_pieces /* 1: the table came from the multi-thread walk */; } ngsqc_bgzf_member;
int ngsqc_bgzf_scan(const void* bam_bytes, size_t n_bytes, int32_t n_threads, ngsqc_bgzf_member* out, int64_t cap, int64_t* n_members, int64_t* inflated_bytes);

/* ---- CRAM 3.0 / 3.1 input. The reference opens CRAM through the same BamReader (htslib; BamReader.cpp:482-492) with the reference genome the caller names
 * (hts_set_fai_filename). Here every ngsqc_open* entry point takes a CRAM 3.0 or 3.1 file: its container layer (containers, slices, blocks with CRC-32, gzip and rANS
 * 4x8 blocks, the encodings, read features, mate chains, the slices' reference MD5) is decoded on the HOST into BAM records, which reach the device as a BAM image
 * whose BGZF members hold stored blocks - the device path (K1's stored-block copy, record index, walk, depth, counters) is the BAM path. The quality arrays
 * (rANS blocks) of a whole-file handle are decoded on the device into that image (cram_dev.hip; NGSQC_CRAM_DEVICE_QUALS=0: on the host). Index-driven
 * requests on a CRAM (ngsqc_open_regions) decode only the slices whose headers overlap a region - what the .crai names, read from the slice headers themselves;
 * ngsqc_open_head the first two slices; ngsqc_open_range the whole file. ngsqc_set_reference names the genome (FASTA with .fai; NULL / "": none; NGSQC_REFERENCE is the
 * fallback) for files that need one (preservation key RR); without it: NGSQC_E_IO "Error while setting reference genome ...", a genome that does not match a
 * slice's MD5: NGSQC_E_FORMAT. CRAM 3.1: rANS Nx16 blocks are decoded; the adaptive arithmetic coder, fqzcomp and the name tokeniser are NGSQC_E_UNSUPPORTED - a name-tokeniser block only
 * when names are asked for (ngsqc_set_cram_skip: the tools never do) (bzip2 / lzma blocks need libbz2 / liblzma on the machine, loaded on first use). ngsqc_cram_to_bam writes the decoded records as a BAM file
 * (host only; the checker of the decoder: tests compare it record by record with oracle/cram_decode.py). */
int ngsqc_set_reference(const char* fasta_path);
/* What later ngsqc_open* calls on a CRAM need not decode (process-wide, like ngsqc_set_reference). Replaces BamReader::skipTags() and the read-name half of
 * htslib's CRAM_OPT_REQUIRED_FIELDS, src/cppNGS/BamReader.cpp:525-572: the reference's coverage workers drop bases, tags and qualities before they iterate
 * (WorkerLowOrHighCoverage.cpp:28-31). Names: every record carries "*"; tags: no optional fields except RG. Only external blocks that no other series reads are
 * left un-inflated (a series kept in the core block is decoded and dropped). No QC result reads a name; MappingQC's target-region mode reads the DP tag. */
#define NGSQC_CRAM_SKIP_NAMES 1
#define NGSQC_CRAM_SKIP_TAGS  2
int ngsqc_set_cram_skip(int32_t flags);
/* The same choice for the files the CALLING THREAD opens from now on (flags >= 0; -1: back to the process-wide choice). Returns the thread's previous value (-1: none),
 * so that a scope can restore what it found: a function that needs tags for its own reader (Statistics::mapping reads DP) must not change what a reader opened by
 * another thread decodes. */
int32_t ngsqc_set_cram_skip_thread(int32_t flags);
int ngsqc_cram_to_bam(const char* cram_path, const char* bam_path, const ngsqc_named_region* regions, int64_t n_regions);   /* regions NULL / 0: every record */

/* ---- writing the index. The reference never builds one: every indexed path above fails with "Could not load index of BAM/CRAM file"
 * (BamReader.cpp:742-746) until `samtools index` (htslib sam_index_build: hts_idx_push / hts_idx_finish / compress_binning, hts.c) has left a
 * <bam>.bai next to the BAM. ngsqc_write_bai writes that file (bai_path NULL: <path of the handle>.bai) from a handle on the whole BAM: one pass
 * over the tiles (bin, 16 kb windows and run boundaries per record on the device), then htslib's chunk rules on the host. Same bins, chunks, linear
 * index, pseudo-bins and n_no_coor as htslib (pinned on the reference's fixture indices through oracle/bai_build.py); the order of the bins inside a
 * reference is ascending instead of htslib's hash order. NGSQC_E_FORMAT for a BAM that is not sorted by coordinate or reaches behind 2^29 (BAI limit). */
int ngsqc_write_bai(ngsqc_handle* h, const char* bai_path);
/* The CSI form of the same index (hts-specs CSIv1; `samtools index -c -m min_shift`, htslib sam_index_build3 with min_shift > 0): the binning scheme of BAI
 * with min_shift as given (<= 0: 14; 8 .. 30) and depth = the smallest n with longest reference + 256 <= 2^(min_shift + 3 n), loff per bin (the linear
 * index at the bin's first window) in place of the linear index, inside a BGZF container. csi_path NULL: <path of the handle>.csi. Every indexed entry
 * point above takes <bam>.csi before <bam>.bai, as sam_index_load (BamReader.cpp:742) does. The reference holds no .csi fixture: pinned through
 * oracle/csi_build.py, which at BAI's geometry must reproduce the htslib-written .bai fixtures' bins, chunks and (as loff) linear index. */
int ngsqc_write_csi(ngsqc_handle* h, const char* csi_path, int32_t min_shift);
/* The host half alone (several shards' runs merged by the caller; tests): runs = consecutive records of one (reference, bin) in file order, each with the
 * virtual offset of its first record (kind 0; kind 1 entries - "last record of a tile", pos clamped at 0 - only feed the sort check), lidx = first
 * virtual offset per 16 kb window (~0: none) for the windows [lidx_first[t], lidx_first[t + 1]) of reference t, counts = (mapped, unmapped) per reference
 * followed by the pair of the reads without reference. */
typedef struct ngsqc_bai_run { uint64_t voff; int32_t tid; uint32_t bin; int32_t pos; uint32_t kind; } ngsqc_bai_run;
int ngsqc_bai_assemble(const char* bai_path, int32_t n_ref, uint64_t first_record_voff, uint64_t end_voff, const ngsqc_bai_run* runs, int64_t n_runs,
                       const uint64_t* lidx, const int64_t* lidx_first, const int64_t* counts);
/* the same for a CSI index: bins of the runs and windows of lidx in the geometry (min_shift, depth) */
int ngsqc_csi_assemble(const char* csi_path, int32_t min_shift, int32_t depth, int32_t n_ref, uint64_t first_record_voff, uint64_t end_voff, const ngsqc_bai_run* runs, int64_t n_runs,
                       const uint64_t* lidx, const int64_t* lidx_first, const int64_t* counts);

typedef struct ngsqc_shard_summary {
	int64_t n_records;         /* records owned by the shard */
	int64_t first_abs;         /* inflated-stream offset (whole file) of the first record owned; -1: no record starts in the shard */
	int64_t exit_abs;          /* offset of the first record behind the shard (= the next shard's first_abs); -1 with first_abs */
	int64_t max_len;           /* longest l_seq among counted (not secondary / supplementary) records, 0 = none */
	int64_t first_max_ord;     /* shard-local ordinal of the first counted record of that length, -1 = none */
	int64_t first_paired_ord;  /* shard-local ordinal of the first record that sets paired_end, -1 = none */
} ngsqc_shard_summary;
typedef struct ngsqc_shard_fix {
	int64_t gmax;              /* max over shards of max_len */
	int64_t floor_max;         /* max of max_len over EARLIER shards (running maximum carried into this shard) */
	int64_t trim_upto;         /* local ordinals [0, trim_upto) precede the BAM's first record of length gmax */
	int64_t paired_upto;       /* local ordinals [0, paired_upto) precede the BAM's first paired record */
	int32_t paired_end;        /* some shard saw a paired read */
	int32_t reserved;
} ngsqc_shard_fix;
int ngsqc_scan_mapping_partial(ngsqc_handle* h, const ngsqc_mapping_params* p, ngsqc_shard_summary* out);
/* the fused job of a shard (what MappingQC runs on one BAM, src/MappingQC/main.cpp:100-151): the mapping scan in shard form (job->mapping is required;
 * summary in *out, counters later from ngsqc_scan_mapping_finish), the extra depth scan without its prefix sum, the site pileup (result->site_counts:
 * additive over shards). Every BGZF member of the shard is inflated once for all consumers; the shard's first records are kept in the form the
 * cross-shard fix-ups need, so ngsqc_scan_mapping_finish does not inflate anything again. read_qc is not available on shards. */
int ngsqc_run_job_partial(ngsqc_handle* h, const ngsqc_job_desc* job, ngsqc_job_result* result, ngsqc_shard_summary* out);
/* coverage tools on a shard: ngsqc_scan_depth without the final prefix sum (the depth scan has no order-dependent carries) */
int ngsqc_scan_depth_partial(ngsqc_handle* h, const ngsqc_depth_params* p);
/* returns NGSQC_E_FORMAT (message via ngsqc_last_error(NULL)) when the shards' record chains do not join */
int ngsqc_plan_shard_fix(const ngsqc_shard_summary* all, int n_shards, int shard, ngsqc_shard_fix* out);
int ngsqc_scan_mapping_finish(ngsqc_handle* h, const ngsqc_shard_fix* fix, int64_t* counters, double* gc_reads);
/* the un-prefixed difference array (int32[n_slots], device memory owned by the handle) for an in-place SUM all-reduce */
int ngsqc_depth_device(ngsqc_handle* h, void** dev_ptr, int64_t* n_slots);
int ngsqc_depth_diff_copy(ngsqc_handle* h, int32_t* out, int64_t cap);       /* host copy of the same array (CPU collectives) */
int ngsqc_depth_diff_set(ngsqc_handle* h, const int32_t* in, int64_t n);
/* dst's difference array += the arrays of the other shard handles; arrays on other GPUs are pulled with peer copies over xGMI
 * (no host staging). All handles must have scanned the same regions with ngsqc_scan_*_partial. */
int ngsqc_depth_reduce(ngsqc_handle* dst, ngsqc_handle* const* srcs, int n_srcs);
int ngsqc_depth_finalize(ngsqc_handle* h);                                    /* difference array -> per-base depth (K6 prefix sum) */

/* ---- the collective of the multi-GPU path: RCCL over xGMI, one process per GPU (round 4; BASELINE.json north_star: "only a final RCCL reduce of the
 * counter vectors"). The reference is a single process and has no counterpart (its parallelism: Statistics.cpp:2614-2638).
 * Rank 0 makes a unique id and hands it to the other ranks out of band (a file, MPI, the launcher's store); every rank then makes its communicator on its
 * own device. The calls are collective: every rank of the communicator must make them in the same order. librccl is loaded by the first of these calls.
 *   one BAM per GPU (config 4): ngsqc_run_job on every rank -> ngsqc_comm_allreduce_counters (and _i64 for site counts, _f64 for gc_reads)
 *   one BAM over the GPUs     : ngsqc_run_job_partial -> ngsqc_comm_allgather_summaries -> ngsqc_plan_shard_fix -> ngsqc_scan_mapping_finish ->
 *                               ngsqc_comm_allreduce_counters / _f64 / _i64 -> ngsqc_comm_allreduce_depth (in place, device memory) -> ngsqc_depth_finalize */
#define NGSQC_COMM_ID_BYTES 128
typedef struct ngsqc_comm ngsqc_comm;
int ngsqc_comm_unique_id(void* id /* NGSQC_COMM_ID_BYTES, out */);
int ngsqc_comm_init(int rank, int world, const void* id, int device, ngsqc_comm** out);
int ngsqc_comm_rank(const ngsqc_comm* c);
int ngsqc_comm_world(const ngsqc_comm* c);
const char* ngsqc_comm_last_error(const ngsqc_comm* c);
int ngsqc_comm_destroy(ngsqc_comm* c);
/* int64[NGSQC_NCOUNTERS] on the host, in place: SUM; MAX for max_length, paired_end, roi_bases, half_depth, yx_valid */
int ngsqc_comm_allreduce_counters(ngsqc_comm* c, int64_t* counters);
int ngsqc_comm_allreduce_i64(ngsqc_comm* c, int64_t* v, int64_t n, int take_max);   /* host vector in place: SUM (take_max = 0) or MAX */
int ngsqc_comm_allreduce_f64(ngsqc_comm* c, double* v, int64_t n);                  /* host vector in place: SUM (gc_reads) */
int ngsqc_comm_allgather_summaries(ngsqc_comm* c, const ngsqc_shard_summary* mine, ngsqc_shard_summary* all /* [world], rank order */);
int ngsqc_comm_allreduce_depth(ngsqc_comm* c, ngsqc_handle* h);                     /* the handle's int32 difference array, in place on the device */

/* ---- measurement: HIP-event timings (ms) of the stages of the last job on this handle ---- */
typedef struct ngsqc_timings {
	double h2d_ms, inflate_ms, index_ms, scan_ms, finalize_ms, total_ms;
	int64_t inflate_launches, scan_launches;
	int64_t scan_algorithmic_bytes;   /* sum over records of (4 + block_size)  — SURVEY.md §8(d) */
	int64_t compressed_bytes, inflated_bytes, n_records;
	double scan_kernel_ms;            /* K3-K5 kernels only (HIP events around the scan launches on the handle's stream) */
	double depth_kernel_ms;           /* K6 prefix sum + histogram kernels */
	double inflate_huff_ms;           /* K1 phase 1: huff_tokens_kernel, sum over its launches (0 when the group kernel ran) */
	double inflate_lz77_ms;           /* K1 phase 2: lz77 resolve kernel (sum over its launches; overlaps phase 1 of the next member chunk) */
	int64_t inflate_huff_launches;    /* K1 phase-1 launches of the last decode (member chunks of one "round" of decoder lanes) */
	/* tile stream (round 2): stage times are sums of HIP-event intervals on the handle's main stream; with more than one tile
	 * they overlap K1 of the next tile, so their sum exceeds the job's wall time */
	int64_t n_tiles;                  /* tiles of the handle's member table */
	int64_t members_inflated;         /* BGZF members that went through K1 during the last job (== members of the tiles visited) */
	double depth_scan_ms;             /* the extra depth scan of a job */
	double pileup_ms, reads_ms;       /* site pileup / raw-read QC consumers */
	double job_wall_ms;               /* host wall time of the last ngsqc_run_job (setup, all tiles, result copies) */
	int64_t members_second_chance;    /* members whose launch ran out of token pages and that were inflated again with the worst-case pool (since the handle was opened) */
	int64_t members_third_chance;     /* ... and again with the bound that holds for every valid member (members of thousands of DEFLATE blocks) */
	/* record index (round 5): how the tiles of the last decode found their record chains */
	int64_t tiles_chain_on_device;    /* tiles whose chain passed the check on the device (every walker's exit = the next walker's start): no host verification */
	int64_t tiles_scan_fused;         /* ... of which the job's mapping / depth scan rode the chain walk (one read of every record's first line) */
	int64_t walkers_per_member;       /* walkers per BGZF member of the last tile's fast path (NGSQC_WALKERS) */
	/* round 6: the state of the switches that can change a RESULT (all others only change a schedule). NGSQC_SW_* bits; the defaults are NGSQC_SW_VERIFY_CRC alone */
	int64_t switches;
} ngsqc_timings;
#define NGSQC_SW_VERIFY_CRC        1   /* every member's CRC32 is checked against its gzip trailer (NGSQC_VERIFY_CRC=0 turns it off: a measurement switch) */
#define NGSQC_SW_CRAM_IGNORE_MD5   2   /* NGSQC_CRAM_IGNORE_MD5=1: a slice's reference MD5 is not compared (htslib's ignore_md5 option) */
#define NGSQC_SW_CRAM_NO_REFERENCE 4   /* NGSQC_CRAM_NO_REFERENCE=1: bases that come from the genome are N (the reference's skipBases mode; test switch) */
/* The struct grows at its END from round to round and this call writes sizeof(ngsqc_timings) of the LIBRARY: a caller compiled against an older header must be
 * rebuilt - or call ngsqc_get_timings_sized with ITS sizeof: at most that many bytes are written (the fields it knows). ngsqc_abi_version() is the round of the
 * header the library was built from (6). */
int ngsqc_get_timings(const ngsqc_handle* h, ngsqc_timings* t);
int ngsqc_get_timings_sized(const ngsqc_handle* h, void* t, size_t struct_size);
int32_t ngsqc_abi_version(void);

/* library / device info string (static storage) */
const char* ngsqc_version(void);

/* Number of HIP devices this process sees (0 when there is none or the runtime cannot start; never an error): what a caller that opens one handle per device -
 * the reference's per-sample loop of src/MappingQC/main.cpp:60-67 spread over a node, SURVEY.md 8(e) - sizes its rank list with. */
int32_t ngsqc_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* NGSQC_H */
