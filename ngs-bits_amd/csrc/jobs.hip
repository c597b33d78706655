// The consumers of a resident tile (mapping scan, depth scan, site pileup, raw-read QC, index writer) and the jobs made of them: Statistics::mapping / mapping_wgs /
// avgCoverage / lowOrHighCoverage / contamination / StatisticsReads::update behind the C ABI (file:line of each in include/ngsqc.h and INTEGRATION.md section 3).
#include "handle.h"

namespace ngsqc { namespace lib {

// regions -> device tables. Regions must be sorted by start within a tid, non-overlapping, and each tid contiguous.
void setup_regions(ngsqc_handle* h, DepthSet& D, const ngsqc_region* regions, int64_t n, bool with_depth = true)
{
	const int n_ref = (int)h->ref_names.size();
	D.regions.assign(regions, regions + (n > 0 ? n : 0));
	D.doff.assign((size_t)n + 1, 0); D.rlen.assign((size_t)n, 0);
	std::vector<int32_t> rs((size_t)n), re((size_t)n), tf((size_t)std::max(n_ref, 1), 0), tl((size_t)std::max(n_ref, 1), 0);
	std::vector<uint8_t> seen((size_t)std::max(n_ref, 1), 0);
	int64_t slots = 0, bases = 0;
	for (int64_t i = 0; i < n; ++i)
	{
		const ngsqc_region& r = regions[i];
		if (r.tid < 0 || r.tid >= n_ref) throw ArgError("region with invalid reference id");
		if (r.start < 1 || r.end < r.start) throw ArgError("invalid region range");
		if (i > 0 && regions[i - 1].tid == r.tid) { if (regions[i - 1].end >= r.start) throw ArgError("Merged and sorted BED file required for coverage details statistics!"); }
		else { if (seen[r.tid]) throw ArgError("Merged and sorted BED file required for coverage details statistics!"); seen[r.tid] = 1; tf[r.tid] = (int32_t)i; }
		tl[r.tid] = (int32_t)i + 1;
		rs[i] = r.start; re[i] = r.end; D.rlen[i] = r.end - r.start + 1; D.doff[i] = slots;
		slots += (int64_t)D.rlen[i] + 1; bases += D.rlen[i];
	}
	D.doff[n] = slots; D.n_slots = slots; D.roi_bases = bases;
	D.d_reg_start.upload(rs, h->stream); D.d_reg_end.upload(re, h->stream); D.d_reg_len.upload(D.rlen, h->stream);
	D.d_tid_first.upload(tf, h->stream); D.d_tid_last.upload(tl, h->stream);
	std::vector<int64_t> doff(D.doff.begin(), D.doff.begin() + n);
	D.d_doff.upload(doff, h->stream);
	if (with_depth)   // (a read-count scan needs the region tables only)
	{
		D.d_depth.ensure((size_t)slots + 1);
		D.d_tmp.ensure(scan_tmp_bytes(slots) + 64);
		HIPCHK(hipMemsetAsync(D.d_depth.p, 0, ((size_t)slots + 1) * sizeof(int32_t), h->stream));
	}
	HIPCHK(hipStreamSynchronize(h->stream));   // the staging vectors go out of scope
	D.depth_ready = false;
}

void finalize_depth(ngsqc_handle* h, DepthSet& D)
{
	if (D.n_slots > 0)
	{
		launch_depth_prefix(D.d_depth.p, D.n_slots, D.d_tmp.p, h->stream);
		launch_depth_mark_spare(D.d_depth.p, D.d_doff.p, D.d_reg_len.p, (int64_t)D.regions.size(), h->stream);
		HIPCHK(hipStreamSynchronize(h->stream));
	}
	D.depth_ready = true;
}

struct GcTables { DevBuf<int32_t> start, end, bin, tf, tl; };

// ---- consumers of a tile -----------------------------------------------------------------------------------------------------

// K3-K5 on every tile with the order-dependent carries of the reference loop resolved while the tile is resident:
//   bases_trimmed = sum over counted records of (running maximum read length - length)   (Statistics.cpp:428-429,565-568)
//   bases_usable_no_overlap (ROI-less modes) only counts once a paired read has been seen (:879,:1115)
// The scan reduces (longest read, first ordinal reaching it) and (first paired ordinal) per tile; a tile whose longest read does
// not exceed the running maximum carried in contributes n_counted x maximum, otherwise the running maximum is walked over the
// tile's records in front of that read (prefix_fix_kernel) - normally a handful of records of the first tile.
struct ScanState : ngsqc_handle::FusedScan
{
	ScanParams sp{}; DevBuf<unsigned long long> d_counters; DevBuf<uint32_t> d_fix;   // d_fix: scratch of the parallel order-dependent fix-up
	DevBuf<int64_t> d_bq, d_bq_sorted; DevBuf<uint8_t> d_bq_tmp; DevBuf<unsigned long long> d_bq_count; bool bq_ride = false; size_t bq_min = 0;   // MODE_DEPTH with min_baseq riding the walk: records that overlap a region, masked by baseq_tile_kernel behind the walk
	std::vector<unsigned long long> dev;   // device accumulators after the last tile
	bool in_pass_fix = true;               // false: shard protocol (ngsqc_scan_mapping_partial / _finish)
	// running state of the in-pass fix
	long long run_max = 0; bool paired_seen = false; long long sum_runmax = 0, fix_len = 0; unsigned long long prev_total = 0, prev_usable = 0;
	// summary for the shard protocol
	unsigned long long best_key = 0, first_paired = ~0ull;
	double kernel_ms = 0, stage_ms = 0; int64_t launches = 0;

	void begin(ngsqc_handle* h)
	{
		d_counters.ensure(A_DEV_TOTAL);
		std::vector<unsigned long long> init(A_DEV_TOTAL, 0ull); init[A_FIRST_PAIRED] = ~0ull;
		HIPCHK(hipMemcpyAsync(d_counters.p, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		sp.counters = d_counters.p; sp.n_ref = (int32_t)h->ref_names.size();
		// (opt-in, NGSQC_BASEQ_RIDE=1: on the bench's data - four quality levels, half of all bases below 20 - the mask is 75 atomic pairs per record, and the compacted
		// list concentrates them on neighbouring addresses: 10.1 ms of kernels per 48 M reads against 6.2 for K2 + the thread-per-record scan, profiles/r05_scan_probe.txt;
		// with instrument qualities - a few per cent below 20 - the walk's index time, 0.3 against 2.6 ms, is what is left)
		{ const char* e = getenv("NGSQC_BASEQ_RIDE"); bq_ride = sp.mode == MODE_DEPTH && sp.min_baseq > 0 && !(e && atoi(e) == 0); } if (bq_ride) d_bq_count.ensure(1);   // (round 6: on by default - the list's decrements are aggregated in LDS tiles, baseq_tile_kernel; NGSQC_BASEQ_RIDE=0: K2 + the thread-per-record scan)
		bq_min = 0;
		sp.bq_list = nullptr; sp.bq_count = nullptr; sp.bq_cap = 0;
		run_max = 0; paired_seen = false; sum_runmax = 0; fix_len = 0; prev_total = 0; prev_usable = 0; best_key = 0; first_paired = ~0ull;
		kernel_ms = 0; stage_ms = 0; launches = 0;
	}
	// the scan of a tile inside K2's chain walk (index_tile); sgn = -1 takes the tile's contributions back
	void fused_launch(ngsqc_handle* h, const uint8_t* infl, int64_t total, int sgn, const BlockDesc* d_desc, int64_t ne, int64_t prefix, int ksh, int64_t nm, int64_t scan_limit) override
	{
		sp.scan_limit = scan_limit; sp.infl = infl; sp.total = total; sp.recoff = nullptr; sp.n_rec = 0; sp.ord_base = 0;
		sp.long_list = h->d_long.p; sp.long_cap = (int64_t)h->d_long.n; sp.entry_base = nullptr; sp.sgn = sgn; sp.tile_slots = 1;
		if (bq_ride)
		{
			if (sgn > 0) { d_bq.ensure_slack(std::max((size_t)std::max<int64_t>(total / 2048, 1 << 16), bq_min)); HIPCHK(hipMemsetAsync(d_bq_count.p, 0, sizeof(unsigned long long), h->stream)); }   // (one record in fifty overlaps an exome: 340 bytes x 50 = a list entry per 17 KB; sized for one per 2 KB, checked by index_tile)
			sp.bq_list = d_bq.p; sp.bq_count = d_bq_count.p; sp.bq_cap = (int64_t)d_bq.n;
			if (const char* e = getenv("NGSQC_BQ_LIST_CAP")) sp.bq_cap = std::min<int64_t>(sp.bq_cap, std::max<int64_t>(1, atoll(e)));   // (tests: a list that overflows)
		}
		if (sgn > 0)
		{
			unsigned long long* s = h->p_small.p + 40; s[0] = 0; s[1] = ~0ull;
			if (sp.pile.list) HIPCHK(hipMemsetAsync(sp.pile.count, 0, sizeof(unsigned long long), h->stream));   // (the site pileup's candidate list of this tile)
			HIPCHK(hipMemsetAsync(d_counters.p + A_LONG_COUNT, 0, sizeof(unsigned long long), h->stream));
			HIPCHK(hipMemcpyAsync(d_counters.p + A_TILE_KEY, s, 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));   // A_TILE_KEY, A_TILE_PAIRED
		}
		const size_t iv = h->evlog->begin(h->stream, &kernel_ms, &stage_ms);   // (the walk + scan kernel: booked as scan time, not under K2)
		launch_walk_scan(sp, d_desc, ne, prefix, ksh, nm, h->d_start.p, h->d_cnt.p, h->d_next.p, h->d_bad.p, h->d_rel.p, h->stream);
		h->evlog->end(iv, h->stream); launches++;
		sp.sgn = 1; sp.scan_limit = INT64_MAX;
	}
	// round 5: everything the host needs of a tile scanned by the walk arrives with K2's own wait (index_tile) - one copy of the accumulators' head
	unsigned long long fused_bq_cap() override { return bq_ride ? (unsigned long long)sp.bq_cap : ~0ull; }
	void fused_readback(ngsqc_handle* h) override
	{
		HIPCHK(hipMemcpyAsync(h->p_rb.p, d_counters.p, (size_t)A_HIST0 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
		h->p_rb.p[ngsqc_handle::RB_CAND] = 0; h->p_rb.p[ngsqc_handle::RB_BQ] = 0;
		if (bq_ride) HIPCHK(hipMemcpyAsync(h->p_rb.p + ngsqc_handle::RB_BQ, d_bq_count.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
		if (sp.pile.list) HIPCHK(hipMemcpyAsync(h->p_rb.p + ngsqc_handle::RB_CAND, sp.pile.count, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
	}

	void tile(ngsqc_handle* h, const TileCtx& c)
	{
		const bool fused = h->fuse == this && h->fused_tile == c.tile;   // K2's chain walk has scanned the tile's records already
		if (!fused) h->d_long.ensure_slack((size_t)std::max<int64_t>(c.n_rec, 1));   // (fused: the list holds the walk's deferred records - growing it would drop them; index_tile checked that they fit)
		sp.infl = c.infl; sp.total = c.total; sp.recoff = c.recoff /* null: not expanded yet (ensure_recoff) */; sp.n_rec = c.n_rec; sp.ord_base = c.ord_base;
		sp.long_list = h->d_long.p; sp.long_cap = fused ? (int64_t)h->d_long.n : c.n_rec; sp.sgn = 1;
		sp.entry_base = fused ? h->d_base.p : nullptr; sp.tile_slots = fused ? 1 : 0;
		if (!fused && !sp.recoff) sp.recoff = ensure_recoff(h);   // (a tile no walk of THIS job has passed - a single-tile file left resident by an earlier job: the thread-per-record scan reads the offsets)
		if (!fused) { sp.bq_list = nullptr; sp.bq_count = nullptr; sp.bq_cap = 0; }   // (the scan kernel masks low-quality bases record by record)
		if (!fused && bq_ride && h->fuse == this && h->p_rb.p[ngsqc_handle::RB_BQ] > (unsigned long long)d_bq.n) bq_min = (size_t)(h->p_rb.p[ngsqc_handle::RB_BQ] + h->p_rb.p[ngsqc_handle::RB_BQ] / 4);   // (the list was too short for this tile: longer for the next)
		EvLog& ev = *h->evlog;
		const size_t ivs = ev.begin(h->stream, &stage_ms);
		unsigned long long s[16] = {0};
		auto readback = [&]() {
			unsigned long long* q = h->p_small.p;
			HIPCHK(hipMemcpyAsync(q + 0, d_counters.p + A_LONG_COUNT, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
			HIPCHK(hipMemcpyAsync(q + 1, d_counters.p + (fused ? A_TILE_KEY : A_FIRST_MAX_KEY), sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
			HIPCHK(hipMemcpyAsync(q + 2, d_counters.p + (fused ? A_TILE_PAIRED : A_FIRST_PAIRED), sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
			HIPCHK(hipMemcpyAsync(q + 3, d_counters.p + A_TOTAL, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
			HIPCHK(hipMemcpyAsync(q + 4, d_counters.p + A_USABLE, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
			HIPCHK(hipStreamSynchronize(h->stream));
			for (int i = 0; i < 5; ++i) s[i] = q[i];
		};
		if (!fused)
		{
			// per-tile slots: long-record count, (longest read, first ordinal) key
			HIPCHK(hipMemsetAsync(d_counters.p + A_LONG_COUNT, 0, sizeof(unsigned long long), h->stream));
			HIPCHK(hipMemsetAsync(d_counters.p + A_FIRST_MAX_KEY, 0, sizeof(unsigned long long), h->stream));
			const size_t ivk = ev.begin(h->stream, &kernel_ms);
			launch_scan(sp, h->stream);
			ev.end(ivk, h->stream); launches++;
			readback();
		}
		else
		{
			// (index_tile's wait brought the accumulators as the walk left them)
			const unsigned long long* rb = h->p_rb.p;
			s[0] = rb[A_LONG_COUNT]; s[1] = rb[A_TILE_KEY]; s[2] = rb[A_TILE_PAIRED]; s[3] = rb[A_TOTAL]; s[4] = rb[A_USABLE];
		}
		if (s[0])
		{
			sp.recoff = ensure_recoff(h);   // (deferred records are found through the record offsets)
			const size_t ivk = ev.begin(h->stream, &kernel_ms);
			launch_scan_long(sp, (int64_t)s[0], h->stream);
			ev.end(ivk, h->stream); launches++;
			readback();
		}
		if (fused && bq_ride && h->p_rb.p[ngsqc_handle::RB_BQ])
		{
			const size_t ivk = ev.begin(h->stream, &kernel_ms);
			const int64_t nb = (int64_t)h->p_rb.p[ngsqc_handle::RB_BQ];
			d_bq_sorted.ensure_slack((size_t)nb); const size_t tb = baseq_sort_bytes(nb); d_bq_tmp.ensure_slack(tb + 256);
			launch_baseq_list(sp, nb, h->stream, d_bq_sorted.p, d_bq_tmp.p, tb);
			ev.end(ivk, h->stream); launches++;
		}
		if (fused)
		{
			// (entry, k) names -> ordinals in the file: index in the tile = first record of the entry (the scanned counts) + k. Only a tile that holds a longer read
			// than every tile before it, or the file's first paired read, asks (the first tile of a file)
			auto ordinal = [&](unsigned long long name) -> unsigned long long {
				int64_t b0 = 0;
				HIPCHK(hipMemcpyAsync(&b0, h->d_base.p + (name >> NAME_SHIFT), sizeof(int64_t), hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
				return (unsigned long long)(c.ord_base + b0 + (int64_t)(name & ((1ull << NAME_SHIFT) - 1)));
			};
			// a key orders by (length, earlier record): a tile whose longest read is not longer than an earlier tile's never wins - its record is not asked for
			if (s[1]) s[1] = (s[1] >> 40) > (best_key >> 40) ? (s[1] & ~0xFFFFFFFFFFull) | (0xFFFFFFFFFFull - ordinal(0xFFFFFFFFFFull - (s[1] & 0xFFFFFFFFFFull))) : (s[1] & ~0xFFFFFFFFFFull);
			if (s[2] != ~0ull) s[2] = first_paired == ~0ull ? ordinal(s[2]) : first_paired;   // (a later tile's first paired read lies behind the file's first)
		}
		const unsigned long long key = s[1], fp = s[2], total = s[3], usable = s[4];
		if (key > best_key) best_key = key;   // keys order by (length, earlier ordinal): the maximum over tiles is the BAM's first longest read
		if (fp < first_paired) first_paired = fp;
		if (in_pass_fix && sp.mode != MODE_DEPTH && sp.mode != MODE_COUNT)
		{
			const long long tile_max = (long long)(key >> 40);
			const long long f_local = key ? (long long)(0xFFFFFFFFFFull - (key & 0xFFFFFFFFFFull)) - c.ord_base : 0;
			const long long n_counted = (long long)(total - prev_total);
			const bool need_trim = tile_max > run_max;
			const bool need_paired = sp.mode != NGSQC_MODE_ROI && !paired_seen && fp != ~0ull;
			const long long lf = need_trim ? f_local : 0, lp = need_paired ? (long long)fp - c.ord_base : 0;
			unsigned long long fix[3] = {0, 0, 0};
			if (lf > 0 || lp > 0)
			{
				sp.recoff = ensure_recoff(h);
				unsigned long long* q = h->p_small.p;
				q[12] = 0; q[13] = 0; q[14] = (unsigned long long)run_max; q[15] = 0;   // A_FIX_TRIM, A_FIX_LEN, A_FIX_CARRY, A_FIX_CNT
				HIPCHK(hipMemcpyAsync(d_counters.p + A_FIX_TRIM, q + 12, 4 * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
				d_fix.ensure_slack(prefix_fix_scratch_words(std::max(lf, lp)) + 1);
				launch_prefix_fix(sp, lf, lp, nullptr, h->stream, d_fix.p);
				HIPCHK(hipMemcpyAsync(q + 8, d_counters.p + A_FIX_TRIM, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
				HIPCHK(hipMemcpyAsync(q + 9, d_counters.p + A_FIX_LEN, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
				HIPCHK(hipMemcpyAsync(q + 10, d_counters.p + A_FIX_CNT, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
				HIPCHK(hipStreamSynchronize(h->stream));
				fix[0] = q[8]; fix[1] = q[9]; fix[2] = q[10];
			}
			if (need_trim) { sum_runmax += (long long)fix[0] + (n_counted - (long long)fix[2]) * tile_max; run_max = tile_max; }
			else sum_runmax += n_counted * run_max;
			if (sp.mode != NGSQC_MODE_ROI && !paired_seen)
			{
				if (fp != ~0ull) { fix_len += (long long)fix[1]; paired_seen = true; }
				else fix_len += (long long)(usable - prev_usable);   // no paired read yet: every passing record of the tile precedes the first one
			}
			prev_total = total; prev_usable = usable;
		}
		ev.end(ivs, h->stream);
	}
	void end(ngsqc_handle* h)
	{
		dev.assign(A_DEV_TOTAL, 0ull);
		HIPCHK(hipMemcpyAsync(dev.data(), d_counters.p, dev.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	}
};

// the job's first scan consumer rides K2's chain walk while the tiles stream
struct FuseGuard
{
	ngsqc_handle* h;
	FuseGuard(ngsqc_handle* hh, ScanState* sc) : h(hh) { h->fuse = sc; h->fused_tile = -1; h->fuse_ok = true; }
	~FuseGuard() { h->fuse = nullptr; h->fused_tile = -1; }
};

void bind_regions(ScanParams& sp, DepthSet& D)
{
	sp.reg_start = D.d_reg_start.p; sp.reg_end = D.d_reg_end.p; sp.reg_doff = D.d_doff.p;
	sp.tid_reg_first = D.d_tid_first.p; sp.tid_reg_last = D.d_tid_last.p; sp.n_regions = (int64_t)D.regions.size();
	sp.diff = D.d_depth.p;
}

// site pileup of a table of known sites (BamReader::getPileup per site in the reference)
struct PileupState
{
	DevBuf<int32_t> d_pos, d_tf, d_tl, d_bucket; DevBuf<int64_t> d_tb0; DevBuf<uint32_t> d_cnt; DevBuf<unsigned long long> d_nlong;
	int64_t n_sites = 0; int n_ref = 0; int min_mapq = 0, min_baseq = 0, include_npp = 0; double stage_ms = 0;
	// Round 4: when the job's mapping scan rides K2's chain walk, the walk also names the records whose reference span holds a site (scan.hip pile_candidate):
	// the pileup of such a tile runs over that list - 0.15 % of the records of a 30x WGS - instead of reading every record again (24 -> 2 ms per step of the 30x file)
	DevBuf<int64_t> d_cand; DevBuf<unsigned long long> d_ncand; const ngsqc_handle::FusedScan* rider = nullptr; int64_t tiles_from_list = 0;
	static constexpr int64_t CAND_CAP = 4ll << 20;
	void attach(ScanParams& sp, const ngsqc_handle::FusedScan* scan)
	{
		if (n_sites == 0 || getenv("NGSQC_NO_FUSED_PILEUP")) return;
		d_cand.ensure((size_t)CAND_CAP); d_ncand.ensure(1);
		sp.pile.site_pos = d_pos.p; sp.pile.tid_first = d_tf.p; sp.pile.tid_last = d_tl.p; sp.pile.bucket = d_bucket.p; sp.pile.tid_bucket0 = d_tb0.p;
		sp.pile.list = d_cand.p; sp.pile.count = d_ncand.p; sp.pile.cap = CAND_CAP; sp.pile.min_mapq = min_mapq; sp.pile.include_npp = include_npp;
		rider = scan;
	}
	void begin(ngsqc_handle* h, const ngsqc_region* sites, int64_t n, int32_t mq, int32_t bq, int32_t npp)
	{
		n_sites = n; min_mapq = mq; min_baseq = bq; include_npp = npp ? 1 : 0; stage_ms = 0;
		n_ref = (int)h->ref_names.size();
		std::vector<int32_t> pos((size_t)n_sites), tf((size_t)std::max(n_ref, 1), 0), tl((size_t)std::max(n_ref, 1), 0);
		std::vector<uint8_t> seen((size_t)std::max(n_ref, 1), 0);
		for (int64_t i = 0; i < n_sites; ++i)
		{
			const ngsqc_region& r = sites[i];
			if (r.tid < 0 || r.tid >= n_ref) throw ArgError("site with invalid reference id");
			if (r.start < 1 || r.end != r.start) throw ArgError("a site is a single 1-based position (start == end)");
			if (i > 0 && sites[i - 1].tid == r.tid) { if (sites[i - 1].start > r.start) throw ArgError("sites must be sorted by position within a reference"); }
			else { if (seen[(size_t)r.tid]) throw ArgError("sites of one reference must be contiguous"); seen[(size_t)r.tid] = 1; tf[(size_t)r.tid] = (int32_t)i; }
			tl[(size_t)r.tid] = (int32_t)i + 1; pos[(size_t)i] = r.start;
		}
		// 64 kb position buckets per reference (only references that have sites get buckets)
		std::vector<int64_t> tb0((size_t)n_ref + 1, 0); std::vector<int32_t> bucket;
		for (int t = 0; t < n_ref; ++t)
		{
			tb0[(size_t)t] = (int64_t)bucket.size();
			if (tf[(size_t)t] >= tl[(size_t)t]) continue;
			const int64_t nb = (std::max<int64_t>(h->ref_lens[(size_t)t], pos[(size_t)tl[(size_t)t] - 1]) >> PILEUP_BUCKET_SHIFT) + 2;
			int32_t i = tf[(size_t)t];
			for (int64_t b = 0; b < nb; ++b) { const int64_t lo = b << PILEUP_BUCKET_SHIFT; while (i < tl[(size_t)t] && pos[(size_t)i] < lo) ++i; bucket.push_back(i); }
		}
		tb0[(size_t)n_ref] = (int64_t)bucket.size();
		if (bucket.empty()) bucket.push_back(0);
		d_pos.upload(pos, h->stream); d_tf.upload(tf, h->stream); d_tl.upload(tl, h->stream); d_bucket.upload(bucket, h->stream); d_tb0.upload(tb0, h->stream);
		d_cnt.ensure((size_t)n_sites * 8); d_nlong.ensure(1);
		HIPCHK(hipMemsetAsync(d_cnt.p, 0, (size_t)n_sites * 8 * sizeof(uint32_t), h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	}
	void tile(ngsqc_handle* h, const TileCtx& c)
	{
		if (n_sites == 0) return;
		const size_t iv = h->evlog->begin(h->stream, &stage_ms);
		// the tile's candidates when the riding scan saw this tile (and its list held them all: the count came with index_tile's wait), else every record of the tile
		const int64_t* offs = nullptr; int64_t n = c.n_rec;
		if (rider && h->fuse == rider && h->fused_tile == c.tile && (int64_t)h->p_rb.p[ngsqc_handle::RB_CAND] <= CAND_CAP) { offs = d_cand.p; n = (int64_t)h->p_rb.p[ngsqc_handle::RB_CAND]; ++tiles_from_list; }
		if (!offs) offs = ensure_recoff(h);
		h->d_long.ensure_slack((size_t)std::max<int64_t>(n, 1));
		HIPCHK(hipMemsetAsync(d_nlong.p, 0, sizeof(unsigned long long), h->stream));
		launch_pileup(c.infl, offs, n, n_ref, d_pos.p, d_tf.p, d_tl.p, d_bucket.p, d_tb0.p, min_mapq, min_baseq, include_npp, d_cnt.p, h->d_long.p, d_nlong.p, h->stream);
		// records with long CIGARs: a wave each; how many there are stays on the device (no wait between the two kernels)
		launch_pileup_long(c.infl, offs, h->d_long.p, d_nlong.p, n, d_pos.p, d_tl.p, d_bucket.p, d_tb0.p, min_baseq, d_cnt.p, h->stream);
		h->evlog->end(iv, h->stream);
	}
	void end(ngsqc_handle* h, int64_t* counts)
	{
		if (n_sites == 0) return;
		std::vector<uint32_t> out((size_t)n_sites * 8);
		HIPCHK(hipMemcpyAsync(out.data(), d_cnt.p, out.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		for (size_t i = 0; i < out.size(); ++i) counts[i] = (int64_t)out[i];
	}
};

// raw-read QC (StatisticsReads::update). The read-length histogram grows with the longest read seen so far.
struct ReadsState
{
	DevBuf<unsigned long long> d_max, d_acc, d_len, d_cyc; int64_t len_cap = -1; int single_end = 0; double stage_ms = 0;
	void begin(ngsqc_handle* h, int se)
	{
		single_end = se ? 1 : 0; len_cap = -1; stage_ms = 0;
		d_max.ensure(1); d_acc.ensure(RA_TOTAL); d_cyc.ensure((size_t)RQ_CYC * 7);
		HIPCHK(hipMemsetAsync(d_max.p, 0, sizeof(unsigned long long), h->stream));
		HIPCHK(hipMemsetAsync(d_acc.p, 0, RA_TOTAL * sizeof(unsigned long long), h->stream));
		HIPCHK(hipMemsetAsync(d_cyc.p, 0, (size_t)RQ_CYC * 7 * sizeof(unsigned long long), h->stream));
	}
	void tile(ngsqc_handle* h, const TileCtx& c)
	{
		Timer t(h->stream); t.start();
		launch_reads_max(c.infl, c.recoff, c.n_rec, d_max.p, h->stream);
		unsigned long long* s = h->p_small.p + 24;
		HIPCHK(hipMemcpyAsync(s, d_max.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		const int64_t need = (int64_t)*s;
		if (need > len_cap)
		{
			// grow the histogram (keeps the counts of the shorter reads seen so far)
			const int64_t cap2 = std::max<int64_t>(need, len_cap < 0 ? need : len_cap * 2);
			DevBuf<unsigned long long> nw; nw.alloc((size_t)cap2 + 1);
			HIPCHK(hipMemsetAsync(nw.p, 0, ((size_t)cap2 + 1) * sizeof(unsigned long long), h->stream));
			if (len_cap >= 0) HIPCHK(hipMemcpyAsync(nw.p, d_len.p, ((size_t)len_cap + 1) * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
			HIPCHK(hipStreamSynchronize(h->stream));
			std::swap(nw.p, d_len.p); std::swap(nw.n, d_len.n); len_cap = cap2;
		}
		launch_reads(c.infl, c.recoff, c.n_rec, single_end, d_acc.p, d_len.p, len_cap, d_cyc.p, h->stream);
		stage_ms += t.stop();
	}
	void end(ngsqc_handle* h, ngsqc_read_stats* st)
	{
		unsigned long long mx = 0;
		HIPCHK(hipMemcpyAsync(&mx, d_max.p, sizeof(mx), hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
		std::vector<unsigned long long> acc(RA_TOTAL), len((size_t)mx + 1, 0ull), cyc((size_t)RQ_CYC * 7);
		HIPCHK(hipMemcpyAsync(acc.data(), d_acc.p, acc.size() * 8, hipMemcpyDeviceToHost, h->stream));
		if (len_cap >= 0) HIPCHK(hipMemcpyAsync(len.data(), d_len.p, len.size() * 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(cyc.data(), d_cyc.p, cyc.size() * 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		memset(st, 0, sizeof(*st));
		st->c_forward = (int64_t)acc[RA_FWD]; st->c_reverse = (int64_t)acc[RA_REV]; st->bases_sequenced = (int64_t)acc[RA_BASES];
		for (int i = 0; i < 5; ++i) st->bases[i] = (int64_t)acc[RA_A + i];
		for (int i = 0; i < 100; ++i) { st->base_qualities[i] = (int64_t)acc[RA_BQ0 + i]; st->read_qualities[i] = (int64_t)acc[RA_RQ0 + i]; }
		for (int i = 0; i < 60; ++i) { st->qscore_dist_r1[i] = (int64_t)acc[RA_QD0 + i]; st->qscore_dist_r2[i] = (int64_t)acc[RA_QD0 + 60 + i]; }
		st->max_cycles = (int64_t)mx; st->n_unknown_base = (int64_t)acc[RA_BAD_BASE]; st->n_quality_out_of_range = (int64_t)acc[RA_BAD_QUAL];
		h->rq_len_hist.assign(len.begin(), len.end()); h->rq_cyc.assign(cyc.begin(), cyc.end());
	}
};

// ---- BAI / CSI index of the handle's BAM (bai.hip): one pass over the tiles, then the chunk rules on the host. csi: min_shift as given (<= 0: 14), depth from the
// longest reference as sam_index_build3 chooses it (sam.c sam_index: the smallest depth with longest + 256 <= 2^(min_shift + 3 depth)) ----
void write_bai(ngsqc_handle* h, const char* out_path, bool csi, int min_shift)
{
	if (h->n_shards != 1 || h->shard_own_members >= 0 || h->member_off.size() != h->blocks.size()) throw ArgError("an index is written from a handle on the whole BAM (ngsqc_open / ngsqc_open_memory)");
	if (h->from_cram) throw ArgError("the handle is on a CRAM file: its index is a .crai (samtools index), not a .bai / .csi");
	const char* ext = csi ? ".csi" : ".bai";
	const std::string path = out_path ? std::string(out_path) : h->path + ext;
	if (path == ext) throw ArgError("no path for the index");
	const int32_t n_ref = (int32_t)h->ref_names.size();
	int depth = 5;
	if (csi)
	{
		if (min_shift <= 0) min_shift = 14;
		if (min_shift < 8 || min_shift > 30) throw ArgError("min_shift of a CSI index: 8 .. 30");
		int64_t max_len = 0;
		for (int64_t l : h->ref_lens) max_len = std::max(max_len, l);
		max_len += 256;
		depth = 0;
		for (int64_t s = 1ll << min_shift; max_len > s; s <<= 3) ++depth;
	}
	else min_shift = 14;
	// windows per reference: its length in windows of 2^min_shift (BAI: 16 kb) and some room (an alignment may reach behind the end of a circular contig)
	std::vector<int64_t> first((size_t)n_ref + 1, 0);
	const int64_t wmask = (1ll << min_shift) - 1, wmax = 1ll << (3 * depth);
	for (int32_t t = 0; t < n_ref; ++t) first[(size_t)t + 1] = first[(size_t)t] + std::min<int64_t>(wmax, ((std::max<int64_t>(h->ref_lens[(size_t)t], 0) + wmask) >> min_shift) + 8);
	const int64_t n_win = first[(size_t)n_ref];
	if (n_win > (1ll << 28)) throw ArgError("too many index windows: use a larger min_shift");
	DevBuf<int64_t> d_first; DevBuf<unsigned long long> d_lidx, d_counts, d_small; DevBuf<uint64_t> d_key; DevBuf<uint64_t> d_wnd; DevBuf<BaiRun> d_runs;
	d_first.upload(first, h->stream); d_lidx.ensure((size_t)std::max<int64_t>(n_win, 1)); d_counts.ensure(((size_t)n_ref + 1) * 2); d_small.ensure(2);
	HIPCHK(hipMemsetAsync(d_lidx.p, 0xff, (size_t)std::max<int64_t>(n_win, 1) * 8, h->stream));
	HIPCHK(hipMemsetAsync(d_counts.p, 0, ((size_t)n_ref + 1) * 16, h->stream));
	HIPCHK(hipMemsetAsync(d_small.p, 0, 16, h->stream));   // [0] runs of the tile, [1] flags
	HIPCHK(hipStreamSynchronize(h->stream));
	std::vector<BaiRun> runs; std::vector<BaiRun> part;
	const bool dbg = getenv("NGSQC_DEBUG") != nullptr;
	if (dbg) fprintf(stderr, "[bai] n_ref %d, windows %lld\n", n_ref, (long long)n_win);
	stream_tiles(h, [&](const TileCtx& c) {
		if (dbg) fprintf(stderr, "[bai] tile %d: %lld records, u_base %lld\n", c.tile, (long long)c.n_rec, (long long)(h->tile_u_lo - h->tile_prefix));
		if (c.n_rec <= 0) return true;
		d_key.ensure_slack((size_t)c.n_rec); d_wnd.ensure_slack((size_t)c.n_rec); d_runs.ensure_slack((size_t)c.n_rec + 1);
		HIPCHK(hipMemsetAsync(d_small.p, 0, 8, h->stream));
		launch_bai_keys(c.infl, c.recoff, c.n_rec, n_ref, min_shift, depth, d_key.p, d_wnd.p, d_counts.p, d_small.p + 1, h->stream);
		launch_bai_runs(c.infl, c.recoff, c.n_rec, h->tile_u_lo - h->tile_prefix, d_key.p, d_wnd.p, d_first.p, d_lidx.p, d_runs.p, d_small.p, d_small.p + 1, h->stream);
		unsigned long long sm[2] = {0, 0};
		HIPCHK(hipMemcpyAsync(sm, d_small.p, 16, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
		if (sm[1] & BAI_F_BAD_TID) throw FormatError("a record names a reference that the BAM header does not have");
		if (sm[1] & BAI_F_UNSORTED) throw FormatError("unsorted positions: the BAM is not sorted by coordinate (a BAI index needs that)");
		if (sm[1] & BAI_F_TOO_FAR) throw FormatError(csi ? "an alignment ends behind position 2^" + std::to_string(min_shift + 3 * depth) + ": it cannot be stored in a CSI index with these parameters"
		                                                 : std::string("an alignment ends behind position 2^29: it cannot be stored in a BAI index"));
		if (sm[1] & BAI_F_WINDOWS) throw FormatError(csi ? "an alignment reaches more than 8 index windows behind the end of its reference" : "an alignment reaches more than 128 kb behind the end of its reference");
		if (dbg) fprintf(stderr, "[bai]   %llu runs, flags %llu\n", sm[0], sm[1]);
		part.resize((size_t)sm[0]);
		HIPCHK(hipMemcpyAsync(part.data(), d_runs.p, (size_t)sm[0] * sizeof(BaiRun), hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
		// the runs of a tile come in the order of the atomic counter: file order is the order of their offsets (a tile's last-record marker behind a run that starts there)
		std::sort(part.begin(), part.end(), [](const BaiRun& a, const BaiRun& b) { return a.u != b.u ? a.u < b.u : a.kind < b.kind; });
		runs.insert(runs.end(), part.begin(), part.end());
		return true;
	});
	if (dbg) fprintf(stderr, "[bai] tiles done: %zu runs\n", runs.size());
	std::vector<unsigned long long> lidx_u((size_t)std::max<int64_t>(n_win, 1)), cnt(((size_t)n_ref + 1) * 2);
	HIPCHK(hipMemcpyAsync(lidx_u.data(), d_lidx.p, lidx_u.size() * 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipMemcpyAsync(cnt.data(), d_counts.p, cnt.size() * 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	// inflated offset -> virtual offset as bgzf_tell reports a position between two records: a position at the end of a member is offset 0 of the member that
	// follows in the FILE (which may be an empty one: the EOF block)
	const std::vector<BlockDesc>& B = h->blocks;
	auto tell = [&](uint64_t u) -> uint64_t {
		size_t lo = 0, hi = B.size() - 1;
		while (lo < hi) { const size_t mid = (lo + hi + 1) / 2; if (B[mid].upos < u) lo = mid; else hi = mid - 1; }   // the member that holds byte u - 1
		const uint64_t rel = u - B[lo].upos;
		return rel == B[lo].usize ? (B[lo].cpos + B[lo].clen + 8) << 16 : (h->member_off[lo] << 16) | rel;
	};
	if (B.empty()) throw FormatError("empty BAM");
	std::vector<BaiRunV> rv(runs.size());
	for (size_t i = 0; i < runs.size(); ++i) rv[i] = BaiRunV{tell((uint64_t)runs[i].u), runs[i].tid, runs[i].bin, runs[i].pos, runs[i].kind};
	std::vector<uint64_t> lidx(lidx_u.size());
	for (size_t i = 0; i < lidx.size(); ++i) lidx[i] = lidx_u[i] == ~0ull ? ~0ull : tell(lidx_u[i]);
	std::vector<int64_t> counts(cnt.begin(), cnt.end());
	if (dbg) fprintf(stderr, "[bai] assemble\n");
	const std::string e = bai_assemble(path, n_ref, tell((uint64_t)h->first_rec), tell((uint64_t)h->total), rv, lidx, first, counts, csi, min_shift, depth);
	if (dbg) fprintf(stderr, "[bai] assembled: %s\n", e.c_str());
	if (!e.empty()) { if (e.compare(0, 12, "cannot write") == 0) throw IoError(e); throw FormatError(e); }
}
}} // namespace ngsqc::lib

struct ngsqc_handle::Partial
{
	int mode = 0; bool yx = false; ScanState scan; DevBuf<uint8_t> d_ns; GcTables gc; DevBuf<unsigned long long> d_gctab; DevBuf<double> d_gcover;
	// shard protocol: what the order-dependent fix-ups need of the shard's first records (l_seq, counted, passing), kept so that
	// ngsqc_scan_mapping_finish does not inflate the shard's first tile a second time
	static constexpr int64_t HEAD_MAX = 1 << 20; DevBuf<uint32_t> d_head; int64_t head_n = 0;
};

namespace ngsqc { namespace lib {
void mapping_setup(ngsqc_handle* h, const ngsqc_mapping_params* p, ngsqc_handle::Partial& st)
{
	if (!p) throw ArgError("null argument");
	if (p->mode < NGSQC_MODE_ROI || p->mode > NGSQC_MODE_WGS) throw ArgError("invalid mode");
	if (p->mode == NGSQC_MODE_ROI && (!p->regions || p->n_regions <= 0)) throw ArgError("target-region mode needs regions");
	const int n_ref = (int)h->ref_names.size();
	const bool use_regions = p->mode != NGSQC_MODE_NOROI && p->regions && p->n_regions > 0;
	DepthSet& D = h->ds[0];
	setup_regions(h, D, use_regions ? p->regions : nullptr, use_regions ? p->n_regions : 0);
	ScanParams& sp = st.scan.sp; sp = ScanParams{};
	sp.mode = p->mode; sp.min_mapq = p->min_mapq; sp.min_baseq = 0; sp.skip_mismapped = 0;
	sp.tid_x = p->tid_x; sp.tid_y = p->tid_y;
	st.mode = p->mode; const bool yx = st.yx = p->tid_x >= 0 && p->tid_x < n_ref && p->tid_y >= 0 && p->tid_y < n_ref;
	if (!yx) { sp.tid_x = -2; sp.tid_y = -2; }
	sp.len_x = yx ? h->ref_lens[p->tid_x] : 0; sp.len_y = yx ? h->ref_lens[p->tid_y] : 0;
	std::vector<uint8_t> ns((size_t)std::max(n_ref, 1), 0);
	if (p->tid_nonspecial) for (int i = 0; i < n_ref; ++i) ns[i] = p->tid_nonspecial[i];
	st.d_ns.upload(ns, h->stream); sp.tid_nonspecial = st.d_ns.p;
	bind_regions(sp, D);
	// GC chunks
	GcTables& gc = st.gc; DevBuf<unsigned long long>& d_gctab = st.d_gctab; DevBuf<double>& d_gcover = st.d_gcover;
	const bool use_gc = use_regions && p->gc_chunks && p->gc_bin && p->n_gc_chunks > 0;
	d_gctab.alloc(101 * GC_NMAX); d_gcover.alloc(101);
	HIPCHK(hipMemsetAsync(d_gctab.p, 0, 101 * GC_NMAX * sizeof(unsigned long long), h->stream));
	HIPCHK(hipMemsetAsync(d_gcover.p, 0, 101 * sizeof(double), h->stream));
	if (use_gc)
	{
		const int64_t n = p->n_gc_chunks;
		std::vector<int32_t> s((size_t)n), e((size_t)n), b((size_t)n), tf((size_t)std::max(n_ref, 1), 0), tl((size_t)std::max(n_ref, 1), 0);
		for (int64_t i = 0; i < n; ++i)
		{
			const ngsqc_region& r = p->gc_chunks[i];
			if (r.tid < 0 || r.tid >= n_ref) throw ArgError("GC chunk with invalid reference id");
			if (i == 0 || p->gc_chunks[i - 1].tid != r.tid) tf[r.tid] = (int32_t)i;
			tl[r.tid] = (int32_t)i + 1;
			s[i] = r.start; e[i] = r.end; b[i] = p->gc_bin[i] > 100 ? -1 : p->gc_bin[i];
		}
		gc.start.upload(s, h->stream); gc.end.upload(e, h->stream); gc.bin.upload(b, h->stream); gc.tf.upload(tf, h->stream); gc.tl.upload(tl, h->stream);
		HIPCHK(hipStreamSynchronize(h->stream));
		sp.gc_start = gc.start.p; sp.gc_end = gc.end.p; sp.gc_bin = gc.bin.p; sp.tid_gc_first = gc.tf.p; sp.tid_gc_last = gc.tl.p; sp.n_gc = n;
	}
	HIPCHK(hipStreamSynchronize(h->stream));
	sp.gc_tab = d_gctab.p; sp.gc_over = d_gcover.p;
}

// device accumulators -> the reference's counters. gmax / paired_end: of the whole BAM (== this handle's unless it is a shard);
// sum_runmax: sum over counted records of the running maximum read length; fix_len: passing bases in front of the first paired read
void mapping_counters(ngsqc_handle* h, ngsqc_handle::Partial& st, int gmax, bool paired_end, long long sum_runmax, long long fix_len, int64_t* counters, double* gc_reads)
{
	const std::vector<unsigned long long>& dev = st.scan.dev; const bool yx = st.yx;
	auto S = [&](int i) { return (int64_t)dev[i]; };
	for (int i = 0; i < NGSQC_NCOUNTERS; ++i) counters[i] = 0;
	counters[NGSQC_C_AL_TOTAL] = S(A_TOTAL); counters[NGSQC_C_AL_MAPPED] = S(A_MAPPED); counters[NGSQC_C_AL_ONTARGET] = S(A_ONTARGET);
	counters[NGSQC_C_AL_NEARTARGET] = S(A_NEAR); counters[NGSQC_C_AL_DUP] = S(A_DUP); counters[NGSQC_C_AL_PROPER_PAIRED] = S(A_PP);
	counters[NGSQC_C_INSERT_SIZE_READ_COUNT] = S(A_INS_CNT);
	counters[NGSQC_C_BASES_TRIMMED] = sum_runmax - S(A_SUM_LEN);
	counters[NGSQC_C_BASES_MAPPED] = S(A_BASES_MAPPED); counters[NGSQC_C_BASES_CLIPPED] = S(A_CLIPPED); counters[NGSQC_C_INSERT_SIZE_SUM] = S(A_INS_SUM);
	if (st.mode == NGSQC_MODE_ROI)
	{
		counters[NGSQC_C_BASES_USABLE] = S(A_USABLE);
		counters[NGSQC_C_BASES_USABLE_NO_OVERLAP] = S(A_NO_OVERLAP);
	}
	else
	{
		counters[NGSQC_C_BASES_USABLE] = S(A_USABLE) - S(A_CLIPPED);                        // Statistics.cpp:917 / :1183
		counters[NGSQC_C_BASES_USABLE_NO_OVERLAP] = (paired_end ? S(A_USABLE) - fix_len : 0) + S(A_NO_OVERLAP); // :879,:898-901
	}
	counters[NGSQC_C_BASES_USABLE_RAW] = S(A_USABLE_RAW); counters[NGSQC_C_BASES_USABLE_ROI] = S(A_USABLE_ROI);
	for (int i = 0; i < 5; ++i) counters[NGSQC_C_BASES_USABLE_DP0 + i] = S(A_DP0 + i);
	for (int i = 0; i < 4; ++i) counters[NGSQC_C_DP_DIST0 + i] = S(A_DD0 + i);
	counters[NGSQC_C_MAX_LENGTH] = gmax; counters[NGSQC_C_PAIRED_END] = paired_end ? 1 : 0;
	counters[NGSQC_C_ROI_BASES] = h->ds[0].roi_bases;
	counters[NGSQC_C_READS_X] = yx ? S(A_READS_X) : 0; counters[NGSQC_C_READS_Y] = yx ? S(A_READS_Y) : 0;
	counters[NGSQC_C_YX_VALID] = (yx && S(A_READS_X) != 0) ? 1 : 0;
	for (int i = 0; i < 1000; ++i) counters[NGSQC_C_INSERT_HIST0 + i] = S(A_HIST0 + i);
	if (gc_reads)
	{
		std::vector<unsigned long long> tab(101 * GC_NMAX); std::vector<double> over(101);
		HIPCHK(hipMemcpy(tab.data(), st.d_gctab.p, tab.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		HIPCHK(hipMemcpy(over.data(), st.d_gcover.p, over.size() * sizeof(double), hipMemcpyDeviceToHost));
		for (int b = 0; b <= 100; ++b)
		{
			double v = over[b];
			for (int n = 1; n < GC_NMAX; ++n) if (tab[(size_t)b * GC_NMAX + n]) v += (double)tab[(size_t)b * GC_NMAX + n] * (1.0 / (double)n);
			gc_reads[b] = v;
		}
	}
	h->tm.scan_algorithmic_bytes = (int64_t)dev[A_ALG_BYTES];
}

void depth_setup(ngsqc_handle* h, const ngsqc_depth_params* p, DepthSet& D, ScanState& sc)
{
	if (!p || !p->regions || p->n_regions <= 0) throw ArgError("depth scan needs regions");
	setup_regions(h, D, p->regions, p->n_regions);
	ScanParams& sp = sc.sp; sp = ScanParams{};
	sp.mode = MODE_DEPTH; sp.min_mapq = p->min_mapq; sp.min_baseq = p->min_baseq; sp.skip_mismapped = p->skip_mismapped;
	sp.tid_x = -2; sp.tid_y = -2;
	bind_regions(sp, D);
}

// The fused job: every requested consumer sees every tile once.
void run_job(ngsqc_handle* h, const ngsqc_job_desc* j, ngsqc_job_result* r, ngsqc_shard_summary* shard_out = nullptr)
{
	if (!j || !r) throw ArgError("null argument");
	const bool part = shard_out != nullptr;   // a shard: additive results only (mapping: summary now, counters from ngsqc_scan_mapping_finish; depth: the un-prefixed difference arrays)
	const bool do_map = j->mapping != nullptr, do_depth = j->depth != nullptr, do_sites = j->n_sites > 0, do_reads = j->read_qc != 0;
	if (do_map && !part && !r->counters) throw ArgError("mapping job without a counter buffer");
	if (part && (!do_map || do_reads)) throw ArgError("a shard job needs the mapping scan and cannot run the raw-read QC");
	if (do_sites && (!j->sites || !r->site_counts)) throw ArgError("site pileup job without sites / count buffer");
	if (do_reads && !r->read_stats) throw ArgError("raw-read QC job without a result buffer");
	if (j->n_sites < 0) throw ArgError("invalid site count");
	const double w0 = wall_ms();
	dbg_stamp("job: start");
	h->tm.scan_ms = 0; h->tm.scan_kernel_ms = 0; h->tm.scan_launches = 0; h->tm.finalize_ms = 0; h->tm.depth_scan_ms = 0; h->tm.pileup_ms = 0; h->tm.reads_ms = 0; h->tm.scan_algorithmic_bytes = 0;   // (every per-consumer field of the previous job)
	Timer total(h->stream); total.start();
	ngsqc_handle::Partial local_map; ScanState dscan; PileupState pile; ReadsState reads;
	if (part) { delete h->partial; h->partial = new ngsqc_handle::Partial(); }
	ngsqc_handle::Partial& map = part ? *h->partial : local_map;
	if (do_map) { mapping_setup(h, j->mapping, map); map.scan.in_pass_fix = !part; map.scan.begin(h); }
	if (do_depth) { depth_setup(h, j->depth, h->ds[1], dscan); dscan.in_pass_fix = false; dscan.begin(h); }
	if (do_sites) pile.begin(h, j->sites, j->n_sites, j->site_min_mapq, j->site_min_baseq, j->site_include_npp);
	if (do_sites && do_map) pile.attach(map.scan.sp, &map.scan);   // (the pileup's candidates come from the scan that rides K2's chain walk)
	if (do_reads) reads.begin(h, j->read_qc_single_end);
	const double w1 = wall_ms();
	const bool depth_rides = do_depth && (j->depth->min_baseq <= 0 || !(getenv("NGSQC_BASEQ_RIDE") && atoi(getenv("NGSQC_BASEQ_RIDE")) == 0));
	FuseGuard fg(h, do_map ? &map.scan : (depth_rides ? &dscan : nullptr));
	// the record offsets of a tile are only expanded when a consumer reads them: the mapping scan rides the chain walk (deferred long-CIGAR records and the
	// order-dependent fix-ups ask for them), the site pileup works on the walk's candidate list; the extra depth scan and the raw-read QC read every record
	struct LazyGuard { ngsqc_handle* h; ~LazyGuard() { h->lazy_recoff = false; } } lg{h};
	// (round 6: a coverage tool's job - the depth scan alone, riding the walk - does not expand them either: 0.23 ms per tile of the 30x file, 6 % of its scan stage)
	h->lazy_recoff = !part && !do_reads && ((do_map && !do_depth) || (!do_map && depth_rides && !do_sites)) && !getenv("NGSQC_EAGER_RECOFF");
	stream_tiles(h, [&](const TileCtx& c) {
		if (do_map) map.scan.tile(h, c);
		if (part && c.ord_base == 0 && c.n_rec > 0)
		{
			// the shard's first records in the form the cross-shard fix-ups need them
			map.head_n = std::min<int64_t>(c.n_rec, ngsqc_handle::Partial::HEAD_MAX);
			map.d_head.ensure((size_t)map.head_n);
			launch_prefix_capture(map.scan.sp, map.head_n, map.d_head.p, h->stream);
		}
		if (do_depth) dscan.tile(h, c);
		if (do_sites) pile.tile(h, c);
		if (do_reads) reads.tile(h, c);
		return true;
	});
	const double w2 = wall_ms();
	h->tm.scan_ms = 0; h->tm.scan_kernel_ms = 0; h->tm.scan_launches = 0; h->tm.finalize_ms = 0;
	if (do_map)
	{
		map.scan.end(h);
		if (!part)
		{
			Timer fin(h->stream); fin.start();
			finalize_depth(h, h->ds[0]);
			h->tm.finalize_ms = fin.stop();
			mapping_counters(h, map, (int)(map.scan.best_key >> 40), map.scan.first_paired != ~0ull, map.scan.sum_runmax, map.scan.fix_len, r->counters, r->gc_reads);
		}
		else
		{
			const unsigned long long key = map.scan.best_key;
			shard_out->n_records = h->tm.n_records;
			shard_out->first_abs = h->shard_own_members >= 0 ? h->shard_first_abs : (h->tm.n_records ? h->first_rec : -1);
			shard_out->exit_abs = h->shard_own_members >= 0 ? h->shard_exit_abs : (h->tm.n_records ? h->total : -1);
			shard_out->max_len = (int64_t)(key >> 40);
			shard_out->first_max_ord = key ? (int64_t)(0xFFFFFFFFFFull - (key & 0xFFFFFFFFFFull)) : -1;
			shard_out->first_paired_ord = map.scan.first_paired != ~0ull ? (int64_t)map.scan.first_paired : -1;
		}
		h->tm.scan_ms = map.scan.stage_ms; h->tm.scan_kernel_ms = map.scan.kernel_ms; h->tm.scan_launches = map.scan.launches;
	}
	if (do_depth)
	{
		dscan.end(h);
		if (!part) { Timer fin(h->stream); fin.start(); finalize_depth(h, h->ds[1]); h->tm.finalize_ms += fin.stop(); }
		h->tm.depth_scan_ms = dscan.stage_ms;
		if (!do_map) { h->tm.scan_algorithmic_bytes = (int64_t)dscan.dev[A_ALG_BYTES]; h->tm.scan_kernel_ms = dscan.kernel_ms; h->tm.scan_launches = dscan.launches; h->tm.scan_ms = dscan.stage_ms; }
	}
	if (do_sites) { pile.end(h, r->site_counts); h->tm.pileup_ms = pile.stage_ms; }
	if (do_reads) { reads.end(h, r->read_stats); h->tm.reads_ms = reads.stage_ms; }
	h->cur_ds = do_map || !do_depth ? 0 : 1;
	h->tm.total_ms = total.stop();
	h->tm.job_wall_ms = wall_ms() - w0;
	if (getenv("NGSQC_DEBUG")) fprintf(stderr, "[ngsqc] job: setup %.2f ms, tile stream %.2f ms (K1 %.2f), results %.2f ms\n", w1 - w0, w2 - w1, h->tm.inflate_ms, wall_ms() - w2);
}

DepthSet& cur_depth(ngsqc_handle* h) { return h->ds[h->cur_ds]; }

}} // namespace ngsqc::lib

extern "C" {

void ngsqc_close(ngsqc_handle* h)
{
	if (!h) return;
	struct Hold { Hold() { reaper().hold(); } ~Hold() { reaper().unhold(); } } hold_frees;   // (the large buffers go back when this handle is gone)
	if (h->plan_thread.joinable()) h->plan_thread.join();
	if (h->up) { upload_join(h); delete h->up; h->up = nullptr; }
	if (h->stream) { (void)hipSetDevice(h->device); sync_all(h); }
	for (hipStream_t s : {h->stream, h->s_p1[0], h->s_p1[1], h->s_p2, h->s_crc}) if (s) (void)hipStreamDestroy(s);
	for (hipEvent_t e : h->ev_chunk) (void)hipEventDestroy(e);
	for (hipEvent_t e : h->ev_tile) (void)hipEventDestroy(e);
	delete h->partial;
	delete h;
}

int ngsqc_run_job(ngsqc_handle* h, const ngsqc_job_desc* job, ngsqc_job_result* result) { return guarded(h, [&] { run_job(h, job, result); }); }

int ngsqc_depth_select(ngsqc_handle* h, int32_t which)
{
	return guarded(h, [&] { if (which < 0 || which >= N_DEPTH_SETS) throw ArgError("invalid depth set"); h->cur_ds = which; });
}

int ngsqc_scan_mapping(ngsqc_handle* h, const ngsqc_mapping_params* p, int64_t* counters, double* gc_reads)
{
	return guarded(h, [&] {
		if (!p || !counters) throw ArgError("null argument");
		ngsqc_job_desc j{}; j.mapping = p; ngsqc_job_result r{}; r.counters = counters; r.gc_reads = gc_reads;
		run_job(h, &j, &r);
	});
}

// ---- one BAM sharded over several handles: local scan, tiny exchange, local fix-up, additive counters ----
int ngsqc_scan_mapping_partial(ngsqc_handle* h, const ngsqc_mapping_params* p, ngsqc_shard_summary* out)
{
	return guarded(h, [&] {
		if (!p || !out) throw ArgError("null argument");
		ngsqc_job_desc j{}; j.mapping = p; ngsqc_job_result r{};
		run_job(h, &j, &r, out);
	});
}
// the fused job of a shard: the mapping scan in shard form (summary now, counters from ngsqc_scan_mapping_finish), the extra depth scan without its
// prefix sum, the site pileup (its counts are additive over shards) - every BGZF member of the shard is inflated once for all of them
int ngsqc_run_job_partial(ngsqc_handle* h, const ngsqc_job_desc* job, ngsqc_job_result* result, ngsqc_shard_summary* out)
{
	return guarded(h, [&] { if (!out) throw ArgError("null argument"); run_job(h, job, result, out); });
}

int ngsqc_scan_mapping_finish(ngsqc_handle* h, const ngsqc_shard_fix* fix, int64_t* counters, double* gc_reads)
{
	return guarded(h, [&] {
		if (!fix || !counters) throw ArgError("null argument");
		if (!h->partial) throw ArgError("ngsqc_scan_mapping_finish without ngsqc_scan_mapping_partial");
		ngsqc_handle::Partial& st = *h->partial; ScanState& sc = st.scan;
		Timer total(h->stream); total.start();
		// running maximum / "paired seen" on the record prefix the carries of the WHOLE BAM touch: [0, trim_upto) / [0, paired_upto)
		// of this shard, with the running maximum of the earlier shards carried in. Normally empty or a handful of records; the
		// tiles that hold them are visited again (a shard is rarely more than one tile).
		const int64_t f = fix->trim_upto, pidx = st.mode != NGSQC_MODE_ROI ? fix->paired_upto : 0;
		unsigned long long fx[4] = {0, 0, (unsigned long long)std::max<int64_t>(fix->floor_max, 0), 0};   // A_FIX_TRIM, A_FIX_LEN, A_FIX_CARRY, A_FIX_CNT
		if (f > 0 || pidx > 0)
		{
			HIPCHK(hipMemcpyAsync(sc.d_counters.p + A_FIX_TRIM, fx, sizeof(fx), hipMemcpyHostToDevice, h->stream));
			const int64_t upto = std::max(f, pidx);
			if (upto <= st.head_n)   // the prefix lies inside the records captured by the shard job: nothing is inflated again
				launch_prefix_fix(sc.sp, f, pidx, st.d_head.p, h->stream);
			else stream_tiles(h, [&](const TileCtx& c) {
				sc.sp.infl = c.infl; sc.sp.total = c.total; sc.sp.recoff = c.recoff; sc.sp.n_rec = c.n_rec; sc.sp.ord_base = c.ord_base;
				const int64_t lf = std::min<int64_t>(std::max<int64_t>(f - c.ord_base, 0), c.n_rec), lp = std::min<int64_t>(std::max<int64_t>(pidx - c.ord_base, 0), c.n_rec);
				launch_prefix_fix(sc.sp, lf, lp, nullptr, h->stream);
				HIPCHK(hipStreamSynchronize(h->stream));
				return c.ord_base + c.n_rec < upto;
			});
			HIPCHK(hipMemcpyAsync(fx, sc.d_counters.p + A_FIX_TRIM, sizeof(fx), hipMemcpyDeviceToHost, h->stream));
			HIPCHK(hipStreamSynchronize(h->stream));
		}
		// records behind the prefix run at the BAM's maximum
		const long long sum_runmax = (long long)fx[0] + ((long long)sc.dev[A_TOTAL] - (long long)fx[3]) * (long long)fix->gmax;
		mapping_counters(h, st, (int)fix->gmax, fix->paired_end != 0, sum_runmax, (long long)fx[1], counters, gc_reads);
		h->tm.total_ms += total.stop();
	});
}

int ngsqc_depth_device(ngsqc_handle* h, void** dev_ptr, int64_t* n_slots)
{
	return guarded(h, [&] {
		if (!dev_ptr || !n_slots) throw ArgError("null argument");
		DepthSet& D = cur_depth(h);
		if (D.depth_ready) throw ArgError("the depth array is already finalized (prefix-summed)");
		HIPCHK(hipStreamSynchronize(h->stream));
		*dev_ptr = D.d_depth.p; *n_slots = D.n_slots;
	});
}
int ngsqc_depth_diff_copy(ngsqc_handle* h, int32_t* out, int64_t cap)
{
	return guarded(h, [&] {
		DepthSet& D = cur_depth(h);
		if (D.depth_ready) throw ArgError("the depth array is already finalized (prefix-summed)");
		if (cap < D.n_slots || (!out && D.n_slots)) throw ArgError("depth buffer too small");
		if (D.n_slots) HIPCHK(hipMemcpyAsync(out, D.d_depth.p, (size_t)D.n_slots * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	});
}
int ngsqc_depth_diff_set(ngsqc_handle* h, const int32_t* in, int64_t n)
{
	return guarded(h, [&] {
		DepthSet& D = cur_depth(h);
		if (D.depth_ready) throw ArgError("the depth array is already finalized (prefix-summed)");
		if (n != D.n_slots || (!in && n)) throw ArgError("depth buffer size mismatch");
		if (n) HIPCHK(hipMemcpyAsync(D.d_depth.p, in, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	});
}
// SUM of the un-prefixed difference arrays of several shard handles into dst's array. Handles on other devices are read
// through peer copies (xGMI) into a staging buffer on dst's device; nothing passes through host memory.
int ngsqc_depth_reduce(ngsqc_handle* dst, ngsqc_handle* const* srcs, int n_srcs)
{
	return guarded(dst, [&] {
		if (n_srcs < 0 || (n_srcs && !srcs)) throw ArgError("null argument");
		DepthSet& D = cur_depth(dst);
		if (D.depth_ready) throw ArgError("the depth array is already finalized (prefix-summed)");
		DevBuf<int32_t> stage;
		for (int i = 0; i < n_srcs; ++i)
		{
			ngsqc_handle* s = srcs[i];
			if (!s || s == dst) continue;
			DepthSet& S = cur_depth(s);
			if (S.depth_ready || S.n_slots != D.n_slots) throw ArgError("shard depth arrays do not match");
			if (D.n_slots == 0) continue;
			HIPCHK(hipSetDevice(s->device)); HIPCHK(hipStreamSynchronize(s->stream)); HIPCHK(hipSetDevice(dst->device));
			const int32_t* src = S.d_depth.p;
			if (s->device != dst->device)
			{
				stage.ensure((size_t)D.n_slots);
				HIPCHK(hipMemcpyPeerAsync(stage.p, dst->device, S.d_depth.p, s->device, (size_t)D.n_slots * sizeof(int32_t), dst->stream));
				src = stage.p;
			}
			launch_depth_add(D.d_depth.p, src, D.n_slots, dst->stream);
			HIPCHK(hipStreamSynchronize(dst->stream));
		}
	});
}
int ngsqc_depth_finalize(ngsqc_handle* h)
{
	return guarded(h, [&] {
		DepthSet& D = cur_depth(h);
		if (D.depth_ready) return;
		Timer fin(h->stream); fin.start();
		finalize_depth(h, D);
		h->tm.finalize_ms = fin.stop();
	});
}

// Pure host logic (no device): what shard `shard` needs for its fix-up, from the summaries of all shards in file order.
// Also verifies the record chain across shards: a shard's guessed first record must be the previous shard's chain exit.
int ngsqc_plan_shard_fix(const ngsqc_shard_summary* all, int n_shards, int shard, ngsqc_shard_fix* out)
{
	if (!all || !out || n_shards < 1 || shard < 0 || shard >= n_shards) return NGSQC_E_ARG;
	int64_t cur = -1;
	for (int s = 0; s < n_shards; ++s)
	{
		if (all[s].first_abs < 0) continue;
		if (cur >= 0 && all[s].first_abs != cur) { g_open_error = "shard " + std::to_string(s) + " starts at inflated offset " + std::to_string(all[s].first_abs) + " but the previous shard's record chain ends at " + std::to_string(cur); return NGSQC_E_FORMAT; }
		cur = all[s].exit_abs;
	}
	int64_t gmax = 0; int s_max = -1, s_paired = -1;
	for (int s = 0; s < n_shards; ++s) if (all[s].max_len > gmax) { gmax = all[s].max_len; }
	for (int s = 0; s < n_shards; ++s) if (s_max < 0 && gmax > 0 && all[s].max_len == gmax) s_max = s;
	for (int s = 0; s < n_shards; ++s) if (s_paired < 0 && all[s].first_paired_ord >= 0) s_paired = s;
	int64_t floor_max = 0; for (int s = 0; s < shard; ++s) floor_max = std::max(floor_max, all[s].max_len);
	out->gmax = gmax; out->floor_max = floor_max; out->paired_end = s_paired >= 0 ? 1 : 0;
	out->trim_upto = s_max < 0 ? 0 : (shard < s_max ? all[shard].n_records : (shard == s_max ? all[shard].first_max_ord : 0));
	out->paired_upto = s_paired < 0 ? 0 : (shard < s_paired ? all[shard].n_records : (shard == s_paired ? all[shard].first_paired_ord : 0));
	return NGSQC_OK;
}

int ngsqc_site_pileup(ngsqc_handle* h, const ngsqc_region* sites, int64_t n_sites, int32_t min_mapq, int32_t min_baseq, int32_t include_not_properly_paired, int64_t* counts)
{
	return guarded(h, [&] {
		if (n_sites < 0 || (n_sites && (!sites || !counts))) throw ArgError("null argument");
		if (n_sites == 0) return;
		ngsqc_job_desc j{}; j.sites = sites; j.n_sites = n_sites; j.site_min_mapq = min_mapq; j.site_min_baseq = min_baseq; j.site_include_npp = include_not_properly_paired;
		ngsqc_job_result r{}; r.site_counts = counts;
		const int keep = h->cur_ds;
		run_job(h, &j, &r);
		h->cur_ds = keep;
	});
}

int ngsqc_scan_reads(ngsqc_handle* h, int32_t single_end, ngsqc_read_stats* st)
{
	return guarded(h, [&] {
		if (!st) throw ArgError("null argument");
		ngsqc_job_desc j{}; j.read_qc = 1; j.read_qc_single_end = single_end; ngsqc_job_result r{}; r.read_stats = st;
		const int keep = h->cur_ds;
		run_job(h, &j, &r);
		h->cur_ds = keep;
	});
}
int ngsqc_read_length_hist(ngsqc_handle* h, int64_t* out, int64_t cap)
{
	return guarded(h, [&] {
		if (h->rq_len_hist.empty()) throw ArgError("no read statistics: run ngsqc_scan_reads first");
		if (!out || cap < (int64_t)h->rq_len_hist.size()) throw ArgError("read-length buffer too small");
		std::copy(h->rq_len_hist.begin(), h->rq_len_hist.end(), out);
	});
}
int ngsqc_read_cycle_stats(ngsqc_handle* h, int64_t* out, int64_t n_cycles)
{
	return guarded(h, [&] {
		if (h->rq_cyc.empty()) throw ArgError("no read statistics: run ngsqc_scan_reads first");
		if (!out || n_cycles < 0) throw ArgError("invalid cycle buffer");
		const int64_t n = std::min<int64_t>(n_cycles, RQ_CYC);
		std::copy(h->rq_cyc.begin(), h->rq_cyc.begin() + 7 * n, out);
		for (int64_t i = 7 * n; i < 7 * n_cycles; ++i) out[i] = 0;
	});
}

namespace {
void depth_scan(ngsqc_handle* h, const ngsqc_depth_params* p, bool finalize)
{
	Timer total(h->stream); total.start();
	ScanState sc; sc.in_pass_fix = false;
	depth_setup(h, p, h->ds[0], sc);
	sc.begin(h);
	// (round 5: with -min_baseq the records that overlap a region leave the walk for a list and a wave-per-record kernel masks their low-quality bases; rounds 3-4
	// took the thread-per-record path - K2, then the scan kernel - because the decrements inside the walk stalled its lanes: 147 vs 224 ms per 96 M reads)
	// (round 6: a riding depth scan does not have the record offsets expanded - deferred long-CIGAR records ask for them (ensure_recoff): 0.23 ms per tile of the 30x file)
	struct LazyGuard { ngsqc_handle* h; ~LazyGuard() { h->lazy_recoff = false; } } lg{h};
	{ const char* e = getenv("NGSQC_BASEQ_RIDE"); const bool ride = p->min_baseq <= 0 || !(e && atoi(e) == 0); h->lazy_recoff = ride && finalize && !getenv("NGSQC_EAGER_RECOFF"); FuseGuard fg(h, ride ? &sc : nullptr); stream_tiles(h, [&](const TileCtx& c) { sc.tile(h, c); return true; }); }
	sc.end(h);
	h->cur_ds = 0;
	h->tm.scan_ms = sc.stage_ms; h->tm.scan_kernel_ms = sc.kernel_ms; h->tm.scan_launches = sc.launches; h->tm.scan_algorithmic_bytes = (int64_t)sc.dev[A_ALG_BYTES];
	if (finalize) { Timer fin(h->stream); fin.start(); finalize_depth(h, h->ds[0]); h->tm.finalize_ms = fin.stop(); }
	h->tm.total_ms = total.stop();
}
} // namespace

int ngsqc_scan_depth(ngsqc_handle* h, const ngsqc_depth_params* p) { return guarded(h, [&] { depth_scan(h, p, true); }); }

// BedReadCount: reads (mapped, not secondary / supplementary, MAPQ >= min_mapq) overlapping each line of a merged + sorted BED
int ngsqc_region_read_counts(ngsqc_handle* h, const ngsqc_region* regions, int64_t n_regions, int32_t min_mapq, int64_t* counts)
{
	return guarded(h, [&] {
		if (!regions || n_regions <= 0 || !counts) throw ArgError("read counting needs regions and a result buffer");
		const int keep = h->cur_ds;
		DepthSet D;   // private region tables, no depth array: the depth sets of the handle (and what an earlier job left in them) stay as they are
		try { setup_regions(h, D, regions, n_regions, false); }
		catch (ArgError& e)
		{
			if (std::string(e.what()).find("Merged and sorted") != std::string::npos) throw ArgError("Merged and sorted BED file required for coverage calculation!");   // src/BedReadCount/main.cpp:36-39
			throw;
		}
		ScanState sc; sc.in_pass_fix = false;
		ScanParams& sp = sc.sp; sp = ScanParams{};
		sp.mode = MODE_COUNT; sp.min_mapq = min_mapq; sp.tid_x = -2; sp.tid_y = -2;
		bind_regions(sp, D);
		DevBuf<unsigned long long> d_cnt; d_cnt.alloc((size_t)n_regions);
		HIPCHK(hipMemsetAsync(d_cnt.p, 0, (size_t)n_regions * sizeof(unsigned long long), h->stream));
		sp.region_reads = d_cnt.p;
		sc.begin(h);
		{ FuseGuard fg(h, &sc); stream_tiles(h, [&](const TileCtx& c) { sc.tile(h, c); return true; }); }
		HIPCHK(hipMemcpyAsync(counts, d_cnt.p, (size_t)n_regions * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		h->cur_ds = keep;
	});
}
// shard variant: leaves the un-prefixed difference array (additive over shards: ngsqc_depth_reduce / _device / _diff_copy / _diff_set, then ngsqc_depth_finalize)
int ngsqc_scan_depth_partial(ngsqc_handle* h, const ngsqc_depth_params* p) { return guarded(h, [&] { depth_scan(h, p, false); }); }

int ngsqc_depth_stats(ngsqc_handle* h, int32_t hist_cap, int64_t half_depth, int64_t* hist, int64_t* covered)
{
	return guarded(h, [&] {
		DepthSet& D = cur_depth(h);
		if (!D.depth_ready) throw ArgError("no depth array: run ngsqc_scan_mapping / ngsqc_scan_depth first");
		if (hist_cap < 0 || hist_cap > 30000 || !hist || !covered) throw ArgError("invalid histogram request");
		DevBuf<unsigned long long> d_hist; d_hist.alloc((size_t)hist_cap + 2);
		HIPCHK(hipMemsetAsync(d_hist.p, 0, ((size_t)hist_cap + 2) * sizeof(unsigned long long), h->stream));
		launch_depth_hist(D.d_depth.p, D.n_slots, hist_cap, half_depth, d_hist.p, d_hist.p + hist_cap + 1, h->stream);
		std::vector<unsigned long long> out((size_t)hist_cap + 2);
		HIPCHK(hipMemcpyAsync(out.data(), d_hist.p, out.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		for (int i = 0; i <= hist_cap; ++i) hist[i] = (int64_t)out[i];
		*covered = (int64_t)out[(size_t)hist_cap + 1];
	});
}

int ngsqc_depth_copy(ngsqc_handle* h, int32_t* out, int64_t cap)
{
	return guarded(h, [&] {
		DepthSet& D = cur_depth(h);
		if (!D.depth_ready) throw ArgError("no depth array: run ngsqc_scan_mapping / ngsqc_scan_depth first");
		if (cap < D.roi_bases) throw ArgError("depth buffer too small");
		if (D.roi_bases == 0) return;
		DevBuf<int32_t> d_out; d_out.alloc((size_t)D.roi_bases);
		launch_depth_compact(D.d_depth.p, D.d_doff.p, D.d_reg_len.p, (int64_t)D.regions.size(), d_out.p, h->stream);
		HIPCHK(hipMemcpyAsync(out, d_out.p, (size_t)D.roi_bases * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	});
}

namespace {
// locate each line inside the scanned (merged) regions: slot offset of its first base
void locate_lines(ngsqc_handle* h, DepthSet& D, const ngsqc_region* lines, int64_t n, std::vector<int64_t>& slot, std::vector<int32_t>& len, std::vector<int32_t>& lstart)
{
	slot.resize((size_t)n); len.resize((size_t)n); lstart.resize((size_t)n);
	const auto& R = D.regions;
	std::vector<std::pair<int32_t, int32_t>> group(h->ref_names.size(), {0, 0}); // per tid: [first,last) in R
	for (size_t k = 0; k < R.size();) { size_t e = k; while (e < R.size() && R[e].tid == R[k].tid) ++e; group[R[k].tid] = {(int32_t)k, (int32_t)e}; k = e; }
	for (int64_t i = 0; i < n; ++i)
	{
		const ngsqc_region& l = lines[i];
		if (l.start < 1 || l.end < l.start) throw ArgError("invalid line range");
		if (l.tid < 0 || l.tid >= (int32_t)group.size()) throw ArgError("line with invalid reference id");
		int lo = group[l.tid].first, last = group[l.tid].second, hi = last;
		while (lo < hi) { int m = (lo + hi) / 2; if (R[m].end < l.start) lo = m + 1; else hi = m; }
		if (!(lo < last && R[lo].start <= l.start && R[lo].end >= l.end)) throw ArgError("line is not covered by the scanned regions");
		slot[i] = D.doff[lo] + (l.start - R[lo].start); len[i] = l.end - l.start + 1; lstart[i] = l.start;
	}
}
}

int ngsqc_region_sums(ngsqc_handle* h, const ngsqc_region* lines, int64_t n_lines, int64_t* sums)
{
	return guarded(h, [&] {
		DepthSet& D = cur_depth(h);
		if (!D.depth_ready) throw ArgError("no depth array: run ngsqc_scan_depth first");
		if (n_lines <= 0) return;
		if (!lines || !sums) throw ArgError("null argument");
		std::vector<int64_t> slot; std::vector<int32_t> len, ls;
		locate_lines(h, D, lines, n_lines, slot, len, ls);
		DevBuf<int64_t> d_slot; d_slot.upload(slot, h->stream);
		DevBuf<int32_t> d_len; d_len.upload(len, h->stream);
		DevBuf<long long> d_sums; d_sums.alloc((size_t)n_lines);
		launch_line_sums(D.d_depth.p, d_slot.p, d_len.p, n_lines, d_sums.p, h->stream);
		HIPCHK(hipMemcpyAsync(sums, d_sums.p, (size_t)n_lines * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	});
}

int ngsqc_lowhigh_runs(ngsqc_handle* h, const ngsqc_region* lines, int64_t n_lines, int32_t cutoff, int32_t is_high, int32_t saturate254,
                       ngsqc_run* runs, int64_t cap, int64_t* n_runs)
{
	return guarded(h, [&] {
		DepthSet& D = cur_depth(h);
		if (!D.depth_ready) throw ArgError("no depth array: run ngsqc_scan_depth first");
		if (!n_runs) throw ArgError("null argument");
		*n_runs = 0;
		if (n_lines <= 0) return;
		std::vector<int64_t> slot; std::vector<int32_t> len, ls;
		locate_lines(h, D, lines, n_lines, slot, len, ls);
		DevBuf<int64_t> d_slot; d_slot.upload(slot, h->stream);
		DevBuf<int32_t> d_len; d_len.upload(len, h->stream);
		DevBuf<int32_t> d_ls; d_ls.upload(ls, h->stream);
		DevBuf<uint32_t> d_cnt; d_cnt.alloc((size_t)n_lines + 1);
		DevBuf<int64_t> d_base; d_base.alloc((size_t)n_lines + 1);
		DevBuf<uint8_t> d_tmp; d_tmp.alloc(scan_tmp_bytes(n_lines) + 64);
		launch_line_runs(false, D.d_depth.p, d_slot.p, d_len.p, d_ls.p, n_lines, cutoff, is_high, saturate254, d_cnt.p, nullptr, nullptr, h->stream);
		launch_scan_counts(d_cnt.p, n_lines, d_base.p, d_tmp.p, h->stream);
		int64_t total = 0;
		HIPCHK(hipMemcpyAsync(&total, d_base.p + n_lines, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		*n_runs = total;
		if (!runs || cap < total || total == 0) return;
		DevBuf<ngsqc_run> d_runs; d_runs.alloc((size_t)total);
		launch_line_runs(true, D.d_depth.p, d_slot.p, d_len.p, d_ls.p, n_lines, cutoff, is_high, saturate254, d_cnt.p, d_base.p, d_runs.p, h->stream);
		HIPCHK(hipMemcpyAsync(runs, d_runs.p, (size_t)total * sizeof(ngsqc_run), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	});
}


} // extern "C"
