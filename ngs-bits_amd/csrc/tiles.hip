// Host side of libngsqc_hip.so: the C ABI of include/ngsqc.h on top of the HIP kernels (K1 k1_kernels.h / inflate.hip + crc.hip, K2
// index.hip, K3-K5 scan.hip / reads.hip, K6 depth.hip).
//
// A BAM is processed as a STREAM OF TILES (contiguous BGZF-member ranges sized to HBM):
//   * K1 runs as one continuous stream of member chunks over the whole file on its own HIP streams (Huffman phase of chunk
//     c+1 overlaps the LZ77 phase of chunk c; the token scratch is a ring of four chunk slots), writing into one of three
//     tile buffers, queued two tiles ahead of the tile the host works on;
//   * K2 (record index) and every consumer of a tile (mapping scan, depth scan, site pileup, raw-read QC) run on the handle's
//     main stream while K1 already decodes the next tile: each member is inflated exactly once per job, and all consumers
//     of a job see the tile while it is resident (ngsqc_run_job; the single-purpose entry points are jobs with one consumer).
//   * A record that straddles two tiles is carried: its head is copied right in front of the next tile's first member
//     (a fixed prefix area in every tile buffer, so K1 of tile t+1 does not depend on K2 of tile t).
// There is no CPU fallback anywhere in this file: without a HIP device every compute entry point fails with NGSQC_E_DEVICE.
// (round 5: this file is the tile stream itself - K1 chunks of a tile, K2, the resident tile; csrc/handle.h names the other parts)
#include "handle.h"

namespace ngsqc { namespace lib {

// Enqueue K1 of tile t: its chunks continue the file-wide chunk stream (nothing here waits on the host).
void enqueue_k1_tile(ngsqc_handle* h, int t)
{
	const int64_t nb = (int64_t)h->blocks.size();
	uint8_t* out_base = h->buf[t % h->nbuf].p + h->pfx;
	// CRC of a chunk on its own stream behind the chunk's phase 2, beside phase 2 of the next chunk. With the round-3 kernels (2.4 KB LDS and 37 VGPRs per phase-2 wave) the two no longer compete for a CU's LDS: K1 of a
	// 96 M-read shard 100 -> 88 ms.
	const char* eks = getenv("NGSQC_K1_SERIAL"); const bool k1_serial = eks && atoi(eks) != 0;   // profiling: every K1 kernel in line on ONE stream (isolated per-kernel counters)   // 1: the next chunk's phase 1 starts when the whole previous launch is done
	hipStream_t crc_stream = h->s_crc;
	// (A "phased" schedule - a tile's decoder launches together, then its phase-2 launches alone - was measured in round 4: 876 against 896 Mreads/s on a 96 M-read
	// shard, profiles/r04_probe_schedule.txt; removed.)
	const int64_t cA = h->tile_first_chunk[(size_t)t], cB = h->tile_first_chunk[(size_t)t + 1];
	auto launch_p1 = [&](int64_t c) {
		const int64_t c0 = c * h->chunk, cn = std::min(h->chunk, nb - c0);
		hipEvent_t* e4 = &h->ev_chunk[(size_t)(4 * c)];
		static const int p1_streams = getenv("NGSQC_P1_STREAMS") ? atoi(getenv("NGSQC_P1_STREAMS")) : 2;   // (dev: 1 = one decoder launch at a time)
		hipStream_t s1 = k1_serial ? h->s_p2 : h->s_p1[p1_streams >= 2 ? (c & 1) : 0];
		if (c >= h->k1_slots) HIPCHK(hipStreamWaitEvent(s1, h->ev_chunk[(size_t)(4 * (c - h->k1_slots) + 3)], 0));   // the ring slot is free again
		if (h->stream_img) stream_wait_chunk(h, c, s1);
		else if (h->up) { const BlockDesc& lb = h->blocks[(size_t)(c0 + cn - 1)]; upload_wait(h, (size_t)(lb.cpos + lb.clen + 64), s1, k1_serial ? 3 : 1 + (int)(c & 1)); }
		HIPCHK(hipEventRecord(e4[0], s1));
		uint32_t* const pool = h->d_tok.p + (size_t)(c % h->k1_slots) * (size_t)h->slot_pages * K1_PAGE_WORDS;   // the chunk's slot of the token pool ring
		launch_huff_tokens(h->d_comp.p, h->d_kdesc.p + c0, cn, h->d_status.p + c0, pool, (uint32_t)h->slot_pages, h->d_pool_ctr.p + c, h->d_tok_first.p + c0, h->d_tok_cnt.p + c0, h->d_work.p + c,
		                   h->d_order.p + c0, h->p1_wgs, s1);
		HIPCHK(hipEventRecord(e4[1], s1));
	};
	auto launch_p2 = [&](int64_t c) {
		const int64_t c0 = c * h->chunk, cn = std::min(h->chunk, nb - c0);
		hipEvent_t* e4 = &h->ev_chunk[(size_t)(4 * c)];
		uint32_t* const pool = h->d_tok.p + (size_t)(c % h->k1_slots) * (size_t)h->slot_pages * K1_PAGE_WORDS;
		HIPCHK(hipStreamWaitEvent(h->s_p2, e4[1], 0));
		if (c == cA && t >= h->nbuf) HIPCHK(hipStreamWaitEvent(h->s_p2, h->ev_tile[(size_t)(2 * (t - h->nbuf) + 1)], 0));   // the buffer's previous tile is consumed
		HIPCHK(hipEventRecord(e4[2], h->s_p2));
		launch_lz77_resolve(h->d_kdesc.p + c0, cn, out_base, h->d_status.p + c0, pool, h->d_tok_first.p + c0, h->d_tok_cnt.p + c0, h->d_comp.p, h->s_p2);
		HIPCHK(hipEventRecord(e4[3], h->s_p2));
		if (h->stream_img) stream_p2_enqueued(h, c);   // (the slot of this chunk's compressed bytes may be refilled once that event has fired)
		if (h->verify_crc)   // htslib checks every member's CRC32 (bgzf.c); a mismatch fails the read
		{
			if (crc_stream != h->s_p2) HIPCHK(hipStreamWaitEvent(crc_stream, e4[3], 0));
			launch_crc32(h->d_kdesc.p + c0, cn, out_base, h->d_crc.p + c0, h->d_status.p + c0, crc_stream);
		}
	};
	for (int64_t c = cA; c < cB; ++c) { launch_p1(c); launch_p2(c); }
	const int64_t f = h->tiles[(size_t)t].first, m = h->tiles[(size_t)t].second;
	hipStream_t s_last = h->verify_crc ? crc_stream : h->s_p2;
	HIPCHK(hipMemcpyAsync(h->p_status.p + f, h->d_status.p + f, (size_t)m * sizeof(BlockStatus), hipMemcpyDeviceToHost, s_last));
	HIPCHK(hipEventRecord(h->ev_tile[(size_t)(2 * t)], s_last));
	h->tm.inflate_launches++;
	h->k1_enq = h->tile_first_chunk[(size_t)t + 1];
}

// Wait for K1 of tile t, check every member; members whose token stream overflowed the clen + 64 budget (e.g. Huffman-only
// streams of low-entropy data) get a second chance with a worst-case budget.
void finish_k1_tile(ngsqc_handle* h, int t)
{
	HIPCHK(hipEventSynchronize(h->ev_tile[(size_t)(2 * t)]));
	const int64_t f = h->tiles[(size_t)t].first, m = h->tiles[(size_t)t].second;
	std::vector<int64_t> redo;
	for (int64_t i = f; i < f + m; ++i)
	{
		const uint32_t e = h->p_status.p[i].error;
		if (e == K1_ERR_TOKEN_OVERFLOW) redo.push_back(i);
		else if (e) throw FormatError(inflate_error(h, i, e));
	}
	h->tm.members_inflated += m;
	if (redo.empty()) return;
	std::vector<BlockDesc> desc; const uint64_t u_lo = h->blocks[(size_t)f].upos;
	for (int64_t i : redo) { BlockDesc d = h->blocks[(size_t)i]; d.upos -= u_lo; desc.push_back(d); }
	inflate_sync(h, redo, desc, h->buf[t % h->nbuf].p + h->pfx);
}

// d_recoff of the tile that index_tile has just indexed (K2's write pass: the entry-relative offsets kept by the walk, expanded with coalesced stores)
const int64_t* ensure_recoff(ngsqc_handle* h)
{
	const ngsqc_handle::RecoffArgs& a = h->rw;
	if (h->recoff_tile != a.tile || a.tile < 0)
	{
		h->d_recoff.ensure_slack((size_t)std::max<int64_t>(a.n_rec, 1));
		size_t iv = h->evlog->begin(h->stream, &h->tm.index_ms);
		launch_index_write(a.base, a.total, a.desc, a.ne, a.prefix, a.ksh, a.nm, h->d_start.p, h->d_cnt.p, h->d_base.p, h->d_rel.p, h->d_recoff.p, h->stream);
		h->evlog->end(iv, h->stream);
		h->recoff_tile = a.tile;
	}
	return h->d_recoff.p;
}

// K2 for tile t (its members are in buf[t % nbuf] behind the prefix area; carry_len bytes of the previous tile's straddling
// record have been copied right in front of them). Tile-local coordinates: byte 0 = first carried byte.
void index_tile(ngsqc_handle* h, int t)
{
	const bool dbg = getenv("NGSQC_DEBUG") != nullptr;
	const int nt = (int)h->tiles.size();
	const bool last = t == nt - 1;
	const int64_t first = h->tiles[(size_t)t].first, nm = h->tiles[(size_t)t].second;
	if (t == 0) { h->carry_len = 0; h->next_ord_base = 0; h->expected_abs = h->first_rec; }
	const int64_t u_lo = (int64_t)h->blocks[(size_t)first].upos;
	const int64_t u_hi = (int64_t)h->blocks[(size_t)(first + nm - 1)].upos + h->blocks[(size_t)(first + nm - 1)].usize;
	const int64_t prefix = h->carry_len;
	const int64_t total = prefix + (u_hi - u_lo);
	const uint8_t* base = h->buf[t % h->nbuf].p + h->pfx - prefix;
	const BlockDesc* d_desc = h->d_kdesc.p + first;
	// entries: entry 0 = the carried prefix, then the members - on the fast path each cut into 2^ksh pieces with a walker of its own (common.h entry_range;
	// NGSQC_WALKERS = 1 / 2 / 4 / 8, default 1: a tile of the 30x file was 190 k members = three waves per SIMD in round 5 and is 655 k since round 6 - more waves than the chip holds buy nothing, profiles/r05_scan_probe.txt); the general path below (and a shard's first tile, whose chain is anchored by a guess) works on whole members
	const bool anchor_by_guess = t == 0 && h->first_rec < 0;
	int64_t exp0 = prefix ? 0 : (h->expected_abs - u_lo);   // local offset of the first record start of this tile
	// long reads (round 5): the file's first record says what kind of file this is - a record of more than 8 KiB means members that mostly lie inside one record.
	// Then nothing is assumed about member starts, and an entry of the fast path is a group of 16 members (common.h entry_range)
	if (t == 0)
	{
		h->long_reads = false;
		const char* elr = getenv("NGSQC_LONG_READ_MODE");   // 0 / 1: never / always (tests); unset: by the first record
		if (elr) h->long_reads = atoi(elr) != 0 && !anchor_by_guess;
		else if (!anchor_by_guess && exp0 >= 0 && exp0 + 4 <= total)
		{
			// (a sample, not one record: the mean block_size of up to eight records along the chain - an ONT file may well begin with a short read)
			int64_t off = exp0, sum = 0; int n = 0;
			for (; n < 8 && off + 4 <= total; ++n)
			{
				uint32_t bs = 0;
				HIPCHK(hipMemcpyAsync(&bs, base + off, 4, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
				if (bs < 32 || bs > (1u << 30)) break;   // (not a record: K2 will say so)
				sum += bs; off += 4 + (int64_t)bs;
			}
			h->long_reads = n > 0 && sum / n > 8192;
		}
		h->lr_failures = 0;
	}
	int ksh = 0; if (const char* e = getenv("NGSQC_WALKERS")) { const int k = atoi(e); ksh = k >= 8 ? 3 : k >= 4 ? 2 : k >= 2 ? 1 : 0; }
	if (h->long_reads) { ksh = K2_MIN_KSH; if (const char* e = getenv("NGSQC_GROUP_SHIFT")) ksh = -std::min(8, std::max(0, atoi(e))); }
	if (anchor_by_guess || (h->k2_plain && !h->long_reads)) ksh = 0;
	const int64_t ne0 = nm + 1;
	int64_t ne = ksh >= 0 ? (nm << ksh) + 1 : ((nm + (1ll << -ksh) - 1) >> -ksh) + 1;
	auto e_lo = [&](int64_t e) -> int64_t { return e == 0 ? 0 : prefix + ((int64_t)h->blocks[(size_t)(first + e - 1)].upos - u_lo); };   // (whole members: the general path)
	auto e_sz = [&](int64_t e) -> int64_t { return e == 0 ? prefix : (int64_t)h->blocks[(size_t)(first + e - 1)].usize; };
	EvLog& ev = *h->evlog; size_t iv = ev.begin(h->stream, &h->tm.index_ms);   // (the riding scan's kernel is booked as scan time: the interval is cut around it)
	// a shard behind the file header does not know where its first record starts: every member is guessed and the first
	// plausible start anchors the chain (checked against the previous shard's chain exit by ngsqc_plan_shard_fix)
	// (with slack: a later tile has one entry more - its carried prefix - and regrowing means hipFree, which waits for all queued K1 work)
	const size_t ne_max = (size_t)std::max(ne, ne0);   // (the general path below works on whole members whatever the fast path's entries were)
	h->d_start.ensure_slack(ne_max); h->d_cnt.ensure_slack(ne_max + 1); h->d_next.ensure_slack(ne_max + 1); h->d_base.ensure_slack(ne_max + 1); h->d_bad.ensure(4);   // d_bad: {corrupt records, chain violations} + the offset of a record cut by the tile end (int64, -1: none)
	h->d_scan_tmp.ensure_slack(scan_tmp_bytes((int64_t)ne_max) + 64); h->d_rel.ensure_slack((size_t)(ne0 + 1) * K2_REL_STRIDE + 64);
	int64_t from = 0; int rounds = 0; int64_t straddle = -1; bool found_start = !anchor_by_guess; int64_t chain_exit = total;
	const bool tail_may_cut_a_record = h->shard_own_members >= 0 && h->shard + 1 < h->n_shards;   // the members behind a shard end anywhere
	// ---- fast path: one round trip. Guess the first record of every entry, walk every entry's chain, check on the device that every walker's exit is the
	// next walker's start (index_chain_kernel: exact), scan the counts; the host reads back {violations, corrupt records, n_rec} only. An htslib-written
	// file passes (a record starts at every member's first byte, none straddles members or tiles) ----
	const bool assume0 = !anchor_by_guess && !h->k2_plain && !h->long_reads;   // (a file that has looked like an htslib file so far: its members start with a record)
	launch_index_init(d_desc, ne, prefix, ksh, nm, exp0, anchor_by_guess, assume0, h->d_start.p, h->stream);
	HIPCHK(hipMemsetAsync(h->d_bad.p, 0, 2 * sizeof(uint32_t), h->stream)); HIPCHK(hipMemsetAsync(h->d_bad.p + 2, 0xff, sizeof(long long), h->stream));
	h->fused_tile = -1;
	// the job's first scan consumer rides K2's walk when the file has looked like an htslib file so far (one read of every record's first line instead of two)
	const bool try_fuse = h->fuse && (h->fuse_ok || h->long_reads) && !anchor_by_guess && (prefix == 0 || h->long_reads) && !getenv("NGSQC_NO_FUSED_SCAN");   // (long reads: nearly every tile starts inside a carried record)
	const int64_t fuse_limit = h->shard_own_members >= 0 ? prefix + (h->shard_limit - u_lo) : INT64_MAX;   // a shard only scans the records that start in front of its limit
	if (try_fuse)
	{
		h->d_long.ensure_slack((size_t)std::max<int64_t>(total / 160, 1024));   // deferred (long-CIGAR) records of the tile: an estimate, checked below
		if (!assume0 || ksh > 0) launch_index_guess(base, total, d_desc, ne, prefix, ksh, nm, 0, h->d_start.p, (int32_t)h->ref_names.size(), h->stream);   // (one walker per member of an htslib-style file: nothing to guess)
		ev.end(iv, h->stream);
		h->fuse->fused_launch(h, base, total, +1, d_desc, ne, prefix, ksh, nm, fuse_limit);
		iv = ev.begin(h->stream, &h->tm.index_ms);
	}
	else launch_index_count(base, total, d_desc, ne, prefix, ksh, nm, 0, h->d_start.p, h->d_cnt.p, h->d_next.p, h->d_bad.p, (int32_t)h->ref_names.size(), h->d_rel.p, h->stream, !assume0 || ksh > 0);
	launch_index_chain(d_desc, ne, prefix, ksh, nm, exp0, total, h->d_start.p, h->d_next.p, h->d_bad.p + 1, (long long*)(h->d_bad.p + 2), h->stream);
	launch_scan_counts(h->d_cnt.p, ne, h->d_base.p, h->d_scan_tmp.p, h->stream);
	unsigned long long* sm = h->p_small.p + 32;   // [0] = {corrupt, violations} (2 x u32), [1] = record cut by the tile end, [2] = deferred records of the riding scan, [3] = n_rec
	HIPCHK(hipMemcpyAsync(sm, h->d_bad.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipMemcpyAsync(sm + 3, h->d_base.p + ne, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
	sm[2] = 0; if (try_fuse) h->fuse->fused_readback(h);   // (what the riding scan's consumers need of this tile comes with the same wait)
	HIPCHK(hipStreamSynchronize(h->stream));
	if (try_fuse) sm[2] = h->p_rb.p[A_LONG_COUNT];
	const uint32_t n_corrupt = ((const uint32_t*)sm)[0], n_viol = ((const uint32_t*)sm)[1];
	const bool aligned = !anchor_by_guess && n_viol == 0 && !getenv("NGSQC_K2_GENERAL");
	if (try_fuse)
	{
		const bool lists_fit = sm[2] <= (unsigned long long)h->d_long.n && h->p_rb.p[ngsqc_handle::RB_BQ] <= h->fuse->fused_bq_cap();
		if (aligned && n_corrupt == 0 && lists_fit) h->fused_tile = t;
		else if (!aligned || n_corrupt == 0)
		{
			// the chain did not check out (or more deferred records than the list holds): what the riding scan added is taken back, the scan runs behind K2 as usual.
			// This includes a tile whose guessed chains ran into something that looks like a corrupt record (a false start guess): the
			// general path below repairs the chain, so the walk's contributions must go whatever it met (only aligned && corrupt throws, below)
			ev.end(iv, h->stream);
			h->fuse->fused_launch(h, base, total, -1, d_desc, ne, prefix, ksh, nm, fuse_limit);
			iv = ev.begin(h->stream, &h->tm.index_ms);
			if (!aligned) h->fuse_ok = false; else h->d_long.ensure_slack((size_t)sm[2]);
		}
	}
	if (aligned)
	{
		if (n_corrupt) throw FormatError("Could not read next alignment in BAM/CRAM file " + h->path + " (corrupt record chain)");
		h->lr_failures = 0;
		straddle = (int64_t)sm[1]; h->tm.tiles_chain_on_device++; if (h->fused_tile == t) h->tm.tiles_scan_fused++;
		h->tm.walkers_per_member = ksh >= 0 ? 1ll << ksh : -(1ll << -ksh);   // (negative: members per walker)
		if (straddle >= 0 && last && !tail_may_cut_a_record) throw FormatError("Could not read next alignment in BAM/CRAM file " + h->path + " (truncated record)");
		chain_exit = std::max(total, exp0);   // (exp0 > total: the first record of the file starts in a later tile)
	}
	else
	{
	// ---- general path: records cut by tile borders, false guesses, shards that guess their first record. Whole members (ksh = 0): the host verifies that every
	// member's exit lands on the next member's start and repairs the first mismatch, round by round ----
	if (!anchor_by_guess) h->k2_plain = true;
	// (long-read mode bypasses k2_plain: a file that keeps failing the group path - short reads behind a long first read, entries with more records than a name holds - pays
	// the fused walk, its take-back and this path for every tile; after three tiles in a row the mode is dropped for the rest of the file)
	if (h->long_reads && ++h->lr_failures >= 3) h->long_reads = false;
	if (ksh != 0 || assume0)
	{
		ksh = 0; ne = ne0;
		launch_index_init(d_desc, ne, prefix, ksh, nm, exp0, anchor_by_guess, false, h->d_start.p, h->stream);
		HIPCHK(hipMemsetAsync(h->d_bad.p, 0, 2 * sizeof(uint32_t), h->stream));
		launch_index_count(base, total, d_desc, ne, prefix, ksh, nm, 0, h->d_start.p, h->d_cnt.p, h->d_next.p, h->d_bad.p, (int32_t)h->ref_names.size(), h->d_rel.p, h->stream);
	}
	h->p_start.ensure((size_t)ne + 64); h->p_next.ensure((size_t)ne + 64);
	int32_t* start = h->p_start.p; int64_t* next = h->p_next.p;
	bool first_round = true;
	while (true)
	{
		if (!first_round)
		{
			HIPCHK(hipMemsetAsync(h->d_bad.p, 0, sizeof(uint32_t), h->stream));
			launch_index_count(base, total, d_desc, ne, prefix, 0, nm, from, h->d_start.p, h->d_cnt.p, h->d_next.p, h->d_bad.p, (int32_t)h->ref_names.size(), h->d_rel.p, h->stream);
		}
		first_round = false;
		HIPCHK(hipMemcpyAsync(start + from, h->d_start.p + from, (size_t)(ne - from) * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(next + from, h->d_next.p + from, (size_t)(ne - from) * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		// exact verification of the chain: every member's exit must land on the next member's start
		int64_t expected = exp0; int64_t mismatch = -1; straddle = -1; bool anchored = !anchor_by_guess;
		for (int64_t b = 0; b < ne; ++b)
		{
			const int64_t lo = e_lo(b), hi = lo + e_sz(b);
			if (!anchored)
			{
				if (start[b] < 0) continue;          // no plausible record start inside this member
				anchored = true; expected = lo + start[b]; exp0 = expected;
			}
			const int32_t want = expected >= hi ? -1 : (int32_t)(expected - lo);
			if (start[b] != want) { mismatch = b; start[b] = want; break; }
			if (want >= 0)
			{
				const int64_t nx = next[b];
				if (nx == -2) throw FormatError("Could not read next alignment in BAM/CRAM file " + h->path + " (corrupt record chain)");
				if (nx <= -10) { straddle = -(nx + 10); expected = INT64_MAX / 2; }   // the rest of the tile belongs to this record
				else expected = nx;
			}
		}
		if (mismatch < 0)
		{
			found_start = anchored;
			if (!anchored) { expected = total; exp0 = total; }   // no record starts in this tile at all
			// without a straddling record the chain leaves the tile exactly at its end - or behind it, when the first record of the file
			// starts in a later tile (a BAM header longer than the first tile)
			if (straddle < 0 && expected < total) throw FormatError("Could not read next alignment in BAM/CRAM file " + h->path + " (record chain does not end at a member boundary)");
			if (straddle >= 0 && last && !tail_may_cut_a_record) throw FormatError("Could not read next alignment in BAM/CRAM file " + h->path + " (truncated record)");
			chain_exit = straddle < 0 ? expected : total;
			break;
		}
		if (dbg) fprintf(stderr, "[ngsqc] tile %d: chain mismatch at entry %lld (round %d)\n", t, (long long)mismatch, rounds);
		if (++rounds > 100000) throw FormatError("could not resolve the BAM record chain");
		HIPCHK(hipMemcpyAsync(h->d_start.p + mismatch, &start[mismatch], sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
		from = mismatch;
	}
	launch_scan_counts(h->d_cnt.p, ne, h->d_base.p, h->d_scan_tmp.p, h->stream);
	HIPCHK(hipMemcpyAsync(sm + 3, h->d_base.p + ne, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	}
	int64_t n_rec = (int64_t)sm[3];
	// the record offsets: expanded now, or - a job whose consumers all ride the walk - only if somebody asks (ensure_recoff)
	h->rw = ngsqc_handle::RecoffArgs{base, total, d_desc, ne, prefix, n_rec, nm, ksh, t}; h->recoff_tile = -1;
	ev.end(iv, h->stream);
	if (!h->lazy_recoff || h->fused_tile != t || h->shard_own_members >= 0) ensure_recoff(h);
	iv = ev.begin(h->stream, &h->tm.index_ms);
	if (t == 0 && h->shard_own_members >= 0) h->shard_first_abs = (found_start && (n_rec > 0 || straddle >= 0)) ? h->shard_u_base + u_lo + (exp0 - prefix) : -1;
	if (h->shard_own_members >= 0)
	{
		// records that start at or behind the shard limit belong to the next shard (recoff is ascending)
		const int64_t lim = prefix + (h->shard_limit - u_lo);
		if (lim <= total)
		{
			int64_t lo = 0, hi = n_rec;
			while (lo < hi)
			{
				const int64_t mid = (lo + hi) / 2; int64_t v = 0;
				HIPCHK(hipMemcpyAsync(&v, h->d_recoff.p + mid, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
				if (v < lim) lo = mid + 1; else hi = mid;
			}
			int64_t exit_local = -1;
			if (lo < n_rec) { HIPCHK(hipMemcpyAsync(&exit_local, h->d_recoff.p + lo, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream)); }
			else if (straddle >= 0 && straddle >= lim) exit_local = straddle;
			else if (straddle < 0 && last) exit_local = total;
			if (!found_start && last) { h->shard_exit_abs = -1; h->shard_last_tile = t; n_rec = 0; straddle = -1; }   // nothing starts here: a longer record covers the shard
			else if (exit_local >= 0 || last)
			{
				if (exit_local < 0) throw FormatError("a record at the end of shard " + std::to_string(h->shard) + " is longer than the members read behind the shard (raise NGSQC_SHARD_TAIL_MEMBERS)");
				h->shard_exit_abs = h->shard_u_base + u_lo + (exit_local - prefix); h->shard_last_tile = t;
				if (lo < n_rec || (straddle >= 0 && straddle >= lim)) straddle = -1;   // whatever straddles the end of this tile is not ours
				n_rec = lo;
			}
		}
	}
	ev.end(iv, h->stream);
	// ---- publish tile state ----
	h->cur_tile = t; h->tile_prefix = prefix; h->tile_total = total; h->tile_u_lo = u_lo; h->tile_ord_base = h->next_ord_base;
	h->n_rec = n_rec; h->tm.n_records += n_rec;
	h->carry_src = straddle; h->carry_len = straddle >= 0 ? total - straddle : 0;
	if (h->carry_len > h->pfx && !last) throw FormatError("a record that straddles two tiles is larger than the carry area (" + std::to_string(h->carry_len) + " > " + std::to_string(h->pfx) + " bytes; raise NGSQC_CARRY_MAX)");
	h->expected_abs = u_lo + (chain_exit - prefix);   // only meaningful when nothing is carried: where the next record starts (normally the next tile's first byte)
	h->next_ord_base = h->tile_ord_base + n_rec;
	h->decoded = true;
}

TileCtx resident_ctx(ngsqc_handle* h)
{
	const int nt = (int)h->tiles.size(); const int t = h->cur_tile;
	return TileCtx{h->buf[t % h->nbuf].p + h->pfx - h->tile_prefix, h->tile_total, h->recoff_tile == t ? h->d_recoff.p : nullptr /* not expanded: ensure_recoff */, h->n_rec, h->tile_ord_base, t, t == nt - 1};
}

void reset_decode_timings(ngsqc_handle* h)
{
	h->tm.inflate_ms = 0; h->tm.index_ms = 0; h->tm.inflate_launches = 0; h->tm.n_records = 0; h->tm.inflate_huff_ms = 0; h->tm.inflate_lz77_ms = 0;
	h->tm.inflate_huff_launches = 0; h->tm.members_inflated = 0; h->tm.tiles_chain_on_device = 0; h->tm.tiles_scan_fused = 0;
}

void sync_all(ngsqc_handle* h)
{
	(void)hipStreamSynchronize(h->s_p1[0]); (void)hipStreamSynchronize(h->s_p1[1]); (void)hipStreamSynchronize(h->s_p2); (void)hipStreamSynchronize(h->s_crc); (void)hipStreamSynchronize(h->stream);
}

// Visit every tile in file order with the tile resident in HBM (K1 + K2 done) while K1 of the next tile is already running.
// A single-tile file that is already decoded is visited without redoing K1 / K2 (the reference re-reads the file for every
// pass; a resident tile is kept). f returns false to stop early.
void stream_tiles(ngsqc_handle* h, const std::function<bool(const TileCtx&)>& f)
{
	plan_layout(h);
	dbg_stamp("tile stream: layout ready");
	const int nt = (int)h->tiles.size();
	if (nt == 0) { h->decoded = true; h->n_rec = 0; return; }
	if (nt == 1 && h->decoded && h->cur_tile == 0)
	{
		// (the tile may have been left by a job whose consumers never asked for the record offsets: this visitor may)
		try { if (!h->lazy_recoff) ensure_recoff(h); f(resident_ctx(h)); } catch (...) { h->evlog->discard(); throw; }
		h->evlog->resolve(); return;
	}
	reset_decode_timings(h);
	h->decoded = false; h->cur_tile = -1; h->k1_enq = 0; h->k2_plain = false;
	const bool dbg = getenv("NGSQC_DEBUG") != nullptr;
	const char* pe = getenv("NGSQC_PIPELINE"); const bool pipelined = !pe || atoi(pe) != 0;   // 0: K1 of a tile starts only when the previous tile is consumed (stage attribution)
	HIPCHK(hipMemsetAsync(h->d_work.p, 0, (size_t)h->nch * sizeof(unsigned long long), h->stream));
	HIPCHK(hipMemsetAsync(h->d_pool_ctr.p, 0, (size_t)h->nch * sizeof(uint32_t), h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	if (h->stream_img)
	{
		// (the layout thread of ngsqc_open starts the first pass as soon as the ring exists: its first slots fill while the caller still sets up its job)
		if (!(h->up->pass_running && h->up->pass_fresh)) stream_pass_begin(h);
		h->up->pass_fresh = false;
	}
	try
	{
		// K1 is queued nbuf - 1 tiles ahead of the tile the host works on (tile t + nbuf - 1 reuses the buffer of tile t - 1, whose consumers were
		// queued - and their event recorded - in the previous iteration)
		const int ahead = pipelined ? h->nbuf - 1 : 0;
		for (int u = 0; u < std::min(nt, std::max(1, ahead)); ++u) enqueue_k1_tile(h, u);
		for (int t = 0; t < nt; ++t)
		{
			const double d0 = wall_ms();
			if (pipelined && t + ahead < nt) enqueue_k1_tile(h, t + ahead);
			const double d1 = wall_ms();
			finish_k1_tile(h, t);
			const double d2 = wall_ms();
			index_tile(h, t);
			const double d3 = wall_ms();
			const bool go_on = f(resident_ctx(h));
			if (dbg) fprintf(stderr, "[ngsqc] tile %d/%d: enqueue next K1 %.2f ms, wait K1 %.2f ms, K2 %.2f ms, consumers %.2f ms (%lld records)\n", t, nt, d1 - d0, d2 - d1, d3 - d2, wall_ms() - d3, (long long)h->n_rec);
			const bool stop = !go_on || t == h->shard_last_tile;   // (a shard stops at the tile that holds the first record of the next shard)
			if (!stop && t + 1 < nt && h->carry_len > 0)
				HIPCHK(hipMemcpyAsync(h->buf[(t + 1) % h->nbuf].p + h->pfx - h->carry_len, h->buf[t % h->nbuf].p + h->pfx - h->tile_prefix + h->carry_src, (size_t)h->carry_len, hipMemcpyDeviceToDevice, h->stream));
			HIPCHK(hipEventRecord(h->ev_tile[(size_t)(2 * t + 1)], h->stream));
			if (stop) { if (t + 1 < nt) { sync_all(h); h->decoded = nt == 1; } break; }
			if (!pipelined && t + 1 < nt) { HIPCHK(hipStreamSynchronize(h->stream)); enqueue_k1_tile(h, t + 1); }
		}
	}
	catch (...) { if (h->stream_img) stream_pass_end(h); sync_all(h); h->evlog->discard(); h->decoded = false; h->cur_tile = -1; throw; }
	h->evlog->resolve();   // (index / scan / pileup stage times: HIP-event intervals that nobody waited for inside the loop)
	if (h->stream_img)
	{
		// a tile stream that stopped early (a shard's last tile, a consumer that had enough) leaves copies nobody waits for: the pass ends here
		if (h->k1_enq < h->nch) stream_pass_end(h);
		else { upload_finish(h); stream_pass_end(h); }
	}
	// K1 timings: wall time from the first phase-1 start to the last phase-2 end, and the per-kernel sums
	if (h->k1_enq > 0)
	{
		const int64_t c_end = h->k1_enq;
		HIPCHK(hipEventSynchronize(h->ev_chunk[(size_t)(4 * (c_end - 1) + 3)]));
		float ms = 0;
		HIPCHK(hipEventElapsedTime(&ms, h->ev_chunk[0], h->ev_chunk[(size_t)(4 * (c_end - 1) + 3)])); h->tm.inflate_ms = ms;
		for (int64_t c = 0; c < c_end; ++c)
		{
			hipEvent_t* e4 = &h->ev_chunk[(size_t)(4 * c)];
			HIPCHK(hipEventElapsedTime(&ms, e4[0], e4[1])); h->tm.inflate_huff_ms += ms;
			HIPCHK(hipEventElapsedTime(&ms, e4[2], e4[3])); h->tm.inflate_lz77_ms += ms;
		}
		h->tm.inflate_huff_launches = c_end;
	}
	if (nt > 1) { h->decoded = false; }   // (only a single-tile file stays resident)
}

void for_each_tile(ngsqc_handle* h, const std::function<bool(int)>& f) { stream_tiles(h, [&](const TileCtx&) { return f(h->cur_tile); }); }
}} // namespace ngsqc::lib
