// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY (see bamio.hpp header).
// Streaming form of the MappingQC -wgs loop used as bench.py's cpu_baseline ("port"): one thread pulls records
// sequentially out of the BGZF stream through a reused 64 KiB buffer, exactly the execution shape of
// BamReader::getNextAlignment (src/cppNGS/BamReader.h:386-398; the reference adds one htslib inflate helper thread,
// BamReader.cpp:472). The per-record body is the same restatement as stats.hpp (Statistics.cpp:1068-1182), with the
// indexed ROI pass and the chrX/chrY queries evaluated on the fly for every record — i.e. the CPU does LESS work than
// the reference (no second, index-driven re-read), which only makes the reported baseline faster. With a StreamSites the loop also
// does what MappingQC's third pass does (Statistics::contamination, Statistics.cpp:2333-2386: BamReader::getPileup at every known SNV),
// again on the fly: every record is looked up in the sorted sites of its reference and counted with site_pileup()'s rules - the work of
// the GPU's fused job, so that the baseline is timed on the same job.
// tests/test_oracle_stream.py checks that its counters equal mapping_wgs() of stats.hpp.
// ============================================================================
#pragma once
#include "stats.hpp"
#include <chrono>
#include <thread>

namespace orc {

struct StreamStats { int64_t n_records = 0, inflated = 0, compressed = 0; double seconds = 0; };
// the known sites of the contamination pileup: per reference the sorted 1-based positions and the row of each in counts (6 per site: A, C, G, T, N, deletion)
struct StreamSites
{
	std::vector<std::vector<int>> pos; std::vector<std::vector<int64_t>> row; int min_mapq = 1, min_baseq = 13; bool include_not_properly_paired = false; int64_t* counts = nullptr;
	void add(int tid, int p, int64_t r) { if (tid < 0) return; if ((size_t)tid >= pos.size()) { pos.resize((size_t)tid + 1); row.resize((size_t)tid + 1); } pos[(size_t)tid].push_back(p); row[(size_t)tid].push_back(r); }
	void finish()   // sort every reference's sites by position (rows follow)
	{
		for (size_t t = 0; t < pos.size(); ++t)
		{
			std::vector<size_t> o(pos[t].size()); for (size_t i = 0; i < o.size(); ++i) o[i] = i;
			std::stable_sort(o.begin(), o.end(), [&](size_t a, size_t b) { return pos[t][a] < pos[t][b]; });
			std::vector<int> p2(o.size()); std::vector<int64_t> r2(o.size());
			for (size_t i = 0; i < o.size(); ++i) { p2[i] = pos[t][o[i]]; r2[i] = row[t][o[i]]; }
			pos[t].swap(p2); row[t].swap(r2);
		}
	}
};

// begin_off / end_off (multi-threaded baseline only): process the members in [begin_off, end_off) after reading the header
// from the start of the file; begin_off must be the offset of a member at which a record starts (true for every member of an
// htslib-style "aligned" BAM such as bench.py's synthetic input). shared_depth: depth array shared by the threads (atomic adds).
inline MappingResult mapping_wgs_stream(const uint8_t* file, size_t n, const BedFile* roi_in, int min_mapq, int64_t max_records, StreamStats& st,
                                        size_t begin_off = 0, size_t end_off = (size_t)-1, int32_t* shared_depth = nullptr, const StreamSites* sites = nullptr)
{
	MappingResult r;
	std::vector<uint8_t> fv; // bgzf_scan works on a vector; avoid the copy by scanning headers inline
	// --- sequential BGZF member walk with a carry buffer for records that straddle members ---
	std::vector<uint8_t> buf; buf.reserve(1 << 20);
	size_t off = 0; bool header_done = false; std::vector<std::string> names; std::vector<int64_t> lens;
	std::vector<int> num; int tid_x = -1, tid_y = -1;
	BedFile roi; bool roi_available = roi_in != nullptr; if (roi_in) roi = *roi_in;
	if (roi_available && !roi.isMergedAndSorted()) { roi.sort(); roi.merge(); }
	std::vector<size_t> doff(roi.count() + 1, 0);
	for (size_t i = 0; i < roi.count(); ++i) doff[i + 1] = doff[i] + roi.lines[i].length();
	if (!shared_depth) r.depth.assign(doff.back(), 0);
	r.roi_bases = (int64_t)doff.back();
	std::unique_ptr<ChrIndex> roi_index; if (roi_available) roi_index.reset(new ChrIndex(roi));
	uint8_t out[65536 + 8]; size_t consumed = 0; // bytes of buf already parsed
	auto process = [&](const Rec& al)
	{
		const int nm = (al.tid >= 0 && (size_t)al.tid < num.size()) ? num[al.tid] : 0;
		// yxRatio: index query chr:[1,len]
		if (!al.isSecondary() && !al.isSupplementary())
		{
			if (al.tid == tid_x && tid_x >= 0 && tid_y >= 0 && al.pos < lens[tid_x] && al.end() > 0) ++r.reads_x;
			if (al.tid == tid_y && tid_x >= 0 && tid_y >= 0 && al.pos < lens[tid_y] && al.end() > 0) ++r.reads_y;
		}
		// indexed ROI pass (Statistics.cpp:1154-1182)
		if (roi_available && !al.isSecondary() && !al.isSupplementary() && !al.isUnmapped() && nm > 0)
		{
			roi_index->forMatches(nm, al.start(), al.end(), [&](int i){
				if (!al.isDuplicate() && al.mapq >= min_mapq)
				{
					r.bases_usable_roi += al.length();
					const BedLine& reg = roi.lines[i];
					int a = std::max(al.start(), reg.start), b = std::min(al.end(), reg.end);
					if (shared_depth) { int32_t* d = shared_depth + doff[i] - reg.start; for (int p = a; p <= b; ++p) __atomic_fetch_add(d + p, 1, __ATOMIC_RELAXED); }
					else { int* d = r.depth.data() + doff[i] - reg.start; for (int p = a; p <= b; ++p) d[p] += 1; }
				}
			});
		}
		// contamination pileup (site_pileup() of stats.hpp, BamReader.cpp:830-866), for the sites inside the record's span
		if (sites && al.tid >= 0 && (size_t)al.tid < sites->pos.size() && !sites->pos[(size_t)al.tid].empty() && !al.isSecondary() && !al.isSupplementary() && !al.isDuplicate() && !al.isUnmapped()
		    && (al.isProperPair() || sites->include_not_properly_paired) && (int)al.mapq >= sites->min_mapq)
		{
			const std::vector<int>& P = sites->pos[(size_t)al.tid]; const int a0 = al.start(), a1 = al.end();
			for (size_t k = (size_t)(std::lower_bound(P.begin(), P.end(), a0) - P.begin()); k < P.size() && P[k] <= a1; ++k)
			{
				const auto base = extract_base_by_cigar(al, P[k]);
				if (base.second < sites->min_baseq) continue;
				int64_t* c = sites->counts + 6 * sites->row[(size_t)al.tid][k];
				switch (base.first)
				{
					case 'A': ++c[0]; break; case 'C': ++c[1]; break; case 'G': ++c[2]; break; case 'T': ++c[3]; break; case 'N': ++c[4]; break; case '-': ++c[5]; break; case '~': break;
					default: throw Error(std::string("Unknown base '") + base.first + "' in pileup!");
				}
			}
		}
		if (al.isSecondary() || al.isSupplementary()) return;
		++r.al_total;
		if (al.isPaired()) r.paired_end = 1;
		const int length = al.length();
		r.max_length = std::max(r.max_length, length);
		bool spliced = false;
		if (!al.isUnmapped())
		{
			++r.al_mapped; r.bases_mapped += length;
			for (uint32_t i = 0; i < al.n_cigar; ++i) { uint32_t op = al.cigarOp(i); if (op == 4 || op == 5) r.bases_clipped += al.cigarLen(i); else if (op == 3) spliced = true; }
			if (chr_non_special(nm))
			{
				++r.al_ontarget;
				if (!al.isDuplicate() && al.mapq >= min_mapq) { r.bases_usable += length; if (r.paired_end) r.bases_usable_no_overlap += length; }
			}
		}
		if (al.isPaired() && al.isProperPair())
		{
			++r.al_proper_paired;
			if (!spliced)
			{
				const int insert_size = std::abs(al.isize);
				if (insert_size < 1000)
				{
					++r.insert_size_read_count; r.insert_size_sum += insert_size; r.insert_hist[insert_size]++;
					if (al.isRead1() && !al.isDuplicate() && al.mapq >= min_mapq && 2 * length > insert_size) r.bases_usable_no_overlap -= (2 * length) - insert_size;
				}
			}
		}
		if (length < r.max_length && length != -1) r.bases_trimmed += (r.max_length - length);
		if (al.isDuplicate()) ++r.al_dup;
	};
	bool stop = false;
	while (off < n && off < end_off && !stop)
	{
		if (off + 18 > n) throw Error("Truncated BGZF header");
		const uint8_t* p = file + off;
		if (p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) throw Error("Not a BGZF block");
		uint16_t xlen = rd16(p + 10); uint32_t bsize = 0; size_t x = 12, xend = 12 + xlen;
		while (x + 4 <= xend) { uint16_t slen = rd16(p + x + 2); if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2) bsize = rd16(p + x + 4) + 1u; x += 4 + slen; }
		if (!bsize || off + bsize > n) throw Error("Invalid BGZF block");
		uint32_t isize = rd32(p + bsize - 4);
		if (isize)
		{
			z_stream zs; memset(&zs, 0, sizeof(zs));
			if (inflateInit2(&zs, -15) != Z_OK) throw Error("inflateInit2 failed");
			zs.next_in = const_cast<uint8_t*>(p + xend); zs.avail_in = (uInt)(bsize - xend - 8); zs.next_out = out; zs.avail_out = isize;
			int rc = inflate(&zs, Z_FINISH); inflateEnd(&zs);
			if (rc != Z_STREAM_END || zs.total_out != isize) throw Error("BGZF inflate failed");
			if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), out, isize) != rd32(p + bsize - 8)) throw Error("BGZF CRC mismatch");
			// compact the carry buffer, append
			if (consumed) { buf.erase(buf.begin(), buf.begin() + (long)consumed); consumed = 0; }
			buf.insert(buf.end(), out, out + isize);
			st.inflated += isize;
			if (!header_done)
			{
				do
				{
					if (buf.size() < 12) break;
					if (memcmp(buf.data(), "BAM\1", 4) != 0) throw Error("Could not read header from BAM file");
					size_t o = 4; uint32_t l_text = rd32(&buf[o]); o += 4 + (size_t)l_text;
					if (o + 4 > buf.size()) break;
					uint32_t n_ref = rd32(&buf[o]); o += 4; bool ok = true; names.clear(); lens.clear();
					for (uint32_t i = 0; i < n_ref; ++i)
					{
						if (o + 4 > buf.size()) { ok = false; break; }
						uint32_t l_name = rd32(&buf[o]); o += 4;
						if (o + l_name + 4 > buf.size()) { ok = false; break; }
						names.emplace_back((const char*)&buf[o], l_name ? l_name - 1 : 0); o += l_name; lens.push_back(rd32(&buf[o])); o += 4;
					}
					if (!ok) break;
					header_done = true; consumed = o;
					for (auto& nme : names) num.push_back(chr_num(nme));
					for (size_t i = 0; i < num.size(); ++i) { if (num[i] == 1001 && tid_x < 0) tid_x = (int)i; if (num[i] == 1002 && tid_y < 0) tid_y = (int)i; }
				} while (false);
				if (header_done && begin_off > 0) { buf.clear(); consumed = 0; off = begin_off; st.inflated = 0; continue; }   // jump to this thread's member range
			}
			if (header_done)
			{
				while (consumed + 4 <= buf.size())
				{
					uint32_t bs = rd32(&buf[consumed]);
					if (consumed + 4 + bs > buf.size()) break;
					process(parse_rec(&buf[consumed]));
					consumed += 4 + (size_t)bs;
					if (++st.n_records == max_records) { stop = true; break; }
				}
			}
		}
		off += bsize;
	}
	st.compressed = (int64_t)(off - begin_off);
	r.bases_usable -= r.bases_clipped;
	r.yx_valid = (tid_x >= 0 && tid_y >= 0 && r.reads_x != 0) ? 1 : 0;
	if (r.roi_bases > 0 && !shared_depth)
	{
		double avg_depth = (double)r.bases_usable_roi / (double)r.roi_bases;
		int half = (int)std::round(0.5 * avg_depth); r.half_depth = half;
		for (int32_t d : r.depth) if (d >= half) ++r.bases_covered_half;
	}
	return r;
}

// All-cores form of the baseline: T threads, each over a contiguous range of BGZF members (equal compressed bytes), one
// shared depth array. Throughput only: the two order-dependent counters (bases_trimmed, bases_usable_no_overlap) are summed
// without the cross-range carries, so the counters of this form are NOT used for parity.
// per_thread (optional): the threads' results, for the ADDITIVE counters (everything except bases_trimmed, bases_usable_no_overlap,
// max_length, paired_end and the half-depth pair); depth_out (optional): the shared per-base depth array.
inline double mapping_wgs_stream_mt(const uint8_t* file, size_t n, const BedFile* roi_in, int min_mapq, int threads, StreamStats& total,
                                    std::vector<MappingResult>* per_thread = nullptr, std::vector<int32_t>* depth_out = nullptr)
{
	std::vector<size_t> member_off;
	for (size_t off = 0; off + 18 <= n;)
	{
		const uint8_t* p = file + off; uint16_t xlen = rd16(p + 10); uint32_t bsize = 0; size_t x = 12, xend = 12 + xlen;
		while (x + 4 <= xend) { uint16_t slen = rd16(p + x + 2); if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2) bsize = rd16(p + x + 4) + 1u; x += 4 + slen; }
		if (!bsize) throw Error("Invalid BGZF block");
		member_off.push_back(off); off += bsize;
	}
	threads = std::max(1, std::min<int>(threads, (int)member_off.size()));
	BedFile roi; if (roi_in) { roi = *roi_in; if (!roi.isMergedAndSorted()) { roi.sort(); roi.merge(); } }
	size_t slots = 0; for (size_t i = 0; i < roi.count(); ++i) slots += roi.lines[i].length();
	std::vector<int32_t> depth(slots, 0);
	std::vector<StreamStats> st((size_t)threads); std::vector<std::string> errs((size_t)threads);
	if (per_thread) per_thread->assign((size_t)threads, MappingResult());
	std::vector<std::thread> th;
	auto t0 = std::chrono::steady_clock::now();
	for (int t = 0; t < threads; ++t)
		th.emplace_back([&, t] {
			try
			{
				const size_t a = member_off[member_off.size() * (size_t)t / (size_t)threads];
				const size_t b = t + 1 == threads ? n : member_off[member_off.size() * (size_t)(t + 1) / (size_t)threads];
				MappingResult r = mapping_wgs_stream(file, n, roi_in ? &roi : nullptr, min_mapq, -1, st[(size_t)t], a, b, roi_in ? depth.data() : nullptr);
				if (per_thread) (*per_thread)[(size_t)t] = std::move(r);
			}
			catch (std::exception& e) { errs[(size_t)t] = e.what(); }
		});
	for (auto& x : th) x.join();
	const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	for (int t = 0; t < threads; ++t) { if (!errs[(size_t)t].empty()) throw Error(errs[(size_t)t]); total.n_records += st[(size_t)t].n_records; total.inflated += st[(size_t)t].inflated; total.compressed += st[(size_t)t].compressed; }
	total.seconds = secs;
	if (depth_out) depth_out->swap(depth);
	return secs;
}

} // namespace orc
