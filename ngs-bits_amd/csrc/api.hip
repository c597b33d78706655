// The C entry points of include/ngsqc.h that open, close and describe a handle (the jobs' entry points: jobs.hip). No CPU fallback: without a HIP device every
// compute entry point fails with NGSQC_E_DEVICE.
#include "handle.h"

namespace ngsqc { namespace lib { thread_local std::string g_open_error; }}

extern "C" {


int ngsqc_open(const char* bam_path, int device, ngsqc_handle** out) { if (!bam_path) return NGSQC_E_ARG; return open_impl(out, bam_path, nullptr, 0, device); }
int ngsqc_open_memory(const void* bam_bytes, size_t n_bytes, int device, ngsqc_handle** out) { return open_impl(out, nullptr, bam_bytes, n_bytes, device); }
int ngsqc_open_shard(const char* bam_path, int device, int shard, int n_shards, ngsqc_handle** out) { if (!bam_path) return NGSQC_E_ARG; return open_impl(out, bam_path, nullptr, 0, device, shard, n_shards); }
int ngsqc_open_memory_shard(const void* bam_bytes, size_t n_bytes, int device, int shard, int n_shards, ngsqc_handle** out) { return open_impl(out, nullptr, bam_bytes, n_bytes, device, shard, n_shards); }
// the records of a virtual-offset range [beg, end) (both record boundaries, e.g. from ngsqc_bai_range): only the BGZF members of the range go to the device
int ngsqc_open_range(const char* bam_path, int device, uint64_t beg_voff, uint64_t end_voff, ngsqc_handle** out)
{
	if (!bam_path) return NGSQC_E_ARG;
	RangeRequest rq; rq.voff[0] = beg_voff; rq.voff[1] = end_voff;
	return open_impl(out, bam_path, nullptr, 0, device, 0, 1, &rq);
}
// the first records of the file: the BGZF members of the BAM header and n_members behind them (BamReader::info looks at the first reads only, BamReader.cpp:626-641)
int ngsqc_open_head(const char* bam_path, int device, int64_t n_members, ngsqc_handle** out)
{
	if (!bam_path || n_members <= 0) return NGSQC_E_ARG;
	RangeRequest rq; rq.head_members = n_members;
	return open_impl(out, bam_path, nullptr, 0, device, 0, 1, &rq);
}
// the same for a set of named regions (1-based, closed): the range comes from <bam>.bai; NGSQC_E_IO "Could not load index of BAM/CRAM file ..." without one
int ngsqc_open_regions(const char* bam_path, int device, const ngsqc_named_region* regions, int64_t n_regions, ngsqc_handle** out)
{
	if (!bam_path || (!regions && n_regions > 0) || n_regions < 0) return NGSQC_E_ARG;
	RangeRequest rq; rq.by_name = true; rq.regions = regions; rq.n_regions = n_regions;
	return open_impl(out, bam_path, nullptr, 0, device, 0, 1, &rq);
}
int ngsqc_bai_range(const char* bam_path, const ngsqc_region* regions, int64_t n_regions, int32_t n_ref, uint64_t* beg_voff, uint64_t* end_voff, int32_t* found)
{
	if (!bam_path || (!regions && n_regions > 0) || !beg_voff || !end_voff || !found) return NGSQC_E_ARG;
	try
	{
		bool f = false;
		if (!ngsqc::bai_range(bam_path, regions, n_regions, n_ref, *beg_voff, *end_voff, f)) { g_open_error = std::string("Could not load index of BAM/CRAM file ") + bam_path; return NGSQC_E_IO; }   // BamReader.cpp:742-746
		*found = f ? 1 : 0;
		return NGSQC_OK;
	}
	catch (std::exception& e) { g_open_error = e.what(); return NGSQC_E_FORMAT; }
}
int ngsqc_bai_ranges(const char* bam_path, const ngsqc_region* regions, int64_t n_regions, int32_t n_ref, uint64_t* beg_voff, uint64_t* end_voff)
{
	if (!bam_path || n_regions < 0 || (n_regions > 0 && (!regions || !beg_voff || !end_voff))) return NGSQC_E_ARG;
	try
	{
		if (!ngsqc::bai_ranges(bam_path, regions, n_regions, n_ref, beg_voff, end_voff)) { g_open_error = std::string("Could not load index of BAM/CRAM file ") + bam_path; return NGSQC_E_IO; }
		return NGSQC_OK;
	}
	catch (std::exception& e) { g_open_error = e.what(); return NGSQC_E_FORMAT; }
}
int ngsqc_bgzf_scan(const void* bam_bytes, size_t n_bytes, int32_t n_threads, ngsqc_bgzf_member* out, int64_t cap, int64_t* n_members, int64_t* inflated_bytes)
{
	if (!bam_bytes || !n_members || cap < 0 || (!out && cap > 0)) return NGSQC_E_ARG;
	try
	{
		std::vector<BlockDesc> b; std::vector<uint32_t> c; std::vector<uint64_t> f; int64_t total = 0;
		bool pieces = false;
		scan_bgzf((const uint8_t*)bam_bytes, n_bytes, b, c, total, &f, n_threads > 0 ? n_threads : 1, &pieces);
		*n_members = (int64_t)b.size(); if (inflated_bytes) *inflated_bytes = total;
		for (int64_t i = 0; i < std::min<int64_t>(cap, (int64_t)b.size()); ++i)
			out[i] = ngsqc_bgzf_member{f[(size_t)i], b[(size_t)i].cpos, b[(size_t)i].upos, b[(size_t)i].clen, b[(size_t)i].usize, c[(size_t)i], pieces ? 1u : 0u};
		return NGSQC_OK;
	}
	catch (FormatError& e) { g_open_error = e.what(); return NGSQC_E_FORMAT; }
	catch (std::domain_error& e) { g_open_error = e.what(); return NGSQC_E_UNSUPPORTED; }
	catch (std::exception& e) { g_open_error = e.what(); return NGSQC_E_DEVICE; }
}
int ngsqc_set_reference(const char* fasta_path) { ngsqc::cram_set_reference(fasta_path); return NGSQC_OK; }
int32_t ngsqc_set_cram_skip_thread(int32_t flags) { if (flags >= 0 && (flags & ~(NGSQC_CRAM_SKIP_NAMES | NGSQC_CRAM_SKIP_TAGS))) return NGSQC_E_ARG; return ngsqc::cram_set_skip_thread(flags); }
int ngsqc_set_cram_skip(int32_t flags) { if (flags & ~(NGSQC_CRAM_SKIP_NAMES | NGSQC_CRAM_SKIP_TAGS)) return NGSQC_E_ARG; ngsqc::cram_set_skip(flags); return NGSQC_OK; }
int ngsqc_cram_to_bam(const char* cram_path, const char* bam_path, const ngsqc_named_region* regions, int64_t n_regions)
{
	if (!cram_path || !bam_path || n_regions < 0 || (!regions && n_regions > 0)) return NGSQC_E_ARG;
	try
	{
		std::ifstream f(cram_path, std::ios::binary);
		if (!f) { g_open_error = std::string("Could not open BAM/CRAM file ") + cram_path; return NGSQC_E_IO; }
		std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); ngsqc::ByteImage image; std::string err;
		if (!ngsqc::is_cram(d.data(), d.size())) { g_open_error = std::string("not a CRAM file: ") + cram_path; return NGSQC_E_FORMAT; }
		ngsqc::CramSelect sel;
		for (int64_t i = 0; i < n_regions; ++i) sel.regions.push_back(ngsqc::CramSelect::Region{regions[i].chr ? regions[i].chr : "", regions[i].start, regions[i].end});
		// NGSQC_CRAM_PLAN_DUMP=<file> (tests): the records with the quality arrays left blank, as the device path uploads them, and the plan of the quality blocks in <file>
		// (counts, then the arrays of CramQualPlan) - tests/test_cpu_cram.py replays the device kernels of cram_dev.hip on it
		const char* dump = getenv("NGSQC_CRAM_PLAN_DUMP"); ngsqc::CramQualPlan plan;
		const int rc = ngsqc::cram_to_bam_image(d.data(), d.size(), cram_path, image, err, &sel, dump ? &plan : nullptr);
		if (rc != NGSQC_OK) { g_open_error = err; return rc; }
		if (dump)
		{
			std::ofstream pf(dump, std::ios::binary | std::ios::trunc);
			const uint64_t hd[5] = {plan.jobs.size(), plan.tabs.size(), plan.syms.size(), plan.patches.size(), plan.out_bytes};
			static_assert(sizeof(ngsqc::CramQualPlan::Job) == 40 && sizeof(ngsqc::CramQualPlan::Patch) == 24, "plan layout");
			pf.write((const char*)hd, sizeof hd); pf.write((const char*)plan.jobs.data(), (std::streamsize)(plan.jobs.size() * 40)); pf.write((const char*)plan.tabs.data(), (std::streamsize)(plan.tabs.size() * 2));
			pf.write((const char*)plan.syms.data(), (std::streamsize)plan.syms.size()); pf.write((const char*)plan.patches.data(), (std::streamsize)(plan.patches.size() * 24));
		}
		std::ofstream o(bam_path, std::ios::binary | std::ios::trunc);
		if (o) o.write((const char*)image.data(), (std::streamsize)image.size());
		o.close();
		if (!o) { g_open_error = std::string("cannot write ") + bam_path; return NGSQC_E_IO; }
		return NGSQC_OK;
	}
	catch (std::exception& e) { g_open_error = e.what(); return NGSQC_E_DEVICE; }
}
int ngsqc_write_bai(ngsqc_handle* h, const char* bai_path) { return guarded(h, [&] { write_bai(h, bai_path); }); }
int ngsqc_write_csi(ngsqc_handle* h, const char* csi_path, int32_t min_shift) { return guarded(h, [&] { write_bai(h, csi_path, true, min_shift); }); }
static int index_assemble(const char* bai_path, int32_t n_ref, uint64_t first_record_voff, uint64_t end_voff, const ngsqc_bai_run* runs, int64_t n_runs,
                          const uint64_t* lidx, const int64_t* lidx_first, const int64_t* counts, bool csi, int min_shift, int depth);
int ngsqc_bai_assemble(const char* bai_path, int32_t n_ref, uint64_t first_record_voff, uint64_t end_voff, const ngsqc_bai_run* runs, int64_t n_runs,
                       const uint64_t* lidx, const int64_t* lidx_first, const int64_t* counts)
{
	return index_assemble(bai_path, n_ref, first_record_voff, end_voff, runs, n_runs, lidx, lidx_first, counts, false, 14, 5);
}
int ngsqc_csi_assemble(const char* csi_path, int32_t min_shift, int32_t depth, int32_t n_ref, uint64_t first_record_voff, uint64_t end_voff, const ngsqc_bai_run* runs, int64_t n_runs,
                       const uint64_t* lidx, const int64_t* lidx_first, const int64_t* counts)
{
	if (min_shift < 0 || min_shift > 31 || depth < 0 || depth > 10) return NGSQC_E_ARG;
	return index_assemble(csi_path, n_ref, first_record_voff, end_voff, runs, n_runs, lidx, lidx_first, counts, true, min_shift, depth);
}
static int index_assemble(const char* bai_path, int32_t n_ref, uint64_t first_record_voff, uint64_t end_voff, const ngsqc_bai_run* runs, int64_t n_runs,
                          const uint64_t* lidx, const int64_t* lidx_first, const int64_t* counts, bool csi, int min_shift, int depth)
{
	if (!bai_path || n_ref < 0 || n_runs < 0 || (!runs && n_runs > 0) || !lidx_first || !counts || (!lidx && lidx_first[n_ref] > 0)) return NGSQC_E_ARG;
	try
	{
		static_assert(sizeof(ngsqc_bai_run) == sizeof(BaiRunV), "run layout");
		std::vector<BaiRunV> rv((size_t)n_runs);
		if (n_runs) memcpy(rv.data(), runs, (size_t)n_runs * sizeof(BaiRunV));
		const std::vector<int64_t> first(lidx_first, lidx_first + n_ref + 1), cnt(counts, counts + 2 * ((size_t)n_ref + 1));
		const std::vector<uint64_t> L(lidx, lidx + first[(size_t)n_ref]);
		const std::string e = bai_assemble(bai_path, n_ref, first_record_voff, end_voff, rv, L, first, cnt, csi, min_shift, depth);
		if (e.empty()) return NGSQC_OK;
		g_open_error = e;
		return e.compare(0, 12, "cannot write") == 0 ? NGSQC_E_IO : NGSQC_E_FORMAT;
	}
	catch (std::exception& e) { g_open_error = e.what(); return NGSQC_E_DEVICE; }
}
int64_t ngsqc_header_text(const ngsqc_handle* h, char* out, int64_t cap)
{
	if (!h) return -1;
	const int64_t n = (int64_t)h->header_text.size();
	if (out && cap > 0) { const int64_t k = std::min(n, cap - 1); memcpy(out, h->header_text.data(), (size_t)k); out[k] = 0; }
	return n;
}


// the compressed image is on the device (a path is copied in the background while the first job already runs); h2d_ms of the timings is final behind this call
int ngsqc_upload_wait(ngsqc_handle* h) { return guarded(h, [&] { upload_finish(h); }); }

const char* ngsqc_last_error(const ngsqc_handle* h) { return h ? h->err.c_str() : g_open_error.c_str(); }
void ngsqc_set_open_error(const char* msg) { g_open_error = msg ? msg : ""; }   // (comm.hip reports through the same channel)
int ngsqc_n_ref(const ngsqc_handle* h) { return h ? (int)h->ref_names.size() : 0; }
const char* ngsqc_ref_name(const ngsqc_handle* h, int tid) { return (h && tid >= 0 && tid < (int)h->ref_names.size()) ? h->ref_names[tid].c_str() : nullptr; }
int64_t ngsqc_ref_len(const ngsqc_handle* h, int tid) { return (h && tid >= 0 && tid < (int)h->ref_lens.size()) ? h->ref_lens[tid] : -1; }
int64_t ngsqc_n_bgzf_blocks(const ngsqc_handle* h) { return h ? (int64_t)h->blocks.size() : 0; }
int64_t ngsqc_compressed_size(const ngsqc_handle* h) { return h ? (int64_t)h->csize : 0; }
int64_t ngsqc_inflated_size(ngsqc_handle* h) { return h ? h->total : 0; }
int64_t ngsqc_n_records(ngsqc_handle* h)
{
	int64_t n = 0;
	int rc = guarded(h, [&] { stream_tiles(h, [&](const TileCtx& c) { n += c.n_rec; return true; }); });
	return rc == NGSQC_OK ? n : (int64_t)rc;
}

int ngsqc_decode(ngsqc_handle* h) { return guarded(h, [&] { stream_tiles(h, [&](const TileCtx&) { return true; }); HIPCHK(hipStreamSynchronize(h->stream)); }); }
int ngsqc_drop_decoded(ngsqc_handle* h)
{
	// buffers stay allocated (re-used by the next decode); only the decoded STATE is dropped, so the next scan redoes K1+K2
	return guarded(h, [&] { h->decoded = false; h->cur_tile = -1; for (DepthSet& D : h->ds) D.depth_ready = false; h->n_rec = 0; });
}

int ngsqc_copy_inflated(ngsqc_handle* h, uint8_t* out, int64_t cap)
{
	return guarded(h, [&] {
		stream_tiles(h, [&](const TileCtx& c) {
			const int64_t lo = h->tile_u_lo, n = std::min(cap, lo + (h->tile_total - h->tile_prefix)) - lo;   // this tile's own bytes (without the carried prefix)
			if (n > 0) { HIPCHK(hipMemcpyAsync(out + lo, c.infl + h->tile_prefix, (size_t)n, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream)); }
			return true;
		});
	});
}
int ngsqc_copy_record_offsets(ngsqc_handle* h, int64_t* out, int64_t cap)
{
	return guarded(h, [&] {
		int64_t done = 0;
		stream_tiles(h, [&](const TileCtx& c) {
			const int64_t n = std::min(cap - done, c.n_rec);
			if (n > 0)
			{
				HIPCHK(hipMemcpyAsync(out + done, c.recoff, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
				for (int64_t i = 0; i < n; ++i) out[done + i] += h->tile_u_lo - h->tile_prefix;   // tile-local -> stream offset
				done += n;
			}
			return true;
		});
	});
}

static ngsqc_timings timings_of(const ngsqc_handle* h)
{
	ngsqc_timings t = h->tm;
	const char* e1 = getenv("NGSQC_CRAM_IGNORE_MD5"); const char* e2 = getenv("NGSQC_CRAM_NO_REFERENCE");
	t.switches = (h->verify_crc ? NGSQC_SW_VERIFY_CRC : 0) | (e1 && atoi(e1) != 0 ? NGSQC_SW_CRAM_IGNORE_MD5 : 0) | (e2 && atoi(e2) != 0 ? NGSQC_SW_CRAM_NO_REFERENCE : 0);
	return t;
}
int ngsqc_get_timings(const ngsqc_handle* h, ngsqc_timings* t) { if (!h || !t) return NGSQC_E_ARG; *t = timings_of(h); return NGSQC_OK; }
int ngsqc_get_timings_sized(const ngsqc_handle* h, void* t, size_t struct_size)
{
	if (!h || !t) return NGSQC_E_ARG;
	const ngsqc_timings full = timings_of(h);
	memcpy(t, &full, struct_size < sizeof(full) ? struct_size : sizeof(full));
	return NGSQC_OK;
}
int32_t ngsqc_abi_version(void) { return 6; }

int32_t ngsqc_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n < 0) { (void)hipGetLastError(); return 0; }
	return n;
}

const char* ngsqc_version(void) { return "ngsqc-hip 0.3 abi 6 (gfx950; K1 bgzf inflate + crc32, K2 bam record index, K3-K5 scan / pileup / read QC, K6 depth; tile stream)"; }

} // extern "C"

