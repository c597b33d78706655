// Host side of libngsqc_hip.so: the C ABI of include/ngsqc.h on top of the HIP kernels (K1 inflate.hip, K2 index.hip,
// K3-K5 scan.hip, K6 depth.hip). Owns the compressed image, the inflated stream, the record index and the depth array
// in HBM; one HIP stream per handle; stage times are taken with HIP events on that stream.
// There is no CPU fallback anywhere in this file: without a HIP device every compute entry point fails with
// NGSQC_E_DEVICE.
#include "common.h"
#include <algorithm>
#include <cstring>
#include <chrono>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace ngsqc;

namespace {
thread_local std::string g_open_error;

struct FormatError : std::runtime_error { using std::runtime_error::runtime_error; };
struct ArgError : std::runtime_error { using std::runtime_error::runtime_error; };
struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };

template <typename T> struct DevBuf
{
	T* p = nullptr; size_t n = 0;
	void alloc(size_t count) { release(); if (count) { HIPCHK(hipMalloc((void**)&p, count * sizeof(T))); n = count; } }
	void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
	void ensure(size_t count) { if (n < count) alloc(count); }   // keep a big-enough allocation (hipMalloc/hipFree of multi-GB buffers can stall for a second)
	void upload(const std::vector<T>& v, hipStream_t s) { alloc(v.size()); if (!v.empty()) HIPCHK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s)); }
	~DevBuf() { release(); }
	DevBuf() = default; DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
};

struct Timer
{
	hipEvent_t a = nullptr, b = nullptr; hipStream_t s;
	explicit Timer(hipStream_t st) : s(st) { HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); }
	~Timer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
	void start() { HIPCHK(hipEventRecord(a, s)); }
	double stop() { HIPCHK(hipEventRecord(b, s)); HIPCHK(hipEventSynchronize(b)); float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
} // namespace

struct ngsqc_handle
{
	std::string err, path;
	int device = 0; hipStream_t stream = nullptr, stream2 = nullptr; int n_cu = 256;
	std::vector<hipEvent_t> k1_events; DevBuf<unsigned long long> d_k1_work;   // K1 pipeline: 4 events + one queue head per member chunk
	size_t csize = 0;
	std::vector<BlockDesc> blocks; int64_t total = 0;
	DevBuf<uint8_t> d_comp; DevBuf<BlockDesc> d_blocks;
	// decoded state
	bool decoded = false;
	DevBuf<uint8_t> d_infl; DevBuf<BlockStatus> d_status; DevBuf<int64_t> d_recoff; int64_t n_rec = 0; int64_t first_rec = 0;
	std::vector<std::string> ref_names; std::vector<int64_t> ref_lens;
	// region / depth state of the last scan
	std::vector<ngsqc_region> regions; std::vector<int64_t> doff; std::vector<int32_t> rlen; int64_t n_slots = 0; int64_t roi_bases = 0;
	DevBuf<int32_t> d_reg_start, d_reg_end, d_reg_len, d_tid_first, d_tid_last; DevBuf<int64_t> d_doff; DevBuf<int32_t> d_depth;
	bool depth_ready = false;
	std::vector<BlockStatus> h_status; std::vector<int32_t> h_start; std::vector<int64_t> h_next;   // host scratch reused across decodes (no per-step page faults)
	DevBuf<int64_t> d_long; DevBuf<unsigned long long> d_counters; DevBuf<BlockDesc> d_tile_blocks; DevBuf<uint8_t> d_carry_tmp;
	// tiling: the inflated stream is processed in member ranges that fit HBM; tile-local offsets everywhere on the device
	std::vector<std::pair<int64_t, int64_t>> tiles;   // (first member, count)
	int cur_tile = -1; int64_t tile_prefix = 0, tile_total = 0, tile_u_lo = 0, tile_ord_base = 0;
	int64_t carry_len = 0, carry_src = 0, next_ord_base = 0, expected_abs = 0; int64_t n_rec_total = -1;
	DevBuf<uint32_t> d_tok; DevBuf<uint64_t> d_tok_off; DevBuf<uint32_t> d_tok_cnt; int64_t tok_first = -1, tok_n = -1;   // K1 token scratch, kept across decodes
	DevBuf<uint32_t> d_k1_order; int64_t k1_order_chunk = -1;   // queue order of the members inside each K1 chunk (largest first)
	ngsqc_timings tm{};
	// one BAM sharded over several handles: this handle owns the records that START inside members
	// [0, shard_own_members) of its (rebased) member table; the members behind them are only there to complete the last record
	int shard = 0, n_shards = 1;
	int64_t shard_own_members = -1;        // -1: not a shard (every record of the table is owned)
	int64_t shard_limit = -1;              // rebased inflated offset of the first byte that is NOT owned
	int64_t shard_u_base = 0;              // inflated offset (whole file) of the handle's first member
	int64_t shard_first_abs = -1, shard_exit_abs = -1; int shard_last_tile = -1;
	std::vector<int64_t> rq_len_hist, rq_cyc;   // results of the last ngsqc_scan_reads
	struct Partial;                        // state between ngsqc_scan_mapping_partial and ngsqc_scan_mapping_finish
	Partial* partial = nullptr;
};

namespace {

// ---- BGZF member table (host): SAM spec §4.1 ----
void scan_bgzf(const uint8_t* file, size_t n, std::vector<BlockDesc>& blocks, int64_t& total)
{
	if (n >= 4 && memcmp(file, "CRAM", 4) == 0) throw std::domain_error("CRAM input is not supported by the HIP path");
	size_t off = 0; uint64_t upos = 0;
	while (off < n)
	{
		if (off + 18 > n) throw FormatError("truncated BGZF header");
		const uint8_t* p = file + off;
		if (p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) throw FormatError("not a BGZF block (gzip member without BC extra field)");
		uint32_t xlen = rd16(p + 10), bsize = 0; bool found = false;
		size_t x = 12, xend = 12 + (size_t)xlen;
		if (off + xend > n) throw FormatError("truncated BGZF extra field");
		while (x + 4 <= xend) { uint16_t slen = rd16(p + x + 2); if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2) { bsize = rd16(p + x + 4) + 1u; found = true; } x += 4 + slen; }
		if (!found || bsize < xend + 8 || off + bsize > n) throw FormatError("invalid BGZF block size");
		uint32_t isize = rd32(p + bsize - 4);
		if (isize > 65536) throw FormatError("BGZF block inflates to more than 64 KiB");
		if (isize) blocks.push_back(BlockDesc{(uint64_t)(off + xend), upos, (uint32_t)(bsize - xend - 8), isize});
		upos += isize; off += bsize;
	}
	total = (int64_t)upos;
}

void init_device(ngsqc_handle* h, int device)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw std::runtime_error("no HIP device available (libngsqc_hip has no CPU fallback)");
	if (device < 0 || device >= n) throw ArgError("invalid HIP device ordinal");
	h->device = device;
	HIPCHK(hipSetDevice(device));
	HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
	HIPCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
	int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) h->n_cu = cu;
}

void check_status(ngsqc_handle* h, int64_t n_blocks)
{
	for (int64_t i = 0; i < n_blocks; ++i)
		if (h->h_status[(size_t)i].error) throw FormatError("BGZF inflate failed in block " + std::to_string(i) + " (code " + std::to_string(h->h_status[(size_t)i].error) + ")");
}

// K1 dispatcher. Default: two-phase (lane-per-member Huffman -> tokens, wave-per-member LZ77 resolve). A member whose
// token stream overflows its budget (clen + 64 tokens) makes the whole range fall back to the group kernel.
bool inflate_members(ngsqc_handle* h, int64_t first, int64_t n, const BlockDesc* d_desc, uint8_t* d_out_base)
{
	BlockStatus* d_st = h->d_status.p;   // status of the n members of this call
	const char* ev = getenv("NGSQC_INFLATE_VARIANT"); const int variant = ev ? atoi(ev) : 20;
	if (n <= 0) return true;
	if (variant >= 20)
	{
		bool order_dirty = false;
		if (h->tok_first != first || h->tok_n != n)
		{
			order_dirty = true;
			std::vector<uint64_t> off((size_t)n + 1, 0);
			for (int64_t i = 0; i < n; ++i) off[(size_t)i + 1] = off[(size_t)i] + (((uint64_t)h->blocks[(size_t)(first + i)].clen + 64 + 3) & ~3ull);
			h->d_tok_off.upload(off, h->stream);
			if (h->d_tok.n < (size_t)off[(size_t)n] + 16) h->d_tok.alloc((size_t)off[(size_t)n] + 16);
			if (h->d_tok_cnt.n < (size_t)n + 8) h->d_tok_cnt.alloc((size_t)n + 8);
			h->tok_first = first; h->tok_n = n;
		}
		// Phase 1 decodes one member per LANE, so a launch lasts as long as its slowest lane: members are cut into chunks of at
		// most one "round" (every decoder lane gets one member) and phase 2 of chunk c runs on a second stream while phase 1
		// of chunk c+1 decodes: a ragged last round no longer idles the chip (both kernels are VALU-bound, so the overlap itself
		// gains little). 6 phase-1 waves per CU leave LDS for the phase-2 workgroups.
		const char* pe = getenv("NGSQC_K1_PIPELINE"); const bool pipelined = !pe || atoi(pe) != 0;
		const char* se = getenv("NGSQC_K1_SORTED"); const bool sorted_queue = !se || atoi(se) != 0;
		const int64_t lanes = (int64_t)h->n_cu * (pipelined ? 6 : 7) * 64;
		const int64_t nch0 = pipelined ? std::max<int64_t>(1, (n + lanes - 1) / lanes) : 1;
		const int64_t chunk = (((n + nch0 - 1) / nch0) + 63) & ~63ll;   // equal chunks, whole waves
		const int64_t nch = (n + chunk - 1) / chunk;
		while ((int64_t)h->k1_events.size() < 4 * nch) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->k1_events.push_back(e); }
		if (h->k1_order_chunk != chunk || h->d_k1_order.n < (size_t)n || order_dirty)
		{
			// queue order inside every chunk: largest compressed size first (chunk-local indices)
			std::vector<uint32_t> ord((size_t)n);
			for (int64_t c0 = 0; c0 < n; c0 += chunk)
			{
				const int64_t cn = std::min<int64_t>(chunk, n - c0);
				for (int64_t i = 0; i < cn; ++i) ord[(size_t)(c0 + i)] = (uint32_t)i;
				std::stable_sort(ord.begin() + c0, ord.begin() + c0 + cn, [&](uint32_t a, uint32_t b) { return h->blocks[(size_t)(first + c0 + a)].clen > h->blocks[(size_t)(first + c0 + b)].clen; });
			}
			h->d_k1_order.upload(ord, h->stream); h->k1_order_chunk = chunk;
		}
		h->d_k1_work.ensure((size_t)nch);
		HIPCHK(hipMemsetAsync(h->d_k1_work.p, 0, (size_t)nch * sizeof(unsigned long long), h->stream));
		for (int64_t c = 0; c < nch; ++c)
		{
			const int64_t c0 = c * chunk, cn = std::min<int64_t>(chunk, n - c0);
			hipEvent_t* e4 = &h->k1_events[(size_t)(4 * c)];
			HIPCHK(hipEventRecord(e4[0], h->stream));
			launch_huff_tokens(h->d_comp.p, d_desc + c0, cn, d_st + c0, h->d_tok_off.p + c0, h->d_tok.p, h->d_tok_cnt.p + c0, h->d_k1_work.p + c, sorted_queue ? h->d_k1_order.p + c0 : nullptr, h->n_cu * (pipelined ? 6 : 7), h->stream);
			HIPCHK(hipEventRecord(e4[1], h->stream));
			hipStream_t s2 = pipelined ? h->stream2 : h->stream;
			if (pipelined) HIPCHK(hipStreamWaitEvent(s2, e4[1], 0));
			HIPCHK(hipEventRecord(e4[2], s2));
			launch_lz77_resolve(d_desc + c0, cn, d_out_base, d_st + c0, h->d_tok_off.p + c0, h->d_tok.p, h->d_tok_cnt.p + c0, s2);
			HIPCHK(hipEventRecord(e4[3], s2));
		}
		if (pipelined) HIPCHK(hipStreamWaitEvent(h->stream, h->k1_events[(size_t)(4 * (nch - 1) + 3)], 0));
		if (h->h_status.size() < (size_t)n) h->h_status.resize((size_t)n);
		HIPCHK(hipMemcpyAsync(h->h_status.data(), d_st, (size_t)n * sizeof(BlockStatus), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		for (int64_t c = 0; c < nch; ++c)
		{
			float ms = 0; hipEvent_t* e4 = &h->k1_events[(size_t)(4 * c)];
			HIPCHK(hipEventElapsedTime(&ms, e4[0], e4[1])); h->tm.inflate_huff_ms += ms;
			HIPCHK(hipEventElapsedTime(&ms, e4[2], e4[3])); h->tm.inflate_lz77_ms += ms;
		}
		h->tm.inflate_huff_launches += nch;
		bool overflow = false; for (int64_t i = 0; i < n; ++i) if (h->h_status[(size_t)i].error == 100) overflow = true;
		if (!overflow) { check_status(h, n); return true; }
	}
	launch_inflate(h->d_comp.p, d_desc, n, d_out_base, d_st, h->stream);
	if (h->h_status.size() < (size_t)n) h->h_status.resize((size_t)n);
	HIPCHK(hipMemcpyAsync(h->h_status.data(), d_st, (size_t)n * sizeof(BlockStatus), hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	check_status(h, n);
	return true;
}

// inflate the first members until the BAM header (magic, text, reference table) is complete; parse it
// avail: number of leading members whose compressed bytes are resident in d_comp (all of them for an unsharded handle).
// Returns false when more members are needed than are resident.
bool read_header(ngsqc_handle* h, int64_t avail)
{
	int64_t k = std::min<int64_t>(std::min<int64_t>(8, avail), (int64_t)h->blocks.size());
	if (avail < (int64_t)h->blocks.size()) k = avail;
	while (true)
	{
		int64_t bytes = k ? (int64_t)(h->blocks[k - 1].upos + h->blocks[k - 1].usize) : 0;
		DevBuf<uint8_t> tmp; tmp.alloc((size_t)bytes + 64);
		h->d_status.ensure((size_t)std::max<int64_t>(k, 1));
		inflate_members(h, 0, k, h->d_blocks.p, tmp.p);
		std::vector<uint8_t> hb((size_t)bytes);
		if (bytes) HIPCHK(hipMemcpy(hb.data(), tmp.p, (size_t)bytes, hipMemcpyDeviceToHost));
		bool complete = false;
		do
		{
			if (bytes < 12) break;
			if (memcmp(hb.data(), "BAM\1", 4) != 0) throw FormatError("Could not read header from BAM/CRAM file " + h->path);
			size_t o = 4; uint32_t l_text = rd32(&hb[o]); o += 4 + (size_t)l_text;
			if (o + 4 > (size_t)bytes) break;
			uint32_t n_ref = rd32(&hb[o]); o += 4;
			std::vector<std::string> names; std::vector<int64_t> lens; bool ok = true;
			for (uint32_t i = 0; i < n_ref; ++i)
			{
				if (o + 4 > (size_t)bytes) { ok = false; break; }
				uint32_t l_name = rd32(&hb[o]); o += 4;
				if (o + l_name + 4 > (size_t)bytes) { ok = false; break; }
				names.emplace_back((const char*)&hb[o], l_name ? l_name - 1 : 0); o += l_name;
				lens.push_back(rd32(&hb[o])); o += 4;
			}
			if (!ok) break;
			h->ref_names.swap(names); h->ref_lens.swap(lens); h->first_rec = (int64_t)o; complete = true;
		} while (false);
		if (complete) return true;
		if (k >= (int64_t)h->blocks.size()) throw FormatError("Could not read header from BAM/CRAM file " + h->path);
		if (k >= avail) return false;
		k = std::min<int64_t>(std::min<int64_t>(k * 4, avail), (int64_t)h->blocks.size());
	}
}

void upload_compressed(ngsqc_handle* h, const uint8_t* bytes, size_t beg, size_t end)
{
	const size_t n = end - beg;
	h->d_comp.alloc(n + 1024);
	HIPCHK(hipMemsetAsync(h->d_comp.p + n, 0, 1024, h->stream));
	if (n) HIPCHK(hipMemcpyAsync(h->d_comp.p, bytes + beg, n, hipMemcpyHostToDevice, h->stream));
}

constexpr int64_t SHARD_TAIL_MEMBERS = 64;   // members behind a shard that are inflated to complete its last record (NGSQC_SHARD_TAIL_MEMBERS)

void open_common(ngsqc_handle* h, const uint8_t* bytes, size_t n, int device, int shard, int n_shards)
{
	if (n_shards < 1 || shard < 0 || shard >= n_shards) throw ArgError("invalid shard index");
	h->csize = n;
	scan_bgzf(bytes, n, h->blocks, h->total);
	init_device(h, device);
	Timer t(h->stream); t.start();
	h->shard = shard; h->n_shards = n_shards;
	if (n_shards == 1)
	{
		upload_compressed(h, bytes, 0, n);
		h->d_blocks.upload(h->blocks, h->stream);
		h->tm.h2d_ms = t.stop();
		h->tm.compressed_bytes = (int64_t)n; h->tm.inflated_bytes = h->total;
		read_header(h, (int64_t)h->blocks.size());
		return;
	}
	// ---- header: only the first members are sent to the device ----
	const int64_t nb = (int64_t)h->blocks.size();
	for (int64_t k = std::min<int64_t>(8, nb);; k = std::min<int64_t>(k * 4, nb))
	{
		const size_t end = k ? (size_t)(h->blocks[(size_t)k - 1].cpos + h->blocks[(size_t)k - 1].clen) : 0;
		upload_compressed(h, bytes, 0, end);
		std::vector<BlockDesc> head(h->blocks.begin(), h->blocks.begin() + k);
		h->d_blocks.upload(head, h->stream);
		if (read_header(h, k)) break;
		if (k >= nb) throw FormatError("Could not read header from BAM/CRAM file " + h->path);
	}
	h->tok_first = -1; h->tok_n = -1;
	// ---- member range of this shard: equal compressed bytes, cut at member starts ----
	auto first_member_at = [&](int s) -> int64_t {
		if (s <= 0) return 0;
		if (s >= n_shards) return nb;
		const uint64_t target = (uint64_t)((double)n * (double)s / (double)n_shards);
		int64_t lo = 0, hi = nb;
		while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (h->blocks[(size_t)mid].cpos < target) lo = mid + 1; else hi = mid; }
		return lo;
	};
	const int64_t m0 = first_member_at(shard), m1 = first_member_at(shard + 1);
	int64_t tail = SHARD_TAIL_MEMBERS; if (const char* e = getenv("NGSQC_SHARD_TAIL_MEMBERS")) tail = std::max<int64_t>(0, atoll(e));
	const int64_t m_end = std::min<int64_t>(nb, m1 + (m1 > m0 ? tail : 0));
	std::vector<BlockDesc> own;
	size_t cbeg = 0, cend = 0; int64_t u0 = 0, u_own = 0, u_all = 0;
	if (m1 > m0)
	{
		cbeg = (size_t)(h->blocks[(size_t)m0].cpos & ~15ull);
		cend = (size_t)(h->blocks[(size_t)m_end - 1].cpos + h->blocks[(size_t)m_end - 1].clen);
		u0 = (int64_t)h->blocks[(size_t)m0].upos;
		u_own = (m1 < nb ? (int64_t)h->blocks[(size_t)m1].upos : h->total) - u0;
		u_all = (m_end < nb ? (int64_t)h->blocks[(size_t)m_end].upos : h->total) - u0;
		for (int64_t i = m0; i < m_end; ++i) { BlockDesc d = h->blocks[(size_t)i]; d.cpos -= cbeg; d.upos -= (uint64_t)u0; own.push_back(d); }
	}
	const int64_t first_rec_abs = h->first_rec;
	h->blocks.swap(own);
	h->shard_own_members = m1 - m0; h->shard_limit = u_own; h->shard_u_base = u0; h->total = u_all;
	h->first_rec = first_rec_abs >= u0 ? first_rec_abs - u0 : -1;   // shards behind the header: unknown, guessed by K2 and verified across shards
	if (m1 > m0 && first_rec_abs >= u0 + u_own) { h->blocks.clear(); h->shard_own_members = 0; h->shard_limit = 0; h->total = 0; cbeg = cend = 0; }   // header only: owns no record
	upload_compressed(h, bytes, cbeg, cend);
	h->csize = cend - cbeg;
	h->d_blocks.upload(h->blocks, h->stream);
	h->tm.h2d_ms = t.stop();
	h->tm.compressed_bytes = (int64_t)(cend - cbeg); h->tm.inflated_bytes = u_own;
}

// Member ranges ("tiles") whose inflated bytes + token scratch + record index fit the device. NGSQC_TILE_MEMBERS overrides
// (tests use tiny tiles to exercise the carry logic).
void plan_tiles(ngsqc_handle* h)
{
	if (!h->tiles.empty() || h->blocks.empty()) return;
	const int64_t nb = (int64_t)h->blocks.size();
	int64_t per_tile = nb;
	if (const char* e = getenv("NGSQC_TILE_MEMBERS")) per_tile = std::max<int64_t>(1, atoll(e));
	else
	{
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
		{
			const double per_member = 65536.0 + 4.0 * (65536.0 / 3.0 + 64.0) + 2.0 * 8.0 * 400.0;   // inflated + tokens + record index / long list
			int64_t fit = (int64_t)((double)free_b * 0.80 / per_member);
			per_tile = std::max<int64_t>(4096, std::min<int64_t>(nb, fit));
		}
	}
	for (int64_t f = 0; f < nb; f += per_tile) h->tiles.emplace_back(f, std::min<int64_t>(per_tile, nb - f));
}

// K1 + K2 for tile t. Tiles must be decoded in order (t == 0 or t == cur_tile + 1): a record that starts in one tile and
// ends in the next is carried as a prefix in front of the next tile's members.
void decode_tile(ngsqc_handle* h, int t)
{
	HIPCHK(hipSetDevice(h->device));
	plan_tiles(h);
	const bool dbg = getenv("NGSQC_DEBUG") != nullptr;
	if (t == h->cur_tile && h->decoded) return;
	if (t != 0 && t != h->cur_tile + 1) throw std::runtime_error("internal: tiles must be decoded in order");
	const bool last = t == (int)h->tiles.size() - 1;
	const int64_t first = h->tiles[(size_t)t].first, nm = h->tiles[(size_t)t].second;
	if (t == 0) { h->carry_len = 0; h->next_ord_base = 0; h->expected_abs = h->first_rec; }
	const int64_t u_lo = (int64_t)h->blocks[(size_t)first].upos;
	const int64_t u_hi = (int64_t)h->blocks[(size_t)(first + nm - 1)].upos + h->blocks[(size_t)(first + nm - 1)].usize;
	const int64_t prefix = h->carry_len;
	const int64_t total = prefix + (u_hi - u_lo);
	Timer tmr(h->stream);
	// ---- buffers ----
	h->d_infl.ensure((size_t)total + 64);   // (the carried prefix was staged in d_carry_tmp by finish_tile)
	if (prefix) HIPCHK(hipMemcpyAsync(h->d_infl.p, h->d_carry_tmp.p, (size_t)prefix, hipMemcpyDeviceToDevice, h->stream));
	// tile-local member descriptors: [0] = pseudo member covering the carried prefix (not inflated), then the members
	std::vector<BlockDesc> loc((size_t)nm + 1);
	loc[0] = BlockDesc{0, 0, 0, (uint32_t)prefix};
	for (int64_t i = 0; i < nm; ++i) { BlockDesc d = h->blocks[(size_t)(first + i)]; d.upos = (uint64_t)(prefix + ((int64_t)d.upos - u_lo)); loc[(size_t)i + 1] = d; }
	h->d_tile_blocks.ensure((size_t)nm + 1);
	HIPCHK(hipMemcpyAsync(h->d_tile_blocks.p, loc.data(), loc.size() * sizeof(BlockDesc), hipMemcpyHostToDevice, h->stream));
	h->d_status.ensure((size_t)nm + 1);
	// ---- K1 ----
	tmr.start();
	inflate_members(h, first, nm, h->d_tile_blocks.p + 1, h->d_infl.p);
	h->tm.inflate_ms += tmr.stop(); h->tm.inflate_launches++;
	// ---- K2 (tile-local coordinates; entry 0 is the prefix pseudo member) ----
	tmr.start();
	const int64_t ne = nm + 1;
	std::vector<int32_t>& start = h->h_start; if (start.size() < (size_t)ne) start.resize((size_t)ne);
	std::vector<int64_t>& next = h->h_next; if (next.size() < (size_t)ne) next.resize((size_t)ne);
	// a shard behind the file header does not know where its first record starts: every member is guessed and the first
	// plausible start anchors the chain (checked against the previous shard's chain exit by ngsqc_plan_shard_fix)
	const bool anchor_by_guess = t == 0 && h->first_rec < 0;
	int64_t exp0 = prefix ? 0 : (h->expected_abs - u_lo);   // local offset of the first record start of this tile
	for (int64_t b = 0; b < ne; ++b)
	{
		const int64_t lo = (int64_t)loc[(size_t)b].upos, hi = lo + loc[(size_t)b].usize;
		start[(size_t)b] = anchor_by_guess ? -2 : (hi <= exp0 ? -1 : (lo <= exp0 ? (int32_t)(exp0 - lo) : -2));
	}
	DevBuf<int32_t> d_start; d_start.alloc((size_t)ne); HIPCHK(hipMemcpyAsync(d_start.p, start.data(), (size_t)ne * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
	DevBuf<uint32_t> d_cnt; d_cnt.alloc((size_t)ne + 1);
	DevBuf<int64_t> d_next; d_next.alloc((size_t)ne + 1);
	DevBuf<uint32_t> d_bad; d_bad.alloc(1);
	int64_t from = 0; int rounds = 0; int64_t straddle = -1; bool found_start = !anchor_by_guess;
	const bool tail_may_cut_a_record = h->shard_own_members >= 0 && h->shard + 1 < h->n_shards;   // the members behind a shard end anywhere
	while (true)
	{
		HIPCHK(hipMemsetAsync(d_bad.p, 0, sizeof(uint32_t), h->stream));
		launch_index_count(h->d_infl.p, total, h->d_tile_blocks.p + from, ne - from, d_start.p + from, d_cnt.p + from, d_next.p + from, d_bad.p, (int32_t)h->ref_names.size(), h->stream);
		HIPCHK(hipMemcpyAsync(start.data() + from, d_start.p + from, (size_t)(ne - from) * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(next.data() + from, d_next.p + from, (size_t)(ne - from) * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		// exact verification of the chain: every member's exit must land on the next member's start
		int64_t expected = exp0; int64_t mismatch = -1; straddle = -1; bool anchored = !anchor_by_guess;
		for (int64_t b = 0; b < ne; ++b)
		{
			const int64_t lo = (int64_t)loc[(size_t)b].upos, hi = lo + loc[(size_t)b].usize;
			if (!anchored)
			{
				if (start[(size_t)b] < 0) continue;          // no plausible record start inside this member
				anchored = true; expected = lo + start[(size_t)b]; exp0 = expected;
			}
			const int32_t want = expected >= hi ? -1 : (int32_t)(expected - lo);
			if (start[(size_t)b] != want) { mismatch = b; start[(size_t)b] = want; break; }
			if (want >= 0)
			{
				const int64_t nx = next[(size_t)b];
				if (nx == -2) throw FormatError("Could not read next alignment in BAM/CRAM file " + h->path + " (corrupt record chain)");
				if (nx <= -10) { straddle = -(nx + 10); expected = INT64_MAX / 2; }   // the rest of the tile belongs to this record
				else expected = nx;
			}
		}
		if (mismatch < 0)
		{
			found_start = anchored;
			if (!anchored) { expected = total; exp0 = total; }   // no record starts in this tile at all
			if (straddle < 0 && expected != total && !(expected == INT64_MAX / 2)) throw FormatError("Could not read next alignment in BAM/CRAM file " + h->path + " (record chain does not end at a member boundary)");
			if (straddle >= 0 && last && !tail_may_cut_a_record) throw FormatError("Could not read next alignment in BAM/CRAM file " + h->path + " (truncated record)");
			break;
		}
		if (dbg) fprintf(stderr, "[ngsqc] tile %d: chain mismatch at entry %lld (round %d)\n", t, (long long)mismatch, rounds);
		if (++rounds > 100000) throw FormatError("could not resolve the BAM record chain");
		HIPCHK(hipMemcpyAsync(d_start.p + mismatch, &start[(size_t)mismatch], sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
		from = mismatch;
	}
	DevBuf<int64_t> d_base; d_base.alloc((size_t)ne + 1);
	DevBuf<uint8_t> d_tmp; d_tmp.alloc(scan_tmp_bytes(ne) + 64);
	launch_scan_counts(d_cnt.p, ne, d_base.p, d_tmp.p, h->stream);
	int64_t n_rec = 0;
	HIPCHK(hipMemcpyAsync(&n_rec, d_base.p + ne, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	h->d_recoff.ensure((size_t)std::max<int64_t>(n_rec, 1));
	launch_index_write(h->d_infl.p, total, h->d_tile_blocks.p, ne, d_start.p, d_base.p, h->d_recoff.p, h->stream);
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
	h->tm.index_ms += tmr.stop();
	// ---- publish tile state ----
	h->cur_tile = t; h->tile_prefix = prefix; h->tile_total = total; h->tile_u_lo = u_lo; h->tile_ord_base = h->next_ord_base;
	h->n_rec = n_rec; h->tm.n_records += n_rec;
	h->carry_src = straddle; h->carry_len = straddle >= 0 ? total - straddle : 0;
	h->expected_abs = u_hi;   // only meaningful when nothing is carried (the next record starts at the next tile's first byte)
	h->decoded = true;
}

// Called after a tile has been consumed and before the next one is decoded: stage the bytes of the straddling record.
void finish_tile(ngsqc_handle* h)
{
	if (h->carry_len > 0)
	{
		h->d_carry_tmp.ensure((size_t)h->carry_len + 64);
		HIPCHK(hipMemcpyAsync(h->d_carry_tmp.p, h->d_infl.p + h->carry_src, (size_t)h->carry_len, hipMemcpyDeviceToDevice, h->stream));
	}
	h->next_ord_base = h->tile_ord_base + h->n_rec;
}

void reset_decode_timings(ngsqc_handle* h) { h->tm.inflate_ms = 0; h->tm.index_ms = 0; h->tm.inflate_launches = 0; h->tm.n_records = 0; h->tm.inflate_huff_ms = 0; h->tm.inflate_lz77_ms = 0; h->tm.inflate_huff_launches = 0; }

// whole-file convenience used by the single-tile fast path and the test hooks
void do_decode(ngsqc_handle* h)
{
	plan_tiles(h);
	if (h->tiles.empty()) { h->decoded = true; h->n_rec = 0; return; }
	if (h->tiles.size() == 1) { if (!(h->decoded && h->cur_tile == 0)) { reset_decode_timings(h); decode_tile(h, 0); } return; }
	throw std::runtime_error("internal: do_decode on a multi-tile file");
}

// regions -> device tables. Regions must be sorted by start within a tid, non-overlapping, and each tid contiguous.
void setup_regions(ngsqc_handle* h, const ngsqc_region* regions, int64_t n)
{
	const int n_ref = (int)h->ref_names.size();
	h->regions.assign(regions, regions + (n > 0 ? n : 0));
	h->doff.assign((size_t)n + 1, 0); h->rlen.assign((size_t)n, 0);
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
		rs[i] = r.start; re[i] = r.end; h->rlen[i] = r.end - r.start + 1; h->doff[i] = slots;
		slots += (int64_t)h->rlen[i] + 1; bases += h->rlen[i];
	}
	h->doff[n] = slots; h->n_slots = slots; h->roi_bases = bases;
	h->d_reg_start.upload(rs, h->stream); h->d_reg_end.upload(re, h->stream); h->d_reg_len.upload(h->rlen, h->stream);
	h->d_tid_first.upload(tf, h->stream); h->d_tid_last.upload(tl, h->stream);
	std::vector<int64_t> doff(h->doff.begin(), h->doff.begin() + n);
	h->d_doff.upload(doff, h->stream);
	h->d_depth.ensure((size_t)slots + 1);
	HIPCHK(hipMemsetAsync(h->d_depth.p, 0, ((size_t)slots + 1) * sizeof(int32_t), h->stream));
	h->depth_ready = false;
}

void finalize_depth(ngsqc_handle* h)
{
	if (h->n_slots > 0)
	{
		DevBuf<uint8_t> tmp; tmp.alloc(scan_tmp_bytes(h->n_slots) + 64);
		launch_depth_prefix(h->d_depth.p, h->n_slots, tmp.p, h->stream);
		launch_depth_mark_spare(h->d_depth.p, h->d_doff.p, h->d_reg_len.p, (int64_t)h->regions.size(), h->stream);
		HIPCHK(hipStreamSynchronize(h->stream));
	}
	h->depth_ready = true;
}

struct GcTables { DevBuf<int32_t> start, end, bin, tf, tl; };

// Visit every tile in file order with the tile resident in HBM (K1+K2 done). A single-tile file that is already decoded
// is visited without redoing K1/K2 (the reference re-reads the file for every pass; we keep it).
template <class F> void for_each_tile(ngsqc_handle* h, F f)
{
	plan_tiles(h);
	const int nt = (int)h->tiles.size();
	if (nt == 0) return;
	if (nt == 1 && h->decoded && h->cur_tile == 0) { f(0); return; }
	reset_decode_timings(h);
	for (int t = 0; t < nt; ++t)
	{
		decode_tile(h, t);
		const bool go_on = f(t);
		if (!go_on || t == h->shard_last_tile) break;   // (a shard stops at the tile that holds the first record of the next shard)
		if (t + 1 < nt) finish_tile(h);
	}
}

// Order-dependent carries (running maximum of the read length, "a paired read has been seen") on the record prefix
// [0, f) / [0, pidx) of this handle; floor_max = running maximum carried in from earlier shards of the same BAM.
void run_prefix_fix(ngsqc_handle* h, ScanParams& sp, std::vector<unsigned long long>& dev, int64_t f, int64_t pidx, int gmax, int floor_max)
{
	if (f <= 0 && pidx <= 0) return;
	Timer t(h->stream); t.start();
	const unsigned long long carry0 = (unsigned long long)std::max(floor_max, 0);
	HIPCHK(hipMemcpyAsync(h->d_counters.p + A_FIX_CARRY, &carry0, sizeof(carry0), hipMemcpyHostToDevice, h->stream));
	const int64_t upto = std::max(f, pidx);
	// visit the tiles that hold records [0, upto) again (normally only tile 0, usually still resident)
	for_each_tile(h, [&](int) {
		h->d_long.ensure((size_t)std::max<int64_t>(h->n_rec, 1));
		sp.infl = h->d_infl.p; sp.total = h->tile_total; sp.recoff = h->d_recoff.p; sp.n_rec = h->n_rec; sp.ord_base = h->tile_ord_base;
		sp.long_list = h->d_long.p; sp.long_cap = h->n_rec;
		const int64_t lf = std::min<int64_t>(std::max<int64_t>(f - h->tile_ord_base, 0), h->n_rec), lp = std::min<int64_t>(std::max<int64_t>(pidx - h->tile_ord_base, 0), h->n_rec);
		launch_prefix_fix(sp, lf, lp, gmax, h->stream);
		HIPCHK(hipStreamSynchronize(h->stream));
		return h->tile_ord_base + h->n_rec < upto;
	});
	HIPCHK(hipMemcpyAsync(dev.data(), h->d_counters.p, dev.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	h->tm.scan_ms += t.stop();
}

void run_scan(ngsqc_handle* h, ScanParams& sp, std::vector<unsigned long long>& dev, bool do_fix = true)
{
	h->d_counters.ensure(A_DEV_TOTAL);
	std::vector<unsigned long long> init(A_DEV_TOTAL, 0ull); init[A_FIRST_PAIRED] = ~0ull;
	HIPCHK(hipMemcpyAsync(h->d_counters.p, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
	sp.counters = h->d_counters.p; sp.diff = h->d_depth.p; sp.n_ref = (int32_t)h->ref_names.size();
	h->tm.scan_kernel_ms = 0; h->tm.scan_launches = 0; h->tm.scan_ms = 0;
	auto bind_tile = [&]() {
		h->d_long.ensure((size_t)std::max<int64_t>(h->n_rec, 1));
		sp.infl = h->d_infl.p; sp.total = h->tile_total; sp.recoff = h->d_recoff.p; sp.n_rec = h->n_rec; sp.ord_base = h->tile_ord_base;
		sp.long_list = h->d_long.p; sp.long_cap = h->n_rec;
	};
	for_each_tile(h, [&](int) {
		bind_tile();
		Timer t(h->stream); t.start();
		HIPCHK(hipMemsetAsync(h->d_counters.p + A_LONG_COUNT, 0, sizeof(unsigned long long), h->stream));
		Timer tk(h->stream); tk.start();
		launch_scan(sp, h->stream);
		h->tm.scan_kernel_ms += tk.stop();
		unsigned long long n_long = 0;
		HIPCHK(hipMemcpyAsync(&n_long, h->d_counters.p + A_LONG_COUNT, sizeof(n_long), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		h->tm.scan_launches++;
		if (n_long) { tk.start(); launch_scan_long(sp, (int64_t)n_long, h->stream); h->tm.scan_kernel_ms += tk.stop(); h->tm.scan_launches++; }
		h->tm.scan_ms += t.stop();
		return true;
	});
	dev.assign(A_DEV_TOTAL, 0ull);
	HIPCHK(hipMemcpyAsync(dev.data(), h->d_counters.p, dev.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	if (sp.mode != MODE_DEPTH && do_fix)
	{
		const unsigned long long key = dev[A_FIRST_MAX_KEY];
		const int gmax = (int)(key >> 40);
		const int64_t f = key ? (int64_t)(0xFFFFFFFFFFull - (key & 0xFFFFFFFFFFull)) : 0;
		const int64_t pidx = (sp.mode != NGSQC_MODE_ROI && dev[A_FIRST_PAIRED] != ~0ull) ? (int64_t)dev[A_FIRST_PAIRED] : 0;
		run_prefix_fix(h, sp, dev, f, pidx, gmax, 0);
	}
	h->tm.scan_algorithmic_bytes = (int64_t)dev[A_ALG_BYTES];
}

template <typename F> int guarded(ngsqc_handle* h, F f)
{
	if (!h) return NGSQC_E_ARG;
	try { HIPCHK(hipSetDevice(h->device)); f(); return NGSQC_OK; }
	catch (FormatError& e) { h->err = e.what(); return NGSQC_E_FORMAT; }
	catch (ArgError& e) { h->err = e.what(); return NGSQC_E_ARG; }
	catch (IoError& e) { h->err = e.what(); return NGSQC_E_IO; }
	catch (std::domain_error& e) { h->err = e.what(); return NGSQC_E_UNSUPPORTED; }
	catch (std::exception& e) { h->err = e.what(); return NGSQC_E_DEVICE; }
}

int open_impl(ngsqc_handle** out, const char* path, const void* bytes, size_t n, int device, int shard = 0, int n_shards = 1)
{
	if (!out) return NGSQC_E_ARG;
	*out = nullptr;
	ngsqc_handle* h = new ngsqc_handle();
	int rc = NGSQC_OK;
	void* map = nullptr; size_t map_n = 0; int fd = -1;
	try
	{
		if (path)
		{
			h->path = path;
			fd = ::open(path, O_RDONLY);
			if (fd < 0) throw IoError(std::string("Could not open BAM/CRAM file ") + path);
			struct stat st; if (fstat(fd, &st) != 0) throw IoError(std::string("Could not open BAM/CRAM file ") + path);
			map_n = (size_t)st.st_size;
			if (map_n)
			{
				map = mmap(nullptr, map_n, PROT_READ, MAP_PRIVATE, fd, 0);
				if (map == MAP_FAILED) { map = nullptr; throw IoError(std::string("Could not open BAM/CRAM file ") + path); }
			}
			bytes = map; n = map_n;
		}
		else h->path = "<memory>";
		if (!bytes && n) throw ArgError("null BAM buffer");
		open_common(h, (const uint8_t*)bytes, n, device, shard, n_shards);
	}
	catch (FormatError& e) { g_open_error = e.what(); rc = NGSQC_E_FORMAT; }
	catch (ArgError& e) { g_open_error = e.what(); rc = NGSQC_E_ARG; }
	catch (IoError& e) { g_open_error = e.what(); rc = NGSQC_E_IO; }
	catch (std::domain_error& e) { g_open_error = e.what(); rc = NGSQC_E_UNSUPPORTED; }
	catch (std::exception& e) { g_open_error = e.what(); rc = NGSQC_E_DEVICE; }
	if (map) munmap(map, map_n);
	if (fd >= 0) ::close(fd);
	if (rc != NGSQC_OK) { ngsqc_close(h); return rc; }
	*out = h;
	return NGSQC_OK;
}

} // namespace

struct ngsqc_handle::Partial
{
	int mode = 0; bool yx = false; ScanParams sp{}; DevBuf<uint8_t> d_ns; GcTables gc; DevBuf<unsigned long long> d_gctab; DevBuf<double> d_gcover;
	std::vector<unsigned long long> dev;
};

namespace {
void mapping_setup(ngsqc_handle* h, const ngsqc_mapping_params* p, ngsqc_handle::Partial& st)
{
	if (!p) throw ArgError("null argument");
	if (p->mode < NGSQC_MODE_ROI || p->mode > NGSQC_MODE_WGS) throw ArgError("invalid mode");
	if (p->mode == NGSQC_MODE_ROI && (!p->regions || p->n_regions <= 0)) throw ArgError("target-region mode needs regions");
	const int n_ref = (int)h->ref_names.size();
	const bool use_regions = p->mode != NGSQC_MODE_NOROI && p->regions && p->n_regions > 0;
	setup_regions(h, use_regions ? p->regions : nullptr, use_regions ? p->n_regions : 0);
	ScanParams& sp = st.sp; sp = ScanParams{};
	sp.mode = p->mode; sp.min_mapq = p->min_mapq; sp.min_baseq = 0; sp.skip_mismapped = 0;
	sp.tid_x = p->tid_x; sp.tid_y = p->tid_y;
	st.mode = p->mode; const bool yx = st.yx = p->tid_x >= 0 && p->tid_x < n_ref && p->tid_y >= 0 && p->tid_y < n_ref;
	if (!yx) { sp.tid_x = -2; sp.tid_y = -2; }
	sp.len_x = yx ? h->ref_lens[p->tid_x] : 0; sp.len_y = yx ? h->ref_lens[p->tid_y] : 0;
	std::vector<uint8_t> ns((size_t)std::max(n_ref, 1), 0);
	if (p->tid_nonspecial) for (int i = 0; i < n_ref; ++i) ns[i] = p->tid_nonspecial[i];
	st.d_ns.upload(ns, h->stream); sp.tid_nonspecial = st.d_ns.p;
	sp.reg_start = h->d_reg_start.p; sp.reg_end = h->d_reg_end.p; sp.reg_doff = h->d_doff.p;
	sp.tid_reg_first = h->d_tid_first.p; sp.tid_reg_last = h->d_tid_last.p; sp.n_regions = (int64_t)h->regions.size();
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
		sp.gc_start = gc.start.p; sp.gc_end = gc.end.p; sp.gc_bin = gc.bin.p; sp.tid_gc_first = gc.tf.p; sp.tid_gc_last = gc.tl.p; sp.n_gc = n;
	}
	sp.gc_tab = d_gctab.p; sp.gc_over = d_gcover.p;
}

// device accumulators -> the reference's counters. gmax / paired_end: of the whole BAM (== this handle's unless it is a shard)
void mapping_counters(ngsqc_handle* h, ngsqc_handle::Partial& st, int gmax, bool paired_end, int64_t* counters, double* gc_reads)
{
	const std::vector<unsigned long long>& dev = st.dev; const bool yx = st.yx;
		// ---- device accumulators -> the reference's counters ----
	auto S = [&](int i) { return (int64_t)dev[i]; };
	for (int i = 0; i < NGSQC_NCOUNTERS; ++i) counters[i] = 0;
	counters[NGSQC_C_AL_TOTAL] = S(A_TOTAL); counters[NGSQC_C_AL_MAPPED] = S(A_MAPPED); counters[NGSQC_C_AL_ONTARGET] = S(A_ONTARGET);
	counters[NGSQC_C_AL_NEARTARGET] = S(A_NEAR); counters[NGSQC_C_AL_DUP] = S(A_DUP); counters[NGSQC_C_AL_PROPER_PAIRED] = S(A_PP);
	counters[NGSQC_C_INSERT_SIZE_READ_COUNT] = S(A_INS_CNT);
	counters[NGSQC_C_BASES_TRIMMED] = S(A_TOTAL) * gmax - S(A_SUM_LEN) - S(A_FIX_TRIM);
	counters[NGSQC_C_BASES_MAPPED] = S(A_BASES_MAPPED); counters[NGSQC_C_BASES_CLIPPED] = S(A_CLIPPED); counters[NGSQC_C_INSERT_SIZE_SUM] = S(A_INS_SUM);
	if (st.mode == NGSQC_MODE_ROI)
	{
		counters[NGSQC_C_BASES_USABLE] = S(A_USABLE);
		counters[NGSQC_C_BASES_USABLE_NO_OVERLAP] = S(A_NO_OVERLAP);
	}
	else
	{
		counters[NGSQC_C_BASES_USABLE] = S(A_USABLE) - S(A_CLIPPED);                        // Statistics.cpp:917 / :1183
		counters[NGSQC_C_BASES_USABLE_NO_OVERLAP] = (paired_end ? S(A_USABLE) - S(A_FIX_LEN) : 0) + S(A_NO_OVERLAP); // :879,:898-901
	}
	counters[NGSQC_C_BASES_USABLE_RAW] = S(A_USABLE_RAW); counters[NGSQC_C_BASES_USABLE_ROI] = S(A_USABLE_ROI);
	for (int i = 0; i < 5; ++i) counters[NGSQC_C_BASES_USABLE_DP0 + i] = S(A_DP0 + i);
	for (int i = 0; i < 4; ++i) counters[NGSQC_C_DP_DIST0 + i] = S(A_DD0 + i);
	counters[NGSQC_C_MAX_LENGTH] = gmax; counters[NGSQC_C_PAIRED_END] = paired_end ? 1 : 0;
	counters[NGSQC_C_ROI_BASES] = h->roi_bases;
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
}
} // namespace

extern "C" {

int ngsqc_open(const char* bam_path, int device, ngsqc_handle** out) { if (!bam_path) return NGSQC_E_ARG; return open_impl(out, bam_path, nullptr, 0, device); }
int ngsqc_open_memory(const void* bam_bytes, size_t n_bytes, int device, ngsqc_handle** out) { return open_impl(out, nullptr, bam_bytes, n_bytes, device); }
int ngsqc_open_shard(const char* bam_path, int device, int shard, int n_shards, ngsqc_handle** out) { if (!bam_path) return NGSQC_E_ARG; return open_impl(out, bam_path, nullptr, 0, device, shard, n_shards); }
int ngsqc_open_memory_shard(const void* bam_bytes, size_t n_bytes, int device, int shard, int n_shards, ngsqc_handle** out) { return open_impl(out, nullptr, bam_bytes, n_bytes, device, shard, n_shards); }

void ngsqc_close(ngsqc_handle* h)
{
	if (!h) return;
	if (h->stream) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
	if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
	for (hipEvent_t e : h->k1_events) (void)hipEventDestroy(e);
	delete h->partial;
	delete h;
}

const char* ngsqc_last_error(const ngsqc_handle* h) { return h ? h->err.c_str() : g_open_error.c_str(); }
int ngsqc_n_ref(const ngsqc_handle* h) { return h ? (int)h->ref_names.size() : 0; }
const char* ngsqc_ref_name(const ngsqc_handle* h, int tid) { return (h && tid >= 0 && tid < (int)h->ref_names.size()) ? h->ref_names[tid].c_str() : nullptr; }
int64_t ngsqc_ref_len(const ngsqc_handle* h, int tid) { return (h && tid >= 0 && tid < (int)h->ref_lens.size()) ? h->ref_lens[tid] : -1; }
int64_t ngsqc_n_bgzf_blocks(const ngsqc_handle* h) { return h ? (int64_t)h->blocks.size() : 0; }
int64_t ngsqc_compressed_size(const ngsqc_handle* h) { return h ? (int64_t)h->csize : 0; }
int64_t ngsqc_inflated_size(ngsqc_handle* h) { return h ? h->total : 0; }
int64_t ngsqc_n_records(ngsqc_handle* h)
{
	int64_t n = 0;
	int rc = guarded(h, [&] { for_each_tile(h, [&](int) { n += h->n_rec; return true; }); });
	return rc == NGSQC_OK ? n : (int64_t)rc;
}

int ngsqc_decode(ngsqc_handle* h) { return guarded(h, [&] { for_each_tile(h, [&](int) { return true; }); }); }
int ngsqc_drop_decoded(ngsqc_handle* h)
{
	// buffers stay allocated (re-used by the next decode); only the decoded STATE is dropped, so the next scan redoes K1+K2
	return guarded(h, [&] { h->decoded = false; h->cur_tile = -1; h->depth_ready = false; h->n_rec = 0; });
}

int ngsqc_copy_inflated(ngsqc_handle* h, uint8_t* out, int64_t cap)
{
	return guarded(h, [&] {
		for_each_tile(h, [&](int) {
			const int64_t lo = h->tile_u_lo, n = std::min(cap, lo + (h->tile_total - h->tile_prefix)) - lo;   // this tile's own bytes (without the carried prefix)
			if (n > 0) HIPCHK(hipMemcpy(out + lo, h->d_infl.p + h->tile_prefix, (size_t)n, hipMemcpyDeviceToHost));
			return true;
		});
	});
}
int ngsqc_copy_record_offsets(ngsqc_handle* h, int64_t* out, int64_t cap)
{
	return guarded(h, [&] {
		int64_t done = 0;
		for_each_tile(h, [&](int) {
			const int64_t n = std::min(cap - done, h->n_rec);
			if (n > 0)
			{
				HIPCHK(hipMemcpy(out + done, h->d_recoff.p, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
				for (int64_t i = 0; i < n; ++i) out[done + i] += h->tile_u_lo - h->tile_prefix;   // tile-local -> stream offset
				done += n;
			}
			return true;
		});
	});
}

int ngsqc_scan_mapping(ngsqc_handle* h, const ngsqc_mapping_params* p, int64_t* counters, double* gc_reads)
{
	return guarded(h, [&] {
		if (!p || !counters) throw ArgError("null argument");
		Timer total(h->stream); total.start();
		ngsqc_handle::Partial st;
		mapping_setup(h, p, st);
		run_scan(h, st.sp, st.dev);
		Timer fin(h->stream); fin.start();
		finalize_depth(h);
		h->tm.finalize_ms = fin.stop();
		mapping_counters(h, st, (int)(st.dev[A_FIRST_MAX_KEY] >> 40), st.dev[A_FIRST_PAIRED] != ~0ull, counters, gc_reads);
		h->tm.total_ms = total.stop();
	});
}

// ---- one BAM sharded over several handles: local scan, tiny exchange, local fix-up, additive counters ----
int ngsqc_scan_mapping_partial(ngsqc_handle* h, const ngsqc_mapping_params* p, ngsqc_shard_summary* out)
{
	return guarded(h, [&] {
		if (!p || !out) throw ArgError("null argument");
		Timer total(h->stream); total.start();
		delete h->partial; h->partial = new ngsqc_handle::Partial();
		ngsqc_handle::Partial& st = *h->partial;
		mapping_setup(h, p, st);
		run_scan(h, st.sp, st.dev, false);
		const unsigned long long key = st.dev[A_FIRST_MAX_KEY];
		out->n_records = h->tm.n_records;
		out->first_abs = h->shard_own_members >= 0 ? h->shard_first_abs : (h->tm.n_records ? h->first_rec : -1);
		out->exit_abs = h->shard_own_members >= 0 ? h->shard_exit_abs : (h->tm.n_records ? h->total : -1);
		out->max_len = (int64_t)(key >> 40);
		out->first_max_ord = key ? (int64_t)(0xFFFFFFFFFFull - (key & 0xFFFFFFFFFFull)) : -1;
		out->first_paired_ord = st.dev[A_FIRST_PAIRED] != ~0ull ? (int64_t)st.dev[A_FIRST_PAIRED] : -1;
		h->tm.total_ms = total.stop();
	});
}

int ngsqc_scan_mapping_finish(ngsqc_handle* h, const ngsqc_shard_fix* fix, int64_t* counters, double* gc_reads)
{
	return guarded(h, [&] {
		if (!fix || !counters) throw ArgError("null argument");
		if (!h->partial) throw ArgError("ngsqc_scan_mapping_finish without ngsqc_scan_mapping_partial");
		ngsqc_handle::Partial& st = *h->partial;
		Timer total(h->stream); total.start();
		run_prefix_fix(h, st.sp, st.dev, fix->trim_upto, st.mode != NGSQC_MODE_ROI ? fix->paired_upto : 0, (int)fix->gmax, (int)fix->floor_max);
		mapping_counters(h, st, (int)fix->gmax, fix->paired_end != 0, counters, gc_reads);
		h->tm.total_ms += total.stop();
	});
}

int ngsqc_depth_device(ngsqc_handle* h, void** dev_ptr, int64_t* n_slots)
{
	return guarded(h, [&] {
		if (!dev_ptr || !n_slots) throw ArgError("null argument");
		if (h->depth_ready) throw ArgError("the depth array is already finalized (prefix-summed)");
		HIPCHK(hipStreamSynchronize(h->stream));
		*dev_ptr = h->d_depth.p; *n_slots = h->n_slots;
	});
}
int ngsqc_depth_diff_copy(ngsqc_handle* h, int32_t* out, int64_t cap)
{
	return guarded(h, [&] {
		if (h->depth_ready) throw ArgError("the depth array is already finalized (prefix-summed)");
		if (cap < h->n_slots || (!out && h->n_slots)) throw ArgError("depth buffer too small");
		if (h->n_slots) HIPCHK(hipMemcpyAsync(out, h->d_depth.p, (size_t)h->n_slots * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	});
}
int ngsqc_depth_diff_set(ngsqc_handle* h, const int32_t* in, int64_t n)
{
	return guarded(h, [&] {
		if (h->depth_ready) throw ArgError("the depth array is already finalized (prefix-summed)");
		if (n != h->n_slots || (!in && n)) throw ArgError("depth buffer size mismatch");
		if (n) HIPCHK(hipMemcpyAsync(h->d_depth.p, in, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	});
}
int ngsqc_depth_finalize(ngsqc_handle* h)
{
	return guarded(h, [&] {
		if (h->depth_ready) return;
		Timer fin(h->stream); fin.start();
		finalize_depth(h);
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
		const int n_ref = (int)h->ref_names.size();
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
		DevBuf<int32_t> d_pos, d_tf, d_tl, d_bucket; DevBuf<int64_t> d_tb0;
		d_pos.upload(pos, h->stream); d_tf.upload(tf, h->stream); d_tl.upload(tl, h->stream); d_bucket.upload(bucket, h->stream); d_tb0.upload(tb0, h->stream);
		DevBuf<uint32_t> d_cnt; d_cnt.alloc((size_t)n_sites * 8);
		HIPCHK(hipMemsetAsync(d_cnt.p, 0, (size_t)n_sites * 8 * sizeof(uint32_t), h->stream));
		DevBuf<unsigned long long> d_nlong; d_nlong.alloc(1);
		for_each_tile(h, [&](int) {
			h->d_long.ensure((size_t)std::max<int64_t>(h->n_rec, 1));
			HIPCHK(hipMemsetAsync(d_nlong.p, 0, sizeof(unsigned long long), h->stream));
			launch_pileup(h->d_infl.p, h->d_recoff.p, h->n_rec, n_ref, d_pos.p, d_tf.p, d_tl.p, d_bucket.p, d_tb0.p, min_mapq, min_baseq, include_not_properly_paired ? 1 : 0, d_cnt.p, h->d_long.p, d_nlong.p, h->stream);
			unsigned long long n_long = 0;
			HIPCHK(hipMemcpyAsync(&n_long, d_nlong.p, sizeof(n_long), hipMemcpyDeviceToHost, h->stream));
			HIPCHK(hipStreamSynchronize(h->stream));
			if (n_long) { launch_pileup_long(h->d_infl.p, h->d_recoff.p, h->d_long.p, (int64_t)n_long, d_pos.p, d_tl.p, d_bucket.p, d_tb0.p, min_baseq, d_cnt.p, h->stream); HIPCHK(hipStreamSynchronize(h->stream)); }
			return true;
		});
		std::vector<uint32_t> out((size_t)n_sites * 8);
		HIPCHK(hipMemcpyAsync(out.data(), d_cnt.p, out.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		for (size_t i = 0; i < out.size(); ++i) counts[i] = (int64_t)out[i];
	});
}

int ngsqc_scan_reads(ngsqc_handle* h, int32_t single_end, ngsqc_read_stats* st)
{
	return guarded(h, [&] {
		if (!st) throw ArgError("null argument");
		// pass 1: longest counted read (sizes the read-length histogram)
		DevBuf<unsigned long long> d_max; d_max.alloc(1);
		HIPCHK(hipMemsetAsync(d_max.p, 0, sizeof(unsigned long long), h->stream));
		for_each_tile(h, [&](int) { launch_reads_max(h->d_infl.p, h->d_recoff.p, h->n_rec, d_max.p, h->stream); HIPCHK(hipStreamSynchronize(h->stream)); return true; });
		unsigned long long mx = 0;
		HIPCHK(hipMemcpyAsync(&mx, d_max.p, sizeof(mx), hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
		const int64_t len_cap = (int64_t)mx;
		DevBuf<unsigned long long> d_acc, d_len, d_cyc; d_acc.alloc(RA_TOTAL); d_len.alloc((size_t)len_cap + 1); d_cyc.alloc((size_t)RQ_CYC * 7);
		HIPCHK(hipMemsetAsync(d_acc.p, 0, RA_TOTAL * sizeof(unsigned long long), h->stream));
		HIPCHK(hipMemsetAsync(d_len.p, 0, ((size_t)len_cap + 1) * sizeof(unsigned long long), h->stream));
		HIPCHK(hipMemsetAsync(d_cyc.p, 0, (size_t)RQ_CYC * 7 * sizeof(unsigned long long), h->stream));
		// pass 2
		for_each_tile(h, [&](int) { launch_reads(h->d_infl.p, h->d_recoff.p, h->n_rec, single_end ? 1 : 0, d_acc.p, d_len.p, len_cap, d_cyc.p, h->stream); HIPCHK(hipStreamSynchronize(h->stream)); return true; });
		std::vector<unsigned long long> acc(RA_TOTAL), len((size_t)len_cap + 1), cyc((size_t)RQ_CYC * 7);
		HIPCHK(hipMemcpyAsync(acc.data(), d_acc.p, acc.size() * 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(len.data(), d_len.p, len.size() * 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(cyc.data(), d_cyc.p, cyc.size() * 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		memset(st, 0, sizeof(*st));
		st->c_forward = (int64_t)acc[RA_FWD]; st->c_reverse = (int64_t)acc[RA_REV]; st->bases_sequenced = (int64_t)acc[RA_BASES];
		for (int i = 0; i < 5; ++i) st->bases[i] = (int64_t)acc[RA_A + i];
		for (int i = 0; i < 100; ++i) { st->base_qualities[i] = (int64_t)acc[RA_BQ0 + i]; st->read_qualities[i] = (int64_t)acc[RA_RQ0 + i]; }
		for (int i = 0; i < 60; ++i) { st->qscore_dist_r1[i] = (int64_t)acc[RA_QD0 + i]; st->qscore_dist_r2[i] = (int64_t)acc[RA_QD0 + 60 + i]; }
		st->max_cycles = len_cap; st->n_unknown_base = (int64_t)acc[RA_BAD_BASE]; st->n_quality_out_of_range = (int64_t)acc[RA_BAD_QUAL];
		h->rq_len_hist.assign(len.begin(), len.end()); h->rq_cyc.assign(cyc.begin(), cyc.end());
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
	if (!p || !p->regions || p->n_regions <= 0) throw ArgError("depth scan needs regions");
	Timer total(h->stream); total.start();
	setup_regions(h, p->regions, p->n_regions);
	ScanParams sp{};
	sp.mode = MODE_DEPTH; sp.min_mapq = p->min_mapq; sp.min_baseq = p->min_baseq; sp.skip_mismapped = p->skip_mismapped;
	sp.tid_x = -2; sp.tid_y = -2;
	sp.reg_start = h->d_reg_start.p; sp.reg_end = h->d_reg_end.p; sp.reg_doff = h->d_doff.p;
	sp.tid_reg_first = h->d_tid_first.p; sp.tid_reg_last = h->d_tid_last.p; sp.n_regions = (int64_t)h->regions.size();
	std::vector<unsigned long long> dev;
	run_scan(h, sp, dev);
	if (finalize) { Timer fin(h->stream); fin.start(); finalize_depth(h); h->tm.finalize_ms = fin.stop(); }
	h->tm.total_ms = total.stop();
}
} // namespace

int ngsqc_scan_depth(ngsqc_handle* h, const ngsqc_depth_params* p) { return guarded(h, [&] { depth_scan(h, p, true); }); }
// shard variant: leaves the un-prefixed difference array (additive over shards: ngsqc_depth_device / _diff_copy / _diff_set, then ngsqc_depth_finalize)
int ngsqc_scan_depth_partial(ngsqc_handle* h, const ngsqc_depth_params* p) { return guarded(h, [&] { depth_scan(h, p, false); }); }

int ngsqc_depth_stats(ngsqc_handle* h, int32_t hist_cap, int64_t half_depth, int64_t* hist, int64_t* covered)
{
	return guarded(h, [&] {
		if (!h->depth_ready) throw ArgError("no depth array: run ngsqc_scan_mapping / ngsqc_scan_depth first");
		if (hist_cap < 0 || hist_cap > 30000 || !hist || !covered) throw ArgError("invalid histogram request");
		DevBuf<unsigned long long> d_hist; d_hist.alloc((size_t)hist_cap + 2);
		HIPCHK(hipMemsetAsync(d_hist.p, 0, ((size_t)hist_cap + 2) * sizeof(unsigned long long), h->stream));
		launch_depth_hist(h->d_depth.p, h->n_slots, hist_cap, half_depth, d_hist.p, d_hist.p + hist_cap + 1, h->stream);
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
		if (!h->depth_ready) throw ArgError("no depth array: run ngsqc_scan_mapping / ngsqc_scan_depth first");
		if (cap < h->roi_bases) throw ArgError("depth buffer too small");
		if (h->roi_bases == 0) return;
		DevBuf<int32_t> d_out; d_out.alloc((size_t)h->roi_bases);
		launch_depth_compact(h->d_depth.p, h->d_doff.p, h->d_reg_len.p, (int64_t)h->regions.size(), d_out.p, h->stream);
		HIPCHK(hipMemcpyAsync(out, d_out.p, (size_t)h->roi_bases * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	});
}

namespace {
// locate each line inside the scanned (merged) regions: slot offset of its first base
void locate_lines(ngsqc_handle* h, const ngsqc_region* lines, int64_t n, std::vector<int64_t>& slot, std::vector<int32_t>& len, std::vector<int32_t>& lstart)
{
	slot.resize((size_t)n); len.resize((size_t)n); lstart.resize((size_t)n);
	const auto& R = h->regions;
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
		slot[i] = h->doff[lo] + (l.start - R[lo].start); len[i] = l.end - l.start + 1; lstart[i] = l.start;
	}
}
}

int ngsqc_region_sums(ngsqc_handle* h, const ngsqc_region* lines, int64_t n_lines, int64_t* sums)
{
	return guarded(h, [&] {
		if (!h->depth_ready) throw ArgError("no depth array: run ngsqc_scan_depth first");
		if (n_lines <= 0) return;
		if (!lines || !sums) throw ArgError("null argument");
		std::vector<int64_t> slot; std::vector<int32_t> len, ls;
		locate_lines(h, lines, n_lines, slot, len, ls);
		DevBuf<int64_t> d_slot; d_slot.upload(slot, h->stream);
		DevBuf<int32_t> d_len; d_len.upload(len, h->stream);
		DevBuf<long long> d_sums; d_sums.alloc((size_t)n_lines);
		launch_line_sums(h->d_depth.p, d_slot.p, d_len.p, n_lines, d_sums.p, h->stream);
		HIPCHK(hipMemcpyAsync(sums, d_sums.p, (size_t)n_lines * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	});
}

int ngsqc_lowhigh_runs(ngsqc_handle* h, const ngsqc_region* lines, int64_t n_lines, int32_t cutoff, int32_t is_high, int32_t saturate254,
                       ngsqc_run* runs, int64_t cap, int64_t* n_runs)
{
	return guarded(h, [&] {
		if (!h->depth_ready) throw ArgError("no depth array: run ngsqc_scan_depth first");
		if (!n_runs) throw ArgError("null argument");
		*n_runs = 0;
		if (n_lines <= 0) return;
		std::vector<int64_t> slot; std::vector<int32_t> len, ls;
		locate_lines(h, lines, n_lines, slot, len, ls);
		DevBuf<int64_t> d_slot; d_slot.upload(slot, h->stream);
		DevBuf<int32_t> d_len; d_len.upload(len, h->stream);
		DevBuf<int32_t> d_ls; d_ls.upload(ls, h->stream);
		DevBuf<uint32_t> d_cnt; d_cnt.alloc((size_t)n_lines + 1);
		DevBuf<int64_t> d_base; d_base.alloc((size_t)n_lines + 1);
		DevBuf<uint8_t> d_tmp; d_tmp.alloc(scan_tmp_bytes(n_lines) + 64);
		launch_line_runs(false, h->d_depth.p, d_slot.p, d_len.p, d_ls.p, n_lines, cutoff, is_high, saturate254, d_cnt.p, nullptr, nullptr, h->stream);
		launch_scan_counts(d_cnt.p, n_lines, d_base.p, d_tmp.p, h->stream);
		int64_t total = 0;
		HIPCHK(hipMemcpyAsync(&total, d_base.p + n_lines, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		*n_runs = total;
		if (!runs || cap < total || total == 0) return;
		DevBuf<ngsqc_run> d_runs; d_runs.alloc((size_t)total);
		launch_line_runs(true, h->d_depth.p, d_slot.p, d_len.p, d_ls.p, n_lines, cutoff, is_high, saturate254, d_cnt.p, d_base.p, d_runs.p, h->stream);
		HIPCHK(hipMemcpyAsync(runs, d_runs.p, (size_t)total * sizeof(ngsqc_run), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	});
}

int ngsqc_get_timings(const ngsqc_handle* h, ngsqc_timings* t) { if (!h || !t) return NGSQC_E_ARG; *t = h->tm; return NGSQC_OK; }

const char* ngsqc_version(void) { return "ngsqc-hip 0.1 (gfx950; K1 bgzf_inflate, K2 bam_record_index, K3-K5 scan, K6 depth)"; }

} // extern "C"
