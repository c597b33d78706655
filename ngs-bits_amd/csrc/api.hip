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
#include "common.h"
#include <memory>
#include <condition_variable>
#include <mutex>
#include <algorithm>
#include <cstring>
#include <chrono>
#include <atomic>
#include <functional>
#include <thread>
#include <deque>
#include <fstream>
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

// Giving tens of GB back to the driver takes about a second (0.9 - 1.2 s for the buffers of a 60 GB BAM, profiles/r03_tool_probe.txt): large buffers are freed by a
// background thread, so ngsqc_close returns at once; a tool that exits right behind its last close never pays (the driver reclaims a dead process's memory itself), a
// process that goes on opening handles finds the memory free again a moment later (an allocation that fails waits for the thread and tries once more).
struct Reaper
{
	std::mutex mu; std::condition_variable cv; std::deque<std::function<void()>> q; std::thread th; bool stop = false, busy = false; int held = 0;
	// (a hipFree of tens of GB holds the runtime's memory lock for its whole duration: while a handle is being closed the thread holds still, so that the closing
	// thread's own small frees and stream / event teardown do not queue up behind it)
	void hold() { std::lock_guard<std::mutex> g(mu); ++held; }
	void unhold() { { std::lock_guard<std::mutex> g(mu); --held; } cv.notify_all(); }
	void push(void* p)
	{
		int dev = 0; (void)hipGetDevice(&dev);
		task([p, dev] { (void)hipSetDevice(dev); (void)hipFree(p); });
	}
	void task(std::function<void()> f)   // (also: unmapping a file of tens of GB - one page-table entry per 4 KB that a copy went through)
	{
		std::lock_guard<std::mutex> g(mu);
		q.push_back(std::move(f));
		if (!th.joinable()) th = std::thread([this] { run(); });
		cv.notify_all();
	}
	void run()
	{
		std::unique_lock<std::mutex> lk(mu);
		for (;;)
		{
			cv.wait(lk, [&] { return stop || (!q.empty() && held == 0); });
			if (stop) return;   // (the process is going: what is still queued goes with it)
			const std::function<void()> f = std::move(q.front()); q.pop_front(); busy = true;
			lk.unlock(); f(); lk.lock();
			busy = false; cv.notify_all();
		}
	}
	void drain() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return (q.empty() && !busy) || held > 0; }); }
	~Reaper() { { std::lock_guard<std::mutex> g(mu); stop = true; } cv.notify_all(); if (th.joinable()) th.join(); }
};
Reaper& reaper() { static Reaper r; return r; }
constexpr size_t REAP_MIN_BYTES = (size_t)64 << 10;   // (a hipFree waits for the device and costs 5 - 20 ms whatever its size: a handle has about forty buffers)

template <typename T> struct DevBuf
{
	T* p = nullptr; size_t n = 0;
	void alloc(size_t count)
	{
		release();
		if (!count) return;
		hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
		if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); reaper().drain(); e = hipMalloc((void**)&p, count * sizeof(T)); }   // (memory that is still on its way back)
		if (e != hipSuccess) { p = nullptr; throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " at hipMalloc of " + std::to_string(count * sizeof(T)) + " bytes"); }
		n = count;
	}
	void release() { if (p) { if (n * sizeof(T) >= REAP_MIN_BYTES) reaper().push(p); else (void)hipFree(p); p = nullptr; n = 0; } }
	void ensure(size_t count) { if (n < count) alloc(count); }
	void ensure_slack(size_t count) { if (n < count) alloc(count + count / 4); }   // per-tile scratch: growing it means hipFree, and hipFree waits for every queued kernel of the device   // keep a big-enough allocation (hipMalloc/hipFree of multi-GB buffers can stall for a second)
	void upload(const std::vector<T>& v, hipStream_t s) { ensure(v.size()); if (!v.empty()) HIPCHK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s)); }
	~DevBuf() { release(); }
	DevBuf() = default; DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
};

// pinned host memory for the per-tile D2H / H2D exchanges (pageable copies of a few MB cost ~1 ms each)
template <typename T> struct PinBuf
{
	T* p = nullptr; size_t n = 0;
	void ensure(size_t count) { if (n >= count) return; release(); HIPCHK(hipHostMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T), hipHostMallocDefault)); n = count; }
	void release() { if (p) { (void)hipHostFree(p); p = nullptr; n = 0; } }
	~PinBuf() { release(); }
	PinBuf() = default; PinBuf(const PinBuf&) = delete; PinBuf& operator=(const PinBuf&) = delete;
};

struct Timer
{
	hipEvent_t a = nullptr, b = nullptr; hipStream_t s;
	explicit Timer(hipStream_t st) : s(st) { HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); }
	~Timer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
	void start() { HIPCHK(hipEventRecord(a, s)); }
	double stop() { HIPCHK(hipEventRecord(b, s)); HIPCHK(hipEventSynchronize(b)); float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b)); return ms; }
	void mark() { HIPCHK(hipEventRecord(b, s)); }   // end of the interval without waiting for it
	double elapsed() { HIPCHK(hipEventSynchronize(b)); float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

// HIP-event intervals on a stream whose durations are only read after the tile loop (round 5: a Timer::stop() is a host wait - eight of them per tile kept the
// device idle between the kernels of a tile). begin() / end() record; resolve() adds every interval to the sums it was opened for.
struct EvLog
{
	struct Iv { hipEvent_t a, b; double* sum[2]; };
	std::vector<hipEvent_t> pool; size_t used = 0; std::vector<Iv> open;
	hipEvent_t get() { if (used == pool.size()) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); pool.push_back(e); } return pool[used++]; }
	size_t begin(hipStream_t s, double* sum0, double* sum1 = nullptr) { Iv iv{get(), get(), {sum0, sum1}}; HIPCHK(hipEventRecord(iv.a, s)); open.push_back(iv); return open.size() - 1; }
	void end(size_t id, hipStream_t s) { HIPCHK(hipEventRecord(open[id].b, s)); }
	void resolve()
	{
		for (Iv& iv : open)
		{
			float ms = 0;
			if (hipEventSynchronize(iv.b) == hipSuccess && hipEventElapsedTime(&ms, iv.a, iv.b) == hipSuccess) { for (double* q : iv.sum) if (q) *q += ms; }
			else (void)hipGetLastError();
		}
		open.clear(); used = 0;
	}
	void discard() { open.clear(); used = 0; (void)hipGetLastError(); }
	~EvLog() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
};

double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

constexpr int K1_CHUNK_WAVES_PER_CU = 6;   // a K1 chunk = this many decoder waves per CU (x 64 members); the launch itself keeps up to P1_WAVES_PER_CU resident
constexpr int P1_WAVES_PER_CU = 12;       // decoder waves a CU holds (11 KB LDS and <= 128 VGPRs each; 10 / 14 / 16 measured within 2 %)
constexpr int K1_SLOTS_DEFAULT = 3;  // token ring: chunk c uses slot c % slots (phase 1 of the next two chunks runs while phase 2 of c reads; a fourth slot measured the same: 865 / 877 vs 890 / 884 Mreads/s on 96 M reads); NGSQC_TOKEN_SLOTS
constexpr int N_DEPTH_SETS = 2;      // [0] the mapping scan's target region, [1] the extra depth scan of a job (-somatic_custom_bed)

// target regions + per-base depth of one scan
struct DepthSet
{
	std::vector<ngsqc_region> regions; std::vector<int64_t> doff; std::vector<int32_t> rlen; int64_t n_slots = 0; int64_t roi_bases = 0;
	DevBuf<int32_t> d_reg_start, d_reg_end, d_reg_len, d_tid_first, d_tid_last; DevBuf<int64_t> d_doff; DevBuf<int32_t> d_depth; DevBuf<uint8_t> d_tmp;
	bool depth_ready = false;
};

// what a consumer sees of the resident tile (offsets are tile-local; byte 0 is the first carried byte)
struct TileCtx { const uint8_t* infl; int64_t total; const int64_t* recoff; int64_t n_rec; int64_t ord_base; int tile; bool last; };
} // namespace

struct ngsqc_handle
{
	std::string err, path;
	bool from_cram = false;                        // the image is the BAM stream the host made of a CRAM 3.0 file (cram.hip)
	int device = 0; int n_cu = 256;
	hipStream_t stream = nullptr;                 // main stream: K2, consumers, setup copies
	hipStream_t s_p1[2] = {nullptr, nullptr};      // K1 phase 1 (alternating: the next chunk's waves fill in as the previous chunk's finish)
	hipStream_t s_p2 = nullptr;                    // K1 phase 2
	hipStream_t s_crc = nullptr;                   // CRC32 of the inflated members (behind phase 2 of the chunk, beside phase 2 of the next one)
	size_t csize = 0;
	std::vector<BlockDesc> blocks; int64_t total = 0;   // BGZF member table of the handle (a shard: rebased to its range)
	std::vector<uint32_t> crc;                           // CRC32 of every member's inflated bytes (from its BGZF trailer)
	std::vector<uint64_t> member_off;                    // file offset of every member of the table (a handle on the whole file only: what ngsqc_write_bai turns into virtual offsets)
	DevBuf<uint8_t> d_comp;
	std::vector<std::string> ref_names; std::vector<int64_t> ref_lens; int64_t first_rec = 0; std::string header_text;   // (SAM header text of the BAM header)
	// ---- layout of the tile stream (plan_layout) ----
	bool planned = false;
	int64_t chunk = 0, nch = 0;                        // K1 chunk size (members) and count
	std::vector<std::pair<int64_t, int64_t>> tiles;    // (first member, count); whole chunks
	std::vector<int64_t> tile_first_chunk;             // size nt + 1
	int64_t pfx = 0, max_tile_bytes = 0, slot_pages = 0;   // slot_pages: token pool pages of one chunk slot
	DevBuf<BlockDesc> d_kdesc;                         // per member: cpos into d_comp, upos relative to its tile's first member
	DevBuf<uint32_t> d_tok_first, d_tok_cnt, d_order, d_tok, d_crc, d_pool_ctr; DevBuf<unsigned long long> d_work; DevBuf<BlockStatus> d_status;   // d_tok: the token pool ring (k1_slots x slot_pages pages)
	int p1_wgs = 0;                                    // decoder workgroups a launch may keep resident
	DevBuf<uint32_t> d_sync_pool; DevBuf<BlockDesc> d_sync_desc; DevBuf<uint32_t> d_sync_u32; DevBuf<BlockStatus> d_sync_st; DevBuf<unsigned long long> d_sync_work;   // scratch of inflate_sync (kept: a hipFree waits for every queued kernel)
	static constexpr int MAX_TILE_BUFS = 4;
	int k1_slots = K1_SLOTS_DEFAULT;
	DevBuf<uint8_t> buf[MAX_TILE_BUFS]; int nbuf = 2;
	int64_t max_tile_members = 0;   // tile buffers (tile t lives in buf[t % nbuf]): [pfx carried bytes right-aligned][members][64]
	std::vector<hipEvent_t> ev_chunk;                  // 4 per chunk: p1 start/end, p2 start/end
	std::vector<hipEvent_t> ev_tile;                   // 2 per tile: K1 done (status on the host), consumed
	PinBuf<BlockStatus> p_status; PinBuf<int32_t> p_start; PinBuf<int64_t> p_next; PinBuf<unsigned long long> p_small;
	// ---- the resident tile ----
	bool decoded = false; int cur_tile = -1;
	int64_t n_rec = 0; DevBuf<int64_t> d_recoff;
	DevBuf<int64_t> d_long;                            // long-record list of the consumer that runs (scan, pileup: one after the other on the main stream); kept across tiles and jobs
	int64_t tile_prefix = 0, tile_total = 0, tile_u_lo = 0, tile_ord_base = 0;
	int64_t carry_len = 0, carry_src = 0, next_ord_base = 0, expected_abs = 0;
	int64_t k1_enq = 0;                                // chunks enqueued by the running job
	// K2 scratch (kept across tiles)
	DevBuf<int32_t> d_start; DevBuf<uint32_t> d_cnt; DevBuf<int64_t> d_next, d_base; DevBuf<uint32_t> d_bad; DevBuf<uint8_t> d_scan_tmp; DevBuf<uint16_t> d_rel;
	// depth state
	DepthSet ds[N_DEPTH_SETS]; int cur_ds = 0;
	ngsqc_timings tm{};
	// one BAM sharded over several handles: this handle owns the records that START inside members
	// [0, shard_own_members) of its (rebased) member table; the members behind them are only there to complete the last record
	int shard = 0, n_shards = 1;
	int64_t shard_own_members = -1;        // -1: not a shard (every record of the table is owned)
	int64_t shard_limit = -1;              // rebased inflated offset of the first byte that is NOT owned
	int64_t shard_u_base = 0;              // inflated offset (whole file) of the handle's first member
	int64_t shard_first_abs = -1, shard_exit_abs = -1; int shard_last_tile = -1;
	bool verify_crc = true;
	// H2D of the compressed image in the background (ngsqc_open of a path): host threads copy pieces in file order, every piece has an event that
	// the K1 chunk stream waits for; the mapping of the file lives until the last piece is on the device
	struct Upload
	{
		std::vector<std::thread> th; std::mutex mu; std::condition_variable cv;
		std::vector<hipEvent_t> ev; std::vector<char> recorded; size_t piece = 0, n_pieces = 0, bytes = 0; std::atomic<size_t> next{0}; std::atomic<bool> cancel{false};
		std::string err; void* map = nullptr; size_t map_n = 0; int fd = -1; double t0 = 0, t_done = 0; size_t done = 0;
		size_t waited[4] = {0, 0, 0, 0};   // pieces [0, waited[k]) have been waited for by stream slot k (main, s_p1[0], s_p1[1], s_p2)
		// ---- streamed image (round 4): the compressed bytes are never resident as a whole. d_comp is a ring of chunk slots (K1 chunk c reads slot c % slots);
		// every job ("pass") copies the file once more from its mapping, piece by piece in chunk order; a slot is overwritten when phase 2 of the chunk that
		// used it is done (p2_enq: chunks whose phase 2 is enqueued - their ev_chunk events are valid to wait for) ----
		struct SPiece { size_t src, dst, bytes; int64_t chunk; };
		std::vector<SPiece> sp; std::vector<size_t> chunk_first;   // pieces of the pass; first piece of every chunk (size nch + 1)
		std::atomic<int64_t> p2_enq{0}; bool pass_running = false, pass_fresh = false; const uint8_t* src_base = nullptr;   // pass_fresh: started ahead of its job (by the layout thread), nothing consumed yet
	};
	Upload* up = nullptr;
	bool stream_img = false; int comp_slots = 0; size_t comp_slot_bytes = 0;   // streamed image: ring geometry (plan_layout)
	std::vector<uint64_t> chunk_lo;                                           // file offset of the first byte copied for chunk c
	DevBuf<uint8_t> d_sync_comp;                                              // compressed bytes of the members inflate_sync works on (streamed image only)
	std::thread plan_thread; std::string plan_err;   // plan_layout in the background of ngsqc_open (device buffers of the tile stream: allocation overlaps the H2D)
	// the scan that rides K2's chain walk (launch_walk_scan): set by the job for its first scan consumer; fuse_ok turns false when a tile is not laid out like an
	// htslib file (the general K2 path takes over); fused_tile = the tile whose records that scan has already seen
	EvLog ev_store; EvLog* evlog = &ev_store;   // stage times of the running tile stream (resolved at its end)
	// what the host learns about a tile in ONE wait (round 5; p_rb, pinned): [0 .. A_HIST0) the device accumulators of the riding scan after its walk (deferred-record
	// count, the tile's longest / first paired record, totals), [RB_CAND] the site pileup's candidates
	PinBuf<unsigned long long> p_rb; static constexpr int RB_CAND = 64, RB_BQ = 65, RB_TOTAL = 72;
	// record offsets of the resident tile are expanded on demand (ensure_recoff): a job whose consumers all ride the chain walk never reads them
	bool lazy_recoff = false; int recoff_tile = -1;
	struct RecoffArgs { const uint8_t* base = nullptr; int64_t total = 0; const BlockDesc* desc = nullptr; int64_t ne = 0, prefix = 0, n_rec = 0, nm = 0; int ksh = 0; int tile = -1; } rw;
	struct FusedScan   // what K2 needs of such a scan (ScanState)
	{
		virtual void fused_launch(ngsqc_handle* h, const uint8_t* infl, int64_t total, int sgn, const BlockDesc* d_desc, int64_t ne, int64_t prefix, int ksh, int64_t nm, int64_t scan_limit) = 0;
		virtual void fused_readback(ngsqc_handle* h) = 0;     // enqueues the copy of its accumulators (and of what rides with it) into h->p_rb
		virtual unsigned long long fused_bq_cap() = 0;        // entries the list of min_baseq records holds (p_rb[RB_BQ] must not exceed it)
		virtual ~FusedScan() = default;
	};
	FusedScan* fuse = nullptr; bool fuse_ok = true; int fused_tile = -1;
	bool long_reads = false;   // the file's first record is longer than 8 KiB (index_tile): entries are groups of members, nothing is assumed about member starts
	bool k2_plain = false;   // a tile of the running stream did not pass the chain check on the device: the later tiles walk whole members, as the general path needs them
	std::vector<int64_t> rq_len_hist, rq_cyc;   // results of the last raw-read QC pass
	struct Partial;                        // state between ngsqc_scan_mapping_partial and ngsqc_scan_mapping_finish
	Partial* partial = nullptr;
};

namespace {

// NGSQC_DEBUG: where the wall time of an open goes (ms since the first stamp of the process)
void dbg_stamp(const char* what)
{
	static const bool on = getenv("NGSQC_DEBUG") != nullptr; static const double t0 = wall_ms();
	if (on) fprintf(stderr, "[ngsqc] t+%.1f ms %s\n", wall_ms() - t0, what);
}

// ---- BGZF member table (host): SAM spec §4.1 ----
// members of [off, off_end) (off_end: a member start or the end of the file), at most max_members of them; upos continues at `upos`
void walk_bgzf(const uint8_t* file, size_t n, size_t& off, size_t off_end, int64_t max_members, uint64_t& upos, std::vector<BlockDesc>& blocks, std::vector<uint32_t>& crc, std::vector<uint64_t>* file_off = nullptr)
{
	int64_t k = 0;
	while (off < n && off < off_end && k < max_members)
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
		if (isize) { blocks.push_back(BlockDesc{(uint64_t)(off + xend), upos, (uint32_t)(bsize - xend - 8), isize}); crc.push_back(rd32(p + bsize - 8)); if (file_off) file_off->push_back((uint64_t)off); }
		upos += isize; off += bsize; ++k;
	}
}
// Does a BGZF member start at off? (header checks of walk_bgzf, without exceptions) -> its size, 0 = no
uint32_t bgzf_member_at(const uint8_t* file, size_t n, size_t off)
{
	if (off + 18 > n) return 0;
	const uint8_t* p = file + off;
	if (p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return 0;
	const uint32_t xlen = rd16(p + 10); uint32_t bsize = 0; bool found = false;
	size_t x = 12; const size_t xend = 12 + (size_t)xlen;
	if (off + xend > n) return 0;
	while (x + 4 <= xend) { const uint16_t slen = rd16(p + x + 2); if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2) { bsize = rd16(p + x + 4) + 1u; found = true; } x += 4 + slen; }
	if (!found || bsize < xend + 8 || off + bsize > n || rd32(p + bsize - 4) > 65536) return 0;
	return bsize;
}

// The member table with several host threads (NGSQC_WALK_THREADS; the walk touches one page of the mapping per member and is bound by page faults: 1.2 s
// for the 3.2 M members of a 60 GB file with one thread - as long as the H2D copy that runs beside it). Thread k starts at the first offset behind
// k * n / T that begins a chain of three plausible members; the pieces are only accepted when every thread's walk ENDS exactly where the next one started -
// then the concatenation is, by induction from offset 0, the sequential walk. Anything else (no start found, an error anywhere) falls back to that walk,
// which also reports errors at the place the reference would.
bool scan_bgzf_threads(const uint8_t* file, size_t n, int T, std::vector<BlockDesc>& blocks, std::vector<uint32_t>& crc, int64_t& total, std::vector<uint64_t>* file_off)
{
	std::vector<size_t> start((size_t)T + 1, 0); start[(size_t)T] = n;
	for (int k = 1; k < T; ++k)
	{
		size_t o = (size_t)((double)n * (double)k / (double)T); const size_t lim = std::min(n, o + (1u << 18)); bool ok = false;
		o = std::max(o, start[(size_t)k - 1]);
		while (o < lim)
		{
			const void* q = memchr(file + o, 31, lim - o);
			if (!q) break;
			o = (size_t)((const uint8_t*)q - file);
			size_t c = o; int good = 0;
			for (; good < 3; ++good) { const uint32_t bs = bgzf_member_at(file, n, c); if (!bs) break; c += bs; if (c == n) { good = 3; break; } }
			if (good >= 3) { ok = true; break; }
			++o;
		}
		if (!ok) return false;
		start[(size_t)k] = o;
	}
	struct Piece { std::vector<BlockDesc> b; std::vector<uint32_t> c; std::vector<uint64_t> f; uint64_t u = 0; bool ok = false; };
	std::vector<Piece> pc((size_t)T);
	std::vector<std::thread> th;
	for (int k = 0; k < T; ++k)
		th.emplace_back([&, k] {
			Piece& P = pc[(size_t)k];
			try
			{
				size_t off = start[(size_t)k]; uint64_t u = 0;
				walk_bgzf(file, n, off, start[(size_t)k + 1], INT64_MAX, u, P.b, P.c, file_off ? &P.f : nullptr);
				P.u = u; P.ok = off == start[(size_t)k + 1];   // the walk ended exactly at the next piece's start
			}
			catch (...) { P.ok = false; }
		});
	for (auto& t : th) t.join();
	for (const Piece& P : pc) if (!P.ok) return false;
	uint64_t u = 0; size_t m = 0;
	for (const Piece& P : pc) m += P.b.size();
	blocks.reserve(m); crc.reserve(m); if (file_off) file_off->reserve(m);
	for (Piece& P : pc)
	{
		for (BlockDesc& d : P.b) { d.upos += u; blocks.push_back(d); }
		crc.insert(crc.end(), P.c.begin(), P.c.end());
		if (file_off) file_off->insert(file_off->end(), P.f.begin(), P.f.end());
		u += P.u;
	}
	total = (int64_t)u;
	return true;
}

void scan_bgzf(const uint8_t* file, size_t n, std::vector<BlockDesc>& blocks, std::vector<uint32_t>& crc, int64_t& total, std::vector<uint64_t>* file_off = nullptr, int threads = 0, bool* in_pieces = nullptr)
{
	if (in_pieces) *in_pieces = false;
	if (n >= 4 && memcmp(file, "CRAM", 4) == 0) throw std::domain_error("a CRAM file has no BGZF members: ngsqc_open / ngsqc_open_memory decode it (cram.hip)");
	if (threads <= 0) { threads = 8; if (const char* e = getenv("NGSQC_WALK_THREADS")) threads = std::min(64, std::max(1, atoi(e))); }   // (round 4: on by default - the first job races the copy, tests/test_gpu_tools.py)
	if (threads > 1 && n >= ((size_t)threads << 20) && scan_bgzf_threads(file, n, threads, blocks, crc, total, file_off)) { if (in_pieces) *in_pieces = true; return; }
	blocks.clear(); crc.clear(); if (file_off) file_off->clear();
	size_t off = 0; uint64_t upos = 0;
	walk_bgzf(file, n, off, n, INT64_MAX, upos, blocks, crc, file_off);
	total = (int64_t)upos;
}

void init_device(ngsqc_handle* h, int device)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw std::runtime_error("no HIP device available (libngsqc_hip has no CPU fallback)");
	if (device < 0 || device >= n) throw ArgError("invalid HIP device ordinal");
	h->device = device;
	HIPCHK(hipSetDevice(device));
	// K2 and the consumers of a tile run while K1 of the next tile fills the chip: their stream gets the highest priority so that their
	// workgroups take the slots that K1's workgroups free instead of queueing behind K1's remaining grid
	int prio_lo = 0, prio_hi = 0; (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
	HIPCHK(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_hi));
	HIPCHK(hipStreamCreateWithFlags(&h->s_p1[0], hipStreamNonBlocking));
	HIPCHK(hipStreamCreateWithFlags(&h->s_p1[1], hipStreamNonBlocking));
	HIPCHK(hipStreamCreateWithFlags(&h->s_p2, hipStreamNonBlocking));
	HIPCHK(hipStreamCreateWithFlags(&h->s_crc, hipStreamNonBlocking));
	int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) h->n_cu = cu;
	const int pw = P1_WAVES_PER_CU;
	h->p1_wgs = h->n_cu * pw;
	k1_read_switches();
	if (const char* e = getenv("NGSQC_VERIFY_CRC")) h->verify_crc = atoi(e) != 0;
}

std::string inflate_error(const ngsqc_handle* h, int64_t member, uint32_t code)
{
	// what the reference reports when htslib fails on a block (BamReader.h:389-392)
	if (code == K1_ERR_CRC) return "Could not read next alignment in BAM/CRAM file " + h->path + " (BGZF CRC32 mismatch in block " + std::to_string(member) + ")";
	return "Could not read next alignment in BAM/CRAM file " + h->path + " (BGZF inflate failed in block " + std::to_string(member) + ", code " + std::to_string(code) + ")";
}

void upload_wait(ngsqc_handle* h, size_t end_byte, hipStream_t st, int slot);   // (H2D in the background, below)

// Synchronous K1 of a few members on the main stream with private scratch (header read, second chance of members that found the token
// pool of their launch used up). idx: member indices into h->blocks; desc/out: where each one goes. The pool is sized for the worst
// case (two token slots per output byte), so the call is made in batches of bounded scratch; the scratch is kept across calls.
void inflate_sync(ngsqc_handle* h, const std::vector<int64_t>& idx, const std::vector<BlockDesc>& desc, uint8_t* d_out, int level = 0)
{
	// level 0: four token words per output byte (a group holds at least one real word; enough unless a member holds hundreds of DEFLATE blocks: every block has its
	// literal table in the pool), 1024 members per batch = 1.1 GB of pool; level 1: the bound that holds for every valid member (k1_pool_pages_absolute: up to 18 MB per member), 32 per batch
	const int64_t BATCH = level == 0 ? 1024 : 32;
	std::vector<int64_t> idx2; std::vector<BlockDesc> desc2;   // members that need level 1
	(level == 0 ? h->tm.members_second_chance : h->tm.members_third_chance) += (int64_t)idx.size();
	for (int64_t b0 = 0; b0 < (int64_t)idx.size(); b0 += BATCH)
	{
		const int64_t n = std::min<int64_t>(BATCH, (int64_t)idx.size() - b0);
		std::vector<BlockDesc> dd(desc.begin() + b0, desc.begin() + b0 + n); std::vector<uint32_t> crc((size_t)n);
		uint64_t sc = 0, su = 0;
		for (int64_t i = 0; i < n; ++i) { crc[(size_t)i] = h->crc[(size_t)idx[(size_t)(b0 + i)]]; sc += dd[(size_t)i].clen; su += dd[(size_t)i].usize; }
		const uint64_t pages = level == 0 ? k1_pool_pages(sc, su, (uint64_t)n, true) : k1_pool_pages_absolute(sc, su, (uint64_t)n);
		const uint8_t* d_comp = h->d_comp.p;
		if (h->stream_img)
		{
			// the image is not resident: these members' payloads are copied from the mapping into a private buffer (16-byte aligned, 64 bytes of slack each)
			size_t tot = 0; for (BlockDesc& d : dd) { const size_t a = (size_t)(d.cpos & 15u); tot += (a + d.clen + 64 + 15) & ~(size_t)15; }
			std::vector<uint8_t> hc(tot + 1024, 0); size_t o = 0;
			for (BlockDesc& d : dd)
			{
				const size_t a = (size_t)(d.cpos & 15u), src = (size_t)d.cpos - a, len = std::min<size_t>(a + d.clen + 64, h->up->map_n - src);
				memcpy(hc.data() + o, h->up->src_base + src, len);
				d.cpos = o + a; o += (a + d.clen + 64 + 15) & ~(size_t)15;
			}
			h->d_sync_comp.ensure_slack(hc.size());
			HIPCHK(hipMemcpyAsync(h->d_sync_comp.p, hc.data(), hc.size(), hipMemcpyHostToDevice, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
			d_comp = h->d_sync_comp.p;
		}
		else { uint64_t cend = 0; for (const BlockDesc& d : dd) cend = std::max<uint64_t>(cend, d.cpos + d.clen + 64); upload_wait(h, (size_t)cend, h->stream, 0); }
		h->d_sync_desc.ensure_slack((size_t)n); h->d_sync_st.ensure_slack((size_t)n); h->d_sync_work.ensure(2);
		h->d_sync_u32.ensure_slack((size_t)(3 * n + 16));   // [first | count | crc]
		h->d_sync_pool.ensure_slack((size_t)pages * K1_PAGE_WORDS + 16);
		uint32_t* d_first = h->d_sync_u32.p, *d_cnt = d_first + n, *d_crc = d_cnt + n;
		HIPCHK(hipMemcpyAsync(h->d_sync_desc.p, dd.data(), (size_t)n * sizeof(BlockDesc), hipMemcpyHostToDevice, h->stream));
		HIPCHK(hipMemcpyAsync(d_crc, crc.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
		HIPCHK(hipMemsetAsync(h->d_sync_work.p, 0, 2 * sizeof(unsigned long long), h->stream));   // [queue head | pool counter]
		launch_huff_tokens(d_comp, h->d_sync_desc.p, n, h->d_sync_st.p, h->d_sync_pool.p, (uint32_t)pages, (uint32_t*)(h->d_sync_work.p + 1), d_first, d_cnt, h->d_sync_work.p, nullptr, h->p1_wgs, h->stream);
		launch_lz77_resolve(h->d_sync_desc.p, n, d_out, h->d_sync_st.p, h->d_sync_pool.p, d_first, d_cnt, d_comp, h->stream);
		if (h->verify_crc) launch_crc32(h->d_sync_desc.p, n, d_out, d_crc, h->d_sync_st.p, h->stream);
		std::vector<BlockStatus> st((size_t)n);
		HIPCHK(hipMemcpyAsync(st.data(), h->d_sync_st.p, (size_t)n * sizeof(BlockStatus), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		for (int64_t i = 0; i < n; ++i)
			if (st[(size_t)i].error == K1_ERR_TOKEN_OVERFLOW && level == 0) { idx2.push_back(idx[(size_t)(b0 + i)]); desc2.push_back(desc[(size_t)(b0 + i)]); }
			else if (st[(size_t)i].error) throw FormatError(inflate_error(h, idx[(size_t)(b0 + i)], st[(size_t)i].error));
	}
	if (!idx2.empty()) inflate_sync(h, idx2, desc2, d_out, 1);
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
		std::vector<int64_t> idx((size_t)k); std::vector<BlockDesc> desc((size_t)k);
		for (int64_t i = 0; i < k; ++i) { idx[(size_t)i] = i; desc[(size_t)i] = h->blocks[(size_t)i]; }
		inflate_sync(h, idx, desc, tmp.p);
		std::vector<uint8_t> hb((size_t)bytes);
		if (bytes) HIPCHK(hipMemcpy(hb.data(), tmp.p, (size_t)bytes, hipMemcpyDeviceToHost));
		bool complete = false;
		do
		{
			if (bytes < 12) break;
			if (memcmp(hb.data(), "BAM\1", 4) != 0) throw FormatError("Could not read header from BAM/CRAM file " + h->path);
			size_t o = 4; uint32_t l_text = rd32(&hb[o]); o += 4 + (size_t)l_text;
			if (o + 4 > (size_t)bytes) break;
			h->header_text.assign((const char*)&hb[8], (size_t)l_text);
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

// H2D of the compressed image. The source is pageable memory (an mmap of the file, a caller's buffer). One hipMemcpy of it is the default: the
// runtime pins the pages and runs the DMA at 37-56 GB/s on a 14 GB image whose pages are warm (12 GB/s on the 60 GB image right after it was
// generated: first pinning of cold pages). NGSQC_H2D_THREADS=T stages the image through T host threads with pinned buffer pairs instead; measured
// slower on this host (16-CPU quota: 14 / 20 / 26 GB/s at 8 / 4 / 16 threads), kept as a switch for hosts with more cores per GPU.
void upload_compressed(ngsqc_handle* h, const uint8_t* bytes, size_t beg, size_t end)
{
	const size_t n = end - beg;
	h->d_comp.alloc(n + 1024);
	HIPCHK(hipMemsetAsync(h->d_comp.p + n, 0, 1024, h->stream));
	if (!n) return;
	constexpr size_t PIECE = 32u << 20;
	int T = 1; if (const char* e = getenv("NGSQC_H2D_THREADS")) T = std::max(1, atoi(e));
	if (n < 8 * PIECE || T == 1) { HIPCHK(hipMemcpyAsync(h->d_comp.p, bytes + beg, n, hipMemcpyHostToDevice, h->stream)); HIPCHK(hipStreamSynchronize(h->stream)); return; }
	const size_t n_pieces = (n + PIECE - 1) / PIECE;
	std::atomic<size_t> next(0); std::vector<std::string> errs((size_t)T);
	std::vector<std::thread> th;
	for (int t = 0; t < T; ++t)
		th.emplace_back([&, t] {
			uint8_t* pin[2] = {nullptr, nullptr}; hipStream_t st = nullptr; hipEvent_t ev[2] = {nullptr, nullptr};
			try
			{
				HIPCHK(hipSetDevice(h->device));
				HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
				for (int k = 0; k < 2; ++k) { HIPCHK(hipHostMalloc((void**)&pin[k], PIECE, hipHostMallocDefault)); HIPCHK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); }
				for (int k = 0;; k ^= 1)
				{
					const size_t i = next.fetch_add(1); if (i >= n_pieces) break;
					const size_t off = i * PIECE, sz = std::min(PIECE, n - off);
					HIPCHK(hipEventSynchronize(ev[k]));   // the previous DMA out of this buffer is done (an unrecorded event is complete)
					memcpy(pin[k], bytes + beg + off, sz);
					HIPCHK(hipMemcpyAsync(h->d_comp.p + off, pin[k], sz, hipMemcpyHostToDevice, st));
					HIPCHK(hipEventRecord(ev[k], st));
				}
				HIPCHK(hipStreamSynchronize(st));
			}
			catch (std::exception& e) { errs[(size_t)t] = e.what(); }
			for (int k = 0; k < 2; ++k) { if (pin[k]) (void)hipHostFree(pin[k]); if (ev[k]) (void)hipEventDestroy(ev[k]); }
			if (st) (void)hipStreamDestroy(st);
		});
	for (auto& t : th) t.join();
	for (auto& e : errs) if (!e.empty()) throw std::runtime_error(e);
	HIPCHK(hipStreamSynchronize(h->stream));
}

// ---- H2D in the background: pieces of the compressed image in file order, one event per piece ----
// T host threads (NGSQC_H2D_THREADS, default 4) each copy whole pieces with hipMemcpyAsync on their own stream; the source is the mapping of the file
// (pageable: the runtime stages it, a call returns when its piece is staged), so T pieces are in flight and the first K1 chunk starts as soon as its
// pieces have arrived instead of behind the whole image.
void upload_join(ngsqc_handle* h)
{
	ngsqc_handle::Upload* u = h->up;
	if (!u) return;
	u->cancel = true; u->cv.notify_all();
	for (auto& t : u->th) if (t.joinable()) t.join();
	u->th.clear();
	for (hipEvent_t e : u->ev) if (e) (void)hipEventDestroy(e);
	u->ev.clear();
	if (u->map) { void* m = u->map; const size_t n = u->map_n; const int fd = u->fd; reaper().task([m, n, fd] { munmap(m, n); if (fd >= 0) ::close(fd); }); u->map = nullptr; u->fd = -1; }
	if (u->fd >= 0) { ::close(u->fd); u->fd = -1; }
}
void upload_start(ngsqc_handle* h, const uint8_t* bytes, size_t beg, size_t end)
{
	ngsqc_handle::Upload* u = h->up;
	const size_t n = end - beg;
	dbg_stamp("upload: allocating the image buffer");
	h->d_comp.alloc(n + 1024);
	dbg_stamp("upload: image buffer allocated");
	HIPCHK(hipMemsetAsync(h->d_comp.p + n, 0, 1024, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	dbg_stamp("upload: first device operation done");
	u->piece = 64u << 20; if (const char* e = getenv("NGSQC_H2D_PIECE_MB")) u->piece = (size_t)std::max(1, atoi(e)) << 20;
	u->bytes = n; u->n_pieces = (n + u->piece - 1) / u->piece; u->next = 0; u->done = 0; u->cancel = false; u->t0 = wall_ms(); u->t_done = u->t0;
	u->recorded.assign(u->n_pieces, 0); u->ev.assign(u->n_pieces, nullptr);
	for (auto& e : u->ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	int T = 4; if (const char* e = getenv("NGSQC_H2D_THREADS")) T = std::max(1, atoi(e));
	T = (int)std::min<size_t>((size_t)T, std::max<size_t>(u->n_pieces, 1));
	uint8_t* const dst = h->d_comp.p; const uint8_t* const src = bytes + beg; const int device = h->device;
	int delay_us = 0; if (const char* e = getenv("NGSQC_H2D_DELAY_US")) delay_us = std::max(0, atoi(e));   // (tests: a slow link, so that the chunk stream really waits for pieces)
	for (int t = 0; t < T && u->n_pieces; ++t)
		u->th.emplace_back([u, dst, src, device, delay_us] {
			hipStream_t st = nullptr;
			try
			{
				HIPCHK(hipSetDevice(device));
				HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
				for (;;)
				{
					const size_t i = u->next.fetch_add(1);
					if (i >= u->n_pieces || u->cancel) break;
					const size_t off = i * u->piece, sz = std::min(u->piece, u->bytes - off);
					if (delay_us) std::this_thread::sleep_for(std::chrono::microseconds(delay_us));
					HIPCHK(hipMemcpyAsync(dst + off, src + off, sz, hipMemcpyHostToDevice, st));
					HIPCHK(hipEventRecord(u->ev[i], st));
					{ std::lock_guard<std::mutex> g(u->mu); u->recorded[i] = 1; }
					u->cv.notify_all();
				}
				HIPCHK(hipStreamSynchronize(st));
			}
			catch (std::exception& e) { std::lock_guard<std::mutex> g(u->mu); if (u->err.empty()) u->err = e.what(); u->cv.notify_all(); }
			if (st) (void)hipStreamDestroy(st);
			{ std::lock_guard<std::mutex> g(u->mu); if (++u->done == u->th.size()) { u->t_done = wall_ms(); dbg_stamp("upload: last piece on the device"); } }
			u->cv.notify_all();
		});
}
// stream st (slot: 0 main, 1 / 2 the phase-1 streams) may read the compressed bytes [0, end_byte) behind this call
void upload_wait(ngsqc_handle* h, size_t end_byte, hipStream_t st, int slot)
{
	ngsqc_handle::Upload* u = h->up;
	if (!u || !u->n_pieces) return;
	const size_t p1 = std::min(u->n_pieces, (std::min(end_byte, u->bytes) + u->piece - 1) / u->piece);
	for (size_t p = u->waited[slot]; p < p1; ++p)
	{
		{
			std::unique_lock<std::mutex> lk(u->mu);
			u->cv.wait(lk, [&] { return u->recorded[p] || !u->err.empty(); });
			if (!u->err.empty()) throw std::runtime_error("H2D of the compressed image failed: " + u->err);
		}
		HIPCHK(hipStreamWaitEvent(st, u->ev[p], 0));
	}
	if (p1 > u->waited[slot]) u->waited[slot] = p1;
}
// ---- streamed image: one pass of the file through the ring of chunk slots (started by every tile stream) ----
void stream_pass_end(ngsqc_handle* h)
{
	ngsqc_handle::Upload* u = h->up;
	if (!u || !u->pass_running) return;
	u->cancel = true; u->cv.notify_all();
	for (auto& t : u->th) if (t.joinable()) t.join();
	u->th.clear(); u->pass_running = false; u->cancel = false;
}
void stream_pass_begin(ngsqc_handle* h)
{
	ngsqc_handle::Upload* u = h->up;
	stream_pass_end(h);
	if (u->sp.empty()) return;
	while (u->ev.size() < u->sp.size()) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); u->ev.push_back(e); }
	u->recorded.assign(u->sp.size(), 0); u->next = 0; u->done = 0; u->p2_enq = 0; u->err.clear(); u->t0 = wall_ms(); u->t_done = u->t0;
	for (size_t& w : u->waited) w = 0;
	int T = 4; if (const char* e = getenv("NGSQC_H2D_THREADS")) T = std::max(1, atoi(e));
	T = (int)std::min<size_t>((size_t)T, u->sp.size());
	int delay_us = 0; if (const char* e = getenv("NGSQC_H2D_DELAY_US")) delay_us = std::max(0, atoi(e));
	uint8_t* const dst = h->d_comp.p; const int device = h->device; const int slots = h->comp_slots; hipEvent_t* const ev_chunk = h->ev_chunk.data();
	// The source of a piece is the mapping of the file (hipMemcpyAsync stages a pageable source through the runtime's pinned buffers). Reading through the mapping
	// faults in one page-table entry per 4 KB - 15 M of them for a 60 GB file - and tearing them down again costs 0.3 - 0.75 s at close for a 19 GB file. Measured
	// alternatives on a 19 GB BAM (profiles/r04_tool_probe.txt; code removed in round 5): pieces read with pread into pinned buffers of the copier threads, 12.5 GB/s with four threads,
	// 24 GB/s with eight, against 37 GB/s through the mapping; dropping a sent piece's entries with madvise(MADV_DONTNEED) made the job ten times slower (the
	// address-space lock against the other copiers' faults). The mapping stays.
	// (Round 5, profiles/r05_tool_probe.txt: registering each piece of the mapping with hipHostRegister(read only) just before it is sent - so that the DMA engines
	// read the page cache's pages themselves - made the job of a 9.4 GB BAM 0.32 -> 1.40 s, eight copier threads instead of four 1.06 s: neither is kept.)
	u->pass_running = true;
	for (int t = 0; t < T; ++t)
		u->th.emplace_back([u, dst, device, delay_us, slots, ev_chunk] {
			hipStream_t st = nullptr; long last = -1;
			auto drop_last = [&]() { last = -1; };
			try
			{
				HIPCHK(hipSetDevice(device));
				HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
				for (;;)
				{
					const size_t i = u->next.fetch_add(1);
					if (i >= u->sp.size() || u->cancel) break;
					const ngsqc_handle::Upload::SPiece& P = u->sp[i];
					drop_last();
					if (P.chunk >= slots)
					{
						// the slot still holds chunk P.chunk - slots: wait until its phase 2 (the last reader of the compressed bytes) has been enqueued, then until it is done
						const int64_t prev = P.chunk - slots;
						{ std::unique_lock<std::mutex> lk(u->mu); u->cv.wait(lk, [&] { return u->p2_enq.load() > prev || u->cancel.load(); }); }
						if (u->cancel) break;
						HIPCHK(hipEventSynchronize(ev_chunk[4 * prev + 3]));
					}
					if (delay_us) std::this_thread::sleep_for(std::chrono::microseconds(delay_us));
					HIPCHK(hipMemcpyAsync(dst + P.dst, u->src_base + P.src, P.bytes, hipMemcpyHostToDevice, st));
					HIPCHK(hipEventRecord(u->ev[i], st));
					{ std::lock_guard<std::mutex> g(u->mu); u->recorded[i] = 1; }
					u->cv.notify_all();
					last = (long)i;
				}
				HIPCHK(hipStreamSynchronize(st));
				drop_last();
			}
			catch (std::exception& e) { std::lock_guard<std::mutex> g(u->mu); if (u->err.empty()) u->err = e.what(); }
			if (st) (void)hipStreamDestroy(st);
			{ std::lock_guard<std::mutex> g(u->mu); if (++u->done == u->th.size()) u->t_done = wall_ms(); }
			u->cv.notify_all();
		});
}
// stream st may read chunk c's compressed bytes behind this call (the host waits until the copies are issued, the stream for their events)
void stream_wait_chunk(ngsqc_handle* h, int64_t c, hipStream_t st)
{
	ngsqc_handle::Upload* u = h->up;
	for (size_t p = u->chunk_first[(size_t)c]; p < u->chunk_first[(size_t)c + 1]; ++p)
	{
		{
			std::unique_lock<std::mutex> lk(u->mu);
			u->cv.wait(lk, [&] { return u->recorded[p] || !u->err.empty(); });
			if (!u->err.empty()) throw std::runtime_error("H2D of the compressed image failed: " + u->err);
		}
		HIPCHK(hipStreamWaitEvent(st, u->ev[p], 0));
	}
}
void stream_p2_enqueued(ngsqc_handle* h, int64_t c)
{
	ngsqc_handle::Upload* u = h->up;
	{ std::lock_guard<std::mutex> g(u->mu); u->p2_enq = c + 1; }
	u->cv.notify_all();
}

// the whole image is on the device (ngsqc_upload_wait / timings)
void upload_finish(ngsqc_handle* h)
{
	ngsqc_handle::Upload* u = h->up;
	if (!u) return;
	if (h->stream_img && !u->pass_running) return;   // (no pass under way: nothing in flight)
	{ std::unique_lock<std::mutex> lk(u->mu); u->cv.wait(lk, [&] { return u->done == u->th.size() || !u->err.empty(); }); if (!u->err.empty()) throw std::runtime_error("H2D of the compressed image failed: " + u->err); }
	h->tm.h2d_ms = u->t_done - u->t0;
}

constexpr int64_t SHARD_TAIL_MEMBERS = 64;   // members behind a shard that are inflated to complete its last record (NGSQC_SHARD_TAIL_MEMBERS)

void open_common(ngsqc_handle* h, const uint8_t* bytes, size_t n, int device, int shard, int n_shards)
{
	if (n_shards < 1 || shard < 0 || shard >= n_shards) throw ArgError("invalid shard index");
	h->csize = n;
	if (h->up)
	{
		// a path: the copy starts before anything else looks at the file (the BGZF member walk below runs beside it; the header read waits for the first pieces only)
		if (n >= 4 && memcmp(bytes, "CRAM", 4) == 0) throw std::domain_error("a CRAM file has no BGZF members: ngsqc_open / ngsqc_open_memory decode it (cram.hip)");
		dbg_stamp("open: start");
		init_device(h, device);
		dbg_stamp("open: device and streams ready");
		// A large file is STREAMED (round 4): no 60 GB image buffer (its allocation alone took as long as the copy, and a BAM no longer has to fit HBM next to its
		// tiles) - every job copies the file through a ring of K1-chunk slots. NGSQC_STREAM_IMAGE=1 / 0 forces / forbids it, NGSQC_STREAM_IMAGE_MIN_MB moves the
		// threshold (default 4096: smaller files stay resident, so repeated jobs on them do not cross PCIe again).
		{
			const char* es = getenv("NGSQC_STREAM_IMAGE"); size_t min_mb = 4096; if (const char* em = getenv("NGSQC_STREAM_IMAGE_MIN_MB")) min_mb = (size_t)std::max(0, atoi(em));
			h->stream_img = es ? atoi(es) != 0 : (n >> 20) >= min_mb;
		}
		h->up->src_base = bytes; h->up->map_n = n;
		if (!h->stream_img) upload_start(h, bytes, 0, n);
		dbg_stamp("open: upload threads started");
		scan_bgzf(bytes, n, h->blocks, h->crc, h->total, n_shards == 1 ? &h->member_off : nullptr);
		dbg_stamp("open: BGZF member table walked");
	}
	else { scan_bgzf(bytes, n, h->blocks, h->crc, h->total, n_shards == 1 ? &h->member_off : nullptr); init_device(h, device); }
	Timer t(h->stream); t.start();
	h->shard = shard; h->n_shards = n_shards;
	if (n_shards == 1)
	{
		if (!h->up) { upload_compressed(h, bytes, 0, n); h->tm.h2d_ms = t.stop(); }
		h->tm.compressed_bytes = (int64_t)n; h->tm.inflated_bytes = h->total;
		read_header(h, (int64_t)h->blocks.size());
		dbg_stamp("open: BAM header read");
		return;
	}
	// ---- header: only the first members are sent to the device ----
	const int64_t nb = (int64_t)h->blocks.size();
	for (int64_t k = std::min<int64_t>(8, nb);; k = std::min<int64_t>(k * 4, nb))
	{
		const size_t end = k ? (size_t)(h->blocks[(size_t)k - 1].cpos + h->blocks[(size_t)k - 1].clen) : 0;
		upload_compressed(h, bytes, 0, end);
		if (read_header(h, k)) break;
		if (k >= nb) throw FormatError("Could not read header from BAM/CRAM file " + h->path);
	}
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
	std::vector<BlockDesc> own; std::vector<uint32_t> own_crc;
	size_t cbeg = 0, cend = 0; int64_t u0 = 0, u_own = 0, u_all = 0;
	if (m1 > m0)
	{
		cbeg = (size_t)(h->blocks[(size_t)m0].cpos & ~15ull);
		cend = (size_t)(h->blocks[(size_t)m_end - 1].cpos + h->blocks[(size_t)m_end - 1].clen);
		u0 = (int64_t)h->blocks[(size_t)m0].upos;
		u_own = (m1 < nb ? (int64_t)h->blocks[(size_t)m1].upos : h->total) - u0;
		u_all = (m_end < nb ? (int64_t)h->blocks[(size_t)m_end].upos : h->total) - u0;
		for (int64_t i = m0; i < m_end; ++i) { BlockDesc d = h->blocks[(size_t)i]; d.cpos -= cbeg; d.upos -= (uint64_t)u0; own.push_back(d); own_crc.push_back(h->crc[(size_t)i]); }
	}
	const int64_t first_rec_abs = h->first_rec;
	h->blocks.swap(own); h->crc.swap(own_crc);
	h->shard_own_members = m1 - m0; h->shard_limit = u_own; h->shard_u_base = u0; h->total = u_all;
	h->first_rec = first_rec_abs >= u0 ? first_rec_abs - u0 : -1;   // shards behind the header: unknown, guessed by K2 and verified across shards
	if (m1 > m0 && first_rec_abs >= u0 + u_own) { h->blocks.clear(); h->crc.clear(); h->shard_own_members = 0; h->shard_limit = 0; h->total = 0; cbeg = cend = 0; }   // header only: owns no record
	upload_compressed(h, bytes, cbeg, cend);
	h->csize = cend - cbeg;
	h->tm.h2d_ms = t.stop();
	h->tm.compressed_bytes = (int64_t)(cend - cbeg); h->tm.inflated_bytes = u_own;
}

// ---- layout of the tile stream: K1 chunks, tiles (whole chunks), token ring, static device tables -------------------------
// NGSQC_TILE_MEMBERS=k (tests): chunks and tiles of k members. NGSQC_TILE_CHUNKS: chunks per tile (default 2).
// NGSQC_K1_CHUNK_DIV: chunk = one decoder round / div. NGSQC_CARRY_MAX: bytes reserved in front of a tile for a straddling record.
void plan_layout_now(ngsqc_handle* h, bool early_pass = false)
{
	if (h->planned) return;
	const double pl0 = wall_ms();
	const int64_t nb = (int64_t)h->blocks.size();
	h->planned = true;
	if (nb == 0) return;
	int64_t div = 1; if (const char* e = getenv("NGSQC_K1_CHUNK_DIV")) div = std::max<int64_t>(1, atoll(e));
	const int64_t mul = 1;
	// A streamed image is bound by PCIe (60 GB in 1.2 s against 0.57 s of K1), and what a one-shot tool waits for besides the copy is the ALLOCATION of the stream's
	// buffers (28 GB/s when another process has just given the memory back): half-size chunks and one chunk per tile cut the ring, the token pool and the tile
	// buffers from 65 GB to 23 GB for the 30x file; the job stays behind the copy
	if (h->stream_img && !getenv("NGSQC_K1_CHUNK_DIV")) div = 2;
	const int64_t lanes = std::max<int64_t>(64, (int64_t)h->n_cu * K1_CHUNK_WAVES_PER_CU * 64 * mul / div);
	// two K1 chunks per tile (192 M reads, 12 chunks; job Mreads/s | un-pipelined scan-stage share of the HBM roofline): 1 chunk 919 | 0.36, 2 chunks 931-941 | 0.43-0.44, 4 chunks
	// 930 | 0.46. The job barely cares; the chain walk of the fused scan has one thread per MEMBER, so a tile of 195 k members keeps twice the lines in flight of a 97 k one.
	int64_t cpt = h->stream_img ? 1 : 2; if (const char* e = getenv("NGSQC_TILE_CHUNKS")) cpt = std::max<int64_t>(1, atoll(e));
	// Three tile buffers: K1 of tile t+2 is queued before the host waits for tile t, so the decoder waves never run out of queued work while the
	// host reads back K2 / consumer results of tile t (with two buffers the queue ran dry for ~6 ms per tile).
	h->nbuf = 3;
	bool forced = false;
	if (const char* e = getenv("NGSQC_TILE_MEMBERS")) { h->chunk = std::max<int64_t>(1, atoll(e)); cpt = 1; forced = true; }
	else
	{
		const int64_t nch0 = std::max<int64_t>(1, (nb + lanes - 1) / lanes);
		h->chunk = (((nb + nch0 - 1) / nch0) + 63) & ~63ll;   // equal chunks, whole waves
	}
	h->nch = (nb + h->chunk - 1) / h->chunk;
	// token pool of a chunk slot: the pages the chunk with the largest need may take (k1_types.h), queue order inside every chunk
	std::vector<uint32_t> ord((size_t)nb);
	std::vector<int64_t> chunk_bytes((size_t)h->nch, 0);
	double pool_factor = 1.0; if (const char* e = getenv("NGSQC_TOKEN_POOL_FACTOR")) pool_factor = std::max(0.01, atof(e));   // (tests: a small pool forces the second-chance path)
	h->slot_pages = 0;
	for (int64_t c = 0; c < h->nch; ++c)
	{
		const int64_t c0 = c * h->chunk, cn = std::min(h->chunk, nb - c0);
		uint64_t sc = 0, su = 0;
		for (int64_t i = 0; i < cn; ++i) { sc += h->blocks[(size_t)(c0 + i)].clen; su += h->blocks[(size_t)(c0 + i)].usize; ord[(size_t)(c0 + i)] = (uint32_t)i; }
		chunk_bytes[(size_t)c] = (int64_t)su;
		h->slot_pages = std::max<int64_t>(h->slot_pages, (int64_t)((double)k1_pool_pages(sc, su, (uint64_t)cn, false) * pool_factor) + 1);
		// queue order inside the chunk: largest compressed size first (the 64 lanes of a wave finish together)
		std::stable_sort(ord.begin() + c0, ord.begin() + c0 + cn, [&](uint32_t a, uint32_t b) { return h->blocks[(size_t)(c0 + a)].clen > h->blocks[(size_t)(c0 + b)].clen; });
	}
	// the kernels address a literal table by a 32-bit WORD offset into the slot (page * K1_PAGE_WORDS): a slot never holds 2^22 pages or more (16 GiB; poorly
	// compressible payloads could ask for that) - members that then find the pool used up take the second-chance path like any other overflow
	h->slot_pages = std::min<int64_t>(h->slot_pages, (1ll << 22) - 1);
	h->k1_slots = K1_SLOTS_DEFAULT; if (const char* e = getenv("NGSQC_TOKEN_SLOTS")) h->k1_slots = std::min(8, std::max(2, atoi(e)));
	const int64_t n_slots = std::min<int64_t>(h->k1_slots, h->nch);
	// tiles: as many chunks as fit the tile buffers next to the ring (at most cpt)
	int64_t carry_max = 64ll << 20; if (const char* e = getenv("NGSQC_CARRY_MAX")) carry_max = std::max<int64_t>(0, atoll(e));
	if (!forced && h->nch > 1)
	{
		size_t free_b = 0, total_b = 0;
		reaper().drain();   // (memory of a handle that was just closed counts as free)
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
		{
			const double fixed = (double)n_slots * (double)h->slot_pages * (double)K1_PAGE_WORDS * 4.0 + (double)nb * 64.0 + (double)h->nbuf * (double)carry_max;
			int64_t max_chunk = 0; for (int64_t b : chunk_bytes) max_chunk = std::max(max_chunk, b);
			const double avail = (double)free_b * 0.85 - fixed;
			int64_t fit = (int64_t)(avail / (((double)h->nbuf + 0.15) * (double)std::max<int64_t>(max_chunk, 1)));   // the tile buffers + record index / long list
			if (fit < 1) throw std::runtime_error("not enough device memory for one K1 chunk (" + std::to_string(max_chunk) + " inflated bytes)");
			cpt = std::min(cpt, fit);
		}
	}
	if (h->nch <= cpt) cpt = h->nch;
	h->tiles.clear(); h->tile_first_chunk.clear();
	for (int64_t c = 0; c < h->nch; c += cpt)
	{
		const int64_t m0 = c * h->chunk, m1 = std::min(nb, (c + cpt) * h->chunk);
		h->tiles.emplace_back(m0, m1 - m0); h->tile_first_chunk.push_back(c);
	}
	h->tile_first_chunk.push_back(h->nch);
	const int nt = (int)h->tiles.size();
	h->pfx = nt > 1 ? ((carry_max + 255) & ~255ll) : 0;
	// static K1 descriptors: upos relative to the tile's first member
	std::vector<BlockDesc> kd((size_t)nb); h->max_tile_bytes = 0;
	for (int t = 0; t < nt; ++t)
	{
		const int64_t f = h->tiles[(size_t)t].first, m = h->tiles[(size_t)t].second; const uint64_t u_lo = h->blocks[(size_t)f].upos;
		for (int64_t i = f; i < f + m; ++i) { kd[(size_t)i] = h->blocks[(size_t)i]; kd[(size_t)i].upos -= u_lo; }
		h->max_tile_bytes = std::max<int64_t>(h->max_tile_bytes, (int64_t)(h->blocks[(size_t)(f + m - 1)].upos + h->blocks[(size_t)(f + m - 1)].usize - u_lo));
	}
	if (h->stream_img)
	{
		// ring of chunk slots: chunk c's bytes [lo_c, hi_c + 64) go to slot c % slots; a member's cpos becomes its place in that slot (static: d_kdesc is built once)
		h->comp_slots = (int)std::min<int64_t>(8, h->nch);   // (eight half-size chunks = 7.6 GB of the 30x file: the copy runs well ahead of K1, so the host rarely blocks on a piece) if (const char* e = getenv("NGSQC_COMP_SLOTS")) h->comp_slots = (int)std::min<int64_t>(h->nch, std::max(2, atoi(e)));
		h->chunk_lo.assign((size_t)h->nch, 0); std::vector<uint64_t> chunk_hi((size_t)h->nch, 0); size_t slot = 0;
		for (int64_t c = 0; c < h->nch; ++c)
		{
			const int64_t c0 = c * h->chunk, cn = std::min(h->chunk, nb - c0);
			h->chunk_lo[(size_t)c] = h->blocks[(size_t)c0].cpos & ~15ull;
			chunk_hi[(size_t)c] = std::min<uint64_t>(h->blocks[(size_t)(c0 + cn - 1)].cpos + h->blocks[(size_t)(c0 + cn - 1)].clen + 64, h->up->map_n);
			slot = std::max<size_t>(slot, (size_t)(chunk_hi[(size_t)c] - h->chunk_lo[(size_t)c]));
		}
		h->comp_slot_bytes = (slot + 1024 + 4095) & ~(size_t)4095;
		h->d_comp.alloc((size_t)h->comp_slots * h->comp_slot_bytes + 1024);
		HIPCHK(hipMemsetAsync(h->d_comp.p, 0, (size_t)h->comp_slots * h->comp_slot_bytes + 1024, h->stream));   // (the bytes behind a slot's last payload are read as padding)
		size_t piece = 64u << 20; if (const char* e = getenv("NGSQC_H2D_PIECE_MB")) piece = (size_t)std::max(1, atoi(e)) << 20;
		ngsqc_handle::Upload* u = h->up; u->sp.clear(); u->chunk_first.assign((size_t)h->nch + 1, 0);
		for (int64_t c = 0; c < h->nch; ++c)
		{
			const int64_t c0 = c * h->chunk, cn = std::min(h->chunk, nb - c0);
			const size_t base = (size_t)(c % h->comp_slots) * h->comp_slot_bytes;
			for (int64_t i = c0; i < c0 + cn; ++i) kd[(size_t)i].cpos = base + (h->blocks[(size_t)i].cpos - h->chunk_lo[(size_t)c]);
			u->chunk_first[(size_t)c] = u->sp.size();
			for (uint64_t o = h->chunk_lo[(size_t)c]; o < chunk_hi[(size_t)c]; o += piece)
				u->sp.push_back(ngsqc_handle::Upload::SPiece{(size_t)o, base + (size_t)(o - h->chunk_lo[(size_t)c]), (size_t)std::min<uint64_t>(piece, chunk_hi[(size_t)c] - o), c});
		}
		u->chunk_first[(size_t)h->nch] = u->sp.size();
	}
	h->d_kdesc.upload(kd, h->stream); h->d_order.upload(ord, h->stream); h->d_crc.upload(h->crc, h->stream);
	h->d_tok_cnt.ensure((size_t)nb + 8); h->d_tok_first.ensure((size_t)nb + 8); h->d_status.ensure((size_t)nb); h->d_work.ensure((size_t)h->nch); h->d_pool_ctr.ensure((size_t)h->nch);
	dbg_stamp("layout: member tables on the device");
	h->d_tok.ensure((size_t)(n_slots * h->slot_pages) * K1_PAGE_WORDS + 16);
	dbg_stamp("layout: token pool allocated");
	for (int i = 0; i < std::min(nt, h->nbuf); ++i) h->buf[i].ensure((size_t)(h->pfx + h->max_tile_bytes) + 64);
	dbg_stamp("layout: tile buffers allocated");
	h->max_tile_members = 0; for (auto& tl : h->tiles) h->max_tile_members = std::max(h->max_tile_members, tl.second);
	h->p_status.ensure((size_t)nb);
	while ((int64_t)h->ev_chunk.size() < 4 * h->nch) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->ev_chunk.push_back(e); }
	while ((int)h->ev_tile.size() < 2 * nt) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->ev_tile.push_back(e); }
	h->p_small.ensure(64); h->p_rb.ensure((size_t)ngsqc_handle::RB_TOTAL);
	HIPCHK(hipStreamSynchronize(h->stream));   // the host vectors above go out of scope
	h->tm.n_tiles = nt;
	if (h->stream_img && early_pass) { stream_pass_begin(h); h->up->pass_fresh = true; }   // (ngsqc_open's layout thread: the copy starts now)
	if (getenv("NGSQC_DEBUG")) fprintf(stderr, "[ngsqc] layout: %d tiles, %lld chunks, token pool %.1f GB, tile buffers %.1f GB, %.1f ms\n", nt, (long long)h->nch, (double)n_slots * (double)h->slot_pages * K1_PAGE_WORDS * 4e-9, (double)std::min(nt, h->nbuf) * (double)(h->pfx + h->max_tile_bytes) * 1e-9, wall_ms() - pl0);
}

void plan_layout(ngsqc_handle* h)
{
	if (h->plan_thread.joinable())
	{
		h->plan_thread.join();
		if (!h->plan_err.empty()) { const std::string e = h->plan_err; h->plan_err.clear(); throw std::runtime_error(e); }
	}
	plan_layout_now(h);
}

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
		hipStream_t s1 = k1_serial ? h->s_p2 : h->s_p1[c & 1];
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
	// NGSQC_WALKERS = 1 / 2 / 4 / 8, default 1: a tile of the 30x file is 190 k members = three waves per SIMD already, and more waves than the chip holds buy nothing - profiles/r05_scan_probe.txt); the general path below (and a shard's first tile, whose chain is anchored by a guess) works on whole members
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
			uint32_t bs0 = 0;
			HIPCHK(hipMemcpyAsync(&bs0, base + exp0, 4, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
			h->long_reads = bs0 > 8192;
		}
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
template <class F> void stream_tiles(ngsqc_handle* h, F f)
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

template <class F> void for_each_tile(ngsqc_handle* h, F f) { stream_tiles(h, [&](const TileCtx&) { return f(h->cur_tile); }); }

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
	DevBuf<int64_t> d_bq, d_bq_sorted; DevBuf<uint8_t> d_bq_tmp; DevBuf<unsigned long long> d_bq_count; bool bq_ride = false; size_t bq_min = 0;   // MODE_DEPTH with min_baseq riding the walk: records that overlap a region, masked by baseq_list_kernel behind the walk
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
		bq_ride = sp.mode == MODE_DEPTH && sp.min_baseq > 0 && !getenv("NGSQC_BASEQ_INLINE"); if (bq_ride) d_bq_count.ensure(1);
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
void write_bai(ngsqc_handle* h, const char* out_path, bool csi = false, int min_shift = 14)
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

// ---- index-driven partial decode (BamReader::setRegion, src/cppNGS/BamReader.cpp:734-768): a handle over the records of ONE virtual-offset range ----
// Only two parts of the file are looked at: the BGZF members from the start of the file until the BAM header is complete, and the members of the range.
// The range comes from the caller (voff) or from the BAI for a set of named regions (resolved against the header's reference names).
struct RangeRequest { bool by_name = false; uint64_t voff[2] = {0, 0}; const ngsqc_named_region* regions = nullptr; int64_t n_regions = 0; int64_t head_members = 0; };   // head_members > 0: the first records of the file (that many BGZF members from the first record on)

void open_range_common(ngsqc_handle* h, const uint8_t* bytes, size_t n, int device, const RangeRequest& rq)
{
	if (n >= 4 && memcmp(bytes, "CRAM", 4) == 0) throw std::domain_error("a CRAM file has no BGZF members: ngsqc_open / ngsqc_open_memory decode it (cram.hip)");
	init_device(h, device);
	Timer t(h->stream); t.start();
	// ---- header: members from the start of the file, more of them until the header is complete ----
	size_t off = 0; uint64_t upos = 0; std::vector<uint64_t> hdr_off;
	for (int64_t k = 8;; k *= 4)
	{
		walk_bgzf(bytes, n, off, n, k - (int64_t)hdr_off.size() > 0 ? k - (int64_t)hdr_off.size() : k, upos, h->blocks, h->crc, &hdr_off);
		const size_t end = h->blocks.empty() ? 0 : (size_t)(h->blocks.back().cpos + h->blocks.back().clen);
		upload_compressed(h, bytes, 0, end);
		h->total = (int64_t)upos;
		if (read_header(h, (int64_t)h->blocks.size())) break;
		if (off >= n) throw FormatError("Could not read header from BAM/CRAM file " + h->path);
	}
	// ---- the range ----
	uint64_t beg = rq.voff[0], end = rq.voff[1]; bool found = true;
	uint64_t own_end = 0;   // head requests: the records that START in front of this member boundary are the handle's; members behind it only complete the last of them
	if (rq.by_name)
	{
		std::vector<ngsqc_region> regs;
		for (int64_t i = 0; i < rq.n_regions; ++i)
		{
			const std::string want = rq.regions[i].chr ? rq.regions[i].chr : "";
			for (size_t r = 0; r < h->ref_names.size(); ++r) if (h->ref_names[r] == want || chr_norm(h->ref_names[r]) == chr_norm(want)) { regs.push_back(ngsqc_region{(int32_t)r, rq.regions[i].start, rq.regions[i].end}); break; }
		}
		if (!bai_range(h->path, regs.data(), (int64_t)regs.size(), (int32_t)h->ref_names.size(), beg, end, found))
			throw IoError("Could not load index of BAM/CRAM file " + h->path);   // BamReader.cpp:742-746
	}
	if (rq.head_members > 0)
	{
		// from the member that holds the first record on: its virtual offset, and the start of the member head_members further down (or the end of the file)
		size_t k = 0; while (k + 1 < h->blocks.size() && (int64_t)(h->blocks[k].upos + h->blocks[k].usize) <= h->first_rec) ++k;
		found = !h->blocks.empty() && h->first_rec < h->total;
		if (!found && off < n) { walk_bgzf(bytes, n, off, n, 1, upos, h->blocks, h->crc, &hdr_off); h->total = (int64_t)upos; k = h->blocks.size() - 1; found = h->first_rec < h->total; }   // (the header ends exactly at a member end)
		if (found)
		{
			beg = (hdr_off[k] << 16) | (uint64_t)(h->first_rec - (int64_t)h->blocks[k].upos);
			size_t o3 = (size_t)hdr_off[k]; uint64_t u3 = 0; std::vector<BlockDesc> tb; std::vector<uint32_t> tc;
			walk_bgzf(bytes, n, o3, n, rq.head_members, u3, tb, tc);
			end = (uint64_t)o3 << 16;
			// a writer that does not keep records inside one BGZF member (htslib does, bam_write1's bgzf_flush_try; others do not) may cut a record at that boundary:
			// like a shard, the handle takes members behind its own ones to complete it
			int64_t tail = SHARD_TAIL_MEMBERS; if (const char* e = getenv("NGSQC_SHARD_TAIL_MEMBERS")) tail = std::max<int64_t>(0, atoll(e));
			if (o3 < n && tail > 0) { own_end = end; walk_bgzf(bytes, n, o3, n, tail, u3, tb, tc); end = (uint64_t)o3 << 16; if (end == own_end) own_end = 0; }
		}
	}
	const int64_t hdr_first_rec = h->first_rec;   // (inflated offset in the header members' numbering)
	std::vector<BlockDesc> hdr_blocks; hdr_blocks.swap(h->blocks); h->crc.clear();
	h->shard = 0; h->n_shards = 2;                 // like a shard that is not the last one: the last member may end inside a record behind the range
	h->shard_u_base = 0; h->shard_own_members = 0; h->shard_limit = 0; h->total = 0; h->first_rec = 0; h->csize = 0;
	size_t cbeg = 0, cend = 0;
	std::vector<uint64_t> foff;
	const size_t co_beg = (size_t)(beg >> 16), co_end = (size_t)(end >> 16);
	if (found && end > beg)
	{
		if (co_beg >= n || co_end > n) throw ArgError("virtual offset behind the end of the file");
		size_t o2 = co_beg; uint64_t u2 = 0;
		// members from the one that holds `beg` up to the one that holds `end` (inclusive when `end` lies inside it)
		walk_bgzf(bytes, n, o2, (end & 0xffff) ? co_end + 1 : co_end, INT64_MAX, u2, h->blocks, h->crc, &foff);
		// A virtual offset may name an EMPTY member (bgzf_tell of a record that starts right behind a member end gives offset 0 of whatever member follows):
		// the walk drops empty members from the table, so the start is checked against the file, not against the first table entry
		if (!bgzf_member_at(bytes, n, co_beg) || (!h->blocks.empty() && foff[0] != co_beg && (beg & 0xffff))) throw ArgError("virtual offset does not name a BGZF block of this file");
	}
	if (found && end > beg && !h->blocks.empty())   // (a range of empty members only: nothing to read)
	{
		int64_t limit = 0;
		if (own_end)
		{
			limit = (int64_t)(h->blocks.back().upos + h->blocks.back().usize);
			for (size_t i = 0; i < foff.size(); ++i) if (foff[i] >= (own_end >> 16)) { limit = (int64_t)h->blocks[i].upos; break; }
		}
		else if ((end & 0xffff) == 0) limit = (int64_t)(h->blocks.back().upos + h->blocks.back().usize);
		else
		{
			if (foff.back() != co_end) throw ArgError("virtual offset does not name a BGZF block of this file");
			limit = (int64_t)h->blocks.back().upos + (int64_t)(end & 0xffff);
		}
		int64_t first = (int64_t)(beg & 0xffff);
		// a range that starts inside the header members: never in front of the first record
		for (size_t i = 0; i < hdr_off.size(); ++i) if (hdr_off[i] == co_beg) first = std::max<int64_t>(first, hdr_first_rec - (int64_t)hdr_blocks[i].upos);
		cbeg = (size_t)(h->blocks.front().cpos & ~15ull);
		cend = (size_t)(h->blocks.back().cpos + h->blocks.back().clen);
		for (auto& d : h->blocks) d.cpos -= cbeg;
		h->total = (int64_t)(h->blocks.back().upos + h->blocks.back().usize);
		h->shard_own_members = (int64_t)h->blocks.size(); h->shard_limit = std::max(limit, first); h->first_rec = first;
	}
	upload_compressed(h, bytes, cbeg, cend);
	h->csize = cend - cbeg;
	h->tm.h2d_ms = t.stop();
	h->tm.compressed_bytes = (int64_t)(cend - cbeg); h->tm.inflated_bytes = h->shard_limit;
}

int open_impl(ngsqc_handle** out, const char* path, const void* bytes, size_t n, int device, int shard = 0, int n_shards = 1, const RangeRequest* range = nullptr)
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
		// CRAM 3.0 (BamReader.cpp:482-492): the container layer is decoded on the host (cram.hip) into a BAM stream in stored BGZF members; from here on the file is a BAM
		// image in memory. Index-driven requests (a .crai names slices, not BGZF members) fall back to the whole file: a superset of what a region needs.
		ByteImage cram_image; CramQualPlan qplan; const uint8_t* cram_src = nullptr;
		const bool from_cram = is_cram((const uint8_t*)bytes, n);
		if (from_cram)
		{
			std::string err;
			// regions: the slices whose headers overlap them (what the .crai of `samtools index` would name; the slice headers themselves are read instead);
			// the first records: the first two slices; a virtual-offset range means nothing in a CRAM: the whole file
			CramSelect sel;
			if (range && range->by_name) for (int64_t i = 0; i < range->n_regions; ++i) sel.regions.push_back(CramSelect::Region{range->regions[i].chr ? range->regions[i].chr : "", range->regions[i].start, range->regions[i].end});
			if (range && range->head_members > 0) sel.max_slices = std::max<int64_t>(2, range->head_members / 4);   // (a slice holds ~10 000 records, a BGZF member ~250: every longer head BamReader::info asks for - x4 each time - brings more slices)
			// the quality arrays (rANS blocks, about half of the records' bytes) stay compressed and are decoded on the device into the uploaded image (cram_dev.hip):
			// whole-file handles only (a shard uploads a part of the image); NGSQC_CRAM_DEVICE_QUALS=0 keeps them on the host
			const char* eq = getenv("NGSQC_CRAM_DEVICE_QUALS");
			const bool dev_quals = n_shards == 1 && (!eq || atoi(eq) != 0);
			cram_src = (const uint8_t*)bytes;
			const int crc = cram_to_bam_image((const uint8_t*)bytes, n, h->path, cram_image, err, &sel, dev_quals ? &qplan : nullptr);
			if (crc == NGSQC_E_FORMAT) throw FormatError(err);
			if (crc == NGSQC_E_IO) throw IoError(err);
			if (crc == NGSQC_E_UNSUPPORTED) throw std::domain_error(err);
			if (crc != NGSQC_OK) throw std::runtime_error(err);
			bytes = cram_image.data(); n = cram_image.size(); h->from_cram = true;
			range = nullptr;   // (regions, a record range, the first records: the whole file holds them)
		}
		const char* ea = getenv("NGSQC_ASYNC_H2D");
		if (path && !from_cram && n_shards == 1 && !range && (!ea || atoi(ea) != 0)) h->up = new ngsqc_handle::Upload();
		if (range) open_range_common(h, (const uint8_t*)bytes, n, device, *range); else open_common(h, (const uint8_t*)bytes, n, device, shard, n_shards);
		if (from_cram && !qplan.jobs.empty())
		{
			// (the BGZF wrapper of the image is our own and its CRC-32s were taken over blank qualities; every CRAM block was CRC-checked on the host)
			h->verify_crc = false;
			const double ms = cram_device_quals(cram_src, qplan, h->d_comp.p, cram_image.size(), h->stream);
			if (getenv("NGSQC_TIMING")) fprintf(stderr, "[ngsqc] cram: %zu quality blocks (%llu bytes, %zu records) decoded on the device in %.3f ms\n", qplan.jobs.size(), (unsigned long long)qplan.out_bytes, qplan.patches.size(), ms);
		}
		if (h->up) { h->up->map = map; h->up->map_n = map_n; h->up->fd = fd; map = nullptr; fd = -1; }   // the mapping lives until the last piece is copied
		const char* ep = getenv("NGSQC_ASYNC_PLAN");
		if (h->up && (!ep || atoi(ep) != 0))
			h->plan_thread = std::thread([h] {
				try { HIPCHK(hipSetDevice(h->device)); dbg_stamp("layout thread: start"); plan_layout_now(h, true); dbg_stamp("layout thread: done"); }
				catch (std::exception& e) { h->plan_err = e.what(); h->planned = false; }
			});
	}
	catch (FormatError& e) { g_open_error = e.what(); rc = NGSQC_E_FORMAT; }
	catch (ArgError& e) { g_open_error = e.what(); rc = NGSQC_E_ARG; }
	catch (IoError& e) { g_open_error = e.what(); rc = NGSQC_E_IO; }
	catch (std::domain_error& e) { g_open_error = e.what(); rc = NGSQC_E_UNSUPPORTED; }
	catch (std::exception& e) { g_open_error = e.what(); rc = NGSQC_E_DEVICE; }
	if (h->plan_thread.joinable() && rc != NGSQC_OK) h->plan_thread.join();
	if (rc != NGSQC_OK && h->up) upload_join(h);   // (the copier threads read the mapping)
	if (map) munmap(map, map_n);
	if (fd >= 0) ::close(fd);
	if (rc != NGSQC_OK) { ngsqc_close(h); return rc; }
	*out = h;
	return NGSQC_OK;
}

} // namespace

struct ngsqc_handle::Partial
{
	int mode = 0; bool yx = false; ScanState scan; DevBuf<uint8_t> d_ns; GcTables gc; DevBuf<unsigned long long> d_gctab; DevBuf<double> d_gcover;
	// shard protocol: what the order-dependent fix-ups need of the shard's first records (l_seq, counted, passing), kept so that
	// ngsqc_scan_mapping_finish does not inflate the shard's first tile a second time
	static constexpr int64_t HEAD_MAX = 1 << 20; DevBuf<uint32_t> d_head; int64_t head_n = 0;
};

namespace {
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
	FuseGuard fg(h, do_map ? &map.scan : (do_depth ? &dscan : nullptr));
	// the record offsets of a tile are only expanded when a consumer reads them: the mapping scan rides the chain walk (deferred long-CIGAR records and the
	// order-dependent fix-ups ask for them), the site pileup works on the walk's candidate list; the extra depth scan and the raw-read QC read every record
	struct LazyGuard { ngsqc_handle* h; ~LazyGuard() { h->lazy_recoff = false; } } lg{h};
	h->lazy_recoff = do_map && !part && !do_depth && !do_reads && !getenv("NGSQC_EAGER_RECOFF");
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

} // namespace

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
	{ FuseGuard fg(h, &sc); stream_tiles(h, [&](const TileCtx& c) { sc.tile(h, c); return true; }); }
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

int ngsqc_get_timings(const ngsqc_handle* h, ngsqc_timings* t) { if (!h || !t) return NGSQC_E_ARG; *t = h->tm; return NGSQC_OK; }

const char* ngsqc_version(void) { return "ngsqc-hip 0.2 (gfx950; K1 bgzf inflate + crc32, K2 bam record index, K3-K5 scan / pileup / read QC, K6 depth; tile stream)"; }

} // extern "C"
