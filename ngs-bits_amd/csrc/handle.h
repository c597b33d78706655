// Internal header of libngsqc_hip.so's host side (round 5: csrc/api.hip cut into image.hip - the compressed image and the layout of the tile stream -, tiles.hip -
// K1 / K2 per tile -, jobs.hip - the consumers and the jobs - and api.hip - the C entry points of include/ngsqc.h that are not jobs). Nothing here is part of the C ABI.
#pragma once
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

namespace ngsqc { namespace lib {
extern thread_local std::string g_open_error;   // message of the last failing ngsqc_open* of this thread (api.hip)

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
inline Reaper& reaper() { static Reaper r; return r; }
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

inline double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

constexpr int K1_CHUNK_WAVES_PER_CU = 5;   // a K1 chunk = this many decoder waves per CU (x 64 members). Round 6: HALF of what a CU holds - two decoder launches run side by side (the two phase-1 streams), and a
                                          // chunk of 6 with room for 10 left the second launch with four waves per CU until the first was done: the full-size step 506 -> 446 ms with 5 (profiles/r06_schedule_probe.txt)
constexpr int P1_WAVES_PER_CU = 10;       // decoder waves a CU holds (15.4 KB LDS each since the token lines are staged there, round 6; 193 VGPRs = two per SIMD)
constexpr int K1_SLOTS_DEFAULT = 8;  // token ring: chunk c uses slot c % slots (phase 1 of the next chunks runs while phase 2 of c reads; round 6: a fourth slot is worth 1.5 % of the full-size step with chunks of 5 waves per CU - 458 -> 451 ms, a fifth 0.4 %; eight - the most the ring holds - with eight chunks per tile); NGSQC_TOKEN_SLOTS
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
}} // namespace ngsqc::lib

using namespace ngsqc::lib;

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
	bool long_reads = false;   // the file's first records are longer than 8 KiB on average (index_tile): entries are groups of members, nothing is assumed about member starts
	int lr_failures = 0;       // tiles in a row that failed the group path of long-read mode (three: the mode is dropped)
	bool k2_plain = false;   // a tile of the running stream did not pass the chain check on the device: the later tiles walk whole members, as the general path needs them
	std::vector<int64_t> rq_len_hist, rq_cyc;   // results of the last raw-read QC pass
	struct Partial;                        // state between ngsqc_scan_mapping_partial and ngsqc_scan_mapping_finish
	Partial* partial = nullptr;
};

namespace ngsqc { namespace lib {
// NGSQC_DEBUG: where the wall time of an open goes (ms since the first stamp of the process)
inline void dbg_stamp(const char* what)
{
	static const bool on = getenv("NGSQC_DEBUG") != nullptr; static const double t0 = wall_ms();
	if (on) fprintf(stderr, "[ngsqc] t+%.1f ms %s\n", wall_ms() - t0, what);
}

// every entry point that works on a handle: the device of the handle, exceptions -> return code + message
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

// ---- image.hip: the BGZF member table, the BAM header, the compressed image on its way to the device, the layout of the tile stream, open ----
void walk_bgzf(const uint8_t* file, size_t n, size_t& off, size_t off_end, int64_t max_members, uint64_t& upos, std::vector<BlockDesc>& blocks, std::vector<uint32_t>& crc, std::vector<uint64_t>* file_off = nullptr);
void scan_bgzf(const uint8_t* file, size_t n, std::vector<BlockDesc>& blocks, std::vector<uint32_t>& crc, int64_t& total, std::vector<uint64_t>* file_off = nullptr, int threads = 0, bool* in_pieces = nullptr);
std::string inflate_error(const ngsqc_handle* h, int64_t member, uint32_t code);
void inflate_sync(ngsqc_handle* h, const std::vector<int64_t>& idx, const std::vector<BlockDesc>& desc, uint8_t* d_out, int level = 0);
void upload_wait(ngsqc_handle* h, size_t end_byte, hipStream_t st, int slot);
void upload_join(ngsqc_handle* h);
void upload_finish(ngsqc_handle* h);
void stream_pass_begin(ngsqc_handle* h);
void stream_pass_end(ngsqc_handle* h);
void stream_wait_chunk(ngsqc_handle* h, int64_t c, hipStream_t st);
void stream_p2_enqueued(ngsqc_handle* h, int64_t c);
void plan_layout(ngsqc_handle* h);
struct RangeRequest { bool by_name = false; uint64_t voff[2] = {0, 0}; const ngsqc_named_region* regions = nullptr; int64_t n_regions = 0; int64_t head_members = 0; };   // head_members > 0: the first records of the file (that many BGZF members from the first record on)
int open_impl(ngsqc_handle** out, const char* path, const void* bytes, size_t n, int device, int shard = 0, int n_shards = 1, const RangeRequest* range = nullptr);

// ---- tiles.hip: the tile stream (K1 chunks, K2, the resident tile) ----
const int64_t* ensure_recoff(ngsqc_handle* h);
void sync_all(ngsqc_handle* h);
// Visit every tile in file order with the tile resident in HBM; f returns false to stop early
void stream_tiles(ngsqc_handle* h, const std::function<bool(const TileCtx&)>& f);
void for_each_tile(ngsqc_handle* h, const std::function<bool(int)>& f);

// ---- jobs.hip: the consumers of a tile and the jobs made of them ----
void write_bai(ngsqc_handle* h, const char* out_path, bool csi = false, int min_shift = 14);
}} // namespace ngsqc::lib
