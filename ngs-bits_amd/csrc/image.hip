// The compressed BAM on its way to the device: BGZF member table (SAM spec 4.1), BAM header, H2D of the image (resident or streamed through a ring of chunk slots),
// layout of the tile stream, and the open paths (whole file, shard, index-driven range: BamReader::BamReader / setRegion, src/cppNGS/BamReader.cpp:462-523,734-768).
#include "handle.h"

namespace ngsqc { namespace lib {
// ---- BGZF member table (host): SAM spec §4.1 ----
// members of [off, off_end) (off_end: a member start or the end of the file), at most max_members of them; upos continues at `upos`
void walk_bgzf(const uint8_t* file, size_t n, size_t& off, size_t off_end, int64_t max_members, uint64_t& upos, std::vector<BlockDesc>& blocks, std::vector<uint32_t>& crc, std::vector<uint64_t>* file_off)
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

void scan_bgzf(const uint8_t* file, size_t n, std::vector<BlockDesc>& blocks, std::vector<uint32_t>& crc, int64_t& total, std::vector<uint64_t>* file_off, int threads, bool* in_pieces)
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
	{
		const bool p2_hi = getenv("NGSQC_P2_PRIO") && atoi(getenv("NGSQC_P2_PRIO")) != 0;   // (dev: the short-lived phase-2 / CRC waves ahead of the decoder's long-lived ones when a CU's LDS frees up)
		HIPCHK(hipStreamCreateWithPriority(&h->s_p2, hipStreamNonBlocking, p2_hi ? prio_hi : 0));
		HIPCHK(hipStreamCreateWithPriority(&h->s_crc, hipStreamNonBlocking, p2_hi ? prio_hi : 0));
	}
	int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) h->n_cu = cu;
	const int pw = getenv("NGSQC_P1_WAVES") ? std::max(1, atoi(getenv("NGSQC_P1_WAVES"))) : P1_WAVES_PER_CU;   // (dev: decoder waves per CU a launch keeps resident)
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
void inflate_sync(ngsqc_handle* h, const std::vector<int64_t>& idx, const std::vector<BlockDesc>& desc, uint8_t* d_out, int level)
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
	const int64_t cw = getenv("NGSQC_K1_CHUNK_WAVES") ? std::max(1, atoi(getenv("NGSQC_K1_CHUNK_WAVES"))) : K1_CHUNK_WAVES_PER_CU;   // (dev: chunk size in decoder waves per CU)
	const int64_t lanes = std::max<int64_t>(64, (int64_t)h->n_cu * cw * 64 * mul / div);
	// two K1 chunks per tile (192 M reads, 12 chunks; job Mreads/s | un-pipelined scan-stage share of the HBM roofline): 1 chunk 919 | 0.36, 2 chunks 931-941 | 0.43-0.44, 4 chunks
	// 930 | 0.46. The job barely cares; the chain walk of the fused scan has one thread per MEMBER, so a tile of 195 k members keeps twice the lines in flight of a 97 k one.
	// (round 6: four chunks of 5 waves per CU; 17 -> 10 tiles of the 30x file, the full-size step 451 -> 446 (three chunks) -> 443 ms: fewer tile boundaries, where the chunk stream runs thin)
	// (round 6, with the decoder at three waves per SIMD: EIGHT chunks per tile - as many as fit the HBM beside the image of the 30x file, the memory bound below - and eight token slots;
	// 5 tiles of the 30x file. Chunks per tile | full-size step: 4 | 433 ms, 5 | 427-429, 6 | 424, 7 | 422, 8 | 414.6 (K1 wall 405 -> 370 ms: the chunk stream runs thin at every tile
	// boundary), profiles/r06_schedule_probe.txt. A tile of eight chunks is 655 360 walkers = exactly two rounds of the coverage tools' walk)
	int64_t cpt = h->stream_img ? 1 : 8; if (const char* e = getenv("NGSQC_TILE_CHUNKS")) cpt = std::max<int64_t>(1, atoll(e));
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

// ---- index-driven partial decode (BamReader::setRegion, src/cppNGS/BamReader.cpp:734-768): a handle over the records of ONE virtual-offset range ----
// Only two parts of the file are looked at: the BGZF members from the start of the file until the BAM header is complete, and the members of the range.
// The range comes from the caller (voff) or from the BAI for a set of named regions (resolved against the header's reference names).

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

int open_impl(ngsqc_handle** out, const char* path, const void* bytes, size_t n, int device, int shard, int n_shards, const RangeRequest* range)
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

}} // namespace ngsqc::lib
