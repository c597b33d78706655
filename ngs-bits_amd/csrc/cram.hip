// CRAM 3.0 input (hts-specs CRAMv3). The reference reads CRAM through htslib's sam_read1 under the same BamReader as BAM (src/cppNGS/BamReader.cpp:482-492:
// hts_set_fai_filename with the reference genome; :525-572: the required-fields switches). First slice of that row: the CONTAINER layer - file definition,
// containers, slices, blocks (CRC-32 checked), the block codecs raw / gzip / rANS 4x8 order 0 and 1, the encodings EXTERNAL / HUFFMAN / BYTE_ARRAY_LEN /
// BYTE_ARRAY_STOP / BETA / SUBEXP / GAMMA, read features -> CIGAR / bases / qualities, mate chains inside a slice and detached mates, the slice's reference MD5 -
// runs on the HOST (slices in parallel on host threads) and hands the records to the device as a BAM stream in BGZF members with stored blocks, so that K1's
// stored-block path, K2 (record index), K3 / K4 (walk, depth, counters) run unchanged on the GPU. The codecs are the next step onto the device (rANS has four
// interleaved states per block and thousands of blocks per file). Checked against oracle/cram_decode.py, which is pinned on the reference's CRAM fixtures.
// Host code only: no kernel in this file.
#include "common.h"
#include "host_crc.h"
#include <algorithm>
#include <atomic>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>
#include <chrono>
#include <functional>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>
#include <dlfcn.h>

namespace ngsqc {

namespace {
struct CramError : std::runtime_error { using std::runtime_error::runtime_error; };      // a damaged file: reported like a BAM that cannot be read
struct FormatError : std::runtime_error { using std::runtime_error::runtime_error; };
struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };

struct Cur
{
	const uint8_t* d = nullptr; size_t n = 0, p = 0;
	Cur() {}
	Cur(const uint8_t* d_, size_t n_, size_t p_ = 0) : d(d_), n(n_), p(p_) {}
	uint8_t byte() { if (p >= n) throw CramError("truncated CRAM data"); return d[p++]; }
	const uint8_t* take(size_t k) { if (k > n - p || p > n) throw CramError("truncated CRAM data"); const uint8_t* r = d + p; p += k; return r; }
	uint32_t u32() { const uint8_t* q = take(4); return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); }
	int32_t i32() { return (int32_t)u32(); }
	int32_t itf8()
	{
		if (n - p >= 5 && p <= n)   // (the common case: no end test per byte)
		{
			const uint8_t* q = d + p; const uint32_t b0 = q[0];
			if (b0 < 0x80) { p += 1; return (int32_t)b0; }
			if (b0 < 0xc0) { p += 2; return (int32_t)(((b0 & 0x3f) << 8) | q[1]); }
			if (b0 < 0xe0) { p += 3; return (int32_t)(((b0 & 0x1f) << 16) | ((uint32_t)q[1] << 8) | q[2]); }
			if (b0 < 0xf0) { p += 4; return (int32_t)(((b0 & 0x0f) << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]); }
			p += 5; return (int32_t)(((b0 & 0x0f) << 28) | ((uint32_t)q[1] << 20) | ((uint32_t)q[2] << 12) | ((uint32_t)q[3] << 4) | (q[4] & 0x0fu));
		}
		const uint32_t b0 = byte(); uint32_t v;
		if (b0 < 0x80) v = b0;
		else if (b0 < 0xc0) v = ((b0 & 0x3f) << 8) | byte();
		else if (b0 < 0xe0) { v = (b0 & 0x1f) << 16; v |= (uint32_t)byte() << 8; v |= byte(); }
		else if (b0 < 0xf0) { v = (b0 & 0x0f) << 24; v |= (uint32_t)byte() << 16; v |= (uint32_t)byte() << 8; v |= byte(); }
		else { v = (b0 & 0x0f) << 28; v |= (uint32_t)byte() << 20; v |= (uint32_t)byte() << 12; v |= (uint32_t)byte() << 4; v |= byte() & 0x0fu; }
		return (int32_t)v;
	}
	int64_t ltf8()
	{
		const uint32_t b0 = byte(); int k = 0;
		while (k < 8 && (b0 & (0x80u >> k))) ++k;
		uint64_t v = k == 8 ? 0 : (b0 & (0xffu >> (k + 1)));
		for (int i = 0; i < k; ++i) v = (v << 8) | byte();
		return (int64_t)v;
	}
	std::vector<int32_t> array_itf8() { const int32_t k = itf8(); if (k < 0 || (size_t)k > n) throw CramError("bad CRAM array"); std::vector<int32_t> a((size_t)k); for (auto& x : a) x = itf8(); return a; }
};

// ---------------------------------------------------------------------------------------------------------------- rANS 4x8 (CRAMv3 section 13)
struct RansTable { uint16_t F[256]; uint16_t C[256]; uint8_t L[4096]; bool set = false; };
void rans_read_freqs(Cur& c, RansTable& t)
{
	memset(t.F, 0, sizeof t.F); t.set = true;
	int sym = c.byte(), last = sym, rle = 0;
	for (;;)
	{
		int f = c.byte();
		if (f >= 0x80) f = ((f & 0x7f) << 8) | c.byte();
		t.F[sym] = (uint16_t)f;
		if (rle) { --rle; ++sym; if (sym > 255) throw CramError("bad rANS frequency table"); }
		else { sym = c.byte(); if (sym == last + 1) rle = c.byte(); }
		last = sym;
		if (sym == 0) break;
	}
	uint32_t acc = 0;
	for (int s = 0; s < 256; ++s)
	{
		t.C[s] = (uint16_t)acc;
		if (acc + t.F[s] > 4096) throw CramError("rANS frequencies exceed 4096");
		memset(t.L + acc, s, t.F[s]); acc += t.F[s];
	}
	if (acc < 4096) memset(t.L + acc, 0, 4096 - acc);
}
void rans_decode(const uint8_t* d, size_t n, std::vector<uint8_t>& out, size_t expect)   // expect: the decoded size the block header names (the stream's own size field must agree BEFORE anything is decoded)
{
	if (n < 9) throw CramError("truncated rANS block");
	const int order = d[0];
	const uint32_t n_out = (uint32_t)d[5] | ((uint32_t)d[6] << 8) | ((uint32_t)d[7] << 16) | ((uint32_t)d[8] << 24);
	if ((size_t)n_out != expect) throw CramError("rANS block of another size than its block header says");
	out.assign(n_out, 0);
	if (!n_out) return;
	Cur c(d, n, 9);
	uint32_t R[4];
	auto renorm = [&](uint32_t x) { while (x < (1u << 23)) x = (x << 8) | c.byte(); return x; };
	if (order == 0)
	{
		std::unique_ptr<RansTable> t(new RansTable()); rans_read_freqs(c, *t);
		for (int j = 0; j < 4; ++j) R[j] = c.u32();
		for (uint32_t i = 0; i < n_out; ++i)
		{
			const int j = i & 3; const uint32_t m = R[j] & 0xfffu; const uint8_t s = t->L[m]; out[i] = s;
			R[j] = renorm((uint32_t)t->F[s] * (R[j] >> 12) + m - t->C[s]);
		}
		return;
	}
	if (order != 1) throw CramError("unknown rANS order");
	std::vector<RansTable> tabs(256);
	int ctx = c.byte(), last = ctx, rle = 0;
	for (;;)
	{
		rans_read_freqs(c, tabs[(size_t)ctx]);
		if (rle) { --rle; ++ctx; if (ctx > 255) throw CramError("bad rANS context table"); }
		else { ctx = c.byte(); if (ctx == last + 1) rle = c.byte(); }
		last = ctx;
		if (ctx == 0) break;
	}
	for (int j = 0; j < 4; ++j) R[j] = c.u32();
	const uint32_t q = n_out >> 2; uint32_t idx[4] = {0, q, 2 * q, 3 * q}; uint8_t prev[4] = {0, 0, 0, 0};
	auto step = [&](int j) {
		const RansTable& t = tabs[prev[j]];
		if (!t.set) throw CramError("rANS order-1 context without a table");
		const uint32_t m = R[j] & 0xfffu; const uint8_t s = t.L[m]; out[idx[j]++] = s;
		R[j] = renorm((uint32_t)t.F[s] * (R[j] >> 12) + m - t.C[s]); prev[j] = s;
	};
	for (uint32_t i = 0; i < q; ++i) for (int j = 0; j < 4; ++j) step(j);
	while (idx[3] < n_out) step(3);
}

// ---------------------------------------------------------------------------------------------------------------- rANS Nx16 (CRAM 3.1, block method 5)
// hts-specs CRAMcodecs "rANS Nx16" (what htslib links as htscodecs' rANS_static4x16pr.c; neither is in /root/reference): one byte of flags - 0x01 order 1, 0x04 32
// interleaved states instead of 4, 0x08 striped, 0x10 no size, 0x20 stored, 0x40 run lengths, 0x80 bit packing - 7-bit big-endian varints, frequencies that add
// up to a power of two (scaled up to 4096 / 1 << shift by the reader), 16-bit renormalisation. oracle/cram_decode.py rans_nx16_decode is the same text in Python;
// the reference holds no CRAM 3.1 file, so both are held against oracle/cram_encode.py's writer of the same specification only.
struct Nx16
{
	static uint32_t u7(Cur& c) { uint32_t v = 0; for (int k = 0; k < 5; ++k) { const uint8_t b = c.byte(); v = (v << 7) | (b & 0x7fu); if (!(b & 0x80u)) return v; } throw CramError("bad varint in a rANS Nx16 block"); }
	static void alphabet(Cur& c, bool (&A)[256])
	{
		memset(A, 0, sizeof A);
		int sym = c.byte(), last = sym, rle = 0;
		for (;;)
		{
			A[sym] = true;
			if (rle) { --rle; ++sym; if (sym > 255) throw CramError("bad rANS Nx16 alphabet"); }
			else { sym = c.byte(); if (sym == last + 1) rle = c.byte(); }
			last = sym;
			if (sym == 0) break;
		}
	}
	struct Tab { uint32_t F[256]; uint32_t C[256]; std::vector<uint8_t> L; bool set = false; };
	static void finish(Tab& t, int bits)
	{
		uint64_t tot = 0; for (int s = 0; s < 256; ++s) tot += t.F[s];
		if (tot != 0 && tot != (1ull << bits))
		{
			if (tot > (1ull << bits)) throw CramError("rANS Nx16 frequencies exceed their total");
			int sh = 0; while (tot < (1ull << bits)) { tot *= 2; ++sh; }
			for (int s = 0; s < 256; ++s) t.F[s] <<= sh;
		}
		t.L.assign((size_t)1 << bits, 0); uint64_t acc = 0;
		for (int s = 0; s < 256; ++s)
		{
			t.C[s] = (uint32_t)acc;
			if (acc + t.F[s] > (1ull << bits)) throw CramError("rANS Nx16 frequencies exceed their total");
			memset(t.L.data() + acc, s, t.F[s]); acc += t.F[s];
		}
		t.set = true;
	}
	static uint32_t renorm(Cur& c, uint32_t x) { if (x < (1u << 15)) { const uint8_t* q = c.take(2); x = (x << 16) | (uint32_t)q[0] | ((uint32_t)q[1] << 8); } return x; }
	static void order0(Cur& c, size_t n, int N, std::vector<uint8_t>& out)
	{
		bool A[256]; alphabet(c, A);
		std::unique_ptr<Tab> t(new Tab()); memset(t->F, 0, sizeof t->F);
		for (int s = 0; s < 256; ++s) if (A[s]) t->F[s] = u7(c);
		finish(*t, 12);
		uint32_t R[32]; for (int j = 0; j < N; ++j) R[j] = c.u32();
		out.assign(n, 0);
		for (size_t i = 0; i < n; ++i)
		{
			const int j = (int)(i % (size_t)N); const uint32_t f = R[j] & 0xfffu; const uint8_t s = t->L[f]; out[i] = s;
			R[j] = renorm(c, t->F[s] * (R[j] >> 12) + f - t->C[s]);
		}
	}
	static void order1(Cur& c, size_t n, int N, std::vector<uint8_t>& out)
	{
		const uint8_t comp = c.byte(); const int shift = comp >> 4;
		if (shift < 1 || shift > 12) throw CramError("bad frequency precision in a rANS Nx16 block");
		std::vector<uint8_t> tbuf; Cur tc;
		if (comp & 1)
		{
			const uint32_t ulen = u7(c), clen = u7(c);
			if (ulen > (1u << 20)) throw CramError("bad table size in a rANS Nx16 block");
			Cur sub(c.take(clen), clen); order0(sub, ulen, 4, tbuf); tc = Cur(tbuf.data(), tbuf.size());
		}
		Cur& t = (comp & 1) ? tc : c;
		bool A[256]; alphabet(t, A);
		std::vector<Tab> T(256);
		for (int i = 0; i < 256; ++i)
		{
			if (!A[i]) continue;
			memset(T[(size_t)i].F, 0, sizeof T[(size_t)i].F); int run = 0;
			for (int j = 0; j < 256; ++j)
			{
				if (!A[j]) continue;
				if (run) { --run; continue; }
				const uint32_t f = u7(t); T[(size_t)i].F[j] = f;
				if (f == 0) run = t.byte();
			}
			finish(T[(size_t)i], shift);
		}
		uint32_t R[32]; for (int j = 0; j < N; ++j) R[j] = c.u32();
		const size_t q = n / (size_t)N; size_t idx[32]; uint8_t last[32];
		for (int j = 0; j < N; ++j) { idx[j] = (size_t)j * q; last[j] = 0; }
		out.assign(n, 0); const uint32_t mask = (1u << shift) - 1u;
		auto step = [&](int j) {
			const Tab& tb = T[last[j]];
			if (!tb.set) throw CramError("rANS Nx16 order-1 context without a table");
			const uint32_t f = R[j] & mask; const uint8_t s = tb.L[f]; out[idx[j]++] = s;
			R[j] = renorm(c, tb.F[s] * (R[j] >> shift) + f - tb.C[s]); last[j] = s;
		};
		for (size_t i = 0; i < q; ++i) for (int j = 0; j < N; ++j) step(j);
		while (idx[N - 1] < n) step(N - 1);
	}
	// expect: the decoded size the caller knows (a block's raw size; a stripe's share); have_n: the stream may leave its size out
	static void decode(Cur& c, std::vector<uint8_t>& out, size_t expect, int depth = 0)
	{
		if (depth > 2) throw CramError("rANS Nx16 stripes nested too deeply");
		const uint8_t flags = c.byte();
		size_t n = expect;
		if (!(flags & 0x10)) { n = u7(c); if (n != expect) throw CramError("rANS Nx16 block of another size than its header says"); }
		const int N = (flags & 0x04) ? 32 : 4;
		if (flags & 0x08)
		{
			const int k = c.byte(); if (k == 0) throw CramError("rANS Nx16 stripe count of zero");
			std::vector<uint32_t> clen((size_t)k); for (auto& x : clen) x = u7(c);
			out.assign(n, 0); std::vector<uint8_t> part;
			for (int j = 0; j < k; ++j)
			{
				const size_t un = n / (size_t)k + ((n % (size_t)k) > (size_t)j ? 1 : 0);
				Cur sub(c.take(clen[(size_t)j]), clen[(size_t)j]); decode(sub, part, un, depth + 1);
				for (size_t i = 0; i < un; ++i) out[i * (size_t)k + (size_t)j] = part[i];
			}
			return;
		}
		size_t pack_len = 0, rle_len = 0; int nsym = 0; uint8_t P[256];
		if (flags & 0x80) { pack_len = n; nsym = c.byte(); for (int i = 0; i < nsym; ++i) P[i] = c.byte(); n = u7(c); }
		std::vector<uint8_t> mbuf; Cur meta; bool Lr[256]; memset(Lr, 0, sizeof Lr);
		if (flags & 0x40)
		{
			rle_len = n; const uint32_t mlen = u7(c); n = u7(c);
			if (mlen & 1) { const uint8_t* q = c.take(mlen / 2); meta = Cur(q, mlen / 2); }
			else { const uint32_t cm = u7(c); Cur sub(c.take(cm), cm); order0(sub, mlen / 2, 4, mbuf); meta = Cur(mbuf.data(), mbuf.size()); }
			int k = meta.byte(); if (k == 0) k = 256;
			for (int i = 0; i < k; ++i) Lr[meta.byte()] = true;
		}
		// (sizes come from the file: nothing larger than what the caller expects may be asked for - packing and run lengths only ever shrink the coded stream)
		if (n > (expect > 64 ? expect : 64) * 2 + 1024) throw CramError("rANS Nx16 block with an implausible inner size");
		std::vector<uint8_t> data;
		if (flags & 0x20) { const uint8_t* q = c.take(n); data.assign(q, q + n); }
		else if (flags & 0x01) order1(c, n, N, data);
		else order0(c, n, N, data);
		if (flags & 0x40)
		{
			std::vector<uint8_t> o; o.reserve(rle_len);
			for (uint8_t s : data)
			{
				size_t k = 1; if (Lr[s]) k = (size_t)u7(meta) + 1;
				if (o.size() + k > rle_len) throw CramError("rANS Nx16 run lengths do not add up");
				o.insert(o.end(), k, s);
			}
			if (o.size() != rle_len) throw CramError("rANS Nx16 run lengths do not add up");
			data.swap(o);
		}
		if (flags & 0x80)
		{
			std::vector<uint8_t> o(pack_len);
			if (nsym <= 1) { if (pack_len && nsym == 0) throw CramError("rANS Nx16 packing without symbols"); if (pack_len) memset(o.data(), P[0], pack_len); data.swap(o); }
			else if (nsym <= 16)
			{
				const int per = nsym <= 2 ? 8 : (nsym <= 4 ? 4 : 2), bits = nsym <= 2 ? 1 : (nsym <= 4 ? 2 : 4);
				if (data.size() < (pack_len + (size_t)per - 1) / (size_t)per) throw CramError("rANS Nx16 packed data too short");
				for (size_t i = 0; i < pack_len; ++i) { const uint32_t v = (data[i / (size_t)per] >> (bits * (int)(i % (size_t)per))) & ((1u << bits) - 1u); if ((int)v >= nsym) throw CramError("rANS Nx16 packed symbol out of range"); o[i] = P[v]; }
				data.swap(o);
			}
		}
		if (data.size() != expect) throw CramError("rANS Nx16 block of another size than its header says");
		out.swap(data);
	}
};
void rans_nx16_decode(const uint8_t* d, size_t n, std::vector<uint8_t>& out, size_t expect) { Cur c(d, n); Nx16::decode(c, out, expect); }

// the plan of one rANS block for the device decoder (cram_dev.hip): false = not eligible (the host decodes it): more than 64 different symbols, a short block,
// tables that do not parse
bool rans_plan(const uint8_t* d, size_t n, uint64_t file_off, size_t rsize, CramQualPlan::Job& job, std::vector<uint16_t>& tabs, std::vector<uint8_t>& syms)
{
	if (n < 9 || d[0] > 1) return false;
	const int order = d[0];
	const uint32_t n_out = (uint32_t)d[5] | ((uint32_t)d[6] << 8) | ((uint32_t)d[7] << 16) | ((uint32_t)d[8] << 24);
	if ((size_t)n_out != rsize || n_out < 2048) return false;
	Cur c(d, n, 9);
	std::vector<RansTable> T(order ? 256 : 1);
	try
	{
		if (!order) rans_read_freqs(c, T[0]);
		else
		{
			int ctx = c.byte(), last = ctx, rle = 0;
			for (;;)
			{
				rans_read_freqs(c, T[(size_t)ctx]);
				if (rle) { --rle; ++ctx; if (ctx > 255) return false; }
				else { ctx = c.byte(); if (ctx == last + 1) rle = c.byte(); }
				last = ctx;
				if (ctx == 0) break;
			}
		}
	}
	catch (CramError&) { return false; }
	if (n - c.p < 16) return false;
	bool used[256]; memset(used, 0, sizeof used); used[0] = order == 1;   // (order 1 starts in context 0)
	for (size_t t = 0; t < T.size(); ++t) if (T[t].set) { if (order) used[t] = true; for (int x = 0; x < 256; ++x) if (T[t].F[x]) used[x] = true; }
	int ns = 0; uint8_t list[64];
	for (int x = 0; x < 256; ++x) if (used[x]) { if (ns == 64) return false; list[ns++] = (uint8_t)x; }
	if (ns == 0) return false;
	syms.assign(320, 0xff);
	for (int k = 0; k < ns; ++k) { syms[(size_t)k] = list[k]; syms[64 + (size_t)list[k]] = (uint8_t)k; }
	const int rows = order ? ns : 1;
	tabs.assign((size_t)rows * (size_t)(ns + 1), 0);
	for (int r = 0; r < rows; ++r)
	{
		const RansTable& t = T[order ? (size_t)list[r] : 0];
		if (!t.set) continue;   // (a context nothing is coded in: a row of zeros - the kernel flags its use)
		uint16_t* C = &tabs[(size_t)r * (size_t)(ns + 1)];
		for (int k = 0; k < ns; ++k) C[k] = t.C[list[k]];
		C[ns] = (uint16_t)(t.C[list[ns - 1]] + t.F[list[ns - 1]]);
	}
	job = CramQualPlan::Job{file_off + c.p, 0, (uint32_t)(n - c.p), n_out, 0, 0, (uint32_t)order, (uint32_t)ns};
	return true;
}

// ---------------------------------------------------------------------------------------------------------------- bzip2 / lzma blocks
// (what `samtools view -O cram,use_bzip2 / use_lzma` writes for some series.) The image carries the runtime libraries but not their headers: the two one-shot
// entry points are declared here as bzlib.h / lzma.h declare them and the libraries are loaded on first use; without them such a block is NGSQC_E_UNSUPPORTED.
typedef int (*bz2_decompress_fn)(char* dest, unsigned int* dest_len, char* source, unsigned int source_len, int small, int verbosity);   // BZ2_bzBuffToBuffDecompress
typedef int (*lzma_decode_fn)(uint64_t* memlimit, uint32_t flags, const void* allocator, const uint8_t* in, size_t* in_pos, size_t in_size, uint8_t* out, size_t* out_pos, size_t out_size);   // lzma_stream_buffer_decode
void* load_symbol(const char* const* libs, const char* name)
{
	for (; *libs; ++libs) if (void* h = dlopen(*libs, RTLD_NOW | RTLD_GLOBAL)) if (void* f = dlsym(h, name)) return f;
	return nullptr;
}
void bz2_block(const uint8_t* raw, size_t csize, size_t rsize, std::vector<uint8_t>& out)
{
	static const char* const libs[] = {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so", nullptr};
	static bz2_decompress_fn fn = (bz2_decompress_fn)load_symbol(libs, "BZ2_bzBuffToBuffDecompress");
	if (!fn) throw std::domain_error("CRAM block compressed with bzip2 and no libbz2 on this machine");
	out.assign(rsize ? rsize : 1, 0); unsigned int got = (unsigned int)out.size();
	if (fn((char*)out.data(), &got, (char*)const_cast<uint8_t*>(raw), (unsigned int)csize, 0, 0) != 0 || got != rsize) throw CramError("bzip2 block of the CRAM file does not decompress");
	out.resize(rsize);
}
void lzma_block(const uint8_t* raw, size_t csize, size_t rsize, std::vector<uint8_t>& out)
{
	static const char* const libs[] = {"liblzma.so.5", "liblzma.so", nullptr};
	static lzma_decode_fn fn = (lzma_decode_fn)load_symbol(libs, "lzma_stream_buffer_decode");
	if (!fn) throw std::domain_error("CRAM block compressed with lzma and no liblzma on this machine");
	out.assign(rsize ? rsize : 1, 0); uint64_t memlimit = 1ull << 31; size_t in_pos = 0, out_pos = 0;
	if (fn(&memlimit, 0, nullptr, raw, &in_pos, csize, out.data(), &out_pos, rsize) != 0 || out_pos != rsize) throw CramError("lzma block of the CRAM file does not decompress");
	out.resize(rsize);
}

// What a caller does not need of the records (the reference's readers say so too: BamReader::skipBases / skipTags / skipQualities set htslib's
// CRAM_OPT_REQUIRED_FIELDS, BamReader.cpp:525-572): read names and / or optional fields. Their blocks - gzip, mostly: a third of a CRAM's bytes - are then
// not inflated and the records carry "*" / no tags. Only external blocks that nothing else reads can be left out; a series in the core block is decoded and dropped.
std::atomic<int> g_cram_skip{0};
thread_local int tl_cram_skip = -1;   // a thread's own choice for the files IT opens (-1: the process-wide one) - a guard around one function's opens must not change what another thread's open decodes
} // namespace
void cram_set_skip(int flags) { g_cram_skip = flags & 3; }
int cram_set_skip_thread(int flags) { const int old = tl_cram_skip; tl_cram_skip = flags < 0 ? -1 : (flags & 3); return old; }
int cram_skip() { return tl_cram_skip >= 0 ? tl_cram_skip : g_cram_skip.load(); }
namespace {

// ---------------------------------------------------------------------------------------------------------------- blocks, containers
struct Blk { int method = 0, ctype = 0; int32_t cid = 0; const uint8_t* p = nullptr; size_t n = 0; std::vector<uint8_t> own; const uint8_t* raw = nullptr; size_t raw_n = 0; bool lazy = false; };
uint32_t crc_of(const uint8_t* p, size_t n) { return host_crc32(p, n); }   // (host_crc.h: carry-less multiplication where the CPU has it, else zlib)
void read_block(Cur& c, Blk& b, int32_t lazy_cid = -1, const std::set<int32_t>* skip_ids = nullptr)   // lazy_cid: an external rANS block with this content id stays compressed (b.lazy; b.n = its decoded size); skip_ids: external blocks nobody will read
{
	const size_t start = c.p;
	b.method = c.byte(); b.ctype = c.byte(); b.cid = c.itf8(); const int32_t csize = c.itf8(), rsize = c.itf8();
	if (csize < 0 || rsize < 0 || rsize > (1 << 30)) throw CramError("bad CRAM block sizes");
	const uint8_t* raw = c.take((size_t)csize);
	const size_t crc_at = c.p; const uint32_t crc = c.u32();
	if (crc_of(c.d + start, crc_at - start) != crc) throw CramError("CRAM block CRC mismatch");
	b.raw = raw; b.raw_n = (size_t)csize;
	if (lazy_cid >= 0 && b.ctype == 4 && b.cid == lazy_cid && b.method == 4) { b.lazy = true; b.p = nullptr; b.n = (size_t)rsize; return; }
	if (skip_ids && b.ctype == 4 && skip_ids->count(b.cid)) { b.p = nullptr; b.n = 0; return; }   // (CRC checked, not inflated)
	if (b.method == 0) { b.p = raw; b.n = (size_t)csize; }
	else if (b.method == 1)
	{
		b.own.assign((size_t)rsize, 0);
		z_stream z; memset(&z, 0, sizeof z);
		if (inflateInit2(&z, 15 + 16) != Z_OK) throw CramError("zlib init failed");
		uint8_t none = 0;
		z.next_in = const_cast<Bytef*>(raw); z.avail_in = (uInt)csize; z.next_out = rsize ? b.own.data() : &none; z.avail_out = (uInt)rsize;
		const int rc = inflate(&z, Z_FINISH); const size_t got = z.total_out; inflateEnd(&z);
		if (rc != Z_STREAM_END || got != (size_t)rsize) throw CramError("gzip block of the CRAM file does not inflate");
		b.p = b.own.data(); b.n = b.own.size();
	}
	else if (b.method == 4) { rans_decode(raw, (size_t)csize, b.own, (size_t)rsize); b.p = b.own.data(); b.n = b.own.size(); }
	else if (b.method == 2) { bz2_block(raw, (size_t)csize, (size_t)rsize, b.own); b.p = b.own.data(); b.n = b.own.size(); }
	else if (b.method == 3) { lzma_block(raw, (size_t)csize, (size_t)rsize, b.own); b.p = b.own.data(); b.n = b.own.size(); }
	else if (b.method == 5) { rans_nx16_decode(raw, (size_t)csize, b.own, (size_t)rsize); b.p = b.own.data(); b.n = b.own.size(); }
	else throw std::domain_error("CRAM 3.1 block codec " + std::to_string(b.method) + " (6: adaptive arithmetic coder, 7: fqzcomp, 8: name tokeniser) is not supported by the HIP path");
	if (b.n != (size_t)rsize) throw CramError("CRAM block inflates to another size than its header says");
}
struct ContainerHdr { int32_t length = 0, ref_id = 0, start = 0, span = 0, n_records = 0, n_blocks = 0; int64_t counter = 0, bases = 0; std::vector<int32_t> landmarks; };
void read_container_header(Cur& c, ContainerHdr& k)
{
	const size_t start = c.p;
	k.length = c.i32(); k.ref_id = c.itf8(); k.start = c.itf8(); k.span = c.itf8(); k.n_records = c.itf8(); k.counter = c.ltf8(); k.bases = c.ltf8();
	k.n_blocks = c.itf8(); k.landmarks = c.array_itf8();
	const size_t crc_at = c.p; const uint32_t crc = c.u32();
	if (crc_of(c.d + start, crc_at - start) != crc) throw CramError("CRAM container header CRC mismatch");
	if (k.length < 0) throw CramError("bad CRAM container length");
}

// ---------------------------------------------------------------------------------------------------------------- encodings
enum { E_NULL = 0, E_EXTERNAL = 1, E_HUFFMAN = 3, E_BYTE_ARRAY_LEN = 4, E_BYTE_ARRAY_STOP = 5, E_BETA = 6, E_SUBEXP = 7, E_GAMMA = 9 };
struct Enc
{
	int kind = E_NULL; int32_t a = 0, b = 0;
	std::vector<int32_t> syms, lens; std::unique_ptr<Enc> e1, e2;
	struct Code { int len; uint32_t code; int32_t sym; }; std::vector<Code> codes; int max_len = 0;   // canonical Huffman codes, ascending by (len, code)
	uint32_t first_code[33] = {}, first_at[33] = {}, n_of_len[33] = {};   // per code length: its first code, where its codes start in `codes`, how many there are (they are consecutive)
	bool present = false;
};
void read_encoding(Cur& c, Enc& e)
{
	e.present = true;
	e.kind = c.itf8(); const int32_t n = c.itf8();
	if (n < 0) throw CramError("bad encoding parameters");
	Cur p(c.take((size_t)n), (size_t)n);
	switch (e.kind)
	{
	case E_NULL: break;
	case E_EXTERNAL: e.a = p.itf8(); break;
	case E_HUFFMAN:
	{
		e.syms = p.array_itf8(); e.lens = p.array_itf8();
		if (e.syms.size() != e.lens.size() || e.syms.empty()) throw CramError("bad Huffman encoding");
		std::vector<size_t> order(e.syms.size());
		for (size_t i = 0; i < order.size(); ++i) order[i] = i;
		std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return e.lens[x] != e.lens[y] ? e.lens[x] < e.lens[y] : e.syms[x] < e.syms[y]; });
		uint32_t code = 0; int last = 0;
		for (size_t i : order)
		{
			if (e.lens[i] < 0 || e.lens[i] > 31) throw CramError("bad Huffman code length");
			code <<= (e.lens[i] - last); last = e.lens[i];
			if (e.n_of_len[e.lens[i]]++ == 0) { e.first_code[e.lens[i]] = code; e.first_at[e.lens[i]] = (uint32_t)e.codes.size(); }
			e.codes.push_back(Enc::Code{e.lens[i], code, e.syms[i]}); ++code;
			e.max_len = std::max(e.max_len, e.lens[i]);
		}
		break;
	}
	case E_BYTE_ARRAY_LEN: e.e1.reset(new Enc()); e.e2.reset(new Enc()); read_encoding(p, *e.e1); read_encoding(p, *e.e2); break;
	case E_BYTE_ARRAY_STOP: e.a = p.byte(); e.b = p.itf8(); break;
	case E_BETA: e.a = p.itf8(); e.b = p.itf8(); if (e.b < 0 || e.b > 32) throw CramError("bad BETA encoding"); break;
	case E_SUBEXP: e.a = p.itf8(); e.b = p.itf8(); break;
	case E_GAMMA: e.a = p.itf8(); break;
	default: throw std::domain_error("CRAM encoding " + std::to_string(e.kind) + " is not supported by the HIP path");
	}
}
inline uint16_t ds_key(const char* k) { return (uint16_t)(((uint8_t)k[0] << 8) | (uint8_t)k[1]); }
struct CompHdr
{
	bool RN = true, AP = true, RR = true; uint8_t SM[5] = {0, 0, 0, 0, 0}; bool has_sm = false;
	std::vector<std::vector<std::pair<uint16_t, uint8_t>>> TD;   // per tag line: (2-char tag, type)
	std::map<uint16_t, Enc> ds; std::map<int32_t, Enc> tags;
	char subst[5][4];   // [reference base ACGTN][code] -> read base
	int32_t qs_only_id = -1;   // content id of the external block that ONLY the QS series reads (-1: none): the block the device may decode
	bool skip_rn = false; std::set<int32_t> skip_tags, skip_ids;   // cram_skip(): read names / tag keys that are not decoded, and the external blocks that are then not inflated
	const Enc& series(const char* k) const
	{
		auto it = ds.find(ds_key(k));
		if (it == ds.end()) throw CramError(std::string("CRAM data series ") + k + " is used but has no encoding");
		return it->second;
	}
};
void read_compression_header(const uint8_t* d, size_t n, CompHdr& h)
{
	Cur c(d, n);
	c.itf8();
	for (int32_t i = c.itf8(); i > 0; --i)
	{
		const uint8_t* k = c.take(2); const std::string key((const char*)k, 2);
		if (key == "RN") h.RN = c.byte() != 0;
		else if (key == "AP") h.AP = c.byte() != 0;
		else if (key == "RR") h.RR = c.byte() != 0;
		else if (key == "SM") { memcpy(h.SM, c.take(5), 5); h.has_sm = true; }
		else if (key == "TD")
		{
			const int32_t len = c.itf8(); if (len < 0) throw CramError("bad tag dictionary");
			const uint8_t* td = c.take((size_t)len); size_t o = 0;
			while (o < (size_t)len)
			{
				size_t e = o; while (e < (size_t)len && td[e]) ++e;
				std::vector<std::pair<uint16_t, uint8_t>> line;
				for (size_t x = o; x + 3 <= e; x += 3) line.emplace_back((uint16_t)((td[x] << 8) | td[x + 1]), td[x + 2]);
				h.TD.push_back(std::move(line)); o = e + 1;
			}
		}
		else throw CramError("unknown CRAM preservation key " + std::to_string((int)k[0]) + " " + std::to_string((int)k[1]));
	}
	if (h.TD.empty()) h.TD.emplace_back();
	c.itf8();
	for (int32_t i = c.itf8(); i > 0; --i) { const uint8_t* k = c.take(2); Enc& e = h.ds[(uint16_t)((k[0] << 8) | k[1])]; read_encoding(c, e); }
	c.itf8();
	for (int32_t i = c.itf8(); i > 0; --i) { const int32_t key = c.itf8(); read_encoding(c, h.tags[key]); }
	{
		auto qs = h.ds.find(ds_key("QS"));
		if (qs != h.ds.end() && qs->second.kind == E_EXTERNAL)
		{
			const int32_t id = qs->second.a; int users = 0;
			std::function<void(const Enc&)> walk = [&](const Enc& e) {
				if (e.kind == E_EXTERNAL && e.a == id) ++users;
				if (e.kind == E_BYTE_ARRAY_STOP && e.b == id) ++users;
				if (e.e1) walk(*e.e1);
				if (e.e2) walk(*e.e2);
			};
			for (const auto& kv : h.ds) walk(kv.second);
			for (const auto& kv : h.tags) walk(kv.second);
			if (users == 1) h.qs_only_id = id;
		}
	}
	// what cram_skip() lets go: a series is left out only if all it reads are external blocks that no series which stays reads
	if (const int skip = cram_skip())
	{
		std::function<bool(const Enc&, std::set<int32_t>&)> ids_of = [&](const Enc& e, std::set<int32_t>& out) -> bool {   // false: it reads the core block (bits between other series' bits)
			bool ext = true;
			if (e.kind == E_EXTERNAL) out.insert(e.a);
			else if (e.kind == E_BYTE_ARRAY_STOP) out.insert(e.b);
			else if (e.kind == E_BYTE_ARRAY_LEN) { if (e.e1) ext = ids_of(*e.e1, out) && ext; if (e.e2) ext = ids_of(*e.e2, out) && ext; }
			else if (e.kind == E_HUFFMAN && e.syms.size() <= 1) { /* a constant: no bits at all */ }
			else ext = false;
			return ext;
		};
		std::set<int32_t> needed, rn_ids; std::map<int32_t, std::set<int32_t>> tag_ids;
		bool rn_seen = false, rn_ok = false;
		for (const auto& kv : h.ds)
		{
			if ((skip & 1) && kv.first == ds_key("RN")) { rn_seen = true; rn_ok = ids_of(kv.second, rn_ids); continue; }
			std::set<int32_t> tmp; ids_of(kv.second, tmp); needed.insert(tmp.begin(), tmp.end());
		}
		for (const auto& kv : h.tags)
		{
			std::set<int32_t> tmp; const bool ok = ids_of(kv.second, tmp);
			if ((skip & 2) && ok) tag_ids[kv.first] = tmp; else needed.insert(tmp.begin(), tmp.end());
		}
		auto free_of_needed = [&](const std::set<int32_t>& ids) { for (int32_t id : ids) if (needed.count(id)) return false; return true; };
		// (a tag that shares a block with a series that stays must stay too: until nothing changes)
		auto settle_tags = [&]() {
			for (bool changed = true; changed;)
			{
				changed = false;
				for (auto it = tag_ids.begin(); it != tag_ids.end();)
					if (!free_of_needed(it->second)) { needed.insert(it->second.begin(), it->second.end()); it = tag_ids.erase(it); changed = true; } else ++it;
			}
		};
		// RN is decided FIRST where it cannot be skipped (it reads core bits): its blocks are needed like any other series' before a tag may let go of one it shares
		if (rn_seen && !rn_ok) needed.insert(rn_ids.begin(), rn_ids.end());
		settle_tags();
		if (rn_seen && rn_ok)
		{
			if (free_of_needed(rn_ids)) { h.skip_rn = true; h.skip_ids.insert(rn_ids.begin(), rn_ids.end()); }
			else { needed.insert(rn_ids.begin(), rn_ids.end()); settle_tags(); }   // RN stays because it shares a block with a series that stays: a tag that shares one with RN stays, too
		}
		for (const auto& kv : tag_ids) { h.skip_tags.insert(kv.first); h.skip_ids.insert(kv.second.begin(), kv.second.end()); }
		if (h.qs_only_id >= 0 && h.skip_ids.count(h.qs_only_id)) h.qs_only_id = -1;
	}
	// substitution matrix: for every reference base the four other bases in the order of their 2-bit codes
	const char B[6] = "ACGTN";
	for (int r = 0; r < 5; ++r)
	{
		int k = 0;
		for (int x = 0; x < 4; ++x) h.subst[r][x] = 'N';
		for (int o = 0; o < 5; ++o) { if (o == r) continue; h.subst[r][(h.SM[r] >> (6 - 2 * k)) & 3] = B[o]; ++k; }
	}
}
struct SliceHdr { int32_t ref_id = 0, start = 0, span = 0, n_records = 0, n_blocks = 0, embedded_ref = -1; int64_t counter = 0; std::vector<int32_t> content_ids; uint8_t md5[16]; };
void read_slice_header(const uint8_t* d, size_t n, SliceHdr& s)
{
	Cur c(d, n);
	s.ref_id = c.itf8(); s.start = c.itf8(); s.span = c.itf8(); s.n_records = c.itf8(); s.counter = c.ltf8(); s.n_blocks = c.itf8();
	s.content_ids = c.array_itf8(); s.embedded_ref = c.itf8(); memcpy(s.md5, c.take(16), 16);
	if (s.n_records < 0 || s.n_records > (1 << 27) || s.n_blocks < 0 || s.n_blocks > (1 << 20)) throw CramError("bad CRAM slice header");
}

// ---------------------------------------------------------------------------------------------------------------- MD5 (RFC 1321) of a reference stretch
void md5_of(const uint8_t* data, size_t n, uint8_t out[16])
{
	static const uint32_t K[64] = {
		0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
		0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
		0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
		0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
	static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
	                          4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
	uint32_t a0 = 0x67452301, b0 = 0xefcdab89, c0 = 0x98badcfe, d0 = 0x10325476;
	auto block = [&](const uint8_t* p) {
		uint32_t M[16];
		for (int i = 0; i < 16; ++i) M[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
		uint32_t A = a0, B = b0, C = c0, D = d0;
		for (int i = 0; i < 64; ++i)
		{
			uint32_t F; int g;
			if (i < 16) { F = (B & C) | (~B & D); g = i; }
			else if (i < 32) { F = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
			else if (i < 48) { F = B ^ C ^ D; g = (3 * i + 5) & 15; }
			else { F = C ^ (B | ~D); g = (7 * i) & 15; }
			F = F + A + K[i] + M[g]; A = D; D = C; C = B; B = B + ((F << S[i]) | (F >> (32 - S[i])));
		}
		a0 += A; b0 += B; c0 += C; d0 += D;
	};
	size_t o = 0;
	for (; o + 64 <= n; o += 64) block(data + o);
	uint8_t tail[128]; const size_t rem = n - o; memset(tail, 0, sizeof tail); memcpy(tail, data + o, rem); tail[rem] = 0x80;
	const size_t tl = rem + 9 <= 64 ? 64 : 128; const uint64_t bits = (uint64_t)n * 8;
	for (int i = 0; i < 8; ++i) tail[tl - 8 + i] = (uint8_t)(bits >> (8 * i));
	block(tail); if (tl == 128) block(tail + 64);
	const uint32_t r[4] = {a0, b0, c0, d0};
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(r[i] >> (8 * j));
}

// ---------------------------------------------------------------------------------------------------------------- the reference genome (FASTA + .fai)
class RefGenome
{
public:
	bool open(const std::string& fasta, std::string& err)
	{
		std::ifstream fai(fasta + ".fai");
		if (!fai) { err = "no index " + fasta + ".fai"; return false; }
		std::string line;
		while (std::getline(fai, line))
		{
			std::istringstream is(line); Entry e; std::string name;
			if (!(is >> name >> e.len >> e.offset >> e.line_bases >> e.line_bytes) || e.len < 0 || e.offset < 0 || e.line_bases <= 0 || e.line_bytes < e.line_bases || e.line_bytes > e.line_bases + 2)
			{ err = "damaged index " + fasta + ".fai"; return false; }
			idx_[name] = e;
		}
		fd_ = ::open(fasta.c_str(), O_RDONLY);
		if (fd_ < 0) { err = "cannot open " + fasta; return false; }
		struct stat st; if (fstat(fd_, &st) != 0) { err = "cannot open " + fasta; return false; }
		n_ = (size_t)st.st_size;
		if (n_) { void* m = mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd_, 0); if (m == MAP_FAILED) { err = "cannot map " + fasta; return false; } map_ = (const uint8_t*)m; }
		return true;
	}
	~RefGenome() { if (map_) munmap(const_cast<uint8_t*>(map_), n_); if (fd_ >= 0) ::close(fd_); }
	bool has(const std::string& name) const { return idx_.count(name) != 0; }
	int64_t length(const std::string& name) const { auto it = idx_.find(name); return it == idx_.end() ? -1 : it->second.len; }
	// the whole contig, upper case (loaded on first use; slices of one contig share it). The cache keeps the four contigs used last (evicting the
	// smallest NAME threw chr10..chr22 out on every load once chr7..chr9 were in); a contig is read OUTSIDE the lock - workers of other contigs do not wait for it,
	// workers of the same contig wait for the one that loads it
	std::shared_ptr<const std::string> contig(const std::string& name)
	{
		std::unique_lock<std::mutex> g(mu_);
		while (true)
		{
			auto c = cache_.find(name);
			if (c != cache_.end()) { c->second.used = ++tick_; return c->second.seq; }
			if (!loading_.count(name)) break;
			cv_.wait(g);
		}
		auto it = idx_.find(name);
		if (it == idx_.end()) return nullptr;
		const Entry e = it->second;
		loading_.insert(name);
		g.unlock();
		std::shared_ptr<std::string> s; std::string err;
		try
		{
			s = std::make_shared<std::string>(); s->resize((size_t)e.len);
			int64_t got = 0; size_t o = (size_t)e.offset;
			while (got < e.len)
			{
				const int64_t k = std::min<int64_t>(e.line_bases, e.len - got);
				if (o > n_ || (size_t)k > n_ - o) throw CramError("reference genome is shorter than its index says");
				for (int64_t i = 0; i < k; ++i) { uint8_t ch = map_[o + (size_t)i]; if (ch >= 'a' && ch <= 'z') ch = (uint8_t)(ch - 32); (*s)[(size_t)(got + i)] = (char)ch; }
				got += k; o += (size_t)e.line_bytes;
			}
		}
		catch (const std::exception& ex) { err = ex.what(); s.reset(); }
		g.lock();
		loading_.erase(name);
		if (s)
		{
			while (cache_.size() >= 4)
			{
				auto old = cache_.begin();
				for (auto c = cache_.begin(); c != cache_.end(); ++c) if (c->second.used < old->second.used) old = c;
				cache_.erase(old);
			}
			cache_[name] = Cached{s, ++tick_};
		}
		cv_.notify_all();
		if (!s) throw CramError(err);
		return s;
	}
private:
	struct Entry { int64_t len = 0, offset = 0, line_bases = 0, line_bytes = 0; };
	struct Cached { std::shared_ptr<const std::string> seq; uint64_t used = 0; };
	std::map<std::string, Entry> idx_; std::map<std::string, Cached> cache_; std::set<std::string> loading_; std::mutex mu_; std::condition_variable cv_; uint64_t tick_ = 0;
	int fd_ = -1; const uint8_t* map_ = nullptr; size_t n_ = 0;
};

// ---------------------------------------------------------------------------------------------------------------- decoders of one slice
struct NeedHostQuals {};
struct BitReader
{
	const uint8_t* d = nullptr; size_t n = 0, at = 0;   // at: the next bit, counted from the block's first (most significant first within a byte)
	uint32_t bits(int k)   // k <= 32
	{
		if (k <= 0) return 0;
		if (at + (size_t)k > n * 8) throw CramError("CRAM core block is too short");
		const size_t b = at >> 3; const int off = (int)(at & 7), nb = (off + k + 7) >> 3;   // at most five bytes hold the k bits
		uint64_t w = 0;
		for (int i = 0; i < nb; ++i) w = (w << 8) | d[b + (size_t)i];
		at += (size_t)k;
		return (uint32_t)((w >> (nb * 8 - off - k)) & ((1ull << k) - 1));
	}
	uint32_t bit1() { if (at >= n * 8) throw CramError("CRAM core block is too short"); const uint32_t v = (d[at >> 3] >> (7 - (at & 7))) & 1u; ++at; return v; }
};
struct Dec
{
	BitReader core; std::map<int32_t, Cur> ext; std::vector<Cur*> flat;   // flat: the cursors by content id (ids are small numbers in practice)
	int32_t defer_id = -1; uint64_t defer_pos = 0, defer_n = 0, last_src = 0; bool took = false;   // the quality block that stays compressed: only its cursor moves
	void index()
	{
		flat.clear();
		for (auto& kv : ext) if (kv.first >= 0 && kv.first < 4096) { if ((size_t)kv.first >= flat.size()) flat.resize((size_t)kv.first + 1, nullptr); flat[(size_t)kv.first] = &kv.second; }
	}
	Cur& block(int32_t id)
	{
		if (id == defer_id && defer_id >= 0) throw NeedHostQuals();   // something other than a quality ARRAY reads the block: the slice is decoded again with the block on the host
		if (id >= 0 && (size_t)id < flat.size() && flat[(size_t)id]) return *flat[(size_t)id];
		auto it = ext.find(id); if (it == ext.end()) throw CramError("CRAM external block " + std::to_string(id) + " is missing in a slice"); return it->second;
	}
	int32_t integer(const Enc& e)
	{
		switch (e.kind)
		{
		case E_EXTERNAL: return block(e.a).itf8();
		case E_HUFFMAN:
		{
			if (e.max_len == 0) return e.codes[0].sym;
			uint32_t code = 0;
			for (int len = 1; len <= e.max_len; ++len)
			{
				code = (code << 1) | core.bit1();
				if (e.n_of_len[len] && code >= e.first_code[len] && code - e.first_code[len] < e.n_of_len[len]) return e.codes[e.first_at[len] + (code - e.first_code[len])].sym;
			}
			throw CramError("bad Huffman code in the CRAM core block");
		}
		case E_BETA: return (int32_t)core.bits(e.b) - e.a;
		case E_GAMMA: { int k = 0; while (core.bit1() == 0) { if (++k > 31) throw CramError("bad GAMMA code"); } return (int32_t)((1u << k) | core.bits(k)) - e.a; }
		case E_SUBEXP:
		{
			int i = 0; while (core.bit1() == 1) { if (++i > 31) throw CramError("bad SUBEXP code"); }
			const int nb = i == 0 ? e.b : i + e.b - 1;
			if (nb < 0 || nb > 31) throw CramError("bad SUBEXP code");
			const uint32_t v = i == 0 ? core.bits(nb) : ((1u << nb) | core.bits(nb));
			return (int32_t)v - e.a;
		}
		default: throw CramError("CRAM encoding " + std::to_string(e.kind) + " cannot give an integer");
		}
	}
	uint8_t byte(const Enc& e) { return e.kind == E_EXTERNAL ? block(e.a).byte() : (uint8_t)(integer(e) & 0xff); }
	// a series with its cursor looked up once per slice (EXTERNAL: the common case); what cannot be resolved here goes the general way, with its errors, when it is read
	struct Ser { const Enc* e = nullptr; Cur* c = nullptr; operator const Enc&() const { return *e; } };
	Ser ser(const Enc& e)
	{
		Ser s; s.e = &e;
		if (e.kind == E_EXTERNAL && !(e.a == defer_id && defer_id >= 0) && e.a >= 0 && (size_t)e.a < flat.size()) s.c = flat[(size_t)e.a];
		return s;
	}
	int32_t integer(const Ser& s) { return s.c ? s.c->itf8() : integer(*s.e); }
	uint8_t byte(const Ser& s) { return s.c ? s.c->byte() : byte(*s.e); }
	void bytes_n(const Enc& e, size_t k, std::vector<uint8_t>& out)
	{
		if (e.kind == E_EXTERNAL && e.a == defer_id && defer_id >= 0)
		{
			if (defer_pos + k > defer_n) throw CramError("truncated CRAM data");
			out.assign(k, 0); last_src = defer_pos; defer_pos += k; took = true; return;
		}
		if (e.kind == E_EXTERNAL) { const uint8_t* p = block(e.a).take(k); out.assign(p, p + k); return; }
		out.resize(k); for (size_t i = 0; i < k; ++i) out[i] = byte(e);
	}
	void array(const Enc& e, std::vector<uint8_t>& out)
	{
		if (e.kind == E_BYTE_ARRAY_STOP)
		{
			Cur& c = block(e.b); size_t q = c.p;
			while (q < c.n && c.d[q] != (uint8_t)e.a) ++q;
			if (q >= c.n) throw CramError("CRAM byte array without its stop byte");
			out.assign(c.d + c.p, c.d + q); c.p = q + 1; return;
		}
		if (e.kind == E_BYTE_ARRAY_LEN) { const int32_t k = integer(*e.e1); if (k < 0 || k > (1 << 28)) throw CramError("byte array length out of range"); bytes_n(*e.e2, (size_t)k, out); return; }
		throw CramError("CRAM encoding " + std::to_string(e.kind) + " cannot give a byte array");
	}
};

enum { BAM_FPAIRED = 1, BAM_FUNMAP = 4, BAM_FMUNMAP = 8, BAM_FREVERSE = 16, BAM_FMREVERSE = 32, BAM_FREAD1 = 64 };
enum { CF_QUAL_ARRAY = 1, CF_DETACHED = 2, CF_MATE_DOWNSTREAM = 4, CF_NO_SEQ = 8 };
struct Feature { char code; int32_t pos; int32_t v = 0; uint8_t q = 0; uint32_t boff = 0, blen = 0; };   // boff / blen: the feature's bytes in the record's scratch buffer
struct RecInfo { uint32_t bf = 0, cf = 0; int32_t ref_id = -1, pos = 0, end = 0, mate_line = -1, mf = 0, ns = -1, np = 0, ts = 0; int32_t mate_ref = -1, mate_pos = 0; int64_t tlen = 0; bool tlen_set = false; size_t off = 0; };

struct DecodeEnv
{
	const std::vector<std::string>* ref_names = nullptr; const std::vector<std::string>* rg_ids = nullptr;
	RefGenome* genome = nullptr; bool no_reference = false, ignore_md5 = false; std::string path;
};

inline uint32_t reg2bin14(int64_t beg, int64_t end)
{
	--end;
	if (beg >> 14 == end >> 14) return (uint32_t)(4681 + (beg >> 14));
	if (beg >> 17 == end >> 17) return (uint32_t)(585 + (beg >> 17));
	if (beg >> 20 == end >> 20) return (uint32_t)(73 + (beg >> 20));
	if (beg >> 23 == end >> 23) return (uint32_t)(9 + (beg >> 23));
	if (beg >> 26 == end >> 26) return (uint32_t)(1 + (beg >> 26));
	return 0;
}
inline void put32(std::vector<uint8_t>& o, size_t at, uint32_t v) { o[at] = (uint8_t)v; o[at + 1] = (uint8_t)(v >> 8); o[at + 2] = (uint8_t)(v >> 16); o[at + 3] = (uint8_t)(v >> 24); }
inline void add32(std::vector<uint8_t>& o, uint32_t v) { const size_t at = o.size(); o.resize(at + 4); put32(o, at, v); }
inline void add16(std::vector<uint8_t>& o, uint32_t v) { o.push_back((uint8_t)v); o.push_back((uint8_t)(v >> 8)); }

// one slice -> BAM records (SAM spec 4.2), as htslib's cram_decode_slice + cram_to_bam build them
void decode_slice(const CompHdr& ch, const SliceHdr& sh, std::vector<Blk>& blocks, const DecodeEnv& env, std::vector<uint8_t>& out, std::vector<CramQualPlan::Patch>* patches = nullptr)
{
	Dec D; bool have_core = false;
	for (Blk& b : blocks)
	{
		if (b.ctype == 5 && !have_core) { D.core.d = b.p; D.core.n = b.n; have_core = true; }
		else if (b.ctype == 4 && b.lazy) { D.defer_id = b.cid; D.defer_n = b.n; }
		else if (b.ctype == 4) D.ext[b.cid] = Cur(b.p, b.n);
	}
	if (!have_core) throw CramError("CRAM slice without a core block");
	D.index();
	const uint8_t* embedded = nullptr; size_t embedded_n = 0;
	if (sh.embedded_ref >= 0) { Cur& e = D.block(sh.embedded_ref); embedded = e.d; embedded_n = e.n; }
	// the reference stretch of a single-reference slice (and its MD5: htslib refuses a genome that does not match, "md5sum reference mismatch")
	std::shared_ptr<const std::string> contig; int32_t contig_id = -3;
	auto contig_of = [&](int32_t ref_id) -> const std::string* {
		if (ref_id == contig_id) return contig.get();
		contig_id = ref_id; contig.reset();
		if (ref_id < 0 || (size_t)ref_id >= env.ref_names->size() || !env.genome) return nullptr;
		contig = env.genome->contig((*env.ref_names)[(size_t)ref_id]);
		return contig.get();
	};
	const bool needs_ref = ch.RR && !embedded && !env.no_reference;
	if (needs_ref && sh.ref_id >= 0)
	{
		const std::string* c = contig_of(sh.ref_id);
		if (!c) throw IoError("Error while setting reference genome for cram file " + env.path + ": no sequence '" + ((size_t)sh.ref_id < env.ref_names->size() ? (*env.ref_names)[(size_t)sh.ref_id] : std::string("?")) + "'");
		static const uint8_t zero[16] = {0};
		if (!env.ignore_md5 && memcmp(sh.md5, zero, 16) != 0 && sh.start >= 1 && sh.span > 0)
		{
			const int64_t a = (int64_t)sh.start - 1, b = std::min<int64_t>(a + sh.span, (int64_t)c->size());
			uint8_t m[16]; md5_of((const uint8_t*)c->data() + std::min<int64_t>(a, (int64_t)c->size()), (size_t)std::max<int64_t>(b - a, 0), m);
			if (memcmp(m, sh.md5, 16) != 0) throw FormatError("md5sum reference mismatch for cram file " + env.path + ": the reference genome is not the one the file was written with (sequence '" + (*env.ref_names)[(size_t)sh.ref_id] + "')");
		}
	}
	auto ref_bases = [&](int32_t ref_id, int64_t p0, int64_t n, uint8_t* dst) {   // n reference bases from 0-based p0 (upper case; 'N' outside)
		if (n <= 0) return;
		if (embedded)
		{
			const int64_t o = p0 - ((int64_t)sh.start - 1);
			for (int64_t i = 0; i < n; ++i) { const int64_t x = o + i; uint8_t c = x >= 0 && (size_t)x < embedded_n ? embedded[x] : (uint8_t)'N'; if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32); dst[i] = c; }
			return;
		}
		if (env.no_reference) { memset(dst, 'N', (size_t)n); return; }
		const std::string* c = contig_of(ref_id);
		if (!c) throw IoError("Error while setting reference genome for cram file " + env.path + ": a read needs bases of a sequence the genome does not hold");
		if (p0 >= 0 && (uint64_t)(p0 + n) <= c->size()) { memcpy(dst, c->data() + p0, (size_t)n); return; }
		for (int64_t i = 0; i < n; ++i) { const int64_t x = p0 + i; dst[i] = x >= 0 && (size_t)x < c->size() ? (uint8_t)(*c)[(size_t)x] : (uint8_t)'N'; }
	};

	const size_t nrec = (size_t)sh.n_records;
	{ size_t raw = 0; for (const Blk& b : blocks) raw += b.n; out.reserve(out.size() + 2 * raw + 64 * nrec); }   // (a guess that saves the first doublings; the vector still grows when it is short)
	std::vector<RecInfo> recs(nrec);
	std::vector<uint8_t> name, seq, qual, tmp, tagbytes, fbytes; std::vector<Feature> feats; std::vector<uint32_t> cigar;
	const Dec::Ser eBF = D.ser(ch.series("BF")), eCF = D.ser(ch.series("CF")), eRL = D.ser(ch.series("RL")), eAP = D.ser(ch.series("AP")), eRG = D.ser(ch.series("RG")), eTL = D.ser(ch.series("TL"));
	struct Lazy   // the other series: looked up once, an error only when a record needs one the header does not define
	{
		const CompHdr& ch; Dec& D; const char* key; Dec::Ser s; bool tried = false;
		Lazy(const CompHdr& c, Dec& d, const char* k) : ch(c), D(d), key(k) {}
		const Dec::Ser& get() { if (!tried) { tried = true; auto it = ch.ds.find(ds_key(key)); if (it != ch.ds.end()) s = D.ser(it->second); } if (!s.e) throw CramError(std::string("CRAM data series ") + key + " is used but has no encoding"); return s; }
	};
	Lazy sRI(ch, D, "RI"), sRN(ch, D, "RN"), sMF(ch, D, "MF"), sNS(ch, D, "NS"), sNP(ch, D, "NP"), sTS(ch, D, "TS"), sNF(ch, D, "NF"), sFN(ch, D, "FN"), sFC(ch, D, "FC"), sFP(ch, D, "FP"), sBA(ch, D, "BA"),
	     sQS(ch, D, "QS"), sBS(ch, D, "BS"), sIN(ch, D, "IN"), sSC(ch, D, "SC"), sHC(ch, D, "HC"), sPD(ch, D, "PD"), sDL(ch, D, "DL"), sRS(ch, D, "RS"), sBB(ch, D, "BB"), sQQ(ch, D, "QQ"), sMQ(ch, D, "MQ");
	// the tag encodings of every tag line, resolved once
	std::vector<std::vector<const Enc*>> td_enc(ch.TD.size());
	for (size_t l = 0; l < ch.TD.size(); ++l) for (const auto& tg : ch.TD[l])
	{
		const int32_t key = ((int32_t)(tg.first >> 8) << 16) | ((int32_t)(tg.first & 0xff) << 8) | tg.second;
		auto it = ch.tags.find(key); td_enc[l].push_back(it == ch.tags.end() ? nullptr : &it->second);
	}
	int64_t prev_pos = sh.start;
	static const uint8_t nt16[256] = {
		15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,0,15,15,
		15,1,14,2,13,15,15,4,11,15,15,12,15,3,15,15, 15,15,5,6,8,15,7,9,15,10,15,15,15,15,15,15, 15,1,14,2,13,15,15,4,11,15,15,12,15,3,15,15, 15,15,5,6,8,15,7,9,15,10,15,15,15,15,15,15,
		15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
		15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15};
	for (size_t i = 0; i < nrec; ++i)
	{
		RecInfo& r = recs[i];
		r.bf = (uint32_t)D.integer(eBF); r.cf = (uint32_t)D.integer(eCF);
		r.ref_id = sh.ref_id == -2 ? D.integer(sRI.get()) : sh.ref_id;
		const int32_t rl = D.integer(eRL);
		if (rl < 0 || rl > (1 << 28)) throw CramError("read length out of range");
		const int32_t ap = D.integer(eAP);
		if (ch.AP) { prev_pos += ap; r.pos = (int32_t)prev_pos; } else r.pos = ap;
		const int32_t rg = D.integer(eRG);
		name.clear(); bool have_name = false;
		if (ch.RN && !ch.skip_rn) { D.array(sRN.get(), name); have_name = true; }
		if (r.cf & CF_DETACHED)
		{
			r.mf = D.integer(sMF.get());
			if (!ch.RN && !ch.skip_rn) { D.array(sRN.get(), name); have_name = true; }
			r.ns = D.integer(sNS.get()); r.np = D.integer(sNP.get()); r.ts = D.integer(sTS.get());
		}
		else if (r.cf & CF_MATE_DOWNSTREAM)
		{
			const int32_t nf = D.integer(sNF.get());
			if (nf < 0 || i + (size_t)nf + 1 >= nrec) throw CramError("CRAM mate chain leaves the slice");
			r.mate_line = (int32_t)(i + (size_t)nf + 1);
		}
		const int32_t tl = D.integer(eTL);
		if (tl < 0 || (size_t)tl >= ch.TD.size()) throw CramError("bad tag line index");
		tagbytes.clear();
		for (size_t ti = 0; ti < ch.TD[(size_t)tl].size(); ++ti)
		{
			const auto& tg = ch.TD[(size_t)tl][ti]; const Enc* te = td_enc[(size_t)tl][ti];
			if (!te) throw CramError("CRAM tag without an encoding");
			if (!ch.skip_tags.empty() && ch.skip_tags.count(((int32_t)(tg.first >> 8) << 16) | ((int32_t)(tg.first & 0xff) << 8) | tg.second)) continue;   // (cram_skip: its block was not inflated)
			D.array(*te, tmp);
			tagbytes.push_back((uint8_t)(tg.first >> 8)); tagbytes.push_back((uint8_t)(tg.first & 0xff)); tagbytes.push_back(tg.second);
			tagbytes.insert(tagbytes.end(), tmp.begin(), tmp.end());
		}
		cigar.clear(); int32_t mapq = 0; bool have_seq = true;
		seq.assign((size_t)rl, 'N'); qual.assign((size_t)rl, 0xff);
		auto add_op = [&](uint32_t op, int64_t n) {
			if (n <= 0) return;
			if (!cigar.empty() && (cigar.back() & 15u) == op) cigar.back() += (uint32_t)n << 4; else cigar.push_back(((uint32_t)n << 4) | op);
		};
		if (!(r.bf & BAM_FUNMAP))
		{
			const int32_t fn = D.integer(sFN.get());
			if (fn < 0 || fn > 2 * rl + 64) throw CramError("feature count out of range");
			feats.clear(); feats.resize((size_t)fn); fbytes.clear(); int32_t fpos = 0;
			auto take = [&](const Enc& e, Feature& f) { D.array(e, tmp); f.boff = (uint32_t)fbytes.size(); f.blen = (uint32_t)tmp.size(); fbytes.insert(fbytes.end(), tmp.begin(), tmp.end()); };
			for (Feature& f : feats)
			{
				f.code = (char)D.byte(sFC.get()); fpos += D.integer(sFP.get()); f.pos = fpos;
				switch (f.code)
				{
				case 'B': f.v = D.byte(sBA.get()); f.q = D.byte(sQS.get()); break;
				case 'X': f.v = D.byte(sBS.get()); break;
				case 'I': take(sIN.get(), f); break;
				case 'S': take(sSC.get(), f); break;
				case 'H': f.v = D.integer(sHC.get()); break;
				case 'P': f.v = D.integer(sPD.get()); break;
				case 'D': f.v = D.integer(sDL.get()); break;
				case 'N': f.v = D.integer(sRS.get()); break;
				case 'i': f.v = D.byte(sBA.get()); break;
				case 'b': take(sBB.get(), f); break;
				case 'q': take(sQQ.get(), f); break;
				case 'Q': f.q = D.byte(sQS.get()); break;
				default: throw CramError("unknown CRAM read feature code " + std::to_string((int)(uint8_t)f.code));
				}
			}
			mapq = D.integer(sMQ.get());
			const bool qarr = (r.cf & CF_QUAL_ARRAY) != 0;
			if (qarr) D.bytes_n(sQS.get(), (size_t)rl, qual);
			// read features -> CIGAR, bases, qualities (CRAMv3 section 10.6); between features the read follows the reference
			int64_t ref_pos = (int64_t)r.pos - 1, read_pos = 0;
			// The reference is fetched a stretch at a time, ahead of the features: seq[read_pos .. fill_end) holds the reference bases of the current alignment as long as
			// ref_pos - read_pos is what it was at the fetch (a match moves both; insertions, clips, deletions and skips shift one against the other). One copy per
			// stretch instead of one per feature; the window bounds what a read with many shifts (long reads) fetches in vain.
			int64_t fill_delta = 0, fill_end = -1;
			auto have = [&](int64_t upto) {   // upto > read_pos: seq[read_pos .. upto) = reference
				const bool same = fill_delta == ref_pos - read_pos;
				if (same && fill_end >= upto) return;
				const int64_t from = same && fill_end > read_pos ? fill_end : read_pos, to = std::min<int64_t>(rl, std::max<int64_t>(upto, from + 128));
				ref_bases(r.ref_id, ref_pos + (from - read_pos), to - from, seq.data() + from);
				fill_delta = ref_pos - read_pos; fill_end = to;
			};
			auto match_to = [&](int64_t upto) {
				const int64_t n = upto - read_pos;
				if (n > 0) { have(upto); add_op(0, n); ref_pos += n; read_pos = upto; }
			};
			auto span_ok = [&](int64_t at, size_t n) { if (at < 0 || at + (int64_t)n > rl) throw CramError("CRAM read feature outside the read"); };
			for (const Feature& f : feats)
			{
				const int64_t at = (int64_t)f.pos - 1;
				if (f.code == 'Q') { span_ok(at, 1); if (!qarr) qual[(size_t)at] = f.q; continue; }
				if (f.code == 'q') { span_ok(at, f.blen); if (!qarr && f.blen) memcpy(qual.data() + at, fbytes.data() + f.boff, f.blen); continue; }
				if (at < read_pos) throw CramError("CRAM read features are not ordered");
				span_ok(at, 0);
				match_to(at);
				switch (f.code)
				{
				case 'B': span_ok(at, 1); seq[(size_t)at] = (uint8_t)f.v; if (!qarr) qual[(size_t)at] = f.q; add_op(0, 1); ++ref_pos; ++read_pos; break;
				case 'X':
				{
					span_ok(at, 1);
					have(at + 1); const uint8_t rb = seq[(size_t)at];
					const int ri = rb == 'A' ? 0 : rb == 'C' ? 1 : rb == 'G' ? 2 : rb == 'T' ? 3 : 4;
					seq[(size_t)at] = (uint8_t)ch.subst[ri][f.v & 3]; add_op(0, 1); ++ref_pos; ++read_pos; break;
				}
				case 'I': span_ok(at, f.blen); if (f.blen) memcpy(seq.data() + at, fbytes.data() + f.boff, f.blen); add_op(1, (int64_t)f.blen); read_pos += (int64_t)f.blen; break;
				case 'i': span_ok(at, 1); seq[(size_t)at] = (uint8_t)f.v; add_op(1, 1); ++read_pos; break;
				case 'S': span_ok(at, f.blen); if (f.blen) memcpy(seq.data() + at, fbytes.data() + f.boff, f.blen); add_op(4, (int64_t)f.blen); read_pos += (int64_t)f.blen; break;
				case 'b': span_ok(at, f.blen); if (f.blen) memcpy(seq.data() + at, fbytes.data() + f.boff, f.blen); add_op(0, (int64_t)f.blen); ref_pos += (int64_t)f.blen; read_pos += (int64_t)f.blen; break;
				case 'D': add_op(2, f.v); ref_pos += f.v; break;
				case 'N': add_op(3, f.v); ref_pos += f.v; break;
				case 'H': add_op(5, f.v); break;
				case 'P': add_op(6, f.v); break;
				}
			}
			match_to(rl);
			r.end = cigar.empty() ? r.pos : (int32_t)ref_pos;
			if (r.cf & CF_NO_SEQ) have_seq = false;
		}
		else
		{
			if (r.cf & CF_NO_SEQ) have_seq = false; else D.bytes_n(sBA.get(), (size_t)rl, seq);
			if (r.cf & CF_QUAL_ARRAY) D.bytes_n(sQS.get(), (size_t)rl, qual);
			r.end = r.pos;
		}
		// ---- the BAM record; flag, mate fields and template length are patched when the slice's chains are resolved ----
		r.off = out.size();
		const size_t l_seq = have_seq ? (size_t)rl : 0;
		if (!have_name) name.assign(1, '*');
		if (name.size() > 254) throw CramError("read name longer than 254 bytes");
		const int64_t pos0 = (int64_t)r.pos - 1, end0 = cigar.empty() ? pos0 + 1 : (int64_t)r.end;
		// more than 65535 operations do not fit n_cigar_op: the BAM convention (SAM spec 4.2.2, htslib bam_write1) - a placeholder <l_seq>S<reference length>N and
		// the operations in a CG:B,I tag behind the record's other tags, which every reader of the image (K2 / K3, like htslib's bam_tag2cigar) puts back
		const bool cg = cigar.size() > 65535;
		const int64_t ref_len = end0 - pos0;
		if (cg && (l_seq >= (1u << 28) || ref_len < 0 || ref_len >= (1ll << 28))) throw CramError("a read with more than 65535 CIGAR operations is too long for the BAM placeholder CIGAR");
		const std::string* rg_id = rg >= 0 && (size_t)rg < env.rg_ids->size() ? &(*env.rg_ids)[(size_t)rg] : nullptr;
		const size_t n_cig = cg ? 2 : cigar.size();
		const size_t rec_bytes = 36 + name.size() + 1 + 4 * n_cig + (l_seq + 1) / 2 + l_seq + tagbytes.size() + (rg_id ? rg_id->size() + 4 : 0) + (cg ? 8 + 4 * cigar.size() : 0);
		out.resize(r.off + rec_bytes);   // (the record's size is known: one growth of the vector, the fields written in place)
		uint8_t* w = out.data() + r.off;
		auto w32 = [&](uint32_t v) { w[0] = (uint8_t)v; w[1] = (uint8_t)(v >> 8); w[2] = (uint8_t)(v >> 16); w[3] = (uint8_t)(v >> 24); w += 4; };
		auto w16 = [&](uint32_t v) { w[0] = (uint8_t)v; w[1] = (uint8_t)(v >> 8); w += 2; };
		auto wn = [&](const void* p, size_t k) { if (k) memcpy(w, p, k); w += k; };
		w32((uint32_t)(rec_bytes - 4));   // block_size
		w32((uint32_t)r.ref_id); w32((uint32_t)(r.pos - 1));
		*w++ = (uint8_t)(name.size() + 1); *w++ = (uint8_t)mapq;
		w16(pos0 < 0 ? 4680u : reg2bin14(pos0, end0));
		w16((uint32_t)n_cig); w16(r.bf); w32((uint32_t)l_seq);   // flag, mate fields and template length are patched when the slice's chains are resolved
		w32(0xffffffffu); w32(0xffffffffu); w32(0);   // next_refID, next_pos, tlen
		wn(name.data(), name.size()); *w++ = 0;
		if (cg) { w32(((uint32_t)l_seq << 4) | 4u); w32(((uint32_t)ref_len << 4) | 3u); }
		else for (uint32_t c : cigar) w32(c);
		for (size_t x = 0; x + 1 < l_seq; x += 2) *w++ = (uint8_t)((nt16[seq[x]] << 4) | nt16[seq[x + 1]]);
		if (l_seq & 1) *w++ = (uint8_t)(nt16[seq[l_seq - 1]] << 4);
		if (D.took) { if (patches && l_seq) patches->push_back(CramQualPlan::Patch{(uint64_t)(w - out.data()), D.last_src, (uint32_t)l_seq, 0u}); D.took = false; }
		wn(qual.data(), l_seq);
		wn(tagbytes.data(), tagbytes.size());
		if (rg_id) { *w++ = 'R'; *w++ = 'G'; *w++ = 'Z'; wn(rg_id->data(), rg_id->size()); *w++ = 0; }
		if (cg)
		{
			*w++ = 'C'; *w++ = 'G'; *w++ = 'B'; *w++ = 'I'; w32((uint32_t)cigar.size());
			for (uint32_t c : cigar) w32(c);
		}
		if (w != out.data() + out.size()) throw std::logic_error("CRAM record size bookkeeping");
	}
	// ---- mates (htslib cram_decode_slice_xref) ----
	for (size_t i = 0; i < nrec; ++i)
	{
		RecInfo& r = recs[i];
		if (r.cf & CF_DETACHED)
		{
			r.mate_ref = r.ns; r.mate_pos = r.np; r.tlen = r.ts; r.tlen_set = true;
			if (r.mf & 1) r.bf |= BAM_FMREVERSE;
			if (r.mf & 2) r.bf |= BAM_FMUNMAP;
			continue;
		}
		if (r.mate_line < 0) continue;
		if (!r.tlen_set)
		{
			int64_t left = r.pos, right = r.end; int left_cnt = 0; int32_t ref = r.ref_id; size_t j = i; std::vector<size_t> chain;
			for (;;)
			{
				RecInfo& m = recs[j]; chain.push_back(j);
				if (m.pos < left) { left = m.pos; left_cnt = 1; } else if (m.pos == left) ++left_cnt;
				if (m.end > right) right = m.end;
				if (m.ref_id != ref) ref = -1;
				if (m.mate_line == -1) { m.mate_line = (int32_t)i; break; }
				if ((size_t)m.mate_line <= j || (size_t)m.mate_line >= nrec) throw CramError("bad CRAM mate chain");
				j = (size_t)m.mate_line;
			}
			const int64_t tlen = right - left + 1;
			for (size_t x : chain)
			{
				RecInfo& m = recs[x]; m.tlen_set = true;
				if (ref == -1) m.tlen = 0;
				else if (m.pos == left && (left_cnt == 1 || (m.bf & BAM_FREAD1))) m.tlen = tlen;
				else m.tlen = -tlen;
			}
		}
		const RecInfo& mate = recs[(size_t)r.mate_line];
		r.mate_ref = mate.ref_id; r.mate_pos = mate.pos;
		r.bf |= BAM_FPAIRED;
		if (mate.bf & BAM_FUNMAP) { r.bf |= BAM_FMUNMAP; r.tlen = 0; }
		if (r.bf & BAM_FUNMAP) r.tlen = 0;
		if (mate.bf & BAM_FREVERSE) r.bf |= BAM_FMREVERSE;
	}
	for (const RecInfo& r : recs)
	{
		out[r.off + 18] = (uint8_t)r.bf; out[r.off + 19] = (uint8_t)(r.bf >> 8);
		put32(out, r.off + 24, (uint32_t)r.mate_ref); put32(out, r.off + 28, (uint32_t)(r.mate_pos - 1)); put32(out, r.off + 32, (uint32_t)(int32_t)r.tlen);
	}
}

// fn(i) for i in [0, n) on up to `threads` host threads (exceptions: the first one is rethrown)
template <class F> void parallel_for(size_t n, int threads, F fn)
{
	threads = (int)std::min<size_t>((size_t)std::max(threads, 1), std::max<size_t>(n, 1));
	if (threads <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
	std::atomic<size_t> next(0); std::mutex mu; std::exception_ptr err;
	auto work = [&] { for (;;) { const size_t i = next.fetch_add(1); if (i >= n) return; try { fn(i); } catch (...) { std::lock_guard<std::mutex> g(mu); if (!err) err = std::current_exception(); return; } } };
	std::vector<std::thread> pool;
	for (int t = 1; t < threads; ++t) pool.emplace_back(work);
	work();
	for (auto& t : pool) t.join();
	if (err) std::rethrow_exception(err);
}
int host_threads()
{
	int nt = (int)std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 32);
	if (const char* et = getenv("NGSQC_CRAM_THREADS")) nt = std::max(1, atoi(et));
	return nt;
}

struct SliceJob
{
	const CompHdr* ch = nullptr; SliceHdr sh; size_t blocks_at = 0; std::vector<uint8_t> out;
	bool has_q = false; CramQualPlan::Job qjob; std::vector<uint16_t> qtabs; std::vector<uint8_t> qsyms; std::vector<CramQualPlan::Patch> patches;   // the slice's quality block, left to the device
};

std::mutex g_ref_mu; std::string g_reference;
} // namespace

void cram_set_reference(const char* fasta) { std::lock_guard<std::mutex> g(g_ref_mu); g_reference = fasta ? fasta : ""; }
std::string cram_reference()
{
	{ std::lock_guard<std::mutex> g(g_ref_mu); if (!g_reference.empty()) return g_reference; }
	const char* e = getenv("NGSQC_REFERENCE");
	return e ? e : "";
}
bool is_cram(const uint8_t* d, size_t n) { return n >= 4 && memcmp(d, "CRAM", 4) == 0; }

namespace {
// the whole CRAM as an uncompressed BAM stream ("BAM\1", header, records in file order). Throws FormatError / IoError / std::domain_error.
void bgzf_store_pieces(const std::vector<std::pair<const uint8_t*, size_t>>& pieces, ByteImage& image);

void cram_to_bam_stream_impl(const uint8_t* d, size_t n, const std::string& path, ByteImage& image, const CramSelect* sel, CramQualPlan* defer)
{
	std::vector<uint8_t> stream;   // (the BAM header only: the records stay in the slices' buffers until they are framed)
	try
	{
		const auto t0 = std::chrono::steady_clock::now();
		auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
		if (n < 26 || memcmp(d, "CRAM", 4) != 0) throw CramError("not a CRAM file");
		if (d[4] != 3 || d[5] > 1) throw std::domain_error("CRAM " + std::to_string(d[4]) + "." + std::to_string(d[5]) + " input is not supported by the HIP path (CRAM 3.0 and 3.1 only)");
		Cur c(d, n, 26);
		// ---- the SAM header ----
		ContainerHdr k; read_container_header(c, k);
		size_t end = c.p + (size_t)k.length;
		if (end > n) throw CramError("truncated CRAM data");
		std::string text;
		{
			Blk b; read_block(c, b);
			if (b.ctype != 0 || b.n < 4) throw CramError("the first container does not hold the SAM header");
			const int32_t l_text = (int32_t)((uint32_t)b.p[0] | ((uint32_t)b.p[1] << 8) | ((uint32_t)b.p[2] << 16) | ((uint32_t)b.p[3] << 24));
			if (l_text < 0 || (size_t)l_text + 4 > b.n) throw CramError("damaged SAM header of the CRAM file");
			text.assign((const char*)b.p + 4, (size_t)l_text);
		}
		c.p = end;
		std::vector<std::string> ref_names, rg_ids; std::vector<int64_t> ref_lens;
		{
			std::istringstream is(text); std::string line;
			while (std::getline(is, line))
			{
				const bool sq = line.compare(0, 3, "@SQ") == 0, rg = line.compare(0, 3, "@RG") == 0;
				if (!sq && !rg) continue;
				std::string nm; int64_t ln = 0; size_t o = 3;
				while (o < line.size())
				{
					const size_t e2 = line.find('\t', o + 1); const std::string f = line.substr(o + 1, (e2 == std::string::npos ? line.size() : e2) - o - 1);
					if (sq && f.compare(0, 3, "SN:") == 0) nm = f.substr(3);
					if (sq && f.compare(0, 3, "LN:") == 0) ln = atoll(f.c_str() + 3);
					if (rg && f.compare(0, 3, "ID:") == 0) nm = f.substr(3);
					if (e2 == std::string::npos) break;
					o = e2;
				}
				if (sq) { ref_names.push_back(nm); ref_lens.push_back(ln); } else rg_ids.push_back(nm);
			}
		}
		// ---- containers -> slice jobs ----
		// a selection (index-driven requests): named regions -> reference ids (names as in the header, with or without "chr", like the BAM path); a slice is kept when
		// its own header says it can hold a record of a region (multi-reference slices always) - the slice headers are read, the other blocks only skipped, so no
		// .crai is needed to find them. A region on a sequence the file does not have selects nothing.
		std::vector<ngsqc_region> want;
		if (sel) for (const CramSelect::Region& g : sel->regions)
		{
			int32_t tid = -1;
			for (size_t i = 0; i < ref_names.size() && tid < 0; ++i) if (ref_names[i] == g.chr) tid = (int32_t)i;
			for (size_t i = 0; i < ref_names.size() && tid < 0; ++i) if (chr_norm(ref_names[i]) == chr_norm(g.chr)) tid = (int32_t)i;   // (the same rule as the BAM index path, api.hip open_range_common)
			if (tid >= 0) want.push_back(ngsqc_region{tid, g.start, g.end});
		}
		const bool by_region = sel && !sel->regions.empty(); size_t n_seen = 0;
		std::vector<std::unique_ptr<CompHdr>> headers; std::vector<SliceJob> jobs; std::vector<Blk> blocks; bool any_rr = false, eof = false;
		while (c.p < n)
		{
			read_container_header(c, k); end = c.p + (size_t)k.length;
			if (end > n) throw CramError("truncated CRAM data");
			if (k.n_records == 0 && k.ref_id == -1 && k.start == 4542278) { eof = true; c.p = end; continue; }
			if (k.n_blocks == 0) { c.p = end; continue; }
			Blk hb; read_block(c, hb);
			if (hb.ctype != 1) throw CramError("compression header expected");
			headers.emplace_back(new CompHdr()); read_compression_header(hb.p, hb.n, *headers.back());
			any_rr = any_rr || headers.back()->RR;
			while (c.p < end)
			{
				Blk sb; read_block(c, sb);
				if (sb.ctype != 2) throw CramError("slice header expected");
				SliceJob j; j.ch = headers.back().get(); read_slice_header(sb.p, sb.n, j.sh); j.blocks_at = c.p;
				// (the blocks are inflated by the slice's worker: skip over them here)
				for (int32_t b = 0; b < j.sh.n_blocks; ++b) { c.byte(); c.byte(); c.itf8(); const int32_t cs = c.itf8(); c.itf8(); if (cs < 0) throw CramError("bad CRAM block sizes"); c.take((size_t)cs + 4); }
				++n_seen;
				bool keep = true;
				if (by_region)
				{
					keep = j.sh.ref_id == -2;
					if (j.sh.ref_id >= 0) for (const ngsqc_region& g : want) keep = keep || (g.tid == j.sh.ref_id && (int64_t)g.start <= (int64_t)j.sh.start + j.sh.span - 1 && (int64_t)g.end >= j.sh.start);
				}
				if (sel && sel->max_slices > 0 && n_seen > (size_t)sel->max_slices) keep = false;
				if (keep) jobs.push_back(std::move(j));
			}
			c.p = end;
			// a head request (BamReader::info re-opens with a head four times as long each round) has what it asked for: the rest of the file is not walked
			if (sel && sel->max_slices > 0 && n_seen >= (size_t)sel->max_slices && !by_region) break;
		}
		(void)eof;   // (htslib warns about a missing EOF container and goes on)
		// ---- the genome ----
		DecodeEnv env; env.ref_names = &ref_names; env.rg_ids = &rg_ids; env.path = path;
		const char* e1 = getenv("NGSQC_CRAM_NO_REFERENCE"); env.no_reference = e1 && atoi(e1) != 0;
		const char* e2 = getenv("NGSQC_CRAM_IGNORE_MD5"); env.ignore_md5 = e2 && atoi(e2) != 0;
		RefGenome genome;
		bool need_genome = false;   // a slice of mapped reads whose bases are neither all in the file (RR = false) nor in an embedded reference block
		for (const SliceJob& j : jobs) need_genome = need_genome || (j.ch->RR && j.sh.embedded_ref < 0 && j.sh.ref_id != -1);
		if (any_rr && need_genome && !env.no_reference)
		{
			const std::string fasta = cram_reference(); std::string err;
			if (fasta.empty() || !genome.open(fasta, err)) throw IoError("Error while setting reference genome '" + fasta + "'for cram file " + path);   // BamReader.cpp:486-489
			// checkChromosomeLengths (BamReader.cpp:491): the genome must hold the file's sequences at their lengths
			for (size_t i = 0; i < ref_names.size(); ++i)
			{
				const int64_t l = genome.length(ref_names[i]);
				if (l >= 0 && l != ref_lens[i]) throw FormatError("The length of chromosome '" + ref_names[i] + "' in the reference genome '" + fasta + "' differs from the length in the BAM/CRAM file '" + path + "'");
			}
			env.genome = &genome;
		}
		// (the plan of the device quality decoder holds 24 bytes per record: above 10^8 records - 2.4 GB - the blocks stay on the host until the plan has a per-block form)
		{ int64_t n_records = 0; for (const SliceJob& j : jobs) n_records += j.sh.n_records; if (n_records > 100000000ll) defer = nullptr; }
		const double t_parse = since();
		// ---- slices in parallel ----
		const int nthreads = (int)std::min<size_t>((size_t)host_threads(), std::max<size_t>(jobs.size(), 1));
		std::atomic<size_t> next(0); std::mutex err_mu; std::exception_ptr first_err;
		std::atomic<long long> us_blocks(0), us_records(0);   // (summed over the workers: NGSQC_TIMING)
		auto now_us = [] { return (long long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		auto work = [&] {
			for (;;)
			{
				const size_t i = next.fetch_add(1);
				if (i >= jobs.size()) return;
				{ std::lock_guard<std::mutex> g(err_mu); if (first_err) return; }
				try
				{
					SliceJob& j = jobs[i]; Cur bc(d, n, j.blocks_at);
					std::vector<Blk> bl((size_t)j.sh.n_blocks);
					const long long tb0 = now_us();
					for (Blk& b : bl) read_block(bc, b, defer ? j.ch->qs_only_id : -1, j.ch->skip_ids.empty() ? nullptr : &j.ch->skip_ids);
					auto on_host = [&](Blk& b) { rans_decode(b.raw, b.raw_n, b.own, b.n); if (b.own.size() != b.n) throw CramError("CRAM block inflates to another size than its header says"); b.p = b.own.data(); b.lazy = false; };
					for (Blk& b : bl) if (b.lazy) { if (rans_plan(b.raw, b.raw_n, (uint64_t)(b.raw - d), b.n, j.qjob, j.qtabs, j.qsyms)) j.has_q = true; else on_host(b); }
					const long long tb1 = now_us(); us_blocks += tb1 - tb0;
					try { decode_slice(*j.ch, j.sh, bl, env, j.out, j.has_q ? &j.patches : nullptr); }
					catch (NeedHostQuals&)
					{
						// a record takes single qualities out of the block (features without a quality array): this slice's block is decoded here after all
						for (Blk& b : bl) if (b.lazy) on_host(b);
						j.has_q = false; j.out.clear(); j.patches.clear();
						decode_slice(*j.ch, j.sh, bl, env, j.out, nullptr);
					}
					us_records += now_us() - tb1;
				}
				catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!first_err) first_err = std::current_exception(); return; }
			}
		};
		std::vector<std::thread> pool;
		for (int t = 1; t < nthreads; ++t) pool.emplace_back(work);
		work();
		for (auto& t : pool) t.join();
		if (first_err) std::rethrow_exception(first_err);
		const double t_decode = since();
		// ---- BAM header + records ----
		size_t total = 12 + text.size();
		for (size_t i = 0; i < ref_names.size(); ++i) total += 9 + ref_names[i].size();
		for (const SliceJob& j : jobs) total += j.out.size();
		(void)total;
		stream.insert(stream.end(), {'B', 'A', 'M', 1}); add32(stream, (uint32_t)text.size()); stream.insert(stream.end(), text.begin(), text.end());
		add32(stream, (uint32_t)ref_names.size());
		for (size_t i = 0; i < ref_names.size(); ++i)
		{
			add32(stream, (uint32_t)ref_names[i].size() + 1); stream.insert(stream.end(), ref_names[i].begin(), ref_names[i].end()); stream.push_back(0);
			add32(stream, (uint32_t)ref_lens[i]);
		}
		size_t n_q = 0; uint64_t at = stream.size(); std::vector<uint64_t> base_of(jobs.size());
		std::vector<std::pair<const uint8_t*, size_t>> pieces; pieces.emplace_back(stream.data(), stream.size());
		for (size_t ji = 0; ji < jobs.size(); ++ji) { base_of[ji] = at; at += jobs[ji].out.size(); pieces.emplace_back(jobs[ji].out.data(), jobs[ji].out.size()); }
		const size_t stream_bytes = (size_t)at;
		bgzf_store_pieces(pieces, image);   // (header + the slices' records, cut into stored BGZF members - in parallel, straight out of the slices' buffers)
		for (SliceJob& j : jobs) std::vector<uint8_t>().swap(j.out);
		for (size_t ji = 0; ji < jobs.size(); ++ji)
		{
			SliceJob& j = jobs[ji];
			if (j.has_q && defer)
			{
				const uint64_t base = base_of[ji];
				CramQualPlan::Job q = j.qjob; q.out_off = defer->out_bytes; q.tab_off = (uint32_t)defer->tabs.size(); q.sym_off = (uint32_t)defer->syms.size();
				defer->jobs.push_back(q); defer->tabs.insert(defer->tabs.end(), j.qtabs.begin(), j.qtabs.end()); defer->syms.insert(defer->syms.end(), j.qsyms.begin(), j.qsyms.end());
				for (CramQualPlan::Patch pt : j.patches) { pt.dst += base; pt.src += q.out_off; defer->patches.push_back(pt); }
				defer->out_bytes += q.n_out; ++n_q;
			}
		}
		if (getenv("NGSQC_TIMING"))
			fprintf(stderr, "[ngsqc] cram: %zu slices on %d host threads: structure %.1f ms, blocks + records %.1f ms, BAM stream of %zu bytes %.1f ms; quality blocks left to the device: %zu (%llu bytes); summed over the threads: CRC + block codecs %.1f ms, records %.1f ms\n",
			        jobs.size(), nthreads, t_parse, t_decode - t_parse, stream_bytes, since() - t_decode, n_q, defer ? (unsigned long long)defer->out_bytes : 0ull, (double)us_blocks.load() / 1e3, (double)us_records.load() / 1e3);
	}
	catch (CramError& e) { throw FormatError("Could not read next alignment in BAM/CRAM file " + path + " (" + e.what() + ")"); }
}
} // namespace

// NGSQC_OK or NGSQC_E_FORMAT / NGSQC_E_IO / NGSQC_E_UNSUPPORTED / NGSQC_E_DEVICE with the message in err
int cram_to_bam_image(const uint8_t* d, size_t n, const std::string& path, ByteImage& image, std::string& err, const CramSelect* sel, CramQualPlan* defer)
{
	if (defer) *defer = CramQualPlan();
	try { cram_to_bam_stream_impl(d, n, path, image, sel, defer); return NGSQC_OK; }
	catch (FormatError& e) { err = e.what(); return NGSQC_E_FORMAT; }
	catch (IoError& e) { err = e.what(); return NGSQC_E_IO; }
	catch (std::domain_error& e) { err = e.what(); return NGSQC_E_UNSUPPORTED; }
	catch (std::exception& e) { err = e.what(); return NGSQC_E_DEVICE; }
}

// the stream in BGZF members with STORED deflate blocks (RFC 1951 3.2.4) and the EOF member: what K1's stored-block path copies on the device
namespace {
// the concatenation of the pieces in BGZF members with STORED deflate blocks (RFC 1951 3.2.4) and the EOF member: what K1's stored-block path copies on the device.
// Member m holds stream bytes [m * 65280, ...): 18 bytes of header, 5 of the stored block, the bytes, CRC-32 and size - every member at a known place, filled in parallel.
void bgzf_store_pieces(const std::vector<std::pair<const uint8_t*, size_t>>& pieces, ByteImage& image)
{
	std::vector<uint64_t> start(pieces.size() + 1, 0);
	for (size_t i = 0; i < pieces.size(); ++i) start[i + 1] = start[i] + pieces[i].second;
	const uint64_t total = start.back(); const size_t piece = 0xff00, nm = (size_t)((total + piece - 1) / piece);
	image.make((size_t)total + nm * 31 + 28);
	parallel_for(nm, host_threads(), [&](size_t m) {
		const uint64_t s0 = (uint64_t)m * piece; const size_t n = (size_t)std::min<uint64_t>(piece, total - s0);
		uint8_t* o = image.data() + m * (piece + 31);
		const uint32_t bsize = (uint32_t)(n + 5 + 25);
		const uint8_t h[23] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 255u), (uint8_t)(bsize >> 8), 1, (uint8_t)(n & 255), (uint8_t)(n >> 8), (uint8_t)(~n & 255), (uint8_t)((~n >> 8) & 255)};
		memcpy(o, h, 23);
		size_t pi = (size_t)(std::upper_bound(start.begin(), start.end(), s0) - start.begin()) - 1, done = 0;   // the piece that holds stream byte s0
		while (done < n)
		{
			while (pieces[pi].second == 0 || s0 + done >= start[pi + 1]) ++pi;
			const uint64_t in = s0 + done - start[pi]; const size_t k = (size_t)std::min<uint64_t>(n - done, pieces[pi].second - in);
			memcpy(o + 23 + done, pieces[pi].first + in, k); done += k;
		}
		const uint32_t crc = crc_of(o + 23, n), isize = (uint32_t)n;
		memcpy(o + 23 + n, &crc, 4); memcpy(o + 27 + n, &isize, 4);
	});
	static const uint8_t eof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	memcpy(image.data() + image.size() - 28, eof, 28);
}
} // namespace

} // namespace ngsqc
