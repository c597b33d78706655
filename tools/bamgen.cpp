// Synthetic BAM generator (test / bench input only — not part of the product path, not part of the oracle).
// Produces a coordinate-sorted, BGZF-compressed (zlib) BAM shaped like the configs in SURVEY.md §8(d):
//   short-read WGS: 2x150 bp paired-end, fragment ~N(400,90), 30x depth, flags/MAPQ/CIGAR mix as specified there
//   long-read (ONT-like): log-normal read lengths, one CIGAR op per ~12 bp, CG:B,I tag when > 65535 ops
// Deterministic for a given (seed, n_reads, mode): work is cut into fixed-size chunks with their own RNG streams and
// compressed by a thread pool; the output does not depend on the thread count.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <zlib.h>

namespace {

struct Rng { uint64_t s; explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) { next(); next(); }
	uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s * 0x2545F4914F6CDD1Dull; }
	double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
	uint32_t below(uint32_t n) { return (uint32_t)(uni() * n); }
	double normal() { double u = std::max(uni(), 1e-300), v = uni(); return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v); }
};

const char* HG38_NAMES[25] = {"chr1","chr2","chr3","chr4","chr5","chr6","chr7","chr8","chr9","chr10","chr11","chr12","chr13","chr14","chr15","chr16","chr17","chr18","chr19","chr20","chr21","chr22","chrX","chrY","chrM"};
const int64_t HG38_LENS[25] = {248956422,242193529,198295559,190214555,181538259,170805979,159345973,145138636,138394717,133797422,135086622,133275309,114364328,107043718,101991189,90338345,83257441,80373285,58617616,64444167,46709983,50818468,156040895,57227415,16569};

struct Params
{
	int64_t n_reads; uint64_t seed; int mode;       // mode 0 = short-read WGS, 1 = long-read
	double depth; int first_contig; int level; int aligned; int threads; int64_t start_pos;
	int flavor = 0;   // bit 0: SEQ from a synthetic reference genome (overlapping reads share sequence, 0.5 % mismatches); bits 1-2: quality model 0 = 4 levels, 1 = 8 levels (NovaSeq-style bins), 2 = 40 levels (HiSeq-style decay)
};

void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x & 255); v.push_back((x >> 8) & 255); v.push_back((x >> 16) & 255); v.push_back((x >> 24) & 255); }
void put16(std::vector<uint8_t>& v, uint16_t x) { v.push_back(x & 255); v.push_back((x >> 8) & 255); }

// one BGZF member. The z_stream is kept per thread and reset per member (deflateInit2 allocates ~260 KB: with a fresh stream per
// member the generator spent most of its time in mmap / page faults); deflateReset gives the same bytes as a fresh stream.
struct ZCache { z_stream zs; int level = -99; bool live = false; ~ZCache() { if (live) deflateEnd(&zs); } };
void bgzf_block(const uint8_t* data, size_t n, int level, std::vector<uint8_t>& out)
{
	static thread_local ZCache zc;
	static thread_local uint8_t buf[70000];
	z_stream& zs = zc.zs;
	if (!zc.live || zc.level != level)
	{
		if (zc.live) deflateEnd(&zs);
		memset(&zs, 0, sizeof(zs));
		deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
		zc.live = true; zc.level = level;
	}
	else deflateReset(&zs);
	zs.next_in = const_cast<uint8_t*>(data); zs.avail_in = (uInt)n; zs.next_out = buf; zs.avail_out = sizeof(buf);
	deflate(&zs, Z_FINISH);
	size_t clen = zs.total_out;
	static const uint8_t hdr[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
	out.insert(out.end(), hdr, hdr + 12);
	out.push_back('B'); out.push_back('C'); put16(out, 2); put16(out, (uint16_t)(clen + 25));
	out.insert(out.end(), buf, buf + clen);
	put32(out, (uint32_t)crc32(crc32(0, nullptr, 0), data, (uInt)n)); put32(out, (uint32_t)n);
}

// BGZF writer with htslib's behaviour (aligned=1: a record never straddles members unless larger than a member)
struct Bgzf
{
	std::vector<uint8_t> out, cur; int level, aligned;
	static constexpr size_t BLOCK = 0xff00;
	Bgzf(int lvl, int al) : level(lvl), aligned(al) { cur.reserve(BLOCK); }
	void flush() { if (!cur.empty()) { bgzf_block(cur.data(), cur.size(), level, out); cur.clear(); } }
	void write(const uint8_t* p, size_t n, bool record_start)
	{
		if (aligned && record_start && cur.size() + n > BLOCK) flush();
		while (n) { size_t k = std::min(n, BLOCK - cur.size()); cur.insert(cur.end(), p, p + k); p += k; n -= k; if (cur.size() == BLOCK) flush(); }
	}
};

const int QLEVELS[4] = {2, 12, 23, 37};

void add_aux_common(std::vector<uint8_t>& r, Rng& g, int nm, int len)
{
	r.push_back('N'); r.push_back('M'); r.push_back('C'); r.push_back((uint8_t)nm);
	r.push_back('A'); r.push_back('S'); r.push_back('C'); r.push_back((uint8_t)std::min(255, std::max(0, len - 5 * nm)));
	r.push_back('X'); r.push_back('S'); r.push_back('C'); r.push_back((uint8_t)g.below(100));
	char md[32]; int k = snprintf(md, sizeof(md), "%d", len);
	r.push_back('M'); r.push_back('D'); r.push_back('Z'); r.insert(r.end(), md, md + k + 1);
	const char* rg = "sample1_L001";
	r.push_back('R'); r.push_back('G'); r.push_back('Z'); r.insert(r.end(), rg, rg + strlen(rg) + 1);
}

void finish_record(std::vector<uint8_t>& r) { uint32_t bs = (uint32_t)r.size() - 4; r[0] = bs & 255; r[1] = (bs >> 8) & 255; r[2] = (bs >> 16) & 255; r[3] = (bs >> 24) & 255; }

void core(std::vector<uint8_t>& r, int32_t tid, int32_t pos, uint8_t l_name, uint8_t mapq, uint16_t n_cigar, uint16_t flag, int32_t l_seq, int32_t mtid, int32_t mpos, int32_t isize)
{
	put32(r, 0); put32(r, (uint32_t)tid); put32(r, (uint32_t)pos);
	r.push_back(l_name); r.push_back(mapq); put16(r, 4680); put16(r, n_cigar); put16(r, flag);
	put32(r, (uint32_t)l_seq); put32(r, (uint32_t)mtid); put32(r, (uint32_t)mpos); put32(r, (uint32_t)isize);
}

// base of the synthetic reference at (tid, pos): a pure function, so that reads that overlap on the genome carry the same sequence without a stored genome
inline uint32_t ref_base(int32_t tid, int64_t pos) { uint64_t x = ((uint64_t)(uint32_t)tid << 40) ^ (uint64_t)pos; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return (uint32_t)(x & 3); }

void seq_qual(std::vector<uint8_t>& r, Rng& g, int len, int flavor = 0, int32_t tid = 0, int64_t pos = 0)
{
	static const uint8_t B[4] = {1, 2, 4, 8};
	if (flavor & 1)
	{
		// the read's bases follow the reference from its position on (CIGAR details are ignored: what matters here is the redundancy between overlapping reads)
		for (int i = 0; i < (len + 1) / 2; ++i)
		{
			uint32_t b0 = ref_base(tid, pos + 2 * i), b1 = ref_base(tid, pos + 2 * i + 1);
			const uint64_t x = g.next();
			if ((x & 255) == 0) b0 = (uint32_t)(x >> 8) & 3;        // ~0.4 % sequencing errors / variants per base
			if (((x >> 16) & 255) == 0) b1 = (uint32_t)(x >> 24) & 3;
			r.push_back((uint8_t)((B[b0] << 4) | B[b1]));
		}
	}
	else for (int i = 0; i < (len + 1) / 2; ++i) { uint64_t x = g.next(); r.push_back((uint8_t)((B[x & 3] << 4) | B[(x >> 2) & 3])); }
	const int qm = (flavor >> 1) & 3;
	if (qm == 0)
	{
		int q = 3;
		for (int i = 0; i < len; ++i) { uint64_t x = g.next(); if ((x & 15) == 0) q = (int)((x >> 4) & 3); else if ((x & 15) == 1 && i > len * 3 / 4) q = (int)((x >> 4) % 3); r.push_back((uint8_t)QLEVELS[q]); }
	}
	else if (qm == 1)
	{
		// eight quality bins (RTA3-style), long runs of the top bin, dips towards the read end
		static const int Q8[8] = {2, 10, 15, 20, 25, 30, 35, 40};
		int q = 7;
		for (int i = 0; i < len; ++i)
		{
			const uint64_t x = g.next(); const uint32_t u = (uint32_t)(x & 63);
			if (u < 3) q = 7 - (int)((x >> 8) % (i > len * 2 / 3 ? 6 : 3)); else if (u < 9) q = std::min(7, q + 1);
			r.push_back((uint8_t)Q8[q]);
		}
	}
	else
	{
		// forty levels: a slowly decaying mean with per-base noise (HiSeq 2000-style, un-binned)
		double mean = 38.0;
		for (int i = 0; i < len; ++i)
		{
			mean -= 0.05 + 0.0004 * i;
			const uint64_t x = g.next(); const int noise = (int)(x & 7) - 3 + (((x >> 8) & 31) == 0 ? -(int)((x >> 16) & 15) : 0);
			r.push_back((uint8_t)std::min(41, std::max(2, (int)std::lround(mean) + noise)));
		}
	}
}

uint8_t draw_mapq(Rng& g) { double u = g.uni(); return u < 0.85 ? 60 : (u < 0.91 ? 0 : (uint8_t)(1 + g.below(59))); }

// short-read record at (tid,pos)
void short_read(std::vector<uint8_t>& r, Rng& g, int32_t tid, int32_t pos, int64_t contig_len, uint64_t serial, int flavor = 0)
{
	r.clear();
	int len = g.uni() < 0.05 ? 50 + (int)g.below(100) : 150;
	int frag = (int)std::lround(400 + 90 * g.normal()); frag = std::min(900, std::max(150, frag));
	uint16_t flag = 0x1; bool read1 = g.next() & 1; flag |= read1 ? 0x40 : 0x80;
	bool fwd = g.next() & 1; flag |= fwd ? 0x20 : 0x10;
	double u = g.uni();
	bool unmapped = u < 0.005, secondary = !unmapped && u < 0.008, supp = !unmapped && !secondary && u < 0.011;
	if (g.uni() < 0.97 && !unmapped) flag |= 0x2;
	if (g.uni() < 0.08 && !unmapped) flag |= 0x400;
	if (unmapped) flag |= 0x4; if (secondary) flag |= 0x100; if (supp) flag |= 0x800;
	uint8_t mapq = unmapped ? 0 : draw_mapq(g);
	char name[48]; int nl = snprintf(name, sizeof(name), "A00%03u:%u:H%05XDSXY:%u:%04u:%05u:%05u", (unsigned)(serial % 7) + 100, 45u + (unsigned)(serial % 3), (unsigned)((serial >> 20) & 0xFFFFF), 1 + (unsigned)(serial % 4), 1101 + (unsigned)g.below(1578), (unsigned)g.below(32000), (unsigned)g.below(32000));
	// CIGAR
	uint32_t cig[6]; int nc = 0; int ref_len = len; int nm = (int)g.below(3);
	if (!unmapped)
	{
		double c = g.uni();
		if (supp) { int h = 20 + (int)g.below(80); h = std::min(h, len - 20); cig[nc++] = ((uint32_t)h << 4) | 5; cig[nc++] = ((uint32_t)len << 4) | 0; }
		else if (c < 0.93) cig[nc++] = ((uint32_t)len << 4) | 0;
		else if (c < 0.97) { int s = 1 + (int)g.below(60); s = std::min(s, len - 10); if (g.next() & 1) { cig[nc++] = ((uint32_t)s << 4) | 4; cig[nc++] = ((uint32_t)(len - s) << 4) | 0; } else { cig[nc++] = ((uint32_t)(len - s) << 4) | 0; cig[nc++] = ((uint32_t)s << 4) | 4; } ref_len = len - s; }
		else
		{
			int il = 1; while (g.uni() > 0.4 && il < 20) ++il;
			int a = 10 + (int)g.below((uint32_t)(len - 30));
			bool ins = g.next() & 1;
			if (c < 0.99)
			{
				if (ins) { cig[nc++] = ((uint32_t)a << 4); cig[nc++] = ((uint32_t)il << 4) | 1; cig[nc++] = ((uint32_t)(len - a - il) << 4); ref_len = len - il; }
				else { cig[nc++] = ((uint32_t)a << 4); cig[nc++] = ((uint32_t)il << 4) | 2; cig[nc++] = ((uint32_t)(len - a) << 4); ref_len = len + il; }
			}
			else
			{
				int s = 1 + (int)g.below(20); int m = len - s; a = std::min(a, m - il - 5); if (a < 1) a = 1;
				cig[nc++] = ((uint32_t)s << 4) | 4; cig[nc++] = ((uint32_t)a << 4); cig[nc++] = ((uint32_t)il << 4) | 2; cig[nc++] = ((uint32_t)(m - a) << 4); ref_len = m + il;
			}
			nm += il;
		}
	}
	if (pos + ref_len > contig_len) pos = (int32_t)std::max<int64_t>(0, contig_len - ref_len);
	int32_t mpos = fwd ? pos + frag - len : pos - (frag - len); if (mpos < 0) mpos = 0;
	int32_t isize = (flag & 0x2) ? (fwd ? frag : -frag) : (g.uni() < 0.5 ? 0 : (int32_t)(g.below(200000)) - 100000);
	core(r, tid, pos, (uint8_t)(nl + 1), mapq, (uint16_t)nc, flag, len, tid, mpos, isize);
	r.insert(r.end(), name, name + nl + 1);
	for (int i = 0; i < nc; ++i) put32(r, cig[i]);
	seq_qual(r, g, len, flavor, tid, pos);
	add_aux_common(r, g, nm, len);
	finish_record(r);
}

// long-read record: many small ops (=, X, I, D), soft clips, CG tag when > 65535 ops
int long_read(std::vector<uint8_t>& r, Rng& g, int32_t tid, int32_t pos, int64_t contig_len, uint64_t serial)
{
	r.clear();
	double l = std::exp(9.6 + 0.75 * g.normal()); int len = (int)std::min(500000.0, std::max(500.0, l));
	uint16_t flag = (g.next() & 1) ? 0x10 : 0;
	double u = g.uni(); bool unmapped = u < 0.005, secondary = !unmapped && u < 0.008, supp = !unmapped && !secondary && u < 0.03;
	if (unmapped) flag |= 0x4; if (secondary) flag |= 0x100; if (supp) flag |= 0x800;
	uint8_t mapq = unmapped ? 0 : draw_mapq(g);
	std::vector<uint32_t> cig; int64_t ref_len = 0; int q = 0;
	if (!unmapped)
	{
		int s1 = (int)g.below(60), s2 = (int)g.below(60); int rem = len - s1 - s2;
		if (s1) { cig.push_back(((uint32_t)s1 << 4) | 4); q += s1; }
		while (rem > 0)
		{
			int m = 1 + (int)g.below(22); m = std::min(m, rem); cig.push_back(((uint32_t)m << 4) | 7); rem -= m; ref_len += m;
			if (rem <= 0) break;
			double c = g.uni();
			if (c < 0.4) { cig.push_back((1u << 4) | 8); rem -= 1; ref_len += 1; }
			else if (c < 0.7) { int k = 1 + (int)g.below(3); k = std::min(k, rem); cig.push_back(((uint32_t)k << 4) | 1); rem -= k; }
			else { int k = 1 + (int)g.below(4); cig.push_back(((uint32_t)k << 4) | 2); ref_len += k; }
		}
		if (s2) cig.push_back(((uint32_t)s2 << 4) | 4);
	}
	if (pos + ref_len > contig_len) pos = (int32_t)std::max<int64_t>(0, contig_len - ref_len);
	char name[48]; int nl = snprintf(name, sizeof(name), "%08x-%04x-%04x-%04x-%012llx", (unsigned)g.next(), (unsigned)(g.next() & 0xffff), (unsigned)(g.next() & 0xffff), (unsigned)(g.next() & 0xffff), (unsigned long long)(serial & 0xffffffffffffull));
	bool use_cg = cig.size() > 65535;
	core(r, tid, pos, (uint8_t)(nl + 1), mapq, (uint16_t)(use_cg ? 2 : cig.size()), flag, len, -1, -1, 0);
	r.insert(r.end(), name, name + nl + 1);
	if (use_cg) { put32(r, ((uint32_t)len << 4) | 4); put32(r, ((uint32_t)ref_len << 4) | 3); }
	else for (uint32_t c : cig) put32(r, c);
	seq_qual(r, g, len);
	add_aux_common(r, g, (int)g.below(200), len > 250 ? 250 : len);
	if (use_cg) { r.push_back('C'); r.push_back('G'); r.push_back('B'); r.push_back('I'); put32(r, (uint32_t)cig.size()); for (uint32_t c : cig) put32(r, c); }
	finish_record(r);
	return (int)ref_len;
}

std::vector<uint8_t> header_bytes()
{
	std::vector<uint8_t> h = {'B', 'A', 'M', 1};
	std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
	for (int i = 0; i < 25; ++i) text += std::string("@SQ\tSN:") + HG38_NAMES[i] + "\tLN:" + std::to_string(HG38_LENS[i]) + "\n";
	text += "@RG\tID:sample1_L001\tSM:sample1\tPL:ILLUMINA\n@PG\tID:bamgen\tPN:bamgen\n";
	put32(h, (uint32_t)text.size()); h.insert(h.end(), text.begin(), text.end());
	put32(h, 25);
	for (int i = 0; i < 25; ++i) { uint32_t l = (uint32_t)strlen(HG38_NAMES[i]) + 1; put32(h, l); h.insert(h.end(), HG38_NAMES[i], HG38_NAMES[i] + l); put32(h, (uint32_t)HG38_LENS[i]); }
	return h;
}

constexpr int64_t CHUNK = 32768; // reads per chunk (short); long-read mode uses CHUNK/64

// The image is assembled in one anonymous mapping sized from an upper bound (untouched pages cost nothing), so the
// caller gets it without another copy. Chunks are generated in segments of a few chunks per thread: the threads compress
// into per-slot buffers that are reused from segment to segment, then copy their slots to the final offsets in parallel
// (peak memory = the image + one segment; a full 30x WGS image is ~60 GB).
struct Image { uint8_t* p = nullptr; size_t n = 0, cap = 0; };

Image generate(const Params& P)
{
	const int64_t chunk = P.mode == 0 ? CHUNK : CHUNK / 64;
	const int64_t n_chunks = (P.n_reads + chunk - 1) / chunk;
	const double mean_len = P.mode == 0 ? 147.0 : std::exp(9.6 + 0.75 * 0.75 / 2);
	const double gap = mean_len / P.depth; // mean distance between read starts
	// chunk c covers genomic offsets [c*chunk*gap, (c+1)*chunk*gap) of the concatenated genome starting at (first_contig,start_pos)
	const int T = P.threads > 0 ? P.threads : (int)std::max(1u, std::thread::hardware_concurrency());
	std::vector<uint8_t> head;
	{ Bgzf z(P.level, 1); auto h = header_bytes(); z.write(h.data(), h.size(), false); z.flush(); head.swap(z.out); }
	static const uint8_t eof[28] = {0x1f,0x8b,0x08,0x04,0,0,0,0,0,0xff,0x06,0,0x42,0x43,0x02,0,0x1b,0,0x03,0,0,0,0,0,0,0,0,0};
	// upper bound of the image: stored-block worst case of every member (inflated size + 5 bytes per 64 KB + 26 bytes of BGZF frame)
	// on top of a generous bound of the inflated record bytes
	const double rec_bound = P.mode == 0 ? 420.0 : 1.3 * 1.6e6;   // bytes per record, far above the mean (long reads: capped lengths, ~13 bytes per 12 bp)
	size_t cap = head.size() + 28 + (size_t)((double)P.n_reads * (P.mode == 0 ? rec_bound : std::min(rec_bound, 40.0 * mean_len))) + (size_t)n_chunks * 70000 + (1u << 20);
	Image img; img.cap = cap;
	void* m = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
	if (m == MAP_FAILED) return img;
	img.p = (uint8_t*)m;
	memcpy(img.p, head.data(), head.size());
	size_t pos = head.size();
	const int64_t seg = std::max<int64_t>(1, std::min<int64_t>(n_chunks, (int64_t)T * 4));
	std::vector<std::vector<uint8_t>> slot((size_t)seg);
	std::vector<size_t> off((size_t)seg + 1);
	bool overflow = false;
	for (int64_t c0 = 0; c0 < n_chunks && !overflow; c0 += seg)
	{
		const int64_t cn = std::min(seg, n_chunks - c0);
		std::atomic<int64_t> next(0);
		auto worker = [&] {
			std::vector<uint8_t> rec;
			Bgzf z(P.level, P.aligned);
			while (true)
			{
				const int64_t k = next.fetch_add(1); if (k >= cn) break;
				const int64_t c = c0 + k;
				Rng g(P.seed * 1000003ull + (uint64_t)c);
				z.out.swap(slot[(size_t)k]); z.out.clear(); z.cur.clear();
				int64_t n = std::min(chunk, P.n_reads - c * chunk);
				double o0 = (double)c * (double)chunk * gap, span = (double)n * gap;
				// sorted offsets inside the chunk: sorted uniforms
				std::vector<double> offs((size_t)n); for (auto& o : offs) o = o0 + g.uni() * span;
				std::sort(offs.begin(), offs.end());
				for (int64_t i = 0; i < n; ++i)
				{
					int64_t o = (int64_t)offs[i] + P.start_pos; int tid = P.first_contig;
					while (tid < 24 && o >= HG38_LENS[tid]) { o -= HG38_LENS[tid]; ++tid; }
					if (o >= HG38_LENS[tid]) o = HG38_LENS[tid] - 1;
					// a read keeps its start: one that would reach behind the end of the contig starts early enough instead (moving single reads back by their own
					// reference length left the last ~150 bp of every contig out of coordinate order, which `samtools index` - and ngsqc_write_bai - refuse)
					{ const int64_t room = P.mode == 0 ? 176 : 640000; if (o > HG38_LENS[tid] - room) o = std::max<int64_t>(0, HG38_LENS[tid] - room); }
					uint64_t serial = (uint64_t)(c * chunk + i);
					if (P.mode == 0) short_read(rec, g, tid, (int32_t)o, HG38_LENS[tid], serial, P.flavor); else long_read(rec, g, tid, (int32_t)o, HG38_LENS[tid], serial);
					z.write(rec.data(), rec.size(), true);
				}
				z.flush();
				z.out.swap(slot[(size_t)k]);
			}
		};
		{ std::vector<std::thread> th; for (int t = 0; t < std::min<int64_t>(T, cn); ++t) th.emplace_back(worker); for (auto& t : th) t.join(); }
		off[0] = pos; for (int64_t k = 0; k < cn; ++k) off[(size_t)k + 1] = off[(size_t)k] + slot[(size_t)k].size();
		if (off[(size_t)cn] + 28 > cap) { overflow = true; break; }
		std::atomic<int64_t> nx(0);
		auto copier = [&] { while (true) { const int64_t k = nx.fetch_add(1); if (k >= cn) break; memcpy(img.p + off[(size_t)k], slot[(size_t)k].data(), slot[(size_t)k].size()); } };
		{ std::vector<std::thread> th; for (int t = 0; t < std::min<int64_t>(T, cn); ++t) th.emplace_back(copier); for (auto& t : th) t.join(); }
		pos = off[(size_t)cn];
	}
	if (overflow) { munmap(img.p, img.cap); img.p = nullptr; return img; }
	// NOTE: clamping a read to the contig end can move it before its predecessor by < 1 read length at contig ends;
	// records stay sorted by (tid, original offset); the QC path does not depend on strict order.
	memcpy(img.p + pos, eof, 28); pos += 28;
	img.n = pos;
	return img;
}

} // namespace

extern "C" {
// returns an anonymous mapping of *cap_out bytes whose first *n_out bytes are the BAM image (release with bamgen_release);
// mode 0 short-read WGS, 1 long-read; aligned=1 htslib-style member alignment. NULL when memory could not be mapped.
// flavor: see Params (0 = the SURVEY.md 8(d) shape: random SEQ, 4-level QUAL)
uint8_t* bamgen_generate_map2(int64_t n_reads, uint64_t seed, int mode, double depth, int first_contig, int64_t start_pos, int level, int aligned, int threads, int flavor, size_t* n_out, size_t* cap_out)
{
	Params P{n_reads, seed, mode, depth, first_contig, level, aligned, threads, start_pos}; P.flavor = flavor;
	Image img = generate(P);
	*n_out = img.n; *cap_out = img.cap;
	return img.p;
}
uint8_t* bamgen_generate_map(int64_t n_reads, uint64_t seed, int mode, double depth, int first_contig, int64_t start_pos, int level, int aligned, int threads, size_t* n_out, size_t* cap_out)
{
	Params P{n_reads, seed, mode, depth, first_contig, level, aligned, threads, start_pos};
	Image img = generate(P);
	*n_out = img.n; *cap_out = img.cap;
	return img.p;
}
void bamgen_release(uint8_t* p, size_t cap) { if (p) munmap(p, cap); }
}

#ifdef BAMGEN_MAIN
int main(int argc, char** argv)
{
	if (argc < 3) { fprintf(stderr, "usage: bamgen OUT.bam N_READS [seed=20260821] [mode=0] [depth=30] [first_contig=0] [level=6] [aligned=1]\n"); return 2; }
	size_t n = 0, cap = 0;
	uint8_t* p = bamgen_generate_map(atoll(argv[2]), argc > 3 ? strtoull(argv[3], 0, 10) : 20260821ull, argc > 4 ? atoi(argv[4]) : 0, argc > 5 ? atof(argv[5]) : 30.0,
	                                 argc > 6 ? atoi(argv[6]) : 0, 0, argc > 7 ? atoi(argv[7]) : 6, argc > 8 ? atoi(argv[8]) : 1, 0, &n, &cap);
	if (!p) return 1;
	FILE* f = fopen(argv[1], "wb"); if (!f) return 1; fwrite(p, 1, n, f); fclose(f); bamgen_release(p, cap);
	return 0;
}
#endif
