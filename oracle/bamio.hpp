// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the BGZF/BAM reading that ngs-bits gets from htslib.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/. The product (ngs-bits_amd/) never links this.
//
// htslib is an un-vendored dependency of the reference (version unpinned in-tree:
// /root/reference/.MISSING_LARGE_BLOBS:1 htslib/htslib_linux.zip). This file restates
// the published formats it implements (SAM spec v1 §4.1 BGZF, §4.2 BAM) and the
// htslib behaviours the reference relies on at its call sites:
//   sam_read1 / sam_itr_next   src/cppNGS/BamReader.h:386-398
//   bam_endpos                 src/cppNGS/BamReader.h:91-94
//   bam_aux_get / bam_aux2i    src/cppNGS/BamReader.cpp:286-297
//   sam_itr_queryi overlap     src/cppNGS/BamReader.cpp:734-768
//   CG:B,I long-CIGAR swap     (htslib bam_tag2cigar, done inside bam_read1)
// ============================================================================
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include <algorithm>
#include <zlib.h>

namespace orc {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

static inline uint16_t rd16(const uint8_t* p){ return (uint16_t)(p[0] | (p[1]<<8)); }
static inline uint32_t rd32(const uint8_t* p){ return (uint32_t)p[0] | ((uint32_t)p[1]<<8) | ((uint32_t)p[2]<<16) | ((uint32_t)p[3]<<24); }
static inline int32_t  rdi32(const uint8_t* p){ return (int32_t)rd32(p); }

static inline std::vector<uint8_t> read_file(const std::string& path)
{
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) throw Error("Could not open BAM/CRAM file " + path); // BamReader.cpp:467
	std::vector<uint8_t> buf;
	fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
	buf.resize((size_t)n);
	if (n>0 && fread(buf.data(), 1, (size_t)n, f)!=(size_t)n) { fclose(f); throw Error("Could not read file " + path); }
	fclose(f);
	return buf;
}

// One BGZF member (SAM spec §4.1): gzip header with FEXTRA subfield 'B','C' (BSIZE = total block size - 1),
// raw DEFLATE payload, CRC32, ISIZE.
struct BgzfBlock { size_t coff; uint32_t csize; size_t uoff; uint32_t usize; };

static inline std::vector<BgzfBlock> bgzf_scan(const std::vector<uint8_t>& file)
{
	std::vector<BgzfBlock> blocks;
	size_t off = 0, uoff = 0;
	while (off < file.size())
	{
		if (off + 18 > file.size()) throw Error("Truncated BGZF header");
		const uint8_t* p = file.data() + off;
		if (p[0]!=31 || p[1]!=139 || p[2]!=8 || !(p[3] & 4)) throw Error("Not a BGZF block");
		uint16_t xlen = rd16(p+10);
		uint32_t bsize = 0; bool found = false;
		size_t x = 12, xend = 12 + xlen;
		while (x + 4 <= xend)
		{
			uint16_t slen = rd16(p+x+2);
			if (p[x]=='B' && p[x+1]=='C' && slen==2) { bsize = rd16(p+x+4) + 1u; found = true; }
			x += 4 + slen;
		}
		if (!found || off + bsize > file.size()) throw Error("Invalid BGZF block");
		uint32_t isize = rd32(p + bsize - 4);
		blocks.push_back({off, bsize, uoff, isize});
		uoff += isize;
		off += bsize;
	}
	return blocks;
}

static inline void bgzf_inflate_block(const std::vector<uint8_t>& file, const BgzfBlock& b, uint8_t* out)
{
	const uint8_t* p = file.data() + b.coff;
	uint16_t xlen = rd16(p+10);
	size_t hdr = 12 + xlen;
	if (b.usize==0) return;
	z_stream zs; memset(&zs, 0, sizeof(zs));
	if (inflateInit2(&zs, -15)!=Z_OK) throw Error("inflateInit2 failed");
	zs.next_in = const_cast<uint8_t*>(p + hdr);
	zs.avail_in = (uInt)(b.csize - hdr - 8);
	zs.next_out = out;
	zs.avail_out = b.usize;
	int rc = inflate(&zs, Z_FINISH);
	inflateEnd(&zs);
	if (rc!=Z_STREAM_END || zs.total_out!=b.usize) throw Error("BGZF inflate failed");
	uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), out, b.usize);
	if (crc != rd32(p + b.csize - 8)) throw Error("BGZF CRC mismatch");
}

// A decoded view of one BAM record (SAM spec §4.2). Pointers point into the inflated stream.
struct Rec
{
	int32_t tid, pos;          // pos is 0-based
	uint8_t mapq; uint16_t flag;
	uint32_t n_cigar; int32_t l_seq; int32_t mtid, mpos, isize;
	const uint8_t* qname; uint32_t l_qname;
	const uint8_t* cigar_raw;  // u32[n_cigar] little endian (possibly the CG:B,I payload)
	const uint8_t* seq; const uint8_t* qual;
	const uint8_t* aux; size_t aux_len;
	uint32_t block_size;

	// BAM flag helpers == BamAlignment accessors (BamReader.h:128-206)
	bool isPaired() const { return flag & 0x1; }
	bool isProperPair() const { return flag & 0x2; }
	bool isUnmapped() const { return flag & 0x4; }
	bool isRead1() const { return flag & 0x40; }
	bool isSecondary() const { return flag & 0x100; }
	bool isDuplicate() const { return flag & 0x400; }
	bool isSupplementary() const { return flag & 0x800; }
	uint32_t cigarOp(uint32_t i) const { return rd32(cigar_raw + 4*i) & 0xf; }
	uint32_t cigarLen(uint32_t i) const { return rd32(cigar_raw + 4*i) >> 4; }
	int start() const { return pos + 1; }                    // BamReader.h:80-83
	int length() const { return l_seq; }                     // BamReader.h:98-115 (BAM branch)
	// htslib bam_endpos: pos + rlen, rlen = 0 if unmapped else sum of ref-consuming ops (M,D,N,=,X); rlen==0 -> 1.
	int end() const
	{
		int64_t rlen = 0;
		if (!isUnmapped())
		{
			for (uint32_t i=0; i<n_cigar; ++i)
			{
				uint32_t op = cigarOp(i);
				if (op==0 || op==2 || op==3 || op==7 || op==8) rlen += cigarLen(i);
			}
		}
		if (rlen==0) rlen = 1;
		return (int)(pos + rlen);
	}
};

// Linear aux scan == htslib bam_aux_get; returns pointer to the type byte or nullptr.
static inline const uint8_t* aux_find(const uint8_t* aux, size_t len, const char tag[2])
{
	const uint8_t* p = aux; const uint8_t* end = aux + len;
	while (p + 3 <= end)
	{
		bool hit = (p[0]==(uint8_t)tag[0] && p[1]==(uint8_t)tag[1]);
		const uint8_t* t = p + 2;
		if (hit) return t;
		uint8_t type = *t; const uint8_t* v = t + 1;
		size_t sz;
		switch (type)
		{
			case 'A': case 'c': case 'C': sz = 1; break;
			case 's': case 'S': sz = 2; break;
			case 'i': case 'I': case 'f': sz = 4; break;
			case 'd': sz = 8; break;
			case 'Z': case 'H': { const uint8_t* q = v; while (q<end && *q) ++q; sz = (size_t)(q - v) + 1; break; }
			case 'B':
			{
				if (v + 5 > end) return nullptr;
				uint8_t st = v[0]; uint32_t n = rd32(v+1);
				size_t es = (st=='c'||st=='C') ? 1 : (st=='s'||st=='S') ? 2 : 4;
				sz = 5 + es*(size_t)n; break;
			}
			default: return nullptr;
		}
		p = v + sz;
	}
	return nullptr;
}

// == BamAlignment::tagi (BamReader.cpp:286-297): 0 if absent; bam_aux2i semantics for integer types, else 0.
static inline int aux_tagi(const Rec& r, const char tag[2])
{
	const uint8_t* t = aux_find(r.aux, r.aux_len, tag);
	if (!t) return 0;
	switch (*t)
	{
		case 'c': return (int8_t)t[1];
		case 'C': return t[1];
		case 's': return (int16_t)rd16(t+1);
		case 'S': return rd16(t+1);
		case 'i': return (int32_t)rd32(t+1);
		case 'I': return (int)(int64_t)rd32(t+1);
		default: return 0;
	}
}

// Parse one record at data[off]; applies the CG:B,I long-CIGAR rule of htslib's bam_read1/bam_tag2cigar:
// if n_cigar>0, tid>=0, pos>=0, first op is S with length == l_seq, and a CG:B,I tag with >= n_cigar entries exists,
// the real CIGAR is the tag payload.
static inline Rec parse_rec(const uint8_t* d)
{
	Rec r;
	r.block_size = rd32(d);
	const uint8_t* c = d + 4;
	r.tid = rdi32(c); r.pos = rdi32(c+4);
	r.l_qname = c[8]; r.mapq = c[9];
	uint16_t n_cigar = rd16(c+12); r.flag = rd16(c+14);
	r.l_seq = rdi32(c+16); r.mtid = rdi32(c+20); r.mpos = rdi32(c+24); r.isize = rdi32(c+28);
	r.qname = c + 32;
	r.cigar_raw = r.qname + r.l_qname;
	r.n_cigar = n_cigar;
	r.seq = r.cigar_raw + 4*(size_t)n_cigar;
	r.qual = r.seq + ((size_t)r.l_seq + 1)/2;
	r.aux = r.qual + r.l_seq;
	const uint8_t* end = c + r.block_size;
	if (r.aux > end) throw Error("Corrupt BAM record");
	r.aux_len = (size_t)(end - r.aux);
	if (n_cigar>0 && r.tid>=0 && r.pos>=0)
	{
		uint32_t c0 = rd32(r.cigar_raw);
		if ((c0 & 0xf)==4 && (int32_t)(c0>>4)==r.l_seq)
		{
			const uint8_t* t = aux_find(r.aux, r.aux_len, "CG");
			if (t && t[0]=='B' && t[1]=='I')
			{
				uint32_t n = rd32(t+2);
				if (n >= n_cigar && n < (1u<<29)) { r.cigar_raw = t + 6; r.n_cigar = n; }
			}
		}
	}
	return r;
}

// Whole-file BAM loaded in memory: header + inflated record stream + per-record offsets.
struct BamFile
{
	std::string path;
	std::vector<uint8_t> data;               // inflated stream
	std::vector<std::string> ref_names; std::vector<int64_t> ref_lens;
	size_t first_rec = 0;
	std::vector<size_t> rec_off;              // offset of every record (block_size word)
	size_t n_blocks = 0; size_t csize = 0;
	// index for region queries: per tid, record ordinals in file order + prefix max of end (0-based exclusive)
	std::vector<std::vector<uint32_t>> tid_recs;
	std::vector<std::vector<int64_t>> tid_pmax_end;
	bool sorted = true;

	void load(const std::string& p)
	{
		path = p;
		std::vector<uint8_t> file = read_file(p);
		csize = file.size();
		auto blocks = bgzf_scan(file);
		n_blocks = blocks.size();
		size_t total = blocks.empty() ? 0 : blocks.back().uoff + blocks.back().usize;
		data.resize(total);
		for (auto& b : blocks) bgzf_inflate_block(file, b, data.data() + b.uoff);
		parse_header();
		index_records();
	}

	void parse_header()
	{
		if (data.size() < 12 || memcmp(data.data(), "BAM\1", 4)!=0) throw Error("Could not read header from BAM/CRAM file " + path);
		size_t o = 4; uint32_t l_text = rd32(&data[o]); o += 4 + l_text;
		uint32_t n_ref = rd32(&data[o]); o += 4;
		for (uint32_t i=0; i<n_ref; ++i)
		{
			uint32_t l_name = rd32(&data[o]); o += 4;
			ref_names.emplace_back((const char*)&data[o], l_name ? l_name-1 : 0); o += l_name;
			ref_lens.push_back(rd32(&data[o])); o += 4;
		}
		first_rec = o;
	}

	void index_records()
	{
		size_t o = first_rec;
		tid_recs.assign(ref_names.size(), {});
		tid_pmax_end.assign(ref_names.size(), {});
		int32_t last_tid = -2; int32_t last_pos = -1;
		while (o + 4 <= data.size())
		{
			uint32_t bs = rd32(&data[o]);
			if (o + 4 + bs > data.size()) throw Error("Truncated BAM record");
			Rec r = parse_rec(&data[o]);
			uint32_t ord = (uint32_t)rec_off.size();
			rec_off.push_back(o);
			if (r.tid>=0 && (size_t)r.tid<ref_names.size())
			{
				auto& pm = tid_pmax_end[r.tid];
				int64_t e = r.end();
				pm.push_back(pm.empty() ? e : std::max(pm.back(), e));
				tid_recs[r.tid].push_back(ord);
				if (last_tid>=0 && (r.tid<last_tid || (r.tid==last_tid && r.pos<last_pos))) sorted = false;
				last_tid = r.tid; last_pos = r.pos;
			}
			o += 4 + bs;
		}
	}

	size_t count() const { return rec_off.size(); }
	Rec rec(size_t i) const { return parse_rec(&data[rec_off[i]]); }

	// == BamReader::setRegion + getNextAlignment loop (BamReader.cpp:734-768): htslib iterator over
	// sam_itr_queryi(idx, tid, start-1, end): records with tid==t && pos < end && bam_endpos > start-1, in file order.
	template <class F> void forRegion(int tid, int start1, int end1, F f) const
	{
		if (tid<0 || (size_t)tid>=ref_names.size()) return;
		const auto& recs = tid_recs[tid]; const auto& pm = tid_pmax_end[tid];
		int64_t beg = (int64_t)start1 - 1, end = end1;
		size_t i = (size_t)(std::upper_bound(pm.begin(), pm.end(), beg) - pm.begin());
		for (; i<recs.size(); ++i)
		{
			Rec r = rec(recs[i]);
			if (r.pos >= end) break;
			if ((int64_t)r.end() > beg) f(r);
		}
	}
};

} // namespace orc
