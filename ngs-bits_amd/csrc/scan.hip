// K3/K4/K5 — per-record decode + classify, target-region depth scatter, counter reduction.
//
// One lane per BAM record (short reads); records with long CIGARs (or a possible CG:B,I long-CIGAR tag) are deferred to
// a wave-per-record kernel that strides the CIGAR across the 64 lanes and combines with wave reductions / prefix sums.
// Loop bodies restated from (not translated line by line):
//   Statistics::mapping(bed..)  src/cppNGS/Statistics.cpp:416-574      (MODE_ROI)
//   Statistics::mapping(bam..)  src/cppNGS/Statistics.cpp:830-916      (MODE_NOROI)
//   Statistics::mapping_wgs     src/cppNGS/Statistics.cpp:1068-1182    (MODE_WGS: pass 1 + the indexed ROI pass fused)
//   Statistics::yxRatio         src/cppNGS/Statistics.cpp:2659-2691    (two counters instead of two indexed re-reads)
//   WorkerAverageCoverage*/WorkerLowOrHighCoverage* filters and depth loops (MODE_DEPTH)
//   BamAlignment::qualities     src/cppNGS/BamReader.cpp:210-255       (min_baseq mask as sparse decrements)
// Depth is accumulated as a DIFFERENCE array (+1 at overlap start, -1 behind overlap end; one spare slot per region):
// the reference increments the whole reference span [start,end] of a read (RegionDepth::incrementRegion,
// Statistics.cpp:45-53), which is exactly a prefix sum over these differences. All arithmetic is integer.
#include "common.h"
#include <algorithm>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace ngsqc {

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint16_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
// The refresh of a per-lane cache (region search, next known site) is a rare branch that leaves loaded values in registers; it waits for them ITSELF (round 6): a
// load still pending at the join makes the compiler wait with vmcnt(0) at the values' first use behind the join - in the walk that is a wait for the next record's
// prefetched header in EVERY trip, taken or not (worth about 1 % on the 30x walk: 2.72 against 2.75 ms per 48 M records, profiles/r06_kernel_stats_serial.txt).
__device__ __forceinline__ void settle_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// per-lane partial sums (A_* in common.h), reduced per wave at the end of the kernel: one atomic per wave and counter
struct Acc
{
	// v: sums that an adversarial record can make large (lengths, clipped bases, overlaps); n: plain counts (and insert sizes below 1000) - a thread of one launch
	// sees far fewer than 2^22 records, so 32 bits hold them (round 5: nine VGPRs less in the walk). A slot is kept in ONE of the two; the other stays zero and
	// costs nothing (the compiler drops it)
	long long v[A_COUNT]; uint32_t n[A_COUNT]; int max_len; unsigned long long best_key; unsigned long long first_paired;
	// A thread meets its records in ascending position (a member's record chain; a strided range of a sorted file), so the answer of the last region search holds for
	// the next records too: lower_region(start1) == rc_idx for every start1 in (rc_lo, rc_hi] on reference rc_tid. Likewise the next known site of the pileup
	// candidates. Round 4: the walk spends its time in the address unit (every lane its own line, ~15 load instructions per record) - these two caches take the
	// region-table and site-table loads out of the common record.
	int rc_tid = -2, rc_lo = 0, rc_hi = 0, rc_idx = 0, rc_last = 0, rc_start = 0;
	int pc_tid = -2, pc_lo = 0, pc_next = 0;
	int ns_tid = -2; bool ns_val = false;   // tid_nonspecial[ns_tid] (round 5: one dependent load per record less - a walker changes reference a few times per file)
};

struct RecView
{
	const uint8_t* core;   // points at refID (record + 4)
	uint32_t bs; int32_t tid, pos; uint32_t l_name, mapq, n_cigar_raw, flag; int32_t l_seq, isize;
	const uint8_t* cigar; uint32_t n_cigar;  // effective CIGAR (may be the CG tag payload)
};

// the fixed part of a record as the chain walk keeps it one record ahead: block_size, refID, pos, (l_read_name | mapq | bin), (n_cigar_op | flag), l_seq, tlen
struct Hdr { uint32_t bs, tid, pos, w, w2, l_seq, isize; };
__device__ __forceinline__ Hdr load_hdr(const uint8_t* p)
{
	Hdr h; uint32_t a[4], b[2];
	__builtin_memcpy(a, p, 16); __builtin_memcpy(b, p + 16, 8);
	h.bs = a[0]; h.tid = a[1]; h.pos = a[2]; h.w = a[3]; h.w2 = b[0]; h.l_seq = b[1]; h.isize = ld32(p + 32);
	return h;
}
__device__ __forceinline__ RecView make_rec(const uint8_t* infl, int64_t off, const Hdr& h)
{
	RecView r; const uint8_t* p = infl + off;
	r.bs = h.bs; r.core = p + 4; r.tid = (int32_t)h.tid; r.pos = (int32_t)h.pos;
	r.l_name = h.w & 0xff; r.mapq = (h.w >> 8) & 0xff; r.n_cigar_raw = h.w2 & 0xffff; r.flag = h.w2 >> 16;
	r.l_seq = (int32_t)h.l_seq; r.isize = (int32_t)h.isize;
	r.cigar = p + 36 + r.l_name; r.n_cigar = r.n_cigar_raw;
	return r;
}
__device__ __forceinline__ RecView load_rec(const uint8_t* infl, int64_t off)
{
	RecView r; const uint8_t* p = infl + off;
	r.bs = ld32(p); r.core = p + 4;
	r.tid = (int32_t)ld32(p + 4); r.pos = (int32_t)ld32(p + 8);
	uint32_t w = ld32(p + 12), w2 = ld32(p + 16);
	r.l_name = w & 0xff; r.mapq = (w >> 8) & 0xff; r.n_cigar_raw = w2 & 0xffff; r.flag = w2 >> 16;
	r.l_seq = (int32_t)ld32(p + 20); r.isize = (int32_t)ld32(p + 32);
	r.cigar = p + 36 + r.l_name; r.n_cigar = r.n_cigar_raw;
	return r;
}
__device__ __forceinline__ const uint8_t* rec_qual(const RecView& r) { return r.core + 32 + r.l_name + 4ull * r.n_cigar_raw + ((uint32_t)r.l_seq + 1) / 2; }
__device__ __forceinline__ const uint8_t* rec_aux(const RecView& r) { return rec_qual(r) + (uint32_t)r.l_seq; }
__device__ __forceinline__ const uint8_t* rec_end(const RecView& r) { return r.core + r.bs; }

// linear aux scan (what htslib's bam_aux_get does); returns pointer to the type byte or nullptr
__device__ static const uint8_t* aux_find(const uint8_t* p, const uint8_t* end, uint8_t t0, uint8_t t1)
{
	while (p + 3 <= end)
	{
		const uint8_t* t = p + 2;
		if (p[0] == t0 && p[1] == t1) return t;
		uint8_t type = *t; const uint8_t* v = t + 1; size_t sz;
		switch (type)
		{
			case 'A': case 'c': case 'C': sz = 1; break;
			case 's': case 'S': sz = 2; break;
			case 'i': case 'I': case 'f': sz = 4; break;
			case 'd': sz = 8; break;
			case 'Z': case 'H': { const uint8_t* q = v; while (q < end && *q) ++q; sz = (size_t)(q - v) + 1; break; }
			case 'B': { if (v + 5 > end) return nullptr; uint8_t st = v[0]; uint32_t n = ld32(v + 1); size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4; sz = 5 + es * (size_t)n; break; }
			default: return nullptr;
		}
		p = v + sz;
	}
	return nullptr;
}
// BamAlignment::tagi (BamReader.cpp:286-297)
__device__ static int aux_tagi(const RecView& r, uint8_t t0, uint8_t t1)
{
	const uint8_t* t = aux_find(rec_aux(r), rec_end(r), t0, t1);
	if (!t) return 0;
	switch (*t)
	{
		case 'c': return (int8_t)t[1];
		case 'C': return t[1];
		case 's': return (int16_t)ld16(t + 1);
		case 'S': return ld16(t + 1);
		case 'i': return (int32_t)ld32(t + 1);
		case 'I': return (int)ld32(t + 1);
		default: return 0;
	}
}

// first region index in [first,last) with reg_end >= s  (merged regions: ends are sorted too)
__device__ __forceinline__ int lower_region(const int32_t* __restrict__ reg_end, int first, int last, int s)
{
	int a = first, b = last;
	while (a < b) { int m = (a + b) >> 1; if (reg_end[m] < s) a = m + 1; else b = m; }
	return a;
}

__device__ __forceinline__ void diff_add(const ScanParams& p, int reg, int a, int b) // +1 on [a,b] of region reg
{
	int32_t* d = p.diff + p.reg_doff[reg] - p.reg_start[reg];
	atomicAdd(d + a, p.sgn); atomicAdd(d + b + 1, -p.sgn);
}

__device__ static void gc_hit(const ScanParams& p, int tid, int s, int e)
{
	if (!p.n_gc) return;
	int first = p.tid_gc_first[tid], last = p.tid_gc_last[tid];
	if (first >= last) return;
	int i0 = lower_region(p.gc_end, first, last, s);
	int i1 = i0; while (i1 < last && p.gc_start[i1] <= e) ++i1;
	int n = i1 - i0;
	if (n <= 0) return;
	for (int i = i0; i < i1; ++i)
	{
		int bin = p.gc_bin[i];
		if (bin < 0) continue;
		if (n < GC_NMAX) atomicAdd(&p.gc_tab[(size_t)bin * GC_NMAX + n], (unsigned long long)(long long)p.sgn);
		else atomicAdd(&p.gc_over[bin], (double)p.sgn / (double)n);
	}
}

// min_baseq mask (BamAlignment::qualities): M-op bases with qual < min_baseq are NOT counted -> point decrement.
// Quirk kept: '=' / 'X' / 'H' / 'P' advance neither index.
__device__ static void baseq_decrements(const ScanParams& p, const RecView& r, int start1, int i0, int i1)
{
	const uint8_t* q = rec_qual(r);
	uint32_t ai = 0; int gi = 0;
	for (uint32_t k = 0; k < r.n_cigar; ++k)
	{
		uint32_t c = ld32(r.cigar + 4ull * k); uint32_t op = c & 15u, len = c >> 4;
		if (op == 0)
		{
			for (uint32_t j = 0; j < len && ai + j < (uint32_t)r.l_seq; ++j)   // (a CIGAR longer than SEQ: htslib would read past the qualities)
			{
				if (q[ai + j] < p.min_baseq)
				{
					int pos1 = start1 + gi + (int)j;
					for (int i = i0; i < i1; ++i) if (pos1 >= p.reg_start[i] && pos1 <= p.reg_end[i]) { int32_t* d = p.diff + p.reg_doff[i] - p.reg_start[i]; atomicAdd(d + pos1, -p.sgn); atomicAdd(d + pos1 + 1, p.sgn); }
				}
			}
			ai += len; gi += (int)len;
		}
		else if (op == 2 || op == 3) gi += (int)len;
		else if (op == 1 || op == 4) ai += len;
	}
}

// A list that the walk's lanes append to (the site pileup's candidates, the records deferred to the wave-per-record kernel): the entries of a WAVE are collected in
// LDS and leave in batches - one global atomic per batch for the exact count (no holes), and no lane waits for an atomic of its own (round 5: every appending lane
// waited for its atomic on one address; a long-read file appends every record). buf: 64 entries + the fill count, of the one-wave workgroup.
struct WaveList { unsigned long long e[64]; uint32_t n; uint32_t pad; };
typedef volatile __attribute__((address_space(3))) WaveList* WaveListP;   // (an LDS pointer as such: no generic pointer, no aperture test)
#define WAVE_LIST_P(x) ((WaveListP)(&(x)))
__device__ __forceinline__ void wave_list_flush(WaveListP b, int64_t* list, unsigned long long* count, int64_t cap, uint32_t rank, uint32_t n_act)
{
	// (called by the n_act active lanes, ranks 0 .. n_act - 1)
	const uint32_t cur = b->n;
	unsigned long long base = 0;
	if (rank == 0) base = atomicAdd(count, (unsigned long long)cur);
	base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
	for (uint32_t i = rank; i < cur; i += n_act) if ((long long)(base + i) < cap) list[base + i] = (int64_t)b->e[i];
}
__device__ __forceinline__ void wave_list_append(WaveListP b, int64_t* list, unsigned long long* count, int64_t cap, unsigned long long value)
{
	const unsigned long long m = __builtin_amdgcn_ballot_w64(true);   // the lanes that append now
	const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)), n = (uint32_t)__popcll(m);
	uint32_t cur = b->n;
	if (cur + n > 64u) { wave_list_flush(b, list, count, cap, rank, n); cur = 0; }
	b->e[cur + rank] = value;
	if (rank == 0) b->n = cur + n;
}
// the end of the wave's walk (all lanes)
__device__ __forceinline__ void wave_list_close(WaveListP b, int64_t* list, unsigned long long* count, int64_t cap)
{
	if (b->n) wave_list_flush(b, list, count, cap, threadIdx.x & 63u, 64u);
}

// The list of the records whose qualities baseq_tile_kernel masks behind the walk (MODE_DEPTH with min_baseq). Round 6: a WAVE takes the list's slots in blocks of
// BQ_BLOCK from the global counter and its lanes fill the block by rank (ballot of the lanes that append in this trip of the walk) - round 5 had every lane wait for
// its own atomic on ONE address, a round trip in nearly every trip of the wave's loop (some lane of the 64 met a region): the walk of -min_baseq took twice the time
// of the same walk without. A block's unused tail is filled with BQ_HOLE (the radix sort moves the holes behind the records; the mask kernel skips them).
// state: two LDS words of the one-wave workgroup {next free slot, end of the block}.
constexpr uint32_t BQ_BLOCK = 64;
constexpr unsigned long long BQ_OFF_MASK = (1ull << 36) - 1ull, BQ_HOLE = ~0ull;
__device__ __forceinline__ void bq_append(const ScanParams& p, uint32_t* state_, unsigned long long entry)
{
	volatile __attribute__((address_space(3))) uint32_t* state = (volatile __attribute__((address_space(3))) uint32_t*)state_;   // (an LDS pointer; written by whichever lane has rank 0: never kept in a register)
	const unsigned long long m = __builtin_amdgcn_ballot_w64(true);   // the lanes that append now
	const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)), n = (uint32_t)__popcll(m);
	uint32_t cur = 0;
	if (rank == 0)
	{
		cur = state[0]; uint32_t end = state[1];
		if (cur + n > end)
		{
			for (uint32_t k = cur; k < end; ++k) if ((long long)k < p.bq_cap) p.bq_list[k] = (int64_t)BQ_HOLE;
			cur = (uint32_t)atomicAdd(p.bq_count, (unsigned long long)BQ_BLOCK); end = cur + BQ_BLOCK;
			state[1] = end;
		}
		state[0] = cur + n;
	}
	cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);   // (the first active lane is the one of rank 0)
	const long long k = (long long)cur + rank;
	if (k < p.bq_cap) p.bq_list[k] = (int64_t)entry;
}
// the end of the wave's walk: the rest of its last block
__device__ __forceinline__ void bq_close(const ScanParams& p, const uint32_t* state_)
{
	const volatile __attribute__((address_space(3))) uint32_t* state = (const volatile __attribute__((address_space(3))) uint32_t*)state_;
	const long long k = (long long)state[0] + (threadIdx.x & 63u);
	if (k < (long long)state[1] && k < p.bq_cap) p.bq_list[k] = (int64_t)BQ_HOLE;
}

// Everything after the CIGAR sums are known. ord = ordinal of the record in the file.
template <int MODE>
__device__ static void classify(const ScanParams& p, const RecView& r, long long ord, long long ref_len, long long clip, bool spliced, Acc& a, uint32_t* lds_hist)
{
	const uint32_t flag = r.flag;
	const bool unmapped = flag & 0x4, secondary = flag & 0x100, supp = flag & 0x800, dup = flag & 0x400;
	const bool paired = flag & 0x1, proper = flag & 0x2, read1 = flag & 0x40;
	long long rlen = unmapped ? 0 : ref_len; if (rlen == 0) rlen = 1;
	const int start1 = r.pos + 1, end1 = (int)(r.pos + rlen);   // 1-based closed (BamReader.h:80-94)
	const int length = r.l_seq;
	const bool tid_ok = r.tid >= 0 && r.tid < p.n_ref;
	a.v[A_ALG_BYTES] += 4 + (long long)r.bs;

	if (MODE == MODE_COUNT)
	{
		// BedReadCount (src/BedReadCount/main.cpp:55-66): mapped, not secondary / supplementary, MAPQ >= min_mapq; +1 for every overlapped line
		if (unmapped || secondary || supp || (int)r.mapq < p.min_mapq || !tid_ok) return;
		int first = p.tid_reg_first[r.tid], last = p.tid_reg_last[r.tid];
		if (first >= last) return;
		for (int i = lower_region(p.reg_end, first, last, start1); i < last && p.reg_start[i] <= end1; ++i) atomicAdd(&p.region_reads[i], (unsigned long long)(long long)p.sgn);
		return;
	}
	if (MODE == 3)
	{
		// coverage-tool filter: WorkerAverageCoverage.cpp:41-45 / WorkerLowOrHighCoverage.cpp:47-49
		if (dup || secondary || supp || unmapped || (int)r.mapq < p.min_mapq) return;
		if (p.skip_mismapped && !proper && r.mapq < 20) return;
		if (!tid_ok) return;
		// (round 6: the per-lane cache of the last search's answer, as in MODE_WGS - a walker's records lie a few bases apart, the binary search over an exome's 200 000
		// regions was 17 dependent loads per record: what made this mode 3 ms per step slower than MODE_WGS on the same walk)
		if (a.rc_tid != r.tid || start1 <= a.rc_lo || start1 > a.rc_hi)
		{
			const int first = p.tid_reg_first[r.tid], lastr = p.tid_reg_last[r.tid];
			const int j0 = first < lastr ? lower_region(p.reg_end, first, lastr, start1) : lastr;
			a.rc_tid = r.tid; a.rc_idx = j0; a.rc_last = lastr;
			a.rc_lo = j0 > first ? p.reg_end[j0 - 1] : INT32_MIN;
			a.rc_hi = j0 < lastr ? p.reg_end[j0] : INT32_MAX;
			a.rc_start = j0 < lastr ? p.reg_start[j0] : INT32_MAX;
			settle_loads();
		}
		const int last = a.rc_last;
		if (a.rc_start > end1) return;   // (the first region that ends at or behind the read's start begins behind its end: no overlap)
		int i0 = a.rc_idx, i1 = i0;
		for (; i1 < last && p.reg_start[i1] <= end1; ++i1) diff_add(p, i1, max(start1, p.reg_start[i1]), min(end1, p.reg_end[i1]));
		if (p.min_baseq > 0 && i1 > i0)
		{
			if (!p.bq_list || i0 >= (1 << 28)) baseq_decrements(p, r, start1, i0, i1);   // (a list entry holds the region index in 28 bits)
			else if (p.sgn > 0) bq_append(p, lds_hist + 1000, (unsigned long long)(r.core - 4 - p.infl) | ((unsigned long long)(uint32_t)i0 << 36));   // (offset in the tile: < 2^36; first overlapped region)
		}
		return;
	}

	// chrY/chrX read counts: index query chr:[1,len] returns tid==t && pos < len && endpos > 0, minus secondary/supplementary
	if (!secondary && !supp)
	{
		if (r.tid == p.tid_x && r.pos < p.len_x && r.pos + rlen > 0) a.n[A_READS_X]++;
		if (r.tid == p.tid_y && r.pos < p.len_y && r.pos + rlen > 0) a.n[A_READS_Y]++;
	}

	// indexed ROI pass of mapping_wgs (Statistics.cpp:1154-1182), fused: per (read, overlapped region) pair
	if (MODE == NGSQC_MODE_WGS && p.n_regions && !secondary && !supp && !unmapped && tid_ok)
	{
		if (a.rc_tid != r.tid || start1 <= a.rc_lo || start1 > a.rc_hi)
		{
			const int first = p.tid_reg_first[r.tid], last = p.tid_reg_last[r.tid];
			const int i0 = first < last ? lower_region(p.reg_end, first, last, start1) : last;
			a.rc_tid = r.tid; a.rc_idx = i0; a.rc_last = last;
			a.rc_lo = i0 > first ? p.reg_end[i0 - 1] : INT32_MIN;   // (start1 > reg_end[i0 - 1]: the search would still pass i0 - 1)
			a.rc_hi = i0 < last ? p.reg_end[i0] : INT32_MAX;        // (start1 <= reg_end[i0]: it would still stop at i0)
			a.rc_start = i0 < last ? p.reg_start[i0] : INT32_MAX;
			settle_loads();
		}
		const int last = a.rc_last;
		if (a.rc_start <= end1)   // (the first region that ends at or behind the read's start begins inside the read's span: an overlap)
		{
			for (int i = a.rc_idx; i < last && p.reg_start[i] <= end1; ++i)
			{
				gc_hit(p, r.tid, start1, end1);
				if (!dup && (int)r.mapq >= p.min_mapq)
				{
					a.v[A_USABLE_ROI] += length;
					diff_add(p, i, max(start1, p.reg_start[i]), min(end1, p.reg_end[i]));
				}
			}
		}
	}

	if (secondary || supp) return;
	a.n[A_TOTAL]++;
	if (paired && (unsigned long long)ord < a.first_paired) a.first_paired = (unsigned long long)ord;
	a.v[A_SUM_LEN] += length;
	if (length > a.max_len) a.max_len = length;
	{
		// (max length, first ordinal reaching it) for the running-max "trimmed bases" rule (Statistics.cpp:428-429,565-568)
		unsigned long long key = ((unsigned long long)(uint32_t)length << 40) | (0xFFFFFFFFFFull - (unsigned long long)ord);
		if (key > a.best_key) a.best_key = key;
	}

	if (!unmapped)
	{
		a.n[A_MAPPED]++;
		a.v[A_BASES_MAPPED] += length;
		a.v[A_CLIPPED] += clip;
		if (MODE == NGSQC_MODE_ROI)
		{
			if (tid_ok)
			{
				int first = p.tid_reg_first[r.tid], last = p.tid_reg_last[r.tid];
				if (first < last)
				{
					int n0 = lower_region(p.reg_end, first, last, start1 - 250);
					if (n0 < last && p.reg_start[n0] <= end1 + 250)
					{
						a.n[A_NEAR]++;
						int i0 = lower_region(p.reg_end, n0, last, start1);
						if (i0 < last && p.reg_start[i0] <= end1)
						{
							a.n[A_ONTARGET]++;
							int dp = aux_tagi(r, 'D', 'P');
							if (dp != 0) { int bin = min(dp, 4); bin = bin < 1 ? 0 : bin - 1; a.n[A_DD0 + bin]++; }
							if (!dup && (int)r.mapq >= p.min_mapq)
							{
								int dpi = min(max(dp, 0), 4);
								for (int i = i0; i < last && p.reg_start[i] <= end1; ++i)
								{
									int s = max(p.reg_start[i], start1), e = min(p.reg_end[i], end1);
									long long n = e - s + 1;
									a.v[A_USABLE] += n; a.v[A_DP0 + dpi] += n; a.v[A_USABLE_RAW] += n * ((long long)dp + 1);
									a.v[A_NO_OVERLAP] += n;
									diff_add(p, i, s, e);
								}
								int insert_size = abs(r.isize);
								if (read1 && paired && proper && !spliced && 2 * length > insert_size)
								{
									int ovl = 2 * length - insert_size;
									int os = r.isize > 0 ? start1 + length - ovl : start1;
									int oe = os + ovl - 1;
									for (int i = lower_region(p.reg_end, first, last, os); i < last && p.reg_start[i] <= oe; ++i)
										a.v[A_NO_OVERLAP] -= (min(p.reg_end[i], oe) - max(p.reg_start[i], os) + 1);
								}
							}
							gc_hit(p, r.tid, start1, end1);
						}
					}
				}
			}
		}
		else
		{
			if (tid_ok && a.ns_tid != r.tid) { a.ns_tid = r.tid; a.ns_val = p.tid_nonspecial[r.tid] != 0; }
			if (tid_ok && a.ns_val)
			{
				a.n[A_ONTARGET]++;
				if (!dup && (int)r.mapq >= p.min_mapq) a.v[A_USABLE] += length; // "no overlap" share is resolved with first_paired_idx afterwards
			}
		}
	}

	if (paired && proper)
	{
		a.n[A_PP]++;
		if (!spliced)
		{
			int insert_size = abs(r.isize);
			if (insert_size < 1000)
			{
				a.n[A_INS_CNT]++; a.n[A_INS_SUM] += (uint32_t)insert_size;
				atomicAdd(&lds_hist[insert_size], 1u);
				if (MODE != NGSQC_MODE_ROI && read1 && !dup && (int)r.mapq >= p.min_mapq && 2 * length > insert_size) a.v[A_NO_OVERLAP] -= (2 * length) - insert_size;
			}
		}
	}
	if (dup) a.n[A_DUP]++;
}

__device__ __forceinline__ long long wave_sum(long long v)
{
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

__device__ static void flush(const ScanParams& p, Acc& a, uint32_t* lds_hist)
{
	const int lane = threadIdx.x & 63;
	for (int i = 0; i < A_COUNT; ++i) { long long s = wave_sum(a.v[i] + (long long)a.n[i]); if (lane == 0 && s) atomicAdd(&p.counters[i], (unsigned long long)(p.sgn * s)); }
	int m = a.max_len; unsigned long long bk = a.best_key, fp = a.first_paired;
	for (int o = 32; o > 0; o >>= 1)
	{
		m = max(m, __shfl_xor(m, o));
		unsigned long long t = __shfl_xor(bk, o); if (t > bk) bk = t;
		t = __shfl_xor(fp, o); if (t < fp) fp = t;
	}
	if (lane == 0 && p.sgn > 0)
	{
		if (m > 0) atomicMax(&p.counters[A_MAX_LEN], (unsigned long long)m);
		if (bk) atomicMax(&p.counters[p.tile_slots ? A_TILE_KEY : A_FIRST_MAX_KEY], bk);
		if (fp != ~0ull) atomicMin(&p.counters[p.tile_slots ? A_TILE_PAIRED : A_FIRST_PAIRED], fp);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < 1000; i += blockDim.x) if (lds_hist[i]) atomicAdd(&p.counters[A_HIST0 + i], (unsigned long long)(p.sgn * (long long)lds_hist[i]));
}

// CIGAR sums of a short record + classify; false: the record goes to the wave-per-record kernel (long CIGAR, possible CG:B,I tag).
// The first four operations come with ONE 16-byte load (97 % of the short reads of a WGS have at most three; the bytes behind a shorter CIGAR are the
// record's own sequence, and the tile buffers end with 64 spare bytes): the walk waits for one round trip instead of one per operation.
// The first four CIGAR operations of a record whose fields fit, WITHOUT a branch: a divergent if / else around the loads made the compiler wait for them at the
// join (s_waitcnt vmcnt(0) in front of the next record's header request). Operation 0 is a 4-byte load; operations 1-3 a 12-byte load whose address, for a
// record of one operation (93 % of a WGS), is the record's own header - bytes that are in the cache already - instead of 12 more bytes that cross into the next
// 64-byte sector four times as often as the first four.
__device__ __forceinline__ void load_cigar4(const RecView& r, uint32_t (&c4)[4])
{
	const bool any = r.n_cigar_raw >= 1 && r.n_cigar_raw <= (uint32_t)LONG_CIGAR, many = any && r.n_cigar_raw > 1;
	const uint8_t* p0 = any ? r.cigar : r.core; const uint8_t* p1 = many ? r.cigar + 4 : r.core;
	uint32_t a = ld32(p0), b[3]; __builtin_memcpy(b, p1, 12);
	c4[0] = any ? a : 0u; c4[1] = many ? b[0] : 0u; c4[2] = many ? b[1] : 0u; c4[3] = many ? b[2] : 0u;
}
template <int MODE>
__device__ __forceinline__ bool scan_record(const ScanParams& p, RecView& r, long long ord, Acc& a, uint32_t* lds_hist, long long* ref_len_out = nullptr, const uint32_t* c4_in = nullptr)
{
	if (r.n_cigar_raw > (uint32_t)LONG_CIGAR) return false;
	uint32_t c4[4];
	if (c4_in) { c4[0] = c4_in[0]; c4[1] = c4_in[1]; c4[2] = c4_in[2]; c4[3] = c4_in[3]; } else load_cigar4(r, c4);
	if (r.n_cigar_raw > 0 && r.tid >= 0 && r.pos >= 0 && (c4[0] & 15u) == 4 && (int32_t)(c4[0] >> 4) == r.l_seq) return false;   // possible CG:B,I long CIGAR (htslib bam_tag2cigar)
	long long ref_len = 0, clip = 0; bool spliced = false;
	#pragma unroll
	for (uint32_t k = 0; k < 4; ++k)
	{
		if (k < r.n_cigar)
		{
			const uint32_t c = c4[k], op = c & 15u, len = c >> 4;
			if ((0x18Du >> op) & 1u) ref_len += len;            // M,D,N,=,X  (bits 0,2,3,7,8)
			else if (op == 4 || op == 5) clip += len;
			if (op == 3) spliced = true;
		}
	}
	for (uint32_t k = 4; k < r.n_cigar; ++k)
	{
		uint32_t c = ld32(r.cigar + 4ull * k); uint32_t op = c & 15u, len = c >> 4;
		if ((0x18Du >> op) & 1u) ref_len += len;
		else if (op == 4 || op == 5) clip += len;
		if (op == 3) spliced = true;
	}
	if (ref_len_out) *ref_len_out = ref_len;
	classify<MODE>(p, r, ord, ref_len, clip, spliced && !(r.flag & 0x4u), a, lds_hist);
	return true;
}

// The site pileup of a job riding the same walk (round 4): a record that passes the pileup's read filters (BamReader.cpp:830-836) and whose reference span holds
// at least one known site (or whose span is not known here: a deferred long-CIGAR record) leaves its tile-local offset in a list - 0.15 % of the records of a 30x
// WGS for the 29 k contamination sites - and pileup_kernel runs over that list instead of reading every record of the tile a second time.
__device__ __forceinline__ void pile_candidate(const ScanParams& p, const RecView& r, int64_t o, bool span_known, long long ref_len, Acc& a, WaveListP wl)
{
	const uint32_t flag = r.flag;
	if (flag & (0x100u | 0x800u | 0x400u | 0x4u)) return;
	if (!(flag & 0x2u) && !p.pile.include_npp) return;
	if ((int)r.mapq < p.pile.min_mapq || r.tid < 0 || r.tid >= p.n_ref) return;
	if (ref_len == 0) ref_len = 1;
	const int start1 = r.pos + 1, end1 = span_known ? (int)(r.pos + ref_len) : INT32_MAX;   // (a deferred record: its span is not known here - a candidate if any site lies behind its start)
	if (a.pc_tid != r.tid || start1 <= a.pc_lo || start1 > a.pc_next)
	{
		// the first site at or behind start1 (and the one in front of it): they answer every record that starts between them; a reference without sites: never
		const int first = p.pile.tid_first[r.tid], last = p.pile.tid_last[r.tid];
		a.pc_tid = r.tid; a.pc_lo = INT32_MIN; a.pc_next = INT32_MAX;
		if (first < last)
		{
			const int64_t b0 = p.pile.tid_bucket0[r.tid], nbk = p.pile.tid_bucket0[r.tid + 1] - b0;
			int64_t bi = start1 > 0 ? (int64_t)(start1 >> PILEUP_BUCKET_SHIFT) : 0; if (bi >= nbk) bi = nbk - 1;
			int i = p.pile.bucket[b0 + bi];
			while (i < last && p.pile.site_pos[i] < start1) ++i;
			if (i < last) a.pc_next = p.pile.site_pos[i];
			if (i > first) a.pc_lo = p.pile.site_pos[i - 1];
		}
		settle_loads();
	}
	if (a.pc_next == INT32_MAX || a.pc_next > end1) return;
#ifdef NGSQC_NO_WAVE_LISTS
	const unsigned long long k = atomicAdd(p.pile.count, 1ull);
	if ((long long)k < p.pile.cap) p.pile.list[k] = o;
#else
	wave_list_append(wl, p.pile.list, p.pile.count, p.pile.cap, (unsigned long long)o);
#endif
}

template <int MODE>
__global__ __launch_bounds__(256) void scan_kernel(const ScanParams p)
{
	__shared__ uint32_t lds_hist[1000];
	for (int i = threadIdx.x; i < 1000; i += blockDim.x) lds_hist[i] = 0;
	__syncthreads();
	Acc a; for (int i = 0; i < A_COUNT; ++i) { a.v[i] = 0; a.n[i] = 0; } a.max_len = 0; a.best_key = 0; a.first_paired = ~0ull;
	const long long stride = (long long)gridDim.x * blockDim.x;
	for (long long li = (long long)blockIdx.x * blockDim.x + threadIdx.x; li < p.n_rec; li += stride)
	{
		const long long ord = p.ord_base + li;   // ordinal in the file; li = index inside the resident tile
		RecView r = load_rec(p.infl, p.recoff[li]);
		if (!scan_record<MODE>(p, r, ord, a, lds_hist))
		{
			unsigned long long k = atomicAdd(&p.counters[A_LONG_COUNT], 1ull);
			if ((long long)k < p.long_cap) p.long_list[k] = li;
		}
	}
	flush(p, a, lds_hist);
}

// wave-per-record path: long CIGARs (ONT) and CG-tag records
template <int MODE>
__global__ __launch_bounds__(256) void scan_long_kernel(const ScanParams p, long long n_long)
{
	__shared__ uint32_t lds_hist[1000];
	for (int i = threadIdx.x; i < 1000; i += blockDim.x) lds_hist[i] = 0;
	__syncthreads();
	Acc a; for (int i = 0; i < A_COUNT; ++i) { a.v[i] = 0; a.n[i] = 0; } a.max_len = 0; a.best_key = 0; a.first_paired = ~0ull;
	const int lane = threadIdx.x & 63;
	const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
	// Round 6: a record of this kernel used to cost its wave a dozen DEPENDENT round trips (list entry -> entry base -> record offset -> header -> first
	// operation -> seven steps of the CIGAR, each waited for on its own), 64 us per record with nothing in between. Now a wave takes a CONTIGUOUS stretch of the
	// list, its lanes resolve the offsets of up to 64 records at once (one chain of round trips per 64 records), the next record's header is requested while
	// this one is processed, and the CIGAR's 16-byte loads are issued four steps (1 024 operations) at a time.
	const long long per_wave = (n_long + n_waves - 1) / n_waves, w_end = min(n_long, (wave + 1) * per_wave);
	for (long long wb = wave * per_wave; wb < w_end; wb += 64)
	{
		const int nb = (int)min(64ll, w_end - wb);
		long long my_off = 0, my_ord = 0;
		if (lane < nb)
		{
			long long li = p.long_list[wb + lane];
			if (p.entry_base) { my_ord = li; li = p.entry_base[li >> NAME_SHIFT] + (li & ((1ll << NAME_SHIFT) - 1)); }   // a tile scanned by the chain walk: records are compared by their (entry, k) names
			else my_ord = p.ord_base + li;
			my_off = p.recoff[li];
		}
		auto lane_ll = [&](long long v, int j) -> long long {
			return (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)v >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, j));
		};
		Hdr h_nx = load_hdr(p.infl + lane_ll(my_off, 0));
	for (int jr = 0; jr < nb; ++jr)
	{
		const long long off = lane_ll(my_off, jr), ord = lane_ll(my_ord, jr);
		const Hdr h_cur = h_nx;
		if (jr + 1 < nb) h_nx = load_hdr(p.infl + lane_ll(my_off, jr + 1));
		RecView r = make_rec(p.infl, off, h_cur);
		// the CIGAR, four operations per lane and step, four steps per request group (a read of 20 kb has ~1 700 operations: two groups). What lies behind the last
		// operation (the record's bases; the next record or the tile's spare bytes behind a CG:B,I array) is loaded and not looked at.
		long long ref_len = 0, clip = 0; int spl = 0; uint32_t c_first = 0; bool cg_checked = false;
		for (int pass = 0; pass < 2; ++pass)
		{
			ref_len = 0; clip = 0; spl = 0;
			for (uint32_t g0 = 0; g0 < r.n_cigar; g0 += 1024u)
			{
				const uint32_t k0 = g0 + 4u * (uint32_t)lane;
				uint32_t c4[4][4];
				#pragma unroll
				for (uint32_t u = 0; u < 4u; ++u)
				{
					const uint32_t kk = k0 + 256u * u;
					if (kk < r.n_cigar) __builtin_memcpy(c4[u], r.cigar + 4ull * kk, 16); else { c4[u][0] = c4[u][1] = c4[u][2] = c4[u][3] = 0u; }
				}
				if (k0 == 0) c_first = c4[0][0];
				#pragma unroll
				for (uint32_t u = 0; u < 4u; ++u)
				{
					#pragma unroll
					for (uint32_t j = 0; j < 4u; ++j)
					{
						if (k0 + 256u * u + j < r.n_cigar)
						{
							const uint32_t c = c4[u][j], op = c & 15u, len = c >> 4;
							if ((0x18Du >> op) & 1u) ref_len += len;
							else if (op == 4 || op == 5) clip += len;
							if (op == 3) spl = 1;
						}
					}
				}
			}
			if (cg_checked) break;
			cg_checked = true;
			// CG:B,I substitution (htslib bam_tag2cigar): first op kS with k == l_seq, tag present with >= n_cigar entries - then the sums are taken again over the tag's array
			bool redo = false;
			if (r.n_cigar_raw > 0 && r.tid >= 0 && r.pos >= 0)
			{
				const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c_first, 0);
				if ((c0 & 15u) == 4 && (int32_t)(c0 >> 4) == r.l_seq)
				{
					unsigned long long cg = 0; uint32_t n = 0;
					if (lane == 0)
					{
						const uint8_t* t = aux_find(rec_aux(r), rec_end(r), 'C', 'G');
						if (t && t[0] == 'B' && t[1] == 'I') { n = ld32(t + 2); if (n >= r.n_cigar_raw && n < (1u << 29)) cg = (unsigned long long)(uintptr_t)(t + 6); }
					}
					cg = __shfl(cg, 0); n = __shfl(n, 0);
					if (cg) { r.cigar = (const uint8_t*)(uintptr_t)cg; r.n_cigar = n; redo = true; }
				}
			}
			if (!redo) break;
		}
		ref_len = wave_sum(ref_len); clip = wave_sum(clip); spl = __any(spl);
		if (MODE == 3 && p.min_baseq > 0)
		{
			// coverage filter first (same as classify), then the per-base mask with a wave prefix sum over CIGAR ops
			const uint32_t flag = r.flag;
			bool pass = !(flag & 0x400) && !(flag & 0x100) && !(flag & 0x800) && !(flag & 0x4) && (int)r.mapq >= p.min_mapq
			            && !(p.skip_mismapped && !(flag & 0x2) && r.mapq < 20) && r.tid >= 0 && r.tid < p.n_ref;
			if (pass)
			{
				long long rlen = ref_len ? ref_len : 1;
				const int start1 = r.pos + 1, end1 = (int)(r.pos + rlen);
				int first = p.tid_reg_first[r.tid], last = p.tid_reg_last[r.tid];
				int i0 = first < last ? lower_region(p.reg_end, first, last, start1) : last, i1 = i0;
				while (i1 < last && p.reg_start[i1] <= end1) ++i1;
				if (lane == 0) for (int i = i0; i < i1; ++i) diff_add(p, i, max(start1, p.reg_start[i]), min(end1, p.reg_end[i]));
				if (i1 > i0)
				{
					const uint8_t* q = rec_qual(r);
					long long ai_base = 0, gi_base = 0;
					for (uint32_t k0 = 0; k0 < r.n_cigar; k0 += 64)
					{
						uint32_t k = k0 + lane; uint32_t op = 15, len = 0;
						if (k < r.n_cigar) { uint32_t c = ld32(r.cigar + 4ull * k); op = c & 15u; len = c >> 4; }
						long long da = (op == 0 || op == 1 || op == 4) ? len : 0, dg = (op == 0 || op == 2 || op == 3) ? len : 0;
						long long sa = da, sg = dg; // inclusive wave scan
						for (int o = 1; o < 64; o <<= 1) { long long ta = __shfl_up(sa, o), tg = __shfl_up(sg, o); if (lane >= o) { sa += ta; sg += tg; } }
						long long ai = ai_base + sa - da, gi = gi_base + sg - dg;
						if (op == 0)
						{
							for (uint32_t j = 0; j < len && ai + (long long)j < (long long)r.l_seq; ++j)
							{
								if (q[ai + j] < p.min_baseq)
								{
									int pos1 = start1 + (int)gi + (int)j;
									for (int i = i0; i < i1; ++i) if (pos1 >= p.reg_start[i] && pos1 <= p.reg_end[i]) { int32_t* d = p.diff + p.reg_doff[i] - p.reg_start[i]; atomicAdd(d + pos1, -1); atomicAdd(d + pos1 + 1, 1); }
								}
							}
						}
						ai_base += __shfl(sa, 63); gi_base += __shfl(sg, 63);
					}
				}
			}
			if (lane == 0) a.v[A_ALG_BYTES] += 4 + (long long)r.bs;
			continue;
		}
		if (lane == 0)
		{
			classify<MODE>(p, r, ord, ref_len, clip, spl != 0 && !(r.flag & 0x4u), a, lds_hist);
		}
	}
	}
	flush(p, a, lds_hist);
}

// ---- MODE_DEPTH with min_baseq behind the riding walk (round 5): the records that overlap a region (passed the coverage filters, at most LONG_CIGAR operations) were
// compacted into a list by the walk; here a LANE per record masks its low-quality bases (BamAlignment::qualities, BamReader.cpp:210-255: baseq_decrements above).
// Inside the walk the same loop stalled 63 lanes for the one whose record overlapped a region (224 vs 147 ms per 96 M reads, round 3); on the compacted list every
// lane has a record. (A wave per record was tried first: 20 dependent loads per record with nothing to overlap them - 5.6 ms per 48 M reads.) ----
static int scan_grid_cap(long long n) { const long long wgs = (n + 255) / 256; return (int)(wgs < 1 ? 1 : (wgs < 2048 ? wgs : 2048)); }
// Round 6: the decrements of 256 neighbouring records of the sorted list are first counted in an LDS TILE of the difference array (BQ_TILE slots behind the lowest
// slot any of them touches: the records are coordinate-sorted and the array holds the regions back to back, so a chunk's decrements lie within a few thousand slots)
// and leave as ONE atomic per slot: diff[o] += dec[o - 1] - dec[o]. At 30x every position of a region is masked by about fifteen reads - the round-5 kernel sent
// two global atomics per masked base to neighbouring addresses (1.8 * 10^9 per step of the bench, 26 G/s: what -min_baseq cost), this one two per position.
// A decrement outside the tile (a chunk that spans more than the tile) goes to the array directly.
constexpr int BQ_TILE = 4096;
constexpr uint32_t BQ_BITS = 256;   // read positions whose "below min_baseq" flags a lane keeps as bits (four words of 64; longer reads: the rest byte by byte)
// Round 6 (second half): what a masked base cost was not the atomics but FOUR DEPENDENT LOADS of the region table per base (reg_end, reg_start, reg_end, reg_doff:
// ~75 masked bases per record of the bench data) behind a serial 16-byte quality loop. Now a lane (1) asks for its record's header and - at the same time - for the
// bounds of the region the walk found, (2) asks for all qualities at once, 64 bytes per request group, and turns them into bits "below the threshold" (the bytes of a
// dword tested together, the four flags gathered with one multiplication), (3) walks the CIGAR once per overlapped region: an M stretch is cut to the region's bounds
// in registers, and only the set bits of that range are visited - each one LDS atomic. No load depends on a base.
__global__ __launch_bounds__(256) void baseq_tile_kernel(const ScanParams p, long long n)
{
	__shared__ int32_t dec[BQ_TILE + 1];
	__shared__ unsigned long long s_base;
	const uint32_t thr = 0x01010101u * (uint32_t)(p.min_baseq & 127);
	for (long long c0 = (long long)blockIdx.x * 256; c0 < n; c0 += (long long)gridDim.x * 256)
	{
		for (int k = threadIdx.x; k <= BQ_TILE; k += 256) dec[k] = 0;
		if (threadIdx.x == 0) s_base = ~0ull;
		__syncthreads();
		const long long li = c0 + threadIdx.x;
		const unsigned long long e = li < n ? (unsigned long long)p.bq_list[li] : BQ_HOLE;
		const bool have = (e & BQ_OFF_MASK) != BQ_OFF_MASK;   // (a hole: the unused tail of a wave's block)
		RecView r{}; int start1 = 0, end1 = 0, i0 = 0, i1 = 0; int rs = 0, re = 0; long long doff = 0;
		uint32_t c4[4] = {0, 0, 0, 0};
		unsigned long long low[BQ_BITS / 64] = {0, 0, 0, 0};
		if (have)
		{
			i0 = (int)(e >> 36);   // (the walk's region search: not repeated)
			const Hdr hd = load_hdr(p.infl + (int64_t)(e & BQ_OFF_MASK));
			rs = p.reg_start[i0]; re = p.reg_end[i0]; doff = p.reg_doff[i0];
			r = make_rec(p.infl, (int64_t)(e & BQ_OFF_MASK), hd);
			load_cigar4(r, c4);
			const uint8_t* q = rec_qual(r);
			const uint32_t nq = min((uint32_t)r.l_seq, BQ_BITS);
			// (behind the qualities lie the record's tags / the next record / the tile buffer's 64 spare bytes: bits at or behind l_seq are cleared below)
			#pragma unroll
			for (uint32_t c = 0; c < BQ_BITS / 64; ++c)
				if (64u * c < nq)
				{
					uint32_t w[16]; __builtin_memcpy(w, q + 64u * c, 64);
					unsigned long long mm = 0;
					#pragma unroll
					for (int k = 0; k < 16; ++k)
					{
						// bit 7 of a byte of y = (x | 0x80) - min_baseq is set iff the byte's low seven bits reach the threshold, bit 7 of x iff the byte is 128 or more
						// (0xff: no quality): clear in both = below. The four flags (bits 7, 15, 23, 31) move to bits 21..24 of the product: no two terms meet in a bit
						const uint32_t lt = (~(((w[k] | 0x80808080u) - thr) | w[k]) & 0x80808080u) >> 7;
						mm |= (unsigned long long)(((lt * 0x00204081u) >> 21) & 15u) << (4 * k);
					}
					const uint32_t left = nq - 64u * c;
					low[c] = left >= 64u ? mm : mm & ((1ull << left) - 1ull);
				}
			if (p.min_baseq > 127) { low[0] = low[1] = low[2] = low[3] = 0; }   // (a threshold the byte trick does not hold for: every base goes the byte way below)
			long long ref_len = 0;
			#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) if (k < r.n_cigar && ((0x18Du >> (c4[k] & 15u)) & 1u)) ref_len += c4[k] >> 4;
			for (uint32_t k = 4; k < r.n_cigar; ++k) { const uint32_t c = ld32(r.cigar + 4ull * k); if ((0x18Du >> (c & 15u)) & 1u) ref_len += c >> 4; }
			if (ref_len == 0) ref_len = 1;
			start1 = r.pos + 1; end1 = (int)(r.pos + ref_len);
			const int last = p.tid_reg_last[r.tid];
			i1 = i0; if (i1 < last && rs <= end1) ++i1;
			while (i1 > i0 && i1 < last && p.reg_start[i1] <= end1) ++i1;
			if (i1 > i0) atomicMin(&s_base, (unsigned long long)(doff + (long long)(max(start1, rs) - rs)));
		}
		__syncthreads();
		const long long base = (long long)s_base;
		if (have && i1 > i0)
		{
			// BamAlignment::qualities (BamReader.cpp:210-255), as baseq_decrements: M bases below min_baseq; '=' / 'X' / 'H' / 'P' advance neither index.
			// (regions of an unmerged BED may overlap: every region that holds the position, as the reference's per-line depth does - a pass per region)
			const uint8_t* q = rec_qual(r);
			const bool by_byte = p.min_baseq > 127;
			for (int i = i0; i < i1; ++i)
			{
				if (i > i0) { rs = p.reg_start[i]; re = p.reg_end[i]; doff = p.reg_doff[i]; }
				uint32_t ai = 0; int gi = 0;
				for (uint32_t k = 0; k < r.n_cigar; ++k)
				{
					const uint32_t c = k < 4 ? (k == 0 ? c4[0] : k == 1 ? c4[1] : k == 2 ? c4[2] : c4[3]) : ld32(r.cigar + 4ull * k), op = c & 15u, len = c >> 4;
					if (op == 0)
					{
						const uint32_t n_m = ai < (uint32_t)r.l_seq ? min(len, (uint32_t)r.l_seq - ai) : 0u;   // (a CIGAR longer than SEQ: htslib would read past the qualities)
						// base j of the stretch lies at reference position p0 + j; inside the region: j in [j_lo, j_hi)
						const long long p0 = (long long)start1 + gi;
						const long long j_lo = max(0ll, (long long)rs - p0), j_hi = min((long long)n_m, (long long)re - p0 + 1);
						if (j_lo < j_hi)
						{
							const uint32_t a_lo = ai + (uint32_t)j_lo, a_hi = ai + (uint32_t)j_hi;        // read positions [a_lo, a_hi)
							const long long slot0 = doff + (p0 - (long long)ai - rs);                       // slot of read position x: slot0 + x
							auto masked = [&](uint32_t x) {
								const long long o = slot0 + (long long)x, rel = o - base;
								if (rel >= 0 && rel < BQ_TILE) atomicAdd(&dec[rel], 1);
								else { atomicAdd(p.diff + o, -p.sgn); atomicAdd(p.diff + o + 1, p.sgn); }
							};
							const uint32_t b_hi = by_byte ? a_lo : min(a_hi, BQ_BITS);
							for (uint32_t cw = a_lo >> 6; cw * 64u < b_hi; ++cw)
							{
								unsigned long long m = cw == 0 ? low[0] : cw == 1 ? low[1] : cw == 2 ? low[2] : low[3];
								if (a_lo > 64u * cw) m &= ~0ull << (a_lo - 64u * cw);
								if (b_hi < 64u * cw + 64u) m &= (1ull << (b_hi - 64u * cw)) - 1ull;
								while (m) { const uint32_t bit = (uint32_t)__builtin_ctzll(m); m &= m - 1ull; masked(64u * cw + bit); }
							}
							for (uint32_t x = max(a_lo, b_hi); x < a_hi; ++x) if (q[x] < p.min_baseq) masked(x);   // (reads longer than BQ_BITS)
						}
						ai += len; gi += (int)len;
					}
					else if (op == 2 || op == 3) gi += (int)len;
					else if (op == 1 || op == 4) ai += len;
				}
			}
		}
		__syncthreads();
		if (base >= 0 && s_base != ~0ull)
			for (int k = threadIdx.x; k <= BQ_TILE; k += 256)
			{
				const int v = (k ? dec[k - 1] : 0) - (k < BQ_TILE ? dec[k] : 0);
				if (v) atomicAdd(p.diff + base + k, v * p.sgn);
			}
		__syncthreads();
	}
}
// The walk's lanes append to the list in whatever order their atomics arrive: neighbouring entries lie anywhere in a 12 GB tile, and a lane-per-record pass over
// that order pays a TLB and HBM miss for every access (10.6 ms per 48 M reads, against 2.8 ms for the same loop inside the thread-per-record scan). The entries
// are sorted by their offset first (rocPRIM radix sort on the low 36 bits: ~1 M keys per tile of an exome), then neighbouring lanes read neighbouring records.
size_t baseq_sort_bytes(int64_t n)
{
	size_t bytes = 0;
	(void)rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)n, 0, 36);
	return bytes;
}
void launch_baseq_list(const ScanParams& p, int64_t n, hipStream_t s, int64_t* d_sorted, void* d_tmp, size_t tmp_bytes)
{
	if (!p.bq_list || n <= 0) return;
	ScanParams q = p;
	if (d_sorted && d_tmp)
	{
		if (rocprim::radix_sort_keys(d_tmp, tmp_bytes, (unsigned long long*)p.bq_list, (unsigned long long*)d_sorted, (size_t)n, 0, 36, s) != hipSuccess) throw std::runtime_error("rocprim::radix_sort_keys failed");
		q.bq_list = d_sorted;
	}
	hipLaunchKernelGGL(baseq_tile_kernel, dim3(scan_grid_cap(n)), dim3(256), 0, s, q, (long long)n); KCHECK();
}

// ---- the scan fused into K2's chain walk ----
// What index_count_kernel does (one thread per entry walks the entry's record chain, validates every record like bam_read1, keeps the entry-relative
// offsets) - and every record the walk passes is scanned right there: the walk has the record's first line in its hands, the separate scan kernel would
// fetch it a second time (a third of the inflated tile per pass). A record is named (entry << NAME_SHIFT | k) until the counts are scanned; the order-dependent
// results (first longest read, first paired read) are kept per tile in that form (A_TILE_KEY / A_TILE_PAIRED), the host turns them into ordinals.
// The launch happens before the chain is known to be right (index_chain_kernel): sgn = -1 takes a tile's contributions back - the same walks from the same
// starts, so whatever a false start guess made a walker add is subtracted again.
// Round 5: (a) an entry is a PIECE of a member (common.h entry_range): four walkers per member instead of one, a quarter of the dependent chain each and four
// times the lines in flight; (b) the fixed 36 bytes of the NEXT record are requested before this record is processed (round 3/4 asked for its block_size only
// and then waited for the fields): one round trip per record is the CIGAR's, the header's is hidden behind the scan of the record in front; (c) the offsets go
// out as 16-byte stores of eight (round 4: a 2-byte store per record and lane - 64 partial sectors per instruction, 32.6 bytes of HBM writes per record).
// (Round 6, probed and not kept - profiles/r06_walk_probe.txt: the first 96 bytes of the next record by LDS-DMA, six global_load_lds_dwordx4 per lane, header and CIGAR
// read out of LDS: one request per record instead of two dependent ones, and 20 % SLOWER - the walk is bound by what the address unit and the L1 process per lane,
// not by the CIGAR's round trip: 96 bytes per lane and record instead of 52.)
template <int MODE, int WAVES>
__global__ __launch_bounds__(64, WAVES) void walk_scan_kernel(const ScanParams p, const BlockDesc* __restrict__ blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm,
                                                        const int32_t* __restrict__ start, uint32_t* __restrict__ cnt, int64_t* __restrict__ next_abs,
                                                        uint32_t* __restrict__ bad, uint16_t* __restrict__ rel)
{
	__shared__ uint32_t lds_hist[1002];   // (1000, 1001: the wave's block of the min_baseq list, bq_append)
#ifndef NGSQC_NO_WAVE_LISTS
	__shared__ WaveList wl_long, wl_pile;
	if (threadIdx.x == 0) { wl_long.n = 0; wl_pile.n = 0; }
#endif
	if (threadIdx.x < 2) lds_hist[1000 + threadIdx.x] = 0;
	for (int i = threadIdx.x; i < 1000; i += blockDim.x) lds_hist[i] = 0;
	__syncthreads();
	Acc a; for (int i = 0; i < A_COUNT; ++i) { a.v[i] = 0; a.n[i] = 0; } a.max_len = 0; a.best_key = 0; a.first_paired = ~0ull;
	const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b < n_entries)
	{
		int64_t lo, hi; entry_range(blocks, b, prefix, ksh, nm, lo, hi);
		const int32_t s = start[b];   // (>= 0, or -1: nothing starts here; the guess kernel has resolved every -2)
		if (s < 0) { cnt[b] = 0; next_abs[b] = -1; }
		else
		{
			const uint32_t stride = rel_stride(ksh);
			uint64_t* const rel8 = (uint64_t*)(rel + b * stride);   // (16-byte aligned: stride is a multiple of 8)
			uint64_t pk_lo = 0, pk_hi = 0;                           // the last eight offsets, oldest in the low bits
			int64_t o = lo + s; uint32_t n = 0; int64_t res = 0; bool stop = false;
			Hdr nx; nx.bs = 0; nx.tid = nx.pos = nx.w = nx.w2 = nx.l_seq = nx.isize = 0;
			if (o < hi) nx = load_hdr(p.infl + o);
			while (o < hi)
			{
				const uint32_t bs = nx.bs;
				if (o + 4 > p.total) { res = -(o + 10); stop = true; break; }
				if (bs < 32) { res = -2; stop = true; break; }
				if (o + 4 + (int64_t)bs > p.total) { res = -(o + 10); stop = true; break; }
				const int64_t o_next = o + 4 + (int64_t)bs;
				RecView r = make_rec(p.infl, o, nx);   // (bs >= 32 and the record ends inside the tile: all 36 bytes were loaded)
				if (!record_fields_fit(r.l_name, r.n_cigar_raw, r.l_seq, r.bs)) { res = -2; stop = true; break; }
				// this record's CIGAR is requested BEFORE the next record's header: loads return in order as far as s_waitcnt vmcnt can tell, so the other way round
				// the wait for the CIGAR (a line that is mostly here already) was a wait for the next header too - an HBM miss per record that nothing overlapped
				uint32_t c4[4]; load_cigar4(r, c4);
				// (unconditional - the last record of the entry asks for its own header again - so that the compiler knows three loads follow the CIGAR's and waits for the
				// CIGAR with s_waitcnt vmcnt(3). o_next < total: the 36 bytes lie inside the tile buffer, which ends with 64 spare bytes; what lies behind total is never used)
				__builtin_amdgcn_sched_barrier(0);   // (the scheduler must not move the header's loads in front of the CIGAR's: see above)
				nx = load_hdr(p.infl + (o_next < hi ? o_next : o));
				__builtin_amdgcn_sched_barrier(0);
				if (n >= (1u << NAME_SHIFT) - 1u) { res = -3; stop = true; break; }   // more records than a name holds (a group of members of a short-read file): not for this path
				if (n < stride)
				{
					pk_lo = (pk_lo >> 16) | (pk_hi << 48); pk_hi = (pk_hi >> 16) | ((uint64_t)(uint16_t)(o - lo) << 48);
					if ((n & 7u) == 7u) { rel8[(n >> 3) * 2] = pk_lo; rel8[(n >> 3) * 2 + 1] = pk_hi; }
				}
				if (o < p.scan_limit)
				{
					const long long name = (long long)((b << NAME_SHIFT) | (int64_t)n);
					long long ref_len = 0;
					const bool scanned = scan_record<MODE>(p, r, name, a, lds_hist, &ref_len, c4);
#ifdef NGSQC_NO_WAVE_LISTS
					if (!scanned && p.sgn > 0) { unsigned long long k = atomicAdd(&p.counters[A_LONG_COUNT], 1ull); if ((long long)k < p.long_cap) p.long_list[k] = name; }
					if (p.pile.list && p.sgn > 0) pile_candidate(p, r, o, scanned, ref_len, a, nullptr);
#else
					if (!scanned && p.sgn > 0) wave_list_append(WAVE_LIST_P(wl_long), p.long_list, &p.counters[A_LONG_COUNT], p.long_cap, (unsigned long long)name);
					if (p.pile.list && p.sgn > 0) pile_candidate(p, r, o, scanned, ref_len, a, WAVE_LIST_P(wl_pile));
#endif
				}
				++n; o = o_next;
			}
			if ((n & 7u) != 0 && n < stride)
			{
				// the last, partial group of eight: moved down to the low bits
				for (uint32_t k = n & 7u; k < 8u; ++k) { pk_lo = (pk_lo >> 16) | (pk_hi << 48); pk_hi >>= 16; }
				rel8[(n >> 3) * 2] = pk_lo; rel8[(n >> 3) * 2 + 1] = pk_hi;
			}
			cnt[b] = n; next_abs[b] = stop ? res : o;
			if (stop && res == -2) atomicAdd(bad, 1u);
		}
	}
	if (MODE == 3 && p.bq_list && p.sgn > 0) bq_close(p, lds_hist + 1000);
#ifndef NGSQC_NO_WAVE_LISTS
	if (p.sgn > 0) { wave_list_close(WAVE_LIST_P(wl_long), p.long_list, &p.counters[A_LONG_COUNT], p.long_cap); if (p.pile.list) wave_list_close(WAVE_LIST_P(wl_pile), p.pile.list, p.pile.count, p.pile.cap); }
#endif
	flush(p, a, lds_hist);
}

template <int WAVES>
static void launch_walk_scan_w(const ScanParams& p, const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, int32_t* d_start, uint32_t* d_cnt, int64_t* d_next_abs, uint32_t* d_bad, uint16_t* d_rel, hipStream_t s)
{
	const int grid = (int)((n_entries + 63) / 64);
	switch (p.mode)
	{
		case NGSQC_MODE_ROI: hipLaunchKernelGGL((walk_scan_kernel<NGSQC_MODE_ROI, WAVES>), dim3(grid), dim3(64), 0, s, p, d_blocks, n_entries, prefix, ksh, nm, d_start, d_cnt, d_next_abs, d_bad, d_rel); break;
		case NGSQC_MODE_NOROI: hipLaunchKernelGGL((walk_scan_kernel<NGSQC_MODE_NOROI, WAVES>), dim3(grid), dim3(64), 0, s, p, d_blocks, n_entries, prefix, ksh, nm, d_start, d_cnt, d_next_abs, d_bad, d_rel); break;
		case NGSQC_MODE_WGS: hipLaunchKernelGGL((walk_scan_kernel<NGSQC_MODE_WGS, WAVES>), dim3(grid), dim3(64), 0, s, p, d_blocks, n_entries, prefix, ksh, nm, d_start, d_cnt, d_next_abs, d_bad, d_rel); break;
		case MODE_COUNT: hipLaunchKernelGGL((walk_scan_kernel<MODE_COUNT, WAVES>), dim3(grid), dim3(64), 0, s, p, d_blocks, n_entries, prefix, ksh, nm, d_start, d_cnt, d_next_abs, d_bad, d_rel); break;
		default: hipLaunchKernelGGL((walk_scan_kernel<3, WAVES>), dim3(grid), dim3(64), 0, s, p, d_blocks, n_entries, prefix, ksh, nm, d_start, d_cnt, d_next_abs, d_bad, d_rel); break;
	}
	KCHECK();
}
// NGSQC_WALK_WAVES = 3 / 4: the register budget the walk is compiled for (waves per SIMD: 170 / 128 VGPRs)
void launch_walk_scan(const ScanParams& p, const BlockDesc* d_blocks, int64_t n_entries, int64_t prefix, int ksh, int64_t nm, int32_t* d_start, uint32_t* d_cnt, int64_t* d_next_abs, uint32_t* d_bad, uint16_t* d_rel, hipStream_t s)
{
	if (n_entries <= 0) return;
	// MODE_DEPTH (the coverage tools) is compiled for FIVE waves per SIMD (95 VGPRs, no scratch): 5 x 4 SIMDs x 256 CUs x 64 lanes = 327 680 walkers are resident at once,
	// which is exactly a tile of four K1 chunks - with the 99 VGPRs of the budget of three (four resident waves, 262 144 walkers) a tile of the 30x file took a second,
	// 23 % full round of workgroups: scan kernels 30.7 ms per step with 17 tiles, 34.9 ms with the 10 tiles the K1 schedule wants (profiles/r06_bench_full_30x.json)
	int waves = p.mode == 3 ? 5 : 3; if (const char* e = getenv("NGSQC_WALK_WAVES")) waves = atoi(e);
	if (waves >= 5 && p.mode == 3) { const int grid = (int)((n_entries + 63) / 64); hipLaunchKernelGGL((walk_scan_kernel<3, 5>), dim3(grid), dim3(64), 0, s, p, d_blocks, n_entries, prefix, ksh, nm, d_start, d_cnt, d_next_abs, d_bad, d_rel); KCHECK(); }
	else if (waves >= 4) launch_walk_scan_w<4>(p, d_blocks, n_entries, prefix, ksh, nm, d_start, d_cnt, d_next_abs, d_bad, d_rel, s);
	else launch_walk_scan_w<3>(p, d_blocks, n_entries, prefix, ksh, nm, d_start, d_cnt, d_next_abs, d_bad, d_rel, s);
}

// ---- order-dependent fix-ups on a record prefix ----
// A_FIX_TRIM += sum over counted records with (tile-local) ordinal < upto_max of running_max_i, A_FIX_CNT += their number
// A_FIX_LEN  += sum over "passing" records with ordinal < upto_paired of length               [single workgroup, sequential chunks]
// The running maximum starts at A_FIX_CARRY (records before this prefix) and is left there for the next call.
// what the order-dependent fix-ups need of a record: l_seq | counted << 30 | passing << 31
__device__ __forceinline__ uint32_t head_word(const ScanParams& p, long long ord)
{
	RecView r = load_rec(p.infl, p.recoff[ord]);
	const uint32_t flag = r.flag;
	const bool counted = !(flag & 0x100) && !(flag & 0x800);
	if (!counted) return 0u;
	const bool tid_ok = r.tid >= 0 && r.tid < p.n_ref;
	const bool passing = !(flag & 0x4) && tid_ok && p.tid_nonspecial[r.tid] && !(flag & 0x400) && (int)r.mapq >= p.min_mapq;
	return ((uint32_t)r.l_seq & 0x3fffffffu) | (1u << 30) | (passing ? 1u << 31 : 0u);
}
// the first n records of a shard in that form: ngsqc_scan_mapping_finish then works without the inflated bytes
__global__ void prefix_capture_kernel(const ScanParams p, long long n, uint32_t* __restrict__ out)
{
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = head_word(p, i);
}

__global__ __launch_bounds__(256) void prefix_fix_kernel(const ScanParams p, long long upto_max, long long upto_paired, const uint32_t* __restrict__ head)
{
	__shared__ int sh[256]; __shared__ int carry;
	if (threadIdx.x == 0) carry = (int)p.counters[A_FIX_CARRY];   // running maximum carried in from earlier tiles
	__syncthreads();
	long long upto = upto_max > upto_paired ? upto_max : upto_paired;   // tile-local limits
	long long s_trim = 0, s_len = 0, s_cnt = 0;
	for (long long base = 0; base < upto; base += 256)
	{
		long long ord = base + threadIdx.x;
		int len = 0; bool counted = false, passing = false;
		if (ord < upto)
		{
			const uint32_t w = head ? head[ord] : head_word(p, ord);
			counted = (w >> 30) & 1u; passing = w >> 31; len = (int)(w & 0x3fffffffu);
		}
		sh[threadIdx.x] = counted ? len : 0; __syncthreads();
		for (int d = 1; d < 256; d <<= 1) { int t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0; __syncthreads(); sh[threadIdx.x] = max(sh[threadIdx.x], t); __syncthreads(); }
		int run = max(carry, sh[threadIdx.x]);
		if (counted && ord < upto_max) { s_trim += run; ++s_cnt; }
		if (passing && ord < upto_paired) s_len += len;
		__syncthreads();
		if (threadIdx.x == 255) carry = run;
		__syncthreads();
	}
	if (threadIdx.x == 0) p.counters[A_FIX_CARRY] = (unsigned long long)carry;
	s_trim = wave_sum(s_trim); s_len = wave_sum(s_len); s_cnt = wave_sum(s_cnt);
	if ((threadIdx.x & 63) == 0)
	{
		if (s_trim) atomicAdd(&p.counters[A_FIX_TRIM], (unsigned long long)s_trim);
		if (s_len) atomicAdd(&p.counters[A_FIX_LEN], (unsigned long long)s_len);
		if (s_cnt) atomicAdd(&p.counters[A_FIX_CNT], (unsigned long long)s_cnt);
	}
}

static int scan_grid(long long n, int per_wg)
{
	long long wgs = (n + per_wg - 1) / per_wg;
	long long cap = 256 * 8;   // (more workgroups were tried for the wave-per-record kernels in round 5: every wave ends with atomics on the same counters - 150 k waves took 1.5 ms longer than 8 k)
	return (int)(wgs < 1 ? 1 : (wgs < cap ? wgs : cap));
}

void launch_scan(const ScanParams& p, hipStream_t s)
{
	if (p.n_rec <= 0) return;
	int grid = scan_grid(p.n_rec, 256);
	switch (p.mode)
	{
		case NGSQC_MODE_ROI: hipLaunchKernelGGL(scan_kernel<NGSQC_MODE_ROI>, dim3(grid), dim3(256), 0, s, p); break;
		case NGSQC_MODE_NOROI: hipLaunchKernelGGL(scan_kernel<NGSQC_MODE_NOROI>, dim3(grid), dim3(256), 0, s, p); break;
		case NGSQC_MODE_WGS: hipLaunchKernelGGL(scan_kernel<NGSQC_MODE_WGS>, dim3(grid), dim3(256), 0, s, p); break;
		case MODE_COUNT: hipLaunchKernelGGL(scan_kernel<MODE_COUNT>, dim3(grid), dim3(256), 0, s, p); break;
		default: hipLaunchKernelGGL(scan_kernel<3>, dim3(grid), dim3(256), 0, s, p); break;
	}
	KCHECK();
}

void launch_scan_long(const ScanParams& p, int64_t n_long, hipStream_t s)
{
	if (n_long <= 0) return;
	int grid = scan_grid(n_long, 4);
	switch (p.mode)
	{
		case NGSQC_MODE_ROI: hipLaunchKernelGGL(scan_long_kernel<NGSQC_MODE_ROI>, dim3(grid), dim3(256), 0, s, p, (long long)n_long); break;
		case NGSQC_MODE_NOROI: hipLaunchKernelGGL(scan_long_kernel<NGSQC_MODE_NOROI>, dim3(grid), dim3(256), 0, s, p, (long long)n_long); break;
		case NGSQC_MODE_WGS: hipLaunchKernelGGL(scan_long_kernel<NGSQC_MODE_WGS>, dim3(grid), dim3(256), 0, s, p, (long long)n_long); break;
		case MODE_COUNT: hipLaunchKernelGGL(scan_long_kernel<MODE_COUNT>, dim3(grid), dim3(256), 0, s, p, (long long)n_long); break;
		default: hipLaunchKernelGGL(scan_long_kernel<3>, dim3(grid), dim3(256), 0, s, p, (long long)n_long); break;
	}
	KCHECK();
}

// ---- the same fix-up in parallel (round 5: the single workgroup above took 1.1 ms for the 150 k records in front of an ONT tile's longest read). Three launches:
// head words + the longest counted read of every block of 1024 records; the running maximum in front of every block (one workgroup over the block maxima, seeded
// with A_FIX_CARRY, which then moves on to the overall maximum); per block the running maximum of every record from that seed and the three sums. ----
constexpr int PF_T = 256, PF_ITEMS = 4, PF_BLK = PF_T * PF_ITEMS;
__global__ __launch_bounds__(PF_T) void prefix_fix_max_kernel(const ScanParams p, long long upto, const uint32_t* __restrict__ head, uint32_t* __restrict__ words, int32_t* __restrict__ block_max)
{
	__shared__ int sh[PF_T];
	const long long base = (long long)blockIdx.x * PF_BLK + (long long)threadIdx.x * PF_ITEMS;
	int m = 0;
	#pragma unroll
	for (int i = 0; i < PF_ITEMS; ++i)
	{
		const long long ord = base + i;
		if (ord < upto) { const uint32_t w = head ? head[ord] : head_word(p, ord); words[ord] = w; if ((w >> 30) & 1u) m = max(m, (int)(w & 0x3fffffffu)); }
	}
	sh[threadIdx.x] = m; __syncthreads();
	for (int d = PF_T / 2; d > 0; d >>= 1) { if ((int)threadIdx.x < d) sh[threadIdx.x] = max(sh[threadIdx.x], sh[threadIdx.x + d]); __syncthreads(); }
	if (threadIdx.x == 0) block_max[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void prefix_fix_seed_kernel(const ScanParams p, int32_t* block_max, long long n_blocks)   // block_max[b] -> maximum in front of block b (carry included)
{
	__shared__ int sh[256]; __shared__ int carry;
	if (threadIdx.x == 0) carry = (int)p.counters[A_FIX_CARRY];
	__syncthreads();
	for (long long base = 0; base < n_blocks; base += 256)
	{
		const long long j = base + threadIdx.x;
		const int v = j < n_blocks ? block_max[j] : 0;
		sh[threadIdx.x] = v; __syncthreads();
		for (int d = 1; d < 256; d <<= 1) { const int t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0; __syncthreads(); sh[threadIdx.x] = max(sh[threadIdx.x], t); __syncthreads(); }
		const int incl = max(carry, sh[threadIdx.x]);
		const int excl = threadIdx.x ? max(carry, sh[threadIdx.x - 1]) : carry;
		if (j < n_blocks) block_max[j] = excl;
		__syncthreads();
		if (threadIdx.x == 255) carry = incl;
		__syncthreads();
	}
	if (threadIdx.x == 0) p.counters[A_FIX_CARRY] = (unsigned long long)carry;
}
__global__ __launch_bounds__(PF_T) void prefix_fix_sum_kernel(const ScanParams p, long long upto_max, long long upto_paired, const uint32_t* __restrict__ words, const int32_t* __restrict__ block_seed)
{
	__shared__ int sh[PF_T];
	const long long upto = upto_max > upto_paired ? upto_max : upto_paired;
	const long long base = (long long)blockIdx.x * PF_BLK + (long long)threadIdx.x * PF_ITEMS;
	uint32_t w[PF_ITEMS]; int m = 0;
	#pragma unroll
	for (int i = 0; i < PF_ITEMS; ++i) { const long long ord = base + i; w[i] = ord < upto ? words[ord] : 0u; if ((w[i] >> 30) & 1u) m = max(m, (int)(w[i] & 0x3fffffffu)); }
	sh[threadIdx.x] = m; __syncthreads();
	for (int d = 1; d < PF_T; d <<= 1) { const int t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0; __syncthreads(); sh[threadIdx.x] = max(sh[threadIdx.x], t); __syncthreads(); }
	int run = max(block_seed[blockIdx.x], threadIdx.x ? sh[threadIdx.x - 1] : 0);   // the running maximum in front of this thread's first record
	long long s_trim = 0, s_len = 0, s_cnt = 0;
	#pragma unroll
	for (int i = 0; i < PF_ITEMS; ++i)
	{
		const long long ord = base + i;
		const bool counted = (w[i] >> 30) & 1u, passing = w[i] >> 31; const int len = (int)(w[i] & 0x3fffffffu);
		if (counted) run = max(run, len);
		if (counted && ord < upto_max) { s_trim += run; ++s_cnt; }
		if (passing && ord < upto_paired) s_len += len;
	}
	s_trim = wave_sum(s_trim); s_len = wave_sum(s_len); s_cnt = wave_sum(s_cnt);
	if ((threadIdx.x & 63) == 0)
	{
		if (s_trim) atomicAdd(&p.counters[A_FIX_TRIM], (unsigned long long)s_trim);
		if (s_len) atomicAdd(&p.counters[A_FIX_LEN], (unsigned long long)s_len);
		if (s_cnt) atomicAdd(&p.counters[A_FIX_CNT], (unsigned long long)s_cnt);
	}
}
size_t prefix_fix_scratch_words(int64_t upto) { return upto > 4096 ? (size_t)upto + (size_t)((upto + PF_BLK - 1) / PF_BLK) + 64 : 0; }

void launch_prefix_fix(const ScanParams& p, int64_t upto_max, int64_t upto_paired, const uint32_t* d_head, hipStream_t s, uint32_t* d_scratch)
{
	if (upto_max <= 0 && upto_paired <= 0) return;
	const int64_t upto = std::max(upto_max, upto_paired);
	if (!d_scratch || upto <= 4096) { hipLaunchKernelGGL(prefix_fix_kernel, dim3(1), dim3(256), 0, s, p, (long long)upto_max, (long long)upto_paired, d_head); KCHECK(); return; }
	const int64_t nb = (upto + PF_BLK - 1) / PF_BLK;
	uint32_t* words = d_scratch; int32_t* bmax = (int32_t*)(d_scratch + upto);
	hipLaunchKernelGGL(prefix_fix_max_kernel, dim3((int)nb), dim3(PF_T), 0, s, p, (long long)upto, d_head, words, bmax); KCHECK();
	hipLaunchKernelGGL(prefix_fix_seed_kernel, dim3(1), dim3(256), 0, s, p, bmax, (long long)nb); KCHECK();
	hipLaunchKernelGGL(prefix_fix_sum_kernel, dim3((int)nb), dim3(PF_T), 0, s, p, (long long)upto_max, (long long)upto_paired, words, bmax); KCHECK();
}
void launch_prefix_capture(const ScanParams& p, int64_t n, uint32_t* d_head, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(prefix_capture_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, s, p, (long long)n, d_head); KCHECK();
}

// ---- site pileup: BamReader::getPileup (src/cppNGS/BamReader.cpp:809-885, SNP counts) for a table of known sites ----
// The reference runs one indexed query per site (10^5 for Statistics::contamination, Statistics.cpp:2333-2386) and extracts
// the base of every overlapping read with BamAlignment::extractBaseByCIGAR (BamReader.cpp:307-374). Here every record
// looks up the sites inside its reference span (binary search in the per-tid sorted site table: a short read meets a
// site with probability ~0.5 %) and walks its CIGAR once per hit. counts[site][0..5] = A, C, G, T, N, deletion;
// [6] = a base Pileup::inc would throw on (other IUPAC codes), [7] = a position the CIGAR walk could not find.
// (single exit, no early returns inside the loops: the straightforward version with returns was miscompiled by hipcc 7.2 -
// three out of four reads came back as "nothing to count")
__device__ static int pileup_base(const RecView& r, int pos, int& qual_out)   // 0..4 = A,C,G,T,N; 5 = deletion; 6 = other letter; -1 = nothing to count; -2 = not found
{
	uint32_t other_ops = 0;                  // cigarIsOnlyInsertion looks at the CORE cigar (BamReader.cpp:90-100)
	for (uint32_t k = 0; k < r.n_cigar_raw; ++k) { const uint32_t op = ld32(r.core + 32 + r.l_name + 4ull * k) & 15u; other_ops |= (op != 1u && op != 4u) ? 1u : 0u; }
	int res = other_ops ? -2 : -1, q = -1;
	bool done = other_ops == 0;
	int read_pos = 0, genome_pos = r.pos;   // start() - 1
	for (uint32_t k = 0; k < r.n_cigar; ++k)
	{
		if (!done)
		{
			const uint32_t c = ld32(r.cigar + 4ull * k); const uint32_t op = c & 15u; const int len = (int)(c >> 4);
			const bool ref_op = op == 0 || op == 2 || op == 3 || op == 7 || op == 8, read_op = op == 0 || op == 1 || op == 4 || op == 7 || op == 8;
			genome_pos += ref_op ? len : 0; read_pos += read_op ? len : 0;
			if (op == 2 && genome_pos >= pos) { res = 5; q = 255; done = true; }
			else if (op == 3 && genome_pos >= pos) { res = -1; done = true; }
			else if (op == 4 && read_pos >= r.l_seq) { res = -1; done = true; }
			else if (op == 6 || op > 8) { res = -2; done = true; }
			else if (genome_pos >= pos)
			{
				const int ap = read_pos - (genome_pos + 1 - pos);
				if (ap < 0 || ap >= r.l_seq) { res = -2; done = true; }   // CIGAR longer than SEQ
				else
				{
					const uint8_t* seq = r.core + 32 + r.l_name + 4ull * r.n_cigar_raw;
					const int nib = (seq[ap >> 1] >> ((~ap & 1) << 2)) & 15;
					q = rec_qual(r)[ap];
					res = nib == 1 ? 0 : nib == 2 ? 1 : nib == 4 ? 2 : nib == 8 ? 3 : nib == 15 ? 4 : 6;
					done = true;
				}
			}
		}
	}
	qual_out = q;
	return res;
}

__global__ __launch_bounds__(256) void pileup_kernel(const uint8_t* __restrict__ infl, const int64_t* __restrict__ recoff, long long n_rec, int n_ref,
                                                     const int32_t* __restrict__ site_pos, const int32_t* __restrict__ tid_first, const int32_t* __restrict__ tid_last,
                                                     const int32_t* __restrict__ bucket, const int64_t* __restrict__ tid_bucket0,
                                                     int min_mapq, int min_baseq, int include_npp, uint32_t* __restrict__ counts,
                                                     int64_t* __restrict__ long_list, unsigned long long* __restrict__ long_count)
{
	for (long long li = (long long)blockIdx.x * blockDim.x + threadIdx.x; li < n_rec; li += (long long)gridDim.x * blockDim.x)
	{
		RecView r = load_rec(infl, recoff[li]);
		const uint32_t flag = r.flag;
		if (flag & (0x100 | 0x800 | 0x400 | 0x4)) continue;                 // BamReader.cpp:830
		if (!(flag & 0x2) && !include_npp) continue;                         // :831
		if ((int)r.mapq < min_mapq) continue;                                // :836
		if (r.tid < 0 || r.tid >= n_ref) continue;
		const int first = tid_first[r.tid], last = tid_last[r.tid];
		if (first >= last) continue;
		// CG:B,I long CIGAR (htslib bam_tag2cigar)
		if (r.n_cigar_raw > 0 && r.pos >= 0)
		{
			const uint32_t c0 = ld32(r.cigar);
			if ((c0 & 15u) == 4 && (int32_t)(c0 >> 4) == r.l_seq)
			{
				const uint8_t* t = aux_find(rec_aux(r), rec_end(r), 'C', 'G');
				if (t && t[0] == 'B' && t[1] == 'I') { const uint32_t n = ld32(t + 2); if (n >= r.n_cigar_raw && n < (1u << 29)) { r.cigar = t + 6; r.n_cigar = n; } }
			}
		}
		if (r.n_cigar > (uint32_t)LONG_CIGAR) { long_list[atomicAdd(long_count, 1ull)] = li; continue; }   // wave-per-record path (pileup_long_kernel)
		long long ref_len = 0;
		for (uint32_t k = 0; k < r.n_cigar; ++k) { const uint32_t c = ld32(r.cigar + 4ull * k); if ((0x18Du >> (c & 15u)) & 1u) ref_len += c >> 4; }
		if (ref_len == 0) ref_len = 1;                                        // bam_endpos
		const int start1 = r.pos + 1, end1 = (int)(r.pos + ref_len);
		// first site with pos >= start1: bucket[tid][start1 >> 16] = first site at or behind the bucket's first position, then a short scan
		const int64_t b0 = tid_bucket0[r.tid], nbk = tid_bucket0[r.tid + 1] - b0;
		int64_t bi = start1 > 0 ? (int64_t)(start1 >> PILEUP_BUCKET_SHIFT) : 0; if (bi >= nbk) bi = nbk - 1;
		int a = bucket[b0 + bi];
		while (a < last && site_pos[a] < start1) ++a;
		for (int i = a; i < last && site_pos[i] <= end1; ++i)
		{
			int q; const int base = pileup_base(r, site_pos[i], q);
			if (base == -2) atomicAdd(&counts[8ull * i + 7], 1u);
			else if (base >= 0 && q >= min_baseq) atomicAdd(&counts[8ull * i + base], 1u);
		}
	}
}

// Records with long CIGARs (ONT: ~1 op per 12 bp, thousands of ops): ONE WAVE PER RECORD, the CIGAR streamed in file order (round 5; rounds 2-4 gave every lane a
// contiguous slice of the operations to walk by itself: a 500 kb read of 40 000 operations was 2 x 625 dependent loads per lane, and one such record set the
// kernel's time - 5.8 ms per tile of the ONT-like shard). Pass 1: the reference length (four operations per lane and step, 16-byte loads) gives the span and with it
// the known sites inside the read - none for four reads in five. Pass 2: 64 operations per step, one per lane, coalesced; a wave prefix sum gives every operation
// the genome / read position behind it; the sites are met in ascending order, each at the first operation that carries the genome position to or past it, exactly
// where the sequential walk stops (BamReader.cpp:307-374): a deletion counts as deletion, a skip as nothing, anything else gives the base; a soft clip that
// exhausts the read in front of that operation ends the walk for this and every later site (:346-353). The stream ends behind the last site of the span.
__global__ __launch_bounds__(256) void pileup_long_kernel(const uint8_t* __restrict__ infl, const int64_t* __restrict__ recoff, const int64_t* __restrict__ long_list, const unsigned long long* __restrict__ n_long_dev,
                                                          const int32_t* __restrict__ site_pos, const int32_t* __restrict__ tid_last,
                                                          const int32_t* __restrict__ bucket, const int64_t* __restrict__ tid_bucket0,
                                                          int min_baseq, uint32_t* __restrict__ counts)
{
	const int lane = threadIdx.x & 63;
	const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
	const long long n_long = (long long)*n_long_dev;   // (round 5: the count stays on the device - the host does not wait for pileup_kernel to learn it)
	for (long long w = wave; w < n_long; w += n_waves)
	{
		RecView r = load_rec(infl, recoff[long_list[w]]);      // (passed the read filters in pileup_kernel)
		if (r.n_cigar_raw > 0 && r.pos >= 0)
		{
			const uint32_t c0 = ld32(r.cigar);
			if ((c0 & 15u) == 4 && (int32_t)(c0 >> 4) == r.l_seq)
			{
				unsigned long long cg = 0; uint32_t n = 0;
				if (lane == 0)
				{
					const uint8_t* t = aux_find(rec_aux(r), rec_end(r), 'C', 'G');
					if (t && t[0] == 'B' && t[1] == 'I') { n = ld32(t + 2); if (n >= r.n_cigar_raw && n < (1u << 29)) cg = (unsigned long long)(uintptr_t)(t + 6); }
				}
				cg = __shfl(cg, 0); n = __shfl(n, 0);
				if (cg) { r.cigar = (const uint8_t*)(uintptr_t)cg; r.n_cigar = n; }
			}
		}
		{
			uint32_t other = 0;
			for (uint32_t k = lane; k < r.n_cigar_raw; k += 64) { const uint32_t op = ld32(r.core + 32 + r.l_name + 4ull * k) & 15u; other |= (op != 1u && op != 4u) ? 1u : 0u; }
			if (__builtin_amdgcn_ballot_w64(other != 0) == 0) continue;   // insertion / soft-clip only: the reference skips the read (BamReader.cpp:845)
		}
		// ---- pass 1: the reference length ----
		long long ref_len = 0;
		for (uint32_t k0 = 4u * lane; k0 < r.n_cigar; k0 += 256u)
		{
			uint32_t c4[4] = {0u, 0u, 0u, 0u};
			if (k0 + 4u <= r.n_cigar) __builtin_memcpy(c4, r.cigar + 4ull * k0, 16);
			else for (uint32_t j = 0; k0 + j < r.n_cigar; ++j) c4[j] = ld32(r.cigar + 4ull * (k0 + j));
			#pragma unroll
			for (uint32_t j = 0; j < 4u; ++j) if (k0 + j < r.n_cigar && ((0x18Du >> (c4[j] & 15u)) & 1u)) ref_len += c4[j] >> 4;
		}
		ref_len = wave_sum(ref_len);
		const bool no_ref_ops = ref_len == 0;
		if (ref_len == 0) ref_len = 1;   // bam_endpos
		const int start1 = r.pos + 1, end1 = (int)(r.pos + ref_len);
		const int last = tid_last[r.tid];
		const int64_t b0 = tid_bucket0[r.tid], nbk = tid_bucket0[r.tid + 1] - b0;
		int64_t bi = start1 > 0 ? (int64_t)(start1 >> PILEUP_BUCKET_SHIFT) : 0; if (bi >= nbk) bi = nbk - 1;
		int cur = bucket[b0 + bi];
		while (cur < last && site_pos[cur] < start1) ++cur;
		int site_end = cur; while (site_end < last && site_pos[site_end] <= end1) ++site_end;
		if (cur >= site_end) continue;
		if (no_ref_ops) { if (lane == 0) for (int i = cur; i < site_end; ++i) atomicAdd(&counts[8ull * i + 7], 1u); continue; }   // "Could not find position": no operation ever reaches it
		// ---- pass 2: the operations in order ----
		long long g_base = r.pos, rp_base = 0; bool s_seen = false;   // genome position (0-based start - 1 + consumed = 1-based position of the last consumed base) and read position behind the chunks so far
		for (uint32_t k0 = 0; k0 < r.n_cigar && cur < site_end && !s_seen; k0 += 64u)
		{
			const uint32_t k = k0 + (uint32_t)lane;
			uint32_t op = 15u; long long len = 0;
			if (k < r.n_cigar) { const uint32_t c = ld32(r.cigar + 4ull * k); op = c & 15u; len = c >> 4; }
			const bool ref_op = (0x18Du >> op) & 1u, read_op = op == 0 || op == 1 || op == 4 || op == 7 || op == 8;
			long long g = ref_op ? len : 0, rp = read_op ? len : 0;
			#pragma unroll
			for (int o = 1; o < 64; o <<= 1) { const long long a = __shfl_up(g, o), b = __shfl_up(rp, o); if (lane >= o) { g += a; rp += b; } }
			g += g_base; rp += rp_base;   // positions BEHIND this lane's operation
			const uint64_t m_s = __builtin_amdgcn_ballot_w64(op == 4u && rp >= (long long)r.l_seq);   // soft clips that exhaust the read
			while (cur < site_end)
			{
				const int pos = site_pos[cur];
				const uint64_t m_hit = __builtin_amdgcn_ballot_w64(ref_op && g >= (long long)pos);
				if (!m_hit) break;                                   // this site lies behind the chunk
				const int l = __builtin_ctzll(m_hit);
				if (m_s & ((1ull << l) - 1ull)) { s_seen = true; break; }   // the walk ended at a soft clip in front of the operation
				if (lane == l)
				{
					if (op == 2u) atomicAdd(&counts[8ull * cur + 5], 1u);                                            // deleted base, quality 255
					else if (op != 3u)
					{
						const long long ap = rp - (g + 1 - pos);
						if (ap < 0 || ap >= r.l_seq) atomicAdd(&counts[8ull * cur + 7], 1u);   // CIGAR longer than SEQ
						else
						{
							const uint8_t* seq = r.core + 32 + r.l_name + 4ull * r.n_cigar_raw;
							const int nib = (seq[ap >> 1] >> ((~ap & 1) << 2)) & 15, q = rec_qual(r)[ap];
							const int base = nib == 1 ? 0 : nib == 2 ? 1 : nib == 4 ? 2 : nib == 8 ? 3 : nib == 15 ? 4 : 6;
							if (q >= min_baseq) atomicAdd(&counts[8ull * cur + base], 1u);
						}
					}
				}
				++cur;
			}
			if (m_s) s_seen = true;   // (a later site's operation lies behind it)
			g_base = __shfl(g, 63); rp_base = __shfl(rp, 63);
		}
	}
}

void launch_pileup(const uint8_t* infl, const int64_t* recoff, int64_t n_rec, int n_ref, const int32_t* site_pos, const int32_t* tid_first, const int32_t* tid_last,
                   const int32_t* bucket, const int64_t* tid_bucket0, int min_mapq, int min_baseq, int include_npp, uint32_t* counts,
                   int64_t* long_list, unsigned long long* long_count, hipStream_t s)
{
	if (n_rec <= 0) return;
	const int grid = (int)std::min<int64_t>((n_rec + 255) / 256, 256 * 32);
	hipLaunchKernelGGL(pileup_kernel, dim3(grid), dim3(256), 0, s, infl, recoff, (long long)n_rec, n_ref, site_pos, tid_first, tid_last, bucket, tid_bucket0, min_mapq, min_baseq, include_npp, counts, long_list, long_count); KCHECK();
}
void launch_pileup_long(const uint8_t* infl, const int64_t* recoff, const int64_t* long_list, const unsigned long long* d_n_long, int64_t n_long_max, const int32_t* site_pos, const int32_t* tid_last,
                        const int32_t* bucket, const int64_t* tid_bucket0, int min_baseq, uint32_t* counts, hipStream_t s)
{
	if (n_long_max <= 0) return;
	const int grid = (int)std::min<int64_t>((n_long_max + 3) / 4, 256 * 16);   // (sized for the most there can be; the waves stride over what there is)
	hipLaunchKernelGGL(pileup_long_kernel, dim3(grid), dim3(256), 0, s, infl, recoff, long_list, d_n_long, site_pos, tid_last, bucket, tid_bucket0, min_baseq, counts); KCHECK();
}

} // namespace ngsqc
