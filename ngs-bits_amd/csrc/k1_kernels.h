// K1 - BGZF inflate in two phases (the kernels; inflate.hip holds the launchers).
//
//   phase 1  huff_tokens_kernel : ONE LANE PER BGZF MEMBER. Every lane Huffman-decodes its own raw-DEFLATE stream into tokens
//            (literal | match{len,dist} | raw run of a stored block); 64 members advance per wave instruction.
//            * The symbol loop is BRANCH-FREE: a trip decodes up to two literal/length symbols (the second one speculatively, it
//              only counts when the first is a literal) and one distance, every lane runs the same instructions, lanes that
//              cannot take part (header states, input not staged) are masked by selects. Header parsing, member changes and
//              stored blocks run in a separate slow section that the wave enters when enough lanes wait for it.
//            * Canonical Huffman decode runs out of REGISTERS: per code length one word {left-aligned code limit | fields}; the
//              length is found by a 4-level binary search whose pivots are picked with v_cndmask; the last pivot the search
//              went right of is the word of the code's own length.
//            * The decoder lanes keep NO table of the literal symbols: a literal leaves as its index in the block's literal
//              table (the literals sorted by (code length, value)), which the header pass writes to the token pool and phase 2
//              applies. LDS per lane: a 32-byte window of the compressed input, 16 words of per-length counts / offsets, the
//              <= 32 length symbols and <= 32 distance symbols in code order - 42 words instead of the 89 a symbol plane took,
//              laid out element-major (word k of lane l at k*64+l: any per-lane access pattern is bank-conflict free).
//            * Bit reader: a bit cursor into the input window; a trip reads the three window words under the cursor and funnels
//              them (v_alignbit) into 64 fresh bits - no bit buffer to shift, no refill state.
//            * Tokens leave in groups of four slots (two trips), one 16-byte store per group, into pages of a pool that the
//              lanes allocate from with one atomic per page.
//   phase 2  lz77_groups_kernel : ONE WAVE PER MEMBER. A batch is 64 token groups (one per lane, <= 256 tokens) cut to <= P2_BMAX bytes: a
//            wave prefix sum places every token; per TOKEN one word says what a byte of it needs to know (literal value from the block's
//            literal table | near / self-overlapping / far match | raw run), and a flag marks the token's last byte. Per 64-byte chunk
//            the flags become a lane mask (ballot) from which every OUTPUT BYTE gets its owner token (v_mbcnt) and, from the token's
//            word, its source; a self-overlapping match is put in periodic form (offset mod distance). Sub-batches of four chunks:
//            pass 1 classifies their bytes and issues every gather that reaches behind the batch (HBM) - one wait - pass 2 resolves the
//            chunks front to back: a source in an earlier chunk comes from the LDS staging bytes, one in the same chunk from the source
//            LANE (iterating only while a lane's source is itself pending). Every byte is written once to LDS and once to HBM.
//
// Written against the wave vocabulary of wave.h only (see there). Integer work, no MFMA. RFC 1951; the reference reaches
// zlib's inflate through htslib's bgzf.c under BamReader::getNextAlignment (src/cppNGS/BamReader.h:386-392).
#pragma once
#include "k1_types.h"

namespace ngsqc { namespace k1 {

// ---------------------------------------------------------------------------------------------------------------- phase 1
// LDS words of a lane (element-major)
constexpr int P1_W_RING = 0;      // compressed input window: word k of the member's piece stream sits in slot k & 7; slots 8, 9 mirror 0, 1, so
                                  // that the three words under a bit cursor are always slots s, s + 1, s + 2
constexpr int P1_W_LINFO = 10;    // per code length l (1..15) of the literal/length code, fields on byte boundaries (the ISA reads them as sub-dword operands): literals n (bits 0..15) | first literal index << 16 | ((first non-literal index - n) & 255) << 24;
                                  // word 0: the same for the all-ones code (see LimTab); during the header passes: counts, then placement cursors
constexpr int P1_W_NONLIT = 26;   // the symbols 256.. of the block in code order, one byte each: (symbol - 256) * 4 - LANE-major, 32 bytes per lane (round 6: the address is one v_and_or; the element-major form cost four VALU per lookup, and the decoder is what the chip's VALU issue pays for)
constexpr int P1_W_DSYM = 34;     // the distance symbols in code order, one byte each: symbol * 4 (same layout)
constexpr int P1_LANE_W = 42;
constexpr int P1_RING_SLOTS = 8;
constexpr int P1_TRIPS = 4;             // trips between two service blocks = the four words of a token group (one word per trip)
#ifndef NGSQC_P1_WAVES_PER_SIMD
#define NGSQC_P1_WAVES_PER_SIMD 3
#endif
constexpr int P1_WAVES_PER_SIMD = NGSQC_P1_WAVES_PER_SIMD;   // register budget of the decoder. Round 6: THREE waves per SIMD = 168 VGPRs and 8 bytes of scratch. The budget of four (128 VGPRs) is out of reach, the compiler then
                                                             // kept 193 VGPRs = TWO waves per SIMD, eight decoder waves per CU whatever the LDS allowed; with three a CU holds the ten its LDS has room for: the full-size step 444 -> 435 ms,
                                                             // K1 wall 430 -> 415 ms, although the kernel alone got slower (9.0 -> 9.6 ms per launch: the two spilled dwords), profiles/r06_schedule_probe.txt
constexpr int P1_TAB_W = 128;     // per workgroup: base | extra bits << 16 of the symbols 256..287 (words 0..31) and the distance symbols (words 32..63); the rest is padding (an index byte of a damaged stream may point behind the tables)
#ifndef NGSQC_P1_LINE_GROUPS
#define NGSQC_P1_LINE_GROUPS 4
#endif
constexpr int P1_LINE_GROUPS = NGSQC_P1_LINE_GROUPS;   // token groups a lane collects in LDS before they leave for the pool as consecutive 16-byte stores: 4 = whole 64-byte lines (a single 16-byte store per
                                                       // lane reached HBM as a partial line: 3.1 x write amplification), 2 = 32-byte sectors and 2 KB less LDS per wave - which is what decides how
                                                       // many resolve / scan waves fit beside the CU's ten decoder waves. Measured at full size (profiles/r06_schedule_probe.txt): 2 -> 437.5 / 440.6 ms per step, un-pipelined inflate 454 ms;
                                                       // 4 -> 441.7 ms, 446 ms: within the spread - whole lines stay (less write traffic)
static_assert(P1_LINE_GROUPS == 2 || P1_LINE_GROUPS == 4, "line staging");
constexpr int P1_STAGE_W = 4 * P1_LINE_GROUPS * 64;
constexpr int P1_LDS_W = P1_LANE_W * 64 + P1_TAB_W + P1_STAGE_W;   // 15 360 B (13 312 with half lines) per one-wave workgroup

enum { S_SYM = 0, S_NEXT = 1, S_HDR = 2, S_P1 = 3, S_P2 = 4, S_RAW = 5, S_FINISH = 6, S_DONE = 7 };   // 1..6: the slow states
constexpr uint32_t TAB_EOB = 0x80000000u, TAB_BAD = 0x40000000u;

struct P1Lds
{
	uint32_t* base; int lane;
	K1_DEV uint8_t* bytes() const { return (uint8_t*)base; }
	K1_DEV uint32_t lane4() const { return (uint32_t)lane * 4u; }
	K1_DEV uint32_t& at(int k) const { return base[k * 64 + lane]; }
	K1_DEV const uint32_t* ring(uint32_t abit) const { return (const uint32_t*)(bytes() + wv::and_or(abit << 3, 0x700u, lane4())); }   // slot (abit >> 5) & 7
	K1_DEV void stage(uint32_t wr, uint32_t a, uint32_t b, uint32_t c, uint32_t d) const   // a piece = four words, wr a multiple of 4
	{
		uint32_t* p = &at((int)(wr & 4u)); p[0] = a; p[64] = b; p[128] = c; p[192] = d;
		if ((wr & 4u) == 0) { at(8) = a; at(9) = b; }
	}
	K1_DEV uint32_t& linfo(uint32_t l) const { return at(P1_W_LINFO + (int)(l & 15u)); }
	K1_DEV uint32_t linfo_of(uint32_t w) const { return *(const uint32_t*)(bytes() + P1_W_LINFO * 256 + wv::and_or(w, 0xf00u, lane4())); }   // word index = bits 8..11 of a limit word
	K1_DEV uint8_t& sym_byte(int w0, uint32_t i) const { return bytes()[w0 * 256 + wv::and_or(i, 31u, (uint32_t)lane << 5)]; }   // byte i & 31 of the lane's 32-byte table
};

// Canonical decode out of registers. With v = the next 15 stream bits MSB-first and, for k = 1..15,
//   limit_k = (first_code_k + count_k) << (15 - k)   (the left-aligned upper bound of the codes of length <= k; non-decreasing in k)
// the code length is n + 1 with n = #{k : v >= limit_k}, and the code is the ((v - limit_n) >> (15 - (n + 1)))-th of its length
// (limit_0 = 0). Word W_k (k = 1..15, w[k - 1]) describes what follows when v >= limit_k is the last true comparison:
//   bits 17..31 limit_k | 12..15 code length n + 1 | 8..11 (L) index of the LINFO word | 6..10 (D) index of the length's first symbol | 0..4 32 - length
// The comparison is a plain 32-bit one of vx = v << 17 | 0x1ffff against W_k; t = (vx - W_n) >> (W_n & 31).
// limit_k = 0x8000 (every code is at most k long) does not fit 15 bits: it is stored as 0x7fff, which is wrong for v = 0x7fff only - the
// all-ones code, the last code of the longest length lmax - so the words k >= lmax describe exactly that code (LINFO word 0 / last symbol).
// An incomplete code (zlib accepts a single code of length 1, and an empty distance code): the words k >= lmax decode - as a code of length 1 - to entries 30 / 31
// of the symbol bytes, which the header pass points at invalid entries of the base / extra-bits table.
constexpr uint32_t W0_L = (1u << 12) | (1u << 8) | 31u, W0_D = (1u << 12) | 31u;   // W_0: limit 0, length 1, first symbol 0
struct LimTab
{
	uint32_t w[15];
	K1_DEV void decode(uint32_t w0, uint32_t bits, uint32_t& sel, uint32_t& t) const
	{
		const uint32_t vx = wv::brev(bits) | 0x1ffffu;
		// pivots: W_8; W_4 / W_12; W_2, 6, 10, 14; W_1, 3, .. 15. The candidates of a level are narrowed by the EARLIEST comparison first (its
		// result is known while the later comparisons are still being made), so the select that waits for the level's own predecessor is the last one
		const bool c1 = vx >= w[7];
		const uint32_t p2 = c1 ? w[11] : w[3];
		const uint32_t a3 = c1 ? w[9] : w[1], b3 = c1 ? w[13] : w[5];
		const uint32_t a4 = c1 ? w[8] : w[0], b4 = c1 ? w[10] : w[2], d4 = c1 ? w[12] : w[4], e4 = c1 ? w[14] : w[6];
		const bool c2 = vx >= p2;
		const uint32_t p3 = c2 ? b3 : a3;
		const uint32_t x4 = c2 ? d4 : a4, y4 = c2 ? e4 : b4;
		const bool c3 = vx >= p3;
		const uint32_t p4 = c3 ? y4 : x4;
		const bool c4 = vx >= p4;
		uint32_t s = c1 ? w[7] : w0; s = c2 ? p2 : s; s = c3 ? p3 : s; s = c4 ? p4 : s;
		sel = s; t = (vx - s) >> (s & 31u);
	}
	// Word W_k (k = 1..15) of a code with lmax / total / incomplete as CodeShape found them. code = first code of length k, c = codes of length k,
	// off = index of the first symbol of length k + 1.
	K1_DEV static uint32_t word(int k, uint32_t code, uint32_t c, uint32_t off, uint32_t lmax, uint32_t total, bool incomplete, bool dist)
	{
		uint32_t lim = ((code + c) << (15 - k)) & 0xffffu; if (lim > 0x7fffu) lim = 0x7fffu;
		if ((uint32_t)k < lmax) return (lim << 17) | ((uint32_t)(k + 1) << 12) | (dist ? off << 6 : (uint32_t)(k + 1) << 8) | (uint32_t)(32 - (k + 1));
		if (!incomplete) return (0x7fffu << 17) | (lmax << 12) | (dist ? (total - 1) << 6 : 0u) | (32u - lmax);
		// an incomplete code (one code of length 1, or no code): what lies behind it decodes - as a code of length 1, index 0 or 1 - to entries 30 / 31 of the symbol
		// bytes, which the header pass points at the invalid entries of the base / extra-bits table (the symbol loop has no test of its own for this)
		return (lim << 17) | (1u << 12) | (dist ? 30u << 6 : 0u) | 31u;
	}
};
// what zlib's inftrees.c checks of a set of code lengths: over-subscribed = error; incomplete = error unless it is a single code of length 1 (or no code at all)
struct CodeShape
{
	int left = 1; uint32_t lmax = 0, total = 0; bool over = false;
	K1_DEV void add(uint32_t l, uint32_t c) { left = (left << 1) - (int)c; over = over || left < 0; lmax = c ? l : lmax; total += c; }
	K1_DEV bool incomplete() const { return left > 0; }
	K1_DEV bool ok() const { return !over && (left == 0 || lmax <= 1); }
};

K1_KERNEL_OCC(64, P1_WAVES_PER_SIMD) void huff_tokens_kernel(const uint8_t* __restrict__ comp, const BlockDesc* __restrict__ blocks, int64_t n_blocks,
                                      uint32_t* __restrict__ pool, uint32_t pool_pages, uint32_t* __restrict__ pool_ctr,
                                      uint32_t* __restrict__ tok_first, uint32_t* __restrict__ tok_count,
                                      BlockStatus* __restrict__ status, unsigned long long* __restrict__ work_counter, const uint32_t* __restrict__ order, int park_hi)
{
	K1_SHARED uint32_t lds[P1_LDS_W];
	const int lane = wv::lane();
	P1Lds L{lds, lane};
	wv::set_priority(park_hi >> 8); park_hi &= 255;   // (upper bits: wave priority of the decoder waves, NGSQC_P1_PRIO)
	const wv::u32x4* const comp_q = (const wv::u32x4*)comp;
	uint32_t* const tab = lds + P1_LANE_W * 64;
	uint32_t* const stage = lds + P1_LANE_W * 64 + P1_TAB_W + lane * (4 * P1_LINE_GROUPS);   // lane-major; a lane's four groups rotated by lane / 4, so that sixteen lanes' 16-byte accesses meet sixteen different bank quads
	{
		// RFC 1951 §3.2.5: symbol 256 ends the block, 257..285 are the 29 length symbols, 286/287 (only in the fixed code) are invalid; 30 distance symbols
		const uint32_t i = (uint32_t)lane;
		if (i < 32)
		{
			uint32_t v = TAB_BAD | TAB_EOB;   // (286 / 287 and the entries an incomplete code decodes to: they leave the symbol loop the way the end of a block does)
			if (i == 0) v = TAB_EOB;
			else if (i < 30)
			{
				const uint32_t ls = i - 1;
				const uint32_t eb = ls < 8 ? 0u : (ls == 28 ? 0u : (ls - 4) >> 2);
				const uint32_t bs = ls < 8 ? ls + 3 : (ls == 28 ? 258u : ((4u + ((ls - 4) & 3u)) << eb) + 3u);
				v = bs | (eb << 16);
			}
			tab[i] = v;
		}
		else
		{
			const uint32_t ds = i - 32;
			uint32_t v = TAB_BAD;
			if (ds < 30)
			{
				const uint32_t eb = ds < 4 ? 0u : (ds >> 1) - 1u;
				const uint32_t bs = ds < 4 ? ds + 1 : ((2u + (ds & 1u)) << eb) + 1u;
				v = bs | (eb << 16);
			}
			tab[i] = v;
		}
	}
	wv::barrier();

	// ---- per-lane decoder state (what lives across the symbol loop) ----
	uint32_t state = S_NEXT;
	uint32_t b = 0;                                         // the lane's member (handed out one at a time from a global counter)
	uint64_t q0 = 0; uint32_t n_q = 0, next_q = 0;         // 16-byte pieces of this member: comp_q[q0 + i], i < n_q
	uint32_t abit = 0, abit_end = 0;                       // bit cursor / end of the payload, both counted from the start of piece 0
	uint32_t wr = 0;                                       // window words staged so far
	uint32_t usize = 0, out_n = 0, err = 0, bfinal = 0, raw_left = 0;
	wv::u32x4 pf = wv::make4(0, 0, 0, 0); bool pf_valid = false;
	uint32_t* tptr = nullptr; uint32_t tleft = 0, ngr = 0; // next group of the member's current page, groups left in front of the page's link group, groups written
	uint32_t ttab = 0, ttab_left = 0;                      // the lane's next free literal table (pool word offset), tables left in its table page
	uint32_t sl[P1_TRIPS];                                // the token words of the trips between two service blocks
	#pragma unroll
	for (int i = 0; i < P1_TRIPS; ++i) sl[i] = K1_TOK_NOOP;
	LimTab limL, limD;
	#pragma unroll
	for (int i = 0; i < 15; ++i) { limL.w[i] = 31u; limD.w[i] = 31u; }

	auto exhausted = [&]() -> bool { return next_q >= n_q && !pf_valid; };   // every piece of the member is in the window (what lies behind it is never consumed by a valid stream)
	auto ready = [&](uint32_t bits) -> bool { return (int)(wr * 32u - abit) >= (int)bits || exhausted(); };
	auto window = [&](uint32_t ab) -> uint32_t { const uint32_t* p = L.ring(ab); return wv::alignbit(p[64], p[0], ab); };   // the 32 stream bits at bit position ab
	auto seek = [&](uint32_t target) {   // synchronous restart of the reader at bit position `target`
		next_q = target >> 7; wr = next_q * 4; pf_valid = false;
		wv::u32x4 c0 = next_q < n_q ? comp_q[q0 + next_q] : wv::make4(0, 0, 0, 0); ++next_q;
		wv::u32x4 c1 = next_q < n_q ? comp_q[q0 + next_q] : wv::make4(0, 0, 0, 0); ++next_q;
		L.stage(wr, c0.x, c0.y, c0.z, c0.w); L.stage(wr + 4, c1.x, c1.y, c1.z, c1.w); wr += 8;
		abit = target;
	};
	// one page of the pool (a lane's atomic); false: the pool is used up
	auto new_page = [&](uint32_t& page) -> bool {
		page = wv::atomic_add_u32(pool_ctr, 1u);
		if (page < pool_pages) return true;
		err = K1_ERR_TOKEN_OVERFLOW; state = S_FINISH; return false;
	};
	// A group first goes to the lane's stage; the four groups of a 64-byte line leave for the pool as four stores to consecutive addresses (gi: the group's index in its page)
	constexpr uint32_t LG = (uint32_t)P1_LINE_GROUPS;
	auto staged = [&](uint32_t slot) -> wv::u32x4* { return (wv::u32x4*)(stage + (((slot + ((uint32_t)lane >> 2)) & (LG - 1u)) << 2)); };
	auto flush_line = [&](uint32_t* line, uint32_t n) {
		#pragma unroll
		for (uint32_t k = 0; k < LG; ++k) if (k < n) *(wv::u32x4*)(line + 4 * k) = *staged(k);
	};
	auto put_group = [&](uint32_t gi, uint32_t a, uint32_t bb, uint32_t c, uint32_t d) {
		*staged(gi & (LG - 1u)) = wv::make4(a, bb, c, d);
		if ((gi & (LG - 1u)) == LG - 1u) flush_line(tptr - 4 * (LG - 1u), LG);
		tptr += 4;
	};
	auto emit = [&](uint32_t a, uint32_t bb, uint32_t c, uint32_t d) {
		if (tleft == 0)
		{
			uint32_t page;
			if (!new_page(page)) return;
			if (ngr) put_group(K1_PAGE_GROUPS - 1, K1_TOK_LINK, page, 0u, 0u); else tok_first[b] = page;   // (the link is the last group of its line)
			tptr = pool + (uint64_t)page * K1_PAGE_WORDS; tleft = K1_PAGE_GROUPS - 1;
		}
		put_group(K1_PAGE_GROUPS - 1 - tleft, a, bb, c, d); --tleft; ++ngr;
	};
	// A piece is always in flight; it enters the window as soon as four slots in front of the cursor's word are free. The service block first
	// commits (that is the only place that waits for memory: for what the PREVIOUS service block issued), then stores and requests.
	auto input_commit = [&]() {
		if (pf_valid && (int)(wr - (abit >> 5)) <= P1_RING_SLOTS - 4) { L.stage(wr, pf.x, pf.y, pf.z, pf.w); wr += 4; pf_valid = false; }
	};
	auto input_request = [&]() {
		if (!pf_valid && next_q < n_q && state != S_DONE && state != S_NEXT) { pf = comp_q[q0 + next_q]; ++next_q; pf_valid = true; }
	};
	auto input_service = [&]() { input_commit(); input_request(); };

	// One trip of the symbol loop: up to two literal/length symbols and one distance. At most 15 + (15 + 5) + (15 + 13) = 63 bits.
	// alim: the lane decodes while its cursor is in front of this bit (set by the service block: the window words in front of it are staged, or the member has no
	// more input; 0: the lane does not decode). Round 6, an instruction diet (the decoder's VALU count is what the pipelined K1 pays for): no test for invalid codes
	// (an incomplete code decodes to table entries that end the loop / fail the distance check), LINFO fields and symbol bytes on byte boundaries.
	auto trip = [&](uint32_t& slot, uint32_t& alim) {
		const bool act = abit < alim;
		const uint32_t* p = L.ring(abit);
		const uint32_t p0 = p[0], p1 = p[64], p2 = p[128];
		const uint32_t w0 = wv::alignbit(p1, p0, abit), w1 = wv::alignbit(p2, p1, abit);   // 64 stream bits from the cursor on
		uint32_t sa, ta, sb, tb;
		limL.decode(W0_L, w0, sa, ta);
		const uint32_t len1 = wv::bfe(sa, 12, 4);
		const uint32_t li1 = L.linfo_of(sa);
		limL.decode(W0_L, wv::alignbit(w1, w0, len1), sb, tb);   // the second symbol as if the first were a literal
		const uint32_t len2 = wv::bfe(sb, 12, 4);
		const uint32_t li2 = L.linfo_of(sb);
		const bool lit1 = ta < (li1 & 0xffffu), lit2 = tb < (li2 & 0xffffu);
		const uint32_t tok1 = ta + ((li1 >> 16) & 255u), tok2 = tb + ((li2 >> 16) & 255u);   // literal: index in the block's literal table
		// the symbol >= 256 of the trip (if any): the first symbol, else the second
		const uint32_t e = lit1 ? tb + (li2 >> 24) : ta + (li1 >> 24);   // index in code order (mod 32: sym_byte masks it)
		const uint32_t ub = lit1 ? len1 + len2 : len1;                   // bits up to and including its code (<= 30); both literals: the bits of the trip
		const bool nonlit = !(lit1 && lit2);
		const uint32_t lt = *(const uint32_t*)((const uint8_t*)tab + L.sym_byte(P1_W_NONLIT, e));
		const uint32_t eb = wv::bfe(lt, 16, 3);
		const uint32_t mlen = (lt & 0x1ffu) + wv::bfe(wv::alignbit(w1, w0, ub), 0, eb);
		const uint32_t u2 = ub + eb;                                     // <= 35
		const uint32_t wd = (uint32_t)((((uint64_t)w1 << 32) | w0) >> u2);   // >= 29 valid bits: a distance takes at most 28
		uint32_t sd, td;
		limD.decode(W0_D, wd, sd, td);
		const uint32_t dl = wv::bfe(sd, 12, 4);
		const uint32_t dt = *(const uint32_t*)((const uint8_t*)tab + 128 + L.sym_byte(P1_W_DSYM, td + wv::bfe(sd, 6, 5)));
		const uint32_t deb = wv::bfe(dt, 16, 4);
		const uint32_t mdist = (dt & (TAB_BAD | 0x7fffu)) + wv::bfe(wd, dl, deb);   // (an invalid distance symbol: farther back than any output)
		const bool stop = nonlit && (int32_t)lt < 0, match = nonlit && (int32_t)lt >= 0;   // stop: the end of the block, or an invalid length symbol
		const uint32_t used = match ? u2 + dl + deb : ub;
		const uint32_t tokm = (mlen << 15) + mdist - ((3u << 15) + 1u);                 // = (mlen - 3) << 15 | (mdist - 1)
		// the output position behind the trip's literals, then behind its match (a lane that does not decode adds nothing). A trip that turns out bad
		// still writes its word and moves the cursors: the member ends with an error and phase 2 never looks at its tokens
		const uint32_t o1 = out_n + ((act && lit1) ? 1u : 0u) + ((act && lit1 && lit2) ? 1u : 0u);
		const uint32_t o2 = o1 + ((act && match) ? mlen : 0u);
		const bool bad = (match && mdist > o1) || o2 > usize;
		if ((alim != 0) & !act) K1_STAT(3);
		// the trip's word: two literals | a literal, then a match | a literal (in front of the end of the block) | a match | nothing
		const uint32_t w_lit = lit2 ? (K1_TOK_LIT2 | tok1 | (tok2 << 8)) : (match ? (K1_TOK_LITMATCH | (tok1 << 23) | tokm) : tok1);
		const uint32_t w_non = match ? (K1_TOK_MATCH | tokm) : K1_TOK_NOOP;
		slot = act ? (lit1 ? w_lit : w_non) : K1_TOK_NOOP;
		if (act) K1_STAT(1);
		abit += act ? used : 0u; out_n = o2;
		if (act && (bad || stop))
		{
			alim = 0;
			if (bad) { err = !match ? 3u : (dt & TAB_BAD) ? 12u : mdist > o1 ? 13u : 3u; state = S_FINISH; }
			else if (lt & TAB_BAD) { err = 10u; state = S_FINISH; }   // 286 / 287, or a code that does not exist in an incomplete set
			else state = bfinal ? (uint32_t)S_FINISH : (uint32_t)S_HDR;
		}
	};

	// Header parsing, member changes, stored blocks: everything that is not the symbol loop. Runs until no lane is left in a slow state;
	// the lanes that decode symbols wait (the wave enters only when park_hi lanes need it, or no lane decodes).
	auto slow_section = [&]() {
		// state of a lane's header parse (dead outside this section: a lane never leaves it in the middle of a header)
		uint64_t ccl_lo = 0, ccl_hi = 0; uint32_t limC[7];       // code-length alphabet: sorted symbols (19 x 5 bit), limit/delta words per length 1..7
		#pragma unroll
		for (int i = 0; i < 7; ++i) limC[i] = 0;
		uint32_t h_i = 0, h_n = 0, h_nlit = 0, h_prev = 0, hdr_abit = 0, cur_tab = 0; bool fixed = false, has_eob = false;
		for (;;)
		{
			if (wv::ballot(state - 1u < 6u) == 0) break;
			if (lane == 0) K1_STAT(2);
			K1_WSTAT(1);
			input_service();
			if (state == S_FINISH)
			{
				if (!err && out_n != usize) err = 14;
				if (!err && abit > abit_end) err = 15;   // consumed bits behind the payload: a truncated stream
				{
					// the groups of the member's last, unfinished line
					const uint32_t n_st = (K1_PAGE_GROUPS - 1 - tleft) & (LG - 1u);
					if (ngr && n_st) flush_line(tptr - 4 * n_st, n_st);
				}
				tok_count[b] = ngr; status[b].produced = out_n; status[b].error = err;
				state = S_NEXT;
			}
			else if (state == S_NEXT)
			{
				// members leave the queue in the caller's order (largest compressed size first): the 64 lanes of a wave decode members of
				// nearly equal size and finish together, and the launch ends with its smallest members
				const int64_t nb = (int64_t)wv::atomic_inc(work_counter);
				if (nb >= n_blocks) state = S_DONE;
				else
				{
					b = order ? order[nb] : (uint32_t)nb;
					const BlockDesc bd = blocks[b];
					q0 = bd.cpos >> 4; const uint32_t mis16 = (uint32_t)(bd.cpos & 15); usize = bd.usize;
					n_q = (mis16 + bd.clen + 15) / 16 + 1;
					abit_end = (mis16 + bd.clen) * 8;
					tleft = 0; ngr = 0; out_n = 0; err = 0; bfinal = 0;
					seek(mis16 * 8);
					state = S_HDR;
				}
			}
			else if (state == S_HDR)
			{
				if (ready(80))   // enough input staged for the fixed part of the header (<= 74 bits) or a stored-block header
				{
					if (abit > abit_end) { err = 15; state = S_FINISH; }   // the stream ran past its payload
					else
					{
						uint32_t win = window(abit);
						bfinal = win & 1u; const uint32_t btype = (win >> 1) & 3u; abit += 3;
						#pragma nounroll
						for (uint32_t l = 0; l < 16; ++l) L.linfo(l) = 0;
						h_i = 0; h_prev = 0; has_eob = false; fixed = false;
						if (btype == 0)
						{
							abit = (abit + 7u) & ~7u;   // piece 0 starts on a byte boundary of the stream, so this is the stream's byte alignment
							win = window(abit); abit += 32;
							const uint32_t lo = win & 0xffffu, hi = win >> 16;
							if ((lo ^ hi) != 0xffffu) { err = 2; state = S_FINISH; }
							else { raw_left = lo; state = lo ? (uint32_t)S_RAW : (bfinal ? (uint32_t)S_FINISH : (uint32_t)S_HDR); }
						}
						else if (btype == 1)
						{
							// fixed code: lengths 8 (0..143), 9 (144..255), 7 (256..279), 8 (280..287); 32 distance codes of length 5 - run through the
							// two header passes like a dynamic block whose code lengths are known (no bits consumed)
							fixed = true; h_nlit = 288; h_n = 320; hdr_abit = abit; state = S_P1;
						}
						else if (btype == 2)
						{
							h_nlit = ((win >> 3) & 31u) + 257; const uint32_t ndist = ((win >> 8) & 31u) + 1, ncl = ((win >> 13) & 15u) + 4; abit += 14;
							h_n = h_nlit + ndist;
							if (h_nlit > 286 || ndist > 30) { err = 5; state = S_FINISH; }
							else
							{
								// 19 code-length code lengths (3 bits each, permuted order)
								const uint64_t ORD_LO = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
								const uint64_t ORD_HI = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
								uint64_t cll = 0;   // 3 bits per symbol index
								for (uint32_t i = 0; i < ncl; ++i)
								{
									const uint32_t v = window(abit) & 7u; abit += 3;
									const uint32_t s = (uint32_t)((i < 12 ? ORD_LO >> (5 * i) : ORD_HI >> (5 * (i - 12))) & 31u);
									cll |= (uint64_t)v << (3 * s);
								}
								// counting sort of the 19 symbols by code length (counts / cursors: 5 bits per length in one 64-bit word)
								uint64_t cc = 0;
								#pragma nounroll
								for (uint32_t sy = 0; sy < 19; ++sy) cc += 1ull << (5u * ((uint32_t)(cll >> (3 * sy)) & 7u));
								uint64_t cur = 0; uint32_t k = 0, ccode = 0; int cleft = 1; uint32_t cmax = 0;
								#pragma unroll
								for (int l = 1; l <= 7; ++l)
								{
									const uint32_t cnt = (uint32_t)(cc >> (5 * l)) & 31u;
									cur |= (uint64_t)k << (5 * l);
									uint32_t lim = (ccode + cnt) << (7 - l); if (lim > 0x80u) lim = 0x80u;
									limC[l - 1] = lim | (((k - ccode) & 0xffffu) << 16);
									ccode = (ccode + cnt) << 1; k += cnt;
									cleft = (cleft << 1) - (int)cnt; if (cnt) cmax = (uint32_t)l;
								}
								ccl_lo = 0; ccl_hi = 0;
								#pragma nounroll
								for (uint32_t sy = 0; sy < 19; ++sy)
								{
									const uint32_t l = (uint32_t)(cll >> (3 * sy)) & 7u;
									if (l)
									{
										const uint32_t o = (uint32_t)(cur >> (5 * l)) & 31u; cur += 1ull << (5 * l);
										if (o < 12) ccl_lo |= (uint64_t)sy << (5 * o); else ccl_hi |= (uint64_t)sy << (5 * (o - 12));
									}
								}
								// zlib (inftrees.c): an over-subscribed set is an error, an incomplete one only passes as a single code of length 1
								if (cleft < 0 || (cleft > 0 && cmax != 1)) { err = 6; state = S_FINISH; }
								else { hdr_abit = abit; state = S_P1; }
							}
						}
						else { err = 4; state = S_FINISH; }
					}
				}
			}
			else if (state == S_P1 || state == S_P2)
			{
				// one code-length-alphabet symbol per trip (RFC 1951 §3.2.7); pass 1 counts, pass 2 places the symbols in code order
				if (fixed || ready(14))
				{
					const uint32_t win = fixed ? 0u : window(abit);
					int sym = -1; uint32_t len = 0;
					if (fixed) sym = h_i < 144 ? 8 : h_i < 256 ? 9 : h_i < 280 ? 7 : h_i < 288 ? 8 : 5;
					else
					{
						// canonical decode of the 19-symbol alphabet (lengths 1..7) from registers
						const uint32_t v = wv::brev(win) >> 25;   // next 7 bits, MSB-first
						uint32_t n = 0, sel = 0;
						#pragma unroll
						for (int l = 6; l >= 0; --l) { const bool ge = v >= (limC[l] & 0xffffu); n += ge ? 1u : 0u; sel = ge ? sel : limC[l]; }
						if (n < 7)
						{
							const uint32_t k = (v >> (6 - n)) + (uint32_t)(int)(int16_t)(sel >> 16);
							if (k < 19) { sym = (int)(k < 12 ? (ccl_lo >> (5 * k)) & 31u : (ccl_hi >> (5 * (k - 12))) & 31u); len = n + 1; }
						}
					}
					if (sym < 0) { err = 6; state = S_FINISH; }
					else
					{
						const uint32_t x = win >> len;   // the repeat count's extra bits (len <= 7, at most 7 more)
						uint32_t rep = 1, val = (uint32_t)sym, used = len;
						if (sym == 16) { if (h_i == 0) { err = 7; state = S_FINISH; } rep = 3 + (x & 3u); used += 2; val = h_prev; }
						else if (sym == 17) { rep = 3 + (x & 7u); used += 3; val = 0; }
						else if (sym == 18) { rep = 11 + (x & 127u); used += 7; val = 0; }
						abit += used;
						if (h_i + rep > h_n) { err = 8; state = S_FINISH; }
						else if (err == 0)
						{
							if (val != 0)
							{
								if (state == S_P1)
								{
									// symbols [h_i, h_i + rep) of length val: literals (< 256), the symbols 256.., distance symbols
									const uint32_t e_i = h_i + rep;
									const uint32_t na = h_i >= 256u ? 0u : (e_i < 256u ? e_i : 256u) - h_i;
									const uint32_t lo_b = h_i > 256u ? h_i : 256u, hi_b = e_i < h_nlit ? e_i : h_nlit;
									const uint32_t nbb = hi_b > lo_b ? hi_b - lo_b : 0u;
									L.linfo(val) += na | (nbb << 9) | ((rep - na - nbb) << 18);
									has_eob = has_eob || (h_i <= 256u && e_i > 256u);
								}
								else for (uint32_t k = 0; k < rep; ++k)
								{
									const uint32_t i = h_i + k; uint32_t li = L.linfo(val);
									if (i < 256u) { ((uint8_t*)pool)[(uint64_t)cur_tab * 4u + (li & 511u)] = (uint8_t)i; li += 1u; }
									else if (i < h_nlit) { L.sym_byte(P1_W_NONLIT, (li >> 9) & 511u) = (uint8_t)((i - 256u) * 4u); li += 1u << 9; }
									else { L.sym_byte(P1_W_DSYM, li >> 18) = (uint8_t)((i - h_nlit) * 4u); li += 1u << 18; }
									L.linfo(val) = li;
								}
							}
							h_i += rep; if (sym < 16) h_prev = (uint32_t)sym; else if (sym != 16) h_prev = 0;
							if (h_i == h_n)
							{
								if (state == S_P1)
								{
									// zlib's checks of the two sets (inftrees.c; inflate.c: "missing end-of-block")
									CodeShape shL, shD;
									#pragma nounroll
									for (uint32_t l = 1; l <= 15; ++l) { const uint32_t v = L.linfo(l); shL.add(l, (v & 511u) + ((v >> 9) & 511u)); shD.add(l, v >> 18); }
									if (!shL.ok() || !shD.ok() || !has_eob) { err = 5; state = S_FINISH; }
									else
									{
										// the decode words of both codes from the counts; counts -> placement cursors (first index of every length in the three symbol orders)
										uint32_t oa = 0, ob = 0, oc = 0, codeL = 0, codeD = 0;
										#pragma unroll
										for (int l = 1; l <= 15; ++l)
										{
											const uint32_t v = L.linfo((uint32_t)l), ca = v & 511u, cb = (v >> 9) & 511u, cd = v >> 18;
											L.linfo((uint32_t)l) = oa | (ob << 9) | (oc << 18);
											oa += ca; ob += cb; oc += cd;
											limL.w[l - 1] = LimTab::word(l, codeL, ca + cb, 0u, shL.lmax, shL.total, shL.incomplete(), false);
											limD.w[l - 1] = LimTab::word(l, codeD, cd, oc, shD.lmax, shD.total, shD.incomplete(), true);
											codeL = (codeL + ca + cb) << 1; codeD = (codeD + cd) << 1;
										}
										// the block's literal table: 256 bytes of the pool, announced to phase 2 in the token stream
										if (ttab_left == 0) { uint32_t page; if (new_page(page)) { ttab = page * K1_PAGE_WORDS; ttab_left = K1_TABLES_PER_PAGE; } }
										if (err == 0)
										{
											cur_tab = ttab; ttab += K1_TABLE_WORDS; --ttab_left;
											emit(K1_TOK_TABLE, cur_tab, 0u, 0u);
										}
										if (err == 0) { seek(hdr_abit); h_i = 0; h_prev = 0; state = S_P2; }
									}
								}
								else
								{
									// cursors (now the END of every length) -> what the symbol loop reads: literals of the length, first literal index, first index of the symbols 256..
									uint32_t pa = 0, pb = 0, pc = 0; bool last_non = false; int leftL = 1, leftD = 1;
									#pragma nounroll
									for (uint32_t l = 1; l <= 15; ++l)
									{
										const uint32_t v = L.linfo(l), ea = v & 511u, eb2 = (v >> 9) & 511u, ec = v >> 18, nl = ea - pa, nn = eb2 - pb;
										L.linfo(l) = nl | ((pa & 255u) << 16) | (((pb - nl) & 255u) << 24);
										if (nl + nn) last_non = nn != 0;
										leftL = (leftL << 1) - (int)(nl + nn); leftD = (leftD << 1) - (int)(ec - pc);
										pa = ea; pb = eb2; pc = ec;
									}
									L.linfo(0) = last_non ? (((pb - 1u) & 255u) << 24) : (1u | (((pa - 1u) & 255u) << 16));   // the all-ones code: the last symbol of the longest length
									// an incomplete set (the check above let it pass: a single code of length 1 / no distance code): the decode words of the bits behind its codes
									// lead to entries 30 and 31 of the symbol bytes (free: at most one symbol is in use) - the invalid entries of the base / extra-bits tables
									if (leftL > 0) { L.linfo(0) = 30u << 24; L.sym_byte(P1_W_NONLIT, 30u) = 30u * 4u; L.sym_byte(P1_W_NONLIT, 31u) = 31u * 4u; }
									if (leftD > 0) { L.sym_byte(P1_W_DSYM, 30u) = 30u * 4u; L.sym_byte(P1_W_DSYM, 31u) = 31u * 4u; }
									state = S_SYM;
								}
							}
						}
					}
				}
			}
			else if (state == S_RAW)
			{
				// a stored block leaves as raw runs: phase 2 copies the bytes from the compressed input
				const uint32_t n = raw_left < 256u ? raw_left : 256u, off8 = abit >> 3, mis16 = (uint32_t)(blocks[b].cpos & 15);
				if (out_n + n > usize) { err = 3; state = S_FINISH; }
				else if ((off8 + n) * 8u > abit_end) { err = 15; state = S_FINISH; }
				else
				{
					emit(K1_TOK_RAW | ((n - 1u) << 16) | (off8 - mis16), K1_TOK_NOOP, K1_TOK_NOOP, K1_TOK_NOOP);
					if (err == 0)
					{
						abit += 8u * n; out_n += n; raw_left -= n;
						if (raw_left == 0) { seek(abit); state = bfinal ? (uint32_t)S_FINISH : (uint32_t)S_HDR; }
					}
				}
			}
		}
	};

	for (;;)
	{
		// ---- service: commit prefetched input, store the token groups of the last trips, issue the next prefetch ----
		input_commit();
		if (((sl[0] ^ K1_TOK_NOOP) | (sl[1] ^ K1_TOK_NOOP) | (sl[2] ^ K1_TOK_NOOP) | (sl[3] ^ K1_TOK_NOOP)) != 0) emit(sl[0], sl[1], sl[2], sl[3]);
		input_request();
		// A lane that reaches a header (or the end of its member) parks until park_hi lanes are parked or no lane decodes symbols; then the
		// wave runs ONLY the slow states until every parked lane is back in the symbol loop.
		{
			const uint64_t slow_m = wv::ballot(state - 1u < 6u);
			if (slow_m != 0)
			{
				if ((int)wv::popc64(slow_m) >= park_hi || wv::ballot(state == S_SYM) == 0) slow_section();
			}
			else if (wv::ballot(state != S_DONE) == 0) break;
		}
		if (lane == 0) K1_STAT(0);
		K1_WSTAT(0);
		uint32_t alim = state == S_SYM ? (exhausted() ? 0xffffffffu : (wr - 2u) * 32u) : 0u;   // (wr >= 8 once a member is open)
		#pragma unroll
		for (int i = 0; i < P1_TRIPS; ++i) trip(sl[i], alim);
	}
}

// ---------------------------------------------------------------------------------------------------------------- phase 2
// ITEM centred resolve (round 4). A batch is up to 64 token groups (one per lane, <= 256 words) cut to <= P2_BMAX output bytes; a wave prefix sum
// places every word. Literal bytes go straight to the LDS staging bytes (translated through the block's literal table). Every match (and raw run)
// is cut into ITEMS of at most 16 bytes - a match of up to 16 bytes is one item, a longer one is covered by 16-byte items whose last one overlaps
// its predecessor - and the items of the batch are listed in LDS in output order. Then ONE LANE PER ITEM, 64 items per step:
//   far item   (every source byte lies in front of the batch) four dword loads from the member's output in HBM at the OVERLAPPING offsets
//              0, min(4, len - 4), min(8, len - 4), len - 4, which cover any length 4..16 exactly; they are issued one step ahead
//   near item  the source reaches into the batch (or the 16 bytes in front of it, which stay in LDS): the same two pieces from LDS, once the
//              items that write its source are done - a 64-bit lane mask per item (from a bit mask of item starts), tested against the
//              ballot of unfinished lanes; the lowest unfinished lane never waits, so the rounds terminate
//   distance 1 the byte in front of the item, repeated; an item that overlaps its own source otherwise (rare) is copied byte by byte
// and the item's bytes are written to LDS with two stores at the same offsets (lengths 1..3: a 16-bit and an 8-bit store). When all steps
// are done the batch leaves LDS for HBM as whole dwords (256 consecutive bytes per store instruction).
constexpr int P2_BMAX = 2032;                  // output bytes per batch (an item's place in the batch has 11 bits); at least one whole group (4 x 259 bytes) always fits
constexpr int P2_GROUPS = 64;                  // groups per batch: one per lane
constexpr int P2_HIST = 16;                    // output bytes in front of the batch that stay in LDS (a near item's source starts at most 15 bytes in front)
constexpr int P2_IMAX = 192;                   // items per batch: a batch is cut so that its items fill whole steps of 64 (round 6: 129 -> 109 steps per member of the bench data)
constexpr int P2_NW = P2_BMAX / 32 + 1;        // words of the item-start bit mask (one lane each in the count scan)
constexpr int P2_LONG = P2_IMAX / 2;           // matches of more than 16 bytes in a batch (each has at least two items)
constexpr int P2_TRASH = P2_BMAX + 15;         // a staging byte nobody reads: where the stores of lanes without a literal go (no branch around a store)
static_assert(P2_BMAX >= 4 * 259 && P2_BMAX < 2048 && P2_BMAX % 8 == 0 && P2_NW <= 64 && P2_HIST % 4 == 0 && P2_IMAX >= 4 * 17 && P2_IMAX % 64 == 0, "phase-2 batch geometry");
// item word: first byte (batch-relative) | (length - 1) << 11 | (distance - 1) << 15; a raw run: bit 31 | payload offset << 15. An item of a distance-1 match
// (a run of one byte: the usual self-overlapping match of BAM data) carries bit 30 and the distance to the byte IN FRONT OF THE MATCH instead: every item of
// the run repeats that byte, none of them waits for its predecessor
struct P2Lds
{
	uint32_t it[P2_IMAX + 8];                              // (the last word: where the item words of lanes without an item go)
	uint32_t ib[2 * 64];                                   // per 32 output bytes: {bit mask of the first bytes of NEAR items, near items that start in front of them}
	uint32_t lg[2 * P2_LONG + 8];                            // the batch's matches of more than 16 bytes: {first byte | length << 11 | index of the first item << 20, key}; then (pass A / B) the near items
	alignas(16) uint8_t val[P2_HIST + P2_BMAX + 16];       // [history | the batch's bytes | slack]
	alignas(4) uint8_t lit[256];                           // the literal table of the current DEFLATE block
};

// Round 6: the kernel's instruction count, not its memory traffic, is what the pipelined K1 pays for (the chip's instruction issue is what phase 1, phase 2 and the CRC
// share), and two thirds of the instructions were scalar: exec-mask bookkeeping of nested branches. So the control flow is FLAT - a store that a lane does not take
// goes to a trash byte instead of behind a branch, the length classes of an item are predicates side by side, rare kinds (raw runs, runs of a byte, an item that
// overlaps its source) leave through one ballot-guarded block - the items of a long match come from a compacted list (a lane per long match) instead of four
// divergent loops, a batch is cut to whole steps of 64 items, and the batch leaves LDS eight bytes per lane.
K1_KERNEL_OCC(64, 8) void lz77_groups_kernel(const uint32_t* __restrict__ pool, const uint32_t* __restrict__ tok_first, const uint32_t* __restrict__ tok_count,
                                      const BlockDesc* __restrict__ blocks, int64_t n_blocks, uint8_t* __restrict__ out_base, BlockStatus* __restrict__ status, const uint8_t* __restrict__ comp)
{
	K1_SHARED P2Lds S;
	const int lane = wv::lane();
	const wv::u32x4 noop4 = wv::make4(K1_TOK_NOOP, K1_TOK_NOOP, K1_TOK_NOOP, K1_TOK_NOOP);
	uint8_t* const vb = S.val + P2_HIST;   // byte j of the batch
	for (int64_t b = wv::block_id(); b < n_blocks; b += wv::grid_size())
	{
		if (status[b].error) continue;
		const uint32_t ngroups = tok_count[b];
		const uint32_t usize = blocks[b].usize;
		const wv::ByteBuf out = wv::ByteBuf::make(out_base + blocks[b].upos, usize);            // stores: exact bounds
		const wv::ByteBuf outld = wv::ByteBuf::make(out_base + blocks[b].upos, usize + 3u);     // a source dword may end up to three bytes behind the item's source (mapped: the next member or the buffer's slack)
		const wv::ByteBuf cin = wv::ByteBuf::make(const_cast<uint8_t*>(comp) + blocks[b].cpos, blocks[b].clen + 3u);   // raw runs (stored blocks) are copied from here (the gzip trailer follows)
		// the member's pages: logical group g lives in page g / (K1_PAGE_GROUPS - 1); a batch spans at most two pages
		uint32_t pg_cur = tok_first[b], pg_first = 0;
		uint32_t pg_next = ngroups > K1_PAGE_GROUPS - 1 ? pool[(uint64_t)pg_cur * K1_PAGE_WORDS + K1_PAGE_WORDS - 3] : 0u;
		auto group = [&](uint32_t g) -> wv::u32x4 {
			const uint32_t rel = g - pg_first; const bool nx = rel >= K1_PAGE_GROUPS - 1;
			return ((const wv::u32x4*)(pool + (uint64_t)(nx ? pg_next : pg_cur) * K1_PAGE_WORDS))[nx ? rel - (K1_PAGE_GROUPS - 1) : rel];
		};
		// (a page's link word is only meaningful when the member has groups behind that page: a member that ends with its page was never linked further)
		auto next_page = [&]() {
			pg_cur = pg_next; pg_first += K1_PAGE_GROUPS - 1;
			pg_next = ngroups - pg_first > K1_PAGE_GROUPS - 1 && ngroups > pg_first ? pool[(uint64_t)pg_cur * K1_PAGE_WORDS + K1_PAGE_WORDS - 3] : 0u;
		};
		uint32_t P = 0, fail = 0;   // bytes written so far
		wv::u32x4 nxt = (uint32_t)lane < ngroups ? group((uint32_t)lane) : noop4;
		for (uint32_t g0 = 0; g0 < ngroups;)
		{
			// ---- place the batch: one group per lane, a prefix sum over (bytes | items << 17) ----
			bool valid = g0 + (uint32_t)lane < ngroups;
			const uint32_t t0 = nxt.x, t1 = nxt.y, t2 = nxt.z, t3 = nxt.w;
			// a table group switches the literal table for the words behind it: it ends the batch in front of it, or (first group) is consumed here
			const uint64_t tabm = wv::ballot(valid && t0 == K1_TOK_TABLE);
			if (tabm & 1ull)
			{
				K1_WSTAT(18);
				wv::barrier();   // (the previous batch's readers of S.lit are done)
				((uint32_t*)S.lit)[lane] = pool[(uint64_t)wv::readlane(t1, 0) + (uint32_t)lane];
				wv::barrier();
				g0 += 1;
				if (g0 - pg_first >= K1_PAGE_GROUPS - 1) next_page();
				nxt = g0 + (uint32_t)lane < ngroups ? group(g0 + (uint32_t)lane) : noop4;
				continue;
			}
			if (tabm) valid = valid && ((1ull << lane) & (tabm - 1ull) & ~tabm) != 0;   // lanes in front of the first table group
			const uint32_t tt[4] = {t0, t1, t2, t3};
			uint32_t ll[4], cl[4], s = 0, c = 0;   // per word: output bytes, bytes that are copied (match / raw run)
			#pragma unroll
			for (int k = 0; k < 4; ++k)
			{
				// (selects, not branches: a match | one or two literals | a raw run | nothing)
				const uint32_t t = tt[k];
				const uint32_t lm = ((t >> 15) & 255u) + 3u, lr = ((t >> 16) & 255u) + 1u;
				const bool ism = t >= K1_TOK_MATCH, isl = t < K1_TOK_RAW, isr = !isl && t < 2u * K1_TOK_RAW;
				const uint32_t cp = ism ? lm : (isr ? lr : 0u);
				const uint32_t lt = isl ? 1u + ((t >> 16) & 1u) : t >> 31;   // literal bytes of the word
				cl[k] = valid ? cp : 0u;
				ll[k] = valid ? cp + lt : 0u;
				s += ll[k]; c += (cl[k] + 15u) >> 4;
			}
			const uint32_t E = wv::scan_incl(s | (c << 17));   // (64 groups: < 2^17 bytes, < 2^15 items)
			const uint32_t Eb = E & 0x1ffffu, Ei = E >> 17;
			// the longest prefix of groups whose output fits the staging buffer and whose items fill at most three steps (the sums are non-decreasing: the ballot is a prefix mask)
			const uint32_t ng = wv::popc64(wv::ballot(valid && Eb <= (uint32_t)P2_BMAX && Ei <= (uint32_t)P2_IMAX));
			if (ng == 0) { fail = 18; break; }   // a group longer than 4 x 259 bytes: not a token stream of phase 1
			const uint32_t B = wv::readlane(Eb, (int)ng - 1), NI = wv::readlane(Ei, (int)ng - 1);
			if (P + B > usize) { fail = 16; break; }
			S.ib[2 * lane] = 0u;
			wv::barrier();
			// per word: literal bytes to their place, the FIRST item of the copied part; a match of more than 16 bytes (one in four) is listed for the second pass
			auto item_key = [&](uint32_t t) -> uint32_t {   // the item word without its position and length
				return t >= K1_TOK_MATCH ? ((t & 0x7fffu) == 0u ? 0x40000000u : (t & 0x7fffu) << 15) : 0x80000000u | ((t & 0xffffu) << 15);
			};
			uint32_t nlong = 0;
			{
				const bool inb = (uint32_t)lane < ng;
				uint32_t st = Eb - s, ix = Ei - c;
				#pragma unroll
				for (int k = 0; k < 4; ++k)
				{
					const uint32_t t = tt[k], n = inb ? cl[k] : 0u, l = inb ? ll[k] : 0u;
					K1_WSTAT(11);
					const bool lmt = (t >> 31) != 0u, isl = t < K1_TOK_RAW;
					const bool l1 = l != 0u && (lmt || isl), l2 = isl && l == 2u;
					const uint32_t i1 = lmt ? (t >> 23) & 255u : t & 255u;
					vb[l1 ? st : (uint32_t)P2_TRASH] = S.lit[i1];
					vb[l2 ? st + 1u : (uint32_t)P2_TRASH] = S.lit[(t >> 8) & 255u];
					const bool has = n != 0u;
					const uint32_t d = has ? st + l - n : 0u, key = item_key(t);
					S.it[has ? ix : (uint32_t)(P2_IMAX + 7)] = d | (((n < 16u ? n : 16u) - 1u) << 11) | key;
					const uint64_t lgm = wv::ballot(n > 16u);
					if (lgm != 0ull)
					{
						if (n > 16u)
						{
							const uint32_t j = nlong + wv::mbcnt(lgm);
							S.lg[2 * j] = d | (n << 11) | (ix << 20); S.lg[2 * j + 1] = key;
						}
						nlong += wv::popc64(lgm);
					}
					st += l; ix += (n + 15u) >> 4;
				}
			}
			wv::barrier();
			// the further items of the long matches: a lane per match
			for (uint32_t j0 = 0; j0 < nlong; j0 += 64u)
			{
				const uint32_t j = j0 + (uint32_t)lane;
				if (j < nlong)
				{
					const uint32_t w = S.lg[2 * j], key = S.lg[2 * j + 1], ms = w & 0x7ffu, n = (w >> 11) & 511u, ix = w >> 20;
					for (uint32_t o = 16u; o < n; o += 16u)
					{
						K1_WSTAT(12);
						const uint32_t off = o + 16u <= n ? o : n - 16u, d = ms + off;   // (the last item of a long match overlaps its predecessor)
						S.it[ix + (o >> 4)] = d | (15u << 11) | (key + ((key >> 30) ? off << 15 : 0u));   // (raw run: payload offset of the item; run of a byte: distance to the byte in front of the match)
					}
				}
			}
			wv::barrier();
			// stores of earlier batches must be complete before this batch loads from the window behind P
			wv::wait_vm0();
			// the next batch's groups are requested now; they arrive while this batch is resolved
			{
				const uint32_t g1 = g0 + ng;
				if (g1 - pg_first >= K1_PAGE_GROUPS - 1) next_page();
				nxt = g1 + (uint32_t)lane < ngroups ? group(g1 + (uint32_t)lane) : noop4;
			}

			if (lane == 0) K1_STAT(7);
			K1_WSTAT(10);
			// ---- the items, in two passes (round 6) ----
			// Pass A, 64 items per step: a FAR item (its source lies in front of the batch: seven in ten on BAM data) is two overlapping loads from the member's output in HBM, issued a
			// step ahead, and two stores into the staging bytes - no dependencies, no rounds; a NEAR item is only moved to a compact list (S.lg, in output order) and its first byte
			// marked in the bit mask. Pass B resolves the near items, 64 per step, in dependency rounds: what they read is then in LDS except what other near items of the same step
			// write. (Round 5 resolved every step of 64 mixed items in rounds: 109 steps and 370 rounds per member of the bench data, each round a walk over three length classes.)
			// An item's bytes travel in two overlapping pieces: lengths 9..16 as the two 8-byte words at 0 and len - 8, lengths 4..8 as the two dwords at 0 and
			// len - 4, three bytes as one dword (stored as a 16-bit and an 8-bit piece)
			struct Far { uint32_t w, k, w0, w1; uint64_t lo, hi; };   // k: 0 nothing to store, 1 / 2 / 3 loaded item of 9..16 / 4..8 / 3 bytes, 4 a rare kind (raw run, far item of a run, one or two bytes)
			uint32_t nn = 0;   // near items so far
			auto fetch = [&](uint32_t i0) -> Far {
				Far q; const uint32_t idx = i0 + (uint32_t)lane; K1_WSTAT(13);
				const bool in = idx < NI;
				q.w = in ? S.it[idx] : 0u; q.lo = q.hi = 0; q.w0 = q.w1 = 0;
				const uint32_t d = q.w & 0x7ffu, len = wv::bfe(q.w, 11, 4) + 1u, dist = wv::bfe(q.w, 15, 15) + 1u;
				const int src = (int)d - (int)dist;
				const bool raw = (q.w >> 31) != 0u, run = (q.w >> 30) == 1u;
				const bool far = src + (run ? 1 : (int)len) <= 0;
				const bool near = in && !raw && !far;
				q.k = !in || near ? 0u : (raw || run || len < 3u ? 4u : (len >= 9u ? 1u : (len >= 4u ? 2u : 3u)));
				// near items: to the list
				const uint64_t nm = wv::ballot(near);
				const uint32_t dn = near ? d : 0u;
				S.lg[near ? nn + wv::mbcnt(nm) : (uint32_t)(2 * P2_LONG + 7)] = q.w;
				wv::lds_or32(&S.ib[2 * (dn >> 5)], near ? 1u << (dn & 31u) : 0u);
				nn += wv::popc64(nm);
				// far items: the loads (a run's byte: one dword)
				const uint32_t a = P + (uint32_t)src;
				const bool ldf = in && !near && !raw, wide = len >= 9u && !run;
				if (ldf && wide) { q.lo = outld.load64(a); q.hi = outld.load64(a + len - 8u); }
				if (ldf && !wide) { q.w0 = outld.load32(a); q.w1 = outld.load32(a + (len >= 4u && !run ? len - 4u : 0u)); }
				return q;
			};
			auto store_far = [&](Far& q) {
				K1_WSTAT(14);
				const uint32_t d = q.w & 0x7ffu, len = wv::bfe(q.w, 11, 4) + 1u;
				uint8_t* const pd = vb + d; uint8_t* const pd2 = pd + len - (len >= 9u ? 8u : 4u);
				if (wv::ballot(q.k == 4u) != 0ull)
				{
					// (rare) a stored block's bytes come from the compressed input; a loaded item of a run repeats its byte; a raw item of one or two bytes
					const bool raw = (q.w >> 31) != 0u;
					if (q.k == 4u && raw)
					{
						const uint32_t a = wv::bfe(q.w, 15, 16);
						if (len >= 9u) { q.lo = cin.load64(a); q.hi = cin.load64(a + len - 8u); }
						else { q.w0 = cin.load32(a); q.w1 = cin.load32(a + (len >= 4u ? len - 4u : 0u)); }
					}
					wv::wait_vm0();
					if (q.k == 4u)
					{
						if (!raw) { const uint32_t r4 = (q.w0 & 255u) * 0x01010101u; q.w0 = q.w1 = r4; q.lo = q.hi = ((uint64_t)r4 << 32) | r4; }
						if (len >= 9u) { wv::lds_store64u(pd, q.lo); wv::lds_store64u(pd2, q.hi); }
						else if (len >= 4u) { wv::lds_store32u(pd, q.w0); wv::lds_store32u(pd2, q.w1); }
						else
						{
							if (len >= 2u) wv::lds_store16u(pd, q.w0);
							vb[d + len - 1u] = (uint8_t)(q.w0 >> (8u * (len - 1u)));
						}
					}
				}
				// (predicates of the length, not comparisons of q.k with 1, 2, 3: those become a decision tree of nested branches)
				const bool ld = q.k != 0u && q.k != 4u;
				if (ld && len >= 9u) { wv::lds_store64u(pd, q.lo); wv::lds_store64u(pd2, q.hi); }
				if (ld && len >= 4u && len < 9u) { wv::lds_store32u(pd, q.w0); wv::lds_store32u(pd2, q.w1); }
				if (ld && len == 3u) { wv::lds_store16u(pd, q.w0); pd[2] = (uint8_t)(q.w0 >> 16); }
			};
			wv::barrier();   // (the list of long matches in S.lg has been read)
			{
				// two steps per trip: the loads of one step's far items are in flight while the other step is stored (only loads are in flight: they return in order)
				Far qa = fetch(0u), qb = qa;
				#pragma nounroll
				for (uint32_t i0 = 0; i0 < NI; i0 += 128u)
				{
					const bool hb = i0 + 64u < NI;
					if (hb) { qb = fetch(i0 + 64u); wv::wait_vm4(); } else wv::wait_vm0();
					store_far(qa);
					if (hb)
					{
						if (i0 + 128u < NI) { qa = fetch(i0 + 128u); wv::wait_vm4(); } else wv::wait_vm0();
						store_far(qb);
					}
				}
			}
			wv::barrier();
			{
				// near items that start in front of each 32-byte piece
				const uint32_t cn = lane < P2_NW ? wv::bcnt(S.ib[2 * lane]) : 0u;
				const uint32_t inc = wv::scan_incl(cn);
				if (lane < P2_NW) S.ib[2 * lane + 1] = inc - cn;
			}
			wv::barrier();
			auto starts_before = [&](uint32_t p) -> uint32_t { return S.ib[2 * (p >> 5) + 1] + wv::bcnt(wv::bfe(S.ib[2 * (p >> 5)], 0, p & 31u)); };   // near items that start in front of byte p
			#pragma nounroll
			for (uint32_t i0 = 0; i0 < nn; i0 += 64u)
			{
				if (lane == 0) K1_STAT(4);
				K1_WSTAT(19);
				const uint32_t idx = i0 + (uint32_t)lane;
				const uint32_t w = idx < nn ? S.lg[idx] : 0u;
				const uint32_t d = w & 0x7ffu, len = wv::bfe(w, 11, 4) + 1u, dist = wv::bfe(w, 15, 15) + 1u;
				const bool run = (w >> 30) == 1u;
				const int src = (int)d - (int)dist;
				uint8_t* const pd = vb + d; uint8_t* const pd2 = pd + len - (len >= 9u ? 8u : 4u);
				const uint8_t* const ps = vb + src; const uint8_t* const ps2 = ps + len - (len >= 9u ? 8u : 4u);
				// the lanes of this step that may write into the item's source: the near items that start in (src - 16, src + slen), as far as they lie in front of this
				// item (an exact first lane - from a second bit plane of item ends - saved one round in seventy)
				uint64_t dep = 0;
				{
					const uint32_t slen = run ? 1u : len;   // source bytes
					const int lo = (int)starts_before((uint32_t)(src > 15 ? src - 15 : 0)) - (int)i0;
					int hi = (int)starts_before((uint32_t)(src + (int)slen)) - 1 - (int)i0;
					hi = hi < lane ? hi : lane - 1;
					const int l0 = lo > 0 ? lo : 0;
					if (hi >= l0) dep = ((2ull << (hi - l0)) - 1ull) << l0;
				}
				// (the three length classes as predicates that live across the rounds - scalar masks; compared inside the loop, a kind number becomes a decision tree of nested branches)
				const bool c9 = len >= 9u, c4 = len >= 4u && len < 9u, c3 = len == 3u;
				uint32_t x = idx >= nn ? 0u : (run || dist < len ? 4u : 2u);   // 0 done, 2 near item of two pieces, 4 near item of a rare kind (a run of a byte, an item that overlaps its own source)
				const bool any4 = wv::ballot(x == 4u) != 0ull;                 // (wave-uniform: the rounds of a step without such an item do not look for one)
				for (;;)
				{
					if (lane == 0) K1_STAT(6);
					K1_WSTAT(15);
					const uint64_t open = wv::ballot(x != 0u);
					if (open == 0ull) break;
					const bool free = (open & dep) == 0ull;   // the lane's source is complete
					if (any4)
					{
						if (free && x == 4u)
						{
							if (run)
							{
								const uint32_t r4 = (uint32_t)*ps * 0x01010101u; const uint64_t r8 = ((uint64_t)r4 << 32) | r4;
								if (len >= 9u) { wv::lds_store64u(pd, r8); wv::lds_store64u(pd2, r8); }
								else if (len >= 4u) { wv::lds_store32u(pd, r4); wv::lds_store32u(pd2, r4); }
								else { wv::lds_store16u(pd, r4); pd[2] = (uint8_t)r4; }
							}
							else for (uint32_t k = 0; k < len; ++k) { K1_WSTAT(16); pd[k] = ps[k]; }   // an item that overlaps its own source with a distance of 2..15: byte by byte
						}
					}
					const bool g = free && x == 2u;
					if (g && c9) { const uint64_t u = wv::lds_load64u(ps), v = wv::lds_load64u(ps2); wv::lds_store64u(pd, u); wv::lds_store64u(pd2, v); }
					if (g && c4) { const uint32_t u = wv::lds_load32u(ps), v = wv::lds_load32u(ps2); wv::lds_store32u(pd, u); wv::lds_store32u(pd2, v); }
					if (g && c3) { const uint32_t u = wv::lds_load32u(ps); wv::lds_store16u(pd, u); pd[2] = (uint8_t)(u >> 16); }
					x = free ? 0u : x;
					wv::barrier();
				}
			}
			// ---- the batch leaves for HBM: eight bytes per lane and store, the last 1..7 bytes one by one ----
			wv::barrier();
			{
				const uint32_t B8 = B & ~7u;
				#pragma nounroll
				for (uint32_t j0 = 0; j0 < B8; j0 += 512u)
				{
					const uint32_t j = j0 + 8u * (uint32_t)lane; K1_WSTAT(17);
					if (j < B8) out.store64(P + j, wv::lds_load64(vb + j));
				}
				if ((uint32_t)lane < (B & 7u)) out.store(P + B8 + (uint32_t)lane, vb[B8 + (uint32_t)lane]);
			}
			// the last P2_HIST bytes stay in LDS in front of the next batch
			{
				const uint32_t h = lane < P2_HIST / 4 ? wv::lds_load32u(vb + (int)B - P2_HIST + 4 * lane) : 0u;
				wv::barrier();
				if (lane < P2_HIST / 4) wv::lds_store32(S.val + 4 * lane, h);
				wv::barrier();
			}
			P += B; g0 += ng;
		}
		if (lane == 0)
		{
			if (fail) status[b].error = fail;
			else if (P != usize) status[b].error = 17;
			status[b].produced = P;
		}
	}
}

} } // namespace ngsqc::k1
