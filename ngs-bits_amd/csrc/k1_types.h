// Plain data types of K1 (BGZF inflate): shared by the device code, the host side of the library and the wave-emulation test
// harness (tests/emul), therefore free of HIP includes.
#pragma once
#include <cstdint>

namespace ngsqc {

// One BGZF member as the device sees it (SAM spec §4.1). cpos = byte offset of the raw DEFLATE payload inside the
// compressed image in HBM, clen = payload bytes, upos/usize = where its output goes in the inflated stream.
struct BlockDesc { uint64_t cpos; uint64_t upos; uint32_t clen; uint32_t usize; };

// Per-member result of K1: bytes produced (must equal usize) and an error code (0 = ok).
struct BlockStatus { uint32_t produced; uint32_t error; };

enum { K1_ERR_CRC = 20, K1_ERR_TOKEN_OVERFLOW = 100 };   // BlockStatus.error values the host treats specially

// Token stream between phase 1 and phase 2. It lives in a POOL of pages that the decoder lanes of one launch allocate from with an
// atomic counter (a member takes what its token stream needs; nothing is sized per member in advance). A page is K1_PAGE_WORDS
// 32-bit words = K1_PAGE_GROUPS groups of four words; a group is one 16-byte store of a decoder lane = the four trips between two
// service blocks, ONE WORD PER TRIP (round 4; round 3 wrote two slots per trip, four bytes per literal). The last group of a page is the
// link to the member's next page. Token words (a trip decodes up to two literal/length symbols and one distance):
//   literal + match : bit 31 | literal index << 23 | (length - 3) << 15 | (distance - 1)     a literal, then a match
//   match           : bit 30 | (length - 3) << 15 | (distance - 1)                           (bits 23..29 zero)
//   literals        : 0x000000aa (one) or 0x0001bbaa (two: aa first)                         indices into the literal table of the current
//                     DEFLATE block (the literals sorted by (code length, value)); phase 2 translates them - the decoder lanes keep no symbol
//                     table for literals
//   raw run         : 0x10000000 | (length - 1) << 16 | byte offset in the member's payload  (stored blocks: 1..256 bytes copied from the input)
//   no-op           : 0x3fffffff  (a trip in which the lane produced nothing)
// Groups with a special first word (the other words are not tokens):
//   table     : {K1_TOK_TABLE, pool word offset of a 256-byte literal table, 0, 0}  - the block that starts here uses that table
//   link      : {K1_TOK_LINK, index of the member's next page, 0, 0}               - always group K1_PAGE_GROUPS - 1 of a page
constexpr uint32_t K1_TOK_NOOP = 0x3fffffffu, K1_TOK_TABLE = 0x3ffffffeu, K1_TOK_LINK = 0x3ffffffdu;
constexpr uint32_t K1_TOK_MATCH = 0x40000000u, K1_TOK_LITMATCH = 0x80000000u, K1_TOK_LIT2 = 0x00010000u, K1_TOK_RAW = 0x10000000u;
constexpr uint32_t K1_PAGE_WORDS = 1024, K1_PAGE_GROUPS = K1_PAGE_WORDS / 4, K1_TABLE_WORDS = 64, K1_TABLES_PER_PAGE = K1_PAGE_WORDS / K1_TABLE_WORDS;

// Pages a launch over members with these sizes may need: the expected token volume of BAM data (a word per trip: about 1.8 bytes of
// tokens per compressed byte on the bench data, 2.3 on literal-heavy 40-level qualities) with a margin - 3 bytes per compressed byte -
// bounded by the worst case (a word per output byte and its three no-op neighbours), plus per member one partly used token page and a
// share of a table page. A launch that runs out of pages reports K1_ERR_TOKEN_OVERFLOW for the members it could not finish; the host
// repeats those with a worst-case pool.
inline uint64_t k1_pool_pages(uint64_t sum_clen, uint64_t sum_usize, uint64_t n_members, bool worst_case)
{
	const uint64_t worst = 4 * sum_usize + 64 * n_members;   // token words
	uint64_t words = worst_case ? worst : (3 * sum_clen) / 4;
	if (!worst_case && words > worst) words = worst;
	return words / (K1_PAGE_WORDS - 4) + 2 * n_members + 64;
}

// The bound that holds for EVERY valid member (third chance, a few members at a time): besides the token words (at most four per output byte: a group holds
// at least one real word) every DEFLATE block may cost a table group, a 256-byte literal table and a padded last group = 76 words, and the smallest block (an empty fixed-Huffman one: 10 bits) lets a
// member hold 0.8 blocks per payload byte - zlib's flush markers make such streams.
inline uint64_t k1_pool_pages_absolute(uint64_t sum_clen, uint64_t sum_usize, uint64_t n_members)
{
	const uint64_t words = 4 * sum_usize + 64 * sum_clen + 128 * n_members;
	return words / (K1_PAGE_WORDS - 4) + 3 * n_members + 64;
}

} // namespace ngsqc
