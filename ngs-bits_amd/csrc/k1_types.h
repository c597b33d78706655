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

// Token stream between phase 1 and phase 2: groups of four 32-bit words, one group per 16-byte store of a decoder lane.
//   literal : the byte value (bits 8..31 zero)
//   match   : bit 31 | (length - 3) << 23 | (distance - 1)           (bits 15..22 zero)
//   no-op   : 0xffffffff  (a trip in which the lane produced nothing; only inside a group that also holds real tokens)
constexpr uint32_t K1_TOK_NOOP = 0xffffffffu;

} // namespace ngsqc
