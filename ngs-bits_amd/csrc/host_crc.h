// CRC-32 (gzip polynomial, reflected) of a host buffer: the check sums of CRAM blocks and containers, and of the stored BGZF members the CRAM path frames its records in
// (csrc/cram.hip). zlib's crc32 runs at about 1 GB/s on the build host; with the carry-less multiplication of x86 (PCLMULQDQ) the message is folded 64 bytes at a time
// (V. Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009: the constants below are x^k mod P of that paper for this
// polynomial) - what htslib gets from libdeflate / zlib-ng under the reference's BamReader. Without the instruction (checked once at run time): zlib.
#pragma once
#include <cstddef>
#include <cstdint>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace ngsqc {

#if defined(__x86_64__)
#define NGSQC_CLMUL_TARGET __attribute__((target("pclmul,sse4.1")))
NGSQC_CLMUL_TARGET inline __m128i crc_fold(__m128i acc, __m128i k, __m128i next)   // acc * x^(distance) + next
{
	return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc, k, 0x00), _mm_clmulepi64_si128(acc, k, 0x11)), next);
}
NGSQC_CLMUL_TARGET inline __m128i crc_ld(const uint8_t* q) { return _mm_loadu_si128(reinterpret_cast<const __m128i*>(q)); }
// state ~crc in, state out; n >= 64 and a multiple of 16
NGSQC_CLMUL_TARGET inline uint32_t crc32_fold_pclmul(const uint8_t* p, size_t n, uint32_t state)
{
	const __m128i k_64 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);    // x^(512+64), x^512 mod P: four lanes, 64 bytes apart
	const __m128i k_16 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);    // x^(128+64), x^128 mod P: one lane onto the next 16 bytes
	const __m128i k_fin = _mm_set_epi64x(0, 0x0163cd6124);              // x^64 mod P
	const __m128i k_bar = _mm_set_epi64x(0x01f7011641, 0x01db710641);   // Barrett: mu, P
	__m128i a = _mm_xor_si128(crc_ld(p), _mm_cvtsi32_si128((int)state)), b = crc_ld(p + 16), c = crc_ld(p + 32), d = crc_ld(p + 48);
	p += 64; n -= 64;
	for (; n >= 64; p += 64, n -= 64) { a = crc_fold(a, k_64, crc_ld(p)); b = crc_fold(b, k_64, crc_ld(p + 16)); c = crc_fold(c, k_64, crc_ld(p + 32)); d = crc_fold(d, k_64, crc_ld(p + 48)); }
	a = crc_fold(a, k_16, b); a = crc_fold(a, k_16, c); a = crc_fold(a, k_16, d);
	for (; n >= 16; p += 16, n -= 16) a = crc_fold(a, k_16, crc_ld(p));
	// 128 -> 64 -> 32 bits
	const __m128i lo32 = _mm_setr_epi32(~0, 0, ~0, 0);
	__m128i t = _mm_xor_si128(_mm_srli_si128(a, 8), _mm_clmulepi64_si128(a, k_16, 0x10));
	t = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(t, lo32), k_fin, 0x00), _mm_srli_si128(t, 4));
	__m128i u = _mm_clmulepi64_si128(_mm_and_si128(t, lo32), k_bar, 0x10);
	u = _mm_clmulepi64_si128(_mm_and_si128(u, lo32), k_bar, 0x00);
	return (uint32_t)_mm_extract_epi32(_mm_xor_si128(t, u), 1);
}
#endif

inline uint32_t host_crc32(const uint8_t* p, size_t n)
{
#if defined(__x86_64__)
	static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
	if (have && n >= 64)
	{
		const size_t body = n & ~(size_t)15;
		const uint32_t s = crc32_fold_pclmul(p, body, 0xFFFFFFFFu);   // (the state after the body; zlib's interface takes and gives the complemented value)
		return (uint32_t)crc32((uLong)(s ^ 0xFFFFFFFFu), p + body, (uInt)(n - body));
	}
#endif
	uLong c = crc32(0L, Z_NULL, 0);
	while (n) { const uInt k = (uInt)(n > (1u << 30) ? (1u << 30) : n); c = crc32(c, p, k); p += k; n -= k; }   // (zlib takes 32-bit lengths)
	return (uint32_t)c;
}

} // namespace ngsqc
