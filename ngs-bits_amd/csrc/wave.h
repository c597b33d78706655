// Wave-level primitives of the K1 kernels (gfx950, wave64). k1_kernels.h is written against this small vocabulary only, so that
// the same kernel text also runs under the wave emulator of the test suite (tests/emul/wave_emul.h: 64 fibers per wave on the
// CPU, every cross-lane operation a rendezvous) - a way to check the kernels' logic without a GPU, not a product path: the
// library only ever contains what this header maps to.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#define K1_KERNEL(bounds) __global__ __launch_bounds__(bounds)
#define K1_KERNEL_OCC(bounds, waves_per_simd) __global__ __launch_bounds__(bounds, waves_per_simd)   // + a register budget for that many waves per SIMD
#define K1_SHARED __shared__
#define K1_DEV __device__ __forceinline__
#define K1_STAT(i) ((void)0)   // (instrumentation hooks of the wave emulator)
#define K1_WSTAT(i) ((void)0)

namespace ngsqc { namespace wv {

using u32x4 = uint4;
K1_DEV u32x4 make4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return make_uint4(a, b, c, d); }

K1_DEV int lane() { return (int)(threadIdx.x & 63u); }
K1_DEV int64_t block_id() { return (int64_t)blockIdx.x; }
K1_DEV int64_t grid_size() { return (int64_t)gridDim.x; }

// ---- cross-lane (call in wave-uniform control flow only) ----
K1_DEV uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
K1_DEV uint32_t readlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
K1_DEV uint32_t shfl(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v); }
K1_DEV void barrier() { __builtin_amdgcn_wave_barrier(); }   // orders LDS traffic between the lanes of the wave (a compiler fence: the wave runs in lockstep)
// inclusive prefix sum on the DPP network: Hillis-Steele inside each 16-lane row (row_shr:1/2/4/8, out-of-row sources read 0),
// then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2-3
K1_DEV uint32_t scan_incl(uint32_t x)
{
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
	return x;
}

// ---- memory ----
K1_DEV void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }
K1_DEV void wait_vm4() { __builtin_amdgcn_s_waitcnt(0x0F74); }   // vmcnt(4): all but the last four vector-memory operations are done (loads return in order)   // vmcnt(0): every vector-memory load has returned, every store is acknowledged
K1_DEV unsigned long long atomic_inc(unsigned long long* p) { return atomicAdd(p, 1ull); }
K1_DEV uint32_t atomic_add_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
K1_DEV void lds_or(unsigned long long* p, unsigned long long v) { atomicOr(p, v); }
K1_DEV void lds_or32(uint32_t* p, uint32_t v) { atomicOr(p, v); }
// a dword of LDS at any byte address (gfx950 runs with unaligned access mode: one ds_read_b32) / at a 4-aligned one
K1_DEV uint32_t lds_load32u(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
K1_DEV uint32_t lds_load32(const uint8_t* p) { return *(const uint32_t*)p; }
K1_DEV void lds_store32(uint8_t* p, uint32_t v) { *(uint32_t*)p = v; }
K1_DEV void lds_store32u(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
K1_DEV uint64_t lds_load64u(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
K1_DEV void lds_store64u(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
K1_DEV uint64_t lds_load64(const uint8_t* p) { return *(const uint64_t*)p; }   // (8-aligned)
K1_DEV void lds_store16u(uint8_t* p, uint32_t v) { const uint16_t h = (uint16_t)v; __builtin_memcpy(p, &h, 2); }

// A byte range in HBM behind a buffer resource: 32-bit offsets (one VALU add per address) and hardware bounds clamping
// (loads outside read 0, stores outside are dropped). The descriptor lives in SGPRs: its inputs are made wave-uniform first.
struct ByteBuf
{
	__amdgpu_buffer_rsrc_t rs;
	K1_DEV static ByteBuf make(uint8_t* p, uint32_t bytes)
	{
		const uint64_t a = (uint64_t)p;
		const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
		ByteBuf b; b.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
		return b;
	}
	K1_DEV uint32_t load(uint32_t off) const { return __builtin_amdgcn_raw_buffer_load_b8(rs, (int)off, 0, 0); }
	K1_DEV void store(uint32_t off, uint32_t v) const { __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, rs, (int)off, 0, 0); }
	// a dword at any byte offset (in range: off + 4 <= bytes; a dword that is not wholly inside the range reads 0 / is dropped)
	K1_DEV uint32_t load32(uint32_t off) const { return __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0); }
	K1_DEV void store32(uint32_t off, uint32_t v) const { __builtin_amdgcn_raw_buffer_store_b32(v, rs, (int)off, 0, 0); }
	K1_DEV void store64(uint32_t off, uint64_t v) const { typedef uint32_t v2 __attribute__((ext_vector_type(2))); v2 x; x.x = (uint32_t)v; x.y = (uint32_t)(v >> 32); __builtin_amdgcn_raw_buffer_store_b64(x, rs, (int)off, 0, 0); }   // (8 bytes at any byte offset, wholly inside the range)
	K1_DEV uint64_t load64(uint32_t off) const { typedef uint32_t v2 __attribute__((ext_vector_type(2))); const v2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0); return ((uint64_t)r.y << 32) | r.x; }
};

// wave priority for instruction arbitration on its SIMD (0..3)
K1_DEV void set_priority(int p) { if (p == 1) __builtin_amdgcn_s_setprio(1); else if (p == 2) __builtin_amdgcn_s_setprio(2); else if (p >= 3) __builtin_amdgcn_s_setprio(3); }

// ---- bit arithmetic ----
K1_DEV uint32_t brev(uint32_t x) { return __brev(x); }
K1_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }   // ({hi,lo} >> (sh & 31)) & 0xffffffff
K1_DEV uint32_t bfe(uint32_t x, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(x, off, width); }        // (x >> off) & ((1 << width) - 1), width 0 -> 0
K1_DEV uint32_t popc64(uint64_t x) { return (uint32_t)__popcll(x); }
K1_DEV uint32_t mbcnt(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }   // bits of m below this lane
K1_DEV uint32_t mbcnt_add(uint64_t m, uint32_t a) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, a)); }   // a + bits of m below this lane
K1_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
K1_DEV uint32_t bcnt(uint32_t x) { return (uint32_t)__popc(x); }
K1_DEV uint32_t ctz32(uint32_t x) { return (uint32_t)__builtin_ctz(x); }     // x != 0
K1_DEV uint32_t clz32(uint32_t x) { return (uint32_t)__builtin_clz(x); }     // x != 0
K1_DEV uint32_t ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }   // x != 0
K1_DEV uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }   // v_perm_b32: result byte k = byte sel.byte[k] of {hi, lo} (0..3: lo, 4..7: hi)
// (x & m) | y in ONE instruction, m a constant kept in a scalar register (v_and_or_b32: a VOP3 takes no literal, and the compiler turns an OR of disjoint bits into
// an ADD it cannot fuse with the AND - the decoder's LDS addresses are made of exactly that)
K1_DEV uint32_t and_or(uint32_t x, uint32_t m, uint32_t y) { uint32_t r; asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(m), "v"(y)); return r; }
K1_DEV uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }               // v_bfi_b32

} } // namespace ngsqc::wv
