// K1 launchers: the two kernels of k1_kernels.h compiled for gfx950 against the wave vocabulary of wave.h.
#include "common.h"
#include "wave.h"
#include "k1_kernels.h"
#include <cstdlib>
#include <algorithm>

namespace ngsqc {

void launch_huff_tokens(const uint8_t* d_comp, const BlockDesc* d_blocks, int64_t n_blocks, BlockStatus* d_status,
                        const uint64_t* d_tok_off, uint32_t* d_tok, uint32_t* d_tok_count, unsigned long long* d_work, const uint32_t* d_order, int max_wgs, hipStream_t s)
{
	if (n_blocks <= 0) return;
	// d_work: the launch's member queue head (zeroed by the caller). One-wave workgroups of 23 KB LDS: six per CU.
	const int64_t wgs = (n_blocks + 63) / 64;
	const int grid1 = (int)(wgs < max_wgs ? wgs : max_wgs);
	const char* pe = getenv("NGSQC_P1_PARK"); int park_hi = pe ? atoi(pe) : 32;   // (16: 731-743, 32: 756 Mreads/s on a 96 M-read shard)
	const char* pr = getenv("NGSQC_P1_PRIO"); park_hi = (park_hi & 255) | ((pr ? atoi(pr) : 0) << 8);
	hipLaunchKernelGGL(k1::huff_tokens_kernel, dim3(grid1), dim3(64), 0, s, d_comp, d_blocks, n_blocks, d_tok_off, d_tok, d_tok_count, d_status, d_work, d_order, park_hi);
	KCHECK();
}

void launch_lz77_resolve(const BlockDesc* d_blocks, int64_t n_blocks, uint8_t* d_out, BlockStatus* d_status,
                         const uint64_t* d_tok_off, const uint32_t* d_tok, const uint32_t* d_tok_count, hipStream_t s)
{
	if (n_blocks <= 0) return;
	// one member per one-wave workgroup, handed out by the dispatcher (NGSQC_P2_WGS caps the grid: the waves then stride over the members)
	const char* e2 = getenv("NGSQC_P2_WGS"); const int64_t cap2 = e2 ? std::max<int64_t>(1, atoll(e2)) : (int64_t)1 << 20;
	const int grid2 = (int)(n_blocks < cap2 ? n_blocks : cap2);
	const char* ep = getenv("NGSQC_P2_LDS_PAD"); const int pad = ep ? std::max(0, atoi(ep)) : 0;   // measurement switch: extra LDS per workgroup = fewer phase-2 waves beside the decoder waves
	hipLaunchKernelGGL(k1::lz77_groups_kernel, dim3(grid2), dim3(64), pad, s, d_tok, d_tok_off, d_tok_count, d_blocks, n_blocks, d_out, d_status); KCHECK();
}

} // namespace ngsqc
