// K1 launchers: the two kernels of k1_kernels.h compiled for gfx950 against the wave vocabulary of wave.h.
#include "common.h"
#include "wave.h"
#include "k1_kernels.h"
#include <cstdlib>
#include <algorithm>

namespace ngsqc {

// NGSQC_P1_PARK (tests: the slow section entered by few / many waiting lanes), read when a handle is opened. (Round 5: the other schedule switches of rounds 3-4 -
// decoder priority, LDS padding of either phase, a capped phase-2 grid - all measured within +-2 % of the defaults, profiles/r04_probe_schedule.txt - are gone.)
static struct { int park_hi = 32; int p1_prio = 0; } g_sw;
void k1_read_switches()
{
	const char* e;
	g_sw.park_hi = (e = getenv("NGSQC_P1_PARK")) ? atoi(e) : 32;                       // lanes that wait for the slow section before the wave enters it
	g_sw.p1_prio = (e = getenv("NGSQC_P1_PRIO")) ? std::min(3, std::max(0, atoi(e))) : 0;   // (dev) s_setprio of the decoder waves
}

void launch_huff_tokens(const uint8_t* d_comp, const BlockDesc* d_blocks, int64_t n_blocks, BlockStatus* d_status,
                        uint32_t* d_pool, uint32_t pool_pages, uint32_t* d_pool_ctr, uint32_t* d_tok_first, uint32_t* d_tok_count,
                        unsigned long long* d_work, const uint32_t* d_order, int max_wgs, hipStream_t s)
{
	if (n_blocks <= 0) return;
	// d_work: the launch's member queue head, d_pool_ctr: pages taken from the launch's token pool (both zeroed by the caller). One-wave workgroups.
	const int64_t wgs = (n_blocks + 63) / 64;
	const int grid1 = (int)(wgs < max_wgs ? wgs : max_wgs);
	hipLaunchKernelGGL(k1::huff_tokens_kernel, dim3(grid1), dim3(64), 0, s, d_comp, d_blocks, n_blocks, d_pool, pool_pages, d_pool_ctr, d_tok_first, d_tok_count, d_status, d_work, d_order, (g_sw.park_hi & 255) | (g_sw.p1_prio << 8));
	KCHECK();
}

void launch_lz77_resolve(const BlockDesc* d_blocks, int64_t n_blocks, uint8_t* d_out, BlockStatus* d_status,
                         const uint32_t* d_pool, const uint32_t* d_tok_first, const uint32_t* d_tok_count, const uint8_t* d_comp, hipStream_t s)
{
	if (n_blocks <= 0) return;
	// one member per one-wave workgroup, handed out by the dispatcher (the waves stride over the members of a grid that is capped)
	const int grid2 = (int)std::min<int64_t>(n_blocks, (int64_t)1 << 20);
	hipLaunchKernelGGL(k1::lz77_groups_kernel, dim3(grid2), dim3(64), 0, s, d_pool, d_tok_first, d_tok_count, d_blocks, n_blocks, d_out, d_status, d_comp); KCHECK();
}

} // namespace ngsqc
