// K6 — consumers of the per-base depth array left in HBM by the scan:
//   depth histogram + half-depth count   Statistics.cpp:606-645, 1185-1204
//   per-line depth sums (BedCoverage)    WorkerAverageCoverage.cpp:47-55,135-155  (sum of overlaps == sum of depth over the line)
//   threshold run-length extraction      WorkerLowOrHighCoverage.cpp:76-107, 210-235
// Layout: region i owns slots [doff_i, doff_i+len_i] — len_i bases plus one spare slot that receives the "-1 behind the
// end" of the difference array; after the prefix sum the spare slot is overwritten with -1 so streaming kernels skip it.
#include "common.h"

namespace ngsqc {

__global__ void depth_mark_spare(int32_t* depth, const int64_t* __restrict__ doff, const int32_t* __restrict__ len, int64_t n_regions)
{
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_regions) depth[doff[i] + len[i]] = -1;
}

__global__ __launch_bounds__(256) void depth_hist_kernel(const int32_t* __restrict__ depth, int64_t n_slots, int32_t cap, long long half,
                                                         unsigned long long* __restrict__ hist, unsigned long long* __restrict__ covered)
{
	extern __shared__ uint32_t lh[];
	for (int i = threadIdx.x; i <= cap; i += blockDim.x) lh[i] = 0;
	__syncthreads();
	long long cov = 0;
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += stride)
	{
		int32_t d = depth[s];
		if (d < 0) continue;
		atomicAdd(&lh[d < cap ? d : cap], 1u);
		if ((long long)d >= half) ++cov;
	}
	for (int o = 32; o > 0; o >>= 1) cov += __shfl_xor(cov, o);
	if ((threadIdx.x & 63) == 0 && cov) atomicAdd(covered, (unsigned long long)cov);
	__syncthreads();
	for (int i = threadIdx.x; i <= cap; i += blockDim.x) if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

// concatenated per-base depth without the spare slots: one workgroup per region
__global__ __launch_bounds__(256) void depth_compact_kernel(const int32_t* __restrict__ depth, const int64_t* __restrict__ doff, const int32_t* __restrict__ len,
                                                            int64_t n_regions, int32_t* __restrict__ out)
{
	for (int64_t i = blockIdx.x; i < n_regions; i += gridDim.x)
	{
		const int32_t* src = depth + doff[i]; int32_t* dst = out + (doff[i] - i);
		for (int j = threadIdx.x; j < len[i]; j += blockDim.x) dst[j] = src[j];
	}
}

// one wave per line: sum of depth over slots [slot, slot+n)
__global__ __launch_bounds__(256) void line_sums_kernel(const int32_t* __restrict__ depth, const int64_t* __restrict__ slot, const int32_t* __restrict__ n,
                                                        int64_t n_lines, long long* __restrict__ sums)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
	for (int64_t l = wave; l < n_lines; l += n_waves)
	{
		const int32_t* d = depth + slot[l]; long long acc = 0;
		for (int j = lane; j < n[l]; j += 64) acc += d[j];
		for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
		if (lane == 0) sums[l] = acc;
	}
}

// runs of pred(depth) inside each line; WRITE=false counts runs per line, WRITE=true emits (line,start,end) at base[line]
template <bool WRITE>
__global__ __launch_bounds__(256) void line_runs_kernel(const int32_t* __restrict__ depth, const int64_t* __restrict__ slot, const int32_t* __restrict__ n,
                                                        const int32_t* __restrict__ line_start, int64_t n_lines, int32_t cutoff, int32_t is_high, int32_t sat,
                                                        uint32_t* __restrict__ cnt, const int64_t* __restrict__ base, ngsqc_run* __restrict__ runs)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
	for (int64_t l = wave; l < n_lines; l += n_waves)
	{
		const int32_t* d = depth + slot[l]; const int len = n[l];
		uint32_t n_start = 0, n_end = 0;
		for (int j0 = 0; j0 < len; j0 += 64)
		{
			int j = j0 + lane;
			auto pred = [&](int k) -> bool { if (k < 0 || k >= len) return false; int v = d[k]; if (sat && v > 254) v = 254; return is_high ? v >= cutoff : v < cutoff; };
			bool cur = pred(j), prev = pred(j - 1), next = pred(j + 1);
			bool is_start = cur && !prev, is_end = cur && !next;
			unsigned long long ms = __ballot(is_start), me = __ballot(is_end);
			if (WRITE)
			{
				unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
				if (is_start) { ngsqc_run& r = runs[base[l] + n_start + __popcll(ms & lt)]; r.line = l; r.start = line_start[l] + j; }
				if (is_end) runs[base[l] + n_end + __popcll(me & lt)].end = line_start[l] + j;
			}
			n_start += (uint32_t)__popcll(ms); n_end += (uint32_t)__popcll(me);
		}
		if (!WRITE && lane == 0) cnt[l] = n_start;
	}
}

__global__ __launch_bounds__(256) void depth_add_kernel(int32_t* __restrict__ dst, const int32_t* __restrict__ src, int64_t n)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
void launch_depth_add(int32_t* d_dst, const int32_t* d_src, int64_t n, hipStream_t s)
{
	if (n <= 0) return;
	const int64_t wgs = (n + 255) / 256;
	hipLaunchKernelGGL(depth_add_kernel, dim3((int)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, s, d_dst, d_src, n); KCHECK();
}

void launch_depth_mark_spare(int32_t* d_depth, const int64_t* d_doff, const int32_t* d_len, int64_t n_regions, hipStream_t s)
{
	if (n_regions <= 0) return;
	hipLaunchKernelGGL(depth_mark_spare, dim3((int)((n_regions + 255) / 256)), dim3(256), 0, s, d_depth, d_doff, d_len, n_regions); KCHECK();
}

void launch_depth_hist(const int32_t* d_depth, int64_t n_slots, int32_t cap, int64_t half, unsigned long long* d_hist, unsigned long long* d_cov, hipStream_t s)
{
	if (n_slots <= 0) return;
	int64_t wgs = (n_slots + 256 * 16 - 1) / (256 * 16);
	int grid = (int)(wgs < 1 ? 1 : (wgs < 1024 ? wgs : 1024));
	size_t lds = (size_t)(cap + 1) * sizeof(uint32_t);
	if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)depth_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   // cfDNA: 20 000 bins = 80 KB of the CU's 160 KB
	hipLaunchKernelGGL(depth_hist_kernel, dim3(grid), dim3(256), lds, s, d_depth, n_slots, cap, (long long)half, d_hist, d_cov); KCHECK();
}

void launch_depth_compact(const int32_t* d_depth, const int64_t* d_doff, const int32_t* d_len, int64_t n_regions, int32_t* d_out, hipStream_t s)
{
	if (n_regions <= 0) return;
	int grid = (int)(n_regions < 4096 ? n_regions : 4096);
	hipLaunchKernelGGL(depth_compact_kernel, dim3(grid), dim3(256), 0, s, d_depth, d_doff, d_len, n_regions, d_out); KCHECK();
}

void launch_line_sums(const int32_t* d_depth, const int64_t* d_slot, const int32_t* d_n, int64_t n_lines, long long* d_sums, hipStream_t s)
{
	if (n_lines <= 0) return;
	int64_t wgs = (n_lines + 3) / 4;
	int grid = (int)(wgs < 2048 ? wgs : 2048);
	hipLaunchKernelGGL(line_sums_kernel, dim3(grid), dim3(256), 0, s, d_depth, d_slot, d_n, n_lines, d_sums); KCHECK();
}

void launch_line_runs(bool write, const int32_t* d_depth, const int64_t* d_slot, const int32_t* d_n, const int32_t* d_line_start, int64_t n_lines,
                      int32_t cutoff, int32_t is_high, int32_t sat, uint32_t* d_cnt, const int64_t* d_base, ngsqc_run* d_runs, hipStream_t s)
{
	if (n_lines <= 0) return;
	int64_t wgs = (n_lines + 3) / 4;
	int grid = (int)(wgs < 2048 ? wgs : 2048);
	if (write) hipLaunchKernelGGL(line_runs_kernel<true>, dim3(grid), dim3(256), 0, s, d_depth, d_slot, d_n, d_line_start, n_lines, cutoff, is_high, sat, d_cnt, d_base, d_runs);
	else hipLaunchKernelGGL(line_runs_kernel<false>, dim3(grid), dim3(256), 0, s, d_depth, d_slot, d_n, d_line_start, n_lines, cutoff, is_high, sat, d_cnt, d_base, d_runs);
	KCHECK();
}

} // namespace ngsqc
