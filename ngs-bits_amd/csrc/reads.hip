// Raw-read QC pass: StatisticsReads::update(const BamAlignment&) (src/cppNGS/StatisticsReads.cpp:83-158), the loop behind
// `MappingQC -read_qc` (src/MappingQC/main.cpp:83-98). HBM-streaming histogramming over SEQ and QUAL of every primary
// record: ONE WAVE PER RECORD, one lane per cycle (64 consecutive quality bytes per load). Per-cycle base counts and
// quality sums of the first RQ_CYC cycles live in registers of the lane that owns the cycle (a wave only touches HBM for
// them once, at the end); quality histograms are LDS atomics, flushed once per workgroup; the read-length histogram is
// run-length cached per wave (sorted WGS reads almost all have the same length). All arithmetic is integer except the
// per-read mean quality, which is the reference's double division followed by its round / floor.
#include "common.h"
#include <algorithm>

namespace ngsqc {

namespace {
__device__ __forceinline__ uint32_t rd32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x)
{
	#pragma unroll
	for (int o = 32; o > 0; o >>= 1) x += (uint32_t)__shfl_xor((int)x, o);
	return x;
}
} // namespace

// max l_seq over the records the pass will count (not secondary / supplementary)
__global__ __launch_bounds__(256) void reads_max_kernel(const uint8_t* __restrict__ infl, const int64_t* __restrict__ recoff, long long n_rec, unsigned long long* __restrict__ out_max)
{
	long long m = 0;
	for (long long li = (long long)blockIdx.x * blockDim.x + threadIdx.x; li < n_rec; li += (long long)gridDim.x * blockDim.x)
	{
		const uint8_t* p = infl + recoff[li];
		const uint32_t flag = rd32(p + 16) >> 16;
		if (flag & (0x100 | 0x800)) continue;
		const long long l = (int32_t)rd32(p + 20);
		m = l > m ? l : m;
	}
	#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { long long t = __shfl_xor(m, o); m = t > m ? t : m; }
	if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out_max, (unsigned long long)m);
}

// acc layout (u64): see ReadsAcc in common.h
__global__ __launch_bounds__(256) void reads_kernel(const uint8_t* __restrict__ infl, const int64_t* __restrict__ recoff, long long n_rec, int single_end,
                                                    unsigned long long* __restrict__ acc, unsigned long long* __restrict__ len_hist, long long len_cap,
                                                    unsigned long long* __restrict__ cyc)
{
	__shared__ uint32_t s_bq[100], s_rq[100], s_qd[120];
	for (int i = threadIdx.x; i < 100; i += 256) { s_bq[i] = 0; s_rq[i] = 0; }
	for (int i = threadIdx.x; i < 120; i += 256) s_qd[i] = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
	uint32_t cA[RQ_PASSES], cC[RQ_PASSES], cG[RQ_PASSES], cT[RQ_PASSES], cN[RQ_PASSES], q1[RQ_PASSES], q2[RQ_PASSES];
	#pragma unroll
	for (int k = 0; k < RQ_PASSES; ++k) { cA[k] = cC[k] = cG[k] = cT[k] = cN[k] = q1[k] = q2[k] = 0; }
	unsigned long long tA = 0, tC = 0, tG = 0, tT = 0, tN = 0;        // bases behind the per-cycle window (per lane)
	unsigned long long n_fwd = 0, n_rev = 0, n_bases = 0, bad_base = 0, bad_qual = 0;   // wave-uniform
	long long run_len = -1; unsigned long long run_cnt = 0;
	for (long long li = wave; li < n_rec; li += n_waves)
	{
		const uint8_t* p = infl + recoff[li];
		const uint32_t w = rd32(p + 12), w2 = rd32(p + 16);
		const uint32_t flag = w2 >> 16, l_name = w & 0xff, n_cigar = w2 & 0xffff;
		if (flag & (0x100 | 0x800)) continue;                                   // StatisticsReads.cpp:86
		const bool fwd = single_end ? true : (flag & 0x40) != 0;                // :90-106
		const int cycles = (int32_t)rd32(p + 20);
		if (fwd) ++n_fwd; else ++n_rev;
		if (cycles > 0) n_bases += (unsigned long long)cycles;
		if (cycles == run_len) ++run_cnt;
		else
		{
			if (run_cnt && lane == 0) atomicAdd(&len_hist[run_len < len_cap ? run_len : len_cap], run_cnt);
			run_len = cycles < 0 ? 0 : cycles; run_cnt = 1;
		}
		const uint8_t* seq = p + 36 + l_name + 4ull * n_cigar;
		const uint8_t* qual = seq + ((uint32_t)cycles + 1) / 2;
		uint32_t qsum = 0; bool bb = false, bq = false;
		#pragma unroll
		for (int k = 0; k < RQ_PASSES; ++k)
		{
			const int i = k * 64 + lane;
			if (i < cycles)
			{
				const uint32_t nib = (seq[i >> 1] >> ((~i & 1) << 2)) & 15u, q = qual[i];
				cA[k] += nib == 1; cC[k] += nib == 2; cG[k] += nib == 4; cT[k] += nib == 8; cN[k] += nib == 15;
				bb |= !(nib == 1 || nib == 2 || nib == 4 || nib == 8 || nib == 15);
				qsum += q; if (fwd) q1[k] += q; else q2[k] += q;
				if (q < 100) atomicAdd(&s_bq[q], 1u); else bq = true;
			}
		}
		for (int i = RQ_CYC + lane; i < cycles; i += 64)
		{
			const uint32_t nib = (seq[i >> 1] >> ((~i & 1) << 2)) & 15u, q = qual[i];
			tA += nib == 1; tC += nib == 2; tG += nib == 4; tT += nib == 8; tN += nib == 15;
			bb |= !(nib == 1 || nib == 2 || nib == 4 || nib == 8 || nib == 15);
			qsum += q;
			if (q < 100) atomicAdd(&s_bq[q], 1u); else bq = true;
		}
		qsum = wave_sum_u32(qsum);
		if (__builtin_amdgcn_ballot_w64(bb)) ++bad_base;
		if (__builtin_amdgcn_ballot_w64(bq)) ++bad_qual;
		if (lane == 0 && cycles > 0)
		{
			const double mean = (double)qsum / (double)cycles;                 // :150
			const long long r = (long long)round(mean);                        // read_qualities_[std::round(mean)]++
			if (r >= 0 && r < 100) atomicAdd(&s_rq[r], 1u);
			double v = mean < 0.0 ? 0.0 : (mean > 60.0 ? 60.0 : mean);           // Histogram(0,60,1).inc(mean, true)
			int b = (int)floor((v - 0.0) / (60.0 - 0.0) * 60.0); b = b < 0 ? 0 : (b > 59 ? 59 : b);
			atomicAdd(&s_qd[(fwd ? 0 : 60) + b], 1u);
		}
	}
	if (run_cnt && lane == 0) atomicAdd(&len_hist[run_len < len_cap ? run_len : len_cap], run_cnt);
	// ---- flush ----
	#pragma unroll
	for (int k = 0; k < RQ_PASSES; ++k)
	{
		unsigned long long* c = cyc + 7ull * (k * 64 + lane);
		if (cA[k]) atomicAdd(c + 0, (unsigned long long)cA[k]); if (cC[k]) atomicAdd(c + 1, (unsigned long long)cC[k]);
		if (cG[k]) atomicAdd(c + 2, (unsigned long long)cG[k]); if (cT[k]) atomicAdd(c + 3, (unsigned long long)cT[k]);
		if (cN[k]) atomicAdd(c + 4, (unsigned long long)cN[k]);
		if (q1[k]) atomicAdd(c + 5, (unsigned long long)q1[k]); if (q2[k]) atomicAdd(c + 6, (unsigned long long)q2[k]);
		tA += cA[k]; tC += cC[k]; tG += cG[k]; tT += cT[k]; tN += cN[k];
	}
	#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { tA += __shfl_xor(tA, o); tC += __shfl_xor(tC, o); tG += __shfl_xor(tG, o); tT += __shfl_xor(tT, o); tN += __shfl_xor(tN, o); }
	if (lane == 0)
	{
		atomicAdd(&acc[RA_FWD], n_fwd); atomicAdd(&acc[RA_REV], n_rev); atomicAdd(&acc[RA_BASES], n_bases);
		atomicAdd(&acc[RA_A], tA); atomicAdd(&acc[RA_C], tC); atomicAdd(&acc[RA_G], tG); atomicAdd(&acc[RA_T], tT); atomicAdd(&acc[RA_N], tN);
		if (bad_base) atomicAdd(&acc[RA_BAD_BASE], bad_base);
		if (bad_qual) atomicAdd(&acc[RA_BAD_QUAL], bad_qual);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < 100; i += 256) { if (s_bq[i]) atomicAdd(&acc[RA_BQ0 + i], (unsigned long long)s_bq[i]); if (s_rq[i]) atomicAdd(&acc[RA_RQ0 + i], (unsigned long long)s_rq[i]); }
	for (int i = threadIdx.x; i < 120; i += 256) if (s_qd[i]) atomicAdd(&acc[RA_QD0 + i], (unsigned long long)s_qd[i]);
}

void launch_reads_max(const uint8_t* infl, const int64_t* recoff, int64_t n_rec, unsigned long long* d_max, hipStream_t s)
{
	if (n_rec <= 0) return;
	const int grid = (int)std::min<int64_t>((n_rec + 255) / 256, 256 * 16);
	hipLaunchKernelGGL(reads_max_kernel, dim3(grid), dim3(256), 0, s, infl, recoff, (long long)n_rec, d_max); KCHECK();
}

void launch_reads(const uint8_t* infl, const int64_t* recoff, int64_t n_rec, int single_end, unsigned long long* d_acc, unsigned long long* d_len_hist, int64_t len_cap,
                  unsigned long long* d_cyc, hipStream_t s)
{
	if (n_rec <= 0) return;
	// 2048 workgroups: the 32-bit LDS bins of a workgroup see n_bases / 2048 increments at most (< 2^32 for any tile that fits HBM)
	const int64_t wgs = std::min<int64_t>((n_rec + 3) / 4, 256 * 8);
	hipLaunchKernelGGL(reads_kernel, dim3((int)wgs), dim3(256), 0, s, infl, recoff, (long long)n_rec, single_end, d_acc, d_len_hist, (long long)len_cap, d_cyc); KCHECK();
}

} // namespace ngsqc
