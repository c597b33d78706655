// CRAM on the device, first codec: the QUALITY arrays. A CRAM slice stores the qualities of its records in one external block (series QS), rANS 4x8 coded
// (CRAMv3 section 13; order 1 in htslib's files) - about half of the bytes of the BAM records the slice decodes to. The host (cram.hip) builds every record with
// its quality bytes left blank and a plan: per block the position of its four rANS states in the CRAM image and its frequency tables in a compact form (the
// symbols that occur - at most 64 - and per context the cumulative frequencies), per record where its qualities go. Here: one lane per block decodes it
// (four interleaved states over one byte stream: sequential by construction; a file has one block per slice, i.e. thousands), then one lane per record copies
// its qualities into the BAM image that K1 reads (stored BGZF members: the payload of member m starts at m * 65311 + 23).
#include "common.h"
#include <cstring>

namespace ngsqc {
namespace {
// One WORKGROUP per block, the job's tables in LDS (a search step is an LDS read, not a dependent global load). The four rANS states live in four lanes - every round
// each lane decodes the symbol of its state,
// the lanes count the bytes their renormalisation takes (0, 1 or 2), a prefix over the four lanes gives each its place in the shared byte stream (the order the
// sequential decoder reads them in: state 0 first), and the stream pointer moves on by the sum. Order 1 writes four quarters of the output, one per lane; what is left
// behind the quarters belongs to state 3 alone.
// The symbol of a state by bisection over the cumulative row (6 LDS reads for 64 symbols). (Round 4 went through four versions - one lane per block with the
// tables in global memory, one lane with the tables in LDS, four lanes with a scan from the front: 800 -> 480 -> 150 -> 38 ms for the test twin's largest block,
// profiles/r04_cram_device_quals.txt; only the last one is kept.)
__global__ __launch_bounds__(64) void cram_rans_lds_kernel(const uint8_t* __restrict__ in, const CramQualPlan::Job* __restrict__ jobs, int n_jobs, const uint16_t* __restrict__ tabs,
                                                                             const uint8_t* __restrict__ syms, uint8_t* __restrict__ out, unsigned int* __restrict__ status)
{
	__shared__ uint16_t sC[65 * 64]; __shared__ uint8_t sSym[64]; __shared__ int sK0;
	const int j = (int)blockIdx.x, lane = (int)threadIdx.x;
	if (j >= n_jobs) return;
	const CramQualPlan::Job jb = jobs[j];
	const int ns = (int)jb.nsym, row = ns + 1, rows = jb.order ? ns : 1;
	if (jb.in_len < 16 || ns < 1 || ns > 64) { if (lane == 0) atomicOr(status, 1u); return; }
	for (int x = lane; x < rows * row; x += 64) sC[x] = tabs[jb.tab_off + x];
	if (lane < ns) sSym[lane] = syms[jb.sym_off + lane];
	if (lane == 0) sK0 = syms[jb.sym_off + 64];   // the row of context 0
	__syncthreads();
	if (lane >= 4) return;
	const uint8_t* p = in + jb.in_off; const uint8_t* const end = p + jb.in_len;
	uint8_t* const o = out + jb.out_off; const uint32_t n = jb.n_out;
	bool bad = false;
	auto sym_of = [&](uint32_t x, const uint16_t* C, uint32_t& v) -> int {   // the symbol index of state x in row C; v: the state behind it, before renormalisation
		const uint32_t m = x & 0xfffu; int k = 0;
		// the LAST k with C[k] <= m: behind it C[k + 1] > m, so that symbol has a frequency (symbols without one repeat the value of their successor)
		int hi = ns;
		while (hi - k > 1) { const int mid = (k + hi) >> 1; if ((uint32_t)C[mid] <= m) k = mid; else hi = mid; }
		const uint32_t c0 = C[k], f = (uint32_t)C[k + 1] - c0;
		if (f == 0 || m < c0 || m >= (uint32_t)C[k + 1]) { bad = true; v = x; return 0; }
		v = f * (x >> 12) + m - c0;
		return k;
	};
	{
		uint32_t x = (uint32_t)p[4 * lane] | ((uint32_t)p[4 * lane + 1] << 8) | ((uint32_t)p[4 * lane + 2] << 16) | ((uint32_t)p[4 * lane + 3] << 24);
		p += 16;   // (every lane tracks the shared stream pointer)
		// one round: the lane's symbol (when it has one), then the renormalisation bytes in the order of the states
		auto round = [&](bool active, const uint16_t* C) -> int {
			uint32_t v = x; int s = 0, cnt = 0;
			if (active) { s = sym_of(x, C, v); uint32_t t = v; while (t < (1u << 23) && cnt < 3) { t <<= 8; ++cnt; } }
			const int c0 = __shfl(cnt, 0), c1 = __shfl(cnt, 1), c2 = __shfl(cnt, 2), c3 = __shfl(cnt, 3);
			const int my = lane == 0 ? 0 : lane == 1 ? c0 : lane == 2 ? c0 + c1 : c0 + c1 + c2;
			if (p + c0 + c1 + c2 + c3 > end) bad = true;
			else for (int b = 0; b < cnt; ++b) v = (v << 8) | p[my + b];
			p += c0 + c1 + c2 + c3;
			if (active) x = v;
			return s;
		};
		if (jb.order == 0)
		{
			for (uint32_t i = 0; i < n; i += 4)
			{
				const bool act = i + (uint32_t)lane < n;
				const int s = round(act, sC);
				if (act && !bad) o[i + (uint32_t)lane] = sSym[s];
				if (__any(bad)) break;
			}
		}
		else
		{
			const uint32_t q = n >> 2; uint32_t idx = (uint32_t)lane * q; int pk = sK0;
			if (pk >= ns) bad = true;
			for (uint32_t i = 0; i < q; ++i)
			{
				const int s = round(!bad, sC + pk * row);
				if (!bad) { o[idx++] = sSym[s]; pk = s; }
				if (__any(bad)) break;
			}
			if (lane == 3 && !bad)   // what the quarters leave over: state 3 alone, bytes one after the other
				while (idx < n)
				{
					uint32_t v; const int s = sym_of(x, sC + pk * row, v);
					while (v < (1u << 23)) { if (p >= end) { bad = true; break; } v = (v << 8) | *p++; }
					if (bad) break;
					x = v; o[idx++] = sSym[s]; pk = s;
				}
		}
	}
	if (bad) atomicOr(status, 2u);
}

__global__ __launch_bounds__(256) void cram_patch_kernel(const CramQualPlan::Patch* __restrict__ P, int64_t n, const uint8_t* __restrict__ qs, uint64_t qs_bytes, uint8_t* __restrict__ image, uint64_t image_bytes,
                                                         unsigned int* __restrict__ status)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const CramQualPlan::Patch p = P[i];
	if (p.src + p.len > qs_bytes) { atomicOr(status, 4u); return; }
	for (uint32_t b = 0; b < p.len; ++b)
	{
		const uint64_t s = p.dst + b, at = (s / 65280ull) * 65311ull + 23ull + (s % 65280ull);
		if (at >= image_bytes) { atomicOr(status, 8u); return; }
		image[at] = qs[p.src + b];
	}
}

template <typename T> T* dev_copy(const T* host, size_t n, hipStream_t s)
{
	T* d = nullptr; HIPCHK(hipMalloc((void**)&d, std::max<size_t>(n, 1) * sizeof(T)));
	if (n) HIPCHK(hipMemcpyAsync(d, host, n * sizeof(T), hipMemcpyHostToDevice, s));
	return d;
}
} // namespace

double cram_device_quals(const uint8_t* cram_image, const CramQualPlan& plan, uint8_t* d_image, size_t image_bytes, hipStream_t s)
{
	if (plan.jobs.empty()) return 0.0;
	// the blocks' byte streams, packed one behind the other (the only part of the CRAM file the device sees)
	std::vector<CramQualPlan::Job> jobs = plan.jobs; std::vector<uint8_t> in; size_t total = 0;
	for (const auto& j : jobs) total += j.in_len;
	in.reserve(total);
	for (auto& j : jobs) { const uint64_t at = in.size(); in.insert(in.end(), cram_image + j.in_off, cram_image + j.in_off + j.in_len); j.in_off = at; }
	uint8_t* d_in = dev_copy(in.data(), in.size(), s); CramQualPlan::Job* d_jobs = dev_copy(jobs.data(), jobs.size(), s);
	uint16_t* d_tabs = dev_copy(plan.tabs.data(), plan.tabs.size(), s); uint8_t* d_syms = dev_copy(plan.syms.data(), plan.syms.size(), s);
	CramQualPlan::Patch* d_patch = dev_copy(plan.patches.data(), plan.patches.size(), s);
	uint8_t* d_out = nullptr; HIPCHK(hipMalloc((void**)&d_out, std::max<uint64_t>(plan.out_bytes, 1)));
	unsigned int* d_status = nullptr; HIPCHK(hipMalloc((void**)&d_status, sizeof(unsigned int))); HIPCHK(hipMemsetAsync(d_status, 0, sizeof(unsigned int), s));
	hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
	HIPCHK(hipEventRecord(e0, s));
	hipLaunchKernelGGL(cram_rans_lds_kernel, dim3((unsigned)jobs.size()), dim3(64), 0, s, d_in, d_jobs, (int)jobs.size(), d_tabs, d_syms, d_out, d_status);
	KCHECK();
	if (!plan.patches.empty())
	{
		hipLaunchKernelGGL(cram_patch_kernel, dim3((unsigned)((plan.patches.size() + 255) / 256)), dim3(256), 0, s, d_patch, (int64_t)plan.patches.size(), d_out, (uint64_t)plan.out_bytes, d_image, (uint64_t)image_bytes, d_status); KCHECK();
	}
	HIPCHK(hipEventRecord(e1, s));
	unsigned int st = 0; HIPCHK(hipMemcpyAsync(&st, d_status, sizeof st, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
	float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	(void)hipFree(d_in); (void)hipFree(d_jobs); (void)hipFree(d_tabs); (void)hipFree(d_syms); (void)hipFree(d_patch); (void)hipFree(d_out); (void)hipFree(d_status);
	if (st) throw std::runtime_error("a quality block of the CRAM file does not decode on the device (rANS status " + std::to_string(st) + ")");
	return (double)ms;
}
} // namespace ngsqc
