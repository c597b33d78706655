#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__global__ void k(const uint8_t* infl, int64_t total, int64_t o, int32_t n_ref)
{
	const uint8_t* r = infl + o;
	uint32_t bs = ld32u(r);
	int32_t tid = (int32_t)ld32u(r + 4), pos = (int32_t)ld32u(r + 8);
	uint32_t w = ld32u(r + 12), w2 = ld32u(r + 16);
	int32_t l_seq = (int32_t)ld32u(r + 20), mtid = (int32_t)ld32u(r + 24), mpos = (int32_t)ld32u(r + 28);
	uint32_t l_name = w & 0xff, n_cigar = w2 & 0xffff;
	uint64_t need = 32ull + l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
	printf("o=%lld bs=%u tid=%d pos=%d lname=%u ncig=%u lseq=%d mtid=%d mpos=%d need=%llu last=%d nref=%d total=%lld\n", (long long)o, bs, tid, pos, l_name, n_cigar, l_seq, mtid, mpos, (unsigned long long)need, (int)r[36 + l_name - 1], n_ref, (long long)total);
}
int main()
{
	// one fake record at unaligned offset 3
	std::vector<uint8_t> h(256, 0); size_t o = 3;
	auto p32 = [&](size_t at, uint32_t v) { memcpy(&h[at], &v, 4); };
	p32(o, 100); p32(o + 4, 0); p32(o + 8, 12345); h[o + 12] = 5; h[o + 13] = 60; p32(o + 16, (0x63u << 16) | 1); p32(o + 20, 20); p32(o + 24, 0); p32(o + 28, 777);
	memcpy(&h[o + 36], "abcd", 5);
	uint8_t* d; hipMalloc((void**)&d, 256); hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, (int64_t)256, (int64_t)o, 25);
	hipDeviceSynchronize();
	return 0;
}
