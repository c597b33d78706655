// dev probe: do gfx950 global / LDS accesses of 4 and 16 bytes work at every byte alignment (what the compiler assumes on amdhsa)?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int off)
{
	__shared__ uint8_t s[8192];
	const int l = threadIdx.x;
	for (int i = l; i < 8192; i += 64) s[i] = 0;
	__builtin_amdgcn_wave_barrier();
	uint32_t v; __builtin_memcpy(&v, in + off + 4 * l, 4);                 // unaligned global dword load
	__builtin_memcpy(s + off + 4 * l, &v, 4);                              // unaligned LDS dword store
	__builtin_amdgcn_wave_barrier();
	uint32_t w; __builtin_memcpy(&w, s + off + 4 * l, 4);                  // unaligned LDS dword load
	__builtin_memcpy(out + off + 4 * l, &w, 4);                            // unaligned global dword store
	uint4 g; __builtin_memcpy(&g, in + 1024 + off + 16 * l, 16);           // unaligned global 16-byte load
	__builtin_memcpy(s + 1024 + off + 16 * l, &g, 16);                     // unaligned LDS 16-byte store
	__builtin_amdgcn_wave_barrier();
	uint4 q; __builtin_memcpy(&q, s + 1024 + off + 16 * l, 16);            // unaligned LDS 16-byte load
	__builtin_memcpy(out + 1024 + off + 16 * l, &q, 16);                   // unaligned global 16-byte store
	unsigned long long d; __builtin_memcpy(&d, s + 1024 + off + 8 * l, 8); // unaligned LDS 8-byte load
	__builtin_memcpy(out + 4096 + off + 8 * l, &d, 8);
}
int main()
{
	std::vector<uint8_t> h(8192), r(8192);
	for (int i = 0; i < 8192; ++i) h[i] = (uint8_t)(i * 131 + 7);
	uint8_t *di, *dout; hipMalloc(&di, 8192); hipMalloc(&dout, 8192); hipMemcpy(di, h.data(), 8192, hipMemcpyHostToDevice);
	int bad = 0;
	for (int off = 0; off < 16; ++off)
	{
		hipMemset(dout, 0, 8192);
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, off);
		if (hipDeviceSynchronize() != hipSuccess) { printf("off %d: launch failed\n", off); return 2; }
		hipMemcpy(r.data(), dout, 8192, hipMemcpyDeviceToHost);
		const bool a = !memcmp(r.data() + off, h.data() + off, 256), b = !memcmp(r.data() + 1024 + off, h.data() + 1024 + off, 1024), c = !memcmp(r.data() + 4096 + off, h.data() + 1024 + off, 512);
		printf("off %2d: dword %s  x4 %s  lds8 %s\n", off, a ? "ok" : "BAD", b ? "ok" : "BAD", c ? "ok" : "BAD");
		bad += !(a && b && c);
	}
	printf("unaligned probe: %s\n", bad ? "FAILED" : "all ok");
	return bad ? 1 : 0;
}
