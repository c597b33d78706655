// The one exchange of the multi-GPU path, inside the boundary (north_star: "only a final RCCL reduce of the counter vectors over xGMI"): a communicator over
// RCCL for one process per GPU, and the collectives of the shard protocol of include/ngsqc.h - SUM / MAX of the counter vector, all-gather of the 48-byte shard
// summaries, SUM of gc_reads / site counts, and the in-place SUM of the int32 difference array on the library's own device memory. The reference has no
// counterpart (it is a single process: its only parallelism is the worker pool of Statistics.cpp:2614-2638).
// librccl is loaded when the first communicator is made (dlopen), not when libngsqc_hip.so is: the single-GPU tools do not pay for it.
#include "common.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>
#include <mutex>

extern "C" void ngsqc_set_open_error(const char* msg);   // (api.hip: the message behind ngsqc_last_error(NULL))

namespace {

struct Rccl
{
	void* so = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
	std::string err;
};
Rccl& rccl()
{
	static Rccl r; static std::once_flag once;
	std::call_once(once, [] {
		for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (r.so) break; }
		if (!r.so) { const char* e = dlerror(); r.err = std::string("librccl could not be loaded: ") + (e ? e : "?"); return; }   // (dlerror() clears the message: one call)
		auto sym = [&](const char* n) { void* p = dlsym(r.so, n); if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + n; return p; };
		r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId"); r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
		r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy"); r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
		r.AllGather = (decltype(r.AllGather))sym("ncclAllGather"); r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
	});
	return r;
}

}   // namespace

struct ngsqc_comm
{
	ncclComm_t comm = nullptr; int rank = 0, world = 1, device = 0;
	hipStream_t stream = nullptr; void* d_buf = nullptr; size_t d_cap = 0;   // staging of the small host vectors
	std::string err;
	void* staging(size_t bytes)
	{
		if (bytes > d_cap) { if (d_buf) (void)hipFree(d_buf); d_buf = nullptr; d_cap = 0; HIPCHK(hipMalloc(&d_buf, bytes + 4096)); d_cap = bytes + 4096; }
		return d_buf;
	}
};

namespace {

static_assert(sizeof(ncclUniqueId) == NGSQC_COMM_ID_BYTES, "NGSQC_COMM_ID_BYTES is RCCL's unique-id size");

int fail(ngsqc_comm* c, int code, const std::string& msg) { if (c) c->err = msg; ngsqc_set_open_error(msg.c_str()); return code; }
#define RCCLCHK(c, expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) throw std::runtime_error(std::string("RCCL error: ") + rccl().GetErrorString(_r) + " at " #expr); } while (0)

template <class F> int guarded(ngsqc_comm* c, F&& f)
{
	try { if (!c) return fail(nullptr, NGSQC_E_ARG, "no communicator"); HIPCHK(hipSetDevice(c->device)); f(); return NGSQC_OK; }
	catch (std::exception& e) { return fail(c, NGSQC_E_DEVICE, e.what()); }
}
// a host vector through the communicator's device buffer: H2D, the collective(s), D2H
template <class T, class F> void through_device(ngsqc_comm* c, T* v, size_t n, size_t extra_bytes, F&& collective)
{
	T* d = (T*)c->staging(n * sizeof(T) + extra_bytes);
	HIPCHK(hipMemcpyAsync(d, v, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
	collective(d);
	HIPCHK(hipMemcpyAsync(v, d, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
}

}   // namespace

extern "C" {

int ngsqc_comm_unique_id(void* id)
{
	if (!id) return fail(nullptr, NGSQC_E_ARG, "no buffer for the unique id");
	Rccl& r = rccl(); if (!r.err.empty()) return fail(nullptr, NGSQC_E_DEVICE, r.err);
	ncclUniqueId u; const ncclResult_t rc = r.GetUniqueId(&u);
	if (rc != ncclSuccess) return fail(nullptr, NGSQC_E_DEVICE, std::string("RCCL error: ") + r.GetErrorString(rc));
	memcpy(id, &u, sizeof(u)); return NGSQC_OK;
}

int ngsqc_comm_init(int rank, int world, const void* id, int device, ngsqc_comm** out)
{
	if (!out) return NGSQC_E_ARG;
	*out = nullptr;
	if (!id || world < 1 || rank < 0 || rank >= world) return fail(nullptr, NGSQC_E_ARG, "invalid rank / world size / unique id");
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(nullptr, NGSQC_E_DEVICE, "no HIP device available (libngsqc_hip has no CPU fallback)");
	if (device < 0 || device >= n_dev) return fail(nullptr, NGSQC_E_ARG, "invalid HIP device ordinal");
	Rccl& r = rccl(); if (!r.err.empty()) return fail(nullptr, NGSQC_E_DEVICE, r.err);
	ngsqc_comm* c = new ngsqc_comm(); c->rank = rank; c->world = world; c->device = device;
	try
	{
		HIPCHK(hipSetDevice(device));
		HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
		ncclUniqueId u; memcpy(&u, id, sizeof(u));
		RCCLCHK(c, r.CommInitRank(&c->comm, world, u, rank));
	}
	catch (std::exception& e) { const std::string m = e.what(); if (c->stream) (void)hipStreamDestroy(c->stream); delete c; return fail(nullptr, NGSQC_E_DEVICE, m); }
	*out = c; return NGSQC_OK;
}

int ngsqc_comm_rank(const ngsqc_comm* c) { return c ? c->rank : -1; }
int ngsqc_comm_world(const ngsqc_comm* c) { return c ? c->world : 0; }
const char* ngsqc_comm_last_error(const ngsqc_comm* c) { return c ? c->err.c_str() : ""; }

int ngsqc_comm_destroy(ngsqc_comm* c)
{
	if (!c) return NGSQC_OK;
	(void)hipSetDevice(c->device);
	if (c->comm) (void)rccl().CommDestroy(c->comm);
	if (c->d_buf) (void)hipFree(c->d_buf);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c; return NGSQC_OK;
}

// The counter vector of a whole-BAM job (one BAM per GPU) or of ngsqc_scan_mapping_finish (one BAM over the GPUs): SUM, and MAX for the slots that are
// not additive (max_length, paired_end, roi_bases, half_depth, yx_valid) - two small collectives on the same device buffer.
int ngsqc_comm_allreduce_counters(ngsqc_comm* c, int64_t* counters)
{
	return guarded(c, [&] {
		if (!counters) throw std::runtime_error("no counter vector");
		static const int MAXED[5] = {NGSQC_C_MAX_LENGTH, NGSQC_C_PAIRED_END, NGSQC_C_ROI_BASES, NGSQC_C_HALF_DEPTH, NGSQC_C_YX_VALID};
		int64_t mx[5]; for (int i = 0; i < 5; ++i) mx[i] = counters[MAXED[i]];
		through_device(c, counters, (size_t)NGSQC_NCOUNTERS, 64, [&](int64_t* d) {
			int64_t* dm = d + NGSQC_NCOUNTERS;
			HIPCHK(hipMemcpyAsync(dm, mx, sizeof(mx), hipMemcpyHostToDevice, c->stream));
			RCCLCHK(c, rccl().AllReduce(d, d, (size_t)NGSQC_NCOUNTERS, ncclInt64, ncclSum, c->comm, c->stream));
			RCCLCHK(c, rccl().AllReduce(dm, dm, 5, ncclInt64, ncclMax, c->comm, c->stream));
			HIPCHK(hipMemcpyAsync(mx, dm, sizeof(mx), hipMemcpyDeviceToHost, c->stream));
		});
		for (int i = 0; i < 5; ++i) counters[MAXED[i]] = mx[i];
	});
}

int ngsqc_comm_allreduce_i64(ngsqc_comm* c, int64_t* v, int64_t n, int take_max)
{
	return guarded(c, [&] {
		if (n < 0 || (n && !v)) throw std::runtime_error("invalid vector");
		if (n) through_device(c, v, (size_t)n, 0, [&](int64_t* d) { RCCLCHK(c, rccl().AllReduce(d, d, (size_t)n, ncclInt64, take_max ? ncclMax : ncclSum, c->comm, c->stream)); });
	});
}

int ngsqc_comm_allreduce_f64(ngsqc_comm* c, double* v, int64_t n)
{
	return guarded(c, [&] {
		if (n < 0 || (n && !v)) throw std::runtime_error("invalid vector");
		if (n) through_device(c, v, (size_t)n, 0, [&](double* d) { RCCLCHK(c, rccl().AllReduce(d, d, (size_t)n, ncclDouble, ncclSum, c->comm, c->stream)); });
	});
}

// every rank's shard summary, in rank order (the input of ngsqc_plan_shard_fix)
int ngsqc_comm_allgather_summaries(ngsqc_comm* c, const ngsqc_shard_summary* mine, ngsqc_shard_summary* all)
{
	return guarded(c, [&] {
		if (!mine || !all) throw std::runtime_error("no summary buffers");
		const size_t w = sizeof(ngsqc_shard_summary) / sizeof(int64_t);
		int64_t* d = (int64_t*)c->staging((size_t)(c->world + 1) * sizeof(ngsqc_shard_summary));
		HIPCHK(hipMemcpyAsync(d, mine, sizeof(*mine), hipMemcpyHostToDevice, c->stream));
		RCCLCHK(c, rccl().AllGather(d, d + w, w, ncclInt64, c->comm, c->stream));
		HIPCHK(hipMemcpyAsync(all, d + w, (size_t)c->world * sizeof(ngsqc_shard_summary), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	});
}

// the un-prefixed int32 difference array of the handle, summed over the ranks in place on the library's device memory (ngsqc_depth_device): no host copy
int ngsqc_comm_allreduce_depth(ngsqc_comm* c, ngsqc_handle* h)
{
	return guarded(c, [&] {
		void* p = nullptr; int64_t n = 0;
		if (ngsqc_depth_device(h, &p, &n) != NGSQC_OK) throw std::runtime_error(ngsqc_last_error(h));
		if (n > 0)
		{
			HIPCHK(hipDeviceSynchronize());   // (the scan that filled the array ran on the handle's streams)
			RCCLCHK(c, rccl().AllReduce(p, p, (size_t)n, ncclInt32, ncclSum, c->comm, c->stream));
			HIPCHK(hipStreamSynchronize(c->stream));
		}
	});
}

}   // extern "C"
