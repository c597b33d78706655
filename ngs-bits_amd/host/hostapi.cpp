// A small C entry into the host layer for bench.py and the C-ABI tests: the region tables they pass to libngsqc_hip.so come from the PRODUCT's
// BedFile / Chromosome code (load, sort, merge, chunk; BedFile.cpp / Chromosome.cpp of the reference), not from the oracle's BED loader - so the measured path and
// the parity tests exercise host/core.cpp as well. Built as bin/libngsqc_hostapi.so; not part of the C ABI of include/ngsqc.h.
#include "core.hpp"
#include <cstring>
using namespace ngsbits;

extern "C" {

// mode: 0 as loaded, 1 merge(), 2 merge(true, true), 3 sort + merge, 4 merge + chunk(100), 5 sort + merge + chunk(100)  (the operations the reference applies to its ROI:
// Statistics.cpp:352-387, 1000-1010; BedCoverage / BedLowCoverage main.cpp). ref_names: the BAM's references in tid order. out: rows (tid or -1, start 1-based, end).
// Returns the number of lines (also when it exceeds cap: call again with a larger buffer), -1 on an error (message in err).
long long ngsbits_bed_regions(const char* bed_path, const char* const* ref_names, int n_ref, int mode, int* out, long long cap, char* err, int err_cap)
{
	try
	{
		BedFile bed; bed.load(bed_path, false);
		if (mode == 3 || mode == 5) bed.sort();
		if (mode == 1 || mode == 3 || mode == 4 || mode == 5) bed.merge();
		if (mode == 2) bed.merge(true, true);
		if (mode == 4 || mode == 5) bed.chunk(100);
		std::vector<std::pair<int, int>> tid_of;   // Chromosome::num() -> first tid with that number
		for (int t = 0; t < n_ref; ++t) { const int num = Chromosome(ref_names[t]).num(); bool have = false; for (auto& p : tid_of) have = have || p.first == num; if (!have) tid_of.emplace_back(num, t); }
		for (long long i = 0; i < bed.count() && i < cap; ++i)
		{
			int tid = -1; for (auto& p : tid_of) if (p.first == bed[i].chr().num()) tid = p.second;
			out[3 * i] = tid; out[3 * i + 1] = bed[i].start(); out[3 * i + 2] = bed[i].end();
		}
		return bed.count();
	}
	catch (std::exception& e) { if (err && err_cap > 0) { strncpy(err, e.what(), (size_t)err_cap - 1); err[err_cap - 1] = 0; } return -1; }
}

}
