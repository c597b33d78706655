// BedReadCount - drop-in for src/BedReadCount/main.cpp:21-86 on the MI355X path (same flags, defaults, header line, output). The per-read
// loop of readCount() (:33-71) runs as one scan of the BAM on the GPU (ngsqc_region_read_counts: SURVEY.md 8(f)4, a sibling of the coverage tools).
#include "Statistics.hpp"
using namespace ngsbits;

class ConcreteTool : public ToolBase
{
public:
	ConcreteTool(int argc, char** argv) : ToolBase(argc, argv) {}
	void setup() override
	{
		setDescription("Annotates the regions in a BED file with the read count from a BAM/CRAM file.");
		addInfile("bam", "Input BAM/CRAM file.", false);
		addInt("min_mapq", "Minimum mapping quality.", true, 1);
		addInfile("in", "Input BED file (note that overlapping regions will be merged before processing). If unset, reads from STDIN.", true);
		addOutfile("out", "Output BED file. If unset, writes to STDOUT.", true);
		addInfile("ref", "Reference genome for CRAM support (mandatory if CRAM is used).", true);
	}
	void readCount(BedFile& bed_file, const std::string& bam_file, int min_mapq, const std::string& ref_file)
	{
		if (!bed_file.isMergedAndSorted()) NB_THROW(ArgumentException, "Merged and sorted BED file required for coverage calculation!");
		BamReader reader(bam_file, ref_file);
		// lines on chromosomes the BAM does not know never match a read (ChromosomalIndex lookup by name): they keep a count of 0
		std::vector<ngsqc_region> regions; std::vector<long long> line_of;
		for (long long i = 0; i < bed_file.count(); ++i)
		{
			int tid = reader.chromosomeID(bed_file[i].chr());
			if (tid < 0) continue;
			regions.push_back(ngsqc_region{tid, bed_file[i].start(), bed_file[i].end()}); line_of.push_back(i);
		}
		std::vector<int64_t> read_count((size_t)bed_file.count(), 0), got(regions.size(), 0);
		if (!regions.empty()) reader.check(ngsqc_region_read_counts(reader.handle(), regions.data(), (int64_t)regions.size(), min_mapq, got.data()));
		for (size_t k = 0; k < regions.size(); ++k) read_count[(size_t)line_of[k]] = got[k];
		for (long long i = 0; i < bed_file.count(); ++i) bed_file[i].annotations().push_back(std::to_string(read_count[(size_t)i]));
		// --changelog (src/BedReadCount/main.cpp)
		changeLog(2020, 11, 27, "Added CRAM support.");
	}
	void main() override
	{
		BedFile file; file.load(getInfile("in"));
		file.merge(false);
		std::string bam = getInfile("bam");
		readCount(file, bam, getInt("min_mapq"), getInfile("ref"));
		file.clearHeaders();
		file.appendHeader("#chr\tstart\tend\t" + baseName(bam));
		file.store(getOutfile("out"));
	}
};
int main(int argc, char** argv) { ConcreteTool tool(argc, argv); return tool.execute(); }
