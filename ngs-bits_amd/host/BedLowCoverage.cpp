// BedLowCoverage / BedHighCoverage — drop-in for src/BedLowCoverage/main.cpp:16-58 and src/BedHighCoverage/main.cpp:15-51
// on the MI355X path (one source, -DHIGH_COVERAGE selects the second tool).
#include "Statistics.hpp"
using namespace ngsbits;

class ConcreteTool : public ToolBase
{
public:
	ConcreteTool(int argc, char** argv) : ToolBase(argc, argv) {}
	void setup() override
	{
#ifdef HIGH_COVERAGE
		setDescription("Detects high-coverage regions from a BAM/CRAM file.");
#else
		setDescription("Detects low-coverage regions from a BAM/CRAM file.");
#endif
		setExtendedDescription({"Note that only read start/end are used. Thus, deletions in the CIGAR string are treated as covered."});
		addInfile("bam", "Input BAM/CRAM file.", false);
		addInt("cutoff", "Minimum depth to consider a base 'high coverage'.", false);
		addInfile("in", "Input BED file containing the regions of interest. If unset, reads from STDIN.", true);
		addFlag("random_access", "Use random access via index to get reads from BAM/CRAM instead of chromosome-wise sweep. Random access is quite slow, so use it only if a small subset of the file needs to be accessed.");
		addOutfile("out", "Output BED file. If unset, writes to STDOUT.", true);
		addInt("min_mapq", "Minimum mapping quality to consider a read.", true, 1);
		addInt("min_baseq", "Minimum base quality to consider a base.", true, 0);
		addInfile("ref", "Reference genome for CRAM support (mandatory if CRAM is used).", true);
		addInt("threads", "Number of threads used.", true, 1);
		addFlag("debug", "Enable debug output.");
		// --changelog (src/BedLowCoverage/main.cpp)
#ifdef HIGH_COVERAGE
		changeLog(2024, 7, 3, "Added 'random_access' and 'debug' parameters and removed 'wgs' parameter.");
		changeLog(2022, 9, 29, "Added 'threads' parameter.");
		changeLog(2020, 11, 27, "Added CRAM support.");
		changeLog(2020, 5, 26, "Added parameter 'min_baseq'.");
		changeLog(2020, 5, 14, "First version.");
#else
		changeLog(2024, 7, 3, "Added 'random_access' and 'debug' parameters and removed 'wgs' parameter.");
		changeLog(2022, 9, 19, "Added 'threads' parameter.");
		changeLog(2020, 11, 27, "Added CRAM support.");
		changeLog(2020, 5, 26, "Added parameter 'min_baseq'.");
		changeLog(2016, 6, 9, "The BED line name of the input BED file is now passed on to the output BED file.");
#endif
	}
	void main() override
	{
		std::string in = getInfile("in"), bam = getInfile("bam");
		BedFile file; file.load(in); file.merge(true, true);
#ifdef HIGH_COVERAGE
		BedFile output = Statistics::highCoverage(file, bam, getInt("cutoff"), getInt("min_mapq"), getInt("min_baseq"), getInt("threads"), getInfile("ref"), getFlag("random_access"), getFlag("debug"));
#else
		BedFile output = Statistics::lowCoverage(file, bam, getInt("cutoff"), getInt("min_mapq"), getInt("min_baseq"), getInt("threads"), getInfile("ref"), getFlag("random_access"), getFlag("debug"));
		output.appendHeader("#BAM: " + fileName(bam));
		output.appendHeader("#ROI: " + fileName(in));
		output.appendHeader("#ROI regions: " + std::to_string(file.count()));
		output.appendHeader("#ROI bases: " + std::to_string(file.baseCount()));
#endif
		output.store(getOutfile("out"));
	}
};
int main(int argc, char** argv) { ConcreteTool tool(argc, argv); return tool.execute(); }
