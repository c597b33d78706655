// BedCoverage — drop-in for src/BedCoverage/main.cpp:16-67 on the MI355X path (same flags, defaults, header line, output).
#include "Statistics.hpp"
using namespace ngsbits;

class ConcreteTool : public ToolBase
{
public:
	ConcreteTool(int argc, char** argv) : ToolBase(argc, argv) {}
	void setup() override
	{
		setDescription("Annotates a BED file with the average coverage of the regions from one or several BAM/CRAM file(s).");
		addInfileList("bam", "Input BAM/CRAM file(s).", false);
		addInt("min_mapq", "Minimum mapping quality.", true, 1);
		addInfile("in", "Input BED file. If unset, reads from STDIN.", true);
		addInt("decimals", "Number of decimals used in output.", true, 2);
		addOutfile("out", "Output BED file. If unset, writes to STDOUT.", true);
		addInfile("ref", "Reference genome for CRAM support (mandatory if CRAM is used).", true);
		addFlag("clear", "Clear previous annotation columns before annotating (starting from 4th column).");
		addInt("threads", "Number of threads used.", true, 1);
		addFlag("random_access", "Use random access via index to get reads from BAM/CRAM instead of chromosome-wise sweep. Random access is quite slow, especially on CRAM, so use it only if a small subset of the file needs to be accessed.");
		addFlag("debug", "Enable debug output.");
		addFlag("skip_mismapped", "Skip reads with mapping quality less than 20 that are not properly paired (they are often mis-mapped).");
		// --changelog (src/BedCoverage/main.cpp)
		changeLog(2025, 9, 15, "Added 'skip_mismapped' parameter.");
		changeLog(2024, 6, 26, "Added 'random_access' parameter.");
		changeLog(2022, 9, 16, "Added 'threads' parameter and removed 'dup' parameter.");
		changeLog(2022, 8, 12, "Added parameter to clear previous annotation columns.");
		changeLog(2022, 8, 9, "Removed mode parameter (panel mode is always used now).");
		changeLog(2020, 11, 27, "Added CRAM support.");
		changeLog(2017, 6, 2, "Added 'dup' parameter.");
	}
	void main() override
	{
		BedFile file; file.load(getInfile("in"));
		if (getFlag("clear")) { file.clearHeaders(); file.clearAnnotations(); }
		std::string header = "#chr\tstart\tend";
		for (const std::string& bam : getInfileList("bam"))
		{
			Statistics::avgCoverage(file, bam, getInt("min_mapq"), getInt("threads"), getInt("decimals"), getInfile("ref"), getFlag("random_access"), getFlag("skip_mismapped"), getFlag("debug"));
			header += "\t" + baseName(bam);
		}
		file.appendHeader(header);
		file.store(getOutfile("out"));
	}
};
int main(int argc, char** argv) { ConcreteTool tool(argc, argv); return tool.execute(); }
